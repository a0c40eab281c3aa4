import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
import exavatar_release_amd as exa
from exavatar_release_amd import rasterizer as rz
dev = torch.device('cuda:0')
for rep in range(3):
    s0 = torch.cuda.memory_stats(); s0.setdefault('segment.all.allocated', s0.get('num_device_alloc', 0))
    t0 = time.perf_counter()
    r = bench.iteration_throughput(dev, iters=40)
    s1 = torch.cuda.memory_stats(); s1.setdefault('segment.all.allocated', s1.get('num_device_alloc', 0))
    print(rep, {k: (round(v['ms_per_iteration'], 3), round(v['host_ms_per_iteration'], 3)) for k, v in r.items() if isinstance(v, dict)},
          'segments +%d' % (s1['segment.all.allocated'] - s0['segment.all.allocated']), 'pending', len(rz._pending),
          'events', len(rz.overflow_events), 'reserved MB', s1['reserved_bytes.all.current'] >> 20, 'wall %.1f s' % (time.perf_counter() - t0))
