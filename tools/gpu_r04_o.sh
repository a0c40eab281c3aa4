#!/bin/bash
# Round-4 check O (experiment): what a composite's instance buffer sized below capacity_a + capacity_b would buy
# (its backward launches one wave per 64-slot of that buffer; the zero-fill of compose_kernel scales with it too).
R=$GRAFT_REPO_ROOT; cd $R
python - <<'PY'
import torch, exavatar_release_amd as exa
from exavatar_release_amd import scenes, rasterizer as rz
dev = torch.device('cuda:0'); H = W = 1024
scene = {k: v.to(dev) for k, v in scenes.dist_c_scene(100_000, H, W, seed=1).items()}
human = {k: v.to(dev) for k, v in scenes.dist_b_avatar(50_000, seed=2).items()}
cam = {k: t.to(dev) for k, t in scenes.ring_camera(H, W, 7, 200).items()}
exa.config.mode = 'exact'
rend = exa.GaussianRenderer()
with torch.no_grad():
    res = exa.render_iteration(rend, scene, human, human, (H, W), cam, torch.rand(3, device=dev))
torch.cuda.synchronize()
print('needs (64-slot space): scene %d human %d' % (rz._seen_D[(0, 100000, H, W)], rz._seen_D[(0, 50000, H, W)]))
PY
for sc in 1.0 0.5 0.3; do
  for how in graphed sets; do
    echo -n "scale $sc: "; EXA_COMPOSE_CAP_SCALE=$sc timeout 200 python tools/gpu_iteration_profile.py $how 300 2>&1 | tail -1
  done
done
cd /tmp && export TMPDIR=/tmp
for sc in 1.0 0.3; do
  rm -rf /tmp/prof_o; EXA_COMPOSE_CAP_SCALE=$sc timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o -- python $R/tools/gpu_iteration_profile.py graphed 60 > /dev/null 2>&1
  f=$(find /tmp/prof_o -name '*kernel_stats.csv' | head -1)
  echo "scale $sc"; python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if any(k in r['Name'] for k in ('render_bwd', 'compose_kernel', 'merge_kernel', 'render_fwd', 'preprocess_bwd')):
        print('   %-80s %5s %8.1f' % (r['Name'][:80], r['Calls'], float(r['AverageNs']) / 1e3))
PY
done
