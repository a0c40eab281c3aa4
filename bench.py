#!/usr/bin/env python3
"""Benchmark of the hot path BASELINE.json names: forward + backward differentiable Gaussian
rasterize at 1024x1024 with ~150 k avatar-like Gaussians (config C3), view-sharded over N GPUs.

    python bench.py --gpus N --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one rasterizer forward + its backward for one training view with a dense
dL/dimage (inputs already resident in HBM), followed for N > 1 by the RCCL all-reduce of the
Gaussian gradients (14 floats x P = 8.4 MB) through the product's `dist.FlatGradAllReducer`.
How the step is issued (`--launch`): `abi` (default) -- plain kernel launches straight through the C ABI
(`exa_raster_forward_batch` + `exa_raster_backward_batch`) from the product's `StaticRender`: static buffers, the jobs of every
view marshalled once and reading their camera in place from the resident table, two ctypes calls per step, every forward polling
its own overflow report (and re-rendering in place if it had to), gradients written directly into the all-reducer's flat buffers;
`surface` (= `eager`) -- the drop-in autograd surface call by call: `GaussianRasterizer(settings)(...)` + `torch.autograd.grad`
through the compiled autograd node.  The default line ALWAYS carries the surface next to the ABI headline
(`extra_plugin_surface_eager`, also for N > 1).  `--views-in-flight S`: S render slots of the `StaticRender` on S HIP streams, a
step = a group of S views whose gradients are accumulated in slot order (N > 1: one all-reduce per group).  Views: the 200
ring cameras of config C4 dealt by the product's `dist.shard_views` in a fixed stratified order (step i -> view
(i * 123) mod 200: a short run covers the ring like a long one); every rank cycles through its shard, so
per-GPU work is fixed as N grows (weak scaling) and `value` = views rasterized fwd+bwd per second over all ranks.

Other workloads: `--config c2|c1` (same step), `--config c5` = BASELINE configs[4]: 300 k Gaussians
(200 k avatar + 100 k scene), in-kernel SH degree 3, 2048x2048, FORWARD ONLY (no backward context stored), same launch
protocols -- the animation / inference use case of avatar/main/animate.py:64-66.

Prints ONE JSON line on rank 0 (contract in the task statement) with extra objects: `roofline` (dominant kernel,
HIP-event timed inside this script; `traffic` / `secondary` / `rocprof` are quoted from committed profiles ONLY when those were
measured on the build that is loaded -- digest of the library sources -- and are null with the reason otherwise),
`cpu_baseline` (the CPU oracle timed on a bounded sample of the same workload, rank 0 at N = 1 only),
`extra_plugin_surface_eager`, `extra_abi_views_in_flight` (four views in flight through the C ABI), `extra_batched_views`
(K views per batched launch, next to -- never instead of -- the single-view headline) and `extra_exavatar_iteration`
(the five same-camera renders of one ExAvatar training sample, eager, three ways); for N > 1 every rank's own step time with
and without the exchange (`rccl.ranks`).
"""
import argparse
import ctypes
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

HBM_PEAK_GBS = 8000.0           # MI355X HBM3E spec peak (MI355X_MICROARCH.md)
N_VIEWS = 200
VIEW_STRIDE = 123      # step i renders ring view (i * VIEW_STRIDE) mod N_VIEWS (coprime: a permutation)
PROFILE_PREFIXES = ('r06', 'r05', 'r04', 'r03', 'r02')   # profiles/<prefix>_hbm_traffic.json feeds roofline.traffic (newest first)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=200)
    ap.add_argument('--warmup', type=int, default=20)
    ap.add_argument('--config', default='c3', choices=['c3', 'c2', 'c1', 'c5'])
    ap.add_argument('--mode', default=None, choices=['train', 'forward'],
                    help='train = forward + backward (default; c5 defaults to forward)')
    ap.add_argument('--launch', default='abi', choices=['abi', 'surface', 'eager'],
                    help="how a step is issued: 'abi' (default) plain launches straight through the C ABI from pre-marshalled jobs and "
                         "static buffers (exavatar_release_amd.StaticRender); 'surface' (= 'eager') the drop-in autograd surface call by "
                         "call: GaussianRasterizer(settings)(...) + torch.autograd.grad through the compiled autograd node (the default "
                         "line carries it as extra_plugin_surface_eager)")
    ap.add_argument('--views-in-flight', type=int, default=1,
                    help="'abi' only: S render slots of the StaticRender, each on its own HIP stream; a step is then a GROUP of S views "
                         "of this rank's shard whose gradients are accumulated in slot order into one set of arrays (N > 1: one "
                         "all-reduce per group); 1 = the headline, one view at a time")
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-concurrent', action='store_true', help='skip the extras that run next to the headline')
    ap.add_argument('--no-kernel-timing', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the extra_c5_forward / extra_c2 child runs')
    return ap.parse_args()


def build_scene(name):
    from exavatar_release_amd import scenes
    if name == 'c3':
        return scenes.dist_b_avatar(150_000, seed=0), (1024, 1024), 'c3: 150k avatar-like Gaussians (Dist-B), 1024x1024'
    if name == 'c2':
        return scenes.dist_b_avatar(120_000, seed=0), (960, 540), 'c2: 120k avatar-like Gaussians (Dist-B), 540x960 (WxH)'
    if name == 'c5':
        a = scenes.make_config('c5')[0]
        a['sh'] = scenes.sh_from_rgb(a['rgb'], 3, seed=5, rest_sigma=0.1)
        return a, (2048, 2048), 'c5: 200k avatar-like + 100k scene Gaussians, SH degree 3 in-kernel, 2048x2048, forward only'
    return scenes.dist_a_random(10_000, 256, 256, seed=0), (256, 256), 'c1: 10k random Gaussians (Dist-A), 256x256'


def view_settings(k, shape, cfg_name):
    """Raster matrices of ring view k, built like GaussianRenderer.forward does."""
    from exavatar_release_amd import scenes
    from exavatar_release_amd.camera import make_raster_matrices
    H, W = shape
    focal = 1500.0 * (H / 1024.0) if cfg_name != 'c2' else 1500.0 * 960 / 1024
    if cfg_name == 'c1':
        cam = scenes.neutral_camera(H, W)
    else:
        cam = scenes.ring_camera(H, W, k, N_VIEWS, focal=focal)
    tanx, tany, view, proj, campos = make_raster_matrices(cam, shape)
    return dict(tanfovx=tanx, tanfovy=tany, view=view, proj=proj, campos=campos)


def main():
    args = parse()
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: become the torch.distributed.run launch of N ranks (one per GPU, RCCL over
        # xGMI; EXA_BENCH_BACKEND=gloo lets several ranks share one GPU for the smoke test of this very path)
        n_dev = torch.cuda.device_count()
        if n_dev < args.gpus and os.environ.get('EXA_BENCH_BACKEND', 'nccl') == 'nccl':
            print('bench.py: --gpus %d but only %d device(s) visible (RCCL needs one GPU per rank)' % (args.gpus, n_dev),
                  file=sys.stderr)
            sys.exit(2)
        port = os.environ.get('MASTER_PORT') or str(29500 + os.getpid() % 2000)
        cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
               '--master-addr', '127.0.0.1', '--master-port', port, os.path.abspath(__file__)] + sys.argv[1:]
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        os.execv(sys.executable, cmd)
    rank = int(os.environ.get('RANK', 0))
    world = int(os.environ.get('WORLD_SIZE', 1))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    if world != args.gpus:
        print('bench.py: --gpus %d but WORLD_SIZE=%d' % (args.gpus, world), file=sys.stderr)
        sys.exit(2)
    local_rank = local_rank % max(torch.cuda.device_count(), 1)
    torch.cuda.set_device(local_rank)
    device = torch.device('cuda', local_rank)
    import torch.distributed as dist
    if world > 1:
        # RCCL ("nccl") in production; EXA_BENCH_BACKEND=gloo lets the multi-rank code path be exercised with several
        # ranks sharing one GPU (RCCL refuses duplicate devices), which is how it is smoke-tested on a 1-GPU box
        backend = os.environ.get('EXA_BENCH_BACKEND', 'nccl')
        if backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group(backend)

    import exavatar_release_amd as exa
    from exavatar_release_amd import _lib
    from exavatar_release_amd import dist as exa_dist
    from exavatar_release_amd.rasterizer import (GaussianRasterizationSettings, GaussianRasterizer, rasterize_gaussians,
                                                 rasterize_gaussians_batch)

    mode = args.mode or ('forward' if args.config == 'c5' else 'train')
    train = mode == 'train'
    assets, shape, workload = build_scene(args.config)
    H, W = shape
    P = assets['mean_3d'].shape[0]
    use_sh = 'sh' in assets
    sh_degree = 3 if use_sh else 0
    launch = 'surface' if args.launch == 'eager' else args.launch
    S = max(1, args.views_in_flight)
    if S > 1 and (launch != 'abi' or not train):
        raise SystemExit("bench.py: --views-in-flight > 1 needs --launch abi and --mode train")

    # ---- parameters: contiguous leaves; for N > 1 their gradients travel in ONE flat buffer ----
    names = ('mean_3d', 'scale', 'rotation', 'opacity') + (('sh',) if use_sh else ('rgb',))
    params = [assets[k].to(device).contiguous().requires_grad_(train) for k in names]
    n_float = sum(p.numel() for p in params)

    g = torch.Generator().manual_seed(1)
    dL_dimg = torch.randn(3, H, W, generator=g).to(device)
    bg = torch.ones(3, device=device)

    # ---- this rank's shard of the ring views (the product's dealer) -------------------------------------------
    # A fixed, STRATIFIED order so that runs are comparable and a short run is representative: step i takes ring view
    # (i * 123) mod 200 (123 ~ 200 / golden ratio, coprime to 200), so any window of a dozen steps covers the ring
    # evenly.  In ring order the 20 steps of a `--steps 20 --warmup 5` run were views 5..24 -- a 36-degree arc of frontal
    # (heavy) views, 6-7 % slower than the mean over the ring that 200+ steps measure; frontal and side views of
    # an avatar differ by ~35 % in instances.  Training itself shuffles (reference DataLoader shuffle=True).
    ring_order = os.environ.get('EXA_BENCH_VIEW_ORDER', 'stratified') == 'ring'      # rounds 1-2 protocol (value_no_settle)
    view_order = list(range(N_VIEWS)) if ring_order else [(i * VIEW_STRIDE) % N_VIEWS for i in range(N_VIEWS)]
    my_views = exa_dist.shard_views(N_VIEWS, rank, world, order=view_order) if args.config != 'c1' else [0]
    vs = [view_settings(k, shape, args.config) for k in my_views]
    n_my = len(my_views)
    # the views are RESIDENT: one 48-float row per view -- viewmatrix (16) | projmatrix (16) | campos (3) | pad -- and every view's
    # settings point INTO its row: no camera is copied per step by any protocol of this script
    cam_tab = torch.zeros(len(vs), 48)
    for j, v in enumerate(vs):
        cam_tab[j, 0:16] = v['view'].reshape(-1).cpu()
        cam_tab[j, 16:32] = v['proj'].reshape(-1).cpu()
        cam_tab[j, 32:35] = v['campos'].reshape(-1).cpu()
    cam_tab = cam_tab.to(device)

    def settings_of(table, j, v):
        return GaussianRasterizationSettings(
            image_height=H, image_width=W, tanfovx=v['tanfovx'], tanfovy=v['tanfovy'], bg=bg, scale_modifier=1.0,
            viewmatrix=table[j, 0:16].view(4, 4), projmatrix=table[j, 16:32].view(4, 4), sh_degree=sh_degree,
            campos=table[j, 32:35], prefiltered=False, debug=False)
    view_st = [settings_of(cam_tab, j, v) for j, v in enumerate(vs)]

    def make_ctx(kv=1):
        """A launch context of the extras that replay K views per batched call from a hipGraph (batched_throughput) and of the
        per-kernel timing: kv camera slots (static tensors a captured graph reads), settings, mean_2d probes."""
        c = {'cam': cam_tab[:kv].clone()}
        c['settings'] = [settings_of(c['cam'], i, vs[0]) for i in range(kv)]
        c['mean_2d'] = [torch.zeros(P, 3, device=device, requires_grad=train) for _ in range(kv)]
        c['stream'] = None
        c['graph'] = None
        c['kv'] = kv
        return c

    def set_view(i, c):
        """Point context c at views i .. i + kv - 1 of this rank's shard (ONE gather-copy kernel)."""
        if c['kv'] == 1:
            j = i % n_my
            torch.mul(cam_tab[j:j + 1], 1.0, out=c['cam'])                  # one elementwise kernel
        else:
            idx = torch.arange(i * c['kv'], (i + 1) * c['kv'], device=device) % n_my
            torch.index_select(cam_tab, 0, idx, out=c['cam'])

    def raster_step(c):
        """forward (+ backward) of the rasterizer for the kv views of context c through the autograd surface."""
        m3, sc, rot, op, col = params
        kw = dict(shs=col, colors_precomp=None) if use_sh else dict(shs=None, colors_precomp=col)
        if c['kv'] == 1:
            if train:
                color, radii, depth, alpha = rasterize_gaussians(m3, c['mean_2d'][0], kw['shs'], kw['colors_precomp'], op, sc,
                                                                 rot, None, c['settings'][0])
                torch.autograd.grad([color], params + c['mean_2d'], grad_outputs=[dL_dimg])
            else:
                with torch.no_grad():
                    rasterize_gaussians(m3, c['mean_2d'][0], kw['shs'], kw['colors_precomp'], op, sc, rot, None,
                                        c['settings'][0])
            return
        jobs = [dict(means3D=m3, means2D=c['mean_2d'][i], opacities=op, scales=sc, rotations=rot, cov3D_precomp=None,
                     raster_settings=c['settings'][i], **kw) for i in range(c['kv'])]
        if train:
            outs = rasterize_gaussians_batch(jobs)
            torch.autograd.grad([o[0] for o in outs], params + c['mean_2d'], grad_outputs=[dL_dimg] * c['kv'])
        else:
            with torch.no_grad():
                rasterize_gaussians_batch(jobs)

    # ---- calibrate the instance-buffer capacity over this rank's views (stage 1 through the C ABI, untimed) -------
    lib = _lib.load()
    from exavatar_release_amd.rasterizer import _make_settings, _ptr, _stream_ptr, read_header
    sz = _lib.workspace_sizes(P, W, H, 0)
    geom = torch.empty(int(sz.geom_bytes), dtype=torch.uint8, device=device)
    tile = torch.empty(int(sz.tile_bytes), dtype=torch.uint8, device=device)
    radii = torch.empty(P, dtype=torch.int32, device=device)
    D_list, V_list, I_list, T_list = [], [], [], []
    for i in range(n_my):        # every view of the shard: the capacity below provably covers them
        keep = []
        st = _make_settings(view_st[i], device, keep)
        m3, sc, rot, op, col = [t.detach() for t in params]
        _lib.check(lib.exa_raster_forward_bin(ctypes.byref(st), P, 16 if use_sh else 0, _ptr(m3), _ptr(col) if use_sh else None,
                                              None if use_sh else _ptr(col), _ptr(op), _ptr(sc), _ptr(rot), None, _ptr(radii),
                                              _ptr(geom), _ptr(tile), _stream_ptr(device)))
        hdr = read_header(tile)
        D_list.append(hdr[0])        # capacity this view needs (64 * batch slots)
        V_list.append(hdr[3])
        I_list.append(hdr[4])        # 8x8 sub-tile instances (rect count, before the exact footprint test)
        T_list.append(hdr[6])        # 16x16 tile instances = upstream's num_rendered = the D of SURVEY.md 8(d)
    del geom, tile
    D_max, I_mean, V_mean = max(D_list), sum(I_list) / len(I_list), sum(V_list) / len(V_list)
    D_mean = sum(T_list) / len(T_list)
    exa.config.mode = 'capacity'
    exa.config.fixed_capacity = int(D_max) + 64      # every view of the shard was probed: D_max is exact (and every render checks itself)

    # N > 1: gradients go through the product's double-buffered flat all-reducer: the all-reduce of step i overlaps the
    # rasterize of step i + 1 (buffer i & 1); 'abi' writes its gradients straight into the reducer's buffers
    reducer = exa_dist.FlatGradAllReducer(params, average=False, n_buffers=2) if world > 1 else None
    wait_host = [0.0]

    def reducer_wait(k):
        if reducer is not None:
            t_w = time.perf_counter()
            reducer.wait(k)                 # the all-reduce that last read this gradient buffer (two steps ago)
            wait_host[0] += time.perf_counter() - t_w

    # ---- 'abi': the step straight through the C ABI (exavatar_release_amd/static.py) ----------------------------
    # One StaticRender: static inputs, workspaces, images, gradient arrays; the forward / backward jobs of every view of the
    # shard marshalled once, their settings pointing INTO the resident camera table; a step = two ctypes calls = seven kernel
    # launches.  Every forward polls its own zero-copy header report and would repair an overflow in place (on_overflow='repair').
    sr = None
    if launch == 'abi':
        m3, sc, rot, op, col = [t.detach() for t in params]
        sr = exa.StaticRender(m3, op, sc, rot, colors_precomp=None if use_sh else col, shs=col if use_sh else None,
                              image_size=(H, W), capacity=int(D_max) + 64, train=train, slots=S,
                              on_overflow=os.environ.get('EXA_BENCH_OVERFLOW', 'repair'))
        for st_j in view_st:
            sr.add_view(st_j, dL_dcolor=dL_dimg if train else None)
        if train:
            for b in range(2 if (reducer is not None or S > 1) else 1):
                if reducer is not None:
                    bv = reducer.buffer_views(b)       # order of `params`: means3D, scales, rotations, opacities, colour | SH
                    sr.add_grad_outputs(means3D=bv[0], scales=bv[1], rotations=bv[2], opacities=bv[3],
                                        **({'shs': bv[4]} if use_sh else {'colors_precomp': bv[4]}))
                else:
                    sr.add_grad_outputs()

    def step_abi(i, reduce=True):
        k = i & 1
        if reduce:
            reducer_wait(k)
        sr.forward(i % n_my)
        if train:
            sr.backward(k if reducer is not None else 0)
        if reduce and reducer is not None:
            reducer.reduce(k)

    chain_done = [None, None]

    def step_abi_group(i, reduce=True):
        """S views of the shard in flight; their gradients accumulated in slot order into ONE of two alternating sets of arrays
        (N > 1: the all-reducer's two buffers, one all-reduce per group).  No barrier between groups: slot 0, which overwrites a
        set, waits for whatever last read it -- the all-reduce of two groups ago, or that group's chain end -- and the consumer
        of the sum (the all-reduce) is queued on the LAST slot's stream; everything else runs ahead into the next group."""
        k = i & 1
        if reduce and reducer is not None:
            reducer_wait(k)                               # (slot 0's stream is the current stream)
        elif chain_done[k] is not None:
            sr.slot_stream(0).wait_event(chain_done[k])
        for s in range(S):
            sr.forward((i * S + s) % n_my, slot=s)
            sr.backward(k, slot=s, accumulate=s > 0, after=s - 1 if s else None)
        if reduce and reducer is not None:
            with torch.cuda.stream(sr.slot_stream(S - 1)):
                reducer.reduce(k)
        else:
            chain_done[k] = torch.cuda.Event()
            chain_done[k].record(sr.slot_stream(S - 1))

    # ---- 'surface': the drop-in autograd surface call by call ----------------------------------------------------
    # One GaussianRasterizer module per view (the reference builds one per render, module.py:623: ~10 us of nn.Module
    # construction that this loop does not pay), its settings reading the camera in place; torch.autograd.grad as the backward.
    rasts = [GaussianRasterizer(st_j) for st_j in view_st]
    probe = torch.zeros(P, 3, device=device, requires_grad=train)
    surf_in = params + [probe]
    kw_col = dict(shs=params[4]) if use_sh else dict(colors_precomp=params[4])

    def step_surface(i, reduce=True):
        k = i & 1
        if reduce:
            reducer_wait(k)
        if train:
            color = rasts[i % n_my](means3D=params[0], means2D=probe, opacities=params[3], scales=params[1], rotations=params[2],
                                    **kw_col)[0]
            grads = torch.autograd.grad([color], surf_in, grad_outputs=[dL_dimg])
            if reduce and reducer is not None:
                reducer.pack(grads[:5], k)
                reducer.reduce(k)
        else:
            with torch.no_grad():
                rasts[i % n_my](means3D=params[0], means2D=probe, opacities=params[3], scales=params[1], rotations=params[2], **kw_col)

    step = step_surface if launch == 'surface' else step_abi_group if S > 1 else step_abi
    units = S if (launch == 'abi' and S > 1) else 1          # views per step

    def finish():
        if reducer is not None:
            reducer.finish()

    def timed(fn, n, w, settle_n=0):
        """settle_n + w untimed steps, then n steps between synchronisations; seconds."""
        for i in range(settle_n):
            fn(i)
        finish()
        torch.cuda.synchronize()
        for i in range(w):
            fn(i)
        finish()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t_0 = time.perf_counter()
        for i in range(n):
            fn(w + i)
        finish()
        torch.cuda.synchronize()
        return time.perf_counter() - t_0

    # Settle phase (setup, untimed, reported as config.settle_steps): the calibration above is a stop-and-go of 200 tiny
    # launches with a host read-back each, which leaves the GPU in a low power state; the first ~50 steps after it run
    # 4-7 % slower than the steady state that any training run is in.  The W warm-up steps asked for follow it.
    settle = int(os.environ.get('EXA_BENCH_SETTLE_STEPS', '300'))
    wait_host[0] = 0.0
    elapsed_local = timed(step, args.steps, args.warmup, settle)
    if world > 1:
        dist.barrier()
    # (the clock of the contract: barrier + synchronize on both sides of exactly K steps -- `timed` synchronises before it
    #  starts its clock behind a barrier and stops it after finish() + synchronize; the closing barrier above is outside it,
    #  the MAX over ranks below covers a rank that finished late)
    t = torch.tensor([elapsed_local], device=device, dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    wait_us = wait_host[0] / max(args.steps + args.warmup + settle, 1) * 1e6
    if sr is not None:
        sr.check()

    # what the host spends per step (queueing only: steps issued back to back onto an idle device, clock stopped before the
    # device is waited for; 'repair' polls every forward's report, i.e. includes waiting for the scatter stage of each render)
    host_us = None
    if world == 1:
        torch.cuda.synchronize()
        th = time.perf_counter()
        for i in range(32):
            step(i)
        host_us = (time.perf_counter() - th) / 32 * 1e6
        torch.cuda.synchronize()

    # N > 1: the same steps WITHOUT the exchange, per rank -- what the all-reduce costs this rank's step when it is not hidden
    local_ms = None
    if world > 1 and train:
        local_ms = timed(lambda i: step(i, reduce=False), max(args.steps // 2, 4), 2) / max(args.steps // 2, 4) * 1e3
        dist.barrier()

    # every rank reports what ITS communicator says and what ITS steps took: a SCALE line that disappoints can be read against it
    rank_info = {'rank': rank, 'world_size': dist.get_world_size() if world > 1 else 1, 'device': torch.cuda.current_device(),
                 'device_name': torch.cuda.get_device_name(device), 'views': n_my,
                 'step_ms': elapsed_local / args.steps * 1e3, 'step_ms_without_allreduce': local_ms,
                 'allreduce_wait_host_us_per_step': wait_us if world > 1 else None}
    ranks = [rank_info]
    if world > 1:
        ranks = [None] * world
        dist.all_gather_object(ranks, rank_info)
    result = None
    if rank == 0:
        ms_per_step = elapsed / args.steps * 1e3
        value = world * args.steps * units / elapsed
        if not train:
            metric = 'forward renders/sec at %dx%d / %dk Gaussians%s' % (W, H, P // 1000, ', SH deg 3 (BASELINE configs[4])' if use_sh else '')
        elif args.config == 'c3':
            metric = 'train iters/sec (fwd+bwd raster) at 1024x1024 / ~150k Gaussians'
        else:
            metric = 'train iters/sec (fwd+bwd raster) at %dx%d / ~%dk Gaussians (%s)' % (W, H, P // 1000, args.config)
        result = {
            'metric': metric,
            'value': value, 'unit': 'iters/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': ms_per_step, 'higher_is_better': True, 'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32', 'data': 'synthetic',
            'config': {'workload': workload + ', %d ring views dealt by dist.shard_views' % N_VIEWS,
                       'view_order': 'ring order (step i -> view i)' if ring_order else
                                     'step i -> ring view (i * %d) mod %d (stratified: every window of steps covers the ring)' % (VIEW_STRIDE, N_VIEWS),
                       'parity_bar': 'image 1e-4 L-inf off the <= 0.04 % threshold-adjacent (ambiguous) pixels, which may be off by '
                                     '2e-2; grads 1e-3 rel, 1e-1 * mean floor for Gaussians under an ambiguous pixel (tests/helpers.py)',
                       'P': P, 'W': W, 'H': H, 'mode': 'fwd+bwd' if train else 'forward only (no_grad)',
                       'views_per_rank': n_my, 'launch': launch, 'settle_steps': settle, 'host_us_per_step': host_us,
                       'launch_what': {'abi': 'plain kernel launches straight through the C ABI: exa_raster_forward_batch + '
                                              'exa_raster_backward_batch on pre-marshalled jobs and static buffers '
                                              '(exavatar_release_amd.StaticRender), two ctypes calls = eight kernels per step '
                                              '(forward: preprocess_fwd, cell_scatter, subtile_count, subtile_bin, sort_subtiles, '
                                              'render_fwd; backward: render_bwd, preprocess_bwd); every forward polls its own '
                                              'overflow report inside the timed region and would re-render in place (on_overflow=%s; '
                                              'repairs during this run: %d)' % (sr.on_overflow if sr else '-', sr.repairs if sr else 0),
                                       'surface': 'the drop-in autograd surface call by call: GaussianRasterizer(settings)(...) + '
                                                  'torch.autograd.grad through the compiled autograd node (csrc/torch_binding.cpp); one '
                                                  'module per view, built once'}[launch],
                       'view_switch': 'per step, inside the timed region: every view\'s settings read its camera block (48 floats) IN PLACE '
                                      'from the resident table of ring views (no copy)',
                       'views_in_flight_per_gpu': S, 'views_per_step': units,
                       'parallelism': 'view-sharded dp%d, RCCL all-reduce of %d B grads (dist.FlatGradAllReducer)%s'
                                      % (world, n_float * 4, ', one per group of %d views accumulated in slot order' % S if S > 1 else ''),
                       'mean_instances_D': D_mean, 'mean_subtile_instances': I_mean, 'mean_visible_V': V_mean},
            'rccl': {'world_size': dist.get_world_size() if world > 1 else 1,
                     'backend': (dist.get_backend() + (' (RCCL)' if dist.get_backend() == 'nccl' else '')) if world > 1 else None,
                     'ranks': ranks},
        }

    single = S == 1 and world == 1
    # ---- the drop-in autograd surface next to the headline: ALWAYS (rank 0, its own shard, no exchange) -----------------
    if rank == 0 and train and launch == 'abi':
        try:
            result['extra_plugin_surface_eager'] = surface_throughput(step_surface, args, exa)
        except Exception as e:  # noqa: BLE001 -- an extra must never take the headline down
            result['extra_plugin_surface_eager'] = {'error': str(e)[:200]}

    # ---- extras (not the headline): views in flight through the C ABI; K views per batched launch -------
    if rank == 0 and single and launch == 'abi' and train and not args.no_concurrent and args.config != 'c1':
        try:
            result['extra_abi_views_in_flight'] = abi_views_in_flight(4, args, exa, params, use_sh, shape, D_max, view_st, dL_dimg)
        except Exception as e:  # noqa: BLE001
            result['extra_abi_views_in_flight'] = {'error': str(e)[:200]}
        for name, fn in (('extra_batched_views', lambda: batched_throughput(8, 1, args, make_ctx, set_view, raster_step)),
                         ('extra_batched_views_x2', lambda: batched_throughput(8, 2, args, make_ctx, set_view, raster_step))):
            try:
                result[name] = fn()
            except Exception as e:  # noqa: BLE001
                result[name] = {'error': str(e)[:200]}

    # ---- extra: one ExAvatar training sample = five same-camera renders fwd + bwd (model.py:119-167), eager ----
    if rank == 0 and single and not args.no_concurrent and args.config == 'c3' and train:
        try:
            result['extra_exavatar_iteration'] = iteration_throughput(device)
        except Exception as e:  # noqa: BLE001
            result['extra_exavatar_iteration'] = {'error': str(e)[:200]}

    # ---- extra: BASELINE configs[2] as worded -- "~150k Gaussians + SMPL-X LBS": the rasterizer behind a PyTorch LBS ----
    if rank == 0 and single and not args.no_concurrent and args.config == 'c3' and train:
        try:
            result['extra_c3_lbs'] = lbs_throughput(device)
        except Exception as e:  # noqa: BLE001
            result['extra_c3_lbs'] = {'error': str(e)[:200]}

    # ---- per-kernel HIP-event timing (eager, on torch's stream = the stream the kernels run on) -----
    if rank == 0 and not args.no_kernel_timing:
        c1 = make_ctx(1)
        _lib.timing_enable(True)
        acc = {}
        reps = 0
        # work counters of the measured views ("record P, V, D, per-tile list-length histogram with every
        # result"; the pixel-Gaussian evaluations feed the secondary roofline): read back from the workspaces of each render
        from exavatar_release_amd import rasterizer as rz_, stats as exa_stats
        work = []
        exa.config.keep_debug = train and not use_sh
        n_t = min(n_my, 20)
        for i in range(n_t + 2):
            set_view(i, c1)
            # two steps back to back, the second one is read: its launches are queued behind running work, so the
            # event brackets hold the kernels and not the ~5 us a launch needs to reach an idle GPU (agrees with the
            # rocprofv3 --kernel-trace durations of the bench command's step, profiles/)
            raster_step(c1)
            raster_step(c1)
            torch.cuda.synchronize()
            tm = _lib.timing_read()
            if i >= 2:
                reps += 1
                for k, v in tm.items():
                    acc[k] = acc.get(k, 0.0) + v
                if exa.config.keep_debug and rz_._debug_last:
                    work.append(exa_stats.render_stats(rz_._debug_last['tile'], rz_._debug_last['bin'], P, W, H,
                                                       rz_._debug_last['capacity']))
        _lib.timing_enable(False)
        exa.config.keep_debug = False
        rz_._debug_last.clear()
        avg_us = {k: v / reps * 1e3 for k, v in acc.items()}
        V, D, WH = V_mean, D_mean, W * H
        sh_bytes = 12 * 16 * P if use_sh else 0
        alg = {   # algorithmic bytes per launch (SURVEY.md 8(d) per-unit figures)
            'preprocess_fwd': 60 * P + 64 * V + sh_bytes,
            'cell_scatter': 16 * V + 4 * V + 4 * V,
            'subtile_bin': 16 * V + 8 * D,
            'sort_subtiles': 12 * D,
            'render_fwd': 4 * D + 40 * V + 28 * WH,
            # B1 of SURVEY.md 8(d) prices 28 B/px of image reads; 8 of them are dL/ddepth + dL/dalpha, which THIS workload (the
            # ExAvatar training case: both null) never reads -- they are left out, the line says so in `byte_model`
            'render_bwd': 4 * D + 80 * V + 20 * WH,
            'preprocess_bwd': 40 * V + 44 * V + 68 * P,
        }
        dom = max((k for k in avg_us if k in alg and alg[k] > 0), key=lambda k: avg_us[k])
        achieved = alg[dom] / (avg_us[dom] * 1e-6) / 1e9
        fwd_bytes = 60 * P + 88 * V + 40 * D + 28 * WH + sh_bytes
        total_bytes = (128 * P + 252 * V + 44 * D + 48 * WH) if train else fwd_bytes
        result['roofline'] = {
            'bound': 'hbm', 'kernel': dom, 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
            'frac': achieved / HBM_PEAK_GBS,
            'frac_with_survey_B1_bytes': ((alg[dom] + (8 * WH if dom == 'render_bwd' else 0)) / (avg_us[dom] * 1e-6) / 1e9 / HBM_PEAK_GBS),
            'algorithmic_bytes_per_launch': alg[dom], 'avg_launch_us': avg_us[dom],
            'clock': 'HIP events around eager launches of the kernel, this run (2nd of two back-to-back steps, stratified views); '
                     'the rocprofv3 --kernel-trace average of the same kernel inside the bench command\'s step is in profiles/ (1-3 us lower)',
            'byte_model': 'SURVEY.md 8(d) with the run\'s own P, V and D = 16x16 tile instances (header.num_tile_instances); the '
                          'backward blend is priced WITHOUT the 8 B/px of dL/ddepth + dL/dalpha reads of B1 (null in this workload, as '
                          'in ExAvatar training: the kernel instantiation that runs never reads them), i.e. 20 instead of 28 B/px',
            'kernel_avg_us': avg_us,
            'step': {'algorithmic_bytes': total_bytes, 'gpu_us_sum_of_kernels': sum(avg_us.values()),
                     'achieved_GBs_at_measured_step': total_bytes * units / (ms_per_step * 1e-3) / 1e9,
                     'frac_at_measured_step': total_bytes * units / (ms_per_step * 1e-3) / 1e9 / HBM_PEAK_GBS},
        }
        digest = lib_digest()
        tr, src = pmc_traffic(dom, digest) if args.config == 'c3' and train else (None, None)
        result['roofline']['traffic'] = tr
        result['roofline']['traffic_source'] = src
        result['roofline']['library_digest'] = digest
        rp = rocprof_time(dom, digest) if args.config == 'c3' and train else None
        if rp is not None and rp[0] is not None:      # the committed rocprofv3 average of the same kernel (kernel begin -> end, no launch gap in the bracket)
            result['roofline']['rocprof'] = {'avg_launch_us': rp[0], 'frac': alg[dom] / (rp[0] * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                             'source': rp[1]}
        elif rp is not None:
            result['roofline']['rocprof'] = {'avg_launch_us': None, 'source': rp[1]}
        if work:
            result['roofline']['secondary'] = valu_roofline(work, avg_us, digest)
            n_w = len(work)
            result['config']['list_length_histogram'] = {
                'what': 'non-empty 8x8-pixel sub-tile lists by length, mean over the %d measured views; bins start at' % n_w,
                'bin_starts': list(exa_stats.LIST_BINS)[1:], 'lists': [sum(w['list_hist'][b] for w in work) / n_w for b in range(1, len(exa_stats.LIST_BINS))],
                'lists_total': sum(w['lists'] for w in work) / n_w, 'longest': max(w['list_max'] for w in work),
                'subtile_instances': sum(w['instances'] for w in work) / n_w}
        eb = result.get('extra_batched_views')
        if isinstance(eb, dict) and 'ms_per_launch' in eb:
            kb = eb['views_per_launch']
            gbs = total_bytes * kb / (eb['ms_per_launch'] * 1e-3) / 1e9
            result['roofline_batched'] = {'views_per_launch': kb, 'algorithmic_bytes': total_bytes * kb,
                                          'ms_per_launch': eb['ms_per_launch'], 'achieved_GBs': gbs,
                                          'frac': gbs / HBM_PEAK_GBS, 'bound': 'hbm'}

    # ---- other BASELINE configs as driver-visible lines: C5 (configs[4], forward only) and C2 ----------
    if rank == 0 and single and not args.no_concurrent and args.config == 'c3' and train and not args.no_other_configs:
        # free this process's workspaces first: the child runs on the same GPU
        if sr is not None:
            sr.close()
        torch.cuda.empty_cache()
        result['extra_c5_forward'] = other_config('c5', args)
        result['extra_c2'] = other_config('c2', args)
        cold = cold_ring_order(args)
        result['extra_cold_ring_order'] = cold
        result['value_no_settle'] = cold.get('value')

    # ---- CPU baseline: the oracle on a bounded sample of the same workload (rank 0, N = 1) ---------
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result['cpu_baseline'] = cpu_baseline(args.config, assets, shape, train)

    # ---- RCCL smoke at world size 1 (the 8-GPU curve is the driver's to measure): init + one all-reduce ----
    if rank == 0 and world == 1 and single and not args.no_concurrent:
        result['rccl_world1_smoke'] = rccl_world1_smoke(device)

    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def surface_throughput(step_surface, args, exa):
    """The drop-in autograd surface -- GaussianRasterizer(settings)(...) + torch.autograd.grad, what the reference calls at
    avatar/common/nets/module.py:632-640 and differentiates from avatar/main/train.py:46 -- on the headline's workload and views,
    call by call, three ways: through the compiled autograd node (the default), the same with PyTorch's autograd engine told
    to run the backward on the calling thread (torch.autograd.set_multithreading_enabled(False): no hand-over to the device
    thread and back, ~40 us of host time per step), and through the Python node it replaces."""
    n, w = args.steps, max(args.warmup, 10)

    from exavatar_release_amd import rasterizer as rz_

    def run(label):
        for i in range(w + 50):
            step_surface(i, reduce=False)
        torch.cuda.synchronize()
        c_0 = rz_.compiled_calls
        # three windows of n steps, the median is reported: the loop is sensitive to what else the host's cores are doing (the
        # same build read 3 100 .. 7 000 it/s for the Python node on boxes of one pool)
        windows = []
        for _w in range(3):
            t_0 = time.perf_counter()
            for i in range(n):
                step_surface(w + i, reduce=False)
            t_host = time.perf_counter() - t_0           # the Python thread is done queueing; the GPU may still be busy
            torch.cuda.synchronize()
            windows.append(((time.perf_counter() - t_0) / n, t_host / n))
        dt, th = sorted(windows)[1]
        return {'value': 1.0 / dt, 'unit': 'iters/s', 'ms_per_step': dt * 1e3, 'host_ms_per_step': th * 1e3,
                'windows_it_per_s': [round(1.0 / x[0]) for x in windows],
                'renders_through_the_compiled_node': (rz_.compiled_calls - c_0) // 3, 'what': label}

    saved = exa.config.compiled_node
    try:
        res = run('GaussianRasterizer + torch.autograd.grad per step, eager, compiled autograd node (exavatar_release_amd/_exa_torch)')
        with torch.autograd.set_multithreading_enabled(False):
            res['autograd_multithreading_off'] = run('the same with torch.autograd.set_multithreading_enabled(False): backward on the calling thread')
        exa.config.compiled_node = 'off'
        res['python_node'] = run('the same through the Python autograd node (config.compiled_node = "off"): the surface of rounds 1-5')
    finally:
        exa.config.compiled_node = saved
    res['compiled_node_in_use'] = bool(rz_._compiled)
    return res


def abi_views_in_flight(S, args, exa, params, use_sh, shape, D_max, view_st, dL_dimg):
    """S views of the shard in flight through the C ABI: a StaticRender with S render slots (own workspaces, images and HIP
    stream each, shared inputs, plain launches), views dealt round-robin, every view's gradients into its slot's own arrays --
    the independent-steps form (what the headline counts, S at a time) -- and, second, the grouped form a trainer with a
    batch of S views uses: the S gradients ACCUMULATED in slot order into one set (chained per-Gaussian kernels)."""
    H, W = shape
    m3, sc, rot, op, col = [t.detach() for t in params]
    n_v = len(view_st)
    with exa.StaticRender(m3, op, sc, rot, colors_precomp=None if use_sh else col, shs=col if use_sh else None, image_size=(H, W),
                          capacity=int(D_max) + 64, slots=S) as sr:
        for st_j in view_st:
            sr.add_view(st_j, dL_dcolor=dL_dimg)
        sets = [sr.add_grad_outputs() for _ in range(S)]

        def independent(i):
            s = i % S
            sr.forward(i % n_v, slot=s)
            sr.backward(sets[s], slot=s)

        def grouped(i):
            sr.begin()
            for s in range(S):
                sr.forward((i * S + s) % n_v, slot=s)
                sr.backward(sets[0], slot=s, accumulate=s > 0, after=s - 1 if s else None)
            sr.end()
        done = [None, None]

        def pipelined(i):
            k = i & 1
            if done[k] is not None:
                sr.slot_stream(0).wait_event(done[k])           # the set this group overwrites was completed two groups ago
            for s in range(S):
                sr.forward((i * S + s) % n_v, slot=s)
                sr.backward(sets[k], slot=s, accumulate=s > 0, after=s - 1 if s else None)
            done[k] = torch.cuda.Event()
            done[k].record(sr.slot_stream(S - 1))
        out = {'views_in_flight': S}
        for name, fn, per in (('independent', independent, 1), ('grouped', grouped, S), ('grouped_pipelined', pipelined, S)):
            n = max(args.steps // per, 8) if per > 1 else max(args.steps, 8 * S)
            for i in range(40 * S // per):
                fn(i)
            sr.check()
            t_0 = time.perf_counter()
            for i in range(n):
                fn(i)
            sr.check()
            dt = time.perf_counter() - t_0
            out[name] = {'value': n * per / dt, 'unit': 'iters/s', 'ms_per_view': dt / (n * per) * 1e3}
        out['value'], out['unit'] = out['independent']['value'], 'iters/s'
        out['repairs'] = sr.repairs
        out['what'] = ('exa.StaticRender(slots=%d): plain launches through the C ABI on %d HIP streams; independent = every view fwd + bwd '
                       'into its slot\'s own gradient arrays; grouped = groups of %d views whose gradients are accumulated in slot order '
                       'into one set (backward(accumulate=True, after=previous slot)) with a barrier at every group end (what an optimizer '
                       'step per group forces); grouped_pipelined = the same chains into two alternating sets without barriers (gradient '
                       'accumulation, or the all-reduce of a group overlapping the next group: bench.py --views-in-flight)' % (S, S, S))
        return out


def lib_digest():
    """Digest of the sources the loaded library was built from (exavatar_release_amd.build._digest: csrc + header + flags) --
    committed profile summaries carry the digest of the build they measured."""
    try:
        from exavatar_release_amd import build as b
        return b._digest()[:16]
    except Exception:  # noqa: BLE001
        return None


# Issue cost of one wave64 instruction on one SIMD, MEASURED on MI355X with tools/probe/valu_probe.hip at 8 waves per SIMD
# (profiles/r05_valu_probe.txt): v_fma_f32 1.07 ns, v_exp_f32 3.51 ns (v_rcp_f32: the same pipe); a packed v_pk_fma_f32 costs
# 2.17 ns = two plain ones (two results per lane: no extra throughput), a v_cmp + v_cndmask pair ~3.6 ns.
# (MI355X_MICROARCH.md's 157.3 TFLOP/s vector peak = 1024 SIMDs x 64 lanes x 2 FLOP / 1.07 ns x 1.28: spec clock and spec
#  issue rate; the probe's sustained rate is what the blends can be held against.)
N_SIMD = 1024
VALU_NS, TRANS_NS = 1.07, 3.51


def valu_roofline(work, avg_us, digest=None):
    """SURVEY.md 8(d) secondary bound: "fp32 vector + v_exp_f32 throughput for the pixel-Gaussian evaluations per pass" --
    the one that binds the two blends (profiles/*_pmc.md: VALU-issue bound, a sixth of the HBM roofline).  Per kernel: the
    pixel-Gaussian pairs it evaluated in the measured views (from the workspaces), the VALU and transcendental
    wave-instructions per dispatch (SQ_INSTS_VALU of the committed PMC pass of the same workload; transcendentals: one
    v_exp_f32 per pair in the forward, v_exp_f32 + v_rcp_f32 in the backward, from the ISA), the issue time those
    instructions need on 1024 SIMDs, and the fraction of the measured launch that is."""
    n = len(work)
    pairs = {'render_fwd': sum(w['fwd_pairs'] for w in work) / n, 'render_bwd': sum(w['bwd_pairs'] for w in work) / n}
    trans_per_pair = {'render_fwd': 1, 'render_bwd': 2}
    pmc, src = pmc_counters(digest)
    out = {'bound': 'valu', 'unit': 'G wave-instructions/s',
           'peak': N_SIMD / VALU_NS,
           'peak_what': '%d SIMDs / %.2f ns per wave64 fp32 instruction (v_exp_f32 / v_rcp_f32: %.2f ns), measured: '
                        'tools/probe/valu_probe.hip, profiles/r05_valu_probe.txt.  Every VALU instruction is priced as a plain '
                        'one except the transcendentals, so `frac` is a LOWER bound of the share of the launch the VALU pipes '
                        'need: packed (v_pk_*), compare and DPP instructions take two passes -- SQ_ACTIVE_INST_VALU (valu_busy) '
                        'counts those' % (N_SIMD, VALU_NS, TRANS_NS),
           'counters_source': src, 'kernels': {}}
    for k in ('render_fwd', 'render_bwd'):
        if k not in avg_us:
            continue
        ent = {'pixel_gaussian_pairs': pairs[k], 'avg_launch_us': avg_us[k],
               'pairs_per_s': pairs[k] / (avg_us[k] * 1e-6)}
        insts = pmc.get(k, {}).get('SQ_INSTS_VALU') if pmc else None
        if insts:
            trans = pairs[k] / 64.0 * trans_per_pair[k]              # wave-instructions
            t_issue_us = ((insts - trans) * VALU_NS + trans * TRANS_NS) / N_SIMD * 1e-3
            ent.update({'valu_wave_insts_per_launch': insts, 'transcendental_wave_insts_per_launch': trans,
                        'valu_lane_ops_per_pair': insts * 64.0 / max(pairs[k], 1.0),
                        'achieved': insts / (avg_us[k] * 1e-6) / 1e9,
                        'issue_time_us_at_peak': t_issue_us, 'frac': t_issue_us / avg_us[k]})
            c = pmc.get(k, {})
            if c.get('SQ_ACTIVE_INST_VALU') and c.get('SQ_BUSY_CYCLES'):
                # quad-cycles the VALU pipes of the chip were executing / (4 SIMDs per CU x the kernel's busy quad-cycles per CU ...):
                # reported as the counter ratio the PMC summary uses: VALU-active quad-cycles per wave quad-cycle x resident waves
                ent['valu_active_per_wave_cycle'] = c['SQ_ACTIVE_INST_VALU'] / c['SQ_WAVE_CYCLES'] if c.get('SQ_WAVE_CYCLES') else None
        out['kernels'][k] = ent
    dom = max(out['kernels'], key=lambda k: out['kernels'][k]['avg_launch_us']) if out['kernels'] else None
    if dom and 'frac' in out['kernels'][dom]:
        out['kernel'], out['achieved'], out['frac'] = dom, out['kernels'][dom]['achieved'], out['kernels'][dom]['frac']
    return out


def _profile(suffix, digest):
    """(parsed profiles/<newest round>_<suffix>.json, its file name) if it was measured on THIS build -- the summary carries
    the digest of the library sources it was taken with (tools/make_profiles.py) and it must equal the loaded library's --
    else (None, why not).  A number measured on another build is not evidence for this one: the line then carries null."""
    here = os.path.dirname(os.path.abspath(__file__))
    for prefix in PROFILE_PREFIXES:
        name = prefix + '_' + suffix + '.json'
        path = os.path.join(here, 'profiles', name)
        if not os.path.exists(path):
            continue
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception as e:  # noqa: BLE001
            return None, 'profiles/%s is unreadable (%s)' % (name, e)
        have = d.get('library_digest')
        if digest is None or have != digest:
            return None, ('profiles/%s was measured on library build %s, the loaded library is %s: not quoted'
                          % (name, have or '(unstamped)', digest))
        return d, name
    return None, 'no profiles/*_%s.json' % suffix


def pmc_counters(digest=None):
    """(per-kernel SQ counters of the committed PMC pass, source) -- builder-side rocprofv3 --pmc passes of the same C3
    workload ON THE SAME BUILD, not collected in this run; ({}, reason) otherwise."""
    d, name = _profile('pmc', digest)
    if d is None:
        return {}, name
    return d['kernels'], 'profiles/%s (%s; build %s = the loaded library; not collected in this run)' % (name, d.get('what', ''), digest)


def pmc_traffic(kernel, digest=None):
    """(HBM bytes per launch of `kernel`, where the number comes from).  NOT measured in this run: read from the committed
    PMC passes (profiles/<round>_hbm_traffic.json: FETCH_SIZE and WRITE_SIZE collected in separate rocprofv3 --pmc runs
    of the same C3 workload on the same build); (None, reason) otherwise."""
    d, name = _profile('hbm_traffic', digest)
    if d is None:
        return None, name
    try:
        return float(d['kernels'][kernel]['hbm_bytes']), \
            'profiles/%s (builder-side rocprofv3 --pmc passes, %s; build %s = the loaded library; not collected in this run)' % (
                name, d.get('what', 'C3 view 0, exact mode'), digest)
    except Exception as e:  # noqa: BLE001
        return None, 'profiles/%s: %s' % (name, e)


def rocprof_time(kernel, digest=None):
    """(average duration in us of `kernel` by rocprofv3 --kernel-trace --stats, source) from the committed profile of the same
    build (builder-side, NOT measured in this run); (None, reason) otherwise."""
    d, name = _profile('kernel_stats', digest)
    if d is None:
        return None, name
    try:
        return float(d['kernels'][kernel]['avg_us']), 'profiles/%s (%s; build %s = the loaded library; not collected in this run)' % (
            name, d['what'], digest)
    except Exception as e:  # noqa: BLE001
        return None, 'profiles/%s: %s' % (name, e)


def other_config(cfg, args):
    """The headline step of another BASELINE config (c5: 300 k Gaussians, SH 3, 2048^2, forward only, hipGraph = configs[4];
    c2: 120 k, 540x960, fwd+bwd) in a child process on the same GPU, so that the driver's one default run sees them."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--config', cfg, '--steps', str(max(20, min(args.steps, 200))),
           '--warmup', str(max(5, min(args.warmup, 20))), '--no-cpu-baseline', '--no-concurrent', '--no-kernel-timing',
           '--no-other-configs']
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            return {'error': (r.stderr or r.stdout)[-300:]}
        d = json.loads(lines[-1])
        res = {k: d[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'steps', 'warmup', 'dtype', 'config')}
        # step-level HBM roofline of that line from its own P, V, D (SURVEY.md 8(d) byte model; forward only for c5)
        c = d['config']
        P, V, D, WH = c['P'], c['mean_visible_V'], c['mean_instances_D'], c['W'] * c['H']
        if 'forward' in c['mode']:
            nbytes = 60 * P + 88 * V + 40 * D + 28 * WH + (192 * P if cfg == 'c5' else 0)
        else:
            nbytes = 128 * P + 252 * V + 44 * D + 56 * WH
        gbs = nbytes / (d['ms_per_step'] * 1e-3) / 1e9
        res['roofline'] = {'bound': 'hbm', 'level': 'whole step (launch: %s)' % c.get('launch', '?'), 'algorithmic_bytes': nbytes, 'achieved': gbs,
                           'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': gbs / HBM_PEAK_GBS}
        return res
    except Exception as e:  # noqa: BLE001
        return {'error': str(e)[:200]}


def cold_ring_order(args):
    """The headline step measured the way rounds 1-2 measured it -- ring view order, NO settle phase: the timed steps start
    right after the capacity calibration and cover a 36-degree arc of frontal (heavy) views -- in a child process, so that the
    round-over-round trend stays readable next to the settled, stratified headline."""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), '--config', 'c3', '--steps', str(args.steps), '--warmup', str(args.warmup),
           '--no-cpu-baseline', '--no-concurrent', '--no-kernel-timing', '--no-other-configs']
    env = dict(os.environ, EXA_BENCH_SETTLE_STEPS='0', EXA_BENCH_VIEW_ORDER='ring')
    try:
        r = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode != 0 or not lines:
            return {'error': (r.stderr or r.stdout)[-300:]}
        d = json.loads(lines[-1])
        return {'value': d['value'], 'ms_per_step': d['ms_per_step'], 'steps': d['steps'], 'warmup': d['warmup'],
                'protocol': 'ring view order, settle_steps = 0 (the rounds 1-2 protocol; headline: stratified order + %s settle steps)'
                            % os.environ.get('EXA_BENCH_SETTLE_STEPS', '300')}
    except Exception as e:  # noqa: BLE001
        return {'error': str(e)[:200]}


def _timed_replays(ctxs, args, set_view, units_per_step):
    torch.cuda.synchronize()
    S = len(ctxs)

    def run(k0, k1):
        for i in range(k0, k1):
            c = ctxs[i % S]
            if c['stream'] is not None:
                with torch.cuda.stream(c['stream']):
                    set_view(i, c)
                    c['graph'].replay()
            else:
                set_view(i, c)
                c['graph'].replay()
    run(0, 4 * S)
    torch.cuda.synchronize()
    n = max(args.steps // units_per_step, 8)
    t0 = time.perf_counter()
    run(0, n)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    return n * units_per_step / dt, dt / n * 1e3


def iteration_throughput(device, n_scene=100_000, n_human=50_000, H=1024, W=1024, iters=30):
    """The five renders of one ExAvatar training sample -- scene, human, scene + human, refined human, scene + refined
    human (avatar/main/model.py:119-167; SURVEY.md 8d "ExAvatar-iteration equivalents") -- forward + backward through
    the drop-in Python surface, four ways: five sequential GaussianRenderer calls on torch.cat((scene.detach(), human)) as
    the reference writes it, the same five as one batched call (render_many), render_iteration (Gaussian sets shared, the
    composites as merges of the plain renders' sorted lists) -- these three EAGER -- and the product class
    GraphedIteration (render_iteration's kernels replayed from two captured hipGraphs, re-captured when P changes).
    100 k Dist-C scene + 50 k avatar-like human Gaussians at 1024 x 1024."""
    import exavatar_release_amd as exa
    from exavatar_release_amd import scenes
    keys = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')
    saved = (exa.config.mode, exa.config.fixed_capacity)
    exa.config.mode, exa.config.fixed_capacity = 'auto', None
    try:
        scene = {k: v.to(device).requires_grad_(True) for k, v in scenes.dist_c_scene(n_scene, H, W, seed=1).items()}
        human = {k: v.to(device).requires_grad_(True) for k, v in scenes.dist_b_avatar(n_human, seed=2).items()}
        refined = {k: v.detach().clone().requires_grad_(True) for k, v in human.items()}
        cam = {k: t.to(device) for k, t in scenes.ring_camera(H, W, 7, N_VIEWS).items()}
        bg = torch.rand(3, device=device)
        G = torch.randn(3, H, W, device=device)
        rend = exa.GaussianRenderer()
        cat = lambda a, b: {k: torch.cat((a[k].detach(), b[k])) for k in keys}      # noqa: E731

        graphed = exa.GraphedIteration((H, W), device)
        # the reference's photometric objective per render (avatar/main/model.py:197-198, 214-215) through the fused
        # producer of dL/dimg: human renders against the frame inside the person's bbox, the scene outside the mask
        photo = exa.PhotometricLoss()
        target = torch.rand(1, 3, H, W, device=device)
        mask = (torch.rand(1, 1, H, W, device=device) > 0.7).float()
        bbox = torch.tensor([[W // 4, H // 8, W // 2, 3 * H // 4]], dtype=torch.float32)

        def synthetic_loss(out, G_):
            return sum((out[k]['img'] * G_).sum() for k in exa.ITERATION_RENDERS)

        def photometric_loss(out, target_, mask_):
            loss = photo(out['scene']['img'][None], target_, l1_weight=1 - mask_, ssim_mask=1 - mask_)
            for k in exa.ITERATION_RENDERS[1:]:
                loss = loss + photo(out[k]['img'][None], target_, bbox=bbox)
            return loss
        # the same two losses RECORDED into the graph (GraphedIteration(loss_fn=...)): forward + loss + backward = one replay
        graphed_l = exa.GraphedIteration((H, W), device, loss_fn=synthetic_loss)
        graphed_pl = exa.GraphedIteration((H, W), device, loss_fn=photometric_loss)

        def iteration(how):
            if how.endswith('_in_graph'):
                for t in (scene, human, refined):
                    for v in t.values():
                        v.grad = None
                if how == 'graphed_loss_in_graph':
                    graphed_l(scene, human, refined, cam, bg, loss_args=(G,))['loss'].backward()
                else:
                    graphed_pl(scene, human, refined, cam, bg, loss_args=(target, mask))['loss'].backward()
                return
            if how.endswith('_photometric'):
                res = graphed(scene, human, refined, cam, bg) if how.startswith('graphed') else \
                    exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
                loss = photo(res['scene']['img'][None], target, l1_weight=1 - mask, ssim_mask=1 - mask)
                for k in exa.ITERATION_RENDERS[1:]:
                    loss = loss + photo(res[k]['img'][None], target, bbox=bbox)
                for t in (scene, human, refined):
                    for v in t.values():
                        v.grad = None
                loss.backward()
                return
            if how == 'graphed_raster_only':
                # what the headline measures for ONE render, for the five of an iteration: dL/dimg handed straight to the
                # backward (= the loss sum(img * G) without its eleven PyTorch kernels per render)
                res = graphed(scene, human, refined, cam, bg)
                for t in (scene, human, refined):
                    for v in t.values():
                        v.grad = None
                torch.autograd.backward([res[k]['img'] for k in exa.ITERATION_RENDERS], [G] * 5)
                return
            if how == 'graphed':
                res = graphed(scene, human, refined, cam, bg)
                outs = [res[k] for k in exa.ITERATION_RENDERS]
            elif how == 'sets':
                res = exa.render_iteration(rend, scene, human, refined, (H, W), cam, bg)
                outs = [res[k] for k in exa.ITERATION_RENDERS]
            else:
                jobs = [(scene, (H, W), cam), (human, (H, W), cam, bg), (cat(scene, human), (H, W), cam),
                        (refined, (H, W), cam, bg), (cat(scene, refined), (H, W), cam)]
                outs = exa.render_many(rend, jobs) if how == 'batched' else [rend(*j) for j in jobs]
            loss = sum((o['img'] * G).sum() for o in outs)
            for t in (scene, human, refined):
                for v in t.values():
                    v.grad = None
            loss.backward()
        out = {'workload': '%d k Dist-C scene + %d k avatar-like human Gaussians, %dx%d, 5 renders fwd+bwd, eager'
                           % (n_scene // 1000, n_human // 1000, W, H)}
        for how in ('sequential', 'batched', 'sets', 'graphed', 'graphed_raster_only', 'graphed_loss_in_graph', 'sets_photometric',
                    'graphed_photometric', 'graphed_photometric_in_graph'):
            # two iterations with the two-stage protocol first: they record the instance count of every render of THIS
            # scene (the capacity memo is keyed on (P, H, W), and the timed C3 runs above used P = 150 k as well)
            exa.config.mode = 'exact'
            for _ in range(2):
                iteration(how)
            exa.config.mode = 'auto'
            for _ in range(12):
                iteration(how)
            # three windows of `iters` iterations, the median is reported: a window that happens to contain a growth of
            # the caching allocator (hipMalloc) or a collection of the Python GC reads 0.1-0.2 ms per iteration high
            windows = []
            for _w in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    iteration(how)
                t_host = time.perf_counter() - t0        # the Python thread is done queueing; the GPU may still be busy
                torch.cuda.synchronize()
                windows.append(((time.perf_counter() - t0) / iters * 1e3, t_host / iters * 1e3))
            ms, host_ms = sorted(windows)[1]
            out[how] = {'ms_per_iteration': ms, 'renders_per_s': 5e3 / ms, 'host_ms_per_iteration': host_ms,
                        'windows_ms': [round(w[0], 4) for w in windows]}
            if how.endswith('_in_graph'):
                out[how]['what'] = ('exa.GraphedIteration(loss_fn=...): the five forwards, the loss (%s) and the backwards recorded '
                                    'into ONE hipGraph per iteration; loss.backward() hands the gradients the replay computed to '
                                    'the asset tensors' % ('fused PhotometricLoss, as graphed_photometric' if 'photometric' in how
                                                           else 'sum(img * G), as graphed'))
            elif how.endswith('_photometric'):
                out[how]['what'] = ('loss = the fused PhotometricLoss (L1 + SSIM, reference weights) of every render against a '
                                    'target image -- scene outside the mask, the four human renders inside a bbox -- instead of '
                                    'sum(img * G): what a train.py on this package runs per iteration around the rasterizer')
            if how == 'graphed_raster_only':
                out[how]['what'] = ('GraphedIteration with dL/dimg = G handed straight to backward (torch.autograd.backward(imgs, '
                                    '[G] * 5)): the rasterizer work of the iteration alone, as the headline metric counts one render')
            if how == 'graphed':
                out[how]['what'] = ('exa.GraphedIteration: one hipGraph for the five forwards, one for their backwards, same '
                                    'loss in PyTorch between them; captures=%d' % graphed.captures)
        return out
    finally:
        exa.config.mode, exa.config.fixed_capacity = saved


def lbs_throughput(device, P=150_000, H=1024, W=1024, iters=40):
    """BASELINE configs[2] as it is worded: "~150k Gaussians + SMPL-X LBS, 1024x1024".  The rasterizer's inputs are the
    outputs of a PyTorch-ROCm linear blend skinning (exavatar_release_amd.lbs.SyntheticAvatar: 55-joint tree, <= 4 skinning
    weights per vertex, per-vertex offsets, isotropic scales; reference avatar/common/nets/module.py:413-422, 516-586 -- the LBS
    stays in PyTorch, north_star), non-leaf tensors, and its gradients flow on into pose / translation / offsets.  Eager,
    through the drop-in GaussianRenderer: the whole chain, the LBS alone (forward + backward from synthetic gradients at its
    outputs) and the rasterizer alone on detached copies of the same tensors."""
    import exavatar_release_amd as exa
    from exavatar_release_amd import lbs, scenes
    saved = (exa.config.mode, exa.config.fixed_capacity)
    exa.config.mode, exa.config.fixed_capacity = 'auto', None
    try:
        model = lbs.SyntheticAvatar(scenes.dist_b_avatar(P, seed=0)).to(device)
        cam = {k: t.to(device) for k, t in scenes.ring_camera(H, W, 7, N_VIEWS).items()}
        bg = torch.ones(3, device=device)
        G = torch.randn(3, H, W, device=device)
        rend = exa.GaussianRenderer()
        gm, gs, gc = (torch.randn(P, 3, device=device) for _ in range(3))

        def zero():
            for p in model.parameters():
                p.grad = None

        def full():
            zero()
            out = rend(model(), (H, W), cam, bg)
            torch.autograd.backward([out['img']], [G])

        def lbs_only():
            zero()
            a = model()
            torch.autograd.backward([a['mean_3d'], a['scale'], a['rgb']], [gm, gs, gc])

        with torch.no_grad():
            frozen = {k: v.detach().clone() for k, v in model().items()}

        def raster_only():
            a = {k: v.requires_grad_(True) for k, v in frozen.items()}
            for v in a.values():
                v.grad = None
            out = rend(a, (H, W), cam, bg)
            torch.autograd.backward([out['img']], [G])

        res = {'workload': '%d k avatar-like Gaussians behind a synthetic SMPL-X-shaped LBS (55 joints), %dx%d, fwd+bwd; eager through the drop-in '
                           'renderer, and the whole chain replayed from one hipGraph (lbs_plus_raster_graphed)' % (P // 1000, W, H)}
        for name, fn in (('lbs_plus_raster', full), ('lbs_only', lbs_only), ('raster_only', raster_only)):
            exa.config.mode = 'exact'
            fn()
            exa.config.mode = 'auto'
            for _ in range(8):
                fn()
            windows = []
            for _w in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    fn()
                torch.cuda.synchronize()
                windows.append((time.perf_counter() - t0) / iters * 1e3)
            res[name] = {'ms_per_iteration': sorted(windows)[1], 'windows_ms': [round(w, 4) for w in windows]}
            if name == 'lbs_plus_raster':     # the rasterizer's gradients really arrive at the pose
                res['pose_grad_nonzero'] = model.pose.grad is not None and float(model.pose.grad.abs().max()) > 0
        # the same chain -- LBS, render, backward of both -- captured in ONE hipGraph (static parameters in, static gradients
        # out; capacity mode with the capacity the eager calls measured): what is left when the ~300 small PyTorch launches
        # of the LBS no longer go through the host one by one
        try:
            params = list(model.parameters())
            static_grads = [torch.zeros_like(p) for p in params]

            def graph_step():
                out = rend(model(), (H, W), cam, bg)
                grads = torch.autograd.grad([out['img']], params, grad_outputs=[G])
                for o, g_ in zip(static_grads, grads):
                    torch.add(g_, 0.0, out=o)            # (elementwise kernels: memcpy nodes do not replay reliably here)
            exa.config.mode = 'capacity'
            side = torch.cuda.Stream()
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):
                for _ in range(2):
                    graph_step()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                graph_step()
            for _ in range(5):
                graph.replay()
            windows = []
            for _w in range(3):
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                for _ in range(iters):
                    graph.replay()
                torch.cuda.synchronize()
                windows.append((time.perf_counter() - t0) / iters * 1e3)
            res['lbs_plus_raster_graphed'] = {'ms_per_iteration': sorted(windows)[1], 'windows_ms': [round(w, 4) for w in windows],
                                              'pose_grad_nonzero': float(static_grads[[i for i, p_ in enumerate(params) if p_ is model.pose][0]].abs().max()) > 0}
            del graph
        except Exception as e:  # noqa: BLE001
            res['lbs_plus_raster_graphed'] = {'error': str(e)[:200]}
        res['rasterizer_share'] = res['raster_only']['ms_per_iteration'] / res['lbs_plus_raster']['ms_per_iteration']
        res['value'] = 1e3 / res['lbs_plus_raster']['ms_per_iteration']
        res['unit'] = 'iters/s'
        return res
    finally:
        exa.config.mode, exa.config.fixed_capacity = saved


def batched_throughput(K, S, args, make_ctx, set_view, raster_step):
    """K views of this GPU's shard per batched launch (exa_raster_forward_batch / _backward_batch: ONE launch per
    pipeline stage for the K views, gradients of the shared Gaussians summed in the per-Gaussian kernel), S such
    launches in flight on separate streams (S = 2: the stages of one batch overlap those of the next).  Reported next to
    the headline, never instead of it."""
    ctxs = []
    for _ in range(S):
        c = make_ctx(K)
        if S > 1:
            c['stream'] = torch.cuda.Stream()
        set_view(0, c)
        side = c['stream'] or torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for _ in range(2):
                raster_step(c)
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        c['graph'] = torch.cuda.CUDAGraph()
        with torch.cuda.graph(c['graph']):
            raster_step(c)
        ctxs.append(c)
    val, ms = _timed_replays(ctxs, args, set_view, K)
    return {'views_per_launch': K, 'launches_in_flight': S, 'value': val, 'unit': 'iters/s', 'ms_per_launch': ms}


_RCCL_SMOKE = r"""
import os, sys, time, json, torch, torch.distributed as dist
sys.path.insert(0, sys.argv[1])
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', str(29600 + os.getpid() % 2000))
dev = torch.device('cuda', 0); torch.cuda.set_device(dev)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
from exavatar_release_amd.dist import FlatGradAllReducer
x = [torch.ones(1000, 3, device=dev), torch.ones(1000, 4, device=dev)]
red = FlatGradAllReducer(x, average=True, n_buffers=2)
t = torch.ones(1 << 20, device=dev)
torch.cuda.synchronize(); t0 = time.perf_counter()
dist.all_reduce(t); red.start(x); red.start(x); v = red.finish(); torch.cuda.synchronize()
ok = float(t.sum()) == float(1 << 20) and float(v[0].sum()) == 3000.0
print(json.dumps({'ok': bool(ok), 'backend': 'nccl (RCCL)', 'world_size': 1, 'all_reduce_ms': (time.perf_counter() - t0) * 1e3}))
dist.destroy_process_group()
"""


def rccl_world1_smoke(device):
    """RCCL initialisation + all-reduces through the product's FlatGradAllReducer in a world of ONE rank on this GPU
    (all a 1-GPU box can run; the 1/2/4/8-GPU curve is the driver's).  Runs in a child process with a timeout so that a
    communicator problem can never take the benchmark line down with it."""
    import subprocess
    try:
        r = subprocess.run([sys.executable, '-c', _RCCL_SMOKE, os.path.dirname(os.path.abspath(__file__))],
                           capture_output=True, text=True, timeout=120)
        lines = [ln for ln in r.stdout.splitlines() if ln.startswith('{')]
        if r.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {'ok': False, 'error': (r.stderr or r.stdout)[-200:]}
    except Exception as e:  # noqa: BLE001
        return {'ok': False, 'error': str(e)[:200]}


def cpu_baseline(cfg_name, assets, shape, train, max_threads=64):
    """CPU baseline (kind "port"): the C restatement of the rasterizer (oracle/c/raster_oracle.c -- sequential per-pixel
    loops in the shape upstream has them, OpenMP over Gaussians / tiles) on the SAME workload: ring view 0 of the config
    in full, forward + backward with the same dense dL/dimage (forward only for `--mode forward`), median of three runs
    after one warm-up, on min(host threads, 64) threads.  Falls back to the (much slower) PyTorch oracle on a bounded
    sample when the C library cannot be built."""
    try:
        from oracle import c_oracle as co
        co.load()
    except Exception as e:  # noqa: BLE001 -- no gcc on this host: time the PyTorch oracle instead
        res = cpu_baseline_torch(cfg_name, assets, shape, train)
        res['sample'] += ' [C oracle unavailable: %s]' % str(e)[:80]
        return res
    from exavatar_release_amd import scenes
    from oracle import raster_oracle as ro
    H, W = shape
    focal = 1500.0 * (H / 1024.0) if cfg_name != 'c2' else 1500.0 * 960 / 1024
    cam = scenes.neutral_camera(H, W) if cfg_name == 'c1' else scenes.ring_camera(H, W, 0, N_VIEWS, focal=focal)
    G = torch.randn(3, H, W, generator=torch.Generator().manual_seed(1)) if train else None
    use_sh = 'sh' in assets
    so = ro.settings_from_camera(cam, shape, torch.ones(3), 3 if use_sh else 0)
    cores = max(1, min(os.cpu_count() or 1, max_threads))
    co.set_num_threads(cores)

    def run():
        t0 = time.perf_counter()
        r = co.rasterize(assets['mean_3d'], assets['opacity'], shs=assets['sh'] if use_sh else None,
                         colors_precomp=None if use_sh else assets['rgb'], scales=assets['scale'],
                         rotations=assets['rotation'], settings=so, dL_dcolor=G)
        return time.perf_counter() - t0, r['num_rendered']
    run()
    times = sorted(run()[0] for _ in range(3))
    D = run()[1]
    t_med = times[1]
    return {'value': 1.0 / t_med, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
            'sample': 'C restatement of the rasterizer (oracle/c, gcc -O2 + OpenMP) %s of view 0 in full (%d 16x16-tile '
                      'instances) on %d of %d host threads: median of 3 runs %.3f s (min %.3f s)'
                      % ('fwd+bwd' if train else 'forward', D, cores, os.cpu_count() or 1, t_med, times[0])}


def cpu_baseline_torch(cfg_name, assets, shape, train, target_instances=40_000, max_threads=16):
    """Times oracle/raster_oracle.py (fwd + autograd bwd, float32; forward only for `--mode forward`) on a bounded
    sample: the full per-Gaussian stage plus the tiles nearest the image centre holding ~``target_instances`` tile
    instances; the per-tile part is scaled by the instance fraction.  Threads are capped at ``max_threads``
    (hundreds of OpenMP threads make the oracle's many small tensor ops far slower)."""
    from exavatar_release_amd import scenes
    from oracle import raster_oracle as ro
    cores = min(os.cpu_count() or 1, max_threads)
    torch.set_num_threads(cores)
    H, W = shape
    focal = 1500.0 * (H / 1024.0) if cfg_name != 'c2' else 1500.0 * 960 / 1024
    cam = scenes.neutral_camera(H, W) if cfg_name == 'c1' else scenes.ring_camera(H, W, 0, N_VIEWS, focal=focal)
    g = torch.Generator().manual_seed(1)
    G = torch.randn(3, H, W, generator=g)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    use_sh = 'sh' in assets
    so = ro.settings_from_camera(cam, shape, torch.ones(3), 3 if use_sh else 0)

    def run(sub):
        a = {k: v.clone().requires_grad_(train) for k, v in assets.items()}
        t0 = time.perf_counter()
        with torch.set_grad_enabled(train):
            res = ro.rasterize(a['mean_3d'], None, a['opacity'], shs=a['sh'] if use_sh else None,
                               colors_precomp=None if use_sh else a['rgb'], scales=a['scale'], rotations=a['rotation'],
                               settings=so, return_aux=True, tile_subset=sub)
            if train and res[0].requires_grad:
                (res[0] * G).sum().backward()
        return time.perf_counter() - t0, res[4]

    what = 'fwd+autograd-bwd' if train else 'forward'
    t_pre, aux = run([])                       # per-Gaussian stage + list building only
    ranges = aux['ranges']
    counts = (ranges[:, 1] - ranges[:, 0]).tolist()
    D_total = sum(counts)
    order = sorted(range(gx * gy), key=lambda t: ((t % gx) - gx / 2 + 0.5) ** 2 + ((t // gx) - gy / 2 + 0.5) ** 2)
    subset, D_sub = [], 0
    for t in order:
        if D_sub >= min(target_instances, D_total):
            break
        if counts[t]:
            subset.append(t)
            D_sub += counts[t]
    t_sub, _ = run(subset)
    frac = D_sub / max(D_total, 1)
    t_full = t_pre + max(t_sub - t_pre, 0.0) / max(frac, 1e-9)
    if t_full <= 25.0 and frac < 1.0:
        # the whole view fits the 10-30 s budget: time it in full instead of extrapolating
        t_meas, _ = run(None)
        return {'value': 1.0 / t_meas, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
                'sample': 'PyTorch CPU oracle %s of view 0 in full (%d tile instances) on %d of %d '
                          'host threads: %.1f s measured (a %.1f%% sample had predicted %.1f s)'
                          % (what, D_total, cores, os.cpu_count() or 1, t_meas, 100 * frac, t_full)}
    return {'value': 1.0 / t_full, 'unit': 'iters/s', 'cores': cores, 'kind': 'port',
            'sample': 'PyTorch CPU oracle %s of view 0 on %d of %d host threads: full per-Gaussian '
                      'stage (%.2f s) + the %d central tiles holding %.1f%% of the %d tile instances (%.2f s); '
                      'per-tile part scaled by the instance fraction -> %.1f s per iteration'
                      % (what, cores, os.cpu_count() or 1, t_pre, len(subset), 100 * frac, D_total, t_sub, t_full)}


if __name__ == '__main__':
    main()
