"""Worker of tests/test_gpu_edge_cases.py::test_two_rank_reduced_gradient_equals_single_process_sum (launched with
torch.distributed.run, 2 ranks, gloo, both on cuda:0).  Each rank rasterizes its shard of the views forward + backward
through the product surface, packs the gradients with dist.FlatGradAllReducer and all-reduces them (SUM); rank 0 then
renders the union of the views in one process and prints the relative errors as one JSON line."""
import json
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import exavatar_release_amd as exa                      # noqa: E402
from exavatar_release_amd import scenes                 # noqa: E402
from exavatar_release_amd import dist as exa_dist       # noqa: E402

KEYS = ('mean_3d', 'scale', 'rotation', 'opacity', 'rgb')


def main():
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    H, W, f, P, NV = 160, 192, 260.0, 8000, 6
    assets = scenes.dist_b_avatar(P, seed=11)
    g = torch.Generator().manual_seed(12)
    Gs = [torch.randn(3, H, W, generator=g).to(dev) for _ in range(NV)]
    bg = torch.rand(3, generator=g).to(dev)
    cams = [{k: v.to(dev) for k, v in scenes.ring_camera(H, W, 3 * v, 24, focal=f).items()} for v in range(NV)]
    rend = exa.GaussianRenderer()
    params = [assets[k].to(dev).requires_grad_(True) for k in KEYS]
    a = dict(zip(KEYS, params))
    red = exa_dist.FlatGradAllReducer(params, average=False, n_buffers=2)
    mine = exa_dist.shard_views(NV, rank, world, shuffle=True, seed=3)
    assert len(mine) == NV // world
    total = [torch.zeros_like(p) for p in params]
    # one asynchronous all-reduce per step, double-buffered exactly as bench.py drives it: buffer b is read back (and
    # added to the running total) right before step + 2 packs into it again
    def collect(b):
        red.wait(b)
        for t, view in zip(total, red.buffer_views(b)):
            t += view
    for step, v in enumerate(mine):
        b = step % 2
        if step >= 2:
            collect(b)
        out = rend(a, (H, W), cams[v], bg)
        grads = torch.autograd.grad((out['img'] * Gs[v]).sum(), params)
        red.pack(grads, b)
        red.reduce(b)
    for step in range(max(0, len(mine) - 2), len(mine)):
        collect(step % 2)
    torch.cuda.synchronize()
    if rank == 0:
        # single process, union of the views: K views per batched call, gradients summed inside the kernel
        ref_params = [assets[k].to(dev).requires_grad_(True) for k in KEYS]
        ra = dict(zip(KEYS, ref_params))
        outs = exa.render_views(rend, ra, (H, W), cams, bg)
        ref = torch.autograd.grad(sum((o['img'] * G).sum() for o, G in zip(outs, Gs)), ref_params)
        rel = {}
        for k, t, r in zip(KEYS, total, ref):
            scale = float(r.abs().max())
            if k == 'rotation':     # isotropic avatar: dL/drotation is rounding noise around zero; scale by |dL/dscale| |scale|
                scale = max(scale, float(ref[1].abs().max() * ref_params[1].detach().abs().max()))
            rel[k] = float((t - r).abs().max()) / max(scale, 1e-30)
        print(json.dumps({'world': world, 'views': NV, 'rel_err': rel}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
