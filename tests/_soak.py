"""Stateful soak of the path as training uses it: an optimisation loop shaped like the reference's
``avatar/main/train.py:28-57`` -- per iteration the five renders of a sample (``render_iteration`` or ``GraphedIteration``),
the fused ``PhotometricLoss`` against target images, ``Adam`` (eps 1e-15, reference ``avatar/common/base.py:84``), the fused
densification statistics, and every ``densify_every`` iterations clone / split / prune of the scene set with optimizer-state
surgery (``exavatar_release_amd.densify.densify_and_prune`` = reference ``avatar/common/nets/module.py:159-240``), so that P
goes up and down between calls; cameras updated IN PLACE (the binding's caches are keyed on tensor identity + version), a
``no_grad`` evaluation render every 10th step.  Exercises together what only a loop exercises: the capacity memo, the
pending header reports, the pinned slot ring, the camera / settings caches, the aliased ``mean_2d`` probes.

Used by tests/test_gpu_soak.py (single process) and as a worker (``python tests/_soak.py``) under ``torch.distributed.run``
(two ranks, gloo, one GPU): view-sharded data parallel with ``FlatGradAllReducer`` for the gradients,
``reduce_densify_stats`` for the statistics and a seed-synchronised generator for the split samples -- the replicas must stay
bit-identical."""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import exavatar_release_amd as exa                                   # noqa: E402
from exavatar_release_amd import densify as exa_densify                # noqa: E402
from exavatar_release_amd import dist as exa_dist                      # noqa: E402
from exavatar_release_amd import rasterizer as rz                      # noqa: E402
from exavatar_release_amd import scenes                                # noqa: E402

H, W, F = 128, 160, 170.0
N_VIEWS = 8
NAMES = ('mean', 'scale', 'rotation', 'opacity', 'rgb')


def _raw(assets, dev):
    """Trainable parameters in the reference's parametrisation: log scales, opacity logits (module.py:253-257)."""
    return {'mean': torch.nn.Parameter(assets['mean_3d'].to(dev).clone()),
            'scale': torch.nn.Parameter(assets['scale'].to(dev).log()),
            'rotation': torch.nn.Parameter(assets['rotation'].to(dev).clone()),
            'opacity': torch.nn.Parameter(torch.logit(assets['opacity'].to(dev).clamp(1e-3, 1 - 1e-3))),
            'rgb': torch.nn.Parameter(assets['rgb'].to(dev).clone())}


def _act(r):
    return {'mean_3d': r['mean'], 'scale': torch.exp(r['scale']), 'rotation': r['rotation'],
            'opacity': torch.sigmoid(r['opacity']), 'rgb': r['rgb']}


def run(dev, iters=300, mode='auto', graphed=False, densify_every=50, n_scene=1500, n_human=1500, rank=0, world=1, seed=0,
        forget_every=0, loss_in_graph=False):
    """Returns a dict: final parameters (list of tensors), losses, P history, counters.  ``forget_every``: every that many
    iterations the capacity memo is scaled down to 60 % -- the situation of a scene that suddenly needs more tile
    instances than any call before it, i.e. an overflow of every render of the next iteration.  ``loss_in_graph`` (with
    ``graphed``): the photometric loss is recorded into the graph (``GraphedIteration(loss_fn=...)``), the five target images
    travel as ``loss_args``."""
    saved = (exa.config.mode, exa.config.fixed_capacity, exa.config.capacity_growth, exa.config.min_capacity)
    exa.config.mode, exa.config.fixed_capacity = mode, None
    exa.config.capacity_growth, exa.config.min_capacity = 1.05, 64      # tight buffers: overflows WILL happen in 'auto'
    rz.overflow_events.clear()
    try:
        # three sets of EQUAL P at the start (scene, human, refined human): one capacity-memo key, three different D
        scene0 = scenes.dist_a_random(n_scene, H, W, seed=seed + 1, focal=F)
        human0 = scenes.dist_a_random(n_human, H, W, seed=seed + 2, focal=F, z_range=(2.0, 4.0))
        g = torch.Generator().manual_seed(seed + 3)
        # ground truth = the same sets, displaced; its renders are the targets
        tgt_sets = [{k: (v + 0.02 * torch.randn(v.shape, generator=g) if k in ('mean_3d', 'rgb') else v.clone()).to(dev)
                     for k, v in d.items()} for d in (scene0, human0, human0)]
        for d in tgt_sets:
            d['rgb'].clamp_(0, 1)
        cams = [scenes.ring_camera(H, W, 5 * v, 40, radius=3.2, center=(0.0, 0.0, 3.0), focal=F) for v in range(N_VIEWS)]
        rend = exa.GaussianRenderer()
        bg = torch.tensor([0.2, 0.5, 0.3], device=dev)
        targets = []
        with torch.no_grad():
            old = exa.config.mode
            exa.config.mode = 'exact'
            for c in cams:
                cd = {k: t.to(dev) for k, t in c.items()}
                res = exa.render_iteration(rend, *tgt_sets, (H, W), cd, bg)
                targets.append({n: res[n]['img'].clone()[None] for n in exa.ITERATION_RENDERS})
            exa.config.mode = old
        scene, human, refined = _raw(scene0, dev), _raw(human0, dev), _raw(human0, dev)
        groups = [{'params': [p], 'name': '%s_%d' % (n, i), 'lr': 2e-3 if n != 'mean' else 5e-4}
                  for i, r in enumerate((scene, human, refined)) for n, p in r.items()]
        opt = torch.optim.Adam(groups, lr=0.0, eps=1e-15)
        photo = exa.PhotometricLoss()
        cam = {k: t.to(dev).clone() for k, t in cams[0].items()}        # ONE set of camera tensors, updated in place
        stats = [torch.zeros(n_scene, 1, device=dev) for _ in range(3)]
        gen = exa_densify.synchronised_generator(1234 + seed, dev)
        def objective(out, *tgts):
            return sum(photo(out[n]['img'][None], t) for n, t in zip(exa.ITERATION_RENDERS, tgts))
        it = exa.GraphedIteration((H, W), dev, loss_fn=objective if loss_in_graph else None) if graphed else None
        my_views = exa_dist.shard_views(N_VIEWS, rank, world, shuffle=False)
        reducer = None
        losses, p_hist, evals = [], [], []
        for i in range(iters):
            v = my_views[i % len(my_views)]
            for k in cam:
                cam[k].copy_(cams[v][k].to(dev))                       # in place: same tensor objects, new versions
            opt.zero_grad(set_to_none=True)
            if forget_every and i % forget_every == forget_every - 1:
                for key in list(rz._seen_D):
                    rz._seen_D[key] = int(rz._seen_D[key] * 0.6)
            a_s, a_h, a_r = _act(scene), _act(human), _act(refined)
            if graphed and loss_in_graph:
                out = it(a_s, a_h, a_r, cam, bg, tuple(stats), loss_args=tuple(targets[v][n] for n in exa.ITERATION_RENDERS))
                loss = out['loss']
            else:
                if graphed:
                    out = it(a_s, a_h, a_r, cam, bg, tuple(stats))
                else:
                    out = exa.render_iteration(rend, a_s, a_h, a_r, (H, W), cam, bg, tuple(stats))
                loss = sum(photo(out[n]['img'][None], targets[v][n]) for n in exa.ITERATION_RENDERS)
            loss.backward()
            params = [p for r in (scene, human, refined) for p in r.values()]
            if world > 1:
                if reducer is None or reducer.numels != [p.numel() for p in params]:
                    reducer = exa_dist.FlatGradAllReducer(params, average=True)
                red = reducer.start([p.grad for p in params]).finish()
                for p, gr in zip(params, red):
                    p.grad = gr.clone()
            opt.step()
            losses.append(float(loss.detach()))
            if i % 10 == 9:                                            # evaluation render, no autograd
                with torch.no_grad():
                    ev = rend(_act(human), (H, W), cam, bg)
                    evals.append(float(ev['img'].mean()))
            if densify_every and i % densify_every == densify_every - 1 and i + 1 < iters:
                if world > 1:
                    exa_dist.reduce_densify_stats(*stats)
                # thresholds chosen so that the set grows in the first rounds and shrinks later (P up AND down)
                rnd = i // densify_every
                thr = float(torch.nan_to_num(stats[0] / stats[1]).flatten().quantile(0.85 if rnd < 3 else 0.97))
                new, n_c, n_s, n_p = exa_densify.densify_and_prune(
                    scene, opt, stats[0], stats[1], grad_thr=thr, extent=4.0, dense_percent=0.004,
                    opacity_min=0.05 if rnd < 3 else 0.35, generator=gen)
                scene = new
                stats = [torch.zeros(scene['mean'].shape[0], 1, device=dev) for _ in range(3)]
            p_hist.append(int(scene['mean'].shape[0]))
        torch.cuda.synchronize()
        final = [p.detach().clone() for r in (scene, human, refined) for p in r.values()]
        return {'final': final, 'losses': losses, 'p_hist': p_hist, 'evals': evals,
                'overflow_events': list(rz.overflow_events),
                'captures': it.captures if it is not None else 0,
                'retries': it.overflow_retries if it is not None else 0}
    finally:
        exa.config.mode, exa.config.fixed_capacity, exa.config.capacity_growth, exa.config.min_capacity = saved


def main():
    """Two-rank worker: both replicas must end with bit-identical parameters."""
    import torch.distributed as dist
    rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])
    dist.init_process_group('gloo')
    dev = torch.device('cuda', 0)
    torch.cuda.set_device(dev)
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 120
    res = run(dev, iters=iters, mode='auto', graphed=False, densify_every=40, rank=rank, world=world, forget_every=23)
    flat = torch.cat([t.reshape(-1) for t in res['final']]).cpu()
    n = torch.tensor([flat.numel()])
    sizes = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(sizes, n)
    same_len = all(int(s) == int(n) for s in sizes)
    equal = False
    if same_len:
        both = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        equal = all(torch.equal(both[0], b) for b in both[1:])
    if rank == 0:
        print(json.dumps({'world': world, 'iters': iters, 'same_length': same_len, 'replicas_bit_identical': equal,
                          'p_first': res['p_hist'][0], 'p_max': max(res['p_hist']), 'p_last': res['p_hist'][-1],
                          'loss_first': sum(res['losses'][:10]) / 10, 'loss_last': sum(res['losses'][-10:]) / 10,
                          'events': [e[3] for e in res['overflow_events']]}))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == '__main__':
    main()
