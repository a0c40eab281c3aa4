"""Host-side mirror of the reference's render plugin ``GaussianRenderer``
(``avatar/common/nets/module.py:588-647``): same ``forward(gaussian_assets, img_shape, cam_param, bg)``
signature, same output dict ``{img, depthmap, mask, mean_2d, is_vis, radius}``, bound to the gfx950
rasterizer instead of the third-party CUDA extension.

The only intentional differences: tensors are created on the device of the inputs instead of a
hard-coded ``.cuda()``, the default background is created per call (the reference evaluates
``torch.ones(3).cuda()`` at import time, module.py:592), and the 4x4 camera matrices are computed on the host
from one read-back of the camera tensors (same formulas, see ``forward``).
"""
import torch
import torch.nn as nn

from .camera import make_raster_matrices
from .rasterizer import GaussianRasterizationSettings, GaussianRasterizer


class GaussianRenderer(nn.Module):
    def __init__(self):
        super(GaussianRenderer, self).__init__()

    def forward(self, gaussian_assets, img_shape, cam_param, bg=None):
        # assets for the rendering (module.py:594-598)
        mean_3d = gaussian_assets['mean_3d']
        opacity = gaussian_assets['opacity']
        scale = gaussian_assets['scale']
        rotation = gaussian_assets['rotation']
        rgb = gaussian_assets['rgb']
        device = mean_3d.device
        if bg is None:
            bg = torch.ones((3), dtype=torch.float32, device=device)

        # camera matrices in the rasterizer's row-vector convention (module.py:604-608).  The reference builds
        # them from ~25 tiny device ops plus two `float(tan(fov))` read-backs; the settings need tan(fov) as Python
        # floats anyway, so the four camera tensors are fetched in ONE read-back, the same helpers run on the host
        # (camera.make_raster_matrices = the reference's formulas), and the three results go back in one upload:
        # 0.55 ms -> 0.1 ms of host time per render (tools/gpu_host_profile.py), same values.
        cam_host = torch.cat([torch.as_tensor(cam_param[k], dtype=torch.float32).reshape(-1)
                              for k in ('focal', 'princpt', 'R', 't')]).detach().cpu()
        cam_cpu = {'focal': cam_host[0:2], 'princpt': cam_host[2:4], 'R': cam_host[4:13].view(3, 3), 't': cam_host[13:16]}
        tanfovx, tanfovy, view_h, proj_h, campos_h = make_raster_matrices(cam_cpu, img_shape, 0.01, 100.0)
        packed = torch.cat((view_h.reshape(-1), proj_h.reshape(-1), campos_h.reshape(-1))).to(device)
        view_matrix, full_proj_matrix, cam_pos = packed[0:16].view(4, 4), packed[16:32].view(4, 4), packed[32:35]
        raster_settings = GaussianRasterizationSettings(
            image_height=img_shape[0],
            image_width=img_shape[1],
            tanfovx=tanfovx,
            tanfovy=tanfovy,
            bg=bg,
            scale_modifier=1.0,
            viewmatrix=view_matrix,
            projmatrix=full_proj_matrix,
            sh_degree=0,  # dummy: rgb is already computed (module.py:618)
            campos=cam_pos,
            prefiltered=False,
            debug=False,
        )
        rasterizer = GaussianRasterizer(raster_settings=raster_settings)

        # screen-space position probe for the densification gradient (module.py:626-629)
        point_num = mean_3d.shape[0]
        mean_2d = torch.zeros((point_num, 3), dtype=torch.float32, device=device)
        mean_2d.requires_grad = True
        mean_2d.retain_grad()

        render_img, radius, render_depthmap, render_mask = rasterizer(
            means3D=mean_3d,
            means2D=mean_2d,
            shs=None,
            colors_precomp=rgb,
            opacities=opacity,
            scales=scale,
            rotations=rotation,
            cov3D_precomp=None)

        return {'img': render_img,
                'depthmap': render_depthmap,
                'mask': render_mask,
                'mean_2d': mean_2d,
                'is_vis': radius > 0,
                'radius': radius}


_stream_pool = {}


def render_many(renderer, jobs):
    """Multi-render batching: run independent renders concurrently, one HIP stream each.

    The reference issues five renders per training iteration with the same camera -- scene, human, scene+human,
    human (refined), scene+human (refined), ``avatar/main/model.py:129-167`` -- one after the other.  Every
    render is a chain of ~10 dependent launches of latency-bound kernels that leave most of the MI355X idle
    (``profiles/``: 4 renders in flight raise the throughput of one GPU by ~30 %), and the renders are independent
    of each other, so they are put on separate streams here; autograd later runs each render's backward on the
    stream of its forward, so the backward passes overlap as well.  Results are bit-identical to sequential calls
    (the pipeline has no atomics).  This pays when the GPU is the bottleneck (hipGraph replays: bench.py's
    ``extra_views_in_flight``); plain eager calls are bound by ~0.5 ms of host work per render and gain nothing
    (tools/gpu_render_many.py).

    ``jobs``: sequence of ``(gaussian_assets, img_shape, cam_param, bg)`` tuples (``bg`` may be ``None``).
    Returns the list of output dicts of ``renderer.forward``.  Use capacity mode (``config.mode = "capacity"``)
    to also remove the per-render host synchronisation.
    """
    jobs = list(jobs)
    if not jobs:
        return []
    device = jobs[0][0]['mean_3d'].device
    if device.type != 'cuda':
        raise RuntimeError('exavatar_release_amd: render_many runs on a ROCm device only')
    main = torch.cuda.current_stream(device)
    pool = _stream_pool.setdefault(device.index, [])
    while len(pool) < len(jobs):
        pool.append(torch.cuda.Stream(device=device))
    outs = []
    for job, side in zip(jobs, pool):
        assets, img_shape, cam_param = job[0], job[1], job[2]
        bg = job[3] if len(job) > 3 else None
        side.wait_stream(main)
        with torch.cuda.stream(side):
            out = renderer(assets, img_shape, cam_param, bg)
        for v in out.values():          # produced on a side stream, consumed on the caller's stream
            if isinstance(v, torch.Tensor):
                v.record_stream(main)
        outs.append(out)
    for side in pool[:len(jobs)]:
        main.wait_stream(side)
    return outs
