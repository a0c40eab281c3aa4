"""Host-side mirror of the reference's render plugin ``GaussianRenderer``
(``avatar/common/nets/module.py:588-647``): same ``forward(gaussian_assets, img_shape, cam_param, bg)``
signature, same output dict ``{img, depthmap, mask, mean_2d, is_vis, radius}``, bound to the gfx950
rasterizer instead of the third-party CUDA extension.

The only intentional differences: tensors are created on the device of the inputs instead of a
hard-coded ``.cuda()``, the default background is created per call (the reference evaluates
``torch.ones(3).cuda()`` at import time, module.py:592), and the 4x4 camera matrices are computed on the host
from one read-back of the camera tensors (same formulas, see ``forward``).
"""
import ctypes
import time
import warnings

import torch
import torch.nn as nn

from .camera import make_raster_matrices
from .rasterizer import (GaussianRasterizationSettings, rasterize_composites, rasterize_gaussians,
                         rasterize_gaussians_batch, take_is_vis)


_CAM_KEYS = ('focal', 'princpt', 'R', 't')
_cam_cache = []        # most recent first: (tensors, versions, img_shape, device, result); holds references to the key tensors
_CAM_CACHE_SIZE = 4


def _camera_block(cam_param, img_shape, device):
    """Camera matrices in the rasterizer's row-vector convention (module.py:604-608).

    The reference builds them from ~25 tiny device ops plus two ``float(tan(fov))`` read-backs per render; the settings
    need tan(fov) as Python floats anyway, so the four camera tensors are fetched in ONE read-back, the same helpers run
    on the host (camera.make_raster_matrices = the reference's formulas, float32, bit-identical) and the three results go
    back in one upload.  The five renders of an ExAvatar iteration share one ``cam_param`` (model.py:119-167), so the
    result is memoised on the IDENTITY and version counters of the four camera tensors (the cache keeps them alive, an
    in-place update bumps the version): renders 2-5 of an iteration skip the read-back, the host math and the upload.
    """
    shape = (int(img_shape[0]), int(img_shape[1]))
    tens = tuple(cam_param[k] for k in _CAM_KEYS)
    if all(isinstance(t, torch.Tensor) for t in tens):
        vers = tuple(t._version for t in tens)
        for i, (ct, cv, cs, cd, res) in enumerate(_cam_cache):
            if cs == shape and cd == device and cv == vers and all(a is b for a, b in zip(ct, tens)):
                if i:
                    _cam_cache.insert(0, _cam_cache.pop(i))
                return res
    else:
        vers = None
    cam_host = torch.cat([torch.as_tensor(t, dtype=torch.float32).reshape(-1) for t in tens]).detach().cpu()
    cam_cpu = {'focal': cam_host[0:2], 'princpt': cam_host[2:4], 'R': cam_host[4:13].view(3, 3), 't': cam_host[13:16]}
    tanfovx, tanfovy, view_h, proj_h, campos_h = make_raster_matrices(cam_cpu, shape, 0.01, 100.0)
    packed = torch.cat((view_h.reshape(-1), proj_h.reshape(-1), campos_h.reshape(-1))).to(device)
    res = (tanfovx, tanfovy, packed[0:16].view(4, 4), packed[16:32].view(4, 4), packed[32:35])
    if vers is not None:
        _cam_cache.insert(0, (tens, vers, shape, device, res))
        del _cam_cache[_CAM_CACHE_SIZE:]
    return res


_proj_cache = []       # most recent first: (focal tensor, its version, img_shape, (tanfovx, tanfovy, ctypes float[16]))


def _proj_host(cam_param, img_shape):
    """The part of the camera block that depends on focal length and image size only -- tan(fov / 2) as Python floats
    (kernel arguments) and get_proj_matrix's [4, 4] as sixteen host floats -- memoised on the identity + version of the
    ``focal`` tensor: a turntable / animation driver that moves the camera (new ``R``, ``t``) keeps one focal length, so
    this read-back happens once, not per frame."""
    from .camera import get_fov, get_proj_matrix
    shape = (int(img_shape[0]), int(img_shape[1]))
    f = cam_param['focal']
    if isinstance(f, torch.Tensor):
        for i, (cf, cv, cs, res) in enumerate(_proj_cache):
            if cf is f and cv == f._version and cs == shape:
                if i:
                    _proj_cache.insert(0, _proj_cache.pop(i))
                return res
    focal = torch.as_tensor(f, dtype=torch.float32).detach().reshape(-1).cpu()
    fov = get_fov(focal, None, shape)
    proj = get_proj_matrix(focal, None, shape, 0.01, 100.0, 1.0)
    res = (float(torch.tan(fov[0] / 2)), float(torch.tan(fov[1] / 2)),
           (ctypes.c_float * 16)(*[float(v) for v in proj.reshape(-1).tolist()]), float(focal[0]), float(focal[1]))
    if isinstance(f, torch.Tensor):
        _proj_cache.insert(0, (f, f._version, shape, res))
        del _proj_cache[4:]
    return res


def camera_block_device(cam_param, img_shape, out38, expect=None, flag=None):
    """Write the camera block of ``GaussianRenderer.forward`` (module.py:604-608) for DEVICE tensors ``cam_param['R']``
    / ``['t']`` into ``out38`` (float32 [>= 35] on the same device: viewmatrix 16 | projmatrix 16 | campos 3) with ONE
    tiny kernel (``exa_raster_camera_block``): no read-back of the extrinsics, no host matrix code, no upload.
    Returns the intrinsics record ``(tanfovx, tanfovy, proj16, fx, fy)`` it used.  The three columns of ``projmatrix``
    the rasterizer reads have one non-zero term each and equal the host path's; ``campos`` is ``-R^T t`` instead of a
    general matrix inverse (equal up to rounding for a rotation matrix; it only enters the SH view direction).

    ``expect`` + ``flag``: an intrinsics record from an earlier call and a ``(device address, tag)`` of 16 bytes of
    pinned host memory: the focal tensor is then NOT read back -- the kernel compares it with ``expect`` on the device
    and reports into the flag slot (1 = unchanged), which the caller polls after its frame."""
    from . import _lib
    dev = out38.device
    f = cam_param['focal']
    check = expect is not None and flag is not None and isinstance(f, torch.Tensor) and f.device == dev and \
        f.dtype == torch.float32 and f.is_contiguous() and f.numel() == 2
    intr = expect if check else _proj_host(cam_param, img_shape)
    R, t = cam_param['R'], cam_param['t']
    if not (isinstance(R, torch.Tensor) and R.device == dev and R.dtype == torch.float32 and R.is_contiguous()):
        R = torch.as_tensor(R, dtype=torch.float32).to(dev).contiguous()
    if not (isinstance(t, torch.Tensor) and t.device == dev and t.dtype == torch.float32 and t.is_contiguous()):
        t = torch.as_tensor(t, dtype=torch.float32).to(dev).contiguous()
    base = out38.data_ptr()
    _lib.check(_lib.load().exa_raster_camera_block(
        R.data_ptr(), t.data_ptr(), intr[2], base, base + 64, base + 128,
        f.data_ptr() if check else None, intr[3], intr[4], flag[0] if check else None, flag[1] if check else 0,
        ctypes.c_void_p(torch.cuda.current_stream(dev).cuda_stream)))
    return intr, check


_probe_cache = []      # most recent first: (P, device, zeros [P, 3]); treat the probes as read-only


_white_bg = {}


def _white(device):
    """The default background (module.py:596 builds ``torch.ones(3)`` per call): one cached tensor per device -- the rasterizer
    only reads it, and three of the five renders of an iteration use it (an allocation + a fill kernel each otherwise)."""
    w = _white_bg.get(device)
    if w is None:
        w = torch.ones(3, dtype=torch.float32, device=device)
        if device.type == 'cuda' and not torch.cuda.is_current_stream_capturing():   # (allocated during a capture = that graph's pool)
            _white_bg[device] = w
    return w


def _zero_probe(P, device):
    for i, (cp, cd, z) in enumerate(_probe_cache):
        if cp == P and cd == device:
            if i:
                _probe_cache.insert(0, _probe_cache.pop(i))
            break
    else:
        z = torch.zeros((P, 3), dtype=torch.float32, device=device)
        _probe_cache.insert(0, (P, device, z))
        del _probe_cache[4:]
    m = z.detach()
    m.requires_grad = True
    return m


def _sh_degree(gaussian_assets):
    """``None`` for the reference's assets (precomputed ``rgb``); for assets that carry spherical-harmonics
    coefficients ``sh`` [P, M, 3] instead, the degree to evaluate: ``sh_degree`` if given, else the largest one M holds."""
    if 'sh' not in gaussian_assets or gaussian_assets.get('rgb') is not None:
        return None
    if gaussian_assets.get('sh_degree') is not None:
        return int(gaussian_assets['sh_degree'])
    M = int(gaussian_assets['sh'].shape[1])
    deg = int(round(M ** 0.5)) - 1
    if (deg + 1) ** 2 != M or not 0 <= deg <= 3:
        raise ValueError('gaussian_assets["sh"] must hold 1, 4, 9 or 16 coefficients per colour channel (or pass sh_degree)')
    return deg


def _raster_job(gaussian_assets, img_shape, cam_param, bg, densify_stats=None, frozen_assets=None, cam_block=None,
                mean_2d=None):
    """Settings tuple + rasterizer keyword arguments of one render, built exactly as module.py:594-640 does.
    ``frozen_assets``: constant Gaussians blended together with ``gaussian_assets`` (render_many / render_iteration).
    ``cam_block``: a ready ``(tanfovx, tanfovy, viewmatrix, projmatrix, campos)`` instead of ``cam_param`` (device tensors
    a captured hipGraph keeps reading: :class:`graphed.GraphedIteration`); ``mean_2d``: the screen-space probe to use.
    Beyond the reference: assets with ``sh`` [P, M, 3] (and no ``rgb``) are coloured IN the rasterizer -- the reference
    evaluates ``clamp_min(eval_sh(...) + 0.5, 0)`` in PyTorch first (module.py:258-266) and passes ``rgb``."""
    mean_3d = gaussian_assets['mean_3d']
    device = mean_3d.device
    sh_degree = _sh_degree(gaussian_assets)
    if bg is None:
        bg = _white(device)

    tanfovx, tanfovy, view_matrix, full_proj_matrix, cam_pos = cam_block if cam_block is not None else \
        _camera_block(cam_param, img_shape, device)
    raster_settings = GaussianRasterizationSettings(
        image_height=img_shape[0],
        image_width=img_shape[1],
        tanfovx=tanfovx,
        tanfovy=tanfovy,
        bg=bg,
        scale_modifier=1.0,
        viewmatrix=view_matrix,
        projmatrix=full_proj_matrix,
        sh_degree=0 if sh_degree is None else sh_degree,  # 0 = dummy: rgb is already computed (module.py:618)
        campos=cam_pos,
        prefiltered=False,
        debug=False,
    )
    # screen-space position probe for the densification gradient (module.py:626-629: zeros, requires_grad, retain_grad).
    # The rasterizer never reads its VALUES (only its .grad is produced), so every render gets a fresh leaf that aliases one
    # cached block of zeros instead of paying an allocation + a fill kernel per render (retain_grad is a no-op on a leaf).
    if mean_2d is None:
        mean_2d = _zero_probe(mean_3d.shape[0], device)
    frozen = None
    if frozen_assets is not None:
        if (_sh_degree(frozen_assets) is None) != (sh_degree is None):
            raise ValueError('frozen_assets must carry the same colour input (rgb or sh) as gaussian_assets')
        frozen = dict(means3D=frozen_assets['mean_3d'], colors_precomp=frozen_assets['rgb'] if sh_degree is None else None,
                      shs=None if sh_degree is None else frozen_assets['sh'],
                      opacities=frozen_assets['opacity'], scales=frozen_assets['scale'],
                      rotations=frozen_assets['rotation'])
    return dict(raster_settings=raster_settings, means3D=mean_3d, means2D=mean_2d,
                shs=None if sh_degree is None else gaussian_assets['sh'],
                colors_precomp=gaussian_assets['rgb'] if sh_degree is None else None, opacities=gaussian_assets['opacity'],
                scales=gaussian_assets['scale'], rotations=gaussian_assets['rotation'], cov3D_precomp=None,
                densify_stats=densify_stats, frozen=frozen)


def _output_dict(job, outs, is_vis=None):
    """The reference's output dict (module.py:641-647).  ``is_vis`` (= ``radius > 0``): as the forward kernel wrote it
    (``rasterizer.take_is_vis``), so that a render costs no comparison kernel."""
    render_img, radius, render_depthmap, render_mask = outs
    return {'img': render_img,
            'depthmap': render_depthmap,
            'mask': render_mask,
            'mean_2d': job['means2D'],
            'is_vis': is_vis if is_vis is not None else radius > 0,
            'radius': radius}


class GaussianRenderer(nn.Module):
    def __init__(self):
        super(GaussianRenderer, self).__init__()

    def forward(self, gaussian_assets, img_shape, cam_param, bg=None, densify_stats=None):
        """Same call and output dict as module.py:592-647.  ``densify_stats`` (not in the reference): optional
        ``(xyz_grad_accum, track_cnt, radius_max)`` tensors that this render's backward updates in place -- the
        bookkeeping of avatar/main/model.py:279-285 + module.py:155-157 fused into the backward pass."""
        job = _raster_job(gaussian_assets, img_shape, cam_param, bg, densify_stats)
        # the reference instantiates GaussianRasterizer(raster_settings) per call (module.py:623) and calls it with
        # keywords; its forward is exactly this function call (rasterizer.GaussianRasterizer.forward), without building
        # an nn.Module per render
        outs = rasterize_gaussians(job['means3D'], job['means2D'], job['shs'], job['colors_precomp'], job['opacities'],
                                   job['scales'], job['rotations'], None, job['raster_settings'], densify_stats)
        vis = take_is_vis()
        return _output_dict(job, outs[:4], vis[0] if vis else None)


def render_many(renderer, jobs):
    """Multi-render batching: K independent renders in ONE launch per pipeline stage.

    The reference issues five renders per training iteration with the same camera -- scene, human, scene+human,
    human (refined), scene+human (refined), ``avatar/main/model.py:129-167`` -- one after the other, each a chain
    of ~10 dependent launches of latency-bound kernels that leave most of the MI355X idle.  Here the K renders
    become K jobs of one batched call (``exa_raster_forward_batch`` / ``_backward_batch``: every kernel takes the
    jobs in its arguments and picks its own with ``blockIdx.y``): 10 launches instead of 10 K in the forward and 2
    instead of 2 K in the backward, one autograd node, and the renders' workgroups fill the chip together.  Results
    are bit-identical to sequential calls (the pipeline has no atomics).

    ``jobs``: sequence of ``(gaussian_assets, img_shape, cam_param, bg[, densify_stats[, frozen_assets]])`` tuples
    (``bg`` may be ``None``).  ``frozen_assets``: a second asset dict of CONSTANT Gaussians rendered together with
    ``gaussian_assets`` -- what the reference writes as ``torch.cat((scene_asset[key].detach(), human_asset[key]))``
    (``model.py:119-126``), without the five concatenations in the autograd graph and without any backward work for the
    constant part; the render's ``radius`` / ``is_vis`` cover ``cat(frozen, own)``, its ``mean_2d`` probe only the own
    Gaussians.  Returns the list of output dicts of ``renderer.forward``.
    """
    jobs = list(jobs)
    if not jobs:
        return []
    device = jobs[0][0]['mean_3d'].device
    if device.type != 'cuda':
        raise RuntimeError('exavatar_release_amd: render_many runs on a ROCm device only')
    rj = [_raster_job(j[0], j[1], j[2], j[3] if len(j) > 3 else None, j[4] if len(j) > 4 else None,
                      j[5] if len(j) > 5 else None) for j in jobs]
    outs = rasterize_gaussians_batch(rj)
    vis = take_is_vis() or [None] * len(rj)
    return [_output_dict(j, o, v) for j, o, v in zip(rj, outs, vis)]


def render_views(renderer, gaussian_assets, img_shape, cam_params, bg=None):
    """K views of the SAME Gaussians in one batched call (the view shard one GPU holds of a data-parallel step,
    SURVEY.md 8e; the reference loops over samples one by one, ``avatar/main/model.py:81``).  The backward returns
    the SUM of the K views' gradients for the shared tensors (summed inside the per-Gaussian kernel), and one
    ``mean_2d`` probe per view.  Returns the list of K output dicts."""
    return render_many(renderer, [(gaussian_assets, img_shape, cp, bg) for cp in cam_params])


ITERATION_RENDERS = ('scene', 'human', 'scene_human', 'human_refined', 'scene_human_refined')


def render_iteration(renderer, scene_asset, human_asset, human_asset_refined, img_shape, cam_param, bg,
                     scene_densify_stats=None, merge=True, cam_block=None, probes=None):
    """The five same-camera renders of one ExAvatar training sample (``avatar/main/model.py:119-167``) as ONE batched
    call with the Gaussian SETS shared between them:

    ===================== ===================================== ==========================================
    render                reference                             here
    ===================== ===================================== ==========================================
    ``scene``             ``renderer(scene_asset, ...)``        job 0: scene, white background
    ``human``             ``renderer(human_asset, ..., bg)``    job 1: human, ``bg``
    ``scene_human``       ``cat(scene.detach(), human)``        job 2: human + CONSTANT scene prefix
    ``human_refined``     ``renderer(human_asset_refined, bg)`` job 3
    ``scene_human_refined`` ``cat(scene.detach(), refined)``    job 4: refined human + CONSTANT scene prefix
    ===================== ===================================== ==========================================

    The scene tensors enter the composite renders as constants: no ``torch.cat`` / ``CatBackward`` nodes, and the
    backward of jobs 2 and 4 skips every 64-entry batch that blended no human Gaussian (most of the image), writes no
    partial sums and runs no chain rule for the scene (``ExaRasterBackwardJob.grad_first``).  Images are bit-identical
    to the reference's formulation through :func:`render_many`, gradients agree to rounding (the prefix-aware backward
    kernels are separate instantiations: ~1e-7 relative).
    ``scene_densify_stats``: optional ``(xyz_grad_accum, track_cnt, radius_max)`` updated by the scene render's backward
    (``model.py:279-285``).  Returns a dict of the five output dicts keyed by :data:`ITERATION_RENDERS`.

    Shapes (both formulations): a composite's ``radius`` / ``is_vis`` cover ``cat(scene, human)`` (P_scene + P_human rows, as
    the reference's concatenated render returns them) while its ``mean_2d`` probe has the HUMAN's rows only (the scene is a
    constant there and gets no screen-space gradient); the reference reads only ``img`` of the composites.
    ``cam_block`` / ``probes``: see :func:`_raster_job` (``probes``: five ``mean_2d`` tensors in the order of
    :data:`ITERATION_RENDERS`), used by :class:`graphed.GraphedIteration`.

    ``merge=True`` (default): the two composites are not binned at all.  ``scene``, ``human`` and ``human_refined`` are
    three jobs of one batched call whose sorts keep their keys; ``scene_human`` / ``scene_human_refined`` are COMPOSITE
    renders (``exa_raster_forward_compose_batch``, csrc/compose.hip) that reuse those renders' splat records and MERGE their
    sorted per-sub-tile lists -- no preprocess, cell scatter, sub-tile binning or sort for two of the five renders, no
    ``torch.cat`` of the Gaussian tensors either.  Same images bit for bit.  ``merge=False``: the round-2 formulation (five
    jobs, the scene as a constant prefix of the composites).
    """
    device = scene_asset['mean_3d'].device
    if device.type != 'cuda':
        raise RuntimeError('exavatar_release_amd: render_iteration runs on a ROCm device only')
    modes = {_sh_degree(a) is None for a in (scene_asset, human_asset, human_asset_refined)}
    if len(modes) != 1:
        raise ValueError('render_iteration: scene, human and refined human must all carry the same colour input (rgb or sh)')
    pr = probes if probes is not None else (None,) * 5
    if cam_block is None:                    # the five renders share the camera: derive its matrices once
        cam_block = _camera_block(cam_param, img_shape, device)
    kw = dict(cam_block=cam_block)
    if merge and scene_asset['mean_3d'].shape[0] > 0 \
            and human_asset['mean_3d'].shape[0] > 0 and human_asset_refined['mean_3d'].shape[0] > 0:
        # three plain renders whose sorts keep their keys ...
        plain = [_raster_job(scene_asset, img_shape, cam_param, None, scene_densify_stats, mean_2d=pr[0], **kw),
                 _raster_job(human_asset, img_shape, cam_param, bg, mean_2d=pr[1], **kw),
                 _raster_job(human_asset_refined, img_shape, cam_param, bg, mean_2d=pr[3], **kw)]
        outs, handles = rasterize_gaussians_batch(plain, keep_keys=True)
        vis = take_is_vis() or (None, None, None)
        # ... and the two composites as MERGES of their sorted lists (white background, as the reference renders them)
        comp = [_raster_job(human_asset, img_shape, cam_param, None, mean_2d=pr[2], **kw),
                _raster_job(human_asset_refined, img_shape, cam_param, None, mean_2d=pr[4], **kw)]
        couts = rasterize_composites([(handles[0], handles[1]), (handles[0], handles[2])], comp, token=handles.token)
        cvis = take_is_vis()
        res = [_output_dict(plain[k], outs[k], vis[k]) for k in range(3)]

        def composite(k):
            # radius / is_vis cover cat(scene, human), as the reference's concatenated render returns them: written by the
            # composite's own ranges launch (ExaRasterComposeJob.radii_out / is_vis_out), no concatenation kernels
            img, radius, depth, mask = couts[k]
            return {'img': img, 'depthmap': depth, 'mask': mask, 'mean_2d': comp[k]['means2D'], 'is_vis': cvis[k], 'radius': radius}
        return dict(zip(ITERATION_RENDERS, [res[0], res[1], composite(0), res[2], composite(1)]))
    rj = [_raster_job(scene_asset, img_shape, cam_param, None, scene_densify_stats, mean_2d=pr[0], **kw),
          _raster_job(human_asset, img_shape, cam_param, bg, mean_2d=pr[1], **kw),
          _raster_job(human_asset, img_shape, cam_param, None, None, scene_asset, mean_2d=pr[2], **kw),
          _raster_job(human_asset_refined, img_shape, cam_param, bg, mean_2d=pr[3], **kw),
          _raster_job(human_asset_refined, img_shape, cam_param, None, None, scene_asset, mean_2d=pr[4], **kw)]
    outs = rasterize_gaussians_batch(rj)
    vis = take_is_vis() or [None] * 5
    return dict(zip(ITERATION_RENDERS, [_output_dict(j, o, v) for j, o, v in zip(rj, outs, vis)]))


class GraphedRenderer:
    """Forward-only renders of a FIXED number of Gaussians through a captured hipGraph (BASELINE configs[4]).

    The animation / turntable drivers of the reference render the same avatar frame after frame under
    ``torch.no_grad()`` (``avatar/main/animate.py:64-66``, ``animate_view_rot.py:104``, ``get_neutral_pose.py:86``):
    ``P`` and the image size never change, only the Gaussians' values and the camera do.  This class captures the ten
    kernel launches of one forward (no backward context stored) once and replays them per frame: inputs are copied
    into static tensors, the camera block into one 38-float static tensor, and the frame costs one graph launch
    instead of ~0.2 ms of host work::

        gr = GraphedRenderer(point_num, (H, W), device)              # sh_degree=3 for SH inputs ('sh' instead of 'rgb')
        for frame in frames:
            out = gr(human_asset, cam_param_of(frame), bg)           # dict like GaussianRenderer's, minus 'mean_2d'

    The returned tensors are the graph's static outputs: valid until the next call (clone what must survive).
    Instance capacity: measured by one eager render of the first frame, times ``capacity_growth``; after every replay
    the 16-byte header is read back (``check=True``: one host synchronisation per frame, free for drivers that pull the
    image to the host anyway) and a frame that overflowed is re-rendered after re-capturing with the capacity it needs.
    The field of view is baked into the kernel arguments: a change of focal length or image size re-captures.
    """

    _ASSET_KEYS = ('mean_3d', 'scale', 'rotation', 'opacity')

    def __init__(self, point_num, img_shape, device, sh_degree=None, capacity_growth=1.5, check=True, capacity=None):
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('exavatar_release_amd: GraphedRenderer runs on a ROCm device only')
        self.P, self.shape, self.device = int(point_num), (int(img_shape[0]), int(img_shape[1])), device
        self.sh_degree = sh_degree
        self.growth, self.check = float(capacity_growth), bool(check)
        f32 = dict(dtype=torch.float32, device=device)
        P = self.P
        self._in = {'mean_3d': torch.zeros((P, 3), **f32), 'scale': torch.zeros((P, 3), **f32),
                    'rotation': torch.zeros((P, 4), **f32), 'opacity': torch.zeros((P, 1), **f32)}
        if sh_degree is None:
            self._in['rgb'] = torch.zeros((P, 3), **f32)
        else:
            self._in['sh'] = torch.zeros((P, (int(sh_degree) + 1) ** 2, 3), **f32)
        self._cam = torch.zeros(38, **f32)                    # viewmatrix 16 | projmatrix 16 | campos 3 | bg 3
        self._mean_2d = torch.zeros((P, 3), **f32)
        self._capacity = int(capacity) if capacity is not None else None
        self._graph, self._tan, self._outs, self._tile = None, None, None, None
        self._last = {}                                       # key -> (source tensor, its version) of the previous frame
        self._slot = None                                     # (slot, tag) of the header report baked into the graph
        self._bg_src, self._bg_ver = None, None
        self._intr, self._focal_src, self._focal_ver = None, None, None     # last verified intrinsics record / focal tensor
        self.captures = 0

    def close(self):
        """Release the captured graph, its static outputs and the reserved report slot (after waiting for the device: a graph
        must not be destroyed while one of its replays is still executing).  The object captures anew when called again."""
        from . import rasterizer as rz
        if self._graph is not None and not torch.cuda.is_current_stream_capturing():
            # (garbage-collected in the middle of somebody else's stream capture: a device wait is illegal there and would
            #  invalidate that capture; the replays of this graph were queued before the capture began)
            torch.cuda.synchronize(self.device)
        self._graph, self._outs, self._tile, self._tan = None, None, None, None
        if self._slot is not None and rz._hdr_pool is not None:
            rz._hdr_pool.release(self._slot[0])
        self._slot = None

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def __del__(self):
        try:
            if self._graph is not None or self._slot is not None:
                warnings.warn('exavatar_release_amd: GraphedRenderer was garbage-collected with a live capture; call close()',
                              ResourceWarning, stacklevel=2)
                self.close()
        except Exception:  # noqa: BLE001 -- interpreter shutdown
            pass

    @property
    def inputs(self):
        """The static input tensors the graph reads (``mean_3d, scale, rotation, opacity, rgb | sh``): a producer may
        write its results straight into them (``out=``) and pass them back in -- no per-frame copy at all."""
        return self._in

    def _settings(self, tan):
        c = self._cam
        return GaussianRasterizationSettings(
            image_height=self.shape[0], image_width=self.shape[1], tanfovx=tan[0], tanfovy=tan[1], bg=c[35:38],
            scale_modifier=1.0, viewmatrix=c[0:16].view(4, 4), projmatrix=c[16:32].view(4, 4),
            sh_degree=0 if self.sh_degree is None else int(self.sh_degree), campos=c[32:35], prefiltered=False, debug=False)

    def _raster(self, tan):
        i = self._in
        return rasterize_gaussians(i['mean_3d'], self._mean_2d, i.get('sh'), i.get('rgb'), i['opacity'], i['scale'],
                                   i['rotation'], None, self._settings(tan))

    def _capture(self, tan):
        from . import rasterizer as rz
        saved = (rz.config.mode, rz.config.fixed_capacity, rz.config.keep_debug, rz.config.on_overflow)
        rz.config.on_overflow = 'retry'       # (the eager warm-up below repairs an overflow whatever the user's policy: the graph has its own path)
        try:
            with torch.no_grad():
                if self._capacity is None:                    # measure this frame once with the two-stage protocol
                    rz.config.mode, rz.config.fixed_capacity = 'exact', None
                    rz.config.keep_debug = True
                    self._raster(tan)
                    need = rz.read_header(rz._debug_last['tile'])[0]
                    self._capacity = max(int(need * self.growth), 1 << 16)
                self._capacity = (self._capacity + 63) // 64 * 64
                rz.config.mode, rz.config.fixed_capacity, rz.config.keep_debug = 'capacity', self._capacity, True
                side = torch.cuda.Stream(device=self.device)
                side.wait_stream(torch.cuda.current_stream(self.device))
                with torch.cuda.stream(side):                 # warm-up on a side stream, as torch.cuda.graph asks for
                    self._raster(tan)
                torch.cuda.current_stream(self.device).wait_stream(side)
                torch.cuda.synchronize(self.device)
                # the captured call reports its header into ONE pinned host slot (a plain store from the scatter kernel,
                # include/exa_raster.h: host_header): every replay rewrites it, the host resets the tag before a replay and
                # polls it afterwards -- no read-back, no synchronisation per frame
                pool = rz._pool()
                if self._slot is not None and pool is not None:
                    pool.release(self._slot[0])               # the graph that wrote into it is gone
                self._slot = None
                got = pool.reserve() if pool is not None else None   # outside the ring eager renders draw from
                if got is not None:
                    self._slot = (got[0], got[1])
                    rz._capture_report = [self._slot]
                try:
                    self._graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(self._graph):
                        self._outs = self._raster(tan)
                finally:
                    rz._capture_report = None
                self._tile = rz._debug_last['tile']           # the captured call's tile workspace (graph-private pool)
                rz._debug_last.clear()
                self._tan = tan
                self.captures += 1
        finally:
            rz.config.mode, rz.config.fixed_capacity, rz.config.keep_debug, rz.config.on_overflow = saved

    def __call__(self, gaussian_assets, cam_param, bg=None):
        from . import rasterizer as rz
        dev = self.device
        if bg is None:
            bg = torch.ones(3, dtype=torch.float32, device=dev)
        with torch.no_grad():
            for k in self._in:
                src = gaussian_assets[k]
                if tuple(src.shape) != tuple(self._in[k].shape):
                    raise ValueError('GraphedRenderer: %s has shape %s, captured for %s (P is fixed)'
                                     % (k, tuple(src.shape), tuple(self._in[k].shape)))
                # no copy for a tensor the caller wrote straight into ``self.inputs[k]``, nor for the very tensor object
                # of the previous frame if nothing wrote to it since (opacity / rotation / SH rest coefficients of an
                # animated avatar: 58 MB per frame for 300 k Gaussians at degree 3)
                if src is self._in[k]:
                    self._last.pop(k, None)                   # the static tensor no longer mirrors the previous source
                    continue
                last = self._last.get(k)
                if last is not None and last[0] is src and last[1] == src._version:
                    continue
                self._in[k].copy_(src)
                self._last[k] = (src, src._version)
            # camera: ONE kernel from the device-resident extrinsics (a new camera per frame costs no read-back, no host
            # matrix code and no upload); tan(fov / 2) and the projection entries come from the focal length, memoised
            if self._bg_src is not bg or self._bg_ver != getattr(bg, '_version', None):
                self._cam[35:38].copy_(torch.as_tensor(bg, dtype=torch.float32).reshape(-1))
                self._bg_src, self._bg_ver = bg, getattr(bg, '_version', None)
        f = cam_param['focal']
        with rz._on_device(dev):
            for _ in range(4):
                # A focal tensor that is the very object (and version) of the last verified frame is trusted; a new object
                # (a data loader hands out a fresh tensor per frame) is compared ON THE DEVICE with the remembered focal
                # length by the camera kernel, which reports into a pinned host word polled after the replay: no read-back.
                pool = rz._pool()
                trusted = self._intr is not None and f is self._focal_src and getattr(f, '_version', None) == self._focal_ver
                flag = None
                if self._intr is not None and not trusted and pool is not None:
                    fslot, ftag, faddr = pool.take()
                    flag = (faddr, ftag)
                intr, checking = camera_block_device(cam_param, self.shape, self._cam, self._intr if flag else None, flag)
                if not checking:                              # intrinsics came from the host path (memo or read-back): verified
                    self._intr, self._focal_src, self._focal_ver = intr, f, getattr(f, '_version', None)
                tan = (intr[0], intr[1])
                if self._graph is None or self._tan != tan:
                    self._capture(tan)
                if self._slot is not None:
                    words, b = rz._hdr_pool.words, 4 * self._slot[0]
                    words[b + 3] = 0                          # (the previous replay's report has been read: no store in flight)
                self._graph.replay()
                if checking:
                    words_f, bf = pool.words, 4 * fslot
                    t_end = time.perf_counter() + 5e-3
                    while words_f[bf + 3] != ftag and time.perf_counter() < t_end:
                        pass
                    if words_f[bf + 3] != ftag:
                        torch.cuda.current_stream(dev).synchronize()
                    if words_f[bf + 3] != ftag or words_f[bf] != 1:
                        self._intr = None                     # the focal length changed: derive it again (one read-back),
                        continue                              # re-capture if tan(fov) moved, and render this frame again
                    self._focal_src, self._focal_ver = f, getattr(f, '_version', None)
                if not self.check:
                    break
                if self._slot is not None:
                    t_end = time.perf_counter() + 5e-3
                    while words[b + 3] != self._slot[1] and time.perf_counter() < t_end:
                        pass
                    if words[b + 3] != self._slot[1]:
                        torch.cuda.current_stream(dev).synchronize()
                if self._slot is not None and words[b + 3] == self._slot[1]:
                    need, overflow = int(words[b]), int(words[b + 1])
                else:
                    need, overflow = rz.read_header(self._tile)[:2]
                if not overflow:
                    break
                self._capacity = int(need * self.growth)      # this frame needs more instances than any before it
                torch.cuda.synchronize(dev)                   # (the graph about to be dropped has a replay in flight)
                self._graph = None
            else:
                raise RuntimeError('exavatar_release_amd: GraphedRenderer could not size its instance buffer')
        color, radii, depth, alpha = self._outs
        return {'img': color, 'depthmap': depth, 'mask': alpha, 'is_vis': radii > 0, 'radius': radii}
