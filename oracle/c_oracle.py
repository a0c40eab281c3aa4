"""ctypes front end of the C restatement of the rasterizer (``oracle/c/raster_oracle.c``).

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE: only ``tests/`` and the ``cpu_baseline`` leg of ``bench.py`` import it.
PARITY UNPINNED, see the header of ``raster_oracle.c``: a second, independently written restatement of the
published algorithm behind reference ``avatar/common/nets/module.py:609-640`` -- sequential per-pixel loops and a
hand-written backward in the shape upstream has them -- held against the vectorised PyTorch oracle
(``oracle/raster_oracle.py``) in ``tests/test_c_oracle.py``.  OpenMP over Gaussians / tiles: also the multi-threaded
CPU baseline of ``bench.py``.

Build: ``make -C oracle/c`` (done by ``__graft_entry__.build()``).
"""
import ctypes
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'c', 'liboracle_c.so')


class OrcSettings(ctypes.Structure):
    _fields_ = [('image_height', ctypes.c_int32), ('image_width', ctypes.c_int32),
                ('tanfovx', ctypes.c_float), ('tanfovy', ctypes.c_float), ('scale_modifier', ctypes.c_float),
                ('sh_degree', ctypes.c_int32), ('bg', ctypes.c_float * 3), ('viewmatrix', ctypes.c_float * 16),
                ('projmatrix', ctypes.c_float * 16), ('campos', ctypes.c_float * 3)]


_lib = None


def available():
    return os.path.exists(LIB_PATH)


def load(build=True):
    """Load (and, if missing, build with gcc) the shared library."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        if not build:
            raise RuntimeError('oracle/c/liboracle_c.so is missing: run `make -C oracle/c`')
        subprocess.run(['make', '-s', '-C', os.path.join(_HERE, 'c')], check=True)
    lib = ctypes.CDLL(LIB_PATH)
    p = ctypes.c_void_p
    lib.exa_oracle_render.restype = ctypes.c_long
    lib.exa_oracle_render.argtypes = [ctypes.POINTER(OrcSettings), ctypes.c_int32, ctypes.c_int32] + [p] * 25
    lib.exa_oracle_num_threads.restype = ctypes.c_int
    lib.exa_oracle_set_num_threads.argtypes = [ctypes.c_int]
    _lib = lib
    return lib


def set_num_threads(n):
    load().exa_oracle_set_num_threads(int(n))


def num_threads():
    return int(load().exa_oracle_num_threads())


def _settings(s):
    st = OrcSettings()
    st.image_height, st.image_width = int(s.image_height), int(s.image_width)
    st.tanfovx, st.tanfovy, st.scale_modifier = float(s.tanfovx), float(s.tanfovy), float(s.scale_modifier)
    st.sh_degree = int(s.sh_degree)
    for name, n in (('bg', 3), ('viewmatrix', 16), ('projmatrix', 16), ('campos', 3)):
        v = torch.as_tensor(getattr(s, name), dtype=torch.float32).reshape(-1).tolist()
        assert len(v) == n, name
        getattr(st, name)[:] = v
    return st


def _f32(t):
    if t is None:
        return None
    return np.ascontiguousarray(torch.as_tensor(t).detach().to(torch.float32).numpy())


def _ptr(a):
    return ctypes.c_void_p(a.ctypes.data) if a is not None else ctypes.c_void_p(0)


def rasterize(means3D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None, cov3D_precomp=None,
              settings=None, dL_dcolor=None, dL_ddepth=None, dL_dalpha=None):
    """Forward (and, with ``dL_dcolor``, backward) of one render; keyword names as at reference module.py:632-640,
    ``settings`` = any 12-field settings tuple (``oracle.raster_oracle.OracleSettings`` /
    ``GaussianRasterizationSettings``) with CPU tensors.

    Returns a dict: ``color [3,H,W]``, ``depth [1,H,W]``, ``alpha [1,H,W]``, ``radii [P] int32``, ``final_T [H,W]``,
    ``n_contrib [H,W] int32``, ``pixel_margin [H,W]`` (``ambiguous = pixel_margin < 1e-4``, the mask
    ``oracle.raster_oracle.ambiguous_pixel_mask`` computes), ``num_rendered`` and -- when ``dL_dcolor`` is given -- ``grads``: a dict with
    ``means2D, means3D, opacities`` and whichever of ``colors_precomp / shs``, ``scales + rotations / cov3D_precomp``
    apply (same shapes as the inputs).
    """
    lib = load()
    st = _settings(settings)
    H, W = st.image_height, st.image_width
    m3, op = _f32(means3D), _f32(opacities)
    P = int(m3.shape[0])
    sh, col, sc, rot, cov = _f32(shs), _f32(colors_precomp), _f32(scales), _f32(rotations), _f32(cov3D_precomp)
    sh_M = int(sh.shape[1]) if sh is not None else 0
    color = np.empty((3, H, W), np.float32)
    depth = np.empty((1, H, W), np.float32)
    alpha = np.empty((1, H, W), np.float32)
    radii = np.zeros((P,), np.int32)
    final_T = np.empty((H, W), np.float32)
    n_contrib = np.empty((H, W), np.int32)
    margin = np.empty((H, W), np.float32)
    gc, gd, ga = _f32(dL_dcolor), _f32(dL_ddepth), _f32(dL_dalpha)
    grads = {}
    if gc is not None:
        grads = {'means2D': np.zeros((P, 3), np.float32), 'means3D': np.zeros((P, 3), np.float32),
                 'opacities': np.zeros((P, 1), np.float32)}
        if col is not None:
            grads['colors_precomp'] = np.zeros((P, 3), np.float32)
        if sh is not None:
            grads['shs'] = np.zeros((P, sh_M, 3), np.float32)
        if sc is not None:
            grads['scales'] = np.zeros((P, 3), np.float32)
            grads['rotations'] = np.zeros((P, 4), np.float32)
        if cov is not None:
            grads['cov3D_precomp'] = np.zeros((P, 6), np.float32)
    g = grads.get
    D = lib.exa_oracle_render(ctypes.byref(st), P, sh_M, _ptr(m3), _ptr(sh), _ptr(col), _ptr(op), _ptr(sc), _ptr(rot),
                              _ptr(cov), _ptr(color), _ptr(depth), _ptr(alpha), _ptr(radii), _ptr(final_T),
                              _ptr(n_contrib), _ptr(margin), _ptr(gc), _ptr(gd), _ptr(ga), _ptr(g('means2D')), _ptr(g('means3D')),
                              _ptr(g('colors_precomp')), _ptr(g('opacities')), _ptr(g('scales')), _ptr(g('rotations')),
                              _ptr(g('shs')), _ptr(g('cov3D_precomp')))
    if D < 0:
        raise RuntimeError('exa_oracle_render failed with status %d' % D)
    out = {'color': torch.from_numpy(color), 'depth': torch.from_numpy(depth), 'alpha': torch.from_numpy(alpha),
           'radii': torch.from_numpy(radii), 'final_T': torch.from_numpy(final_T),
           'n_contrib': torch.from_numpy(n_contrib), 'pixel_margin': torch.from_numpy(margin), 'num_rendered': int(D)}
    if gc is not None:
        out['grads'] = {k: torch.from_numpy(v) for k, v in grads.items()}
    return out


def render(gaussian_assets, img_shape, cam_param, bg=None, dL_dimg=None, dL_ddepth=None, dL_dalpha=None):
    """C twin of ``GaussianRenderer.forward`` (reference module.py:592-647) on asset dicts
    ``{mean_3d, scale, rotation, opacity, rgb}``; ``grads`` are keyed like the assets (+ ``mean_2d``)."""
    from oracle import raster_oracle as ro
    if bg is None:
        bg = torch.ones(3)
    s = ro.settings_from_camera(cam_param, img_shape, bg)
    r = rasterize(gaussian_assets['mean_3d'], gaussian_assets['opacity'], colors_precomp=gaussian_assets['rgb'],
                  scales=gaussian_assets['scale'], rotations=gaussian_assets['rotation'], settings=s,
                  dL_dcolor=dL_dimg, dL_ddepth=dL_ddepth, dL_dalpha=dL_dalpha)
    out = {'img': r['color'], 'depthmap': r['depth'], 'mask': r['alpha'], 'radius': r['radii'], 'is_vis': r['radii'] > 0,
           'final_T': r['final_T'], 'n_contrib': r['n_contrib'], 'pixel_margin': r['pixel_margin'],
           'num_rendered': r['num_rendered']}
    if 'grads' in r:
        gr = r['grads']
        out['grads'] = {'mean_3d': gr['means3D'], 'scale': gr['scales'], 'rotation': gr['rotations'],
                        'opacity': gr['opacities'], 'rgb': gr['colors_precomp'], 'mean_2d': gr['means2D']}
    return out


