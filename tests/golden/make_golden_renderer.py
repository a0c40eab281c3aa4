"""Generates tests/golden/ref_renderer.npz by EXECUTING the reference's own render plugin.

``class GaussianRenderer`` (/root/reference/avatar/common/nets/module.py:588-647) is the calling code of the path this
package replaces.  ``module.py`` cannot be imported here (pytorch3d, smplx, the training config and a ``.cuda()`` default
argument at import time), so the class is cut out of the file with ``ast`` and exec'd UNCHANGED -- as
``make_golden_ssim.py`` does for the loss classes -- in a namespace that holds what ``module.py`` imports for it:

* ``get_fov`` / ``get_view_matrix`` / ``get_proj_matrix``: the reference's own ``utils/transforms.py``, imported;
* ``GaussianRasterizationSettings`` / ``GaussianRasterizer``: a recording stand-in for the third-party extension whose
  ``forward`` is the CPU oracle (``oracle/raster_oracle.py``) -- the extension's source is not in the reference tree
  (SURVEY.md section 0.1), so the rasterizer's own numbers stay "parity unpinned"; what this fixture pins is everything
  the reference's CALLING CODE decides: the twelve settings fields it builds (tan(fov) floats, the transposed view matrix,
  ``view^T proj^T``, the camera position from a matrix inverse), the keyword arguments it passes, the order it unpacks the
  four results in and the six-key dict it returns, incl. ``mean_2d`` as a leaf whose ``.grad`` the backward fills.

``Tensor.cuda()`` is the identity for the duration (no GPU in this container).  Nothing of the reference's text is written
anywhere: only inputs and outputs travel.  Run from the repo root:  python tests/golden/make_golden_renderer.py
"""
import ast
import importlib.util
import os
import sys

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))
REF_MODULE = '/root/reference/avatar/common/nets/module.py'
REF_TRANSFORMS = '/root/reference/avatar/common/utils/transforms.py'


def reference_renderer_class(namespace):
    """exec ``class GaussianRenderer`` of the reference, unchanged, in ``namespace``; returns the class."""
    src = open(REF_MODULE).read()
    lines = src.splitlines()
    for node in ast.parse(src).body:
        if isinstance(node, ast.ClassDef) and node.name == 'GaussianRenderer':
            exec('\n'.join(lines[node.lineno - 1: node.end_lineno]), namespace)
            return namespace['GaussianRenderer']
    raise RuntimeError('class GaussianRenderer not found in ' + REF_MODULE)


def reference_transforms():
    spec = importlib.util.spec_from_file_location('ref_transforms', REF_TRANSFORMS)
    ref = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ref)
    return ref


def cases():
    """(name, assets, (H, W), cam_param, bg): a C1-like random scene and a 540-pixel-wide avatar view (33.75 tiles)."""
    from exavatar_release_amd import scenes
    g = torch.Generator().manual_seed(77)
    H, W = 96, 128
    yield ('c1', scenes.dist_a_random(1500, H, W, seed=4, focal=140.0), (H, W),
           scenes.ring_camera(H, W, 2, 9, radius=0.4, center=(0.0, 0.0, 4.0), focal=140.0), torch.rand(3, generator=g))
    H, W = 72, 540
    cam = scenes.ring_camera(H, W, 5, 24, focal=300.0)
    cam['princpt'] = torch.tensor([W / 2.0 + 7.0, H / 2.0 - 3.0])          # off-centre: accepted and ignored (transforms.py:43-64)
    yield ('w540', scenes.dist_b_avatar(2500, seed=6), (H, W), cam, None)        # None: the reference's default white background


def main():
    from oracle import raster_oracle as ro
    torch.set_num_threads(1)
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        ref_t = reference_transforms()
        seen = {}

        class GaussianRasterizer(nn.Module):
            """Stand-in for the third-party class (module.py:623): records what the calling code hands over, renders with the
            CPU oracle."""
            def __init__(self, raster_settings):
                super().__init__()
                self.raster_settings = raster_settings

            def forward(self, **kw):
                seen['kwargs'] = sorted(kw)
                seen['settings'] = self.raster_settings
                s = self.raster_settings
                so = ro.OracleSettings(image_height=int(s.image_height), image_width=int(s.image_width), tanfovx=s.tanfovx,
                                       tanfovy=s.tanfovy, bg=s.bg, scale_modifier=s.scale_modifier, viewmatrix=s.viewmatrix,
                                       projmatrix=s.projmatrix, sh_degree=s.sh_degree, campos=s.campos, prefiltered=s.prefiltered,
                                       debug=s.debug)
                res = ro.rasterize(settings=so, return_aux=True, **kw)
                seen['aux'] = res[4]
                return res[:4]

        from exavatar_release_amd.rasterizer import GaussianRasterizationSettings      # (a 12-field NamedTuple, module.py:609-622)
        ns = {'torch': torch, 'nn': nn, 'get_fov': ref_t.get_fov, 'get_view_matrix': ref_t.get_view_matrix,
              'get_proj_matrix': ref_t.get_proj_matrix, 'GaussianRasterizationSettings': GaussianRasterizationSettings,
              'GaussianRasterizer': GaussianRasterizer}
        renderer = reference_renderer_class(ns)()
        out = {}
        names = []
        for name, assets, shape, cam, bg in cases():
            g = torch.Generator().manual_seed(len(name))
            G = torch.randn(3, shape[0], shape[1], generator=g)
            a = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
            res = renderer(a, shape, cam) if bg is None else renderer(a, shape, cam, bg)
            assert sorted(res) == ['depthmap', 'img', 'is_vis', 'mask', 'mean_2d', 'radius']
            (res['img'] * G).sum().backward()
            s = seen['settings']
            p = name + '_'
            for k, v in assets.items():
                out[p + 'asset_' + k] = v.numpy()
            for k, v in cam.items():
                out[p + 'cam_' + k] = torch.as_tensor(v).numpy()
            out[p + 'shape'] = np.array(shape)
            out[p + 'bg_given'] = np.array(bg is not None)
            out[p + 'bg'] = s.bg.numpy()
            out[p + 'G'] = G.numpy()
            out[p + 'kwargs'] = np.array(seen['kwargs'])
            out[p + 'settings_scalars'] = np.array([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree,
                                                    float(s.prefiltered), float(s.debug)], dtype=np.float64)
            out[p + 'settings_tan_is_float'] = np.array(isinstance(s.tanfovx, float) and isinstance(s.tanfovy, float))
            out[p + 'viewmatrix'] = s.viewmatrix.numpy()
            out[p + 'projmatrix'] = s.projmatrix.numpy()
            out[p + 'campos'] = s.campos.numpy()
            out[p + 'img'] = res['img'].detach().numpy()
            out[p + 'depthmap'] = res['depthmap'].detach().numpy()
            out[p + 'mask'] = res['mask'].detach().numpy()
            out[p + 'radius'] = res['radius'].numpy()
            out[p + 'is_vis'] = res['is_vis'].numpy()
            out[p + 'mean_2d_is_leaf'] = np.array(res['mean_2d'].is_leaf and res['mean_2d'].requires_grad)
            out[p + 'mean_2d_grad'] = res['mean_2d'].grad.numpy()
            out[p + 'ambiguous'] = ro.ambiguous_pixel_mask(seen['aux'], shape[0], shape[1]).numpy()
            for k in assets:
                out[p + 'grad_' + k] = a[k].grad.numpy()
            names.append(name)
        out['cases'] = np.array(names)
    finally:
        torch.Tensor.cuda = orig_cuda
    path = os.path.join(HERE, 'ref_renderer.npz')
    np.savez_compressed(path, **out)
    print('wrote', path, os.path.getsize(path), 'bytes;', names)


if __name__ == '__main__':
    main()
