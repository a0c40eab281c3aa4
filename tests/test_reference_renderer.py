"""The reference's own calling code at the boundary: ``class GaussianRenderer`` (reference
avatar/common/nets/module.py:588-647).  ``tests/golden/ref_renderer.npz`` holds what that class -- exec'd unchanged by
``tests/golden/make_golden_renderer.py`` -- builds and returns for two scenes (settings tuple, keyword arguments, the six-key
output dict, ``mean_2d.grad``) with the CPU oracle behind it.  Here: this package's renderer builds the same settings bit for
bit, the oracle's twin of the renderer reproduces the outputs, and -- where the reference tree is present (the build
container) -- the reference's class runs UNMODIFIED against this package's names and returns what ``exa.GaussianRenderer``
returns."""
import os

import numpy as np
import pytest
import torch
import torch.nn as nn

import exavatar_release_amd as exa
from exavatar_release_amd import camera, rasterizer, renderer
from oracle import raster_oracle as ro

REF_MODULE = '/root/reference/avatar/common/nets/module.py'


@pytest.fixture(scope='module')
def golden(golden_dir):
    return np.load(os.path.join(golden_dir, 'ref_renderer.npz'))


@pytest.fixture(autouse=True)
def _one_thread():
    """The fixture was generated with one CPU thread (make_golden_renderer.py): the oracle's reductions round differently when
    PyTorch splits them over threads, and the comparisons below are bitwise."""
    n = torch.get_num_threads()
    torch.set_num_threads(1)
    yield
    torch.set_num_threads(n)


def _case(golden, name):
    p = name + '_'
    assets = {k[len(p) + 6:]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith(p + 'asset_')}
    cam = {k[len(p) + 4:]: torch.from_numpy(golden[k]) for k in golden.files if k.startswith(p + 'cam_')}
    shape = tuple(int(v) for v in golden[p + 'shape'])
    bg = torch.from_numpy(golden[p + 'bg']) if bool(golden[p + 'bg_given']) else None
    return assets, shape, cam, bg


def _oracle_rasterize(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
                      densify_stats=None):
    """``rasterizer.rasterize_gaussians`` with the CPU oracle behind it (this container has no GPU)."""
    s = raster_settings
    so = ro.OracleSettings(image_height=int(s.image_height), image_width=int(s.image_width), tanfovx=s.tanfovx, tanfovy=s.tanfovy,
                           bg=s.bg, scale_modifier=s.scale_modifier, viewmatrix=s.viewmatrix, projmatrix=s.projmatrix,
                           sh_degree=s.sh_degree, campos=s.campos, prefiltered=s.prefiltered, debug=s.debug)
    return ro.rasterize(means3D=means3D, means2D=means2D, opacities=opacities, shs=sh, colors_precomp=colors_precomp, scales=scales,
                        rotations=rotations, cov3D_precomp=cov3Ds_precomp, settings=so)


@pytest.mark.parametrize('name', ['c1', 'w540'])
def test_this_packages_renderer_builds_the_reference_callers_settings(golden, name):
    assets, shape, cam, bg = _case(golden, name)
    p = name + '_'
    job = renderer._raster_job(assets, shape, cam, bg)
    s = job['raster_settings']
    assert type(s).__name__ == 'GaussianRasterizationSettings' and s._fields == (
        'image_height', 'image_width', 'tanfovx', 'tanfovy', 'bg', 'scale_modifier', 'viewmatrix', 'projmatrix', 'sh_degree',
        'campos', 'prefiltered', 'debug')
    want = golden[p + 'settings_scalars']
    got = np.array([s.image_height, s.image_width, s.tanfovx, s.tanfovy, s.scale_modifier, s.sh_degree, float(s.prefiltered),
                    float(s.debug)], dtype=np.float64)
    assert np.array_equal(got, want)                                        # tan(fov / 2): the same Python floats
    assert bool(golden[p + 'settings_tan_is_float']) and isinstance(s.tanfovx, float) and isinstance(s.tanfovy, float)
    assert np.array_equal(s.viewmatrix.numpy(), golden[p + 'viewmatrix'])
    assert np.array_equal(s.projmatrix.numpy(), golden[p + 'projmatrix'])
    assert np.array_equal(s.campos.numpy(), golden[p + 'campos'])
    assert np.array_equal(s.bg.numpy(), golden[p + 'bg'])                  # incl. the default white one
    # the keyword arguments the reference passes (module.py:632-640) are the ones this package's forward takes
    passed = sorted(k for k in job if k not in ('raster_settings', 'densify_stats', 'frozen'))
    assert passed == list(golden[p + 'kwargs'])
    assert job['means2D'].is_leaf and job['means2D'].requires_grad and bool(golden[p + 'mean_2d_is_leaf'])
    assert tuple(job['means2D'].shape) == (assets['mean_3d'].shape[0], 3) and float(job['means2D'].detach().abs().max()) == 0.0


@pytest.mark.parametrize('name', ['c1', 'w540'])
def test_the_oracles_renderer_twin_reproduces_the_reference_callers_outputs(golden, name):
    assets, shape, cam, bg = _case(golden, name)
    p = name + '_'
    a = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    out = ro.render(a, shape, cam, bg)
    (out['img'] * torch.from_numpy(golden[p + 'G'])).sum().backward()
    assert sorted(out) == ['depthmap', 'img', 'is_vis', 'mask', 'mean_2d', 'radius']
    for k in ('img', 'depthmap', 'mask', 'radius', 'is_vis'):
        assert np.array_equal(out[k].detach().numpy(), golden[p + k]), k
    assert np.array_equal(out['mean_2d'].grad.numpy(), golden[p + 'mean_2d_grad'])
    for k in assets:
        assert np.array_equal(a[k].grad.numpy(), golden[p + 'grad_' + k]), k


@pytest.mark.skipif(not os.path.exists(REF_MODULE), reason='the reference tree is only present in the build container')
@pytest.mark.parametrize('name', ['c1', 'w540'])
def test_reference_renderer_source_runs_unmodified_against_this_package(golden, name, monkeypatch):
    """"``GaussianRenderer.forward`` from module.py:592-647 runs unmodified against it".  The class
    text is read from the reference tree at test time (never stored in this repository) and exec'd with the five names
    ``module.py`` imports for it bound to THIS package -- ``GaussianRasterizationSettings``, ``GaussianRasterizer``,
    ``get_fov``, ``get_view_matrix``, ``get_proj_matrix`` -- the rasterizer's device call replaced by the CPU oracle for
    both sides (no GPU here; on the GPU box tests/test_gpu_reference_renderer.py holds the HIP path against the fixture)."""
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden'))
    from make_golden_renderer import reference_renderer_class
    monkeypatch.setattr(rasterizer, 'rasterize_gaussians', _oracle_rasterize)
    monkeypatch.setattr(renderer, 'rasterize_gaussians', _oracle_rasterize)
    monkeypatch.setattr(torch.Tensor, 'cuda', lambda self, *a, **k: self)
    ns = {'torch': torch, 'nn': nn, 'get_fov': camera.get_fov, 'get_view_matrix': camera.get_view_matrix,
          'get_proj_matrix': camera.get_proj_matrix, 'GaussianRasterizationSettings': exa.GaussianRasterizationSettings,
          'GaussianRasterizer': exa.GaussianRasterizer}
    ref_renderer = reference_renderer_class(ns)()
    assets, shape, cam, bg = _case(golden, name)
    p = name + '_'
    G = torch.from_numpy(golden[p + 'G'])
    res = {}
    for who, rend in (('reference', ref_renderer), ('package', exa.GaussianRenderer())):
        a = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
        out = rend(a, shape, cam) if bg is None else rend(a, shape, cam, bg)
        (out['img'] * G).sum().backward()
        res[who] = (out, a)
    (o_ref, a_ref), (o_pkg, a_pkg) = res['reference'], res['package']
    assert sorted(o_ref) == sorted(o_pkg) == ['depthmap', 'img', 'is_vis', 'mask', 'mean_2d', 'radius']
    for k in ('img', 'depthmap', 'mask', 'radius', 'is_vis'):
        assert torch.equal(o_ref[k], o_pkg[k]), k
        assert np.array_equal(o_ref[k].detach().numpy(), golden[p + k]), k          # ... and both are the fixture
    assert torch.equal(o_ref['mean_2d'].grad, o_pkg['mean_2d'].grad)
    assert np.array_equal(o_pkg['mean_2d'].grad.numpy(), golden[p + 'mean_2d_grad'])
    for k in assets:
        assert torch.equal(a_ref[k].grad, a_pkg[k].grad), k
