"""CPU tests of the host side: the C-ABI library loads and exports every declared symbol, argument
validation works without a GPU, the Python surface mirrors the reference's plugin interface, and the
product path refuses to run without a ROCm device (no CPU fallback)."""
import ctypes
import os
import re

import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import _lib, scenes

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    src = open(os.path.join(ROOT, 'include', 'exa_raster.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    return sorted(set(re.findall(r'\b(exa_(?:raster|ssim|photo|l1)_\w+)\s*\(', src)))


def test_library_exports_every_symbol_the_header_declares():
    lib = _lib.load()
    names = _declared_functions()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), n
        assert n in _lib.SIGNATURES, 'ctypes signature missing for ' + n
    assert lib.exa_raster_version() == 139
    assert [lib.exa_raster_timing_name(i) for i in range(_lib.TIMING_SLOTS)][1] == b'preprocess_fwd'


def test_settings_struct_layout_matches_c():
    # 4 ints/floats, ptr, float(+pad), 2 ptrs, int(+pad), ptr, 2 ints  on LP64
    assert ctypes.sizeof(_lib.ExaRasterSettings) == 72
    assert _lib.ExaRasterSettings.bg.offset == 16 and _lib.ExaRasterSettings.campos.offset == 56


def test_workspace_sizes():
    s = _lib.workspace_sizes(150_000, 1024, 1024, 1_000_000)
    assert s.geom_bytes == 150_000 * 64
    assert s.bin_bytes >= 29 * 1_000_000 and s.grad_bytes >= 40 * 1_000_000 + 92 * 150_000
    s2 = _lib.workspace_sizes(0, 0, 0, 0)
    assert s2.geom_bytes == 0
    with pytest.raises(RuntimeError):
        _lib.workspace_sizes(-1, 16, 16, 0)


def test_argument_validation_returns_negative_status_without_touching_the_gpu():
    lib = _lib.load()
    st = _lib.ExaRasterSettings()
    st.image_height, st.image_width, st.tanfovx, st.tanfovy = 64, 64, 0.5, 0.5
    null = ctypes.c_void_p(0)
    # NULL device pointers in the settings
    rc = lib.exa_raster_forward_bin(ctypes.byref(st), 10, 0, null, null, null, null, null, null, null, null, null,
                                    null, null)
    assert rc == -2 and b'settings' in lib.exa_raster_last_error()
    fake = ctypes.c_void_p(4096)
    st.bg = st.viewmatrix = st.projmatrix = st.campos = 4096
    # both colours and SHs missing
    rc = lib.exa_raster_forward_bin(ctypes.byref(st), 10, 0, fake, null, null, fake, fake, fake, null, fake, fake,
                                    fake, null)
    assert rc == -1 and b'exactly one' in lib.exa_raster_last_error()
    # scales without rotations
    rc = lib.exa_raster_forward_bin(ctypes.byref(st), 10, 0, fake, null, fake, fake, fake, null, null, fake, fake,
                                    fake, null)
    assert rc == -1
    # sh_M too small for the degree
    st.sh_degree = 3
    rc = lib.exa_raster_forward_bin(ctypes.byref(st), 10, 4, fake, fake, null, fake, fake, fake, null, fake, fake,
                                    fake, null)
    assert rc == -1 and b'sh_M' in lib.exa_raster_last_error()
    st.sh_degree = 0
    # image larger than the cell budget
    st.image_height = st.image_width = 40000
    rc = lib.exa_raster_forward_bin(ctypes.byref(st), 10, 0, fake, null, fake, fake, fake, fake, null, fake, fake,
                                    fake, null)
    assert rc == -1 and b'too large' in lib.exa_raster_last_error()
    with pytest.raises(RuntimeError):
        _lib.check(rc)


def test_batch_job_structs_match_c_layout():
    # LP64: pointer, 2 x int32, then 8-byte fields only
    assert ctypes.sizeof(_lib.ExaRasterForwardJob) == 8 + 8 + 7 * 8 + 8 + 2 * 8 + 8 + 8 + 3 * 8 + 8 + 8 + 8 + 8     # + keep_sorted_keys, host_header, header_tag (padded), is_vis
    assert _lib.ExaRasterForwardJob.keep_sorted_keys.offset == 136 and _lib.ExaRasterForwardJob.host_header.offset == 144
    assert _lib.ExaRasterForwardJob.capacity.offset == 104 and _lib.ExaRasterForwardJob.out_color.offset == 112
    assert ctypes.sizeof(_lib.ExaRasterBackwardJob) == 8 + 8 + 7 * 8 + 8 + 3 * 8 + 8 + 3 * 8 + 8 + 8 * 8 + 3 * 8 + 8 + 3 * 8 + 8 + 8     # + composite fields, dL_dcolor_indirect, accumulate + used_slots
    assert _lib.ExaRasterBackwardJob.grad_first.offset == 8 + 8 + 7 * 8 + 8 + 3 * 8 + 8 + 3 * 8 + 8 + 8 * 8 + 3 * 8
    assert _lib.ExaRasterBackwardJob.grad_ws.offset == 136
    lib = _lib.load()
    # argument validation of the batched entry points, no GPU touched
    jobs = (_lib.ExaRasterForwardJob * 2)()
    assert lib.exa_raster_forward_batch(jobs, 2, 0, None) == -2 and b'settings' in lib.exa_raster_last_error()
    assert lib.exa_raster_forward_batch(None, 2, 0, None) == -1
    assert lib.exa_raster_forward_batch(None, 0, 0, None) == 0
    bj = (_lib.ExaRasterBackwardJob * 1)()
    assert lib.exa_raster_backward_batch(bj, 1, 0, None) == -2
    # constant prefix: 0 <= grad_first <= P, and never together with sum_shared
    st = _lib.ExaRasterSettings()
    st.image_height, st.image_width, st.tanfovx, st.tanfovy = 64, 64, 0.5, 0.5
    st.bg = st.viewmatrix = st.projmatrix = st.campos = 4096
    b = bj[0]
    b.settings = ctypes.pointer(st)
    b.P = 10
    for name in ('means3D', 'colors_precomp', 'opacities', 'scales', 'rotations', 'radii', 'geom_ws', 'tile_ws', 'grad_ws',
                 'dL_dcolor'):
        setattr(b, name, 4096)
    b.grad_first = 11
    assert lib.exa_raster_backward_batch(bj, 1, 0, None) == -1 and b'grad_first' in lib.exa_raster_last_error()
    b.grad_first = -1
    assert lib.exa_raster_backward_batch(bj, 1, 0, None) == -1
    b.grad_first = 3
    assert lib.exa_raster_backward_batch(bj, 1, 1, None) == -1 and b'sum_shared' in lib.exa_raster_last_error()
    b.grad_first = 10                     # nothing trainable: a no-op that touches no pointer
    assert lib.exa_raster_backward_batch(bj, 1, 0, None) == 0
    assert lib.exa_raster_read_header_async(None, None, None) == -2
    assert lib.exa_raster_read_header_full_async(None, None, None) == -2
    assert lib.exa_raster_camera_block(None, None, None, None, None, None, None, 0.0, 0.0, None, 0, None) == -2
    assert lib.exa_raster_host_device_pointer(None, None) == -2
    # two jobs of one batch updating the same densification statistics: rejected (-5) unless sum_shared with the same
    # three arrays in every job
    bj2 = (_lib.ExaRasterBackwardJob * 2)()
    for q in bj2:
        q.settings = ctypes.pointer(st)
        q.P = 10
        for name in ('means3D', 'colors_precomp', 'opacities', 'scales', 'rotations', 'radii', 'geom_ws', 'tile_ws', 'grad_ws',
                     'dL_dcolor'):
            setattr(q, name, 4096)
        q.grad_first = 10
        q.densify_grad_accum = 8192
    assert lib.exa_raster_backward_batch(bj2, 2, 0, None) == -5 and b'densification' in lib.exa_raster_last_error()
    bj2[1].densify_grad_accum = 16384
    assert lib.exa_raster_backward_batch(bj2, 2, 0, None) == 0


def test_header_status_and_struct():
    lib = _lib.load()
    assert ctypes.sizeof(_lib.ExaRasterHeader) == 28
    h = _lib.ExaRasterHeader()
    h.num_rendered, h.overflow = 4096, 0
    assert lib.exa_raster_header_status(ctypes.byref(h)) == 0
    h.overflow = 1
    assert lib.exa_raster_header_status(ctypes.byref(h)) == -4 and b'4096' in lib.exa_raster_last_error()
    assert lib.exa_raster_header_status(None) == -2


def test_python_surface_matches_the_reference_plugin():
    fields = exa.GaussianRasterizationSettings._fields
    assert fields == ('image_height', 'image_width', 'tanfovx', 'tanfovy', 'bg', 'scale_modifier', 'viewmatrix',
                      'projmatrix', 'sh_degree', 'campos', 'prefiltered', 'debug')      # module.py:609-622
    import inspect
    sig = inspect.signature(exa.GaussianRasterizer.forward)
    assert list(sig.parameters)[1:] == ['means3D', 'means2D', 'opacities', 'shs', 'colors_precomp', 'scales',
                                        'rotations', 'cov3D_precomp']
    sig = inspect.signature(exa.GaussianRenderer.forward)
    assert list(sig.parameters)[1:5] == ['gaussian_assets', 'img_shape', 'cam_param', 'bg']   # module.py:592 (+ optional extras)


def _settings():
    H = W = 32
    from exavatar_release_amd.camera import make_raster_matrices
    tanx, tany, view, proj, campos = make_raster_matrices(scenes.neutral_camera(H, W), (H, W))
    return exa.GaussianRasterizationSettings(H, W, tanx, tany, torch.ones(3), 1.0, view, proj, 0, campos, False, False)


def test_rasterizer_rejects_bad_argument_combinations_like_upstream():
    r = exa.GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(Exception, match='SHs or precomputed colors'):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), shs=None, colors_precomp=None,
          scales=torch.ones(4, 3), rotations=torch.ones(4, 4))
    with pytest.raises(Exception, match='scale/rotation pair or precomputed 3D covariance'):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=m, scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4), cov3D_precomp=torch.ones(4, 6))


def test_product_path_has_no_cpu_fallback():
    r = exa.GaussianRasterizer(_settings())
    m = torch.zeros(4, 3)
    with pytest.raises(RuntimeError, match='no CPU path'):
        r(means3D=m, means2D=m, opacities=torch.ones(4, 1), colors_precomp=m, scales=torch.ones(4, 3),
          rotations=torch.ones(4, 4))
    # and nothing in the package imports the oracle
    pkg = os.path.join(ROOT, 'exavatar_release_amd')
    pat = re.compile(r'^\s*(from\s+oracle\b|import\s+oracle\b)|import_module\([\'"]oracle', re.M)
    for fn in os.listdir(pkg):
        if fn.endswith('.py'):
            assert not pat.search(open(os.path.join(pkg, fn)).read()), fn


def test_compiled_autograd_node_is_built_against_this_abi_and_declines_what_it_cannot_run():
    """``_exa_torch`` (csrc/torch_binding.cpp): host C++ over the C ABI.  It must be in the tree (built by
    ``__graft_entry__.build()``), carry the ABI version of the header it was compiled against, and hand a call it cannot run
    -- here: before ``init`` (no GPU in this container), CPU tensors -- back to the Python node by returning None."""
    from exavatar_release_amd import _exa_torch
    assert _exa_torch.abi_version == _lib.load().exa_raster_version()
    m = torch.zeros(4, 3)
    assert _exa_torch.rasterize(_settings(), m, m, None, m, torch.ones(4, 1), torch.ones(4, 3), torch.ones(4, 4), None, 64, True,
                                False, None) is None
    assert _exa_torch.last_decline() != ''
    src = open(os.path.join(ROOT, 'exavatar_release_amd', 'csrc', 'torch_binding.cpp')).read()
    assert '__global__' not in src and 'hipLaunchKernelGGL' not in src          # host code only: the kernels live behind the C ABI
    assert 'oracle' not in src


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    monkeypatch.setattr(_lib, '_lib', None)
    monkeypatch.setattr(_lib, 'LIB_PATH', str(tmp_path / 'nope.so'))
    with pytest.raises(RuntimeError, match='no CPU fallback'):
        _lib.load()


def test_scenes_are_seeded_and_shaped():
    a1, shp, cam = scenes.make_config('c1')
    a2, _, _ = scenes.make_config('c1')
    assert shp == (256, 256) and a1['mean_3d'].shape == (10_000, 3)
    for k in a1:
        assert torch.equal(a1[k], a2[k]) and a1[k].dtype == torch.float32 and a1[k].is_contiguous()
    b = scenes.dist_b_avatar(5000, seed=1)
    assert torch.all(b['opacity'] == 1) and torch.all(b['rotation'][:, 0] == 1)
    assert torch.all(b['scale'][:, 0] == b['scale'][:, 1])
    sh = scenes.sh_from_rgb(a1['rgb'][:7], 3)
    assert sh.shape == (7, 16, 3)


def test_dev_layout_helper_matches_library():
    """tools/_layout.py (used by the developer probes) mirrors carve_tile_ws(): same total size as the C ABI reports."""
    import importlib.util
    import os
    from exavatar_release_amd import _lib
    spec = importlib.util.spec_from_file_location(
        '_layout', os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', '_layout.py'))
    lay = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lay)
    for P, W, H in ((150000, 1024, 1024), (1000, 540, 960), (0, 64, 64), (70001, 1920, 1080)):
        assert lay.tile_offsets(P, W, H)['total'] == int(_lib.workspace_sizes(P, W, H, 0).tile_bytes)


def test_renderer_plumbing_of_sh_assets(monkeypatch):
    """GaussianRenderer with assets that carry SH coefficients instead of rgb: the shs path of the rasterizer with the
    right degree, the reference's call otherwise (module.py:609-640) -- checked on the arguments, no GPU needed."""
    from exavatar_release_amd import renderer as rn
    seen = {}

    def fake(means3D, means2D, sh, colors_precomp, opacities, scales, rotations, cov3Ds_precomp, raster_settings,
             densify_stats=None):
        seen.update(sh=sh, col=colors_precomp, deg=raster_settings.sh_degree, m2=means2D, cov=cov3Ds_precomp,
                    op=opacities, sc=scales, rot=rotations, m3=means3D)
        H, W = raster_settings.image_height, raster_settings.image_width
        return torch.zeros(3, H, W), torch.ones(means3D.shape[0], dtype=torch.int32), torch.zeros(1, H, W), torch.zeros(1, H, W)
    monkeypatch.setattr(rn, 'rasterize_gaussians', fake)
    a = scenes.dist_a_random(50, 32, 32, seed=1)
    cam = scenes.neutral_camera(32, 32)
    out = rn.GaussianRenderer()(a, (32, 32), cam)
    assert seen['sh'] is None and seen['col'] is a['rgb'] and seen['deg'] == 0 and seen['cov'] is None
    assert seen['m3'] is a['mean_3d'] and seen['op'] is a['opacity'] and seen['sc'] is a['scale'] and seen['rot'] is a['rotation']
    assert out['mean_2d'] is seen['m2'] and out['mean_2d'].requires_grad and out['mean_2d'].shape == (50, 3)
    assert set(out) == {'img', 'depthmap', 'mask', 'mean_2d', 'is_vis', 'radius'} and bool(out['is_vis'].all())
    sh = scenes.sh_from_rgb(a['rgb'], 2)
    b = {k: v for k, v in a.items() if k != 'rgb'}
    b['sh'] = sh
    rn.GaussianRenderer()(b, (32, 32), cam)
    assert seen['sh'] is sh and seen['col'] is None and seen['deg'] == 2
    b['sh_degree'] = 1                         # evaluate fewer bands than stored
    rn.GaussianRenderer()(b, (32, 32), cam)
    assert seen['deg'] == 1
    with pytest.raises(ValueError):
        rn.GaussianRenderer()({**{k: v for k, v in a.items() if k != 'rgb'}, 'sh': sh[:, :5]}, (32, 32), cam)
    # the constant prefix must carry the same kind of colour input
    with pytest.raises(ValueError, match='same colour input'):
        rn._raster_job(b, (32, 32), cam, None, None, a)
    j = rn._raster_job(b, (32, 32), cam, None, None, b)
    assert j['frozen']['shs'] is sh and j['frozen']['colors_precomp'] is None


def test_densify_and_prune_screen_space_criterion_and_reference_quirk():
    """``densify_and_prune`` (reference avatar/common/nets/module.py:159-240): the reference's screen-space prune reads a
    ``radius_max`` its own ``densify()`` has just zeroed, so it never fires (default, topology as the reference's); with
    ``reference_radius_reset=False`` the surviving originals keep their radii and the big ones go."""
    from exavatar_release_amd.densify import densify_and_prune
    g = torch.Generator().manual_seed(3)
    P = 40

    def fresh():
        params = {'mean': torch.nn.Parameter(torch.randn(P, 3, generator=g)),
                  'scale': torch.nn.Parameter(torch.full((P, 3), -5.0)),
                  'rotation': torch.nn.Parameter(torch.tensor([[1.0, 0, 0, 0]]).repeat(P, 1)),
                  'opacity': torch.nn.Parameter(torch.full((P, 1), 2.0)),
                  'rgb': torch.nn.Parameter(torch.rand(P, 3, generator=g))}
        opt = torch.optim.Adam([{'params': [p]} for p in params.values()], lr=1e-3)
        sum(p.sum() for p in params.values()).backward()
        opt.step()
        return params, opt
    accum, cnt = torch.zeros(P, 1), torch.ones(P, 1)
    accum[:5] = 1.0                                              # five hot, small Gaussians: cloned
    radius_max = torch.zeros(P)
    radius_max[10:14] = 50.0                                     # four cold ones that covered > 20 px on screen
    kw = dict(grad_thr=0.5, extent=4.0, prune_big=True, radius_max=radius_max, screen_size_max=20, generator=g)
    params, opt = fresh()
    new, n_c, n_s, n_p = densify_and_prune(params, opt, accum, cnt, **kw)
    assert (n_c, n_s, n_p) == (5, 0, 0) and new['mean'].shape[0] == P + 5          # the reference's quirk: nothing pruned
    params, opt = fresh()
    new, n_c, n_s, n_p = densify_and_prune(params, opt, accum, cnt, reference_radius_reset=False, **kw)
    assert (n_c, n_s, n_p) == (5, 0, 4) and new['mean'].shape[0] == P + 5 - 4
    keep = torch.ones(P, dtype=torch.bool)
    keep[10:14] = False
    assert torch.equal(new['mean'][:P - 4], params['mean'][keep])
    st = opt.state[new['mean']]
    assert st['exp_avg'].shape[0] == P + 1 and bool((st['exp_avg'][P - 4:] == 0).all())   # new rows: zero optimizer state


def test_static_render_rejects_what_it_cannot_drive():
    """StaticRender (exavatar_release_amd/static.py) validates before it touches the library: CPU tensors, both colour inputs."""
    P = 8
    z = lambda *s: torch.zeros(*s)     # noqa: E731
    with pytest.raises(ValueError, match='exactly one'):
        exa.StaticRender(z(P, 3), z(P, 1), z(P, 3), z(P, 4), image_size=(16, 16), capacity=64)
    with pytest.raises(ValueError, match='exactly one'):
        exa.StaticRender(z(P, 3), z(P, 1), z(P, 3), z(P, 4), colors_precomp=z(P, 3), shs=z(P, 1, 3), image_size=(16, 16), capacity=64)
    with pytest.raises(ValueError, match='on a GPU'):
        exa.StaticRender(z(P, 3), z(P, 1), z(P, 3), z(P, 4), colors_precomp=z(P, 3), image_size=(16, 16), capacity=64)
    assert 'StaticRender' in exa.__all__ and 'required_capacity' in exa.__all__


def test_header_compiles_as_plain_c_and_cxx_and_links_against_the_library(tmp_path):
    """include/exa_raster.h is the drop-in boundary for NATIVE hosts: it has to compile as C99 and as C++11
    without torch, HIP or any other header of ours, and a C program that takes the address of every function it declares
    has to link against libexa_raster.so (no compute call: there is no GPU here)."""
    import shutil
    import subprocess
    if shutil.which('gcc') is None:
        pytest.skip('no gcc')
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, 'include', 'exa_raster.h')).read()
    names = sorted(set(re.findall(r'\b(exa_(?:raster|l1|photo|ssim)_\w+)\s*\(', hdr)))
    assert set(names) == set(_lib.SIGNATURES), 'binding and header disagree on the exported functions'
    src = tmp_path / 'host.c'
    src.write_text('#include "exa_raster.h"\n#include <stdio.h>\nint main(void) {\n  void* f[] = {%s};\n'
                   '  ExaRasterForwardJob fj; ExaRasterBackwardJob bj; ExaRasterComposeJob cj; (void)fj; (void)bj; (void)cj;\n'
                   '  printf("%%d %%d\\n", (int)(sizeof f / sizeof f[0]), exa_raster_version());\n  return 0;\n}\n'
                   % ', '.join('(void*)' + n for n in names))
    inc = ['-I', os.path.join(root, 'include')]
    subprocess.run(['gcc', '-std=c99', '-Wall', '-Wextra', '-Werror', '-Wno-pedantic', '-fsyntax-only'] + inc + [str(src)], check=True)
    if shutil.which('g++'):
        subprocess.run(['g++', '-std=c++11', '-Wall', '-Wextra', '-Werror', '-fsyntax-only', '-x', 'c++'] + inc + [str(src)], check=True)
    lib = os.path.join(root, 'exavatar_release_amd', 'libexa_raster.so')
    exe = tmp_path / 'host'
    subprocess.run(['gcc', '-std=c99'] + inc + [str(src), lib, '-Wl,-rpath,' + os.path.dirname(lib), '-Wl,--allow-shlib-undefined',
                                                 '-o', str(exe)], check=True)
    env = dict(os.environ, LD_LIBRARY_PATH='/opt/rocm/lib:' + os.environ.get('LD_LIBRARY_PATH', ''))
    out = subprocess.run([str(exe)], check=True, capture_output=True, text=True, env=env).stdout.split()
    assert int(out[0]) == len(names) and int(out[1]) == _lib.load().exa_raster_version()


def test_native_host_example_compiles_and_links(tmp_path):
    """examples/native_host.c (plain C99 + HIP runtime API over include/exa_raster.h; run on the GPU by
    tests/test_gpu_native_host.py) must build with gcc against the in-tree library -- no compute here."""
    import shutil
    import subprocess
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    if shutil.which('gcc') is None or not os.path.exists(os.path.join(rocm, 'include', 'hip', 'hip_runtime_api.h')):
        pytest.skip('no gcc / HIP headers')
    lib = os.path.join(ROOT, 'exavatar_release_amd', 'libexa_raster.so')
    subprocess.run(['gcc', '-std=c99', '-O1', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
                    '-I', os.path.join(rocm, 'include'), os.path.join(ROOT, 'examples', 'native_host.c'), lib,
                    '-L', os.path.join(rocm, 'lib'), '-lamdhip64', '-Wl,-rpath,' + os.path.join(rocm, 'lib'),
                    '-o', str(tmp_path / 'native_host')], check=True)
