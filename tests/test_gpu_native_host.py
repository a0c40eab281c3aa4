"""A NATIVE host of the C ABI: ``examples/native_host.c`` -- plain C99 + the HIP runtime API, no torch, no Python -- compiled with
gcc on the GPU box, given a scene file, runs upstream's two-stage forward, the backward and the fused forward on buffers and a
stream it owns; what it writes is held against the Python surface bit for bit.  The drop-in boundary is the C header alone."""
import os
import shutil
import struct
import subprocess

import numpy as np
import pytest
import torch

import exavatar_release_amd as exa
from exavatar_release_amd import scenes
from exavatar_release_amd.camera import make_raster_matrices

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_native_c_host_equals_the_python_surface(tmp_path):
    assert torch.cuda.is_available(), 'GPU tests need a ROCm device'
    if shutil.which('gcc') is None:
        pytest.skip('no gcc on this box')
    rocm = os.environ.get('ROCM_PATH', '/opt/rocm')
    lib = os.path.join(ROOT, 'exavatar_release_amd', 'libexa_raster.so')
    exe = str(tmp_path / 'native_host')
    subprocess.run(['gcc', '-std=c99', '-O1', '-Wall', '-Werror', '-D__HIP_PLATFORM_AMD__', '-I', os.path.join(ROOT, 'include'),
                    '-I', os.path.join(rocm, 'include'), os.path.join(ROOT, 'examples', 'native_host.c'), lib,
                    '-L', os.path.join(rocm, 'lib'), '-lamdhip64', '-Wl,-rpath,' + os.path.join(rocm, 'lib'),
                    '-Wl,-rpath,' + os.path.dirname(lib), '-o', exe], check=True)
    dev = torch.device('cuda:0')
    H, W, P, f = 120, 152, 4000 + 3, 210.0
    a = scenes.dist_b_avatar(P, seed=13)
    cam = scenes.ring_camera(H, W, 5, 24, focal=f)
    tanx, tany, view, proj, campos = make_raster_matrices(cam, (H, W))
    g = torch.Generator().manual_seed(3)
    bg = torch.rand(3, generator=g)
    G = torch.randn(3, H, W, generator=g)
    scene = tmp_path / 'scene.bin'
    with open(scene, 'wb') as fh:
        fh.write(struct.pack('<3i2f', P, H, W, tanx, tany))
        for t in (bg, view, proj, campos, a['mean_3d'], a['scale'], a['rotation'], a['opacity'], a['rgb'], G):
            fh.write(t.contiguous().numpy().astype('<f4').tobytes())
    out = tmp_path / 'out.bin'
    r = subprocess.run([exe, str(scene), str(out)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.stdout, r.stderr)
    assert 'native_host: ABI' in r.stdout
    raw = np.fromfile(out, dtype='<f4')
    HW = H * W
    sizes = [3 * HW, HW, HW, P, 3 * P, 3 * P, 3 * P, P, 3 * P, 4 * P, 3 * HW, HW, HW]
    assert raw.size == sum(sizes)
    parts, o = [], 0
    for n in sizes:
        parts.append(raw[o:o + n])
        o += n
    color, depth, alpha, radii, d3, d2, dc, dop, dsc, drot, color2, depth2, alpha2 = parts
    # the same render through the Python surface (exact mode = the same two-stage protocol)
    exa.config.mode = 'exact'
    leaves = {k: v.to(dev).requires_grad_(True) for k, v in a.items()}
    m2 = torch.zeros(P, 3, device=dev, requires_grad=True)
    st = exa.GaussianRasterizationSettings(H, W, tanx, tany, bg.to(dev), 1.0, view.to(dev).contiguous(), proj.to(dev).contiguous(), 0,
                                           campos.to(dev).contiguous(), False, False)
    c, r_, d, al = exa.GaussianRasterizer(st)(means3D=leaves['mean_3d'], means2D=m2, opacities=leaves['opacity'],
                                              colors_precomp=leaves['rgb'], scales=leaves['scale'], rotations=leaves['rotation'])
    (c * G.to(dev)).sum().backward()
    eq = lambda x, t: np.array_equal(x, t.detach().cpu().numpy().reshape(-1))        # noqa: E731
    assert eq(color, c) and eq(depth, d) and eq(alpha, al)
    assert np.array_equal(radii.view('<i4'), r_.cpu().numpy())
    assert eq(d3, leaves['mean_3d'].grad) and eq(d2, m2.grad) and eq(dc, leaves['rgb'].grad) and eq(dop, leaves['opacity'].grad)
    assert eq(dsc, leaves['scale'].grad) and eq(drot, leaves['rotation'].grad)
    assert eq(color2, c) and eq(depth2, d) and eq(alpha2, al)        # the fused call with the measured capacity: the same images
