"""Generates the committed golden fixtures.  Run from the repo root IN THE BUILD CONTAINER:

    python tests/golden/make_golden.py

* ref_transforms.npz -- outputs of the REFERENCE's own in-tree helpers on the hot path, produced by
  importing /root/reference/avatar/common/utils/transforms.py (get_fov, get_view_matrix,
  get_proj_matrix: transforms.py:38-70; eval_sh, RGB2SH: transforms.py:112-170;
  get_covariance_matrix: transforms.py:72-80).  The reference hard-codes ``.cuda()``; this script
  runs it on the CPU by making ``Tensor.cuda`` the identity for the duration of the import/calls.
  These pin camera.py and the oracle's SH / covariance restatements to the reference itself.
* oracle_small.npz   -- a small seeded scene with the oracle's float32 outputs and gradients: a
  regression pin for the oracle and a portable golden vector for the GPU parity tests
  (/root/reference does not exist on the GPU box).
"""
import importlib.util
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
HERE = os.path.dirname(os.path.abspath(__file__))


def reference_fixtures():
    orig_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **k: self
    try:
        spec = importlib.util.spec_from_file_location(
            'ref_transforms', '/root/reference/avatar/common/utils/transforms.py')
        ref = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(ref)
        g = torch.Generator().manual_seed(11)
        out = {}
        cams = []
        for i, (H, W, fx, fy) in enumerate([(1024, 1024, 1500.0, 1500.0), (540, 960, 791.0, 805.5),
                                            (960, 540, 1406.25, 1400.0), (256, 256, 375.0, 375.0)]):
            focal = torch.tensor([fx, fy])
            princpt = torch.tensor([W / 2.0 + 3.0, H / 2.0 - 5.0])        # deliberately off-centre: ignored upstream
            A = torch.randn(3, 3, generator=g)
            Q, _ = torch.linalg.qr(A)
            t = torch.randn(3, generator=g)
            out['cam%d_in' % i] = np.array([H, W, fx, fy], dtype=np.float64)
            out['cam%d_R' % i] = Q.numpy()
            out['cam%d_t' % i] = t.numpy()
            out['cam%d_princpt' % i] = princpt.numpy()
            out['cam%d_fov' % i] = ref.get_fov(focal, princpt, (H, W)).numpy()
            out['cam%d_view' % i] = ref.get_view_matrix(Q, t).numpy()
            out['cam%d_proj' % i] = ref.get_proj_matrix(focal, princpt, (H, W), 0.01, 100, 1.0).numpy()
            cams.append(i)
        out['n_cams'] = np.array(len(cams))
        # SH: reference layout is sh[..., C, coeff]
        N = 64
        sh = torch.randn(N, 3, 16, generator=g)
        dirs = torch.randn(N, 3, generator=g)
        dirs = dirs / dirs.norm(dim=1, keepdim=True)
        out['sh_coeff'] = sh.numpy()
        out['sh_dirs'] = dirs.numpy()
        for deg in range(4):
            out['sh_eval_deg%d' % deg] = ref.eval_sh(deg, sh, dirs).numpy()
        rgb = torch.rand(N, 3, generator=g)
        out['rgb'] = rgb.numpy()
        out['rgb2sh'] = ref.RGB2SH(rgb).numpy()
        # covariance R S (R S)^T
        A = torch.randn(N, 3, 3, generator=g)
        Rm, _ = torch.linalg.qr(A)
        S = torch.rand(N, 3, generator=g) * 0.05 + 0.001
        out['cov_R'] = Rm.numpy()
        out['cov_S'] = S.numpy()
        out['cov'] = ref.get_covariance_matrix(S, Rm).numpy()
        np.savez_compressed(os.path.join(HERE, 'ref_transforms.npz'), **out)
    finally:
        torch.Tensor.cuda = orig_cuda


def oracle_fixture():
    from exavatar_release_amd import scenes
    from oracle import raster_oracle as ro
    torch.set_num_threads(1)
    H, W = 40, 56
    P = 300
    assets = scenes.dist_a_random(P, H, W, seed=5, focal=60.0, z_range=(1.5, 4.0))
    cam = scenes.neutral_camera(H, W, focal=60.0)
    g = torch.Generator().manual_seed(1)
    G = torch.randn(3, H, W, generator=g)
    Gd = torch.randn(1, H, W, generator=g)
    Ga = torch.randn(1, H, W, generator=g)
    bg = torch.rand(3, generator=g)
    a = {k: v.clone().requires_grad_(True) for k, v in assets.items()}
    ref = ro.render(a, (H, W), cam, bg, return_aux=True)
    loss = (ref['img'] * G).sum() + (ref['depthmap'] * Gd).sum() + (ref['mask'] * Ga).sum()
    loss.backward()
    out = {k: v.numpy() for k, v in assets.items()}
    out.update(H=np.array(H), W=np.array(W), focal=np.array(60.0), bg=bg.numpy(), G=G.numpy(), Gd=Gd.numpy(),
               Ga=Ga.numpy(), img=ref['img'].detach().numpy(), depth=ref['depthmap'].detach().numpy(),
               alpha=ref['mask'].detach().numpy(), radii=ref['radius'].numpy(),
               ambiguous=ro.ambiguous_pixel_mask(ref['aux'], H, W).numpy(),
               grad_mean_2d=ref['mean_2d'].grad.numpy())
    for k in assets:
        out['grad_' + k] = a[k].grad.numpy()
    np.savez_compressed(os.path.join(HERE, 'oracle_small.npz'), **out)


if __name__ == '__main__':
    if os.path.exists('/root/reference/avatar/common/utils/transforms.py'):
        reference_fixtures()
    else:
        print('reference not present: ref_transforms.npz left as committed')
    oracle_fixture()
    print('golden fixtures written to', HERE)
