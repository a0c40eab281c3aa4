"""Generates tests/golden/ref_ssim.npz by EXECUTING the reference's own loss classes.

``/root/reference/avatar/common/nets/loss.py`` cannot be imported as a module here (it imports lpips, pytorch3d and
the training config at the top), so the source text of ``class RGBLoss`` and ``class SSIM`` is cut out of the file
with ``ast`` and exec'd unchanged in a namespace holding only torch / nn / F / math; ``Tensor.cuda()`` is a no-op for
the duration (this container has no GPU).  Run from the repo root:  python tests/golden/make_golden_ssim.py
Nothing here is read at test time on the GPU box -- only the .npz travels.
"""
import ast
import math
import os

import numpy as np
import torch
import torch.nn as nn
from torch.nn import functional as F

REF = '/root/reference/avatar/common/nets/loss.py'
src = open(REF).read()
tree = ast.parse(src)
ns = {'torch': torch, 'nn': nn, 'F': F, 'math': math, 'np': np}
lines = src.splitlines()
for node in tree.body:
    if isinstance(node, ast.ClassDef) and node.name in ('RGBLoss', 'SSIM'):
        exec('\n'.join(lines[node.lineno - 1: node.end_lineno]), ns)
_cuda = torch.Tensor.cuda
torch.Tensor.cuda = lambda self, *a, **k: self
try:
    g = torch.Generator().manual_seed(2024)
    B, C, H, W = 2, 3, 37, 53
    out = {}
    x = torch.rand(B, C, H, W, generator=g)
    y = (x + 0.2 * torch.randn(B, C, H, W, generator=g)).clamp(0, 1)
    mask = (torch.rand(B, 1, H, W, generator=g) > 0.4).float()
    bbox = torch.tensor([[-3.0, 5.0, 40.0, 60.0]])            # clipped on two sides, as the reference clamps it
    bg = torch.rand(B, 3, generator=g)
    Gm = torch.randn(B, C, H, W, generator=g)
    ssim, rgb = ns['SSIM'](), ns['RGBLoss']()
    for name, kw in (('plain', {}), ('mask', {'mask': mask}), ('bbox', {'bbox': bbox})):
        xi = x.clone().requires_grad_(True)
        m = ssim(xi, y, **kw)
        w_ = Gm[:, :, :m.shape[2], :m.shape[3]]
        (m * w_).sum().backward()
        out['ssim_' + name] = m.detach().numpy()
        out['ssim_' + name + '_grad'] = xi.grad.numpy()
    for name, kw in (('plain', {}), ('bbox', {'bbox': bbox}), ('maskbg', {'mask': mask, 'bg': bg})):
        xi = x.clone().requires_grad_(True)
        m = rgb(xi, y, **kw)
        (m * Gm[:, :, :m.shape[2], :m.shape[3]]).sum().backward()
        out['rgb_' + name] = m.detach().numpy()
        out['rgb_' + name + '_grad'] = xi.grad.numpy()
    # the per-render objectives of avatar/main/model.py:197-198 (human: bbox) and :214-215 (scene: 1 - mask), assembled
    # from the reference's own classes with the weights of avatar/main/config.py:35-36 and train.py:43's .mean()
    W_RGB, W_SSIM = 0.8, 0.2
    xi = x.clone().requires_grad_(True)
    L = (rgb(xi, y, bbox=bbox) * W_RGB).mean() + ((1 - ssim(xi, y, bbox=bbox)) * W_SSIM).mean()
    L.backward()
    out['photo_human'] = L.detach().numpy()
    out['photo_human_grad'] = xi.grad.numpy()
    xi = x.clone().requires_grad_(True)
    L = (rgb(xi, y) * (1 - mask) * W_RGB).mean() + ((1 - ssim(xi, y, mask=1 - mask)) * W_SSIM).mean()
    L.backward()
    out['photo_scene'] = L.detach().numpy()
    out['photo_scene_grad'] = xi.grad.numpy()
    out.update(x=x.numpy(), y=y.numpy(), mask=mask.numpy(), bbox=bbox.numpy(), bg=bg.numpy(), G=Gm.numpy())
finally:
    torch.Tensor.cuda = _cuda
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ref_ssim.npz')
np.savez_compressed(path, **out)
print('wrote', path, {k: v.shape for k, v in out.items()})
