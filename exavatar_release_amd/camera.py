"""Pinhole camera -> 3DGS view / projection matrices.

Host-side mirror of the reference's camera helpers that `GaussianRenderer.forward`
calls before it builds `GaussianRasterizationSettings`:

* ``get_view_matrix``  -- reference ``avatar/common/utils/transforms.py:38-41``
* ``get_proj_matrix``  -- reference ``avatar/common/utils/transforms.py:43-64``
* ``get_fov``          -- reference ``avatar/common/utils/transforms.py:66-70``

Same names, argument meaning and results.  The only difference is that tensors are created on
the device of the inputs instead of a hard-coded ``.cuda()`` so the helpers also serve the CPU
oracle and the CPU tests.  As in the reference ``princpt`` is accepted and ignored: the effective
principal point is the image centre.
"""
import math

import torch


def get_fov(focal, princpt, img_shape):
    """fov = 2*atan(size / (2*focal)); returns tensor [fov_x, fov_y] (transforms.py:66-70)."""
    focal = torch.as_tensor(focal, dtype=torch.float32)
    fov_x = 2 * torch.atan(img_shape[1] / (2 * focal[0]))
    fov_y = 2 * torch.atan(img_shape[0] / (2 * focal[1]))
    return torch.stack((fov_x, fov_y)).float().to(focal.device)


def get_view_matrix(R, t):
    """4x4 world->camera [[R, t], [0, 0, 0, 1]] (transforms.py:38-41); the caller transposes it."""
    Rt = torch.cat((R, t.view(3, 1)), 1)
    last = torch.tensor([0, 0, 0, 1], dtype=Rt.dtype, device=Rt.device).view(1, 4)
    return torch.cat((Rt, last))


def get_proj_matrix(focal, princpt, img_shape, z_near, z_far, z_sign):
    """Symmetric-frustum perspective matrix (transforms.py:43-64). ``z_sign`` is forced to 1."""
    fov = get_fov(focal, princpt, img_shape)
    tan_half_y = math.tan(float(fov[1]) / 2)
    tan_half_x = math.tan(float(fov[0]) / 2)
    top = tan_half_y * z_near
    bottom = -top
    right = tan_half_x * z_near
    left = -right
    z_sign = 1.0
    P = torch.zeros(4, 4, dtype=torch.float32, device=fov.device)
    P[0, 0] = 2.0 * z_near / (right - left)
    P[1, 1] = 2.0 * z_near / (top - bottom)
    P[0, 2] = (right + left) / (right - left)
    P[1, 2] = (top + bottom) / (top - bottom)
    P[3, 2] = z_sign
    P[2, 2] = z_sign * z_far / (z_far - z_near)
    P[2, 3] = -(z_far * z_near) / (z_far - z_near)
    return P


def make_raster_matrices(cam_param, img_shape, z_near=0.01, z_far=100.0):
    """The matrix block of ``GaussianRenderer.forward`` (reference module.py:604-608).

    Returns (tanfovx, tanfovy, view_matrix^T, full_proj = view^T @ proj^T, cam_pos).
    """
    fov = get_fov(cam_param['focal'], cam_param['princpt'], img_shape)
    view = get_view_matrix(cam_param['R'], cam_param['t']).permute(1, 0)
    proj = get_proj_matrix(cam_param['focal'], cam_param['princpt'], img_shape, z_near, z_far, 1.0).permute(1, 0)
    proj = proj.to(view.device)
    full_proj = torch.mm(view, proj)
    cam_pos = view.inverse()[3, :3]
    tanfovx = float(torch.tan(fov[0] / 2))
    tanfovy = float(torch.tan(fov[1] / 2))
    return tanfovx, tanfovy, view.contiguous(), full_proj.contiguous(), cam_pos.contiguous()
