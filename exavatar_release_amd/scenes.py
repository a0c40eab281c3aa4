"""Seeded synthetic Gaussian scenes and cameras for the BASELINE.json configurations.

No dataset or SMPL-X asset ships with the reference, so parity tests and
``bench.py`` use the synthetic distributions that SURVEY.md section 8(d) defines:

* Dist-A "random"       -- config C1 (10 k Gaussians, 256x256).
* Dist-B "avatar-like"  -- configs C2-C5: points on a union of capsules that approximates a 1.7 m
  body at z = 3 m, *isotropic* scale, *identity* rotation, *opacity = 1*, which mirrors what
  ``HumanGaussian.forward`` hands to the renderer (reference module.py:532,561-565).
* Dist-C "scene"        -- anisotropic background Gaussians (reference ``SceneGaussian``,
  module.py:253-272).

Everything is generated on the CPU from ``torch.Generator().manual_seed(seed)`` and returned as
contiguous float32 tensors (the layout ``GaussianRenderer.forward`` receives, module.py:594-598).
"""
import math

import torch


def neutral_camera(H, W, focal=None):
    """R = I, t = 0 looking down +z, focal = 1500 * (H / 1024) (reference get_neutral_pose.py:64-66)."""
    if focal is None:
        focal = 1500.0 * (H / 1024.0)
    return {
        'R': torch.eye(3, dtype=torch.float32),
        't': torch.zeros(3, dtype=torch.float32),
        'focal': torch.tensor([focal, focal], dtype=torch.float32),
        'princpt': torch.tensor([W / 2.0, H / 2.0], dtype=torch.float32),
    }


def ring_camera(H, W, k, n_views, radius=3.0, center=(0.0, 0.0, 3.0), focal=None):
    """k-th of ``n_views`` cameras on a horizontal ring looking at ``center`` (C4: 200 views).

    Same construction idea as the turntable of reference get_neutral_pose.py:76-82: rotate the
    world about the vertical axis through ``center`` and keep the camera ``radius`` away from it.
    """
    if focal is None:
        focal = 1500.0 * (H / 1024.0)
    ang = 2.0 * math.pi * k / n_views
    c, s = math.cos(ang), math.sin(ang)
    R = torch.tensor([[c, 0.0, s], [0.0, 1.0, 0.0], [-s, 0.0, c]], dtype=torch.float32)
    ctr = torch.tensor(center, dtype=torch.float32)
    # x_cam = R (x - ctr) + (0, 0, radius)
    t = -R @ ctr + torch.tensor([0.0, 0.0, radius], dtype=torch.float32)
    return {
        'R': R, 't': t,
        'focal': torch.tensor([focal, focal], dtype=torch.float32),
        'princpt': torch.tensor([W / 2.0, H / 2.0], dtype=torch.float32),
    }


def _unit_quat(n, g):
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=1, keepdim=True)


def dist_a_random(P, H, W, seed=0, z_range=(2.0, 6.0), focal=None):
    """Dist-A: uniform in the view frustum, anisotropic, random rotation / opacity / colour."""
    g = torch.Generator().manual_seed(seed)
    if focal is None:
        focal = 1500.0 * (H / 1024.0)
    z = z_range[0] + (z_range[1] - z_range[0]) * torch.rand(P, generator=g)
    # slightly over-fill the frustum so that tile-rect clipping at the image border is exercised
    u = (torch.rand(P, generator=g) * 1.1 - 0.05) * W
    v = (torch.rand(P, generator=g) * 1.1 - 0.05) * H
    x = (u - W / 2.0) / focal * z
    y = (v - H / 2.0) / focal * z
    mean = torch.stack((x, y, z), 1)
    log_s = math.log(0.005) + (math.log(0.05) - math.log(0.005)) * torch.rand(P, 3, generator=g)
    return {
        'mean_3d': mean.contiguous(),
        'scale': torch.exp(log_s).contiguous(),
        'rotation': _unit_quat(P, g).contiguous(),
        'opacity': torch.sigmoid(torch.randn(P, 1, generator=g)).contiguous(),
        'rgb': torch.rand(P, 3, generator=g).contiguous(),
    }


# (centre_a, centre_b, radius) capsules of a 1.7 m T-less standing body centred at (0, 0, 3);
# image y points down, so the head is at negative y.
_BODY_CAPSULES = [
    ((0.00, -0.78, 0.0), (0.00, -0.62, 0.0), 0.10),   # head
    ((0.00, -0.55, 0.0), (0.00, 0.05, 0.0), 0.16),    # torso
    ((-0.22, -0.45, 0.0), (-0.32, 0.10, 0.0), 0.05),  # left arm
    ((0.22, -0.45, 0.0), (0.32, 0.10, 0.0), 0.05),    # right arm
    ((-0.09, 0.10, 0.0), (-0.11, 0.82, 0.0), 0.075),  # left leg
    ((0.09, 0.10, 0.0), (0.11, 0.82, 0.0), 0.075),    # right leg
]


def _capsule_surface(n, a, b, r, g):
    a = torch.tensor(a)
    b = torch.tensor(b)
    axis = b - a
    L = float(axis.norm())
    axis = axis / L
    # area split between the cylinder wall and the two hemispherical caps
    area_cyl = 2 * math.pi * r * L
    area_cap = 4 * math.pi * r * r
    on_cyl = torch.rand(n, generator=g) < area_cyl / (area_cyl + area_cap)
    # orthonormal frame
    ref = torch.tensor([0.0, 0.0, 1.0]) if abs(float(axis[2])) < 0.9 else torch.tensor([1.0, 0.0, 0.0])
    e1 = torch.linalg.cross(axis, ref)
    e1 = e1 / e1.norm()
    e2 = torch.linalg.cross(axis, e1)
    phi = 2 * math.pi * torch.rand(n, generator=g)
    radial = torch.cos(phi)[:, None] * e1 + torch.sin(phi)[:, None] * e2
    h = torch.rand(n, generator=g) * L
    p_cyl = a + h[:, None] * axis + r * radial
    nrm_cyl = radial
    d = torch.randn(n, 3, generator=g)
    d = d / d.norm(dim=1, keepdim=True)
    top = (d @ axis) > 0
    p_cap = torch.where(top[:, None], b + r * d, a + r * d)
    pts = torch.where(on_cyl[:, None], p_cyl, p_cap)
    nrm = torch.where(on_cyl[:, None], nrm_cyl, d)
    return pts, nrm


def dist_b_avatar(P, seed=0, center=(0.0, 0.0, 3.0), scale_mu=0.003, scale_sigma=0.3):
    """Dist-B: avatar-like. Isotropic scale, identity rotation, opacity 1 (module.py:532,564-565)."""
    g = torch.Generator().manual_seed(seed)
    areas = []
    for a, b, r in _BODY_CAPSULES:
        L = math.dist(a, b)
        areas.append(2 * math.pi * r * L + 4 * math.pi * r * r)
    tot = sum(areas)
    counts = [int(P * ar / tot) for ar in areas]
    counts[1] += P - sum(counts)
    pts, nrms = [], []
    for (a, b, r), n in zip(_BODY_CAPSULES, counts):
        p, nr = _capsule_surface(n, a, b, r, g)
        pts.append(p)
        nrms.append(nr)
    pts = torch.cat(pts)
    nrms = torch.cat(nrms)
    pts = pts + nrms * (0.01 * torch.randn(P, 1, generator=g))
    perm = torch.randperm(P, generator=g)
    pts = pts[perm] + torch.tensor(center)
    s = torch.exp(math.log(scale_mu) + scale_sigma * torch.randn(P, 1, generator=g))
    rot = torch.zeros(P, 4)
    rot[:, 0] = 1.0
    return {
        'mean_3d': pts.float().contiguous(),
        'scale': s.repeat(1, 3).float().contiguous(),
        'rotation': rot.contiguous(),
        'opacity': torch.ones(P, 1),
        'rgb': torch.rand(P, 3, generator=g).contiguous(),
    }


def dist_c_scene(P, H, W, seed=0, focal=None):
    """Dist-C: background scene, Dist-A with z in [3, 15] m and opacity sigmoid(N(0, 1.5))."""
    g = torch.Generator().manual_seed(seed + 7919)
    out = dist_a_random(P, H, W, seed=seed + 104729, z_range=(3.0, 15.0), focal=focal)
    out['opacity'] = torch.sigmoid(1.5 * torch.randn(P, 1, generator=g)).contiguous()
    return out


def cat_assets(a, b):
    """Concatenate two asset dicts (reference model.py:119-127 does this for scene + human)."""
    return {k: torch.cat((a[k], b[k]), 0).contiguous() for k in a}


def sh_from_rgb(rgb, degree, seed=0, rest_sigma=0.1):
    """SH coefficients [P, (degree+1)^2, 3]: DC = RGB2SH(rgb) (transforms.py:169-170), rest ~ N(0, .1)."""
    C0 = 0.28209479177387814
    g = torch.Generator().manual_seed(seed + 31337)
    P = rgb.shape[0]
    M = (degree + 1) ** 2
    sh = rest_sigma * torch.randn(P, M, 3, generator=g)
    sh[:, 0, :] = (rgb - 0.5) / C0
    return sh.contiguous()


# name -> (P, H, W, builder) for the BASELINE.json configs (SURVEY.md section 8a sizes)
def make_config(name, seed=0):
    """Returns (assets, img_shape(H, W), cam_param) for a BASELINE config name.

    'c1' 10 k Dist-A 256x256 | 'c2' 120 k Dist-B 960x540 (H x W) | 'c2l' 540x960 |
    'c3' 150 k Dist-B 1024x1024 | 'c3s' 150 k avatar + 50 k scene 1024x1024 | 'c5' 300 k 2048x2048.
    """
    if name == 'c1':
        H, W = 256, 256
        return dist_a_random(10_000, H, W, seed), (H, W), neutral_camera(H, W)
    if name == 'c2':
        H, W = 960, 540
        return dist_b_avatar(120_000, seed), (H, W), neutral_camera(H, W, focal=1500.0 * 960 / 1024)
    if name == 'c2l':
        H, W = 540, 960
        return dist_b_avatar(120_000, seed), (H, W), neutral_camera(H, W, focal=1500.0 * 540 / 1024)
    if name == 'c3':
        H, W = 1024, 1024
        return dist_b_avatar(150_000, seed), (H, W), neutral_camera(H, W)
    if name == 'c3s':
        H, W = 1024, 1024
        a = cat_assets(dist_c_scene(50_000, H, W, seed), dist_b_avatar(150_000, seed))
        return a, (H, W), neutral_camera(H, W)
    if name == 'c5':
        H, W = 2048, 2048
        a = cat_assets(dist_c_scene(100_000, H, W, seed), dist_b_avatar(200_000, seed))
        return a, (H, W), neutral_camera(H, W)
    raise KeyError(name)
