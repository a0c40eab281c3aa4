"""Synthetic stand-in for the part of ``HumanGaussian.forward`` that sits IN FRONT of the rasterizer
(reference ``avatar/common/nets/module.py:516-586``): per-vertex offsets + SMPL-X linear blend skinning produce
``mean_3d`` / ``scale`` / ``rgb`` as NON-LEAF tensors, so that the rasterizer's gradients flow on into pose, translation
and offset parameters -- BASELINE.json ``configs[2]`` ("~150k Gaussians + SMPL-X LBS").

This stays PyTorch-ROCm, as north_star says ("The SMPL-X LBS/pose-deformation and per-Gaussian offset MLP stay in
PyTorch-ROCm; only the rasterize forward/backward moves"): plain tensor ops, no custom kernels.  The SMPL-X assets are
licence-gated and absent, so the skeleton, the rest pose and the skinning weights are synthetic with the
real model's STRUCTURE: 55 joints in a kinematic tree (``smplx`` body 22 + jaw + 2 eyes + 2 x 15 fingers), every vertex
skinned to at most four joints, one rigid transform per joint composed along the tree
(``smplx/lbs.py:361-417 batch_rigid_transform``), vertex transform = skinning-weight matrix x joint transforms
(``module.py:413-416``), ``xyz' = T_v [xyz + offset, 1] + trans`` (``module.py:418-422``).

What the reference also does here and this module does not: triplane feature lookup + the offset / scale / colour MLPs
(their outputs are plain leaf parameters here), the nearest-vertex search (``knn_points``; stand-in in
``p3d_standins.py``), expression blend shapes.
"""
import math

import torch
import torch.nn as nn

JOINT_NUM = 55

# parents of the 55 SMPL-X joints (pelvis root; 21 body joints, jaw, eyes, 15 + 15 finger joints): the published kinematic
# tree of the model (smplx `parents`), written out because the asset that carries it is not in the tree
SMPLX_PARENTS = (-1, 0, 0, 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 9, 9, 12, 13, 14, 16, 17, 18, 19, 15, 15, 15,
                 20, 25, 26, 20, 28, 29, 20, 31, 32, 20, 34, 35, 20, 37, 38,
                 21, 40, 41, 21, 43, 44, 21, 46, 47, 21, 49, 50, 21, 52, 53)


def axis_angle_to_matrix(aa):
    """Rodrigues' formula for [N, 3] axis-angle vectors -> [N, 3, 3] (what ``module.py:405`` gets from pytorch3d);
    written through sin(x)/x and (1 - cos x)/x^2 so that the zero rotation has a finite gradient."""
    theta2 = (aa * aa).sum(-1, keepdim=True)
    theta = torch.sqrt(theta2 + 1e-12)
    a = (torch.sin(theta) / theta)[..., None]                  # sin(t) / t
    b = ((1.0 - torch.cos(theta)) / (theta2 + 1e-12))[..., None]  # (1 - cos t) / t^2
    x, y, z = aa[:, 0], aa[:, 1], aa[:, 2]
    zero = torch.zeros_like(x)
    K = torch.stack((zero, -z, y, z, zero, -x, -y, x, zero), -1).view(-1, 3, 3)
    eye = torch.eye(3, dtype=aa.dtype, device=aa.device)
    return eye + a * K + b * (K @ K)


def _level_plan(parents):
    """Joints grouped by depth in the tree: ``[(joint ids of the level, position of each one's parent inside the PREVIOUS
    level)]`` + the permutation that puts the concatenated levels back into joint order.  All joints of a level compose with
    their parents in ONE batched matmul (eleven for SMPL-X instead of the reference's 54 sequential ones)."""
    depth = [0] * len(parents)
    for i, p in enumerate(parents):
        depth[i] = 0 if p < 0 else depth[p] + 1
    levels = [[i for i, d in enumerate(depth) if d == lv] for lv in range(max(depth) + 1)]
    plan = [(levels[lv], [levels[lv - 1].index(parents[i]) for i in levels[lv]]) for lv in range(1, len(levels))]
    order = [i for lv in levels for i in lv]
    inverse = [order.index(i) for i in range(len(parents))]
    return plan, inverse


_plans = {}


def joint_transforms(rot, joints, parents=SMPLX_PARENTS):
    """Relative rigid transforms of the joints, rest pose -> posed ([J, 4, 4]): the composition of per-joint transforms
    along the kinematic tree with the rest joint location removed (``smplx/lbs.py:361-417``, second return value)."""
    J, dev = rot.shape[0], rot.device
    key = (tuple(parents), dev)
    if key not in _plans:
        plan, inverse = _level_plan(parents)
        _plans[key] = ([(torch.tensor(ids, device=dev), torch.tensor(ppos, device=dev)) for ids, ppos in plan],
                       torch.tensor(inverse, device=dev), torch.tensor([max(p, 0) for p in parents], device=dev))
    plan, inverse, par = _plans[key]
    rel = joints - joints[par]
    rel = torch.cat((joints[:1], rel[1:]))
    bottom = torch.zeros(J, 1, 4, dtype=rot.dtype, device=dev)
    bottom[:, 0, 3] = 1.0
    local = torch.cat((torch.cat((rot, rel[:, :, None]), 2), bottom), 1)          # [J, 4, 4]
    levels = [local[:1]]
    for ids, ppos in plan:
        levels.append(levels[-1][ppos] @ local[ids])
    world = torch.cat(levels)[inverse]
    # remove the rest-pose joint location: T' = T - [0 | T [j, 0]]  (column 3)
    shift = (world[:, :, :3] @ joints[:, :, None])[:, :, 0]
    return torch.cat((world[:, :, :3], (world[:, :, 3] - shift)[:, :, None]), 2)


class SyntheticAvatar(nn.Module):
    """P avatar-like Gaussians driven by a 55-joint skeleton (module docstring).  Parameters: ``pose`` [55, 3] axis-angle,
    ``trans`` [3], ``mean_offset`` [P, 3], ``scale_log`` [P, 1] (isotropic, ``module.py:532``), ``rgb_logit`` [P, 3]
    (``rgb = (tanh + 1) / 2``, ``module.py:561``).  Buffers: the rest-pose surface points, the rest joints, the dense
    [P, 55] skinning-weight matrix (<= 4 non-zeros per row).  ``forward()`` returns the asset dict the renderer takes."""

    def __init__(self, base_assets, seed=0, pose_sigma=0.15):
        super().__init__()
        g = torch.Generator().manual_seed(seed)
        xyz = base_assets['mean_3d'].clone()
        P = xyz.shape[0]
        lo, hi = xyz.min(0).values, xyz.max(0).values
        # rest joints: spread through the body's bounding box, children near their parents
        joints = torch.zeros(JOINT_NUM, 3)
        joints[0] = (lo + hi) / 2
        for i in range(1, JOINT_NUM):
            step = (hi - lo) * 0.18 * (torch.rand(3, generator=g) - 0.5)
            joints[i] = torch.minimum(torch.maximum(joints[SMPLX_PARENTS[i]] + step, lo), hi)
        # skinning weights: the four nearest joints of every point, weights ~ exp(-distance / 5 cm)
        d = torch.cdist(xyz, joints)
        dist, idx = d.topk(4, dim=1, largest=False)
        w = torch.softmax(-dist / 0.05, dim=1)
        W = torch.zeros(P, JOINT_NUM)
        W.scatter_(1, idx, w)
        self.register_buffer('xyz', xyz)
        self.register_buffer('joints', joints)
        self.register_buffer('skinning_weight', W)
        self.register_buffer('rotation', base_assets['rotation'].clone())     # identity quaternions (module.py:564)
        self.register_buffer('opacity', base_assets['opacity'].clone())       # ones (module.py:565)
        self.pose = nn.Parameter(pose_sigma * torch.randn(JOINT_NUM, 3, generator=g))
        self.trans = nn.Parameter(torch.zeros(3))
        self.mean_offset = nn.Parameter(0.002 * torch.randn(P, 3, generator=g))
        self.scale_log = nn.Parameter(torch.log(base_assets['scale'][:, :1].clone()))
        self.rgb_logit = nn.Parameter(torch.atanh((2 * base_assets['rgb'].clamp(0.02, 0.98) - 1)))
        # skinning is relative to the pose the surface points were generated in: every joint transform is composed with the
        # inverse of the INITIAL pose's (the reference composes "big pose -> zero pose -> image pose", module.py:408-410)
        with torch.no_grad():
            self.register_buffer('rest_inverse', torch.linalg.inv(joint_transforms(axis_angle_to_matrix(self.pose), joints)))

    def forward(self):
        rot = axis_angle_to_matrix(self.pose)
        T_joint = joint_transforms(rot, self.joints)                                       # [55, 4, 4]
        T_joint = T_joint @ self.rest_inverse
        # vertex transforms = skinning weights x joint transforms (module.py:413-416), applied to [xyz + offset, 1]
        # (module.py:418-422).  Only the three affine rows are formed, and the 150 k tiny 4x4 products are written as one
        # broadcast multiply-add: torch.bmm over a batch of P 4x4 matrices takes rocBLAS 3 ms per direction here
        T_vertex = (self.skinning_weight @ T_joint[:, :3, :].reshape(JOINT_NUM, 12)).view(-1, 3, 4)
        xyz = self.xyz + self.mean_offset
        mean_3d = (T_vertex[:, :, :3] * xyz[:, None, :]).sum(-1) + T_vertex[:, :, 3] + self.trans
        return {'mean_3d': mean_3d, 'opacity': self.opacity, 'scale': torch.exp(self.scale_log).repeat(1, 3),
                'rotation': self.rotation, 'rgb': (torch.tanh(self.rgb_logit) + 1) / 2}


def _self_check():
    """(python -m exavatar_release_amd.lbs) at the initial pose the skinned points are the rest points."""
    from . import scenes
    a = scenes.dist_b_avatar(2000, seed=0)
    m = SyntheticAvatar(a)
    out = m()
    err = float((out['mean_3d'] - (m.xyz + m.mean_offset)).abs().max())
    print('max |skinned - rest| at the initial pose: %.2e' % err)
    assert err < 1e-5 and math.isfinite(err)


if __name__ == '__main__':
    _self_check()
