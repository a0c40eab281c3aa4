"""CPU oracle for the differentiable 3D-Gaussian rasterizer behind ExAvatar's ``GaussianRenderer``.

THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only ``tests/``, ``__graft_entry__.smoke()`` and
the ``cpu_baseline`` leg of ``bench.py`` may import it.  The product path
(``exavatar_release_amd``) never falls back to it.

PARITY UNPINNED: the reference tree holds neither the rasterizer's source (it is the un-vendored,
un-pinned pip module ``diff_gaussian_rasterization_depth``, reference
``avatar/common/nets/module.py:11``, ``avatar/README.md:42``, ``environment.yml:272``) nor a single
test, golden image or tolerance for it (SURVEY.md sections 0.1, 0.2, 8c), and the module cannot be
imported or built in this environment.  This file therefore *restates the published algorithm*
of ``graphdeco-inria/diff-gaussian-rasterization`` (pre-antialiasing API, plus the depth / alpha
outputs of the ``-depth`` fork) as listed in SURVEY.md section 8(c), anchored on the reference's
own call site (``module.py:609-640``) and camera conventions (``transforms.py:38-70``).  It is
pinned only by the analytic known-answer tests and float64 finite-difference tests in ``tests/``.

Forward, per Gaussian i, all in ``dtype`` (float32 = the spec):
  1. p_view = [mu, 1] @ viewmatrix          cull if p_view.z <= 0.2
  2. p_hom  = [mu, 1] @ projmatrix ; p_ndc = p_hom.xyz / (p_hom.w + 1e-7)
  3. Sigma3 = R(q) diag(mod*s)^2 R(q)^T      q = (w, x, y, z), not re-normalised
  4. EWA: t = p_view with t.x/t.z, t.y/t.z clamped to +-1.3 tanfov; J; Sigma2 = J Rv Sigma3 Rv^T J^T
     Sigma2 += 0.3 I ; det == 0 -> culled ; conic = Sigma2^-1
  5. radius = ceil(3 sqrt(mid + sqrt(max(0.1, mid^2 - det))))
  6. pix = ((ndc + 1) * size - 1) / 2
  7. 16x16 tile rect, clamped; empty -> culled
  8. per tile ascending (depth fp32 bits, gaussian index)
Per pixel front to back (step 9): skip power > 0, alpha = min(0.99, o exp(power)), skip alpha < 1/255,
stop when T (1 - alpha) < 1e-4.  Outputs (step 10): color = C + T bg, depth = sum z alpha T,
alpha = 1 - T, radii.

Backward = torch.autograd of the forward with the rules of SURVEY.md 8(c): discrete decisions are
constants, ``min(0.99, .)`` is straight-through, a clamped ``t.x`` / ``t.y`` is a constant (upstream's ``x_grad_mul``),
the conic's backward carries upstream's ``1 / (det^2 + 1e-7)`` (``_ConicFromCov2D``),
``means2D`` is added to the NDC position so its gradient is d L / d pix * (W/2, H/2).
"""
from typing import NamedTuple, Optional

import torch

TILE = 16
NEAR_CULL = 0.2
FOV_CLAMP = 1.3
LOWPASS = 0.3
ALPHA_MAX = 0.99
ALPHA_MIN = 1.0 / 255.0
T_EPS = 1e-4

# SH constants: reference avatar/common/utils/transforms.py:82-110
C0 = 0.28209479177387814
C1 = 0.4886025119029199
C2 = [1.0925484305920792, -1.0925484305920792, 0.31539156525252005, -1.0925484305920792, 0.5462742152960396]
C3 = [-0.5900435899266435, 2.890611442640554, -0.4570457994644658, 0.3731763325901154,
      -0.4570457994644658, 1.445305721320277, -0.5900435899266435]


class OracleSettings(NamedTuple):
    """Same 12 fields, same order, as the third-party NamedTuple built at module.py:609-622."""
    image_height: int
    image_width: int
    tanfovx: float
    tanfovy: float
    bg: torch.Tensor
    scale_modifier: float
    viewmatrix: torch.Tensor
    projmatrix: torch.Tensor
    sh_degree: int
    campos: torch.Tensor
    prefiltered: bool
    debug: bool


def eval_sh_color(deg, shs, means3D, campos):
    """In-rasterizer SH colour: clamp_min(eval_sh(dir) + 0.5, 0).

    Follows reference transforms.py:112-167 (polynomials) and module.py:258-266 (direction =
    normalize(mean - cam_pos), +0.5, clamp at 0).  ``shs`` is [P, M, 3] (upstream layout).
    """
    d = means3D - campos[None, :]
    d = d / d.norm(dim=1, keepdim=True)
    x, y, z = d[:, 0:1], d[:, 1:2], d[:, 2:3]
    res = C0 * shs[:, 0]
    if deg > 0:
        res = res - C1 * y * shs[:, 1] + C1 * z * shs[:, 2] - C1 * x * shs[:, 3]
        if deg > 1:
            xx, yy, zz = x * x, y * y, z * z
            xy, yz, xz = x * y, y * z, x * z
            res = (res + C2[0] * xy * shs[:, 4] + C2[1] * yz * shs[:, 5]
                   + C2[2] * (2.0 * zz - xx - yy) * shs[:, 6]
                   + C2[3] * xz * shs[:, 7] + C2[4] * (xx - yy) * shs[:, 8])
            if deg > 2:
                res = (res + C3[0] * y * (3 * xx - yy) * shs[:, 9] + C3[1] * xy * z * shs[:, 10]
                       + C3[2] * y * (4 * zz - xx - yy) * shs[:, 11]
                       + C3[3] * z * (2 * zz - 3 * xx - 3 * yy) * shs[:, 12]
                       + C3[4] * x * (4 * zz - xx - yy) * shs[:, 13]
                       + C3[5] * z * (xx - yy) * shs[:, 14] + C3[6] * x * (xx - 3 * yy) * shs[:, 15])
    return torch.clamp_min(res + 0.5, 0.0)


def quat_to_rotmat(q):
    """R(q) for q = (w, x, y, z), no normalisation."""
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack((
        1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)), 1)
    return R.view(-1, 3, 3)


def cov3d_from_scale_rot(scales, rotations, mod):
    """Sigma3 = R S^2 R^T (cross-check: reference transforms.py:72-80 get_covariance_matrix)."""
    R = quat_to_rotmat(rotations)
    M = R * (mod * scales)[:, None, :]
    return M @ M.transpose(1, 2)


def cov3d_from_packed(c6):
    """Upper-triangular packing [xx, xy, xz, yy, yz, zz] -> 3x3 (upstream cov3D_precomp layout)."""
    xx, xy, xz, yy, yz, zz = [c6[:, i] for i in range(6)]
    return torch.stack((xx, xy, xz, xy, yy, yz, xz, yz, zz), 1).view(-1, 3, 3)


class _ConicFromCov2D(torch.autograd.Function):
    """conic = Sigma2^-1 = (c, -b, a) / det with upstream's backward.

    The forward is the exact inverse.  The backward follows upstream ``computeCov2DCUDA`` (backward): the
    exact partial derivatives, but with the common factor ``1 / det^2`` evaluated as
    ``1 / (det^2 + 1e-7)`` ("denom2inv") -- a guard upstream carries and that the replacement must
    reproduce (det >= 0.09 because of the +0.3 low-pass, so the deviation from the exact derivative is
    <= 1.2e-5 relative).  SURVEY.md 8(c); the HIP kernel uses the same form (preprocess_bwd.hip).
    """

    @staticmethod
    def forward(ctx, a, b, c):
        det = a * c - b * b
        det_safe = torch.where(det == 0, torch.ones_like(det), det)
        det_inv = 1.0 / det_safe
        ctx.save_for_backward(a, b, c, det)
        return torch.stack((c * det_inv, -b * det_inv, a * det_inv), 1)

    @staticmethod
    def backward(ctx, g):
        a, b, c, det = ctx.saved_tensors
        gA, gB, gC = g[:, 0], g[:, 1], g[:, 2]
        d2i = 1.0 / (det * det + 1e-7)
        da = d2i * (-c * c * gA + b * c * gB + (det - a * c) * gC)
        dc = d2i * ((det - a * c) * gA + a * b * gB - a * a * gC)
        db = d2i * (2.0 * b * c * gA - (det + 2.0 * b * b) * gB + 2.0 * a * b * gC)
        return da, db, dc


class _ZeroGradOfCulled(torch.autograd.Function):
    """Identity whose backward zeroes the rows of culled Gaussians (radius == 0).  Upstream's backward kernels return
    at once for them (``if (!(radii[idx] > 0)) return;``): their gradients are exactly zero.  Autograd would also give
    zero -- nothing downstream uses them -- except where the forward of such a Gaussian is not finite (a centre exactly
    on the camera plane: 1 / t.z = inf, and 0 * inf = NaN in the chain rule).  ``holder`` is filled with the visibility
    mask once ``preprocess`` has run; the backward pass reads it."""

    @staticmethod
    def forward(ctx, x, holder):
        ctx.holder = holder
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        vis = ctx.holder[0].view([-1] + [1] * (g.dim() - 1))
        return torch.where(vis, g, torch.zeros_like(g)), None


def preprocess(means3D, means2D, opacities, scales, rotations, cov3D_precomp, s: OracleSettings, dtype):
    """Steps 1-7. Returns a dict of per-Gaussian tensors (differentiable where it must be).

    Written as explicit scalar arithmetic in ONE fixed association order (every ``*`` and ``+`` below
    is a separately rounded IEEE operation, left to right as parenthesised; no fused multiply-add).
    This order is the specification: the HIP ``preprocess`` kernel is compiled with
    ``-ffp-contract=off`` and follows it term by term, so pixel centres, conics, radii and tile
    rects are bit-identical to this oracle in float32 and the per-tile lists agree exactly.
    """
    P = means3D.shape[0]
    H, W = int(s.image_height), int(s.image_width)
    f = lambda v: torch.tensor(float(v), dtype=torch.float32).to(dtype)   # host-side fp32 scalars
    v = [x.to(dtype) for x in s.viewmatrix.reshape(-1)]     # flat row-major of the [4,4] tensor
    p = [x.to(dtype) for x in s.projmatrix.reshape(-1)]
    mu = means3D.to(dtype)
    x, y, z = mu[:, 0], mu[:, 1], mu[:, 2]
    # 1. view-space position: [mu, 1] @ viewmatrix
    pvx = ((v[0] * x + v[4] * y) + v[8] * z) + v[12]
    pvy = ((v[1] * x + v[5] * y) + v[9] * z) + v[13]
    pvz = ((v[2] * x + v[6] * y) + v[10] * z) + v[14]
    # 2. clip space and perspective divide
    hx = ((p[0] * x + p[4] * y) + p[8] * z) + p[12]
    hy = ((p[1] * x + p[5] * y) + p[9] * z) + p[13]
    hw = ((p[3] * x + p[7] * y) + p[11] * z) + p[15]
    pw = 1.0 / (hw + 1e-7)
    ndcx = hx * pw
    ndcy = hy * pw
    if means2D is not None:
        ndcx = ndcx + means2D[:, 0].to(dtype)
        ndcy = ndcy + means2D[:, 1].to(dtype)
    # 3. 3D covariance (6 unique entries)
    if cov3D_precomp is not None:
        c6 = cov3D_precomp.to(dtype)
        S00, S01, S02, S11, S12, S22 = [c6[:, i] for i in range(6)]
    else:
        mod = f(s.scale_modifier)
        sc = scales.to(dtype)
        s0, s1, s2 = mod * sc[:, 0], mod * sc[:, 1], mod * sc[:, 2]
        q = rotations.to(dtype)
        qr, qx, qy, qz = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
        R00 = 1.0 - 2.0 * (qy * qy + qz * qz)
        R01 = 2.0 * (qx * qy - qr * qz)
        R02 = 2.0 * (qx * qz + qr * qy)
        R10 = 2.0 * (qx * qy + qr * qz)
        R11 = 1.0 - 2.0 * (qx * qx + qz * qz)
        R12 = 2.0 * (qy * qz - qr * qx)
        R20 = 2.0 * (qx * qz - qr * qy)
        R21 = 2.0 * (qy * qz + qr * qx)
        R22 = 1.0 - 2.0 * (qx * qx + qy * qy)
        M00, M01, M02 = R00 * s0, R01 * s1, R02 * s2
        M10, M11, M12 = R10 * s0, R11 * s1, R12 * s2
        M20, M21, M22 = R20 * s0, R21 * s1, R22 * s2
        S00 = (M00 * M00 + M01 * M01) + M02 * M02
        S01 = (M00 * M10 + M01 * M11) + M02 * M12
        S02 = (M00 * M20 + M01 * M21) + M02 * M22
        S11 = (M10 * M10 + M11 * M11) + M12 * M12
        S12 = (M10 * M20 + M11 * M21) + M12 * M22
        S22 = (M20 * M20 + M21 * M21) + M22 * M22
    # 4. EWA projection
    tanx, tany = f(s.tanfovx), f(s.tanfovy)
    focal_x = f(W) / (2.0 * tanx)
    focal_y = f(H) / (2.0 * tany)
    limx = 1.3 * tanx
    limy = 1.3 * tany
    tz = pvz
    txtz, tytz = pvx / tz, pvy / tz
    tx = torch.minimum(limx, torch.maximum(-limx, txtz)) * tz
    ty = torch.minimum(limy, torch.maximum(-limy, tytz)) * tz
    # upstream's backward (computeCov2DCUDA) differentiates J with respect to an INDEPENDENT (t.x, t.y, t.z) and
    # multiplies dL/dt.x (dL/dt.y) by 0 when the clamp is active ("x_grad_mul"): a clamped t.x is a constant, also
    # with respect to t.z.  Plain autograd of `clamp(t.x / t.z) * t.z` would keep d t.x / d t.z = +-1.3 tanfov.
    # (Unclamped: d/dt.x = 1 and d/dt.z = 0 either way.)
    tx = torch.where((txtz < -limx) | (txtz > limx), tx.detach(), tx)
    ty = torch.where((tytz < -limy) | (tytz > limy), ty.detach(), ty)
    J00 = focal_x / tz
    J02 = -(focal_x * tx) / (tz * tz)
    J11 = focal_y / tz
    J12 = -(focal_y * ty) / (tz * tz)
    # Rv[i][j] = viewmatrix[j][i]  (rotation part of world->camera)
    T00 = J00 * v[0] + J02 * v[2]
    T01 = J00 * v[4] + J02 * v[6]
    T02 = J00 * v[8] + J02 * v[10]
    T10 = J11 * v[1] + J12 * v[2]
    T11 = J11 * v[5] + J12 * v[6]
    T12 = J11 * v[9] + J12 * v[10]
    U00 = (T00 * S00 + T01 * S01) + T02 * S02
    U01 = (T00 * S01 + T01 * S11) + T02 * S12
    U02 = (T00 * S02 + T01 * S12) + T02 * S22
    U10 = (T10 * S00 + T11 * S01) + T12 * S02
    U11 = (T10 * S01 + T11 * S11) + T12 * S12
    U12 = (T10 * S02 + T11 * S12) + T12 * S22
    a = ((U00 * T00 + U01 * T01) + U02 * T02) + LOWPASS
    b = (U00 * T10 + U01 * T11) + U02 * T12
    c = ((U10 * T10 + U11 * T11) + U12 * T12) + LOWPASS
    det = a * c - b * b
    conic = _ConicFromCov2D.apply(a, b, c)      # exact forward, upstream's 1 / (det^2 + 1e-7) backward
    # 5. radius
    with torch.no_grad():
        mid = 0.5 * (a + c)
        lam = mid + torch.sqrt(torch.clamp_min(mid * mid - det, 0.1))
        r_f = 3.0 * torch.sqrt(lam)
        radius = torch.ceil(r_f).to(torch.int64)
    # 6. pixel centre
    px = ((ndcx + 1.0) * W - 1.0) * 0.5
    py = ((ndcy + 1.0) * H - 1.0) * 0.5
    # 7. tile rect
    gx = (W + TILE - 1) // TILE
    gy = (H + TILE - 1) // TILE
    with torch.no_grad():
        rf = radius.to(dtype)
        # C float -> int conversion truncates toward zero
        x0f, y0f = (px - rf) / TILE, (py - rf) / TILE
        x1f, y1f = ((px + rf) + (TILE - 1)) / TILE, ((py + rf) + (TILE - 1)) / TILE
        big = 1 << 20
        tr = lambda t_: torch.trunc(torch.nan_to_num(t_, nan=0.0).clamp(-big, big)).to(torch.int64)
        x0 = tr(x0f).clamp(0, gx)
        x1 = tr(x1f).clamp(0, gx)
        y0 = tr(y0f).clamp(0, gy)
        y1 = tr(y1f).clamp(0, gy)
        tiles = (x1 - x0) * (y1 - y0)
        vis = (tz > NEAR_CULL) & (det != 0) & (tiles > 0)
        radius = torch.where(vis, radius, torch.zeros_like(radius))
        tiles = torch.where(vis, tiles, torch.zeros_like(tiles))
        # decision margins (relative distance to the nearest discrete flip); only needed when two
        # implementations do NOT share this exact arithmetic (e.g. the float64 oracle vs float32)
        frac = r_f - torch.floor(r_f)
        m_rad = torch.minimum(frac, 1 - frac) / r_f.clamp_min(1.0)

        def edge_margin(t_):
            fr = t_ - torch.floor(t_)
            return torch.minimum(fr, 1 - fr) / (t_.abs() + 1.0)
        m_rect = torch.stack((edge_margin(x0f), edge_margin(x1f), edge_margin(y0f), edge_margin(y1f)), 1).amin(1)
        m_cull = (tz - NEAR_CULL).abs() / NEAR_CULL
        g_margin = torch.minimum(torch.minimum(m_rad, m_rect), m_cull)
    p_view = torch.stack((pvx, pvy, pvz), 1)
    return dict(p_view=p_view, depth=tz, px=px, py=py, conic=conic, radius=radius,
                rect=(x0, y0, x1, y1), tiles_touched=tiles, visible=vis, cov2=(a, b, c),
                gaussian_margin=g_margin, grid=(gx, gy))


def build_tile_lists(pre, dtype):
    """Step 8: instance list sorted by (tile, depth bits, index); returns (sorted idx, ranges[tiles,2])."""
    gx, gy = pre['grid']
    x0, y0, x1, y1 = pre['rect']
    vis = pre['visible']
    idx = torch.nonzero(vis, as_tuple=False).flatten()
    n_tiles = gx * gy
    if idx.numel() == 0:
        return torch.zeros(0, dtype=torch.int64), torch.zeros(n_tiles, 2, dtype=torch.int64)
    w = (x1 - x0)[idx]
    cnt = pre['tiles_touched'][idx]
    rep = torch.repeat_interleave(torch.arange(idx.numel()), cnt)
    start = torch.cumsum(cnt, 0) - cnt
    local = torch.arange(int(cnt.sum())) - start[rep]
    g = idx[rep]
    ty = y0[g] + local // w[rep]
    tx = x0[g] + local % w[rep]
    tile = ty * gx + tx
    d = pre['depth'].detach()[g]
    if dtype == torch.float32:
        dkey = d.contiguous().view(torch.int32).to(torch.int64)       # positive floats: bit order == value order
        key = tile * (1 << 32) + dkey
        order = torch.argsort(key, stable=True)
    else:
        # float64 oracle: order by value, ties by index (lexsort via two stable sorts)
        o1 = torch.argsort(d, stable=True)
        o2 = torch.argsort(tile[o1], stable=True)
        order = o1[o2]
    tile_sorted = tile[order]
    g_sorted = g[order]
    counts = torch.bincount(tile_sorted, minlength=n_tiles)
    ends = torch.cumsum(counts, 0)
    ranges = torch.stack((ends - counts, ends), 1)
    return g_sorted, ranges


def rasterize(means3D, means2D, opacities, shs=None, colors_precomp=None, scales=None, rotations=None,
              cov3D_precomp=None, settings: OracleSettings = None, dtype=torch.float32, return_aux=False,
              tile_subset=None):
    """Full forward. Returns (color[3,H,W], radii[P] int32, depth[1,H,W], alpha[1,H,W]) (+ aux dict).

    Argument names / order and output order follow the call at reference module.py:632-640.
    ``tile_subset`` (iterable of tile ids) restricts the per-tile loop to those tiles (all other
    pixels stay at the background); used only to time a bounded sample for bench.py's cpu_baseline.
    """
    s = settings
    if (shs is None) == (colors_precomp is None):
        raise Exception('Please provide excatly one of either SHs or precomputed colors!')
    if ((scales is None or rotations is None) and cov3D_precomp is None) or \
            ((scales is not None or rotations is not None) and cov3D_precomp is not None):
        raise Exception('Please provide exactly one of either scale/rotation pair or precomputed 3D covariance!')
    H, W = int(s.image_height), int(s.image_width)
    P = means3D.shape[0]
    bg = s.bg.to(dtype)
    vis_holder = []
    if torch.is_grad_enabled():             # culled Gaussians get exactly zero gradients, like upstream (see the class)
        wrap = lambda t_: _ZeroGradOfCulled.apply(t_, vis_holder) if (t_ is not None and t_.requires_grad) else t_   # noqa: E731
        means3D, means2D, opacities, scales = wrap(means3D), wrap(means2D), wrap(opacities), wrap(scales)
        rotations, cov3D_precomp, shs, colors_precomp = wrap(rotations), wrap(cov3D_precomp), wrap(shs), wrap(colors_precomp)
    pre = preprocess(means3D, means2D, opacities, scales, rotations, cov3D_precomp, s, dtype)
    vis_holder.append(pre['visible'])
    if shs is not None:
        colors = eval_sh_color(int(s.sh_degree), shs.to(dtype), means3D.to(dtype), s.campos.to(dtype))
    else:
        colors = colors_precomp.to(dtype)
    opac = opacities.to(dtype).view(-1)
    sorted_idx, ranges = build_tile_lists(pre, dtype)
    gx, gy = pre['grid']

    color = bg.view(3, 1, 1).expand(3, H, W).clone()
    depth = torch.zeros(H, W, dtype=dtype)
    alpha = torch.zeros(H, W, dtype=dtype)
    final_T = torch.ones(H, W, dtype=dtype)
    n_contrib = torch.zeros(H, W, dtype=torch.int32)
    pix_margin = torch.full((H, W), float('inf'), dtype=dtype)

    px, py, conic, z = pre['px'], pre['py'], pre['conic'], pre['depth']
    ranges_l = ranges.tolist()
    for t in (range(gx * gy) if tile_subset is None else tile_subset):
        s0, e0 = ranges_l[t]
        if e0 == s0:
            continue
        tx, ty = t % gx, t // gx
        xa, ya = tx * TILE, ty * TILE
        xb, yb = min(xa + TILE, W), min(ya + TILE, H)
        ids = sorted_idx[s0:e0]
        pix_x = torch.arange(xa, xb, dtype=dtype)
        pix_y = torch.arange(ya, yb, dtype=dtype)
        X = pix_x.repeat(yb - ya)                       # [npix]
        Y = pix_y.repeat_interleave(xb - xa)
        dx = px[ids][:, None] - X[None, :]
        dy = py[ids][:, None] - Y[None, :]
        cn = conic[ids]
        power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
        G = torch.exp(power)
        a_raw = opac[ids][:, None] * G
        a = a_raw + (a_raw.clamp(max=ALPHA_MAX) - a_raw).detach()     # straight-through min(0.99, .)
        with torch.no_grad():
            valid = (power <= 0) & (a >= ALPHA_MIN)
            a_v = torch.where(valid, a, torch.zeros_like(a))
            T_incl = torch.cumprod(1 - a_v, 0)
            keep = valid & (T_incl >= T_EPS)
            # upstream sets done=true at the first failing Gaussian; T_incl is monotone so nothing
            # after it passes either.
            if return_aux:
                idxs = torch.arange(1, ids.numel() + 1, dtype=torch.int32)[:, None].expand_as(keep)
                n_c = torch.where(keep, idxs, torch.zeros_like(idxs)).amax(0)
                T_excl = torch.cat((torch.ones(1, T_incl.shape[1], dtype=dtype), T_incl[:-1]), 0)
                alive = T_excl >= T_EPS            # evaluated before the pixel finished
                m_a = torch.where(alive & (power <= 0), (a - ALPHA_MIN).abs() * 255.0,
                                  torch.full_like(a, float('inf'))).amin(0)
                m_t = torch.where(alive & valid, (T_incl - T_EPS).abs() / T_EPS,
                                  torch.full_like(a, float('inf'))).amin(0)
                m_p = torch.where(alive, power.abs() + (power.abs() > 1e-6) * 1e9,
                                  torch.full_like(a, float('inf'))).amin(0)
                pm = torch.minimum(torch.minimum(m_a, m_t), m_p)
        a_k = torch.where(keep, a, torch.zeros_like(a))
        T_in = torch.cumprod(1 - a_k, 0)
        T_ex = torch.cat((torch.ones(1, T_in.shape[1], dtype=dtype), T_in[:-1]), 0)
        wgt = a_k * T_ex                                              # [n, npix]
        Tf = T_in[-1]
        c_t = wgt.t() @ colors[ids] + Tf[:, None] * bg[None, :]       # [npix, 3]
        d_t = wgt.t() @ z[ids]
        hh, ww = yb - ya, xb - xa
        color[:, ya:yb, xa:xb] = c_t.t().reshape(3, hh, ww)
        depth[ya:yb, xa:xb] = d_t.reshape(hh, ww)
        alpha[ya:yb, xa:xb] = (1 - Tf).reshape(hh, ww)
        if return_aux:
            final_T[ya:yb, xa:xb] = Tf.detach().reshape(hh, ww)
            n_contrib[ya:yb, xa:xb] = n_c.reshape(hh, ww)
            pix_margin[ya:yb, xa:xb] = pm.reshape(hh, ww)

    radii = pre['radius'].to(torch.int32)
    out = (color, radii, depth[None], alpha[None])
    if return_aux:
        aux = dict(pre=pre, sorted_idx=sorted_idx, ranges=ranges, final_T=final_T, n_contrib=n_contrib,
                   pixel_margin=pix_margin, colors=colors.detach())
        return out + (aux,)
    return out


def mark_visible(positions, settings: OracleSettings, dtype=torch.float32):
    """upstream ``markVisible``: p_view.z > 0.2 (unused by ExAvatar)."""
    P = positions.shape[0]
    mu_h = torch.cat((positions.to(dtype), torch.ones(P, 1, dtype=dtype)), 1)
    return (mu_h @ settings.viewmatrix.to(dtype)[:, :3])[:, 2] > NEAR_CULL


def ambiguous_pixel_mask(aux, H, W, rel=1e-4, include_gaussians=False):
    """Pixels whose value may legitimately differ between two fp32 implementations.

    A pixel is ambiguous when one of its discrete decisions (alpha < 1/255, T < 1e-4, power > 0) sits
    within ``rel`` of its threshold (exp() and fused multiply-adds differ by a few ulp between the CPU
    and the GPU; measured: the same float32 oracle on two different host CPUs already moves an alpha by
    1.4e-5 relative, hence the 1e-4 default).  With ``include_gaussians`` it is also ambiguous when it lies in a tile touched by
    a Gaussian whose radius / tile rect / near-cull decision is that close to flipping -- only
    needed against an implementation that does not share the oracle's preprocess arithmetic bit for
    bit (the float64 oracle); the HIP preprocess kernel does share it.
    """
    m = aux['pixel_margin'] < rel
    pre = aux['pre']
    amb_g = torch.nonzero(pre['gaussian_margin'] < rel).flatten() if include_gaussians else torch.zeros(0)
    if amb_g.numel():
        x0, y0, x1, y1 = pre['rect']
        gx, gy = pre['grid']
        for g in amb_g.tolist():
            xa = max(int(x0[g]) - 1, 0) * TILE
            xb = min(int(x1[g]) + 1, gx) * TILE
            ya = max(int(y0[g]) - 1, 0) * TILE
            yb = min(int(y1[g]) + 1, gy) * TILE
            m[ya:yb, xa:xb] = True
    return m


def settings_from_camera(cam_param, img_shape, bg, sh_degree=0):
    """Build settings exactly as ``GaussianRenderer.forward`` does (reference module.py:604-622)."""
    from exavatar_release_amd.camera import make_raster_matrices
    tanfovx, tanfovy, view, full_proj, cam_pos = make_raster_matrices(cam_param, img_shape)
    return OracleSettings(image_height=int(img_shape[0]), image_width=int(img_shape[1]), tanfovx=tanfovx,
                          tanfovy=tanfovy, bg=bg, scale_modifier=1.0, viewmatrix=view, projmatrix=full_proj,
                          sh_degree=sh_degree, campos=cam_pos, prefiltered=False, debug=False)


def render(gaussian_assets, img_shape, cam_param, bg=None, dtype=torch.float32, return_aux=False,
           tile_subset=None):
    """Oracle twin of ``GaussianRenderer.forward`` (reference module.py:592-647), CPU only."""
    if bg is None:
        bg = torch.ones(3)
    s = settings_from_camera(cam_param, img_shape, bg)
    P = gaussian_assets['mean_3d'].shape[0]
    mean_2d = torch.zeros(P, 3, dtype=dtype, requires_grad=True)
    res = rasterize(means3D=gaussian_assets['mean_3d'], means2D=mean_2d, shs=None,
                    colors_precomp=gaussian_assets['rgb'], opacities=gaussian_assets['opacity'],
                    scales=gaussian_assets['scale'], rotations=gaussian_assets['rotation'],
                    cov3D_precomp=None, settings=s, dtype=dtype, return_aux=return_aux, tile_subset=tile_subset)
    out = {'img': res[0], 'depthmap': res[2], 'mask': res[3], 'mean_2d': mean_2d,
           'is_vis': res[1] > 0, 'radius': res[1]}
    if return_aux:
        out['aux'] = res[4]
    return out


def densify_stats_reference(mean_2d_grad, radius, xyz_grad_accum, track_cnt, radius_max):
    """The reference's densification bookkeeping for ONE render, statement by statement
    (avatar/main/model.py:279-285 and SceneGaussian.track_stats, avatar/common/nets/module.py:155-157), on clones:
    ``is_vis = radius > 0`` (module.py:645), returns the three updated statistics."""
    is_vis = radius > 0
    xyz_grad_accum, track_cnt, radius_max = xyz_grad_accum.clone(), track_cnt.clone(), radius_max.clone()
    radius_max[is_vis] = torch.maximum(radius_max[is_vis], radius[is_vis].to(radius_max.dtype))
    xyz_grad_accum[is_vis, :] += torch.norm(mean_2d_grad[is_vis, :2], dim=1, keepdim=True)
    track_cnt[is_vis, :] += 1
    return xyz_grad_accum, track_cnt, radius_max
