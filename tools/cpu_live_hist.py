"""Developer tool (CPU only, uses the oracle): how many of a sub-tile's 64 pixels are still LIVE while the forward blend
walks its list -- histogram over the walked groups of four entries of C3 (ring view argv[1], default 0), and the share of
(pixel, entry) evaluations spent on dead pixels.  Prices pixel-compaction schemes for render_fwd.  ~20 s."""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exavatar_release_amd import scenes
from oracle import raster_oracle as ro
torch.set_num_threads(8)
H = W = 1024
view = int(sys.argv[1]) if len(sys.argv) > 1 else 0
assets = scenes.dist_b_avatar(150000, seed=0)
cam = scenes.ring_camera(H, W, view, 200)
s = ro.settings_from_camera(cam, (H, W), torch.ones(3))
dtype = torch.float32
with torch.no_grad():
    pre = ro.preprocess(assets['mean_3d'], None, assets['opacity'], assets['scale'], assets['rotation'], None, s, dtype)
    sorted_idx, ranges = ro.build_tile_lists(pre, dtype)
    gx, gy = pre['grid']
    px, py, conic = pre['px'], pre['py'], pre['conic']
    a2, b2, c2 = pre['cov2']
    opac = assets['opacity'].view(-1)
    tau2 = 2 * torch.log(255 * opac) + 1e-3
    ex = torch.sqrt(tau2 * a2) * 1.001 + 0.01
    ey = torch.sqrt(tau2 * c2) * 1.001 + 0.01
    bx0 = torch.floor(px - ex); bx1 = torch.ceil(px + ex); by0 = torch.floor(py - ey); by1 = torch.ceil(py + ey)
    hist = torch.zeros(65, dtype=torch.long)        # groups of four walked with k live pixels at their start
    lists = []
    for t in range(gx * gy):
        s0, e0 = ranges[t].tolist()
        if e0 == s0:
            continue
        tx, ty = t % gx, t // gx
        ids = sorted_idx[s0:e0]
        for sy in range(2):
            for sx in range(2):
                ox = tx * 16 + sx * 8; oy = ty * 16 + sy * 8
                sel = (bx0[ids] <= ox + 7) & (bx1[ids] >= ox) & (by0[ids] <= oy + 7) & (by1[ids] >= oy)
                l = ids[sel]
                n = l.numel()
                if n == 0:
                    continue
                X = torch.arange(ox, ox + 8, dtype=dtype).repeat(8); Y = torch.arange(oy, oy + 8, dtype=dtype).repeat_interleave(8)
                dx = px[l][:, None] - X[None, :]; dy = py[l][:, None] - Y[None, :]
                cn = conic[l]
                power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
                a = (opac[l][:, None] * torch.exp(power)).clamp(max=0.99)
                av = torch.where((power <= 0) & (a >= 1 / 255.), a, torch.zeros_like(a))
                Tin = torch.cumprod(1 - av, 0)
                Tex = torch.cat((torch.ones(1, 64), Tin[:-1]), 0)
                live = (Tex >= 1e-4).sum(1)                       # live pixels in front of every entry
                g4 = live[::4]
                walked = int((g4 > 0).sum())
                hist += torch.bincount(g4[:walked], minlength=65)
                lists.append((n, walked * 4))
    tot = int(hist.sum())
    cum = torch.cumsum(hist, 0)
    print('view', view, 'groups of four walked:', tot, ' lists:', len(lists), ' entries listed / walked:',
          sum(a for a, b in lists), sum(b for a, b in lists))
    for k in (1, 2, 4, 8, 16, 32, 48, 63, 64):
        print('  groups with <= %2d live pixels: %5.1f %%' % (k, 100.0 * int(cum[k]) / tot))
    w = torch.arange(65, dtype=torch.float64)
    print('  mean live pixels per walked group: %.1f of 64 -> %.1f %% of the (pixel, entry) evaluations are on dead pixels'
          % (float((hist * w).sum()) / tot, 100 - 100 * float((hist * w).sum()) / tot / 64))
    long_ = sorted(lists, reverse=True)[:10]
    print('  ten longest lists (listed, walked):', long_)

# ---- the same question for the BACKWARD: per entered batch its blended entries in chunks of eight (render_bwd.hip), live
# pixels at the first entry of every chunk
with torch.no_grad():
    hist_b = torch.zeros(65, dtype=torch.long)
    for t in range(gx * gy):
        s0, e0 = ranges[t].tolist()
        if e0 == s0:
            continue
        tx, ty = t % gx, t // gx
        ids = sorted_idx[s0:e0]
        for sy in range(2):
            for sx in range(2):
                ox = tx * 16 + sx * 8; oy = ty * 16 + sy * 8
                sel = (bx0[ids] <= ox + 7) & (bx1[ids] >= ox) & (by0[ids] <= oy + 7) & (by1[ids] >= oy)
                l = ids[sel]
                n = l.numel()
                if n == 0:
                    continue
                X = torch.arange(ox, ox + 8, dtype=dtype).repeat(8); Y = torch.arange(oy, oy + 8, dtype=dtype).repeat_interleave(8)
                dx = px[l][:, None] - X[None, :]; dy = py[l][:, None] - Y[None, :]
                cn = conic[l]
                power = -0.5 * (cn[:, 0:1] * dx * dx + cn[:, 2:3] * dy * dy) - cn[:, 1:2] * dx * dy
                a = (opac[l][:, None] * torch.exp(power)).clamp(max=0.99)
                valid = (power <= 0) & (a >= 1 / 255.)
                av = torch.where(valid, a, torch.zeros_like(a))
                Tin = torch.cumprod(1 - av, 0)
                Tex = torch.cat((torch.ones(1, 64), Tin[:-1]), 0)
                alive = Tex >= 1e-4
                blended = (valid & alive).any(1)
                live = alive.sum(1)
                for b0 in range(0, n, 64):
                    if not bool(alive[b0].any()):
                        break
                    idx = torch.nonzero(blended[b0:b0 + 64]).flatten() + b0
                    if idx.numel():
                        hist_b += torch.bincount(live[idx[::8]], minlength=65)
    tot = int(hist_b.sum()); cum = torch.cumsum(hist_b, 0)
    print('backward: chunks of eight blended entries:', tot)
    for k in (4, 8, 16, 32, 48, 63, 64):
        print('  chunks starting with <= %2d live pixels: %5.1f %%' % (k, 100.0 * int(cum[k]) / tot))
