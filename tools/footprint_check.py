"""Developer check (CPU, numpy float32): the row-interval footprint arithmetic of csrc/binning.hip (subtile_count_kernel)
against the per-pixel alpha rule evaluated by brute force on one 64x64 cell -- no reached sub-tile may be missed ("bad" must be
0); "kept" vs "exact(pixel)" shows how tight the test is, "box" what the bounding rect alone keeps.  python tools/footprint_check.py"""
import numpy as np
rng = np.random.default_rng(0)
f32 = np.float32
LOG2E = f32(1.4426950408889634)
def run(N, big):
    bad = 0; kept = 0; exact = 0; box = 0
    for it in range(N):
        # random covariance (with the 0.3 low-pass), opacity, centre
        s1 = np.exp(rng.uniform(np.log(0.3), np.log(60.0 if big else 6.0))); s2 = np.exp(rng.uniform(np.log(0.3), np.log(60.0 if big else 6.0)))
        th = rng.uniform(0, np.pi)
        R = np.array([[np.cos(th), -np.sin(th)], [np.sin(th), np.cos(th)]])
        cov = R @ np.diag([s1 * s1, s2 * s2]) @ R.T + 0.3 * np.eye(2)
        a, b, c = cov[0, 0], cov[0, 1], cov[1, 1]
        det = a * c - b * b
        ca, cb, cc = c / det, -b / det, a / det
        op = rng.uniform(0.01, 1.0) if rng.uniform() < 0.5 else 1.0
        if 255 * op < 1: continue
        px, py = rng.uniform(0, 64, 2)
        A = f32(-0.5) * LOG2E * f32(ca); B = -LOG2E * f32(cb); C = f32(-0.5) * LOG2E * f32(cc)
        px = f32(px); py = f32(py)
        # brute force per pixel (float32, same formula as the kernels), over an 8x8 grid of sub-tiles = one cell
        X = np.arange(64, dtype=np.float32); Y = np.arange(64, dtype=np.float32)
        dx = (px - X)[None, :]; dy = (py - Y)[:, None]
        p2 = (A * dx) * dx + ((C * dy) * dy + (B * dx) * dy)
        alpha = np.minimum(f32(0.99), f32(op) * np.exp2(p2.astype(np.float32)))
        ok = (alpha >= f32(1 / 255.)) & (p2 <= 0)
        truth = ok.reshape(8, 8, 8, 8).any(axis=(1, 3))        # [sub y, sub x]
        # bounding box (what preprocess gives), in sub-tiles, clipped to the cell
        tau2 = 2 * np.log(255 * op) + 1e-3
        ex = np.sqrt(tau2 * a) * 1.001 + 0.01; ey = np.sqrt(tau2 * c) * 1.001 + 0.01
        x0 = max(int(np.floor((px - ex) / 8)), 0); x1 = min(int(np.floor(np.ceil(px + ex) / 8)) + 1, 8)
        y0 = max(int(np.floor((py - ey) / 8)), 0); y1 = min(int(np.floor(np.ceil(py + ey) / 8)) + 1, 8)
        if x1 <= x0 or y1 <= y0: continue
        # ---- row-interval method in float32 ----
        xl0 = f32(x0 * 8) - px; yl0 = f32(y0 * 8) - py
        xm = max(abs(xl0), abs(xl0 + f32((x1 - x0) * 8))); ym = max(abs(yl0), abs(yl0 + f32((y1 - y0) * 8)))
        mag = abs(A) * xm * xm + abs(B) * xm * ym + abs(C) * ym * ym
        thr = f32(-np.log2(f32(255.0) * f32(op))) - f32(1e-3) - f32(1e-5) * f32(mag)
        rA = f32(1) / A; rC = f32(1) / C
        As = A - f32(0.25) * B * B * rC
        mask = np.zeros((8, 8), bool)
        if not (A < 0 and C < 0 and As < 0):
            mask[y0:y1, x0:x1] = True
        else:
            kA = f32(-0.5) * B * rA; kC = f32(-0.5) * B * rC
            Xf = np.sqrt(thr / As, dtype=np.float32)
            ysr = kC * Xf
            X2 = thr * rA
            D4 = C * As * rA * rA
            for y in range(y0, y1):
                yl = yl0 + f32((y - y0) * 8); yh = yl + f32(7)
                yR = min(max(ysr, yl), yh); yL = min(max(-ysr, yl), yh)
                hR2 = X2 - D4 * yR * yR; hL2 = X2 - D4 * yL * yL
                if hR2 < 0 or hL2 < 0: continue
                xb = kA * yR + np.sqrt(hR2, dtype=np.float32); xa = kA * yL - np.sqrt(hL2, dtype=np.float32)
                m = f32(1e-3) * (f32(1) + max(abs(xa), abs(xb)))
                clo = x0 + int(np.ceil((xa - m - f32(7) - xl0) * f32(0.125)))
                chi = x0 + int(np.floor((xb + m - xl0) * f32(0.125)))
                clo = max(clo, x0); chi = min(chi, x1 - 1)
                if chi >= clo: mask[y, clo:chi + 1] = True
        miss = truth & ~mask
        # truth outside the rect cannot happen if the rect is right
        if miss.any(): bad += 1; print('MISS', it, s1, s2, th, op, px, py, np.argwhere(miss)[:3])
        kept += mask.sum(); exact += truth.sum(); box += (y1 - y0) * (x1 - x0)
    print('N', N, 'big', big, 'bad', bad, 'box', box, 'kept', kept, 'exact(pixel)', exact)
run(4000, False)
run(3000, True)
