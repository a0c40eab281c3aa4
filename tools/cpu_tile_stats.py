"""Developer tool (CPU only, uses the oracle): how the tile edge (16 / 8 / 4 px) changes what a C3 view costs -- entries in
the exact-footprint lists, entries walked until every pixel of the tile is dead, (pixel, splat) pairs evaluated, pairs
that blend.  Usage: python tools/cpu_tile_stats.py [ring view]   (~1 min)

C3 view 0:   16 px: 373 k entries, 318 k walked, 81.4 M pairs, lane efficiency 0.079
              8 px: 709 k entries, 502 k walked, 32.2 M pairs, lane efficiency 0.200   (this design)
              4 px: 1.68 M entries, 972 k walked, 15.5 M pairs, lane efficiency 0.413
              8 px wave, four independent 16-lane quad streams: 299 k trips (vs 502 k walked), lane efficiency 0.335
              (per-ENTRY count; the exact model quad_stream_model.c (removed in round 4; last in commit 44f775f under oracle/c/) -- groups of four, batches of 64, box masks --
               gives 95 k vs 127 k loop trips: x0.75)"""
import sys, os, math, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from exavatar_release_amd import scenes
from oracle import raster_oracle as ro
torch.set_num_threads(8)
H=W=1024
view = int(sys.argv[1]) if len(sys.argv)>1 else 0
assets = scenes.dist_b_avatar(150000, seed=0)
cam = scenes.ring_camera(H, W, view, 200)
s = ro.settings_from_camera(cam, (H,W), torch.ones(3))
dtype=torch.float32
with torch.no_grad():
    pre = ro.preprocess(assets['mean_3d'], None, assets['opacity'], assets['scale'], assets['rotation'], None, s, dtype)
    sorted_idx, ranges = ro.build_tile_lists(pre, dtype)
    gx, gy = pre['grid']
    px, py, conic = pre['px'], pre['py'], pre['conic']
    opac = assets['opacity'].view(-1)
    res = {}
    for TS in (16, 8, 4):
        res[TS] = dict(lists=0, walked=0, pairs=0, useful=0, tiles=0, blended=0)
    quad_trips = 0          # 8 px wave, four independent 16-lane streams (one per 4x4 quad): sum over sub-tiles of max_q walked_q
    quad_blended = 0
    half_trips = 0          # ... two 32-lane streams (8 x 4 px halves)
    for t in range(gx*gy):
        s0,e0 = ranges[t].tolist()
        if e0==s0: continue
        w4 = [[0]*4 for _ in range(4)]
        tx,ty = t%gx, t//gx
        ids = sorted_idx[s0:e0]
        X = torch.arange(tx*16, tx*16+16, dtype=dtype).repeat(16); Y = torch.arange(ty*16, ty*16+16, dtype=dtype).repeat_interleave(16)
        dx = px[ids][:,None]-X[None,:]; dy = py[ids][:,None]-Y[None,:]
        cn = conic[ids]
        power = -0.5*(cn[:,0:1]*dx*dx + cn[:,2:3]*dy*dy) - cn[:,1:2]*dx*dy
        a = (opac[ids][:,None]*torch.exp(power)).clamp(max=0.99)
        valid = (power<=0)&(a>=1/255.)
        av = torch.where(valid,a,torch.zeros_like(a))
        Tin = torch.cumprod(1-av,0)
        Tex = torch.cat((torch.ones(1,256),Tin[:-1]),0)
        alive = Tex>=1e-4
        # stop rule: a splat is not blended if T(1-a) < 1e-4
        blend = valid & alive & (Tin>=1e-4)
        valid = valid.view(-1,16,16); alive = alive.view(-1,16,16); blend = blend.view(-1,16,16)
        for TS in (16, 8, 4):
            n = 16//TS
            for sy in range(n):
                for sx in range(n):
                    v = valid[:, sy*TS:(sy+1)*TS, sx*TS:(sx+1)*TS].reshape(len(ids), -1)
                    al = alive[:, sy*TS:(sy+1)*TS, sx*TS:(sx+1)*TS].reshape(len(ids), -1)
                    bl = blend[:, sy*TS:(sy+1)*TS, sx*TS:(sx+1)*TS].reshape(len(ids), -1)
                    inlist = v.any(1)                       # exact footprint list of this tile
                    L = int(inlist.sum())
                    if L == 0: continue
                    r = res[TS]; r['tiles']+=1; r['lists']+=L
                    alv = al[inlist]                       # per list entry, which pixels are still alive when it arrives
                    any_alive = alv.any(1)
                    walked = int(any_alive.sum())          # entries walked until every pixel is dead (alive is monotone)
                    r['walked']+=walked; r['pairs']+=walked*TS*TS
                    if TS == 4: w4[sy][sx] = walked
                    r['useful']+=int(bl[inlist].sum()); r['blended']+=int(bl[inlist].any(1).sum())
        for sy in range(2):
            for sx in range(2):
                q = [w4[2*sy][2*sx], w4[2*sy][2*sx+1], w4[2*sy+1][2*sx], w4[2*sy+1][2*sx+1]]
                quad_trips += max(q)
                # 8 x 4 halves (upper / lower): a half walks the union of its two quads' entries: bounded by their sum, at least their max
                half_trips += max(max(q[0], q[1]), max(q[2], q[3]))
    for TS in (16,8,4):
        r=res[TS]; print(TS, r, 'lane eff %.3f' % (r['useful']/max(1,r['pairs'])))
    print('8 px wave with four independent quad streams: %d trips (sum over sub-tiles of the longest quad stream) vs %d walked entries today; '
          'pairs evaluated %d (x16 lanes) -> lane efficiency %.3f' % (quad_trips, res[8]['walked'], quad_trips * 64,
                                                                    res[4]['useful'] / max(1, quad_trips * 64)))
