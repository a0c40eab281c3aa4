"""Developer tool (CPU only, uses the oracle): how much of a C3 view's per-sub-tile lists the forward pass enters, how
many of those entries are blended into at least one pixel (what the backward pass keeps after compaction) and what cheaper
conservative tests -- footprint bounding box / exact footprint against the live mask at batch, 16-entry or 4-entry
granularity -- would keep.  Usage: python tools/cpu_blend_stats.py [ring view]   (~10 s)

C3, view 0: 893 k instances, 690 k in entered batches, 409 k blended; box test with the live mask per 16 entries keeps 573 k,
the exact footprint 426 k -- hence the per-batch masks written by the forward pass (csrc/render_fwd.hip)."""
import sys, os, math, torch, time
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
    px, py, conic, z = pre['px'], pre['py'], pre['conic'], pre['depth']
    a2, b2, c2 = pre['cov2']
    opac = assets['opacity'].view(-1)
    tau2 = 2*torch.log(255*opac)+1e-3
    ex = torch.sqrt(tau2*a2)*1.001+0.01
    ey = torch.sqrt(tau2*c2)*1.001+0.01
    bx0 = torch.floor(px-ex); bx1 = torch.ceil(px+ex); by0=torch.floor(py-ey); by1=torch.ceil(py+ey)
    tot = dict(fp_any=0, inst=0, entered=0, exact=0, box_batch=0, box_c16=0, box_g4=0, fp_c16=0, fp_batch=0, nb_ent=0)
    hist = torch.zeros(65, dtype=torch.long)
    for t in range(gx*gy):
        s0,e0 = ranges[t].tolist()
        if e0==s0: continue
        tx,ty = t%gx, t//gx
        ids = sorted_idx[s0:e0]
        for sy in range(2):
            for sx in range(2):
                ox = tx*16+sx*8; oy = ty*16+sy*8
                sel = (bx0[ids] <= ox+7) & (bx1[ids] >= ox) & (by0[ids] <= oy+7) & (by1[ids] >= oy)
                l = ids[sel]
                n = l.numel()
                if n==0: continue
                X = torch.arange(ox, ox+8, dtype=dtype).repeat(8); Y = torch.arange(oy, oy+8, dtype=dtype).repeat_interleave(8)
                dx = px[l][:,None]-X[None,:]; dy = py[l][:,None]-Y[None,:]
                cn = conic[l]
                power = -0.5*(cn[:,0:1]*dx*dx + cn[:,2:3]*dy*dy) - cn[:,1:2]*dx*dy
                a = (opac[l][:,None]*torch.exp(power)).clamp(max=0.99)
                valid = (power<=0)&(a>=1/255.)
                av = torch.where(valid,a,torch.zeros_like(a))
                Tin = torch.cumprod(1-av,0)
                Tex = torch.cat((torch.ones(1,64),Tin[:-1]),0)
                alive = Tex>=1e-4
                touch = valid & alive
                anyt = touch.any(1)
                alive_any = alive.any(1)
                nb = (n+63)//64
                ent=0
                for b in range(nb):
                    if not alive_any[b*64]: break
                    ent+=1
                m = min(n, ent*64)
                tot['fp_any']+=int(valid.any(1).sum())
                tot['inst']+=n; tot['entered']+=m; tot['exact']+=int(anyt[:m].sum()); tot['nb_ent']+=ent
                inbox = (X[None,:] >= bx0[l][:,None]) & (X[None,:] <= bx1[l][:,None]) & (Y[None,:] >= by0[l][:,None]) & (Y[None,:] <= by1[l][:,None])
                idx = torch.arange(m)
                for name, gran in (('box_batch',64),('box_c16',16),('box_g4',4)):
                    st = (idx//gran)*gran
                    al = alive[st]           # alive at start of the granule
                    tot[name] += int((inbox[:m] & al).any(1).sum())
                for name, gran in (('fp_batch',64),('fp_c16',16)):
                    st = (idx//gran)*gran
                    al = alive[st]
                    tot[name] += int((valid[:m] & al).any(1).sum())
                # histogram of compacted batch sizes with box_c16
                st = (idx//16)*16
                flag = (inbox[:m] & alive[st]).any(1)
                for b in range(ent):
                    hist[int(flag[b*64:(b+1)*64].sum())]+=1
    print(view, tot)
    print('hist compacted batch size (box_c16):', hist.tolist())
