"""Per-stage clock stamps of the one-wavefront-per-pose bit-window kernels (-DFPE_TRACE builds only; profiles/build_trace.sh).
usage on the GPU box: FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_seq.py cfg3 4096"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
buf = torch.zeros(256*8*16 + 65536*4, dtype=torch.int64, device='cuda')  # (+ the per-block area of -DFPE_TRACE_ALL_BLOCKS builds)
os.environ["FPE_TRACE_PTR"] = str(buf.data_ptr())
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "headline"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
trav, elev, res, poses, n, extra = synth.make_config(cfgname, B=B)
if "search_radius" in extra: pl.params["searchRadius"] = np.float32(extra["search_radius"])
if "max_leg_search_radius" in extra: pl.set_max_leg_search_radius(extra["max_leg_search_radius"])
pl.gridmapCallback(trav, elev, res)
d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1)).cuda()
nrec = B*n*4
d_nom = torch.zeros(nrec*32, dtype=torch.uint8, device='cuda'); d_cen = torch.zeros(nrec*32, dtype=torch.uint8, device='cuda')
d_def = torch.zeros(nrec*3, dtype=torch.float64, device='cuda'); d_ok = torch.zeros(B*n, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B*12, dtype=torch.float64, device='cuda')
d_sel = torch.zeros(nrec*16, dtype=torch.uint8, device='cuda'); d_ps = torch.zeros(B, dtype=torch.uint8, device='cuda')
for it in range(3):
    buf.zero_()
    pl.plan_device(d_poses.data_ptr(), B, n, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(), d_ok.data_ptr(), d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, d_selected_ptr=d_sel.data_ptr(), d_pose_status_ptr=d_ps.data_ptr())
    torch.cuda.synchronize()
t = buf.cpu().numpy()[:256*8*16].reshape(256, 8, 16).astype(np.float64)
names = {1:"leg start",2:"corners+submap",3:"loads issued",4:"rows arrived+scan",5:"default chk+zCentre",6:"centroid begin",7:"zDefault",8:"spiral",9:"centroid z+stores",10:"commit"}
pts = sorted(names)
prev = 1
for p in pts[1:]:
    dt = t[:, :min(n,8), p] - t[:, :min(n,8), prev]
    print(f"{names[prev]:22s} -> {names[p]:22s} mean {dt.mean():8.0f} clk  median {np.median(dt):8.0f}  p90 {np.percentile(dt,90):8.0f}")
    prev = p
tot = t[:, :min(n,8), 9] - t[:, :min(n,8), 1]
print("cycle total mean", tot.mean(), "median", np.median(tot))
print("cycle-to-cycle", np.mean(t[:,1:,0]-t[:,:-1,0]))
print("kernel span per block", np.mean(t[:,n-1 if n<8 else 7,10]-t[:,0,0]))


life = t[:, 6, 15] - t[:, 6, 14]
print("wavefront lifetime over the first 256 poses: mean", life.mean(), "median", np.median(life), "p90", np.percentile(life, 90), "max", life.max(), " mean/max", life.mean() / life.max())

sp = t[:, 6, 13]; ns = t[:, 6, 12]; nf = t[:, 6, 11]
order = np.argsort(-life)
print("slowest poses: lifetime, clocks in spiral, searches, searches without a hit")
for b in order[:6]: print("  ", b, life[b], sp[b], ns[b], nf[b])
print("median pose:", np.median(life), np.median(sp), np.median(ns), np.median(nf), " corr(life, spiral clocks)", np.corrcoef(life, sp)[0,1])


ok = (t[:, 3, 14] > t[:, 2, 15]) & (t[:, 2, 15] >= t[:, 2, 14]) & (t[:, 2, 14] > t[:, 0, 15]) & (t[:, 0, 15] > t[:, 0, 14])
print("last successful search per pose (n=%d): P rows %.0f, erosion %.0f, ring skip %.0f, rounds %.0f clk; start round %.1f, hit round %.1f" % (ok.sum(),
      np.mean((t[:,0,15]-t[:,0,14])[ok]), np.mean((t[:,2,14]-t[:,0,15])[ok]), np.mean((t[:,2,15]-t[:,2,14])[ok]), np.mean((t[:,3,14]-t[:,2,15])[ok]), np.mean(t[:,4,14][ok]), np.mean(t[:,4,15][ok])))

m8 = min(n, 8)
print("tail of a leg: spiral end -> sums start %.0f, three height sums %.0f, -> staged results %.0f, record stores (to the kernel's stamp 9) %.0f" % (
    np.median(t[:, :m8, 11] - t[:, :m8, 8]), np.median(t[:, :m8, 12] - t[:, :m8, 11]), np.median(t[:, :m8, 13] - t[:, :m8, 12]), np.median(t[:, :m8, 9] - t[:, :m8, 13])))

fl, nf = t[:, 7, 14], t[:, 7, 15]
if nf.max() > 0:
    pro = t[:, 5, 14] - t[:, 6, 14]
    legs, nl = t[:, 5, 15], t[:, 1, 14]
    rest = life - pro - legs - fl
    print("budget per pose (clocks at four wavefronts per SIMD, mean over the first 256 poses): prologue %.0f + %.1f leg searches x %.0f = %.0f + %.1f flushes x %.0f = %.0f "
          "+ everything between the leg searches (feet-polygon centres, commit, hand-offs, priority levels) %.0f = lifetime %.0f" % (
              pro.mean(), nl.mean(), (legs / np.maximum(nl, 1)).mean(), legs.mean(), nf.mean(), (fl / np.maximum(nf, 1)).mean(), fl.mean(), rest.mean(), life.mean()))
    print("shares of the lifetime: prologue %.3f, leg searches %.3f, flushes %.3f, between %.3f" % (pro.mean() / life.mean(), legs.mean() / life.mean(), fl.mean() / life.mean(), rest.mean() / life.mean()))
