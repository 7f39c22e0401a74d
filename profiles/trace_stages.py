"""Per-stage clock stamps of the 8-lane bit-window kernel (s_memtime, -DFPE_TRACE builds only): where a gait cycle's time goes.
usage on the GPU box: FPE_LIB=<trace build of libfpe.so> python3 profiles/trace_stages.py headline 4096"""
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
# stamp points of the 3x3 fast path (leg_fast8m); build the library with -DFPE_TRACE and pass it as FPE_LIB
names = {0:"cycle start",1:"feet-polygon centres",2:"x pass + submap",3:"loads issued",4:"rows arrived + row scan",5:"default check + deposits",7:"centroid selects",8:"spiral (if any leg needs it)",9:"unit fields",10:"commit"}
pts = sorted(names)
prev = 0
for p in pts[1:]:
    dt = t[:, :min(n,8), p] - t[:, :min(n,8), prev]
    print(f"{names[prev]:22s} -> {names[p]:22s} mean {dt.mean():8.0f} clk  median {np.median(dt):8.0f}  p90 {np.percentile(dt,90):8.0f}")
    prev = p
tot = t[:, :min(n,8), 10] - t[:, :min(n,8), 0]
print("cycle total mean", tot.mean(), "median", np.median(tot))
print("cycle-to-cycle", np.mean(t[:,1:,0]-t[:,:-1,0]))
print("kernel span per block", np.mean(t[:,n-1 if n<8 else 7,10]-t[:,0,0]))

print("prologue: entry -> statics/LUT/stance", np.mean(t[:,1,12]-t[:,1,11]), " gate", np.mean(t[:,1,13]-t[:,1,12]), " ytab fill", np.mean(t[:,1,14]-t[:,1,13]), " first cycle start after entry", np.mean(t[:,0,0]-t[:,1,11]))
if t[:,3,11].any():
    print("prologue detail: entry -> pose arrived", np.mean(t[:,3,11]-t[:,1,11]), " -> rank-table head", np.mean(t[:,3,12]-t[:,3,11]), " -> per-leg constants", np.mean(t[:,3,13]-t[:,3,12]),
          " -> offset table copied", np.mean(t[:,3,14]-t[:,3,13]), " -> stance in LDS", np.mean(t[:,1,12]-t[:,3,14]))
print("flush", np.mean(t[:,2,12]-t[:,2,11]), " last commit -> flush start", np.mean(t[:,2,11]-t[:,7,10]))
life = t[:, 2, 12] - t[:, 1, 11]
print("wavefront lifetime (entry -> last flush) over the first 256 blocks: mean", life.mean(), "median", np.median(life), "p90", np.percentile(life, 90), "max", life.max(), " mean/max", life.mean() / life.max())
# cold start: the same stages for the first cycle against the later ones (instruction and data caches are cold per launch)
nc = min(n, 8)
print("cycle total by cycle index:", " ".join(f"{tot[:, c].mean():.0f}" for c in range(nc)))
prev = 0
for p in pts[1:]:
    dt = t[:, :nc, p] - t[:, :nc, prev]
    print(f"  {names[prev]:22s} -> {names[p]:22s} cycle 0 {dt[:,0].mean():7.0f}  cycle 1 {dt[:,1].mean() if nc>1 else 0:7.0f}  cycles 2.. {dt[:,2:].mean() if nc>2 else 0:7.0f}")
    prev = p
# which wavefronts are the slow ones: lifetime against the per-cycle stage that varies
order = np.argsort(life)
cyc_tot = t[:, :nc, 10] - t[:, :nc, 0]
sp = t[:, :nc, 8] - t[:, :nc, 7]
xp = t[:, :nc, 2] - t[:, :nc, 1]
print("slowest 8 wavefronts: lifetime, sum of cycle totals, spiral stage per cycle, x pass per cycle")
for bidx in order[-8:]:
    print(f"  block {bidx}: {life[bidx]:.0f} {cyc_tot[bidx].sum():.0f}  spiral {' '.join(f'{v:.0f}' for v in sp[bidx])}  x {' '.join(f'{v:.0f}' for v in xp[bidx])}")
print("fastest 4:")
for bidx in order[:4]:
    print(f"  block {bidx}: {life[bidx]:.0f} {cyc_tot[bidx].sum():.0f}  spiral {' '.join(f'{v:.0f}' for v in sp[bidx])}  x {' '.join(f'{v:.0f}' for v in xp[bidx])}")
print("correlation of lifetime with the summed spiral stage", np.corrcoef(life, sp.sum(1))[0, 1], " with the summed x pass", np.corrcoef(life, xp.sum(1))[0, 1])
print("spiral stage histogram (clk):", np.histogram(sp, bins=[0, 200, 600, 1000, 1400, 1800, 2500, 4000, 100000])[0])
