"""Per-stage clock stamps of the generic 8-lane bit-window kernel (-DFPE_TRACE builds only; profiles/build_trace.sh).
usage on the GPU box: FPE_LIB=scratch/libfpe_trace.so python3 profiles/trace_stages_generic.py cfg4 32768"""
import os, sys, numpy as np, torch
sys.path.insert(0, '.')
buf = torch.zeros(256*8*16 + 65536*4, dtype=torch.int64, device='cuda')  # (+ the per-block area of -DFPE_TRACE_ALL_BLOCKS builds)
os.environ["FPE_TRACE_PTR"] = str(buf.data_ptr())
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg4"
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
print(pl.describe_plan())
t = buf.cpu().numpy()[:256*8*16].reshape(256, 8, 16).astype(np.float64)
names = {0:"cycle start",1:"centres",2:"x pass+submap",3:"loads issued",4:"rows arrived+scan",5:"default chk+zCentre",6:"centroid begin",7:"zDefault",8:"spiral",9:"centroid z+stores",10:"commit"}
pts = sorted(names)
nc = min(n, 8)
prev = 0
tot = t[:, :nc, 10] - t[:, :nc, 0]
for p in pts[1:]:
    dt = t[:, :nc, p] - t[:, :nc, prev]
    print(f"{names[prev]:22s} -> {names[p]:22s} mean {dt.mean():8.0f} clk ({100*dt.mean()/tot.mean():5.1f} %)  median {np.median(dt):8.0f}  p90 {np.percentile(dt,90):8.0f}")
    prev = p
print("cycle total mean", tot.mean(), "median", np.median(tot))
print("cycle-to-cycle", np.mean(t[:,1:nc,0]-t[:,:nc-1,0]))
print("flush", np.mean(t[:,2,12]-t[:,2,11]))
if n >= 8:
    sel = t[:, 4:8, :]
    okm = (sel[:, :, 11] > sel[:, :, 7]) & (sel[:, :, 12] > sel[:, :, 11]) & (sel[:, :, 13] > sel[:, :, 12]) & (sel[:, :, 8] > sel[:, :, 13])
    def mm(a): return float(np.mean(a[okm])) if okm.any() else float('nan')
    print("spiral detail (cycles 4-7, n=%d): entry %.0f, P rows (incl. rectangle bounds) %.0f, erosion %.0f, scan+exit %.0f clk" % (
        okm.sum(), mm(sel[:, :, 11] - sel[:, :, 7]), mm(sel[:, :, 12] - sel[:, :, 11]), mm(sel[:, :, 13] - sel[:, :, 12]), mm(sel[:, :, 8] - sel[:, :, 13])))
