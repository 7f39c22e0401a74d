import os, sys, numpy as np, torch
sys.path.insert(0, '.')
NB = 65536
buf = torch.zeros(256*8*16 + NB*4, dtype=torch.int64, device='cuda')
os.environ["FPE_TRACE_PTR"] = str(buf.data_ptr())
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
cfgname = sys.argv[1] if len(sys.argv) > 1 else "cfg3"
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
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for it in range(3):
    buf.zero_(); torch.cuda.synchronize()
    e0.record()
    pl.plan_device(d_poses.data_ptr(), B, n, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(), d_ok.data_ptr(), d_st.data_ptr(), stream=torch.cuda.current_stream().cuda_stream, d_selected_ptr=d_sel.data_ptr(), d_pose_status_ptr=d_ps.data_ptr())
    e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
nblk = B if 'seq_kernel' in pl.describe_plan() else (B + 1) // 2  # (8-lane kernels: two poses per workgroup)
t = buf.cpu().numpy()[256*8*16:].reshape(NB, 4)[:nblk].astype(np.float64)
start, end = t[:,0], t[:,1]
hw = t[:,2].astype(np.int64); xcc = t[:,3].astype(np.int64)
t0 = start.min()
print(cfgname, "B", B, "kernel ms (events)", ms, "describe", pl.describe_plan())
print("span clk", end.max()-t0, "=> implied clock GHz", (end.max()-t0)/(ms*1e-3)/1e9)
print("start offsets clk: min %.0f median %.0f p90 %.0f max %.0f" % tuple(np.percentile(start-t0,[0,50,90,100])))
print("lifetime clk: median %.0f p90 %.0f max %.0f" % tuple(np.percentile(end-start,[50,90,100])))
late = (start - t0) > 0.1*(end.max()-t0)
print("blocks starting later than 10%% of the span: %d of %d" % (late.sum(), nblk))
# hw id: wave_id[3:0], simd_id[5:4], pipe[7:6], cu_id[11:8], sh_id[12], se_id[15:13]
cu = (hw>>8)&0xF; sh=(hw>>12)&1; se=(hw>>13)&7; simd=(hw>>4)&3
key = ((xcc&0xF)*8+se)*2*16+sh*16+cu
early = ~late
cnt = np.bincount(key[early], minlength=8*8*2*16)
print("CUs used", (cnt>0).sum(), "blocks per CU among the first-round blocks: min %d median %d max %d" % (cnt[cnt>0].min(), np.median(cnt[cnt>0]), cnt.max()))
sk = key*4+simd
cs = np.bincount(sk[early])
print("first-round waves per SIMD: min %d median %d max %d" % (cs[cs>0].min(), np.median(cs[cs>0]), cs.max()))
print("---- per XCD")
life = end - start
for x in sorted(set(xcc.tolist())):
    m = xcc == x
    s0 = start[m].min()
    so = start[m] - s0
    print("xcc", x, "blocks", m.sum(), "start offs p50 %.0f max %.0f" % (np.median(so), so.max()), "life p10 %.0f p50 %.0f p90 %.0f max %.0f" % tuple(np.percentile(life[m],[10,50,90,100])), "span %.0f" % (end[m].max()-s0),
          "CUs", len(set(key[m].tolist())))
# lifetime by position within launch order
idx = np.arange(nblk)
for lo in range(0, nblk, 512 if nblk > 2048 else 256):
    sl = slice(lo, lo + (512 if nblk > 2048 else 256))
    print("blocks %4d-%4d life p50 %.0f max %.0f  xcc set %s" % (lo, sl.stop - 1, np.median(life[sl]), life[sl].max(), sorted(set(xcc[sl].tolist()))[:8]))
# per-SIMD: sum of lifetimes vs max
sk = key*4+simd
import collections
d = collections.defaultdict(list)
for k_, l_ in zip(sk.tolist(), life.tolist()): d[k_].append(l_)
mx = np.array([max(v) for v in d.values()]); sm = np.array([sum(v) for v in d.values()]); nn = np.array([len(v) for v in d.values()])
print("SIMDs", len(d), "waves per SIMD min/max", nn.min(), nn.max(), "max-life per SIMD p50 %.0f max %.0f; corr(sum of lifetimes on the SIMD, max) %.2f" % (np.median(mx), mx.max(), np.corrcoef(sm, mx)[0,1]))
# hardware wave slot (HW_ID.WAVE_ID) by launch order: do the slots of a SIMD fill in order?
wid = hw & 0xF
step = 512 if nblk > 2048 else 256
print("wave slot by launch order:", "  ".join("%d-%d: %s" % (lo, lo + step - 1, dict(sorted(collections.Counter(wid[lo:lo + step].tolist()).items()))) for lo in range(0, nblk, step)))
