"""Per-workgroup timeline of filter_fused_kernel: start / end / phase marks (wall_clock64, 100 MHz) and the CU each workgroup ran on.
Needs the measurement build (profiles/collect_filters_timeline.sh compiles it: -DFPE_FUSED_TIMELINE); FPE_LIB points at it."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import synth, _capi
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
lib = ctypes.CDLL(os.environ['FPE_LIB'])
lib.fpe_debug_timeline.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
s = torch.cuda.current_stream()
rows, res = (1000, 0.02) if os.environ.get('FPE_PROBE_MAP', '0') == '0' else (2000, 0.01)
kind = os.environ.get('TL_MAP', 'rough')
_, elev = synth.rough_map(rows, rows, res, 5)
if kind == 'noise':
    rng = np.random.default_rng(1)
    ii, jj = np.meshgrid(np.arange(rows), np.arange(rows), indexing='ij')
    elev = (0.1 * ii * res + 0.05 * jj * res + 0.002 * rng.standard_normal((rows, rows))).astype(np.float32)
d_e = torch.from_numpy(elev).cuda()
d_t = torch.empty_like(d_e)
for _ in range(5):
    pl.traversability_device(d_e.data_ptr(), d_t.data_ptr(), rows, rows, res, d_layers_ptr=0, stream=s.cuda_stream)
torch.cuda.synchronize()
buf = np.zeros((8192, 16), dtype=np.uint64)
assert lib.fpe_debug_timeline(buf.ctypes.data, buf.nbytes) == 0
n = int((buf[:, 0] > 0).sum())
b = buf[:n].astype(np.int64)
t0 = b[:, 0].min()
start, m1, m2, end = [(b[:, k] - t0) / 100.0 for k in range(4)]  # us (100 MHz)
hw, xcc = b[:, 6], b[:, 7] & 0xF
cu = ((hw >> 8) & 0xF) | (((hw >> 13) & 0x7) << 4) | (((hw >> 12) & 1) << 7)
cuid = xcc * 256 + cu
print(f"{n} workgroups, kernel span {end.max():.1f} us; life mean {np.mean(end-start):.2f} us (step {np.mean(m1-start):.2f}, moments {np.mean(m2-m1):.2f}, walk {np.mean(end-m2):.2f}); p50 {np.median(end-start):.2f} p95 {np.percentile(end-start,95):.2f} max {np.max(end-start):.2f}")
if b[:, 8].max() > 0:
    ok = (b[:, 8] > 0) & (b[:, 9] > 0) & (b[:, 10] > 0) & (b[:, 11] > 0) & (b[:, 12] > 0) & (b[:, 13] > 0)
    q = lambda a, c: float(np.mean((b[ok, a] - b[ok, c]) / 100.0))
    print(f"  step: tile+tables {q(8,0):.2f}  runs+barrier {q(9,8):.2f}  fold {q(10,9):.2f}  edges+close {q(1,10):.2f} | moments: tile+tables {q(11,1):.2f}  z0+scans+barrier {q(12,11):.2f}  row records {q(13,12):.2f}  normals+stores {q(2,13):.2f}  walk vote/end {q(3,2):.2f}")
wk = end - m2
print(f"  walking phase per workgroup: {int((wk > 3).sum())} of {n} workgroups above 3 us, {int((wk > 10).sum())} above 10 us, longest {wk.max():.1f} us; workgroup lives above 1.5 x the median: {int(((end - start) > 1.5 * np.median(end - start)).sum())}")
print("distinct CUs", len(set(cuid.tolist())), "per XCD:", [int((xcc == x).sum()) for x in range(8)])
# per-XCD finish times and per-CU workgroup counts
for x in range(8):
    m = xcc == x
    cnts = np.bincount(cu[m])
    cnts = cnts[cnts > 0]
    print(f" xcd {x}: first start {start[m].min():.1f} last end {end[m].max():.1f}; CUs {len(cnts)} wgs/CU min {cnts.min()} max {cnts.max()}")
# concurrency over time (whole chip)
ts = np.arange(0, end.max(), 2.0)
conc = [(int(((start <= t) & (end > t)).sum())) for t in ts]
print("resident workgroups every 2 us:", conc)
# start-time histogram: when do the rounds begin
print("start times percentiles (us):", [round(float(np.percentile(start, p)), 1) for p in (0, 10, 25, 38, 50, 75, 90, 100)])
# per-CU max concurrency
mx = []
for c in set(cuid.tolist()):
    m = cuid == c
    ev = sorted([(float(a), 1) for a in start[m]] + [(float(e), -1) for e in end[m]])
    k = best = 0
    for _, d in ev:
        k += d; best = max(best, k)
    mx.append(best)
print("max concurrent workgroups per CU: histogram", np.bincount(mx))
