"""Where do the slow timed blocks of bench.py come from?  Headline launches in blocks of 50 between synchronisations for
~4 s: every block's HIP-event time and wall-clock start; prints the outliers (> 1.5 x the median block), their spacing in time and what the
blocks right after them look like."""
import sys, time
import numpy as np, torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
trav, elev, res, poses, n, extra = synth.make_config("headline")
pl.gridmapCallback(trav, elev, res)
B = poses.shape[0]
dev = torch.device("cuda:0")
d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1)).to(dev)
nrec = B * n * 4
d_nom = torch.zeros(nrec * 32, dtype=torch.uint8, device=dev); d_ok = torch.zeros(B * n, dtype=torch.uint8, device=dev)
d_sel = torch.zeros(nrec * 16, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream()
def block(k=50):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); t0 = time.perf_counter(); e0.record(s)
    for _ in range(k):
        pl.plan_device(d_poses.data_ptr(), B, n, d_nominal_ptr=d_nom.data_ptr(), d_cycle_ok_ptr=d_ok.data_ptr(), stream=s.cuda_stream, d_selected_ptr=d_sel.data_ptr())
    e1.record(s); torch.cuda.synchronize()
    return t0, e0.elapsed_time(e1) / k * 1e3, (time.perf_counter() - t0) / k * 1e6
for _ in range(5): block()
rows = []
t_start = time.perf_counter()
while time.perf_counter() - t_start < 4.0:
    rows.append(block())
t0s = np.array([r[0] for r in rows]) - t_start; ev = np.array([r[1] for r in rows]); wall = np.array([r[2] for r in rows])
med = np.median(ev)
out = np.nonzero(ev > 1.5 * med)[0]
print(f"{len(rows)} blocks of 50 launches in {t0s[-1]:.2f} s: median {med:.2f} us per launch by HIP events ({np.median(wall):.2f} by the wall clock), p10 {np.percentile(ev,10):.2f} p90 {np.percentile(ev,90):.2f}")
print(f"outlier blocks (> 1.5 x median): {len(out)} at t = {[round(float(t0s[i]), 3) for i in out]} s; their per-launch times by events {[round(float(ev[i]), 1) for i in out]} / by wall {[round(float(wall[i]), 1) for i in out]}")
if len(out) > 1:
    print("spacing between outliers (s):", [round(float(x), 3) for x in np.diff(t0s[out])])
for i in out[:4]:
    print(f"  around block {i}: events", [round(float(x), 1) for x in ev[max(0, i - 3):i + 8]])
print("first 12 blocks after the warm-up:", [round(float(x), 1) for x in ev[:12]])
