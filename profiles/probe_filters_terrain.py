"""Filter chain on two extreme terrains: exactly flat ground (every cell takes the literal walks) and a tilted plane with noise (none does)."""
import sys, numpy as np, torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
s = torch.cuda.current_stream()
for rows, res in ((1000, 0.02), (1000, 0.01)):
    for kind in ("flat", "tilted-noise"):
        rng = np.random.default_rng(1)
        ii, jj = np.meshgrid(np.arange(rows), np.arange(rows), indexing="ij")
        elev = np.zeros((rows, rows), np.float32) if kind == "flat" else (0.1 * ii * res + 0.05 * jj * res + rng.normal(0, 1e-3, (rows, rows))).astype(np.float32)
        d_e = torch.from_numpy(elev).cuda(); d_t = torch.empty_like(d_e)
        def run(): pl.traversability_device(d_e.data_ptr(), d_t.data_ptr(), rows, rows, res, stream=s.cuda_stream)
        for _ in range(2): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(s)
        for _ in range(5): run()
        e1.record(s); torch.cuda.synchronize()
        print(rows, res, kind, f"{e0.elapsed_time(e1)/5:.3f} ms per chain")
