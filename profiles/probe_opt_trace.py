"""Stage trace of opt_track_kernel for ONE pose (the service's chain; eight wavefronts): wall-clock stamps (100 MHz) of wavefront 0's
stages and of the first helper's, per gait cycle.  Needs the measurement build (-DFPE_OPT_TRACE; profiles/collect_opt_trace.sh
compiles it into scratch/); FPE_LIB points at it."""
import ctypes, os, sys
import numpy as np
import torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
lib = ctypes.CDLL(os.environ['FPE_LIB'])
lib.fpe_debug_opt_trace.argtypes = [ctypes.c_void_p, ctypes.c_size_t]
pl = FootholdPlanner(0)
trav, elev, res, poses, n, extra = synth.make_config("headline")
pl.gridmapCallback(trav, elev, res)
dev = torch.device("cuda:0")
d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1)).to(dev)
B = 1
d_ok = torch.ones(B * n, dtype=torch.uint8, device=dev)
d_f = torch.zeros(B * n * 4 * 32, dtype=torch.uint8, device=dev); d_c = torch.zeros(B * n * 240, dtype=torch.uint8, device=dev); d_g = torch.zeros(B, dtype=torch.uint8, device=dev)
s = torch.cuda.current_stream()
for it in range(5):
    pl.plan_opt_device(d_poses.data_ptr(), B, n, d_ok.data_ptr(), d_f.data_ptr(), d_c.data_ptr(), d_g.data_ptr(), stream=s.cuda_stream)
    torch.cuda.synchronize()
buf = np.zeros((256, 16), dtype=np.uint64)
assert lib.fpe_debug_opt_trace(buf.ctypes.data, buf.nbytes) == 0
t = buf[:n].astype(np.int64)
us = lambda a, b: (t[:, a] - t[:, b]) / 100.0
print(f"opt track, 1 pose, {n} gait cycles; per cycle (us), mean over the cycles [min .. max]")
rows = [("wavefront 0: cycle start -> gait-cycle submap, nominal index", 1, 0), ("  -> rows scanned (traversability loads)", 2, 1),
        ("  -> centroid method, gather, problem published", 3, 2), ("  -> columns decided and published", 4, 3),
        ("  -> the helpers' results are in", 5, 4), ("  -> merged, winner's point", 6, 5), ("  -> positions (and heights)", 7, 6), ("  -> commit", 8, 7),
        ("helper 1: problem seen -> its pairs' terms, its share of the Dab values", 14, 10), ("  -> shares exchanged, smallest violation", 15, 14),
        ("  -> survivors listed (end of the first part)", 11, 15), ("  -> columns seen (waiting for wavefront 0)", 12, 11),
        ("  -> list evaluated, result handed back", 13, 12)]
for name, a, b in rows:
    d = us(a, b)
    print(f"  {name:75s} {d.mean():6.2f} [{d.min():5.2f} .. {d.max():5.2f}]")
print(f"  cycle total {us(8, 0).mean():.2f}; cycle to cycle {np.diff(t[:, 0]).mean() / 100.0:.2f}; helper 1 sees the problem {((t[:, 10] - t[:, 3]) / 100.0).mean():.2f} after it is published")
