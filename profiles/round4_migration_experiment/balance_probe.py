"""How much of the one-wavefront-per-pose kernels' time is imbalance between SIMDs: the time of the real batch against
the mean of batches made of 4096 copies of ONE of its poses (no variance between wavefronts)."""
import sys, numpy as np
sys.path.insert(0, '.')
import torch
from quadrupedal_foothold_planner_amd import synth, _capi
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
dev = torch.device("cuda:0")
for name in sys.argv[1:] or ["cfg3", "cfg5"]:
    trav, elev, res, poses, n, extra = synth.make_config(name)
    pl.params = _capi.params_yaml()
    if "search_radius" in extra: pl.params["searchRadius"] = np.float32(extra["search_radius"])
    if "max_leg_search_radius" in extra: pl.set_max_leg_search_radius(extra["max_leg_search_radius"])
    pl.gridmapCallback(trav, elev, res)
    B = poses.shape[0]
    bufs = [torch.zeros(B * n * 4 * 32, dtype=torch.uint8, device=dev), torch.zeros(B * n * 4 * 32, dtype=torch.uint8, device=dev), torch.zeros(B * n * 12, dtype=torch.float64, device=dev),
            torch.zeros(B * n, dtype=torch.uint8, device=dev), torch.zeros(B * 12, dtype=torch.float64, device=dev), torch.zeros(B * n * 4 * 16, dtype=torch.uint8, device=dev), torch.zeros(B, dtype=torch.uint8, device=dev)]
    s = torch.cuda.current_stream()
    def timeit(p, reps=6):
        d_poses = torch.from_numpy(p.view(np.uint8).reshape(-1).copy()).to(dev)
        def run(): pl.plan_device(d_poses.data_ptr(), B, n, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(), stream=s.cuda_stream, d_selected_ptr=bufs[5].data_ptr(), d_pose_status_ptr=bufs[6].data_ptr())
        for _ in range(2): run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(s)
        for _ in range(reps): run()
        e1.record(s); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps * 1e3
    for _ in range(3): timeit(poses)
    real = timeit(poses, 12)
    rng = np.random.default_rng(5)
    idx = rng.choice(B, 48, replace=False)
    ts = []
    for i in idx:
        ts.append(timeit(np.repeat(poses[i:i + 1], B)))
    ts = np.array(ts)
    real2 = timeit(poses, 12)
    print(f"{name}: real batch {real:.1f} / {real2:.1f} us; copies of one pose: mean {ts.mean():.1f} median {np.median(ts):.1f} min {ts.min():.1f} max {ts.max():.1f} us  => balanced/real = {ts.mean() / real2:.3f}", flush=True)
