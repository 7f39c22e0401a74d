import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from quadrupedal_foothold_planner_amd import synth, _capi
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
dev = torch.device("cuda:0")
CTL = 64 * 256 + 128 + 2 * 8192 * 4 + 64 * 3 * 256 * 4
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
    d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1).copy()).to(dev)
    def run(): pl.plan_device(d_poses.data_ptr(), B, n, bufs[0].data_ptr(), bufs[1].data_ptr(), bufs[2].data_ptr(), bufs[3].data_ptr(), bufs[4].data_ptr(), stream=s.cuda_stream, d_selected_ptr=bufs[5].data_ptr(), d_pose_status_ptr=bufs[6].data_ptr())
    for _ in range(5): run()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(s); run(); e1.record(s); torch.cuda.synchronize()
    nb = B * 32 + 8192 * 16
    raw = (C.c_ubyte * nb)()
    rc = pl._lib.fpe_debug_mig_read(pl._h, C.c_void_p(s.cuda_stream), raw, C.c_int64(CTL + B * 208), C.c_int64(nb))
    hdr = (C.c_ubyte * 128)()
    pl._lib.fpe_debug_mig_read(pl._h, C.c_void_p(s.cuda_stream), hdr, C.c_int64(64 * 256), C.c_int64(128))
    nev = int(np.frombuffer(hdr, dtype=np.int32)[1])
    a = np.frombuffer(raw, dtype=np.int32)
    wl = a[:B * 8].reshape(B, 8).astype(np.int64)
    ev = a[B * 8:].reshape(8192, 4)[:min(nev, 8192)].astype(np.int64)
    t0 = wl[:, 0].min()
    start, end, exit_, takes, first_take = wl[:, 0] - t0, wl[:, 1] - t0, wl[:, 2] - t0, wl[:, 3], wl[:, 4] - t0
    us = lambda x: x / 100.0
    print(name, "launch %.1f us" % (e0.elapsed_time(e1) * 1e3), "rc", rc, "hand-overs", nev)
    print("  own pose ended/given: p10 %.0f p50 %.0f p90 %.0f max %.0f us" % tuple(us(np.percentile(end, [10, 50, 90, 100]))))
    print("  wavefront exit:       p10 %.0f p50 %.0f p90 %.0f max %.0f us" % tuple(us(np.percentile(exit_, [10, 50, 90, 100]))))
    print("  wavefronts that took poses: %d (takes %d); idle before leaving without a pose: p50 %.1f p90 %.1f max %.1f us" % ((takes > 0).sum(), takes.sum(),
          *us(np.percentile((exit_ - end)[takes == 0], [50, 90, 100]))))
    if nev:
        et = ev[:, 0] - t0
        print("  hand-over times: first %.0f p50 %.0f last %.0f us; by level" % tuple(us(np.percentile(et, [0, 50, 100]))), np.bincount(ev[:, 3] & 255, minlength=3),
              "next cycle p50", np.median(ev[:, 2]))
        w = (first_take - end)[takes > 0]
        print("  wait before the first take: p50 %.1f p90 %.1f max %.1f us" % tuple(us(np.percentile(w, [50, 90, 100]))))
        # poses handed over more than once
        c = np.bincount(ev[:, 1], minlength=B)
        print("  poses handed over: %d, more than once: %d, max %d" % ((c > 0).sum(), (c > 1).sum(), c.max()))
