"""Where a one-pose service call spends its time on the HOST (-DFPE_HOST_TIMING build: FPE_LIB=scratch/libfpe_ht.so)."""
import ctypes, os, sys, time, numpy as np
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
lib = ctypes.CDLL(os.environ['FPE_LIB'])
trav, elev, res, poses, n, extra = synth.make_config("headline", B=64)
pl.gridmapCallback(trav, elev, res)
pos = poses["position"][0].copy()
out = (ctypes.c_double * 16)()
names = ["", "lease + reserve", "prepare_call", "opt constants, pose copy, device pointers", "chain queued", "plan kernel queued", "main stream synchronised",
         "side stream synchronised", "results copied out"]
poll = int(os.environ.get("FPE_PROBE_POLL", "1"))
for label, tun in (("enforcing default (chain beside the plan kernel)", dict(service_opt_gate=2, service_overlap=1, service_poll=poll)),
                   ("exact gates only (plan kernel alone)", dict(service_opt_gate=0))):
    with pl.tuning(**tun):
        for _ in range(20): pl.globalFootholdPlan(8, pos)
        lib.fpe_debug_host_timing(out, 1)
        ts = []
        for _ in range(200):
            t0 = time.perf_counter(); pl.globalFootholdPlan(8, pos); ts.append(time.perf_counter() - t0)
        lib.fpe_debug_host_timing(out, 1)
    inside = sum(out[k] for k in range(1, 9))
    print(f"{label}: median call {np.median(ts)*1e6:.1f} us through ctypes; inside plan_host {inside:.1f} us over {int(out[0])} calls:")
    for k in range(1, 9):
        print(f"    -> {names[k]:45s} {out[k]:7.2f} us")
