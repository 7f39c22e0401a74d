import sys, time, numpy as np
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import synth, _capi
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
trav, elev, res, poses, n, extra = synth.make_config("headline", B=64)
pl.gridmapCallback(trav, elev, res)
def t(f, reps=100):
    f(); f()
    ts=[]
    for _ in range(reps):
        t0=time.perf_counter(); f(); ts.append(time.perf_counter()-t0)
    return np.median(ts)*1e6
p1 = poses[:1]
out1 = pl.plan_outputs(1, 8, products=("nominal","cycle_ok","stance"))
print("plan 1 pose (nominal, ok, stance):", t(lambda: pl.plan(p1, 8, out=out1)))
ok = pl.plan(p1, 8)["cycle_ok"]
print("plan_opt 1 pose with cycle_ok:", t(lambda: pl.plan_opt(p1, 8, ok)))
print("service:", t(lambda: pl.globalFootholdPlan(8, p1["position"][0])))
print("service all tracks:", t(lambda: pl.globalFootholdPlan(8, p1["position"][0], all_tracks=True)))
pl.opt_params["use_inequality_constraints"] = 0
print("plan_opt 1 pose no constraints:", t(lambda: pl.plan_opt(p1, 8, ok)))
p64 = poses[:64]; ok64 = pl.plan(p64, 8)["cycle_ok"]
print("plan_opt 64 poses:", t(lambda: pl.plan_opt(p64, 8, ok64)))
big = synth.make_config("headline")[3]; okb = pl.plan(big, 8)["cycle_ok"]
print("plan_opt 4096 poses:", t(lambda: pl.plan_opt(big, 8, okb), 10))
