#!/usr/bin/env python3
"""Generate the committed golden vectors (tests/golden/*.npz).

PROVENANCE: the reference (lukechencqu/quadrupedal_foothold_planner) has no tests, no golden
vectors and cannot be built in this image (needs ROS1, grid_map_core, Eigen, NLopt), so these
vectors are produced by THIS repo's oracle (oracle/, the CPU restatement of the reference's
algorithm) — "parity unpinned" beyond the analytic KATs in tests/test_oracle_kat.py.  They freeze
the oracle's behaviour so that later edits to either the oracle or the engine are caught.

Inputs are tiny seeded maps (<= 96x96 cells) with steps, holes, NaN and +-inf; outputs are the full
plan products.  Run from the repo root:  python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from oracle import fpo  # noqa: E402
from tests.conftest import oracle_poses, yaml_params  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def small_map(rows, cols, res, seed, bad=0.04):
    rng = np.random.default_rng(seed)
    trav = np.ones((rows, cols), np.float32)
    elev = (0.03 * rng.standard_normal((rows, cols))).astype(np.float32)
    # a step across x, two holes, sprinkled bad cells, NaN / inf
    trav[rows // 3 : rows // 3 + 2, :] = 0.2
    elev[: rows // 3, :] += 0.15
    trav[rows // 2 : rows // 2 + 5, cols // 4 : cols // 4 + 6] = 0.05
    badmask = rng.random((rows, cols)) < bad
    trav[badmask] = rng.uniform(0, 0.6, size=int(badmask.sum())).astype(np.float32)
    trav[rng.random((rows, cols)) < 0.02] = np.nan
    trav[rng.random((rows, cols)) < 0.003] = -np.inf
    elev[rng.random((rows, cols)) < 0.02] = np.nan
    elev[rng.random((rows, cols)) < 0.01] = 11.0
    return trav, elev


CASES = {
    # name: (rows, cols, res, map seed, params overrides, poses, n_cycles, pose extras)
    "trot_2cm": dict(rows=96, cols=64, res=0.02, seed=101, params={}, n=4,
                     poses=[[-0.55, 0.05, 0.0], [-0.50, -0.20, 0.1], [-0.62, 0.25, 0.0], [-0.45, 0.0, 0.0]]),
    "trot_1cm_r015": dict(rows=96, cols=96, res=0.01, seed=102, params={"searchRadius": 0.15, "stepLength": 0.08, "length": 0.2, "width": 0.1, "skew": 0.02},
                          n=3, poses=[[-0.15, 0.0, 0.0], [-0.12, 0.1, 0.0], [-0.2, -0.1, 0.0]]),
    "walk_hex_2cm": dict(rows=96, cols=64, res=0.02, seed=103, params={}, n=3, gait=1, poly=1,
                         poses=[[-0.55, 0.05, 0.0], [-0.48, -0.15, 0.0]]),
    "harsh_2cm": dict(rows=96, cols=64, res=0.02, seed=105, params={}, n=5, bad=0.4,
                      poses=[[-0.55, 0.05, 0.0], [-0.50, -0.20, 0.1], [-0.62, 0.25, 0.0], [-0.45, 0.0, 0.0], [-0.58, -0.1, 0.0], [-0.52, 0.15, 0.0]]),
    "code_defaults_3cm": dict(rows=80, cols=60, res=0.03, seed=104, params="code", n=3,
                              poses=[[-0.5, 0.0, 0.0], [-0.4, 0.2, 0.0]]),
}


def params_for(spec):
    p = yaml_params()
    if spec == "code":
        p["footRadius"], p["defaultFootholdThreshold"], p["stepLength"], p["skew"] = (np.float32(0.03), np.float32(0.7), np.float32(0.2), np.float32(0.1))
        return p
    for k, v in spec.items():
        p[k] = np.float32(v)
    return p


def main():
    for name, c in CASES.items():
        trav, elev = small_map(c["rows"], c["cols"], c["res"], c["seed"], c.get("bad", 0.04))
        p = params_for(c["params"])
        poses = oracle_poses(c["poses"], gait=c.get("gait", 0), leg_poly=c.get("poly", 0))
        m = fpo.OracleMap(trav, elev, c["res"])
        out = m.plan(p, poses, c["n"])
        np.savez_compressed(
            os.path.join(OUT, name + ".npz"), trav=trav, elev=elev, res=c["res"], params=p, poses=poses, n_cycles=c["n"],
            nominal=out["nominal"], centroid=out["centroid"], default=out["default"], cycle_ok=out["cycle_ok"], stance=out["stance"],
        )
        src = np.bincount(out["nominal"]["source"].ravel(), minlength=3).tolist()
        codes = np.bincount(out["centroid"]["code"].ravel(), minlength=7).tolist()
        print(name, "sources(default,candidate,none)", src, "centroid codes", codes, "cycle_ok", int(out["cycle_ok"].sum()), "/", out["cycle_ok"].size)


# Opt track on the same maps / poses / cycle flags: tests/golden/opt/<name>_<variant>.npz hold only the
# optimiser parameters and the opt products (the map and the plan come from the base file).
OPT_VARIANTS = {
    "yaml": {},                                           # constraints on: the reference's infeasible set -> least violation
    "code": {"useInequalityConstraits": 0},               # readParameters' defaults: unconstrained
    "weights": {"w1": 0.7, "w2": 1.3, "w3": 0.45, "w4": 2.1, "wr": 0.9, "wc": 1.15, "skewLowerScale": 0.0, "skewUpperScale": 40.0,
                "lfCurrentRow0": 5.0, "rhCurrentRow0": 27.0},
}


def main_opt():
    os.makedirs(os.path.join(OUT, "opt"), exist_ok=True)
    for name in ("trot_2cm", "trot_1cm_r015", "harsh_2cm", "code_defaults_3cm"):
        z = np.load(os.path.join(OUT, name + ".npz"))
        m = fpo.OracleMap(z["trav"], z["elev"], float(z["res"]))
        p, poses, n = z["params"].view(fpo.PARAMS_DTYPE), z["poses"].view(fpo.POSE_DTYPE), int(z["n_cycles"])
        for vname, ov in OPT_VARIANTS.items():
            if name == "trot_1cm_r015" and vname != "yaml":
                continue  # 31^4 lattice points per problem: one variant is enough for a CPU test
            op = fpo.opt_params_yaml()
            for k, v in ov.items():
                op[k] = v
            o = m.plan_opt(p, op, poses, n, z["cycle_ok"])
            np.savez_compressed(os.path.join(OUT, "opt", f"{name}_{vname}.npz"), base=name, opt_params=op, footholds=o["footholds"],
                                cycles=o["cycles"], gate_fail_cycle=o["gate_fail_cycle"])
            print("opt", name, vname, "status", np.bincount(o["cycles"]["solver_status"].ravel(), minlength=4).tolist(), "gate", o["gate_fail_cycle"].tolist(),
                  "codes", np.bincount(o["cycles"]["centroid_code"].ravel(), minlength=7).tolist())


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "opt":
        main_opt()  # (the base vectors stay as committed in round 1)
    else:
        main()
        main_opt()
