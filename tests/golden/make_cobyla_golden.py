#!/usr/bin/env python3
"""How far is the build-defined optimiser of the opt track (solveLattice, oracle/fpo_opt.cpp = csrc/fpe_opt.hpp) from a COBYLA run?

The reference's opt track calls NLopt's LN_COBYLA (FootholdPlanner.cpp:1116-1211; yaml: stepSize 1, xtol_rel 1e-4, ctol 1e-2, start at
centroidIndex) and truncates the result to integers (cpp:1287-1312).  NLopt is not in the image and is not pinned by the reference, so
its iterates cannot be reproduced; scipy 1.15 — present in the BUILD container only, never on the GPU box — ships Powell's COBYLA, the
algorithm NLopt's implementation derives from.  This script lets scipy's COBYLA drive the oracle's LITERAL chain (oracle.plan_opt_forced:
every cycle's problem is set up by the restated reference code from the feet the previous cycles' COBYLA answers produced), next to the
chain driven by solveLattice, and writes the statistics VERDICT r4 asked for to tests/golden/cobyla_vs_lattice.json:

  * how often the truncated x of the two optimisers is the same per cycle (rows x[0,2,4,6], columns x[1,3,5,7]), and how far apart;
  * the objective gap (lattice minus COBYLA, evaluated with the reference's objective on the truncated x) and the constraint violation;
  * how often the SERVICE GATE verdict (cycle in which getGaitCycleSearchGridMap fails, cpp:920-934) differs — what
    fpe_set_tuning("service_opt_gate", 2) turns into the handler's `return false`.

This is NOT NLopt's COBYLA (another implementation of the same method: different trust-region bookkeeping, different last digits) and it
pins nothing.  It tells the adapter's maintainer how much weight the build-defined gate verdict deserves.

    python3 tests/golden/make_cobyla_golden.py [n_poses_per_map]      (build container; ~2 min for the default)
"""
import json
import os
import sys

import numpy as np
from scipy.optimize import Bounds, minimize

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import fpo  # noqa: E402
from quadrupedal_foothold_planner_amd import _capi, synth  # noqa: E402  (host-side constants and generators only: no GPU, no engine call)
from tests import util  # noqa: E402


def problem_of(rec, op, length_base, skew, res):
    """The reference's objective and constraints (cpp:54-148) of one cycle record, as Python closures."""
    w1, w2, w3, w4, wr, wc = (float(op[k][0]) for k in ("w1", "w2", "w3", "w4", "wr", "wc"))
    nom, cen = rec["nominal_index"].astype(float), rec["centroid_index"].astype(float)
    lf, rh = float(rec["lf_current_row"]), float(rec["rh_current_row"])
    lb_r, sk2 = length_base / res, 2 * skew / res
    t1, t2 = length_base * float(op["hipLowerScale"][0]) / res, length_base * float(op["hipUpperScale"][0]) / res
    t3, t4 = 2 * skew * float(op["skewLowerScale"][0]) / res, 2 * skew * float(op["skewUpperScale"][0]) / res
    wv = np.array([wr, wc] * 4)

    def f(x):
        return (w1 * np.sum(wv * np.abs(x - nom)) + w2 * np.sum(wv * np.abs(x - cen))
                + w3 * (abs(abs(x[0] - x[2]) - lb_r) + abs(abs(x[4] - x[6]) - lb_r))
                + w4 * (abs(abs(0.5 * abs(x[0] - x[2]) - 0.5 * abs(x[4] - x[6])) - sk2)
                        + abs(abs(0.5 * abs(x[4] - x[6]) - 0.5 * abs(lf - rh)) - sk2)))

    def cons(x):
        a, b, c = abs(x[0] - x[2]), abs(x[4] - x[6]), abs(lf - rh)
        return np.array([t1 - a, a - t2, t1 - b, b - t2, t3 - 0.5 * abs(a - b), 0.5 * abs(a - b) - t4, t3 - 0.5 * abs(b - c), 0.5 * abs(b - c) - t4])

    return f, cons


def cobyla(rec, op, length_base, skew, res):
    f, cons = problem_of(rec, op, length_base, skew, res)
    lo, up = rec["x_lower"].astype(float), rec["x_upper"].astype(float)
    x0 = rec["centroid_index"].astype(float)
    if np.any(lo > up) or np.any(x0 < lo) or np.any(x0 > up):
        return x0, f, cons, "invalid-args"  # nlopt_optimize refuses (NLOPT_INVALID_ARGS): x stays x0 (cpp:1224-1226)
    use_cons = bool(op["useInequalityConstraits"][0])
    ctol = float(op["ctol"][0])
    c = [{"type": "ineq", "fun": (lambda x, k=k: -cons(x)[k])} for k in range(8)] if use_cons else []
    # NLopt: initial step 1 in every variable (yaml: manulStepSize, stepSize 1) = rhobeg 1; xtol_rel 1e-4 on indices of order 10^1-10^2
    # ~ a final trust-region radius of 1e-3; constraint tolerance ctol
    r = minimize(f, x0, method="COBYLA", bounds=Bounds(lo, up), constraints=c, options={"rhobeg": 1.0, "tol": 1e-3, "catol": ctol, "maxiter": 5000})
    return np.clip(r.x, lo, up), f, cons, "ok"


def main():
    n_poses = int(sys.argv[1]) if len(sys.argv) > 1 else 160
    params, op = util.to_oracle_params(_capi.params_yaml()), fpo.opt_params_yaml()
    length_base, skew = float(np.float32(params["length"][0])), float(params["skew"][0])
    n_cycles = 8
    stats = {"cycles": 0, "rows_equal": 0, "cols_equal": 0, "all_equal": 0, "row_l1": [], "col_l1": [], "obj_gap": [], "cobyla_violation": [],
             "lattice_violation": [], "poses": 0, "gate_equal": 0, "gate_lattice_only": 0, "gate_cobyla_only": 0, "gate_both_other_cycle": 0,
             "invalid_args": 0}
    fixture = []
    for res, side, seed, bad in ((0.02, 8.0, 1, 0.02), (0.02, 8.0, 11, 0.25), (0.01, 6.0, 2, 0.02), (0.02, 6.0, 61, 0.45)):
        rows = int(round(side / res))
        trav, elev = synth.rough_map(rows, rows, res, seed, bad_frac=bad)
        om = fpo.OracleMap(trav, elev, res)
        poses = synth.poses_in_map(n_poses, side, side, n_cycles, 0.18, seed=seed + 100, margin=0.7)
        op_o = util.to_oracle_poses(poses)
        plan = om.plan(params, op_o, n_cycles, threads=4)
        lat = om.plan_opt(params, op, op_o, n_cycles, plan["cycle_ok"])
        for b in range(n_poses):
            forced = []
            gate_c = 255
            for g in range(n_cycles):
                cyc, gate = om.plan_opt_forced(params, op, op_o[b], n_cycles, plan["cycle_ok"][b], np.array(forced).reshape(-1, 8))
                if gate != 255 and gate <= g:
                    gate_c = gate
                    break
                x, f, cons, st = cobyla(cyc[g], op, length_base, skew, res)
                stats["invalid_args"] += st != "ok"
                forced.append(x)
            else:
                cyc, gate = om.plan_opt_forced(params, op, op_o[b], n_cycles, plan["cycle_ok"][b], np.array(forced).reshape(-1, 8))
                gate_c = gate
            gate_l = int(lat["gate_fail_cycle"][b])
            stats["poses"] += 1
            stats["gate_equal"] += gate_l == gate_c
            stats["gate_lattice_only"] += gate_l != 255 and gate_c == 255
            stats["gate_cobyla_only"] += gate_l == 255 and gate_c != 255
            stats["gate_both_other_cycle"] += gate_l != 255 and gate_c != 255 and gate_l != gate_c
            # per-cycle comparison on the FIRST cycle the two chains share as a problem (cycle 0: same feet, same problem) and, where
            # the chains stayed together, on the later ones
            together = True
            for g in range(min(len(forced), n_cycles)):
                rl = lat["cycles"][b, g]
                if lat["gate_fail_cycle"][b] != 255 and g >= lat["gate_fail_cycle"][b]:
                    break
                same_problem = together and all(np.array_equal(rl[k], cyc[g][k]) for k in ("nominal_index", "centroid_index", "x_lower", "x_upper")) \
                    and rl["lf_current_row"] == cyc[g]["lf_current_row"] and rl["rh_current_row"] == cyc[g]["rh_current_row"]
                if not same_problem:
                    together = False
                    continue
                f, cons = problem_of(rl, op, length_base, skew, res)
                xl, xc = rl["x"].astype(float), np.trunc(forced[g])
                stats["cycles"] += 1
                re_, ce_ = np.array_equal(xl[0::2], xc[0::2]), np.array_equal(xl[1::2], xc[1::2])
                stats["rows_equal"] += re_
                stats["cols_equal"] += ce_
                stats["all_equal"] += re_ and ce_
                stats["row_l1"].append(float(np.abs(xl[0::2] - xc[0::2]).sum()))
                stats["col_l1"].append(float(np.abs(xl[1::2] - xc[1::2]).sum()))
                stats["obj_gap"].append(float(f(xl) - f(xc)))
                stats["cobyla_violation"].append(float(max(cons(xc).max(), 0.0)))
                stats["lattice_violation"].append(float(max(cons(xl).max(), 0.0)))
                if len(fixture) < 64:
                    fixture.append({"map": [res, side, seed, bad], "pose": [float(v) for v in poses["position"][b]], "cycle": g,
                                    "x_lattice": [int(v) for v in xl], "x_cobyla_truncated": [int(v) for v in xc]})
                together = re_ and ce_ and together
    q = lambda v: [float(np.quantile(v, p)) for p in (0.0, 0.1, 0.5, 0.9, 1.0)] if len(v) else None  # noqa: E731
    out = {
        "what": "solveLattice (build-defined optimiser of the opt track) against scipy COBYLA driving the oracle's literal chain; see make_cobyla_golden.py",
        "scipy": __import__("scipy").__version__, "cobyla_options": {"rhobeg": 1.0, "tol": 1e-3, "catol": float(op["ctol"][0]), "x0": "centroidIndex"},
        "poses": stats["poses"], "cycles_compared_on_identical_problems": stats["cycles"],
        "share_rows_equal": stats["rows_equal"] / max(stats["cycles"], 1), "share_cols_equal": stats["cols_equal"] / max(stats["cycles"], 1),
        "share_all_eight_equal": stats["all_equal"] / max(stats["cycles"], 1),
        "row_l1_distance_quantiles_0_10_50_90_100": q(stats["row_l1"]), "col_l1_distance_quantiles": q(stats["col_l1"]),
        "objective_lattice_minus_cobyla_quantiles": q(stats["obj_gap"]),
        "constraint_violation_cobyla_quantiles": q(stats["cobyla_violation"]), "constraint_violation_lattice_quantiles": q(stats["lattice_violation"]),
        "gate_verdict": {"equal_share": stats["gate_equal"] / max(stats["poses"], 1), "lattice_refuses_cobyla_does_not": stats["gate_lattice_only"],
                         "cobyla_refuses_lattice_does_not": stats["gate_cobyla_only"], "both_refuse_other_cycle": stats["gate_both_other_cycle"]},
        "nlopt_invalid_args_problems": stats["invalid_args"],
        "examples": fixture,
    }
    path = os.path.join(ROOT, "tests", "golden", "cobyla_vs_lattice.json")
    json.dump(out, open(path, "w"), indent=1)
    print(json.dumps({k: v for k, v in out.items() if k != "examples"}, indent=1))


if __name__ == "__main__":
    main()
