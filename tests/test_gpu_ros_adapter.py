"""The ROS adapter RUN, not only parsed ("adapter header syntax-checked against a mock only"):
csrc/ros_adapter/fpe_ros_adapter.hpp compiled with g++ against the mock ROS / grid_map types of tests/probe/ros_mock — which
carry storage — linked with the real libfpe.so and run on the GPU by tests/probe/adapter_run.cpp: gridmapCallback's upload
from a column-major grid_map with a circular-buffer start index, then for a list of initial poses every call the adapter
offers (plan, planAllTracks, planWithOptTrack) with the members a node keeps between calls (the never-cleared centroid and
opt messages, lfCurrentRow / rhCurrentRow).  Every number the C++ side hands to the ROS messages must equal what the same
service calls return through the Python binding, bit for bit."""
import os
import subprocess

import numpy as np
import pytest

from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd import build as fbuild
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def build_driver():
    lib = fbuild.build_engine()
    out_dir = os.path.join(ROOT, "tests", "probe", "_build")
    os.makedirs(out_dir, exist_ok=True)
    exe = os.path.join(out_dir, "adapter_run")
    cmd = ["g++", "-std=c++14", "-O1", "-Wall", "-DFPE_WITH_ROS", "-I" + os.path.join(ROOT, "tests", "probe", "ros_mock"),
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "quadrupedal_foothold_planner_amd", "csrc", "ros_adapter"),
           os.path.join(ROOT, "tests", "probe", "adapter_run.cpp"), "-o", exe, "-L" + os.path.dirname(lib), "-l:" + os.path.basename(lib),
           "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return exe


def parse(path):
    """-> list of per-pose dicts; numbers from C99 hex floats."""
    hx = float.fromhex
    poses, cur = [], None
    lines = open(path).read().split("\n")
    k = 0

    def msg(first):
        nonlocal k
        _, success, gc, gcs, n = first.split()
        fh = []
        for _ in range(int(n)):
            k += 1
            x, y, z, foot, cyc = lines[k].split()
            fh.append((hx(x), hx(y), hx(z), int(foot), int(cyc)))
        return {"success": bool(int(success)), "gait_cycles": int(gc), "gait_cycles_succeed": int(gcs), "footholds": fh}

    section = None
    while k < len(lines):
        ln = lines[k]
        t = ln.split()
        if not t:
            k += 1
            continue
        if t[0] == "pose":
            cur = {}
            poses.append(cur)
        elif t[0] in ("plan", "all", "opt"):
            section = t[0]
            cur[section] = {"ok": bool(int(t[1]))}
        elif t[0] in ("msg", "centroid", "optmsg"):
            cur[section][t[0]] = msg(ln)
        elif t[0] in ("rows", "pathN", "pathC", "csN", "fdN", "csC", "fdC", "csO", "fdO"):
            cur[section][t[0]] = np.array([hx(v) for v in t[2:]], dtype=np.float64)
            cur[section][t[0] + "_n"] = int(t[1])
        elif t[0] == "lfrh":
            cur["lfrh"] = (hx(t[1]), hx(t[2]))
        else:
            raise AssertionError("adapter_run: " + ln)
        k += 1
    return poses


def service_report(p, n_cycles, pos):
    """fpe_plan_service_report as the adapter's planAllTracks calls it (the Python binding's all_tracks form goes through
    fpe_plan_service_opt, whose centroid path is interleaved with the opt track's feet centres, cpp:946)."""
    import ctypes as C
    from quadrupedal_foothold_planner_amd._capi import GLOBAL_FOOTHOLDS_DTYPE, TRACK_REPORT_DTYPE, ptr
    msg, cen = np.zeros(1, GLOBAL_FOOTHOLDS_DTYPE), np.zeros(1, GLOBAL_FOOTHOLDS_DTYPE)
    dflt = np.zeros((1 + n_cycles, 4, 3), np.float64)
    nrows = C.c_int32(0)
    rep = np.zeros(2, TRACK_REPORT_DTYPE)
    pos = np.ascontiguousarray(pos, np.float64)
    rc = p._lib.fpe_plan_service_report(p._h, ptr(p.params), ptr(pos), n_cycles, ptr(msg), ptr(cen), ptr(dflt), C.cast(C.byref(nrows), C.c_void_p),
                                        ptr(rep[0:1]), ptr(rep[1:2]))
    if rc == _capi.FPE_E_SERVICE_FALSE:
        return False
    assert rc == _capi.FPE_OK, rc
    out = p._msg(msg[0])
    out["centroid"] = p._msg(cen[0])
    out["default_footholds"] = dflt[: nrows.value].copy()
    out["report"], out["centroid"]["report"] = p._report(rep[0]), p._report(rep[1])
    return out


def as_tuples(m):
    return [(float(f["x"]), float(f["y"]), float(f["z"]), int(f["foot_id"]), int(f["gait_cycle_id"])) for f in m["footholds"]]


def test_the_adapter_runs_and_hands_over_what_the_service_returns(tmp_path):
    exe = build_driver()
    rows, cols, res, n_cycles = 300, 280, 0.02, 6
    trav, elev = synth.rough_map(rows, cols, res, seed=77, bad_frac=0.1)
    position, (si, sj) = (1.5, -0.7), (37, 81)
    # initial poses: inside, near the -y edge (the lateral gate refuses there), far outside (cycle-0 gate)
    init = np.array([[-1.0, 0.2, 0.0], [0.3, -0.4, 0.1], [-0.5, -3.35, 0.0], [40.0, 0.0, 0.0], [0.9, 1.1, 0.0], [-1.6, -1.0, 0.0]], dtype=np.float64)
    init[:, 0] += position[0]
    init[:, 1] += position[1]
    buf = lambda layer: np.ascontiguousarray(np.roll(np.roll(layer, si, axis=0), sj, axis=1).T).astype(np.float32).tobytes()  # column-major + start index
    inp = tmp_path / "in.bin"
    with open(inp, "wb") as f:
        f.write(np.array([rows, cols, si, sj, n_cycles, len(init)], dtype=np.int32).tobytes())
        f.write(np.array([res, position[0], position[1]], dtype=np.float64).tobytes())
        f.write(buf(trav))
        f.write(buf(elev))
        f.write(init.tobytes())
    outp = tmp_path / "out.txt"
    r = subprocess.run([exe, str(inp), str(outp)], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, (r.returncode, r.stderr[-2000:], open(outp).read()[-500:] if os.path.exists(outp) else "")
    got = parse(outp)
    assert len(got) == len(init)

    p = FootholdPlanner(0)
    try:
        p.gridmapCallback(trav, elev, res, position)
        cen_acc, opt_acc, lfrh = [], [], [0.0, 0.0]
        seen_false = seen_true = 0
        for k, pos in enumerate(init):
            g = got[k]
            p.params = _capi.params_yaml()
            p.opt_params = _capi.opt_params_yaml()
            ref = p.globalFootholdPlan(n_cycles, pos)
            assert g["plan"]["ok"] == (ref is not False), k
            if ref is not False:
                m = g["plan"]["msg"]
                assert (m["success"], m["gait_cycles"], m["gait_cycles_succeed"]) == (ref["success"], ref["gait_cycles"], ref["gait_cycles_succeed"])
                assert m["footholds"] == as_tuples(ref), f"pose {k}: nominal message differs"
            # every track: planAllTracks = fpe_plan_service_report, planWithOptTrack = fpe_plan_service_opt (the binding's all_tracks form)
            p.opt_params["lf_current_row0"], p.opt_params["rh_current_row0"] = lfrh
            full = p.globalFootholdPlan(n_cycles, pos, all_tracks=True)
            gate = p.last_service_gate()
            assert g["all"]["ok"] == (full is not False) and g["opt"]["ok"] == (full is not False), k
            if full is False:
                seen_false += 1
                assert g["lfrh"] == tuple(lfrh)
                continue
            seen_true += 1
            a = g["all"]
            two = service_report(p, n_cycles, pos)
            assert two is not False and a["msg"]["footholds"] == as_tuples(two)
            cen_acc += as_tuples(two["centroid"])
            c = a["centroid"]
            assert c["footholds"] == cen_acc, f"pose {k}: the centroid message is appended to (cpp:715)"
            assert c["gait_cycles"] == 77 and c["success"] == two["centroid"]["success"] and c["gait_cycles_succeed"] == two["centroid"]["gait_cycles_succeed"]
            assert a["rows_n"] == two["default_footholds"].shape[0] and np.array_equal(a["rows"], two["default_footholds"].reshape(-1))
            for tag, rep in (("N", two["report"]), ("C", two["centroid"]["report"])):
                assert np.array_equal(a["path" + tag], rep["path"].reshape(-1)) and a["path" + tag + "_n"] == rep["path"].shape[0]
                assert np.array_equal(a["cs" + tag], rep["cog_speed"]) and np.array_equal(a["fd" + tag], rep["feet_distance"])
            o = g["opt"]
            assert o["msg"]["footholds"] == as_tuples(full)
            opt_acc += as_tuples(full["opt"])
            assert o["optmsg"]["footholds"] == opt_acc and o["optmsg"]["gait_cycles"] == 78
            assert np.array_equal(o["csO"], full["opt"]["report"]["cog_speed"]) and np.array_equal(o["fdO"], full["opt"]["report"]["feet_distance"])
            if gate["chain_ran"]:
                lfrh = [gate["lf_current_row"], gate["rh_current_row"]]
            assert g["lfrh"] == tuple(lfrh), f"pose {k}: lfCurrentRow / rhCurrentRow carried to the next call"
        assert seen_true >= 3 and seen_false >= 2, (seen_true, seen_false)
        assert any(v != 0.0 for v in lfrh)
    finally:
        p.close()
