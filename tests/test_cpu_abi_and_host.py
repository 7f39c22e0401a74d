"""CPU-only checks of the product: the C-ABI library builds/loads and exports every symbol of
include/fpe.h, and the host logic (spiral rank table, tile sizing, params, message assembly)
agrees with the oracle.  No compute call is made without a GPU."""
import ctypes as C
import re
import os

import numpy as np
import pytest

from oracle import fpo
from quadrupedal_foothold_planner_amd import _capi, synth
from tests.conftest import yaml_params

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_loads_and_exports_every_declared_symbol():
    L = _capi.lib()
    hdr = open(os.path.join(ROOT, "include", "fpe.h")).read()
    declared = set(re.findall(r"\b(fpe_[a-z_0-9]+)\s*\(", hdr))
    assert declared == set(_capi.EXPORTED_SYMBOLS), declared ^ set(_capi.EXPORTED_SYMBOLS)
    for name in declared:
        assert hasattr(L, name), name
    assert b"gfx950" in L.fpe_version()


def test_record_layouts_match_between_engine_and_oracle():
    assert _capi.PARAMS_DTYPE == fpo.PARAMS_DTYPE
    assert _capi.POSE_DTYPE.itemsize == fpo.POSE_DTYPE.itemsize == 64
    assert _capi.FOOTHOLD_DTYPE.itemsize == 32 and _capi.CENTROID_DTYPE.itemsize == 32
    assert _capi.QUERY_DTYPE.itemsize == fpo.QUERY_DTYPE.itemsize
    assert _capi.MSG_FOOTHOLD_DTYPE.itemsize == 32
    assert _capi.GLOBAL_FOOTHOLDS_DTYPE.itemsize == 8 + 32 * (4 + 4 * 255)


def test_multi_device_shard_offsets_and_exchange_records():
    """fpe_multi_shard_range (the block of device k in fpe_multi_plan / fpe_multi_plan_device) equals dist.shard_range, the
    blocks tile the batch in order, and the two exchange records have the sizes the gather's offset arithmetic uses."""
    from quadrupedal_foothold_planner_amd import dist

    L = _capi.lib()
    for B in (1, 7, 8, 9, 4096, 4099, 262144):
        for n in (1, 2, 3, 8):
            nxt = 0
            for k in range(n):
                first, count = C.c_int32(-1), C.c_int32(-1)
                assert L.fpe_multi_shard_range(B, k, n, C.byref(first), C.byref(count)) == 0
                lo, hi = dist.shard_range(B, k, n)
                assert (first.value, first.value + count.value) == (lo, hi) and first.value == nxt
                nxt = hi
            assert nxt == B
    assert L.fpe_multi_shard_range(8, 3, 3, C.byref(C.c_int32()), C.byref(C.c_int32())) == _capi.FPE_E_INVALID_ARG
    assert _capi.SELECTED_DTYPE.itemsize == 16 and _capi.PACKED_DTYPE.itemsize == 8
    assert C.sizeof(_capi.PlanOut) == 8 * 8 and C.sizeof(_capi.MultiDeviceIO) == 8 + 64 + 8 + 8 and C.sizeof(_capi.ServiceGate) == 24
    # the packed word: (row + 256) | (col + 256) << 14 | valid << 28 | source << 29 (-1 and the few cells a default hit's index
    # can lie outside the map are ordinary values)
    rec = np.zeros((2, 3, 4), dtype=_capi.PACKED_DTYPE)
    rec["cell"][0, 1, 2] = (123 + 256) | ((15870 + 256) << 14) | (1 << 28) | (1 << 29)
    rec["cell"][1, 2, 3] = 255 | (253 << 14) | (2 << 29)
    rec["z"][0, 1, 2] = np.float32(0.25)
    un = _capi.unpack_selected(rec)
    assert (un["row"][0, 1, 2], un["col"][0, 1, 2], un["valid"][0, 1, 2], un["source"][0, 1, 2]) == (123, 15870, 1, 1)
    assert (un["row"][1, 2, 3], un["col"][1, 2, 3], un["valid"][1, 2, 3], un["source"][1, 2, 3]) == (-1, -3, 0, 2)
    assert _capi.PACKED_MAX_CELLS == 15871
    assert un["foot_id"][1, 2].tolist() == [0, 1, 2, 3] and un["gait_cycle_id"][0, :, 0].tolist() == [0, 1, 2]
    assert un["z"][0, 1, 2] == np.float32(0.25)


def test_oracle_lateral_gate_is_the_y_side_of_the_gait_cycle_submap():
    """fpo_gate_lateral: the cycle in which getGaitCycleSearchGridMap's centre y = y0 + g * drift leaves the map (hand values:
    6 x 6 m map, y0 = -3 + 0.010, drift -0.007: cycles 0 and 1 stay at -2.990 / -2.997, cycle 2 is at -3.004: off the map)."""
    m = fpo.OracleMap(np.ones((300, 300), np.float32), np.zeros((300, 300), np.float32), 0.02)
    p = yaml_params()
    from tests.conftest import oracle_poses
    poses = oracle_poses([[0.0, -2.990, 0.0], [1.0, 0.0, 0.0], [0.0, 2.99, 0.0], [0.0, -3.2, 0.0], [-2.95, -2.996, 0.0]])
    assert m.gate_lateral(p, poses, 8).tolist() == [2, 255, 255, 0, 1]
    assert m.gate_lateral(p, poses, 2).tolist() == [255, 255, 255, 0, 1]


def test_params_defaults_match_yaml_and_code():
    y = _capi.params_yaml()
    assert y.tobytes() == yaml_params().tobytes()
    c = _capi.params_code_defaults()[0]
    assert c["footRadius"] == np.float32(0.03) and c["stepLength"] == np.float32(0.2) and c["skew"] == np.float32(0.1)
    assert c["defaultFootholdThreshold"] == np.float32(0.7)


def test_filter_defaults_match_between_engine_and_oracle():
    """fpe_filter_params_defaults (no device needed) and the oracle's defaults are the same published chain, field by
    field, and the two records have the same layout."""
    import ctypes as C

    fp = _capi.FilterParams()
    assert _capi.lib().fpe_filter_params_defaults(C.byref(fp)) == 0
    o = fpo.filter_defaults()[0]
    assert C.sizeof(fp) == fpo.FILTER_PARAMS_DTYPE.itemsize
    pairs = [("normal_radius", "normalRadius"), ("slope_critical", "slopeCritical"), ("step_critical", "stepCritical"),
             ("step_first_radius", "stepFirstRadius"), ("step_second_radius", "stepSecondRadius"),
             ("step_critical_cells", "stepCriticalCells"), ("roughness_critical", "roughnessCritical"), ("roughness_radius", "roughnessRadius")]
    for a, b in pairs:
        assert getattr(fp, a) == o[b], (a, b)
    assert bytes(fp) == o.tobytes()


def test_no_gpu_means_loud_failure_not_fallback():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from quadrupedal_foothold_planner_amd.planner import FootholdPlanner

    with pytest.raises(_capi.EngineUnavailable):
        FootholdPlanner(0)


@pytest.mark.parametrize("res,R", [(0.02, 0.1), (0.01, 0.15), (0.005, 0.1), (0.03, 0.2)])
def test_spiral_rank_table_equals_oracle_iterator_order(res, R):
    """The engine's host-built rank table (prefix of rings 0..nRings-2) must be the oracle's
    SpiralIterator order on a map large enough that nothing is clipped."""
    rows = 600
    m = fpo.OracleMap(np.ones((rows, rows), np.float32), np.zeros((rows, rows), np.float32), res)
    ok, cx, cy = m.get_position(300, 300)
    Rd = float(np.float32(R))
    n_rings = int(np.ceil(Rd / res))
    cells = m.spiral_cells(cx, cy, Rd) - np.array([300, 300])
    table = _capi.spiral_offsets(n_rings)
    # every offset with trunc(norm) == d appears exactly once in ring d
    for d in range(n_rings + 1):
        ring = table[table[:, 2] == d][:, :2]
        want = {(a, b) for a in range(-d - 1, d + 2) for b in range(-d - 1, d + 2) if int(np.sqrt(a * a + b * b)) == d}
        assert {tuple(x) for x in ring.tolist()} == want and len(ring) == len(want)
    # unfiltered rings (<= nRings-2) are a prefix of the oracle's order
    n_unfiltered = int((table[:, 2] <= n_rings - 2).sum())
    assert np.array_equal(table[:n_unfiltered, :2], cells[:n_unfiltered])
    # filtered rings keep the table's relative order
    rank = {tuple(t[:2]): k for k, t in enumerate(table.tolist())}
    ks = [rank[tuple(c)] for c in cells.tolist()]
    assert ks == sorted(ks)


def test_tile_halfwidth_covers_search_footprint():
    L = _capi.lib()
    for res, R, rf in [(0.02, 0.1, 0.02), (0.01, 0.15, 0.02), (0.005, 0.1, 0.02), (0.02, 0.1, 0.03)]:
        H = L.fpe_tile_halfwidth(np.float32(R), np.float32(rf), res)
        n_rings = int(np.ceil(float(np.float32(R)) / res))
        assert H >= n_rings + int(np.ceil(float(np.float32(rf)) / res)) + 1


def test_algorithmic_bytes_per_foothold_matches_survey_table():
    # 508 / 2204 / 4444 / 9212 B
    f = _capi.algorithmic_bytes_per_foothold
    assert f(0.1, 0.02, 0.02) == 508
    assert f(0.1, 0.02, 0.01) == 2204
    assert f(0.15, 0.02, 0.01) == 4444
    assert f(0.1, 0.02, 0.005) == 9212


def test_service_message_for_zero_cycles_needs_no_gpu():
    """gait_cycles = 0: the reference's loop never runs; the response holds the 4 stance entries."""
    L = _capi.lib()
    msg = np.zeros(1, dtype=_capi.GLOBAL_FOOTHOLDS_DTYPE)
    p = _capi.params_yaml()
    pos = np.array([-0.21, -1.87, 0.3])
    # a handle is required by the ABI; without a GPU fpe_create fails, so only run the host part when it exists
    h = C.c_void_p()
    rc = L.fpe_create(0, C.byref(h))
    if rc != _capi.FPE_OK:
        assert rc == _capi.FPE_E_NO_DEVICE
        assert L.fpe_plan_service(None, _capi.ptr(p), _capi.ptr(pos), 0, _capi.ptr(msg)) == _capi.FPE_E_INVALID_ARG
        return
    assert L.fpe_plan_service(h, _capi.ptr(p), _capi.ptr(pos), 0, _capi.ptr(msg)) == _capi.FPE_OK
    m = msg[0]
    assert m["n_footholds"] == 4 and m["success"] == 0 and m["gait_cycles_succeed"] == 0
    assert m["footholds"]["y"][0] == -0.12449999898672104 + -1.87
    L.fpe_destroy(h)


def test_synthetic_generator_is_deterministic():
    a = synth.rough_map(64, 48, 0.02, seed=5)
    b = synth.rough_map(64, 48, 0.02, seed=5)
    assert a[0].tobytes() == b[0].tobytes() and a[1].tobytes() == b[1].tobytes()
    assert np.isnan(a[0]).sum() > 0 and (a[0] < 0.7).sum() > 0
    p = synth.poses_in_map(32, 20.0, 20.0, 8, 0.18, seed=6)
    assert p["position"][:, 0].max() <= 10 - 0.6 - 8 * 0.18


def test_ros_adapter_parses_against_the_mock_ros_types():
    """csrc/ros_adapter/fpe_ros_adapter.hpp cannot be linked here (no ROS1 / grid_map in the image), but it must at
    least be valid C++ against the types it touches: -fsyntax-only with tests/probe/ros_mock (a mock of those few
    types, see its README) catches renamed C-ABI fields, missing includes and signature drift."""
    import subprocess

    hdr = os.path.join(ROOT, "quadrupedal_foothold_planner_amd", "csrc", "ros_adapter", "fpe_ros_adapter.hpp")
    cmd = ["g++", "-std=c++14", "-fsyntax-only", "-Wall", "-Wno-pragma-once-outside-header", "-DFPE_WITH_ROS",
           "-I" + os.path.join(ROOT, "tests", "probe", "ros_mock"), "-I" + os.path.join(ROOT, "include"), "-x", "c++", hdr]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    src = open(hdr).read()
    assert "resp_" not in src.replace("repNominal_", "").replace("repCentroid_", ""), "no shared response members (AsyncSpinner)"


def test_upstream_check_parses_against_the_mock_and_its_harness_runs(tmp_path):
    """tests/probe/upstream_check.cpp is the one-command pin for whoever has grid_map: the same seeded inputs
    through the real grid_map_core and through oracle/fpo_gridmap.hpp, exit 1 at the first difference.  Here its upstream half can
    only be PARSED (-DFPE_WITH_GRID_MAP against tests/probe/ros_mock's declarations), and its harness run with the restatement on
    both sides (every section must report identical cases: the comparison code itself is exercised)."""
    import subprocess

    src = os.path.join(ROOT, "tests", "probe", "upstream_check.cpp")
    inc = ["-I" + os.path.join(ROOT, "oracle")]
    r = subprocess.run(["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Wextra", "-DFPE_WITH_GRID_MAP", "-I" + os.path.join(ROOT, "tests", "probe", "ros_mock")]
                       + inc + [src], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    exe = str(tmp_path / "upstream_check_self")
    r = subprocess.run(["g++", "-std=c++17", "-O1", "-ffp-contract=off"] + inc + [src, "-o", exe], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe, "3000"], capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.count("cases identical") == 5 and "harness self-test only" in r.stdout, r.stdout


def test_the_header_is_plain_c(tmp_path):
    """include/fpe.h is the drop-in boundary: a C ABI.  It must compile as C99 with warnings on (a C or cgo host binds it as
    it is), and a C program that only uses the parameter helpers must link against the library without a C++ runtime of its own."""
    import subprocess

    from quadrupedal_foothold_planner_amd import build as fbuild

    src = tmp_path / "hdr.c"
    src.write_text('#include "fpe.h"\n#include <stdio.h>\nint main(void) { fpe_params p; fpe_opt_params o; fpe_filter_params f;\n'
                   '  if (fpe_params_yaml(&p) || fpe_opt_params_yaml(&o) || fpe_filter_params_defaults(&f)) return 1;\n'
                   '  printf("%s %.3f %.3f\\n", fpe_version(), (double)p.searchRadius, f.normal_radius); return 0; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-fsyntax-only", "-I" + os.path.join(ROOT, "include"), str(src)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib = fbuild.build_engine()
    exe = tmp_path / "hdr"
    r = subprocess.run(["gcc", "-std=c99", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe), "-L" + os.path.dirname(lib),
                        "-l:" + os.path.basename(lib), "-Wl,-rpath," + os.path.dirname(lib), "-Wl,--allow-shlib-undefined"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True)  # (no GPU needed: the helpers only fill structs)
    assert r.returncode == 0 and r.stdout.startswith("fpe ") and "0.100 0.050" in r.stdout, (r.returncode, r.stdout, r.stderr)


def test_build_is_atomic_and_reports_what_it_did():
    from quadrupedal_foothold_planner_amd import build as fbuild

    path = fbuild.build_engine()
    assert os.path.exists(path) and fbuild.LAST_ACTION in ("compiled", "up-to-date")
    assert not [f for f in os.listdir(os.path.dirname(path)) if f.endswith(".so.tmp")], "a temporary build output was left behind"


def test_three_operation_division_by_three_is_exact(tmp_path):
    """getPolygonCenter's x / 3.0 (cpp:2461-2462) is evaluated by the kernels as q = x * c, r = fma(-3, q, x),
    q + r * c with c = RN(1/3) (fpe_kernels.hip::div3; proof in its comment).  The same sequence on the CPU must equal the
    division bit for bit: 20 M random doubles plus mantissa patterns next to the quotient's rounding boundaries."""
    import subprocess

    src = os.path.join(os.path.dirname(__file__), "probe", "div3_check.c")
    exe = str(tmp_path / "div3_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe], check=True, capture_output=True, text=True).stdout.strip()
    assert out == "0", f"{out} mismatches between the three-operation form and x / 3.0"


def test_collective_stand_in_builds_and_exports_what_the_engine_binds():
    """tests/probe/collective_shim.cpp (the device-local stand-in that lets one GPU act as n ranks in the GPU suite) must export
    every entry point csrc/fpe_multi.cpp resolves from its collective library, plus the marker that identifies it."""
    import re
    import subprocess

    from tests.test_gpu_device_api import build_collective_shim

    so = build_collective_shim()
    src = open(os.path.join(ROOT, "quadrupedal_foothold_planner_amd", "csrc", "fpe_multi.cpp")).read()
    wanted = set(re.findall(r'sym\("(nccl\w+)"\)', src))
    assert len(wanted) >= 7
    exported = subprocess.run(["nm", "-D", "--defined-only", so], capture_output=True, text=True).stdout
    for name in sorted(wanted) + ["fpe_test_collective_shim"]:
        assert re.search(r"\b" + name + r"\b", exported), name


def test_a_library_that_predates_the_header_is_unavailable_not_an_attribute_error(tmp_path):
    """FPE_LIB pointing at a library built before a symbol of include/fpe.h existed must surface as EngineUnavailable
    (what callers and tests catch), not as ctypes' AttributeError."""
    import subprocess
    import sys

    src = tmp_path / "old.c"
    src.write_text('const char* fpe_version(void) { return "0"; }\nint fpe_create(int d, void** h) { (void)d; (void)h; return -1; }\n')
    so = tmp_path / "libfpe_old.so"
    assert subprocess.run(["gcc", "-shared", "-fPIC", str(src), "-o", str(so)]).returncode == 0
    code = ("from quadrupedal_foothold_planner_amd import _capi\n"
            "try:\n    _capi.lib()\n    print('loaded')\n"
            "except _capi.EngineUnavailable as e:\n    print('unavailable', 'fpe_abi_version' in str(e) or 'does not export' in str(e))\n")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, cwd=ROOT, env=dict(os.environ, FPE_LIB=str(so)))
    assert r.returncode == 0 and r.stdout.strip() == "unavailable True", r.stdout + r.stderr
