"""CPU tests of the engine's geometry header (csrc/fpe_gridmath.hpp, shared by host and device
code), compiled for the host through tests/probe/gridmath_probe.cpp:
* the division-free index prediction equals the literal getIndexFromPosition everywhere, including
  exact ties; the fast PNPOLY equals the literal one;
* the closed-form per-axis geometry equals the oracle's iterator objects (independent restatements
  of the same assumed grid_map semantics)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest
from hypothesis import given, settings
from hypothesis import strategies as st

from oracle import fpo

HERE = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(HERE, "probe", "gridmath_probe.cpp")
OUT = os.path.join(HERE, "probe", "_build", "libgridmath_probe.so")


@pytest.fixture(scope="module")
def probe():
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    hdr = os.path.join(HERE, "..", "quadrupedal_foothold_planner_amd", "csrc", "fpe_gridmath.hpp")
    if not os.path.exists(OUT) or max(os.path.getmtime(SRC), os.path.getmtime(hdr)) > os.path.getmtime(OUT):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", OUT, SRC])
    L = C.CDLL(OUT)
    d = C.c_double
    L.probe_index_of.argtypes = [d, d, d, d]
    L.probe_index_of_fast.argtypes = [d, d, d, d]
    L.probe_polygon.argtypes = [C.c_void_p, C.c_void_p, C.c_int, d, d, C.c_int]
    L.probe_bbox.argtypes = [C.c_int, C.c_int, d, d, d, d, d, d, C.c_void_p, C.c_void_p]
    L.probe_submap.argtypes = [C.c_int, C.c_int, d, d, d, d, d, d, d, C.c_void_p, C.c_void_p]
    L.probe_cell_pos.restype = d
    L.probe_cell_pos.argtypes = [C.c_int, C.c_int, d, d, d, C.c_int, C.c_int]
    return L


RES = st.sampled_from([0.02, 0.01, 0.005, 0.03, 0.25, 0.1, 1.0 / 3.0])


@settings(max_examples=400, deadline=None)
@given(res=RES, n=st.integers(10, 4000), p=st.floats(-50, 50), frac=st.floats(-0.2, 1.2))
def test_index_prediction_equals_literal_random(probe, res, n, p, frac):
    L_ = n * res
    org = 0.5 * L_
    x = p + org - frac * L_
    assert probe.probe_index_of_fast(x, org, p, res) == probe.probe_index_of(x, org, p, res)


@settings(max_examples=400, deadline=None)
@given(res=RES, n=st.integers(10, 2000), p=st.sampled_from([0.0, 1.5, -3.25, 7.0]), k=st.integers(0, 2000),
       half=st.sampled_from([0.0, 0.5, 1.0, 1.5, 2.0]), ulps=st.integers(-3, 3))
def test_index_prediction_equals_literal_on_ties(probe, res, n, p, k, half, ulps):
    """positions on cell boundaries / centres +- a few ulps: where the prediction must defer"""
    L_ = n * res
    org = 0.5 * L_
    x = p + org - (min(k, n) + half) * res
    x = float(np.nextafter(x, np.inf if ulps > 0 else -np.inf)) if ulps else x
    for _ in range(abs(ulps) - 1 if ulps else 0):
        x = float(np.nextafter(x, np.inf if ulps > 0 else -np.inf))
    assert probe.probe_index_of_fast(x, org, p, res) == probe.probe_index_of(x, org, p, res)


@settings(max_examples=300, deadline=None)
@given(data=st.data())
def test_fast_pnpoly_equals_literal(probe, data):
    n = data.draw(st.integers(0, 8))
    coords = st.floats(-2, 2) | st.sampled_from([0.0, 0.5, -0.5, 1.0])
    vx = np.array([data.draw(coords) for _ in range(n)], dtype=np.float64)
    vy = np.array([data.draw(coords) for _ in range(n)], dtype=np.float64)
    px, py = data.draw(coords), data.draw(coords)
    a = probe.probe_polygon(vx.ctypes.data, vy.ctypes.data, n, px, py, 0)
    b = probe.probe_polygon(vx.ctypes.data, vy.ctypes.data, n, px, py, 1)
    assert a == b == int(fpo.polygon_inside(vx, vy, px, py))


@settings(max_examples=200, deadline=None)
@given(res=st.sampled_from([0.02, 0.01, 0.25]), px=st.sampled_from([0.0, 3.7]), cx=st.floats(-3, 3), cy=st.floats(-3, 3),
       r=st.sampled_from([0.02, 0.03, 0.25, 0.5]))
def test_bbox_and_cells_match_oracle_circle_iterator(probe, res, px, cx, cy, r):
    rows, cols = 240, 200
    rd = float(np.float32(r))
    lit = np.zeros(4, np.int32)
    fast = np.zeros(4, np.int32)
    probe.probe_bbox(rows, cols, res, px, -1.0, cx, cy, rd, lit.ctypes.data, fast.ctypes.data)
    assert lit.tolist() == fast.tolist()
    m = fpo.OracleMap(np.ones((rows, cols), np.float32), np.zeros((rows, cols), np.float32), res, (px, -1.0))
    cells = m.circle_cells(cx, cy, rd)
    # every visited cell lies in the engine's bounding box, in row-major order
    if len(cells):
        assert cells[:, 0].min() >= lit[0] and cells[:, 0].max() < lit[0] + lit[2]
        assert cells[:, 1].min() >= lit[1] and cells[:, 1].max() < lit[1] + lit[3]
    for (i, j) in cells[:3].tolist():
        ok, x, y = m.get_position(i, j)
        assert probe.probe_cell_pos(rows, cols, res, px, -1.0, 0, i) == x
        assert probe.probe_cell_pos(rows, cols, res, px, -1.0, 1, j) == y


@settings(max_examples=200, deadline=None)
@given(res=st.sampled_from([0.02, 0.01, 0.005]), x=st.floats(-2.6, 2.6), y=st.floats(-2.2, 2.2), R=st.sampled_from([0.06, 0.1, 0.15]))
def test_submap_geometry_matches_oracle_getSubmap(probe, res, x, y, R):
    rows, cols = int(round(4.8 / res)), int(round(4.0 / res))
    Rf = np.float32(R)
    lx, ly = float(Rf * 2), float(Rf)
    out = np.zeros(4, np.int32)
    base = np.zeros(2, np.float64)
    ok = probe.probe_submap(rows, cols, res, 0.0, 0.0, x, y, lx, ly, out.ctypes.data, base.ctypes.data)
    m = fpo.OracleMap(np.ones((rows, cols), np.float32), np.zeros((rows, cols), np.float32), res)
    ok_o, o, pl = m.submap_info(x, y, lx, ly)
    assert bool(ok) == ok_o
    if ok_o:
        assert out.tolist() == o.tolist()
        # submap getPosition(0,0) = subPos + (subLen/2 - res/2)
        assert base[0] == pl[0] + (0.5 * pl[2] - 0.5 * res) and base[1] == pl[1] + (0.5 * pl[3] - 0.5 * res)


def test_engine_geometry_reproduces_the_hand_traced_submaps(probe):
    """The hand-derived getSubmap answers of tests/test_oracle_kat.py (border clamp, and the 12-row rectangle of
    double(0.2f) / 0.02 = 10.00000015 cells) on the ENGINE's closed-form geometry."""
    out = np.zeros(4, np.int32)
    base = np.zeros(2, np.float64)
    ok = probe.probe_submap(10, 10, 1.0, 0.0, 0.0, 4.0, 0.0, 4.0, 2.0, out.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p))
    assert ok == 1 and out.tolist() == [0, 4, 4, 3]
    # submap cell (0, 0) = map cell (0, 4): centre (4.5, 0.5)
    assert base.tolist() == [4.5, 0.5]
    assert probe.probe_submap(10, 10, 1.0, 0.0, 0.0, 5.5, 0.0, 4.0, 2.0, out.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p)) == 0
    lx, ly = float(np.float32(0.1) * np.float32(2)), float(np.float32(0.1))
    cx = 2.0 - 0.5 * lx - 0.02 * 49.99999995
    assert probe.probe_submap(200, 200, 0.02, 0.0, 0.0, cx, 0.0, lx, ly, out.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p)) == 1
    assert out[0] == 49 and out[2] == 12
    assert probe.probe_submap(200, 200, 0.02, 0.0, 0.0, cx + 1e-8, 0.0, lx, ly, out.ctypes.data_as(C.c_void_p), base.ctypes.data_as(C.c_void_p)) == 1
    assert out[0] == 49 and out[2] == 11
