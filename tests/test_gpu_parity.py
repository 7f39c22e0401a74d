"""Parity tests proper: the HIP path (through the C ABI) against the oracle on the same seeded
inputs.  Bar: chosen grid indices / flags / x / y bit-exact, z within 1e-6 (north_star)."""
import numpy as np
import pytest

from oracle import fpo
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner, make_poses
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner():
    p = FootholdPlanner(0)
    yield p
    p.close()


def set_params(planner, **kw):
    planner.params = _capi.params_yaml()
    for k, v in kw.items():
        planner.params[k] = v


def test_cfg1_flat_reference_case(planner):
    """BASELINE configs[0]: trot, 8 cycles, 200x200 @2cm flat map, one pose (the reference's own case)."""
    set_params(planner)
    trav, elev, res, poses, n, _ = synth.make_config("cfg1")
    eng, ora = util.run_both(planner, trav, elev, res, poses, n)
    util.assert_plan_equal(eng, ora)
    assert eng["cycle_ok"].all() and (eng["nominal"]["source"] == 0).all()
    # closed form of SURVEY App. B.6
    s = 0.18000000715255737
    assert eng["nominal"][0, 3, 0]["x"] == pytest.approx(-1.0 + s / 2 + 3 * s + (0.21934999525547028 - 0.03999999910593033), abs=1e-12)


def test_cfg2_rough_terrain_batch(planner):
    """BASELINE configs[1] at a size the oracle finishes in seconds (full B=4096)."""
    set_params(planner)
    trav, elev, res, poses, n, _ = synth.make_config("cfg2")
    eng, ora = util.run_both(planner, trav, elev, res, poses, n, threads=8)
    util.assert_plan_equal(eng, ora)
    src = eng["nominal"]["source"]
    assert (src == 1).sum() > 100, "terrain should force spiral searches"
    assert (eng["centroid"]["code"] > 0).sum() > 100, "terrain should exercise the centroid row scan"


@pytest.mark.parametrize("res,R,rows", [(0.01, 0.1, 600), (0.01, 0.15, 600), (0.005, 0.1, 800), (0.03, 0.1, 300), (0.02, 0.06, 400),
                                        (0.005, 0.1587, 700)])  # the last: a centroid rectangle of 65 rows (> one 64-bit row mask)
def test_resolutions_and_radii(planner, res, R, rows):
    set_params(planner, searchRadius=np.float32(R))
    trav, elev = synth.rough_map(rows, rows, res, seed=11)
    side = rows * res
    poses = synth.poses_in_map(96, side, side, 6, 0.18, seed=12, margin=0.7)
    eng, ora = util.run_both(planner, trav, elev, res, poses, 6, threads=8)
    util.assert_plan_equal(eng, ora)


@pytest.mark.parametrize("group", ["4", "8", "16", "64", "65"])
def test_both_lane_groupings_agree_with_the_oracle(planner, group):
    """The chained kernel has two decompositions (16 lanes per leg = one wavefront per pose, 64 lanes
    per leg = one wavefront per leg); both must reproduce the oracle on 2 cm and 1 cm maps."""
    for res, rows, R in [(0.02, 400, 0.1), (0.01, 500, 0.12)]:
        set_params(planner, searchRadius=np.float32(R))
        trav, elev = synth.rough_map(rows, rows, res, seed=13, bad_frac=0.15)
        side = rows * res
        poses = synth.poses_in_map(160, side, side, 6, 0.18, seed=14, margin=0.7)
        poses["gait"][::3] = 1
        with planner.tuning(plan_group=int(group), no_bits=1):  # the direct kernels' lane groupings
            eng, ora = util.run_both(planner, trav, elev, res, poses, 6, threads=8)
        util.assert_plan_equal(eng, ora)
        assert (eng["nominal"]["source"] == 1).sum() > 50


@pytest.mark.parametrize("res,R,rf", [(0.02, 0.1, 0.02), (0.01, 0.1, 0.02), (0.02, 0.1, 0.03)])
def test_literal_disc_walk_equals_offset_table(planner, res, R, rf):
    """fpe_set_tuning("literal_discs") forces the per-candidate f64 bounding-box walk (the path taken when the host
    cannot prove the foot-disc offset table rounding-robust); both must match the oracle."""
    set_params(planner, searchRadius=np.float32(R), footRadius=np.float32(rf))
    rows = 400
    trav, elev = synth.rough_map(rows, rows, res, seed=17, bad_frac=0.2)
    side = rows * res
    poses = synth.poses_in_map(96, side, side, 5, 0.18, seed=18, margin=0.7)
    for literal in (False, True):
        with planner.tuning(literal_discs=int(literal)):
            eng, ora = util.run_both(planner, trav, elev, res, poses, 5, threads=8)
        util.assert_plan_equal(eng, ora)
    assert (eng["nominal"]["source"] == 1).sum() > 50


def test_exact_ties_take_the_exact_division_path(planner):
    """Binary-exact geometry (res 0.25, radii 0.25/0.5, poses on the lattice) puts positions exactly on
    cell boundaries, where the division-free index prediction must defer to the true division."""
    planner.params = _capi.params_yaml()
    for k, v in dict(footRadius=0.25, searchRadius=1.0, stepLength=1.0, length=2.0, width=1.0, l1=0.25, skew=0.25).items():
        planner.params[k] = np.float32(v)
    planner.params["lateralDrift"] = -0.125
    rng = np.random.default_rng(15)
    rows = cols = 160  # 40 x 40 m at 0.25 m
    trav = rng.uniform(0.4, 1.0, size=(rows, cols)).astype(np.float32)
    elev = rng.uniform(-0.5, 0.5, size=(rows, cols)).astype(np.float32)
    xs = rng.integers(-60, 20, size=128) * 0.125
    ys = rng.integers(-60, 60, size=128) * 0.125
    poses = make_poses(np.stack([xs, ys, np.zeros(128)], 1))
    eng, ora = util.run_both(planner, trav, elev, 0.25, poses, 6, threads=8)
    util.assert_plan_equal(eng, ora)
    assert (eng["nominal"]["source"] == 1).sum() > 20


def test_code_default_params_foot_radius_003(planner):
    planner.params = _capi.params_code_defaults()
    trav, elev = synth.rough_map(400, 400, 0.02, seed=21)
    poses = synth.poses_in_map(128, 8.0, 8.0, 6, 0.2, seed=22, margin=0.8)
    eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 6, threads=8)
    util.assert_plan_equal(eng, ora)


def test_rf_first_and_map_offset(planner):
    set_params(planner, RF_FIRST=1)
    trav, elev = synth.rough_map(400, 300, 0.02, seed=31, position=(3.7, -1.3))
    poses = synth.poses_uniform(128, (3.7 - 3.2, 3.7 + 1.0), (-1.3 - 2.2, -1.3 + 2.2), seed=32)
    eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 8, position=(3.7, -1.3), threads=8)
    util.assert_plan_equal(eng, ora)


def test_walk_gait_and_mixed_polygons(planner):
    """Build-defined extensions: walk gait, per-leg radii, hexagon polygons."""
    set_params(planner)
    trav, elev = synth.rough_map(500, 500, 0.01, seed=41)
    poses = synth.poses_in_map(128, 5.0, 5.0, 6, 0.18, seed=42, margin=0.7)
    rng = np.random.default_rng(43)
    poses["gait"] = rng.integers(0, 2, size=128)
    poses["leg_search_radius"] = rng.uniform(0.06, 0.15, size=(128, 4)).astype(np.float32)
    poses["leg_polygon_kind"] = rng.integers(0, 2, size=(128, 4))
    eng, ora = util.run_both(planner, trav, elev, 0.01, poses, 6, threads=8)
    util.assert_plan_equal(eng, ora)
    assert (eng["nominal"]["valid"] == 1).any()


def test_hostile_maps(planner):
    """NaN-rich, -inf/+inf, elevation >= 10, everything blocked."""
    set_params(planner)
    rng = np.random.default_rng(51)
    rows = cols = 300
    trav = rng.uniform(0.5, 1.0, size=(rows, cols)).astype(np.float32)
    trav[rng.random((rows, cols)) < 0.2] = np.nan
    trav[rng.random((rows, cols)) < 0.01] = -np.inf
    trav[rng.random((rows, cols)) < 0.01] = np.inf
    elev = rng.uniform(-0.2, 0.2, size=(rows, cols)).astype(np.float32)
    elev[rng.random((rows, cols)) < 0.1] = np.nan
    elev[rng.random((rows, cols)) < 0.05] = 11.0
    poses = synth.poses_in_map(128, 6.0, 6.0, 6, 0.18, seed=52, margin=0.7)
    eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 6, threads=8)
    util.assert_plan_equal(eng, ora)
    assert (eng["cycle_ok"] == 0).any() and (eng["cycle_ok"] == 1).any(), "commit and skip paths must both run"
    blocked = np.full((rows, cols), 0.1, np.float32)
    eng, ora = util.run_both(planner, blocked, elev, 0.02, poses[:16], 4)
    util.assert_plan_equal(eng, ora)
    assert not eng["cycle_ok"].any()


def test_poses_near_and_outside_the_border(planner):
    set_params(planner)
    trav, elev = synth.rough_map(200, 200, 0.02, seed=61)  # 4 x 4 m
    xs = np.linspace(-2.6, 2.3, 50)
    ys = np.linspace(-2.4, 2.4, 7)
    xyz = np.array([[x, y, 0.0] for x in xs for y in ys])
    poses = make_poses(xyz)
    eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 5, threads=8)
    util.assert_plan_equal(eng, ora)
    assert (eng["centroid"]["code"] == 6).any(), "some legs must fall outside the map"


def test_centroid_rectangle_reaching_the_index_past_the_far_edge(planner):
    """Random-campaign case 502981 (round 2): 280 columns at 4 cm centred at y = 4.8865..., legs next to the low-y edge.
    The rectangle's bounded bottom-right corner rounds to column 280 = size(1); grid_map's getSubmap fails there
    (getBufferRegionsForSubmap), so the centroid result is code 6 — the bit-window kernels used to scan the in-map
    columns while the direct kernels and the oracle read one cell past the row (tests/test_oracle_kat.py has the
    arithmetic).  Both kernel families must agree with the oracle now."""
    from tests.test_gpu_fuzz import make_case
    c = make_case(502981)
    for no_bits in (0, 1):
        with planner.tuning(plan_group=0, literal_discs=0, no_bits=no_bits):
            planner.params = c["params"]
            eng, ora = util.run_both(planner, c["trav"], c["elev"], c["res"], c["poses"], c["n"], position=c["pos"], threads=8)
            util.assert_plan_equal(eng, ora)
    assert (ora["centroid"]["code"][5, 0] == 6).any()


def test_packed_record_of_a_default_hit_whose_index_lies_outside_the_map(planner):
    """Random-campaign case 4400001 (round 4, found on the campaign's second case): a search centre just over the map's
    edge whose foot disc still reaches map cells is a VALID default hit, and its record carries getIndex(centre) (cpp:2016) —
    column -3 / -2 here.  The first form of the 8-byte exchange record kept 14 unsigned bits per index with a code for -1 and
    lost such indices; the record is biased now (include/fpe.h, FPE_PACKED_BIAS)."""
    from tests.test_gpu_fuzz import make_case
    c = make_case(4400001)
    for no_bits in (0, 1):
        with planner.tuning(plan_group=0, literal_discs=0, no_bits=no_bits):
            planner.params = c["params"]
            eng, ora = util.run_both(planner, c["trav"], c["elev"], c["res"], c["poses"], c["n"], position=c["pos"], threads=8,
                                     products=util.ALL_PRODUCTS)
            util.assert_plan_equal(eng, ora)
            nom = eng["nominal"]
            assert ((nom["valid"] == 1) & (nom["col"] < -1)).any(), "the case must hold a valid foothold with an index outside the map"
    set_params(planner)


def test_open_loop_checkFoothold_with_arbitrary_polygons(planner):
    set_params(planner)
    trav, elev = synth.rough_map(400, 400, 0.02, seed=71)
    planner.gridmapCallback(trav, elev, 0.02)
    rng = np.random.default_rng(72)
    n = 2000
    q = np.zeros(n, dtype=_capi.QUERY_DTYPE)
    q["cx"] = rng.uniform(-3.5, 3.5, n)
    q["cy"] = rng.uniform(-3.5, 3.5, n)
    q["search_radius"] = rng.choice(np.array([0.06, 0.1, 0.12], np.float32), n)
    for k in range(n):
        nv = int(rng.integers(0, 9))
        q["n_vertices"][k] = nv
        if nv:
            ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
            rad = rng.uniform(0.05, 0.2, nv)
            q["vx"][k, :nv] = q["cx"][k] + rng.uniform(-0.03, 0.03) + rad * np.cos(ang)
            q["vy"][k, :nv] = q["cy"][k] + rng.uniform(-0.03, 0.03) + rad * np.sin(ang)
    eng = planner.checkFoothold(q)
    ora = fpo.OracleMap(trav, elev, 0.02).search_legs(util.to_oracle_params(planner.params), util.to_oracle_queries(q))
    util.assert_nominal_equal(eng, ora, "checkFoothold")
    assert (eng["source"] == 1).sum() > 20


@pytest.mark.parametrize("res,rows", [(0.01, 700), (0.005, 900)])
def test_open_loop_polygons_on_the_staged_window_path(planner, res, rows):
    """Fine maps: the spiral window is staged through LDS and the polygon test runs per tile column
    (column_crossings): convex polygons take the two-crossing form, star-shaped (non-convex) ones have
    columns with four or more crossings and must fall back to the per-cell PNPOLY — both against the oracle."""
    set_params(planner)
    trav, elev = synth.rough_map(rows, rows, res, seed=73, bad_frac=0.15)
    planner.gridmapCallback(trav, elev, res)
    rng = np.random.default_rng(74)
    n = 700
    half = 0.5 * rows * res - 0.4
    q = np.zeros(n, dtype=_capi.QUERY_DTYPE)
    q["cx"] = rng.uniform(-half, half, n)
    q["cy"] = rng.uniform(-half, half, n)
    q["search_radius"] = rng.choice(np.array([0.06, 0.1, 0.15], np.float32), n)
    for k in range(n):
        nv = int(rng.integers(3, 9))
        q["n_vertices"][k] = nv
        ang = np.sort(rng.uniform(0, 2 * np.pi, nv))
        convex = k % 2 == 0
        rad = np.full(nv, rng.uniform(0.08, 0.2)) if convex else rng.uniform(0.03, 0.2, nv)
        q["vx"][k, :nv] = q["cx"][k] + rng.uniform(-0.03, 0.03) + rad * np.cos(ang)
        q["vy"][k, :nv] = q["cy"][k] + rng.uniform(-0.03, 0.03) + rad * np.sin(ang)
    planner.set_max_leg_search_radius(0.15)
    eng = planner.checkFoothold(q)
    planner.set_max_leg_search_radius(0.0)
    ora = fpo.OracleMap(trav, elev, res).search_legs(util.to_oracle_params(planner.params), util.to_oracle_queries(q))
    util.assert_nominal_equal(eng, ora, "checkFoothold")
    assert (eng["source"] == 1).sum() > 20 and (eng["source"] == 2).sum() > 5


def test_single_blocked_cell_kat_on_gpu(planner):
    set_params(planner)
    trav = np.ones((200, 200), np.float32)
    trav[100, 100] = 0.1
    planner.gridmapCallback(trav, np.zeros((200, 200), np.float32), 0.02)
    cx = 0.5 * 4.0 - 0.01 - 0.02 * 100
    R = float(np.float32(0.1))
    q = np.zeros(1, dtype=_capi.QUERY_DTYPE)
    q["cx"], q["cy"], q["search_radius"], q["n_vertices"] = cx, cx, np.float32(0.1), 4
    q["vx"][0, :4] = [cx + R, cx + R, cx - R, cx - R]
    q["vy"][0, :4] = [cx + 0.5 * R, cx - 0.5 * R, cx - 0.5 * R, cx + 0.5 * R]
    r = planner.checkFoothold(q)[0]
    assert (r["valid"], r["source"], r["row"], r["col"]) == (1, 1, 101, 99)


@pytest.mark.parametrize("rows,cols,si,sj", [(240, 200, 37, 151), (131, 167, 130, 1), (193, 128, 0, 127), (191, 260, 64, 128)])
def test_grid_map_message_layout_is_canonicalised(planner, rows, cols, si, sj):
    """Column-major buffers with a circular-buffer start index give the same plan as the canonical map (sizes on and off the
    transpose's 64 x 64 tiles, row lengths that are and are not a multiple of four cells: 16-byte and 4-byte stores)."""
    set_params(planner)
    trav, elev = synth.rough_map(rows, cols, 0.02, seed=81)
    poses = synth.poses_in_map(64, rows * 0.02, cols * 0.02, 5, 0.18, seed=82, margin=0.7)
    planner.gridmapCallback(trav, elev, 0.02)
    ref = planner.plan(poses, 5)
    # buffer (bi, bj) holds unwrapped ((bi - si) % rows, (bj - sj) % cols); stored column-major
    buf_t = np.roll(trav, (si, sj), axis=(0, 1))
    buf_e = np.roll(elev, (si, sj), axis=(0, 1))
    planner.gridmapCallback(np.ascontiguousarray(buf_t.T), np.ascontiguousarray(buf_e.T), 0.02, start_index=(si, sj),
                            storage_order="col")
    got = planner.plan(poses, 5)
    for k in ref:
        a, b = ref[k], got[k]
        assert a.tobytes() == b.tobytes() or np.array_equal(a, b, equal_nan=True) if a.dtype.names is None else a.tobytes() == b.tobytes(), k


def test_service_message(planner):
    """plan_global_footholds response content (cpp:591-699, 1378-1396, 1574) vs the oracle's cycle flags."""
    set_params(planner)
    trav, elev = synth.rough_map(400, 400, 0.02, seed=1, bad_frac=0.45)  # harsh: some cycles must fail
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    rng = np.random.default_rng(91)
    seen_fail = False
    answered = 0
    for _ in range(40):
        pos = [rng.uniform(-3.2, -2.0), rng.uniform(-3, 3), 0.25]
        msg = util.service_enforced(planner, 8, pos)
        o = omap.plan(util.to_oracle_params(planner.params), util.to_oracle_poses(make_poses([pos])), 8)
        ok = o["cycle_ok"][0]
        if util.oracle_service_gate(omap, planner, pos, 8, o) != 255:
            # the opt track's getGaitCycleSearchGridMap fails in some cycle: the handler returns false (cpp:931-934).  On
            # terrain this harsh the opt track derails often: an untouched centroid result (0,0,0) is far outside the
            # optimiser's box, the optimiser call throws, the feet land on stale positions (cpp:1030-1041, 1224-1226, 1287)
            assert msg is False
            continue
        answered += 1
        assert msg["gait_cycles"] == 8
        assert msg["success"] == bool(ok[-1])
        assert msg["gait_cycles_succeed"] == (int(np.nonzero(ok)[0][-1]) + 1 if ok.any() else 0)
        f = msg["footholds"]
        assert len(f) == 4 + 4 * int(ok.sum())
        assert np.array_equal(np.stack([f["x"][:4], f["y"][:4], f["z"][:4]], 1), o["stance"][0])
        assert f["gait_cycle_id"][:4].tolist() == [0, 0, 0, 0] and f["foot_id"][:4].tolist() == [0, 1, 2, 3]
        k = 4
        for g in range(8):
            if ok[g]:
                for l in range(4):
                    assert f["x"][k] == o["nominal"][0, g, l]["x"] and f["y"][k] == o["nominal"][0, g, l]["y"]
                    assert abs(f["z"][k] - float(o["nominal"][0, g, l]["z"])) <= util.Z_TOL
                    assert f["foot_id"][k] == l and f["gait_cycle_id"][k] == g
                    k += 1
        seen_fail |= not ok.all()
    assert seen_fail, "the harsh map should make some cycles fail (commit/skip path)"
    assert answered >= 5, answered


@pytest.mark.parametrize("group", ["4", "8"])
@pytest.mark.parametrize("B", [1, 2, 3, 5, 7])
def test_ragged_batches_pad_the_last_wavefront(planner, group, B):
    """B not a multiple of the poses-per-wavefront (2 at 8 lanes per leg, 4 at 4): the padding poses of
    the last wavefront must neither store nor disturb the live ones."""
    set_params(planner)
    trav, elev = synth.rough_map(300, 300, 0.02, seed=23, bad_frac=0.2)
    poses = synth.poses_in_map(B, 6.0, 6.0, 5, 0.18, seed=24 + B, margin=0.7)
    for no_bits in (1, 0):  # direct kernel of that grouping, then the bit-window kernel (automatic grouping)
        with planner.tuning(plan_group=int(group) if no_bits else 0, no_bits=no_bits):
            eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 5)
        util.assert_plan_equal(eng, ora)


def test_maximum_gait_cycles_255(planner):
    """uint8 gait_cycles (srv:5): 255 cycles, gait_cycle_id up to 254; trajectories leave the map at the end."""
    set_params(planner)
    trav, elev = synth.rough_map(400, 400, 0.02, seed=25)
    poses = synth.poses_uniform(8, (-3.3, -3.0), (-1.0, 2.5), seed=26)
    eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 255)
    util.assert_plan_equal(eng, ora)
    assert eng["nominal"]["gait_cycle_id"][:, 254, :].tolist() == [[254] * 4] * 8
    assert (eng["centroid"]["code"] == 6).any(), "late cycles run off the map: getSubmap must fail there"
    omap = fpo.OracleMap(trav, elev, 0.02)
    answered = 0
    for b in range(8):
        msg = util.service_enforced(planner, 255, poses["position"][b])
        if util.oracle_service_gate(omap, planner, poses["position"][b], 255) != 255:
            assert msg is False
            continue
        answered += 1
        assert msg["gait_cycles"] == 255 and len(msg["footholds"]) == 4 + 4 * int(ora["cycle_ok"][b].sum())
    assert answered > 0


def test_service_all_tracks(planner):
    """N2: centroid message and default-track rows of the same call (cpp:1338-1348, 1444-1483)."""
    set_params(planner)
    trav, elev = synth.rough_map(400, 400, 0.02, seed=1, bad_frac=0.3)
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    rng = np.random.default_rng(92)
    seen_split = False
    for _ in range(48):
        pos = [rng.uniform(-3.2, -2.0), rng.uniform(-3, 3), 0.0]
        r = util.service_enforced(planner, 6, pos, all_tracks=True)
        o = omap.plan(util.to_oracle_params(planner.params), util.to_oracle_poses(make_poses([pos])), 6)
        if util.oracle_service_gate(omap, planner, pos, 6, o) != 255:
            assert r is False
            continue
        ok = o["cycle_ok"][0].astype(bool)
        cf = r["centroid"]["footholds"]
        assert len(cf) == 4 + 4 * int(ok.sum())
        # centroidGlobalFootholdsMsg_ keeps its own books (cpp:711, 1445-1446): success = some cycle committed,
        # gait_cycles_succeed = last committed cycle + 1, gait_cycles never written; the nominal message's success
        # is the LAST cycle's validity (cpp:1380, 1574)
        assert r["centroid"]["success"] == bool(ok.any())
        assert r["centroid"]["gait_cycles_succeed"] == (int(np.nonzero(ok)[0][-1]) + 1 if ok.any() else 0)
        assert r["centroid"]["gait_cycles"] == 0
        assert r["success"] == bool(ok[-1]) and r["gait_cycles"] == 6
        seen_split |= bool(ok.any() and not ok[-1])
        want = o["centroid"][0][ok].reshape(-1)
        assert np.array_equal(cf["x"][4:], want["x"]) and np.array_equal(cf["y"][4:], want["y"])
        assert np.all(np.abs(cf["z"][4:] - want["z"].astype(np.float64)) <= util.Z_TOL)
        d = r["default_footholds"]
        assert d.shape == (1 + int(ok.sum()), 4, 3)
        assert np.array_equal(d[0], o["stance"][0])
        assert np.array_equal(d[1:, :, :2], o["default"][0][ok][:, :, :2])
        assert np.all(np.abs(d[1:, :, 2] - o["default"][0][ok][:, :, 2]) <= util.Z_TOL)
    assert seen_split, "need a plan whose last cycle fails after an earlier one committed (centroid.success != nominal.success)"


@pytest.mark.parametrize("rf_first", [0, 1])
def test_service_track_reports(planner, rf_first):
    """N2: feet-centre paths (cpp:2191-2196) and KPIs (getHipDistance cpp:2571-2584, getCogSpeed cpp:2587-2623)
    of the nominal and centroid tracks: x/y and the KPIs bit-exact, path z within the z tolerance."""
    set_params(planner, RF_FIRST=rf_first)
    trav, elev = synth.rough_map(400, 400, 0.02, seed=5, bad_frac=0.3)
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    rng = np.random.default_rng(93 + rf_first)
    committed = 0
    for _ in range(24):
        pos = [rng.uniform(-3.2, -2.0), rng.uniform(-3, 3), 0.0]
        r = util.service_enforced(planner, 7, pos, all_tracks=True)
        if util.oracle_service_gate(omap, planner, pos, 7) != 255:
            assert r is False
            continue
        want = omap.plan_products(util.to_oracle_params(planner.params), util.to_oracle_poses(make_poses([pos])), 7)
        for got, w in ((r["report"], want["nominal"]), (r["centroid"]["report"], want["centroid"])):
            # centroidFeetCenterPath also receives the opt track's feet centre of every cycle (cpp:946): its entries
            # alternate centroid, opt (tests/test_gpu_opt.py checks the odd ones against the oracle's opt track)
            path = got["path"][0::2] if got is r["centroid"]["report"] else got["path"]
            assert path.shape == (7, 3) == w["path"].shape
            assert np.array_equal(path[:, :2], w["path"][:, :2])
            assert np.all(np.abs(path[:, 2] - w["path"][:, 2]) <= util.Z_TOL)
            assert np.array_equal(got["feet_distance"], w["feet_distance"])
            assert np.array_equal(got["cog_speed"], w["cog_speed"])
        assert r["centroid"]["report"]["path"].shape == (14, 3)
        committed += len(r["report"]["cog_speed"]) // 2
    assert 0 < committed < 24 * 7, "want both committed and skipped cycles"
    set_params(planner)


def test_errors_are_codes_not_crashes(planner):
    from quadrupedal_foothold_planner_amd.planner import FpeError

    fresh = FootholdPlanner(0)
    with pytest.raises(FpeError) as e:
        fresh.plan(make_poses([[0, 0, 0]]), 2)
    assert e.value.code == _capi.FPE_E_NO_MAP
    fresh.close()
    set_params(planner, searchRadius=np.float32(5.0))
    planner.gridmapCallback(*synth.flat_map(100, 100), 0.02)
    with pytest.raises(FpeError) as e:
        planner.plan(make_poses([[0, 0, 0]]), 2)
    assert e.value.code == _capi.FPE_E_UNSUPPORTED
    set_params(planner)
    with pytest.raises(FpeError) as e:
        planner.plan(make_poses([[np.nan, 0, 0]]), 2)
    assert e.value.code == _capi.FPE_E_INVALID_ARG


def test_concurrent_uploads_and_plans_see_whole_snapshots(planner):
    """SURVEY 8(b) threading row: the reference runs gridmapCallback and the service handler on an
    AsyncSpinner thread pool with no locking (hpp:535, cpp:506).  Here fpe_upload_map and fpe_plan /
    fpe_plan_service are called from different threads at once (ctypes drops the GIL): every plan must
    equal the oracle's plan on ONE of the two maps being swapped — never a mixture, never an error."""
    import threading

    set_params(planner)
    maps = [synth.rough_map(300, 300, 0.02, seed=61, bad_frac=0.1), synth.rough_map(300, 300, 0.02, seed=62, bad_frac=0.25)]
    poses = synth.poses_in_map(48, 6.0, 6.0, 6, 0.18, seed=63, margin=0.7)
    want, gate = [], []
    for trav, elev in maps:
        om = fpo.OracleMap(trav, elev, 0.02)
        o = om.plan(util.to_oracle_params(planner.params), util.to_oracle_poses(poses), 6, threads=4)
        want.append(o)
        gate.append(om.plan_opt(util.to_oracle_params(planner.params), util.to_oracle_opt_params(planner.opt_params),
                                util.to_oracle_poses(poses), 6, o["cycle_ok"])["gate_fail_cycle"])
    assert not np.array_equal(want[0]["nominal"]["row"], want[1]["nominal"]["row"]), "the two maps must plan differently"
    planner.gridmapCallback(maps[0][0], maps[0][1], 0.02)
    stop = threading.Event()
    errors, seen = [], [0, 0]

    def uploader():
        k = 0
        try:
            while not stop.is_set():
                k += 1
                planner.gridmapCallback(maps[k & 1][0], maps[k & 1][1], 0.02)
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    def client(service):
        try:
            for it in range(60):
                if service:
                    msg = planner.globalFootholdPlan(6, poses["position"][it % 48])
                    b = it % 48
                    hits = []
                    for m in (0, 1):
                        if gate[m][b] != 255 or msg is False:  # the handler returns false on this map (cpp:931-934)
                            hits.append(msg is False and gate[m][b] != 255)
                            continue
                        ok = want[m]["cycle_ok"][b].astype(bool)
                        ref = want[m]["nominal"][b][ok].reshape(-1)
                        f = msg["footholds"][4:]
                        hits.append(len(f) == len(ref) and np.array_equal(f["x"], ref["x"]) and np.array_equal(f["y"], ref["y"]))
                    if not any(hits):
                        errors.append(f"service call {it}: response matches neither map")
                else:
                    eng = planner.plan(poses, 6, products=("nominal", "cycle_ok"))
                    which = [m for m in (0, 1) if np.array_equal(eng["nominal"]["row"], want[m]["nominal"]["row"])
                             and np.array_equal(eng["nominal"]["x"], want[m]["nominal"]["x"])
                             and np.array_equal(eng["cycle_ok"], want[m]["cycle_ok"])]
                    if not which:
                        errors.append(f"plan {it}: result is a mixture of snapshots")
                    else:
                        seen[which[0]] += 1
        except Exception as e:  # pragma: no cover
            errors.append(repr(e))

    ths = [threading.Thread(target=uploader)] + [threading.Thread(target=client, args=(s,)) for s in (False, False, True)]
    planner.set_tuning(service_opt_gate=2)  # the service client compares with the oracle's opt-track gate of ANY cycle
    try:
        for t in ths:
            t.start()
        for t in ths[1:]:
            t.join()
        stop.set()
        ths[0].join()
    finally:
        stop.set()
        planner.set_tuning(service_opt_gate=2)  # (the default)
    assert not errors, errors[:3]
    assert seen[0] + seen[1] == 120


@pytest.mark.parametrize("rows,cols", [(1, 1), (2, 3), (5, 5), (7, 9), (16, 4), (33, 17)])
@pytest.mark.parametrize("res", [0.02, 0.005])
def test_tiny_maps_smaller_than_every_window(planner, rows, cols, res):
    """Maps smaller than the search window, the centroid rectangle and even one 16-byte row group: every
    iterator is clamped to the map (or fails like getSubmap does) exactly as in the oracle."""
    set_params(planner)
    rng = np.random.default_rng(rows * 100 + cols)
    trav = rng.uniform(0.5, 1.0, (rows, cols)).astype(np.float32)
    elev = rng.uniform(-0.1, 0.1, (rows, cols)).astype(np.float32)
    poses = make_poses(np.column_stack([rng.uniform(-0.6, 0.3, 24), rng.uniform(-0.3, 0.3, 24), np.zeros(24)]))
    poses["gait"] = rng.integers(0, 2, 24)
    eng, ora = util.run_both(planner, trav, elev, res, poses, 4)
    util.assert_plan_equal(eng, ora)


@pytest.mark.parametrize("ratio", [0.85, 0.9001, 0.95, 1.0, 1.2, 1.45])
def test_foot_radius_ratios_around_the_middle_cell_shortcut(planner, ratio):
    """footRadius / resolution from just below to well above 0.9: from 0.9 on, the 8-lane kernel takes the middle cell
    of an unclamped 3x3 disc box for granted (PlanConsts::midCellInside).  Lattice-aligned and random poses, border
    poses (clamped boxes fall back to the generic walk), default hits and spiral candidates — all against the oracle."""
    res = 0.02
    set_params(planner, footRadius=np.float32(ratio * res))
    trav, elev = synth.rough_map(260, 260, res, seed=int(ratio * 1000), bad_frac=0.12)
    rng = np.random.default_rng(int(ratio * 977))
    xs = rng.uniform(-2.9, 2.0, 160)
    ys = rng.uniform(-2.9, 2.9, 160)
    xs[:40], ys[:40] = np.round(xs[:40] / res) * res, np.round(ys[:40] / res) * res          # cell corners / centres
    xs[40:60], ys[40:60] = np.round(xs[40:60] / res) * res + 0.5 * res, ys[40:60]              # half-cell offsets
    poses = make_poses(np.column_stack([xs, ys, np.zeros(160)]))
    eng, ora = util.run_both(planner, trav, elev, res, poses, 5, threads=8)
    util.assert_plan_equal(eng, ora)
    set_params(planner)


def test_generic_and_3x3_only_kernel_variants_agree(planner):
    """foot radius in [0.9, 1] x resolution launches the 3x3-only variant of the 8-lane kernel (no generic disc issue
    compiled in; clamped boxes at the map border take the direct pass).  fpe_set_tuning("no_mid_variant") forces the generic
    kernel, "no_bits" the direct kernels: both must reproduce the oracle on a map whose border is inside the pose range."""
    set_params(planner)
    trav, elev = synth.rough_map(220, 220, 0.02, seed=81, bad_frac=0.15)
    rng = np.random.default_rng(82)
    poses = make_poses(np.column_stack([rng.uniform(-2.4, 2.3, 200), rng.uniform(-2.3, 2.3, 200), np.zeros(200)]))
    for no_mid in (0, 1):
        for no_bits in (0, 1):
            with planner.tuning(no_mid_variant=no_mid, no_bits=no_bits):
                eng, ora = util.run_both(planner, trav, elev, 0.02, poses, 6, threads=8)
            util.assert_plan_equal(eng, ora)
    assert (eng["nominal"]["valid"] == 0).any() and (eng["centroid"]["code"] == 6).any(), "poses must reach the border"


def test_map_stream_with_changing_sizes_recycles_buffers_correctly(planner):
    """A stream of maps of different sizes and layouts through one engine: snapshot layers and the upload staging layer
    are recycled by size (BufferPool); every plan must see exactly the map uploaded last."""
    set_params(planner)
    rng = np.random.default_rng(101)
    shapes = [(120, 160), (200, 150), (120, 160), (90, 90), (200, 150), (120, 160), (300, 80), (90, 90)]
    for k, (rows, cols) in enumerate(shapes * 2):
        trav, elev = synth.rough_map(rows, cols, 0.02, seed=200 + k, bad_frac=0.15)
        poses = synth.poses_in_map(16, rows * 0.02, cols * 0.02, 3, 0.18, seed=300 + k, margin=0.45)
        if k % 2:  # grid_map message layout: column-major with a circular-buffer start index
            si, sj = int(rng.integers(0, rows)), int(rng.integers(0, cols))
            msg_t = np.ascontiguousarray(np.roll(trav, (si, sj), axis=(0, 1)).T)
            msg_e = np.ascontiguousarray(np.roll(elev, (si, sj), axis=(0, 1)).T)
            planner.gridmapCallback(msg_t, msg_e, 0.02, start_index=(si, sj), storage_order="col")
        else:
            planner.gridmapCallback(trav, elev, 0.02)
        eng = planner.plan(poses, 3)
        ora = fpo.OracleMap(trav, elev, 0.02).plan(util.to_oracle_params(planner.params), util.to_oracle_poses(poses), 3)
        util.assert_plan_equal(eng, ora)


@pytest.mark.parametrize("n", [1, 5, 31, 32, 33, 257])
def test_open_loop_ragged_query_counts(planner, n):
    """fpe_search_legs plans 32 queries per workgroup on small windows (8 lanes each): counts that do not fill the last
    workgroup or wavefront must neither read nor write beyond the arrays."""
    set_params(planner)
    trav, elev = synth.rough_map(200, 200, 0.02, seed=121, bad_frac=0.2)
    planner.gridmapCallback(trav, elev, 0.02)
    rng = np.random.default_rng(122 + n)
    q = np.zeros(n, dtype=_capi.QUERY_DTYPE)
    q["cx"], q["cy"] = rng.uniform(-1.9, 1.9, n), rng.uniform(-1.9, 1.9, n)
    q["search_radius"], q["n_vertices"] = np.float32(0.1), 4
    q["vx"][:, :4] = q["cx"][:, None] + np.array([0.1, 0.1, -0.1, -0.1])
    q["vy"][:, :4] = q["cy"][:, None] + np.array([0.05, -0.05, -0.05, 0.05])
    eng = planner.checkFoothold(q)
    ora = fpo.OracleMap(trav, elev, 0.02).search_legs(util.to_oracle_params(planner.params), util.to_oracle_queries(q))
    util.assert_nominal_equal(eng, ora, "checkFoothold")


@pytest.mark.parametrize("n_cycles", [3, 11])
def test_round3_kernel_paths_at_the_map_border(planner, n_cycles):
    """The round-3 forms of the three bit-window kernel families on maps whose borders lie inside the pose range, with a
    cycle count that leaves the last flush batch partial: generic 8-lane kernels at 1 cm (5x5 disc boxes by box rows, 32-bit
    membership masks finished by flush_unit_g every fourth cycle, four table entries per lane in the scan, E cleared outside
    the map), 64-bit rows at 1 cm / R 0.15 (two lanes per staged record), 96-bit rows at 0.5 cm with per-leg radii on both
    sides of the one-row-slot limit and both polygon kinds (nested erosion, polygon columns within reach only)."""
    rng = np.random.default_rng(900 + n_cycles)
    # generic 8-lane kernel
    set_params(planner)
    trav, elev = synth.rough_map(260, 300, 0.01, seed=91, bad_frac=0.15, nan_frac=0.01)
    xs, ys = rng.uniform(-1.6, 1.4, 160), rng.uniform(-1.7, 1.7, 160)
    xs[:30], ys[:30] = np.round(xs[:30] / 0.01) * 0.01, np.round(ys[:30] / 0.01) * 0.01
    poses = make_poses(np.column_stack([xs, ys, np.zeros(160)]))
    poses["gait"] = rng.integers(0, 2, 160)
    eng, ora = util.run_both(planner, trav, elev, 0.01, poses, n_cycles, threads=8)
    assert planner.describe_plan().startswith("plan_bits_kernel<3, false>"), planner.describe_plan()
    util.assert_plan_equal(eng, ora)
    assert (eng["centroid"]["code"] == 6).any() and (eng["nominal"]["source"] == 1).any()
    # 64-bit rows
    set_params(planner, searchRadius=np.float32(0.15))
    eng, ora = util.run_both(planner, trav, elev, 0.01, poses[:96], n_cycles, threads=8)
    assert planner.describe_plan().startswith("plan_bits_seq_kernel<1, 2>"), planner.describe_plan()
    util.assert_plan_equal(eng, ora)
    # 96-bit rows, mixed radii and polygons
    set_params(planner)
    trav5, elev5 = synth.rough_map(420, 380, 0.005, seed=92, bad_frac=0.2, nan_frac=0.01)
    p5 = make_poses(np.column_stack([rng.uniform(-1.2, 1.0, 64), rng.uniform(-1.1, 1.1, 64), np.zeros(64)]))
    p5["gait"] = rng.integers(0, 2, 64)
    p5["leg_search_radius"] = rng.uniform(0.05, 0.15, (64, 4)).astype(np.float32)
    p5["leg_polygon_kind"] = rng.integers(0, 2, (64, 4))
    planner.set_max_leg_search_radius(0.15)
    try:
        eng, ora = util.run_both(planner, trav5, elev5, 0.005, p5, n_cycles, threads=8)
        assert planner.describe_plan().startswith("plan_bits_seq_kernel<2, 3>"), planner.describe_plan()
        util.assert_plan_equal(eng, ora)
        assert (eng["nominal"]["source"] == 1).any()
    finally:
        planner.set_max_leg_search_radius(0.0)
        set_params(planner)
