"""Parity of the DEVICE-RESIDENT entry points (what bench.py times) and of the open-loop mode against the
chained plan.  Everything goes through the C ABI; torch is only the owner of device buffers and streams."""
import numpy as np
import pytest

from oracle import fpo
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner, FpeError, make_poses
from tests import util

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def planner():
    p = FootholdPlanner(0)
    yield p
    p.close()


def _device_plan(planner, torch, dev, stream, trav, elev, res, poses, n_cycles, upload_stream=None):
    """upload_map_device + plan_device with torch-owned buffers; returns the same dict planner.plan() does."""
    rows, cols = trav.shape
    B = poses.shape[0]
    n_rec = B * n_cycles * 4
    up = upload_stream or stream
    with torch.cuda.stream(up):
        d_trav = torch.from_numpy(trav).to(dev, non_blocking=False)
        d_elev = torch.from_numpy(elev).to(dev, non_blocking=False)
        planner.upload_map_device(d_trav.data_ptr(), d_elev.data_ptr(), rows, cols, res, stream=up.cuda_stream)
    with torch.cuda.stream(stream):
        d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1).copy()).to(dev)
        bufs = {
            "nominal": torch.zeros(n_rec * _capi.FOOTHOLD_DTYPE.itemsize, dtype=torch.uint8, device=dev),
            "centroid": torch.zeros(n_rec * _capi.CENTROID_DTYPE.itemsize, dtype=torch.uint8, device=dev),
            "default": torch.zeros(n_rec * 3, dtype=torch.float64, device=dev),
            "cycle_ok": torch.zeros(B * n_cycles, dtype=torch.uint8, device=dev),
            "stance": torch.zeros(B * 12, dtype=torch.float64, device=dev),
            "selected": torch.zeros(n_rec * _capi.SELECTED_DTYPE.itemsize, dtype=torch.uint8, device=dev),
            "pose_status": torch.zeros(B, dtype=torch.uint8, device=dev),
        }
    stream.synchronize()  # the input copies above ran on `stream`; the plan below is asynchronous on it
    planner.plan_device(d_poses.data_ptr(), B, n_cycles, bufs["nominal"].data_ptr(), bufs["centroid"].data_ptr(),
                        bufs["default"].data_ptr(), bufs["cycle_ok"].data_ptr(), bufs["stance"].data_ptr(),
                        stream=stream.cuda_stream, d_selected_ptr=bufs["selected"].data_ptr(),
                        d_pose_status_ptr=bufs["pose_status"].data_ptr())
    stream.synchronize()
    host = {k: v.cpu().numpy() for k, v in bufs.items()}
    return {
        "nominal": host["nominal"].view(_capi.FOOTHOLD_DTYPE).reshape(B, n_cycles, 4),
        "centroid": host["centroid"].view(_capi.CENTROID_DTYPE).reshape(B, n_cycles, 4),
        "default": host["default"].reshape(B, n_cycles, 4, 3),
        "cycle_ok": host["cycle_ok"].reshape(B, n_cycles),
        "stance": host["stance"].reshape(B, 4, 3),
        "selected": host["selected"].view(_capi.SELECTED_DTYPE).reshape(B, n_cycles, 4),
        "pose_status": host["pose_status"],
    }


def _oracle_plan(planner, trav, elev, res, poses, n_cycles):
    omap = fpo.OracleMap(trav, elev, res)
    op, opo = util.to_oracle_params(planner.params), util.to_oracle_poses(poses)
    ora = omap.plan(op, opo, n_cycles, threads=8)
    ora["pose_status"] = omap.pose_status(op, opo)
    return ora


@pytest.mark.parametrize("case", ["headline_2cm", "rough_1cm_r012", "walk_mixed_1cm"])
def test_device_entry_points_on_a_side_stream(planner, case):
    """fpe_upload_map_device + fpe_plan_device (the calls bench.py times) with torch buffers on a NON-default
    stream, compared record by record with the oracle."""
    import torch

    dev = torch.device("cuda", 0)
    planner.params = _capi.params_yaml()
    if case == "headline_2cm":
        trav, elev = synth.rough_map(1000, 1000, 0.02, seed=1)
        res, n = 0.02, 8
        poses = synth.poses_in_map(1024, 20.0, 20.0, n, 0.18, seed=6)
    elif case == "rough_1cm_r012":
        planner.params["searchRadius"] = np.float32(0.12)
        trav, elev = synth.rough_map(700, 700, 0.01, seed=31, bad_frac=0.1)
        res, n = 0.01, 6
        poses = synth.poses_in_map(256, 7.0, 7.0, n, 0.18, seed=32, margin=0.7)
    else:
        planner.params["searchRadius"] = np.float32(0.15)
        trav, elev = synth.rough_map(600, 600, 0.01, seed=33, bad_frac=0.08)
        res, n = 0.01, 5
        poses = synth.poses_in_map(128, 6.0, 6.0, n, 0.18, seed=34, margin=0.7)
        poses["gait"][::2] = 1
    side = torch.cuda.Stream(device=dev)
    eng = _device_plan(planner, torch, dev, side, trav, elev, res, poses, n)
    util.assert_plan_equal(eng, _oracle_plan(planner, trav, elev, res, poses, n))
    planner.params = _capi.params_yaml()


def test_plan_on_one_stream_right_after_an_upload_on_another(planner):
    """The new snapshot is published before its asynchronous device-to-device upload has completed: a plan on a
    DIFFERENT stream must wait for it (MapSnapshot::ready), and an upload that follows an asynchronous plan must not
    recycle the layers that plan still reads (MapSnapshot::note_use)."""
    import torch

    dev = torch.device("cuda", 0)
    planner.params = _capi.params_yaml()
    s_up, s_plan = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    n = 6
    for k in range(6):
        rows = 600 + 40 * (k % 3)
        trav, elev = synth.rough_map(rows, rows, 0.02, seed=400 + k, bad_frac=0.1)
        poses = synth.poses_in_map(512, rows * 0.02, rows * 0.02, n, 0.18, seed=500 + k, margin=0.7)
        eng = _device_plan(planner, torch, dev, s_plan, trav, elev, 0.02, poses, n, upload_stream=s_up)
        util.assert_plan_equal(eng, _oracle_plan(planner, trav, elev, 0.02, poses, n))


def test_search_legs_device_on_a_side_stream(planner):
    """fpe_search_legs_device (bench.py's open_loop leg) against the oracle's checkFoothold, 2 cm and 1 cm."""
    import torch

    dev = torch.device("cuda", 0)
    side = torch.cuda.Stream(device=dev)
    for res, rows, R in [(0.02, 500, 0.1), (0.01, 500, 0.15)]:
        planner.params = _capi.params_yaml()
        planner.params["searchRadius"] = np.float32(R)
        trav, elev = synth.rough_map(rows, rows, res, seed=41, bad_frac=0.1)
        planner.gridmapCallback(trav, elev, res)
        rng = np.random.default_rng(42)
        nq = 4096
        half = 0.5 * rows * res - 0.4
        q = np.zeros(nq, dtype=_capi.QUERY_DTYPE)
        q["cx"], q["cy"] = rng.uniform(-half, half, nq), rng.uniform(-half, half, nq)
        q["search_radius"], q["n_vertices"] = np.float32(R), 4
        Rd = float(np.float32(R))
        q["vx"][:, :4] = q["cx"][:, None] + np.array([Rd, Rd, -Rd, -Rd])
        q["vy"][:, :4] = q["cy"][:, None] + 0.5 * np.array([Rd, -Rd, -Rd, Rd])
        with torch.cuda.stream(side):
            d_q = torch.from_numpy(q.view(np.uint8).reshape(-1).copy()).to(dev)
            d_o = torch.zeros(nq * _capi.FOOTHOLD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        side.synchronize()
        planner.search_legs_device(d_q.data_ptr(), nq, d_o.data_ptr(), stream=side.cuda_stream)
        side.synchronize()
        eng = d_o.cpu().numpy().view(_capi.FOOTHOLD_DTYPE)
        ora = fpo.OracleMap(trav, elev, res).search_legs(util.to_oracle_params(planner.params), util.to_oracle_queries(q))
        util.assert_nominal_equal(eng, ora, "search_legs_device")
        assert (eng["source"] == 1).sum() > 100
    planner.params = _capi.params_yaml()


@pytest.mark.parametrize("res,rows,R", [(0.02, 500, 0.1), (0.01, 600, 0.12)])
def test_open_loop_fed_with_chained_centres_equals_chained(planner, res, rows, R):
    """the open-loop mode must equal the chained mode when fed the chained centres.  Query (b, g, leg)
    = checkFoothold(centre = the centroid track's next default position, polygon = getSearchPolygon around the
    nominal track's next default position) — both reconstructed on the host from the chained plan's committed
    results with the reference's expression order (cpp:2199-2213, 2411-2418, 2496-2517)."""
    planner.params = _capi.params_yaml()
    planner.params["searchRadius"] = np.float32(R)
    trav, elev = synth.rough_map(rows, rows, res, seed=51, bad_frac=0.12)
    side = rows * res
    n = 6
    poses = synth.poses_in_map(200, side, side, n, 0.18, seed=52, margin=0.7)
    planner.gridmapCallback(trav, elev, res)
    ch = planner.plan(poses, n)
    c = fpo.constants(util.to_oracle_params(planner.params))
    step, step_half = c["step"], c["stepHalf"]
    Rd = float(np.float32(R))
    drift = float(planner.params["lateralDrift"][0])
    B = poses.shape[0]
    q = np.zeros((B, n, 4), dtype=_capi.QUERY_DTYPE)
    for b in range(B):
        y0 = float(poses["position"][b, 1])
        cur = np.zeros((2, 4, 3))  # [centroid, nominal] current feet: shifted stance (setFirstGait, cpp:2693)
        for t in range(2):
            cur[t] = ch["stance"][b]
            cur[t][:, 0] = ch["stance"][b][:, 0] - step_half
        adj = 0.0
        for g in range(n):
            ctr = [fpo.polygon_center(cur[t])[0] for t in range(2)]
            ny = y0 + adj
            for l in range(4):
                cx = (ctr[0] + step) + c["biasX"][l]      # centroid track (cpp:2199, 2414)
                px = (ctr[1] + step) + c["biasX"][l]      # nominal track: polygon centre
                cy = ny + c["biasY"][l]
                q[b, g, l]["cx"], q[b, g, l]["cy"] = cx, cy
                q[b, g, l]["vx"][:4] = [px + Rd, px + Rd, px - Rd, px - Rd]
                q[b, g, l]["vy"][:4] = [cy + 0.5 * Rd, cy - 0.5 * Rd, cy - 0.5 * Rd, cy + 0.5 * Rd]
            if ch["cycle_ok"][b, g]:
                for l in range(4):
                    cen, nom = ch["centroid"][b, g, l], ch["nominal"][b, g, l]
                    cur[0][l] = (cen["x"], cen["y"], float(cen["z"]))
                    cur[1][l] = (nom["x"], nom["y"], float(nom["z"]))
            adj += drift
    q["search_radius"], q["n_vertices"] = np.float32(R), 4
    # a degenerate feet polygon (after a committed "no case" centroid result) gives a non-finite centre, which the
    # host-buffer open-loop entry point rejects as an argument error: such units are left out of the comparison
    usable = np.isfinite(q["cx"]) & (np.abs(q["cx"]) <= 1e6) & np.isfinite(q["vx"]).all(-1)
    for f in ("cx", "cy", "vx", "vy"):
        q[f][~usable] = 0.0
    ol = planner.checkFoothold(q.reshape(-1)).reshape(B, n, 4)
    nom = ch["nominal"]
    assert usable.mean() > 0.9
    for f in ("valid", "source", "row", "col", "x", "y"):
        bad = np.nonzero((ol[f] != nom[f]) & usable)
        assert bad[0].size == 0, f"open-loop {f} differs from the chained plan at {tuple(x[0] for x in bad)}"
    assert np.array_equal(ol["z"][usable].view(np.uint32), nom["z"][usable].view(np.uint32)), "z must be the same f32 bits"
    assert (nom["source"] == 1).sum() > 50
    planner.params = _capi.params_yaml()


def test_service_returns_false_where_the_opt_track_gate_fails(planner):
    """getGaitCycleSearchGridMap (cpp:2307-2349) fails in the first gait cycle when the next feet centre lies off the
    map: the reference's handler returns false (cpp:931-934).  Batch plans report it per pose instead."""
    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(300, 300, 0.02, seed=61)  # 6 x 6 m
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    xs = np.array([-3.4, -3.12, -3.08, -2.0, 0.0, 2.8, 2.9, 2.95, 3.2])
    ys = np.array([0.0, 0.0, 0.0, 3.2, 0.0, 0.0, 0.0, -2.99, 0.0])
    poses = make_poses(np.column_stack([xs, ys, np.zeros(xs.size)]))
    out = planner.plan(poses, 3)
    want = omap.pose_status(util.to_oracle_params(planner.params), util.to_oracle_poses(poses))
    assert np.array_equal(out["pose_status"], want)
    assert want.any() and not want.all()
    for k in range(xs.size):
        r = util.service_enforced(planner, 3, poses["position"][k])
        gate = util.oracle_service_gate(omap, planner, poses["position"][k], 3)
        assert (gate == 0) == bool(want[k] & _capi.FPE_POSE_OPT_SUBMAP_FAILED)  # the plan kernels' bit is the cycle-0 gate
        assert (r is False) == (gate != 255)
        if r is not False:
            assert r["gait_cycles"] == 3


def test_service_gate_kinds_default_advisory_enforce(planner):
    """The handler's `return false` (cpp:931-934) by kind (include/fpe.h, fpe_service_gate).  Exact kinds — the first gait
    cycle (stance feet) and the LATERAL side of any cycle (the drift, cpp:1578: independent of the optimiser) — refuse in every
    mode, without the opt track's chain; the x side of a cycle >= 1 follows the build-defined optimiser: ignored by default
    (chain not run), reported under service_opt_gate = 1, refused under 2.  Checked against the oracle's gate functions."""
    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(300, 300, 0.02, seed=61, bad_frac=0.04)  # 6 x 6 m
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    rng = np.random.default_rng(5)
    # poses near the -y edge (the drift carries the gait-cycle submap out: lateral), near the +x edge (the walk reaches the far
    # edge in a later cycle: x side, build-defined), outside (cycle 0), and well inside (no failure)
    pos = [[rng.uniform(-2, 0), -3.0 + rng.uniform(0.003, 0.045), 0.0] for _ in range(6)]
    pos += [[rng.uniform(1.5, 2.7), rng.uniform(-1, 1), 0.0] for _ in range(8)]
    pos += [[3.3, 0.0, 0.0], [-3.4, 0.5, 0.0]]
    pos += [[rng.uniform(-2.5, -1.0), rng.uniform(-1, 1), 0.0] for _ in range(4)]
    n = 8
    kinds = {k: 0 for k in range(4)}
    n_aborted = 0
    harsh = synth.rough_map(400, 400, 0.02, seed=1, bad_frac=0.45)  # terrain on which the opt track derails: x side, build-defined
    hmap = fpo.OracleMap(harsh[0], harsh[1], 0.02)
    rng2 = np.random.default_rng(91)
    hpos = [[rng2.uniform(-3.2, -2.0), rng2.uniform(-3, 3), 0.25] for _ in range(16)]
    for p in pos + hpos:
        if p is hpos[0]:
            planner.gridmapCallback(harsh[0], harsh[1], 0.02)
            omap = hmap
        for mode in (0, 1, 2):
            refuse, kind, cyc = util.oracle_service_verdict(omap, planner, p, n, opt_gate=mode)
            with planner.tuning(service_opt_gate=mode):
                r = planner.globalFootholdPlan(n, p)
                g = planner.last_service_gate()
            assert (r is False) == refuse, (p, mode, kind, cyc, g)
            assert g["returned_false"] == refuse and g["chain_ran"] == (mode != 0)
            assert (g["fail_kind"], g["fail_cycle"]) == (kind, cyc), (p, mode, g, kind, cyc)
            if r is not False:
                assert r["gait_cycles"] == n
            if mode == 1:
                kinds[kind] += 1
            if mode == 2:  # the engine's default IS mode 2
                planner.set_tuning(service_opt_gate=2)
                assert (planner.globalFootholdPlan(n, p) is False) == refuse
            if mode in (0, 1) and r is not False:
                # every track asked for: the chain runs in both modes; stopped at its gate -> OK, opt products empty, kind reported
                with planner.tuning(service_opt_gate=mode):
                    full = planner.globalFootholdPlan(n, p, all_tracks=True)
                    g2 = planner.last_service_gate()
                assert full is not False and g2["chain_ran"]
                aborted = g2["fail_kind"] == _capi.GATE_BUILD_DEFINED
                assert aborted == (util.oracle_service_gate(omap, planner, p, n) != 255)
                if aborted:
                    n_aborted += 1
                    assert len(full["opt"]["footholds"]) == 0 and full["opt"]["report"]["path"].shape[0] == 0
                    assert full["centroid"]["report"]["path"].shape[0] <= n + 1  # the centroid track's own points only
    assert all(v > 0 for v in kinds.values()), kinds  # every kind occurred
    assert n_aborted > 0
    # the default call equals the enforced one wherever the exact kinds decide
    planner.params = _capi.params_yaml()


def _same(a, b, path="result"):
    """Deep equality of two service results (dicts / lists / arrays / scalars), bit for bit."""
    if isinstance(a, dict):
        assert isinstance(b, dict) and a.keys() == b.keys(), path
        for k in a:
            _same(a[k], b[k], f"{path}[{k!r}]")
    elif isinstance(a, (list, tuple)):
        assert len(a) == len(b), path
        for k, (x, y) in enumerate(zip(a, b)):
            _same(x, y, f"{path}[{k}]")
    elif isinstance(a, np.ndarray):
        assert a.dtype == b.dtype and a.shape == b.shape and a.tobytes() == b.tobytes(), path
    else:
        assert a == b or (a != a and b != b), path


def test_service_with_the_opt_chain_beside_the_plan_kernel_equals_one_after_the_other(planner):
    """A one-pose service call runs the opt track's chain on a second stream BESIDE the plan kernel, on cycle flags of 1, and
    again after it when the nominal track's real flags turn out otherwise (fpe_engine.cpp, plan_host; "service_overlap").
    Every product of the call — all tracks, reports, gate — must be what the sequential call returns, bit for bit, on a map where
    nominal searches do fail (cycles that are not committed between cycles that are)."""
    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(400, 400, 0.02, seed=1, bad_frac=0.3)
    planner.gridmapCallback(trav, elev, 0.02)
    omap = fpo.OracleMap(trav, elev, 0.02)
    rng = np.random.default_rng(2718)
    reran = answered = 0
    for _ in range(40):
        pos = [rng.uniform(-3.2, -2.0), rng.uniform(-3, 3), 0.0]
        res = []
        # (overlap, poll): the overlapped call whose host POLLS the chain's completion word in the pinned arena (the default since
        # round 6), the sequential call, the overlapped call that waits for the side stream's signal
        for overlap, poll in ((1, 1), (0, 1), (1, 0), (1, 1)):
            with planner.tuning(service_overlap=overlap, service_poll=poll, service_opt_gate=1):  # (advisory: the call answers, the chain's verdict is reported)
                r = planner.globalFootholdPlan(6, pos, all_tracks=True)
                res.append((r, planner.last_service_gate()))
        _same(res[0], res[1])
        _same(res[2], res[1])
        _same(res[3], res[1])
        if res[0][0] is not False:
            answered += 1
            ok = omap.plan(util.to_oracle_params(planner.params), util.to_oracle_poses(make_poses([pos])), 6)["cycle_ok"][0].astype(bool)
            reran += int(not ok.all())
    assert answered >= 10 and reran >= 5, "need calls whose nominal flags differ from the flags the chain first ran on"


def test_service_right_after_an_asynchronous_upload_waits_on_the_side_stream_too(planner):
    """the overlapped service call launches the opt track's chain on a SECOND stream; after an asynchronous
    fpe_upload_map_device (elevation -> filters -> upload -> service, all on the device) that stream has to wait for the snapshot's
    upload exactly as the plan's stream does, or the chain reads layers the canonicalise / copy kernels are still writing.  The
    upload's stream is kept busy (a spin kernel queued ahead of the upload), every round uploads a map never seen before into
    recycled buffers, and the overlapped call made while the upload is still pending must equal the sequential call made after
    it has completed — every product, the gate included."""
    import torch

    dev = torch.device("cuda", 0)
    planner.params = _capi.params_yaml()
    s_up = torch.cuda.Stream(device=dev)
    rng = np.random.default_rng(31337)
    flat_t, flat_e = np.ones((400, 400), np.float32), np.zeros((400, 400), np.float32)
    answered = differed = 0
    for k in range(8):
        planner.gridmapCallback(flat_t, flat_e, 0.02)  # the snapshot the pending upload replaces: a chain reading IT answers differently
        pos = [rng.uniform(-3.2, -2.0), rng.uniform(-3, 3), 0.0]
        with planner.tuning(service_overlap=0, service_opt_gate=1):
            on_flat = (planner.globalFootholdPlan(6, pos, all_tracks=True), planner.last_service_gate())
        trav, elev = synth.rough_map(400, 400, 0.02, seed=900 + k, bad_frac=0.25)
        with torch.cuda.stream(s_up):
            d_trav = torch.from_numpy(trav).to(dev)
            d_elev = torch.from_numpy(elev).to(dev)
            torch.cuda._sleep(100_000_000)  # tens of milliseconds (at least) of spinning ahead of the upload on its stream
            planner.upload_map_device(d_trav.data_ptr(), d_elev.data_ptr(), 400, 400, 0.02, stream=s_up.cuda_stream)
        assert not s_up.query(), "the upload must still be pending when the service call starts"
        with planner.tuning(service_overlap=1, service_opt_gate=1):
            early = (planner.globalFootholdPlan(6, pos, all_tracks=True), planner.last_service_gate())
        s_up.synchronize()
        with planner.tuning(service_overlap=0, service_opt_gate=1):
            late = (planner.globalFootholdPlan(6, pos, all_tracks=True), planner.last_service_gate())
        _same(early, late)
        answered += int(late[0] is not False)
        try:
            _same(on_flat, late)
        except AssertionError:
            differed += 1
    assert answered >= 3 and differed >= 6, "the rounds must be able to tell the old snapshot from the new one"


def test_service_survives_a_geometry_the_opt_track_does_not_support(planner):
    """on a fine map with a large search radius the opt track is unsupported (its blocked-row mask holds 128 rows);
    the service call — which no longer needs the chain for its return value — still answers, under every gate mode."""
    planner.params = _capi.params_yaml()
    planner.params["searchRadius"] = np.float32(0.34)
    trav, elev = synth.rough_map(700, 500, 0.005, seed=9)
    planner.gridmapCallback(trav, elev, 0.005)
    for mode in (0, 1, 2):
        with planner.tuning(service_opt_gate=mode):
            r = planner.globalFootholdPlan(2, [-0.9, 0.0, 0.0])
            g = planner.last_service_gate()
        assert r is not False and r["gait_cycles"] == 2 and not g["chain_ran"]
    with pytest.raises(FpeError) as e:  # the opt products themselves stay unsupported, and say so
        planner.globalFootholdPlan(2, [-0.9, 0.0, 0.0], all_tracks=True)
    assert e.value.code == _capi.FPE_E_UNSUPPORTED
    planner.params = _capi.params_yaml()


def test_selected_packed_is_the_selected_record_in_eight_bytes(planner):
    """fpe_selected_packed (the halved exchange record): same grid index, flags and f32 height as `selected` / `nominal`, on
    every kernel family (3x3-only, generic 8-lane, one wavefront per pose, direct)."""
    cases = [(0.02, {}, {}), (0.01, {}, {}), (0.01, {"searchRadius": np.float32(0.15)}, {}), (0.02, {}, {"no_bits": 1})]
    for res, prm, tun in cases:
        planner.params = _capi.params_yaml()
        for k, v in prm.items():
            planner.params[k] = v
        trav, elev = synth.rough_map(260, 240, res, seed=33, bad_frac=0.12, nan_frac=0.01)
        planner.gridmapCallback(trav, elev, res)
        poses = synth.poses_in_map(96, 260 * res, 240 * res, 5, 0.18, seed=34, margin=0.2)
        poses["gait"][::3] = 1
        with planner.tuning(**tun):
            out = planner.plan(poses, 5, products=("nominal", "selected", "selected_packed", "cycle_ok"))
        un = _capi.unpack_selected(out["selected_packed"])
        for f in ("row", "col", "valid", "source", "foot_id", "gait_cycle_id"):
            assert np.array_equal(un[f], out["selected"][f]), (res, prm, tun, f)
        assert np.array_equal(un["z"].view(np.uint32), out["selected"]["z"].view(np.uint32))
        assert (out["selected"]["valid"] == 0).any() and (out["selected"]["source"] == 1).any()
        only = planner.plan(poses, 5, products=("selected_packed",))
        assert only["selected_packed"].tobytes() == out["selected_packed"].tobytes()
        again = planner.plan_outputs(poses.shape[0], 5, products=("selected_packed",), pinned=True)  # caller-owned (pinned) array, re-used
        planner.plan(poses, 5, out=again)
        assert again["selected_packed"].tobytes() == out["selected_packed"].tobytes()
    planner.params = _capi.params_yaml()


def test_multi_plan_device_gathers_over_rccl(planner):
    """fpe_multi_plan_device: the C++ host's device-resident multi-GPU plan with the RCCL all-gather of the selected records
    (ncclCommInitAll / ncclGroupStart / ncclAllGather / ncclGroupEnd behind the C ABI, no Python collective).  One device on
    this box: the communicator has one rank (B % 1 == 0: the direct all-gather), the collective still runs through RCCL; both
    record kinds.  The path uneven batches take on a node — blocks padded to ceil(B / n) poses, staged, gathered in place, put in
    place by device-local copies — is forced with the group's "gather_padded" knob and must give the same bytes; what this
    one-GPU box cannot show is that path with n > 1 (its offsets follow fpe_multi_shard_range, covered on the CPU in
    test_cpu_abi_and_host; a test for >= 2 devices is below)."""
    import torch

    from quadrupedal_foothold_planner_amd.planner import MultiFootholdPlanner

    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(300, 300, 0.02, seed=7)
    poses = synth.poses_in_map(257, 6.0, 6.0, 8, 0.18, seed=8, margin=0.7)
    planner.gridmapCallback(trav, elev, 0.02)
    want = planner.plan(poses, 8, products=("nominal", "selected", "selected_packed", "cycle_ok"))
    mp = MultiFootholdPlanner([0])
    try:
        mp.gridmapCallback(trav, elev, 0.02)
        dev = torch.device("cuda:0")
        B, n = poses.shape[0], 8
        d_poses = torch.from_numpy(poses.view(np.uint8).reshape(B, -1).copy()).to(dev)
        for kind, name, dt in ((_capi.EXCHANGE_SELECTED, "selected", _capi.SELECTED_DTYPE), (_capi.EXCHANGE_PACKED, "selected_packed", _capi.PACKED_DTYPE)):
            d_rec = torch.zeros(B * n * 4 * dt.itemsize, dtype=torch.uint8, device=dev)
            d_all = torch.zeros(B * n * 4 * dt.itemsize, dtype=torch.uint8, device=dev)
            d_ok = torch.zeros(B * n, dtype=torch.uint8, device=dev)
            torch.cuda.synchronize()
            mp.plan_device(B, n, [{"d_poses": d_poses.data_ptr(), name: d_rec.data_ptr(), "cycle_ok": d_ok.data_ptr(),
                                   "d_gathered": d_all.data_ptr()}], record_kind=kind)
            mp.synchronize()
            got = d_all.cpu().numpy().view(dt).reshape(B, n, 4)
            assert got.tobytes() == want[name].tobytes(), name
            assert np.array_equal(d_ok.cpu().numpy().reshape(B, n), want["cycle_ok"])
            # the padded path (what B % n != 0 takes), twice: the staging buffer is reused behind its event
            mp.set_tuning(gather_padded=1)
            for _ in range(2):
                d_all.zero_()
                mp.plan_device(B, n, [{"d_poses": d_poses.data_ptr(), name: d_rec.data_ptr(), "cycle_ok": d_ok.data_ptr(),
                                       "d_gathered": d_all.data_ptr()}], record_kind=kind)
                mp.synchronize()
                assert d_all.cpu().numpy().tobytes() == want[name].tobytes(), name + " (padded)"
            mp.set_tuning(gather_padded=0)
    finally:
        mp.close()


def build_collective_shim():
    """tests/probe/collective_shim.cpp -> tests/probe/_build/libcollective_shim.so (hipcc: host code + the HIP runtime)."""
    import os
    import subprocess

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out_dir = os.path.join(root, "tests", "probe", "_build")
    os.makedirs(out_dir, exist_ok=True)
    so = os.path.join(out_dir, "libcollective_shim.so")
    src = os.path.join(root, "tests", "probe", "collective_shim.cpp")
    if not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        r = subprocess.run(["/opt/rocm/bin/hipcc", "-O1", "-Wall", "-shared", "-fPIC", src, "-o", so], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr
    return so


def test_multi_plan_device_with_2_3_and_8_ranks_on_one_gpu():
    """the gather's n > 1 logic had never executed anywhere (RCCL refuses two ranks on one device; the boxes have one).
    The engine binds its collective library at run time, so a device-local stand-in (tests/probe/collective_shim.cpp, bound through
    FPE_RCCL_LIB in a fresh process) lets ONE GPU stand for 2, 3 and 8 ranks: uneven and even batches, both record kinds, the caller's
    streams and the group's, each call twice (staging reuse), a growing and a shrinking batch (staging growth) — every rank's
    d_gathered byte-equal to the single-device plan of the whole batch.  tests/probe/multi_shim_run.py is the script."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPE_RCCL_LIB=build_collective_shim(), PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "probe", "multi_shim_run.py")], capture_output=True, text=True, env=env, cwd=root,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    assert r.stdout.strip().splitlines()[-1] == "ok 36", r.stdout[-2000:]


def test_a_collective_library_that_does_not_load_is_an_error_not_a_fallback():
    """FPE_RCCL_LIB names the library to bind; when it cannot be loaded the gather fails with FPE_E_UNSUPPORTED and says why — it
    never falls back to another library behind the caller's back."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, torch\n"
        "from quadrupedal_foothold_planner_amd import _capi, synth\n"
        "from quadrupedal_foothold_planner_amd.planner import MultiFootholdPlanner, FpeError\n"
        "mp = MultiFootholdPlanner([0])\n"
        "t, e = synth.rough_map(100, 100, 0.02, seed=1)\n"
        "mp.gridmapCallback(t, e, 0.02)\n"
        "p = synth.poses_in_map(4, 2.0, 2.0, 2, 0.18, seed=2, margin=0.7)\n"
        "d = torch.device('cuda:0')\n"
        "dp = torch.from_numpy(p.view(np.uint8).reshape(4, -1).copy()).to(d)\n"
        "rec = torch.zeros(4 * 2 * 4 * 8, dtype=torch.uint8, device=d); al = torch.zeros_like(rec)\n"
        "try:\n"
        "    mp.plan_device(4, 2, [{'d_poses': dp.data_ptr(), 'selected_packed': rec.data_ptr(), 'd_gathered': al.data_ptr()}], record_kind=_capi.EXCHANGE_PACKED)\n"
        "    print('accepted')\n"
        "except FpeError as x:\n"
        "    print('refused', x.code == _capi.FPE_E_UNSUPPORTED, 'FPE_RCCL_LIB' in str(x))\n"
    )
    env = dict(os.environ, FPE_RCCL_LIB="/nonexistent/librccl_typo.so", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=600)
    assert r.returncode == 0 and r.stdout.strip().splitlines()[-1] == "refused True True", r.stdout[-2000:] + r.stderr[-2000:]


@pytest.mark.skipif(__import__("torch").cuda.device_count() < 2, reason="needs two GPUs (the driver's multi-GPU node; this pool's boxes have one)")
def test_multi_plan_device_uneven_batch_on_two_devices(planner):
    """B % n != 0 on a real multi-device group — every device's d_gathered must hold the single-device plan's records
    of the WHOLE batch (padded in-place all-gather + compaction, csrc/fpe_multi.cpp)."""
    import torch

    from quadrupedal_foothold_planner_amd.planner import MultiFootholdPlanner

    ndev = min(torch.cuda.device_count(), 4)
    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(300, 300, 0.02, seed=7)
    B, n = 64 * ndev + 1, 6
    poses = synth.poses_in_map(B, 6.0, 6.0, n, 0.18, seed=8, margin=0.7)
    planner.gridmapCallback(trav, elev, 0.02)
    want = planner.plan(poses, n, products=("selected_packed",))["selected_packed"].tobytes()
    mp = MultiFootholdPlanner(list(range(ndev)))
    try:
        mp.gridmapCallback(trav, elev, 0.02)
        dt, ios, keep = _capi.PACKED_DTYPE, [], []
        raw = poses.view(np.uint8).reshape(B, -1)
        for k in range(ndev):
            first, count = mp.shard_range(B, k)
            dev = torch.device(f"cuda:{k}")
            d_poses = torch.from_numpy(raw[first:first + count].copy()).to(dev)
            d_rec = torch.zeros(count * n * 4 * dt.itemsize, dtype=torch.uint8, device=dev)
            d_all = torch.zeros(B * n * 4 * dt.itemsize, dtype=torch.uint8, device=dev)
            keep.append((d_poses, d_rec, d_all))
            ios.append({"d_poses": d_poses.data_ptr(), "selected_packed": d_rec.data_ptr(), "d_gathered": d_all.data_ptr()})
        for k in range(ndev):
            torch.cuda.synchronize(k)
        mp.plan_device(B, n, ios, record_kind=_capi.EXCHANGE_PACKED)
        mp.synchronize()
        for k in range(ndev):
            assert keep[k][2].cpu().numpy().tobytes() == want, f"device {k}"
    finally:
        mp.close()


def test_foot_radius_much_larger_than_search_radius(planner):
    """The per-leg LDS tile doubles as float scratch of the ordered height sums; a foot disc much larger than the
    search window must not overrun it (the tile grows with the disc's bounding box)."""
    planner.params = _capi.params_yaml()
    planner.params["searchRadius"] = np.float32(0.02)
    planner.params["footRadius"] = np.float32(0.12)
    trav, elev = synth.rough_map(300, 300, 0.01, seed=71, bad_frac=0.002, nan_frac=0.001)
    poses = synth.poses_in_map(64, 3.0, 3.0, 4, 0.18, seed=72, margin=0.6)
    for group in (0, 8, 65):
        with planner.tuning(plan_group=group, no_bits=int(group != 0)):
            try:
                eng, ora = util.run_both(planner, trav, elev, 0.01, poses, 4, threads=8)
            except FpeError as e:
                assert e.code == _capi.FPE_E_UNSUPPORTED
                continue
        util.assert_plan_equal(eng, ora)
    planner.params = _capi.params_yaml()


@pytest.mark.parametrize("n_engines,B", [(1, 37), (2, 37), (3, 5), (4, 3)])
def test_multi_device_group_shards_a_batch_through_the_c_abi(n_engines, B):
    """fpe_multi_*: a C++ host shards a pose batch over a device list without Python or torch.distributed.  A 1-GPU
    box lists cuda:0 several times (independent engines, one host thread each): uneven shards, more engines than poses
    and the global result order are checked against the oracle."""
    from quadrupedal_foothold_planner_amd.planner import MultiFootholdPlanner

    mp = MultiFootholdPlanner([0] * n_engines)
    assert mp.device_count == n_engines
    trav, elev = synth.rough_map(300, 300, 0.02, seed=91, bad_frac=0.15)
    poses = synth.poses_in_map(B, 6.0, 6.0, 5, 0.18, seed=92, margin=0.7)
    poses["gait"][::4] = 1
    mp.gridmapCallback(trav, elev, 0.02)
    eng = mp.plan(poses, 5)
    omap = fpo.OracleMap(trav, elev, 0.02)
    op, opo = util.to_oracle_params(mp.params), util.to_oracle_poses(poses)
    ora = omap.plan(op, opo, 5, threads=4)
    ora["pose_status"] = omap.pose_status(op, opo)
    util.assert_plan_equal(eng, ora)
    mp.close()


@pytest.mark.parametrize("config,batch", [("headline", 512), ("cfg4", 1024)])
def test_bench_two_ranks_on_one_gpu(config, batch):
    """The N > 1 path of bench.py on the 1-GPU box: `python bench.py --gpus 2` starts its two rank processes itself
    (fresh children of a child that never touches the GPU; FPE_BENCH_SHARE_GPU=1 puts both on cuda:0 with gloo
    collectives — RCCL refuses two ranks on one device).  The JSON must name the 2-rank exchange, carry the exchange
    accounting, and every rank's shard must equal the oracle (`verified`)."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPE_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--config", config, "--batch", str(batch),
                        "--steps", "3", "--warmup", "1", "--blocks", "2", "--no-cpu-baseline", "--gather-every", "2"],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads(r.stdout.strip().splitlines()[-1])
    assert line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert "2 ranks in the process group" in line["config"]["exchange"] and "gloo" in line["config"]["exchange"]
    assert "one collective per 2 steps" in line["config"]["exchange"]  # every step's records, batched (3 steps: a trailing partial batch)
    assert line["config"]["verified"] is True
    assert line["config"]["footholds_per_step"] == 2 * batch * line["config"]["n_cycles"] * 4
    assert line["config"]["exchange_bytes_per_rank"] == batch * line["config"]["n_cycles"] * 4 * 8  # the packed 8-byte record
    assert "fpe_selected_packed" in line["config"]["exchange"]
    assert line["config"]["exchange_alt"]["gather_every"] == 1
    assert line["config"]["step_bound_at_this_n"]["by_direct_links"] in ("plan", "exchange")
    assert line["value"] > 0 and "roofline" in line


@pytest.mark.parametrize("record", ["packed", "selected"])
def test_bench_runs_its_exchange_through_rccl_in_a_one_rank_group(record):
    """RCCL for real on the 1-GPU box (VERDICT r3 task 3a): `bench.py --gpus 1` in a fresh process with the nccl backend and
    the collectives forced at world size 1 (FPE_BENCH_FORCE_NCCL=1): init_process_group("nccl"), all_gather_into_tensor on
    the device records the plan kernel wrote, BatchedFootholdExchange's work-handle waits against RCCL's stream, barrier and
    all_reduce — and the gathered records equal the oracle's."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, FPE_BENCH_FORCE_NCCL="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_PORT", "FPE_BENCH_SHARE_GPU"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--config", "headline", "--batch", "1024",
                        "--steps", "7", "--warmup", "2", "--blocks", "1", "--no-cpu-baseline", "--no-extras", "--gather-every", "3", "--exchange-record", record],
                       env=env, cwd=root, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.strip().splitlines() if l.startswith("{")][-1])
    ex = line["config"]["exchange"]
    assert "backend nccl, 1 rank in the process group" in ex and "collective forced" in ex
    assert ("fpe_selected_packed" in ex) == (record == "packed")
    assert "one collective per 3 steps" in ex and " 4 collectives issued" in ex  # warm-up (2 steps: 1 flush) + 7 timed steps: 3
    assert line["config"]["verified"] is True and line["n_gpus"] == 1


@pytest.mark.parametrize("B", [4096, 1000, 37])
def test_host_plan_into_pinned_arrays_equals_the_pageable_path(planner, B):
    """fpe_plan with pinned destinations (fpe_host_alloc, FootholdPlanner.plan_outputs(pinned=True): one block, the products
    behind one another as in the engine's device arena): neighbours without padding in between leave in ONE DMA transfer
    (B = 4096: all seven), products with padding (B = 1000, 37: sizes that are no multiples of 256) or with a product left
    out in between in separate ones; mixed pinned / pageable destinations as well.  Same bytes as the staged path."""
    planner.params = _capi.params_yaml()
    trav, elev = synth.rough_map(300, 300, 0.02, seed=77)
    rng = np.random.default_rng(B)
    poses = make_poses(np.column_stack([rng.uniform(-2.0, 0.5, B), rng.uniform(-2.0, 2.0, B), np.zeros(B)]))
    poses["gait"] = rng.integers(0, 2, B)
    n = 5
    planner.gridmapCallback(trav, elev, 0.02)
    ref = planner.plan(poses, n)
    out = planner.plan(poses, n, out=planner.plan_outputs(B, n, pinned=True))
    for k in ref:
        assert out[k].tobytes() == ref[k].tobytes(), k
    some = ("nominal", "default", "selected", "pose_status")  # centroid, cycle_ok and stance left out: gaps in the arena order
    out2 = planner.plan(poses, n, products=some, out=planner.plan_outputs(B, n, products=some, pinned=True))
    for k in some:
        assert out2[k].tobytes() == ref[k].tobytes(), k
    mixed = planner.plan_outputs(B, n, pinned=True)
    mixed["centroid"] = np.zeros_like(ref["centroid"])  # a pageable array between pinned neighbours
    out3 = planner.plan(poses, n, out=mixed)
    for k in ref:
        assert out3[k].tobytes() == ref[k].tobytes(), k
