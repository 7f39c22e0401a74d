#!/usr/bin/env python3
"""bench.py — footholds/sec of the chained foothold plan on MI355X.

A "step" = one pass of the hot path over one batch: fpe_plan_device on B poses x N cycles x 4
legs with the map and the poses already resident in HBM, every product of the three tracks
written (nominal, centroid, default, cycle flags, stance, the 16-byte selected records, pose status);
at N>1 GPUs the step also contains the RCCL all-gather of the selected (nominal) footholds, as
north_star names it — issued asynchronously so that it overlaps the next step's plan kernel
(double-buffered; all gathers complete inside the timed region).  Weak scaling: every rank plans its
own B poses (a contiguous shard of the global seeded list of N*B poses).

Launching: `python bench.py --gpus N` starts its N rank processes itself (fresh children, created
before this process touches the GPU); under `python -m torch.distributed.run --nproc-per-node N bench.py
--gpus N` the ranks torchrun created are used as they are.

Prints ONE JSON line (rank 0).  `roofline.achieved` = algorithmic bytes per foothold (SURVEY.md
§8(d): 508 B at 2 cm / R=0.1) x footholds per launch / mean kernel time (HIP events on the launch
stream) — stated against the 8 TB/s HBM peak although this path is latency/ALU bound and the map
is cache resident.  `cpu_baseline` times the oracle (oracle/, "port") on the host cores
of the same box on the same workload.  The last timed step's products and the open-loop outputs are
compared with the oracle (`config.verified`); a mismatch exits non-zero.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--blocks", type=int, default=25, help="timed blocks of --steps steps; ms_per_step is the median block (min / max reported)")
    ap.add_argument("--config", default="headline", help="headline | cfg2 | cfg3 | cfg4 | cfg5 (quadrupedal_foothold_planner_amd.synth.CONFIGS)")
    ap.add_argument("--batch", type=int, default=None, help="poses per GPU (default: the config's B; cfg4: B/8 = one shard)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the ingest / open-loop / D2H side measurements (profiling runs)")
    ap.add_argument("--b2b-seconds", type=float, default=0.25,
                    help="length of the untimed back-to-back pass (>= 200 launches in any case); profiles/collect.sh passes 0 so that the "
                         "profiler's per-kernel average is taken over the TIMED region's launches, like roofline.kernel_ms")
    ap.add_argument("--no-bits", action="store_true", help="run the direct kernels (fpe_set_tuning no_bits=1) instead of the bit-window kernels")
    ap.add_argument("--cpu-seconds", type=float, default=8.0, help="target seconds of oracle work per baseline leg")
    ap.add_argument("--gather-every", type=int, default=8, help="N>1: one all-gather per K steps, carrying all K steps' records (1: a collective per step; measured as config.exchange_alt)")
    ap.add_argument("--exchange-record", default="packed", choices=("packed", "selected"),
                    help="N>1: the record the all-gather moves: the 8-byte fpe_selected_packed (default) or the 16-byte fpe_selected_foothold")
    return ap.parse_args()


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


def spawn_ranks(n):
    """`python bench.py --gpus N` without a launcher: start N rank processes (fresh interpreters — this process has
    not touched the GPU and never will), wait for them, exit with the worst return code.  Rank 0 prints the JSON."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ)
        env.update(RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), HSA_ENABLE_IPC_MODE_LEGACY="0", FPE_BENCH_SPAWNED="1")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env))
    # poll every rank: the first one that fails takes the others down with it (a rank that died before the rendezvous
    # would otherwise leave its peers waiting in init_process_group / barrier until the collective's timeout)
    rc = 0
    live = list(procs)
    while live and rc == 0:
        time.sleep(0.05)
        for p in list(live):
            code = p.poll()
            if code is None:
                continue
            live.remove(p)
            if code != 0:
                rc = abs(code) or 1
    for p in live:  # only reached with rc != 0: stop the exact children this process started
        p.terminate()
    for p in live:
        try:
            p.wait(timeout=10)
        except subprocess.TimeoutExpired:
            p.kill()
    sys.exit(rc)


def cpu_baseline(trav, elev, res, params, poses, n_cycles, target_s):
    """Time the oracle (CPU restatement, all products like the GPU step) on the same workload:
    one thread, and all host cores (std::thread over poses).  The all-cores leg tiles the pose list so
    that every pass holds >= 256 poses per core and thread start-up does not dominate."""
    from oracle import fpo
    from tests.util import to_oracle_params, to_oracle_poses

    omap = fpo.OracleMap(trav, elev, res)
    op, oposes = to_oracle_params(params), to_oracle_poses(poses)
    cores = usable_cores()
    res_ = {}
    for label, threads in (("single", 1), ("all", cores)):
        reps = 1 if threads == 1 else max(1, min(16, (cores * 256 + len(oposes) - 1) // len(oposes)))
        batch = np.tile(oposes, reps)
        per_pass = batch.shape[0] * n_cycles * 4
        out = omap.plan(op, batch, n_cycles, threads=threads)  # warm + allocate
        t0 = time.perf_counter()
        passes = 0
        while True:
            omap.plan(op, batch, n_cycles, threads=threads, out=out)
            passes += 1
            dt = time.perf_counter() - t0
            if dt >= target_s or passes >= 100000:
                break
        res_[label] = (passes * per_pass / dt, passes, dt, per_pass)
        del out
    # "as-written" EMULATION on cfg-1 (the reference's own case: 200x200 @2cm flat map, 1 pose, 8 cycles):
    # the oracle additionally performs the reference's by-value whole-map copies (hpp:94-143, one per
    # spiral candidate at cpp:2100); two layers only, so it understates a real multi-layer map
    from quadrupedal_foothold_planner_amd import synth as _synth
    t1, e1, r1, p1, n1, _ = _synth.make_config("cfg1")
    om1 = fpo.OracleMap(t1, e1, r1)
    op1 = to_oracle_poses(p1)
    om1.plan_as_written(op, op1, n1)
    t0 = time.perf_counter()
    reps, copies = 0, 0
    while time.perf_counter() - t0 < 1.0:
        _, copies = om1.plan_as_written(op, op1, n1)
        reps += 1
    aw = reps * 4 * n1 * len(op1) / (time.perf_counter() - t0)
    return {
        "value": res_["all"][0],
        "unit": "footholds/s",
        "cores": cores,
        "kind": "port",
        "sample": f"oracle/libfpo.so (copy-free CPU restatement of the reference, g++ -O2), same maps/params, all five "
                  f"products; {cores} std::thread workers (= usable host cores: affinity capped by the cgroup CPU quota) over the step's pose list tiled to {res_['all'][3]} footholds per "
                  f"pass x {res_['all'][1]} passes in {res_['all'][2]:.1f} s",
        "single_thread_value": res_["single"][0],
        "single_thread_sample": f"the step's {res_['single'][3]} footholds x {res_['single'][1]} passes in {res_['single'][2]:.1f} s",
        "as_written_emulation_cfg1": {"value": aw, "unit": "footholds/s", "map_copies_per_call": copies,
                                      "note": "emulation of the reference's by-value GridMap copies on cfg-1 (1 thread, 2 layers); "
                                              "not a measurement of the reference"},
    }


def verify_plan(eng, trav, elev, res, params, poses, n_cycles):
    """The timed launch's own outputs against the oracle on the same poses.  Returns (ok, message)."""
    from oracle import fpo
    from tests import util

    omap = fpo.OracleMap(trav, elev, res)
    op, opo = util.to_oracle_params(params), util.to_oracle_poses(poses)
    ora = omap.plan(op, opo, n_cycles, threads=usable_cores())
    ora["pose_status"] = omap.pose_status(op, opo)
    try:
        util.assert_plan_equal(eng, ora)
    except AssertionError as e:
        return False, str(e)[:400]
    return True, ""


def kernel_sources_sha16():
    """sha256 (first 16 hex digits) over the sources of the plan kernels: what profiles/pmc_traffic.json's counters belong to."""
    import hashlib
    h = hashlib.sha256()
    for f in ("fpe_kernels.hip", "fpe_bits.hpp", "fpe_device.hpp", "fpe_gridmath.hpp"):
        with open(os.path.join(ROOT, "quadrupedal_foothold_planner_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        spawn_ranks(args.gpus)  # never returns

    import torch
    import torch.distributed as dist

    from quadrupedal_foothold_planner_amd import _capi, synth
    from quadrupedal_foothold_planner_amd import dist as fdist
    from quadrupedal_foothold_planner_amd.planner import FootholdPlanner

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if world != args.gpus and rank == 0:
        print(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks; using {world}", file=sys.stderr)
    # debugging aid for 1-GPU boxes: FPE_BENCH_SHARE_GPU=1 puts every rank on cuda:0 and uses gloo for
    # the collectives (RCCL refuses two ranks on one device); never set by the driver
    share = os.environ.get("FPE_BENCH_SHARE_GPU") == "1"
    n_dev = torch.cuda.device_count()  # does not initialise the GPU
    if n_dev == 0:
        raise SystemExit("bench.py needs an MI355X: the engine has no CPU path")
    if world > n_dev and not share:
        raise SystemExit(f"bench.py: {world} ranks but only {n_dev} GPU(s) visible (FPE_BENCH_SHARE_GPU=1 shares cuda:0 for functional checks)")
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    backend = "none"
    # FPE_BENCH_FORCE_NCCL=1 (functional check on a 1-GPU box, tests/test_gpu_device_api.py): a process group of ONE rank on
    # the nccl backend (= RCCL), and the exchange's collectives are issued although nobody else takes part — init, the
    # all-gather on device records, the work-handle waits against RCCL's stream all run for real
    force_nccl = world == 1 and os.environ.get("FPE_BENCH_FORCE_NCCL") == "1"
    if world > 1 or force_nccl:
        backend = "gloo" if share else "nccl"
        import datetime
        rendezvous = datetime.timedelta(seconds=int(os.environ.get("FPE_BENCH_RENDEZVOUS_S", "180")))
        if force_nccl:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            if "MASTER_PORT" not in os.environ:
                with socket.socket() as s_:
                    s_.bind(("127.0.0.1", 0))
                    os.environ["MASTER_PORT"] = str(s_.getsockname()[1])
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            dist.init_process_group("nccl", device_id=dev, timeout=rendezvous, rank=0, world_size=1)
        elif share:
            dist.init_process_group("gloo", timeout=rendezvous)
        else:
            dist.init_process_group("nccl", device_id=dev, timeout=rendezvous)
    exchanging = world > 1 or force_nccl

    # ---- workload --------------------------------------------------------------------------------
    cfg = synth.CONFIGS[args.config]
    B = args.batch or (cfg["B"] // 8 if args.config == "cfg4" else cfg["B"])
    n_cycles = cfg["n_cycles"]
    planner = FootholdPlanner(local_rank)
    if args.no_bits:
        planner.set_tuning(no_bits=1)
    params = planner.params
    trav, elev, res, poses_all, n_cycles, extra = synth.make_config(args.config, B=B * world)
    if "search_radius" in extra:
        params["searchRadius"] = np.float32(extra["search_radius"])
    if "max_leg_search_radius" in extra:
        planner.set_max_leg_search_radius(extra["max_leg_search_radius"])
    rows, cols = trav.shape
    # map: replicated; rank 0's copy is broadcast over RCCL, then canonicalised into the engine
    d_trav, d_elev = fdist.broadcast_map(torch.from_numpy(trav), torch.from_numpy(elev), dev)
    torch.cuda.synchronize()
    planner.upload_map_device(d_trav.data_ptr(), d_elev.data_ptr(), rows, cols, res, stream=torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    lo, hi = fdist.shard_range(B * world, rank, world)
    poses = poses_all[lo:hi]
    d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1)).to(dev)
    n_rec = B * n_cycles * 4
    rec = _capi.FOOTHOLD_DTYPE.itemsize
    sel = _capi.SELECTED_DTYPE.itemsize  # 16 B exchange record: grid index + z + flags
    # what the all-gather moves: the 8-byte packed form by default (half the bytes over xGMI), written by the plan kernel itself
    packed = args.exchange_record == "packed" and rows <= _capi.PACKED_MAX_CELLS and cols <= _capi.PACKED_MAX_CELLS
    xrec = _capi.PACKED_DTYPE.itemsize if packed else sel
    xdtype = _capi.PACKED_DTYPE if packed else _capi.SELECTED_DTYPE
    d_nom = torch.zeros(n_rec * rec, dtype=torch.uint8, device=dev)
    d_cen = torch.zeros(n_rec * _capi.CENTROID_DTYPE.itemsize, dtype=torch.uint8, device=dev)
    d_def = torch.zeros(n_rec * 3, dtype=torch.float64, device=dev)
    d_ok = torch.zeros(B * n_cycles, dtype=torch.uint8, device=dev)
    d_st = torch.zeros(B * 12, dtype=torch.float64, device=dev)
    d_ps = torch.zeros(B, dtype=torch.uint8, device=dev)
    d_sel = torch.zeros(n_rec * sel, dtype=torch.uint8, device=dev)

    stream = torch.cuda.current_stream()
    # N>1: every step's 16-byte selected records are exchanged (RCCL all-gather, its own stream).  The plan kernel of step k
    # writes them straight into sub-block k % K of a staging buffer and every K-th step ONE all-gather moves the whole
    # buffer, overlapping the plans of the next batch (quadrupedal_foothold_planner_amd.dist.BatchedFootholdExchange; two
    # staging buffers).  K = --gather-every (default 8: "fewer, larger collectives" — a collective has a fixed cost of tens
    # of microseconds against a 29 us headline step); K = 1 is measured as well (config.exchange_alt).  Every gathered step's
    # footholds reach every rank; the timed region ends after the last all-gather has completed.
    ev_blocks_ = []  # HIP-event time per step of every timed block of the last run()

    def run(batch):
        ex_ = fdist.BatchedFootholdExchange(n_rec * xrec, dev, batch=batch, force_collective=force_nccl) if exchanging else None
        k_ = [0]

        def step():
            k = k_[0]
            k_[0] += 1
            # (acquire waits, stream-ordered, until the gather that last read this buffer is done)
            x_buf = ex_.acquire(k) if ex_ else None
            planner.plan_device(d_poses.data_ptr(), B, n_cycles, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(),
                                d_ok.data_ptr(), d_st.data_ptr(), stream=stream.cuda_stream,
                                d_selected_ptr=(x_buf.data_ptr() if ex_ and not packed else d_sel.data_ptr()),
                                d_pose_status_ptr=d_ps.data_ptr(), d_selected_packed_ptr=(x_buf.data_ptr() if ex_ and packed else 0))
            if ex_:
                ex_.gather(k)

        # W warmup steps, then BLOCKS of exactly K steps, each between barrier + synchronize pairs; a block's time is the MAX
        # over ranks, the reported step is the MEDIAN block (one block of 20 headline launches is half a
        # millisecond — one mean, no spread, invisible to a sampler; 25 blocks make the timed region >= 10 ms and give the
        # spread, reported as ms_per_step_min / _max)
        for _ in range(args.warmup):
            step()
        if ex_:
            ex_.flush(k_[0] - 1)
            ex_.drain()
            k_[0] = ((k_[0] + batch - 1) // batch) * batch  # the timed region starts on a batch boundary
        walls, evs, last_run = [], [], [k_[0] - 1]
        for _ in range(args.blocks):
            ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if exchanging:
                dist.barrier()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ev0.record(stream)
            for _ in range(args.steps):
                step()
            last_run[0] = k_[0] - 1  # the last step that ran (the block's trailing partial batch included)
            if ex_:
                ex_.flush(k_[0] - 1)
                ex_.drain()  # the stream waits for the in-flight all-gathers
                k_[0] = ((k_[0] + batch - 1) // batch) * batch  # the next block starts on a batch boundary
            ev1.record(stream)
            torch.cuda.synchronize()
            if exchanging:
                dist.barrier()
            el = time.perf_counter() - t0
            t = torch.tensor([el], dtype=torch.float64, device=dev)
            if exchanging:
                dist.all_reduce(t, op=dist.ReduceOp.MAX)
            walls.append(float(t.item()))
            evs.append(ev0.elapsed_time(ev1) / args.steps)
        ev_blocks_[:] = evs
        return walls, float(np.median(evs)), ex_, last_run[0]

    # HIP events on the launch stream bracket the K launches of the timed region; at N=1 the region
    # holds nothing but the K plan kernels, so elapsed/K is the mean launch duration (an upper bound
    # of the kernel time: it includes the ~2 us dispatch gap between back-to-back launches).  At N>1
    # the all-gather shares the stream, so a second, kernel-only pass measures the launch duration.
    gather_batch = max(1, args.gather_every) if exchanging else 1
    walls, kernel_ms, ex, last_step = run(gather_batch)
    ev_blocks = list(ev_blocks_)
    elapsed = float(np.median(walls))  # the median block of K steps
    alt = None
    n_collectives = ex.collectives if ex else 0
    if exchanging:
        g = ex.result(last_step)[rank * n_rec * xrec:(rank + 1) * n_rec * xrec]
        mine = np.frombuffer(g.cpu().numpy().tobytes(), dtype=xdtype).reshape(B, n_cycles, 4)
        if packed:
            mine = _capi.unpack_selected(mine)
        mine = mine.reshape(-1)
        x_last = ex.local_block(last_step).clone()
        if gather_batch > 1 and world > 1:
            walls1, _, ex1, _ = run(1)
            el1 = float(np.median(walls1))
            alt = {"gather_every": 1, "value": 4 * n_cycles * B * world * args.steps / el1, "ms_per_step": el1 / args.steps * 1e3,
                   "note": "one all-gather per step (2 x the collectives' fixed cost per 29 us headline step)"}
            del ex1
        kms = []
        for _ in range(args.blocks):
            k0, k1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            k0.record(stream)
            for _ in range(args.steps):
                planner.plan_device(d_poses.data_ptr(), B, n_cycles, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(),
                                    d_ok.data_ptr(), d_st.data_ptr(), stream=stream.cuda_stream, d_selected_ptr=d_sel.data_ptr(),
                                    d_pose_status_ptr=d_ps.data_ptr())
            k1.record(stream)
            torch.cuda.synchronize()
            kms.append(k0.elapsed_time(k1) / args.steps)
        kernel_ms = float(np.median(kms))

    # The launch duration with the queue kept busy: a block of K launches starts from an idle GPU (barrier + synchronize) and its
    # first dispatch pays the queue's wake-up — ~30 us per block, i.e. 1.3 us per step at K = 20, 0.1 us at K = 200 (measured:
    # 25.0 / 24.6 / 23.7 us per launch at K = 20 / 50 / 200 on one box).  One extra, untimed pass of >= 200 back-to-back launches
    # gives the kernel's own steady-state duration; `frac` keeps using the timed region's figure.
    # Sized to keep the GPU busy for >= --b2b-seconds = 0.25 s (the timed region of a 25 us step is ~13 ms of a 20 s run, below what a
    # utilisation sampler at 1-5 Hz can see; this pass — 10 000 launches at the headline — is the run's visibly busy stretch).
    nb2b = int(min(20000, max(200, args.steps, np.ceil(args.b2b_seconds * 1e3 / max(kernel_ms, 1e-3)))))
    s0, s1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    for rep_ in range(2):
        if rep_ == 1:
            s0.record(stream)
        for _ in range(nb2b if rep_ else 20):
            planner.plan_device(d_poses.data_ptr(), B, n_cycles, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(),
                                d_ok.data_ptr(), d_st.data_ptr(), stream=stream.cuda_stream, d_selected_ptr=d_sel.data_ptr(),
                                d_pose_status_ptr=d_ps.data_ptr())
    s1.record(stream)
    torch.cuda.synchronize()
    kernel_ms_b2b = s0.elapsed_time(s1) / nb2b
    footholds_per_step = 4 * n_cycles * B * world
    value = footholds_per_step * args.steps / elapsed
    R = float(params["searchRadius"][0])
    alg_bytes = _capi.algorithmic_bytes_per_foothold(R, float(params["footRadius"][0]), res)
    achieved = alg_bytes * (4 * n_cycles * B) / (kernel_ms * 1e-3) / 1e9  # GB/s, per launch on one GPU
    peak = 8000.0

    # ---- the timed launch's outputs against the oracle (every rank checks its own shard) -----------------
    torch.cuda.synchronize()
    eng = {
        "nominal": d_nom.cpu().numpy().view(_capi.FOOTHOLD_DTYPE).reshape(B, n_cycles, 4),
        "centroid": d_cen.cpu().numpy().view(_capi.CENTROID_DTYPE).reshape(B, n_cycles, 4),
        "default": d_def.cpu().numpy().reshape(B, n_cycles, 4, 3),
        "cycle_ok": d_ok.cpu().numpy().reshape(B, n_cycles),
        "stance": d_st.cpu().numpy().reshape(B, 4, 3),
        "pose_status": d_ps.cpu().numpy(),
    }
    if exchanging:  # the exchanged record of the last step, as the plan kernel wrote it into the staging buffer
        x_host = x_last.cpu().numpy().view(xdtype).reshape(B, n_cycles, 4)
        eng["selected"] = _capi.unpack_selected(x_host) if packed else x_host
    else:
        eng["selected"] = d_sel.cpu().numpy().view(_capi.SELECTED_DTYPE).reshape(B, n_cycles, 4)
    verified, why = verify_plan(eng, trav, elev, res, params, poses, n_cycles)
    if exchanging:
        nom_host = eng["nominal"].reshape(-1)
        for f in ("row", "col", "valid", "source", "foot_id", "gait_cycle_id"):
            if not np.array_equal(mine[f], nom_host[f]):
                verified, why = False, f"all-gather lost this rank's footholds ({f})"
        if not np.array_equal(mine["z"].view(np.uint32), nom_host["z"].view(np.uint32)):
            verified, why = False, "all-gather lost this rank's footholds (z)"
        flag = torch.tensor([0 if verified else 1], dtype=torch.int32, device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MAX)
        if int(flag.item()) and verified:
            verified, why = False, "another rank's outputs differ from the oracle"
    ok_frac = float(eng["cycle_ok"].mean())
    valid_frac = float(eng["nominal"]["valid"].mean())
    src = eng["nominal"]["source"].reshape(-1)
    codes = np.bincount(eng["centroid"]["code"].reshape(-1), minlength=7)[:7] / src.size

    # roofline.traffic: PMC passes cannot run inside bench.py, so this is the counter-measured figure of the committed
    # profile of this configuration — reported only while the kernel the engine launches NOW is the kernel that profile
    # measured (same template instance), and always with its source
    traffic, traffic_source = None, None
    sources_now = kernel_sources_sha16()
    pmc_path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    kernel_now = planner.describe_plan()
    if os.path.exists(pmc_path) and not args.no_bits:
        try:
            ent = json.load(open(pmc_path)).get(args.config, {})
            inst = kernel_now.split(" (")[0]  # e.g. "plan_bits_kernel<2, true>"
            # (the profiled name carries one more template argument, the product shape: "plan_bits_kernel<2, true, 2>")
            if ent and (inst in ent.get("kernel", "") or inst.rstrip(">") + "," in ent.get("kernel", "")):
                traffic = ent.get("hbm_bytes_per_launch")
                traffic_source = {"file": ent.get("file", "profiles/pmc_traffic.json"), "round": ent.get("round"),
                                  "commit": ent.get("commit", "round-2 HEAD"), "kernel": ent.get("kernel"),
                                  "kernel_sources_sha16": ent.get("kernel_sources_sha16"),
                                  "traffic_stale": ent.get("kernel_sources_sha16") != sources_now,
                                  "note": "static: measured by rocprofv3 --pmc passes when that profile was taken, not by this run; "
                                          "traffic_stale: the plan kernels' sources (sha256 of csrc/fpe_kernels.hip, fpe_bits.hpp, fpe_device.hpp, "
                                          "fpe_gridmath.hpp) have changed since that profile"}
            elif ent:
                traffic_source = {"omitted": f"the committed profile measured `{ent.get('kernel')}`, this run launches `{inst}`"}
        except Exception:
            traffic, traffic_source = None, None

    line = {
        "metric": "footholds/sec (4 legs x N cycles x B poses) on 1k^2 @2cm map; `value` = device-resident (poses and every product stay in HBM), "
                  "`value_8d` / `ms_per_step_8d` = SURVEY 8(d)'s wall time of fpe_plan with host buffers, result D2H included (pinned destinations; "
                  "pageable ones in value_survey_8d)",
        "value": value,
        # SURVEY 8(d)'s own definition of the metric, filled in below at N = 1: wall time of fpe_plan with HOST buffers (poses H2D +
        # kernel + all seven products D2H into pinned destinations from fpe_host_alloc); null where the leg does not run
        "value_8d": None,
        "ms_per_step_8d": None,
        "unit": "footholds/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "ms_per_step_is": f"the median of {args.blocks} timed blocks of {args.steps} steps (each block between barrier + synchronize pairs, MAX over ranks)",
        "ms_per_step_min": min(walls) / args.steps * 1e3,
        "ms_per_step_max": max(walls) / args.steps * 1e3,
        "ms_per_step_p10": float(np.percentile(walls, 10)) / args.steps * 1e3,
        "ms_per_step_p90": float(np.percentile(walls, 90)) / args.steps * 1e3,
        "ms_per_step_blocks": [round(w / args.steps * 1e3, 6) for w in walls],
        "ms_per_step_slowest_block": {"index": int(np.argmax(walls)), "of": len(walls),
                                      "is_first_block": bool(int(np.argmax(walls)) == 0),
                                      "blocks_over_1.5x_median": [int(i) for i in np.nonzero(np.array(walls) > 1.5 * elapsed)[0]],
                                      "note": "host wall clock around each block (barrier + synchronize on both sides); block 0 follows the "
                                              "untimed warm-up directly, every later block follows the previous block's synchronize.  An outlier "
                                              "here that the HIP-event time of the same block (kernel_ms_blocks) does not show is host-side "
                                              "(scheduler, a collector pause), not the GPU"},
        "kernel_ms_blocks": [round(e, 6) for e in ev_blocks],
        "timed_region_ms": sum(walls) * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f64",
        "data": "synthetic",
        "config": {
            "workload": f"{args.config}: {rows}x{cols} @{res*100:g}cm rough terrain (seed {cfg.get('terrain')}), "
                        f"B={B} poses/GPU (seed {cfg.get('pose_seed')}), {'walk' if cfg.get('gait') else 'trot'}, "
                        f"{n_cycles} cycles, searchRadius {R:.3g}, all products written; `value` is device-resident "
                        f"(inputs and results stay in HBM; the PCIe-inclusive rate is value_incl_d2h)",
            "poses_per_gpu": B,
            "n_cycles": n_cycles,
            "footholds_per_step": footholds_per_step,
            "generator": synth.GENERATOR_VERSION,
            "kernels": "direct (no_bits)" if args.no_bits else "bit-window where supported",
            "verified": bool(verified),
            "verified_against": "oracle/ on this step's poses: indices/flags/x/y bit-exact, |dz| <= 1e-6, every product",
            "cycle_ok_fraction": ok_frac,
            "valid_leg_fraction": valid_frac,
            "spiral_leg_fraction": float((src == 1).mean()),
            "default_hit_fraction": float((src == 0).mean()),
            "centroid_code_fractions": [float(c) for c in codes],
            "exchange": (f"all_gather_into_tensor of the selected footholds ({xrec} B records written by the plan kernel: "
                         f"{'fpe_selected_packed — biased row | col << 14 | flags in one word, z' if packed else 'fpe_selected_foothold — grid index, z, flags'}): "
                         f"EVERY step's records, one collective per {gather_batch} steps ({gather_batch} x the bytes), overlapped "
                         f"with the plan kernels of the next batch; backend {backend}, "
                         f"{dist.get_world_size()} rank{'s' if dist.get_world_size() != 1 else ''} in the process group, {n_collectives} collectives issued"
                         + (" (FPE_BENCH_FORCE_NCCL: one rank, the collective forced)" if force_nccl else "")) if exchanging else "none",
        },
        "roofline": {
            "bound": "simd-issue",
            "yardstick": "hbm",
            "achieved": achieved,
            "peak": peak,
            "unit": "GB/s",
            "frac": achieved / peak,
            "traffic": traffic,
            "traffic_source": traffic_source,
            "kernel": kernel_now,
            "kernel_sources_sha16": sources_now,
            "kernel_ms": kernel_ms,
            "kernel_ms_back_to_back": kernel_ms_b2b,
            "frac_back_to_back": alg_bytes * (4 * n_cycles * B) / (kernel_ms_b2b * 1e-3) / 1e9 / peak,
            "back_to_back_note": f"one untimed pass of {nb2b} launches with the queue kept busy: the kernel's steady-state launch duration; "
                                 f"`kernel_ms` / `frac` are the timed region's (blocks of {args.steps} launches, each started from an idle GPU)",
            "algorithmic_bytes_per_foothold": alg_bytes,
            "algorithmic_bytes_convention": ("SURVEY 8(d): 4 W^2 + 8 n_foot + 16 with W from fpe_params.searchRadius — charged to EVERY leg"
                                             + ("; this configuration draws per-leg radii from U[0.06, 0.15] (mean window smaller than the "
                                                "0.1 m one charged) and hexagon polygons: the figure is a convention, not the bytes a leg needs"
                                                if args.config == "cfg5" else "")),
            "frac_by_counter_bytes": (traffic / (kernel_ms * 1e-3) / 1e9 / peak) if traffic else None,
            "note": "`bound` is what the counters say limits the kernel (the SIMDs' instruction issue: VALU-active 0.67-0.94 of the SIMDs' "
                    "time, profiles/); `achieved` / `peak` / `frac` keep north_star's yardstick — SURVEY 8(d)'s ALGORITHMIC bytes against the "
                    "8 TB/s HBM peak (`yardstick`) — so that rounds stay comparable; the counter-measured traffic is `traffic` "
                    "(frac_by_counter_bytes): the map and its bit planes are L2 / Infinity-Cache resident (DESIGN.md §4)",
        },
    }
    if not verified:
        line["config"]["verify_error"] = why
    if world > 1:
        # what the exchange moves, so that a scaling line can be read against the links: xGMI is point to point (every
        # peer's block arrives over its own link; a ring would push all world - 1 blocks through one)
        local_bytes = n_rec * xrec
        link_gbs = 64.0  # GB/s per xGMI link and direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s bidirectional per GPU)
        line["config"]["exchange_bytes_per_rank"] = local_bytes
        line["config"]["exchange_inbound_bytes_per_gpu"] = local_bytes * (world - 1)
        line["config"]["exchange_floor_ms"] = {"direct_links": local_bytes / (link_gbs * 1e9) * 1e3,
                                               "ring": local_bytes * (world - 1) / (link_gbs * 1e9) * 1e3,
                                               "plan_kernel_ms": kernel_ms,
                                               "note": "per-step all-gather time a step cannot go below at 64 GB/s per link and direction; when it "
                                                       "exceeds plan_kernel_ms the step is exchange-bound however well the gather overlaps the next plan"}
        floor = line["config"]["exchange_floor_ms"]
        line["config"]["step_bound_at_this_n"] = {
            "by_direct_links": "exchange" if floor["direct_links"] > kernel_ms else "plan",
            "by_ring": "exchange" if floor["ring"] > kernel_ms else "plan",
            "measured_ms_per_step": elapsed / args.steps * 1e3,
            "note": "which of the plan kernel and the all-gather's link floor is the longer per step (the two overlap: the step costs the "
                    "longer one); measured_ms_per_step above both means neither hides the other completely"}
    if alt:
        line["config"]["exchange_alt"] = alt
    extras = rank == 0 and world == 1 and not args.no_extras
    if extras:
        # the §8(d) metric as defined: wall time of fpe_plan with HOST buffers — poses H2D, kernel, results D2H.
        # Three forms: every product into ordinary (pageable) arrays; every product into pinned arrays (fpe_host_alloc:
        # the device writes them by DMA, no copy-out); the 16-byte selected records only (what north_star's consumer reads)
        host_call_stats = {}

        def timed_host(out, reps_h=20):
            # per call, the MEDIAN: one call in a few dozen takes 2-30 ms on this box (a pinned block of an earlier leg being
            # freed by Python's collector synchronises the device); mean and maximum are kept beside it
            planner.plan(poses, n_cycles, out=out)
            ts = []
            for _ in range(reps_h):
                t0 = time.perf_counter()
                planner.plan(poses, n_cycles, out=out)
                ts.append(time.perf_counter() - t0)
            host_call_stats[id(out)] = {"ms_per_call_mean": float(np.mean(ts)) * 1e3, "ms_per_call_max": float(np.max(ts)) * 1e3, "calls": reps_h}
            return float(np.median(ts))

        out_h = planner.plan_outputs(B, n_cycles)
        dt = timed_host(out_h)
        res_bytes = sum(v.nbytes for v in out_h.values())
        out_p = planner.plan_outputs(B, n_cycles, pinned=True)
        dt_p = timed_host(out_p)
        for k in out_h:  # same launch, same results whichever way they travel
            assert out_h[k].tobytes() == out_p[k].tobytes(), k
        out_s = planner.plan_outputs(B, n_cycles, products=("selected",), pinned=True)
        dt_s = timed_host(out_s)
        assert out_s["selected"].tobytes() == out_h["selected"].tobytes()
        out_k = planner.plan_outputs(B, n_cycles, products=("selected_packed",), pinned=True)
        dt_k = timed_host(out_k)
        sel_k = _capi.unpack_selected(out_k["selected_packed"])
        assert all(np.array_equal(sel_k[f], out_h["selected"][f]) for f in ("row", "col", "valid", "source")) and \
            np.array_equal(sel_k["z"], out_h["selected"]["z"], equal_nan=True)
        # the link itself: one plain device -> pinned-host copy of the same number of bytes (what no D2H path can beat)
        d_raw = torch.empty(res_bytes, dtype=torch.uint8, device=dev)
        h_raw = torch.empty(res_bytes, dtype=torch.uint8, pin_memory=True)
        h_raw.copy_(d_raw, non_blocking=True)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(10):
            h_raw.copy_(d_raw, non_blocking=True)
            torch.cuda.synchronize()
        link_gbs = res_bytes / ((time.perf_counter() - t0) / 10) / 1e9
        del d_raw, h_raw
        line["value_incl_d2h"] = {"value": 4 * n_cycles * B / dt, "unit": "footholds/s", "ms_per_call": dt * 1e3,
                                  "plain_d2h_copy_GB/s_on_this_box": link_gbs,
                                  "result_bytes": res_bytes, "pose_bytes": poses.nbytes,
                                  "GB/s_results": res_bytes / dt / 1e9,
                                  "note": "fpe_plan (host buffers, ordinary pageable arrays): poses H2D + kernel + all seven products D2H in chunks "
                                          "through the engine's pinned arena, copied out by the engine's copy threads while later chunks are in "
                                          "flight; SURVEY 8(d) wall-time definition"}
        # the same number under the name of its definition (`value` is the device-resident rate; SURVEY 8(d) defines the
        # metric on the wall time of fpe_plan including the result D2H — this one)
        line["value_survey_8d"] = {"value": 4 * n_cycles * B / dt, "unit": "footholds/s", "ms_per_call": dt * 1e3,
                                   "pinned_destinations": 4 * n_cycles * B / dt_p,
                                   "definition": "SURVEY 8(d): 4 legs x N cycles x B poses / wall time of fpe_plan (host buffers: poses H2D + kernel + all "
                                                 "seven products D2H; map upload excluded), median of 20 calls; `pinned_destinations`: the same call with "
                                                 "result arrays from fpe_host_alloc.  Bound by the PCIe link (plain_d2h_copy_GB/s_on_this_box in "
                                                 "value_incl_d2h), not by the kernel: `value` is 13-18 x this"}
        line["value_8d"] = 4 * n_cycles * B / dt_p
        line["ms_per_step_8d"] = dt_p * 1e3
        line["value_incl_d2h_pinned"] = {"value": 4 * n_cycles * B / dt_p, "unit": "footholds/s", "ms_per_call": dt_p * 1e3,
                                         "result_bytes": res_bytes, "GB/s_results": res_bytes / dt_p / 1e9,
                                         "note": "same call, result arrays from fpe_host_alloc (pinned): every product is written by DMA straight "
                                                 "into the caller's array"}
        line["value_selected_only"] = {"value": 4 * n_cycles * B / dt_s, "unit": "footholds/s", "ms_per_call": dt_s * 1e3,
                                       "result_bytes": out_s["selected"].nbytes,
                                       "note": "fpe_plan asking for the 16-byte selected records only (pinned destination)"}
        line["value_packed_only"] = {"value": 4 * n_cycles * B / dt_k, "unit": "footholds/s", "ms_per_call": dt_k * 1e3,
                                     "result_bytes": out_k["selected_packed"].nbytes,
                                     "note": "fpe_plan asking for the 8-byte packed selected records only (fpe_selected_packed, pinned "
                                             "destination): what a host that needs the chosen cells and heights pays, PCIe included"}
        for name_, out_ in (("value_incl_d2h", out_h), ("value_incl_d2h_pinned", out_p), ("value_selected_only", out_s), ("value_packed_only", out_k)):
            line[name_].update(host_call_stats[id(out_)])
            line[name_]["ms_per_call_is"] = "the median over the calls"
        # ---- the headline kernel's launch structure (VERDICT r3 task 2): three side measurements, each verified ----
        def ev_ms(fn, reps_e):
            for _ in range(20):  # (the legs before this one are host-side copies: the GPU's clocks have to come back up first)
                fn()
            a_, b_ = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            a_.record(stream)
            for _ in range(reps_e):
                fn()
            b_.record(stream)
            torch.cuda.synchronize()
            return a_.elapsed_time(b_) / reps_e

        # (a) the service's response is the NOMINAL track only (cpp:1588): nominal + selected + cycle flags, no default /
        # centroid products, no stance
        d_nom2, d_sel2, d_ok2 = torch.zeros_like(d_nom), torch.zeros_like(d_sel), torch.zeros_like(d_ok)
        ms_nom = ev_ms(lambda: planner.plan_device(d_poses.data_ptr(), B, n_cycles, d_nominal_ptr=d_nom2.data_ptr(), d_cycle_ok_ptr=d_ok2.data_ptr(),
                                                   stream=stream.cuda_stream, d_selected_ptr=d_sel2.data_ptr()), args.steps)
        nom_ok = bool(torch.equal(d_nom2, d_nom) and torch.equal(d_ok2, d_ok) and torch.equal(d_sel2, d_sel))
        line["value_nominal_only"] = {"value": 4 * n_cycles * B / (ms_nom * 1e-3), "unit": "footholds/s", "ms_per_step": ms_nom, "verified": nom_ok,
                                      "kernel": planner.describe_plan(),
                                      "note": "fpe_plan_device asked for {nominal, selected, cycle_ok} only — the service's response (cpp:1588) and the "
                                              "exchange record; outputs byte-identical to the all-products launch's"}
        # (b) two plans in flight: the timed steps alternate between two streams with their own output buffers (independent
        # plans, what concurrent AsyncSpinner service threads produce): the tail of one launch overlaps the ramp of the next
        s2 = torch.cuda.Stream(device=dev)
        set1 = (d_nom, d_cen, d_def, d_ok, d_st, d_sel, d_ps)
        set2 = tuple(torch.zeros_like(t) for t in set1)

        def launch(st, bs):
            planner.plan_device(d_poses.data_ptr(), B, n_cycles, bs[0].data_ptr(), bs[1].data_ptr(), bs[2].data_ptr(), bs[3].data_ptr(),
                                bs[4].data_ptr(), stream=st.cuda_stream, d_selected_ptr=bs[5].data_ptr(), d_pose_status_ptr=bs[6].data_ptr())

        for _ in range(2):
            launch(stream, set1)
            launch(s2, set2)
        torch.cuda.synchronize()
        n2 = max(2, args.steps - args.steps % 2)
        t0 = time.perf_counter()
        for k in range(n2):
            launch(*((stream, set1) if k % 2 == 0 else (s2, set2)))
        torch.cuda.synchronize()
        dt2 = (time.perf_counter() - t0) / n2
        two_ok = all(torch.equal(a_, b_) for a_, b_ in zip(set1, set2))  # set1 was checked against the oracle above
        line["value_two_in_flight"] = {"value": 4 * n_cycles * B / dt2, "unit": "footholds/s", "ms_per_step": dt2 * 1e3, "steps": n2,
                                       "verified": bool(two_ok and verified),
                                       "note": "the same K launches issued round-robin on two streams with double-buffered outputs (host wall clock "
                                               "around them, synchronised on both sides): independent plans overlap each other's ramp and tail; both "
                                               "output sets byte-identical to the verified single-stream launch"}
        del set2
        # (c) one launch of twice the batch: ramp and tail amortised over two rounds of wavefronts
        if args.config in ("headline", "cfg2"):
            _, _, _, poses2, _, _ = synth.make_config(args.config, B=2 * B)
            d_p2 = torch.from_numpy(poses2.view(np.uint8).reshape(-1)).to(dev)
            big = {k: torch.zeros(2 * t.numel(), dtype=t.dtype, device=dev) for k, t in
                   (("nom", d_nom), ("cen", d_cen), ("def", d_def), ("ok", d_ok), ("st", d_st), ("sel", d_sel), ("ps", d_ps))}
            ms_big = ev_ms(lambda: planner.plan_device(d_p2.data_ptr(), 2 * B, n_cycles, big["nom"].data_ptr(), big["cen"].data_ptr(), big["def"].data_ptr(),
                                                       big["ok"].data_ptr(), big["st"].data_ptr(), stream=stream.cuda_stream,
                                                       d_selected_ptr=big["sel"].data_ptr(), d_pose_status_ptr=big["ps"].data_ptr()), max(4, args.steps // 2))
            eng2 = {"nominal": big["nom"].cpu().numpy().view(_capi.FOOTHOLD_DTYPE).reshape(2 * B, n_cycles, 4),
                    "centroid": big["cen"].cpu().numpy().view(_capi.CENTROID_DTYPE).reshape(2 * B, n_cycles, 4),
                    "default": big["def"].cpu().numpy().reshape(2 * B, n_cycles, 4, 3), "cycle_ok": big["ok"].cpu().numpy().reshape(2 * B, n_cycles),
                    "stance": big["st"].cpu().numpy().reshape(2 * B, 4, 3),
                    "selected": big["sel"].cpu().numpy().view(_capi.SELECTED_DTYPE).reshape(2 * B, n_cycles, 4), "pose_status": big["ps"].cpu().numpy()}
            big_ok, _ = verify_plan(eng2, trav, elev, res, params, poses2, n_cycles)
            line["value_double_batch"] = {"value": 4 * n_cycles * 2 * B / (ms_big * 1e-3), "unit": "footholds/s", "ms_per_launch": ms_big, "poses": 2 * B,
                                          "verified": bool(big_ok),
                                          "note": "ONE launch of 2 x B poses (two rounds of wavefronts per SIMD): the per-pose rate with launch ramp and "
                                                  "tail amortised; every product checked against the oracle"}
            del big, d_p2
        if not (nom_ok and two_ok and line.get("value_double_batch", {}).get("verified", True)):
            verified = False
            line["config"]["verified"] = False
            line["config"]["verify_error"] = "a side measurement (nominal-only / two in flight / double batch) differs from the verified launch"
        # the actual drop-in call: plan_global_footholds for ONE pose x 8 cycles (fpe_plan_service: plan kernel + the opt
        # track's chain for the handler's return value, zero-copy through the pinned arena), wall time per call through
        # ctypes; on a steady map and as the first call after a fresh map message (bit planes pre-built by the upload)
        def service_us(fresh_map, no_bits, reps_s=60, opt_gate=2, overlap=1):
            svc = FootholdPlanner(local_rank)
            svc.params = planner.params.copy()
            if no_bits:
                svc.set_tuning(no_bits=1)
            svc.set_tuning(service_opt_gate=opt_gate, service_overlap=overlap)
            svc.gridmapCallback(trav, elev, res)
            pos = poses["position"][0].copy()
            svc.globalFootholdPlan(8, pos)
            ts = []
            for _ in range(reps_s):
                if fresh_map:
                    svc.gridmapCallback(trav, elev, res)
                t0 = time.perf_counter()
                svc.globalFootholdPlan(8, pos)
                ts.append(time.perf_counter() - t0)
            svc.close()
            return float(np.median(ts) * 1e6)

        def service_entry_us(reps_s=200):
            """The C entry point itself: fpe_plan_service called with prebuilt ctypes arguments (what a C or C++ caller pays, plus one
            foreign-function call of ~1 us) — without the Python mirror's per-call work (response dict, copies)."""
            from quadrupedal_foothold_planner_amd import _capi as _c
            svc = FootholdPlanner(local_rank)
            svc.params = planner.params.copy()
            svc.gridmapCallback(trav, elev, res)
            msg = np.zeros(1, dtype=_c.GLOBAL_FOOTHOLDS_DTYPE)
            pos = np.ascontiguousarray(poses["position"][0], dtype=np.float64).copy()
            a_par, a_pos, a_msg = _c.ptr(svc.params), _c.ptr(pos), _c.ptr(msg)
            fn, h = svc._lib.fpe_plan_service, svc._h
            for _ in range(3):
                assert fn(h, a_par, a_pos, 8, a_msg) in (_c.FPE_OK, _c.FPE_E_SERVICE_FALSE)
            ts = []
            for _ in range(reps_s):
                t0 = time.perf_counter()
                fn(h, a_par, a_pos, 8, a_msg)
                ts.append(time.perf_counter() - t0)
            svc.close()
            return float(np.median(ts) * 1e6)

        line["service_latency_us"] = {
            "steady_map": {"bit_window": service_us(False, False), "direct": service_us(False, True)},
            "steady_map_c_entry_point": {"bit_window": service_entry_us()},
            "first_call_after_a_map": {"bit_window": service_us(True, False, 12), "direct": service_us(True, True, 12)},
            "steady_map_one_kernel_after_the_other": {"bit_window": service_us(False, False, overlap=0)},
            "steady_map_exact_gates_only": {"bit_window": service_us(False, False, opt_gate=0)},
            "first_call_after_a_map_exact_gates_only": {"bit_window": service_us(True, False, 12, opt_gate=0)},
            "note": "median wall time of fpe_plan_service (1 pose x 8 cycles, response assembled) per call through the Python mirror "
                    "(FootholdPlanner.globalFootholdPlan: ctypes call + response dict); steady_map_c_entry_point = the C function alone, called "
                    "with prebuilt arguments.  Default "
                    "(service_opt_gate 2, enforce): the plan kernel and, BESIDE it on a second stream, the opt track's chain (cpp:913-1319: eight "
                    "optimiser searches of 14 641 lattice points each, one after the other) whose gate verdict the call honours like the "
                    "reference's handler (cpp:920-934); the chain runs on nominal cycle flags of 1 and again on the real ones in the call where "
                    "they differ (service_overlap, include/fpe.h; *_one_kernel_after_the_other = service_overlap 0).  *_exact_gates_only = fpe_set_tuning('service_opt_gate', 0): the chain is not run; the "
                    "handler's `return false` is decided for its optimiser-independent kinds only — first gait cycle, lateral side of every "
                    "cycle (include/fpe.h, fpe_service_gate) — for latency-critical callers",
        }
        # map ingest: grid_map message layout (column-major, circular-buffer start
        # index) -> canonical HBM layers, device-resident source; HBM-bound transpose, 2 layers
        ir, ic = 4000, 4000
        src_t = torch.rand(ic * ir, dtype=torch.float32, device=dev)
        src_e = torch.rand(ic * ir, dtype=torch.float32, device=dev)
        ing = FootholdPlanner(local_rank)
        for _ in range(2):
            ing.upload_map_device(src_t.data_ptr(), src_e.data_ptr(), ir, ic, 0.005, start_index=(1234, 321),
                                  storage_order="col", stream=stream.cuda_stream)
        torch.cuda.synchronize()
        i0, i1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        reps = 5
        i0.record(stream)
        for _ in range(reps):
            ing.upload_map_device(src_t.data_ptr(), src_e.data_ptr(), ir, ic, 0.005, start_index=(1234, 321),
                                  storage_order="col", stream=stream.cuda_stream)
        i1.record(stream)
        torch.cuda.synchronize()
        ing_ms = i0.elapsed_time(i1) / reps
        ing_bytes = 2 * 2 * ir * ic * 4  # two layers, read + write
        line["ingest"] = {"map": f"{ir}x{ic} f32 x 2 layers, column-major + start index (1234,321), device source",
                          "ms": ing_ms, "GB/s": ing_bytes / (ing_ms * 1e-3) / 1e9, "bytes": ing_bytes,
                          "note": "canonicalise_layer_kernel x2 per upload, snapshot buffers recycled (no hipMalloc in steady state)"}
        # ... and once the map stream is being PLANNED on (the engine then knows a threshold pair): the same upload also leaves the
        # search bit planes of the new snapshot behind — since round 5 written by the traversability layer's own ingest kernel
        # (a ballot per destination row while the tile is in LDS), not by a second pass over the layer
        try:
            ing.params = planner.params.copy()
            ing.plan(synth.poses_in_map(4, ir * 0.005, ic * 0.005, 2, 0.18, seed=3, margin=0.7), 2)  # registers (thrDefault, thrCandidate)
            for _ in range(2):
                ing.upload_map_device(src_t.data_ptr(), src_e.data_ptr(), ir, ic, 0.005, start_index=(1234, 321), storage_order="col", stream=stream.cuda_stream)
            torch.cuda.synchronize()
            i0.record(stream)
            for _ in range(reps):
                ing.upload_map_device(src_t.data_ptr(), src_e.data_ptr(), ir, ic, 0.005, start_index=(1234, 321), storage_order="col", stream=stream.cuda_stream)
            i1.record(stream)
            torch.cuda.synchronize()
            ingp_ms = i0.elapsed_time(i1) / reps
            line["ingest"]["with_bit_planes"] = {"ms": ingp_ms, "extra_ms_for_the_planes": ingp_ms - ing_ms,
                                                 "note": "the same upload on an engine that has planned: bit planes {D, Df, C, F} of the new "
                                                         "snapshot for the pair in use, built inside canonicalise_layer_kernel (round 4: a pass of "
                                                         "its own, ~13 us for this map)"}
        except Exception as e_:  # (never fails the bench line)
            line["ingest"]["with_bit_planes"] = {"error": str(e_)[:200]}
        ing.close()
        del src_t, src_e
        # open-loop mode: one independent checkFoothold query per (leg, cycle, pose) unit —
        # the step's 4*N*B units as fpe_search_legs_device queries with the reference rectangle polygon
        nq = 4 * n_cycles * B
        rng = np.random.default_rng(7)
        q = np.zeros(nq, dtype=_capi.QUERY_DTYPE)
        half_x, half_y = 0.5 * rows * res - 0.5, 0.5 * cols * res - 0.5
        q["cx"], q["cy"] = rng.uniform(-half_x, half_x, nq), rng.uniform(-half_y, half_y, nq)
        q["search_radius"], q["n_vertices"] = np.float32(R), 4
        Rd = float(np.float32(R))
        q["vx"][:, :4] = q["cx"][:, None] + np.array([Rd, Rd, -Rd, -Rd])
        q["vy"][:, :4] = q["cy"][:, None] + 0.5 * np.array([Rd, -Rd, -Rd, Rd])
        d_q = torch.from_numpy(q.view(np.uint8).reshape(-1)).to(dev)
        d_qo = torch.zeros(nq * rec, dtype=torch.uint8, device=dev)
        for _ in range(2):
            planner.search_legs_device(d_q.data_ptr(), nq, d_qo.data_ptr(), stream=stream.cuda_stream)
        q0, q1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        q0.record(stream)
        for _ in range(10):
            planner.search_legs_device(d_q.data_ptr(), nq, d_qo.data_ptr(), stream=stream.cuda_stream)
        q1.record(stream)
        torch.cuda.synchronize()
        ol_ms = q0.elapsed_time(q1) / 10
        from oracle import fpo
        from tests import util as _util
        ol_eng = d_qo.cpu().numpy().view(_capi.FOOTHOLD_DTYPE)
        ol_ora = fpo.OracleMap(trav, elev, res).search_legs(_util.to_oracle_params(params), _util.to_oracle_queries(q))
        ol_ok = True
        try:
            _util.assert_nominal_equal(ol_eng, ol_ora, "open_loop")
        except AssertionError as e:
            ol_ok, verified = False, False
            line["config"]["verified"] = False
            line["config"]["verify_error"] = str(e)[:400]
        line["open_loop"] = {"queries": nq, "ms": ol_ms, "footholds_per_s": nq / (ol_ms * 1e-3), "verified": ol_ok,
                             "spiral_fraction": float((ol_eng["source"] == 1).mean()),
                             "default_hit_fraction": float((ol_eng["source"] == 0).mean()),
                             "roofline_frac_by_convention": alg_bytes * nq / (ol_ms * 1e-3) / 1e9 / peak,
                             "note": "search_legs_kernel: independent checkFoothold queries at random map positions (no centroid/default "
                                     "track, no chain).  roofline_frac_by_convention charges every query the full 508 B window although a "
                                     "default hit reads ~18 cells; the counter-measured bytes are in profiles/ (DESIGN.md)"}
        # the map's producer: elevation layer of this workload's map -> traversability layer through the
        # device filters, device-resident: two launches (step heights; normals + slope + roughness + second step window +
        # weighted sum).  Timed in both forms: traversability only (no layer buffer: what the pipeline elevation -> filters ->
        # fpe_upload_map_device uses; step_height and traversability are the only layers stored) and with all eight layers.
        d_fe = torch.from_numpy(np.ascontiguousarray(elev)).to(dev)
        d_ft = torch.empty_like(d_fe)
        d_fl = torch.empty(8 * rows * cols, dtype=torch.float32, device=dev)

        def _time_filters(layers_ptr, reps=25, blocks=9):
            for _ in range(3):
                planner.traversability_device(d_fe.data_ptr(), d_ft.data_ptr(), rows, cols, res, d_layers_ptr=layers_ptr, stream=stream.cuda_stream)
            ts = []
            for _ in range(blocks):
                f0, f1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                f0.record(stream)
                for _ in range(reps):
                    planner.traversability_device(d_fe.data_ptr(), d_ft.data_ptr(), rows, cols, res, d_layers_ptr=layers_ptr, stream=stream.cuda_stream)
                f1.record(stream)
                torch.cuda.synchronize()
                ts.append(f0.elapsed_time(f1) / reps)
            return float(np.median(ts))
        f_ms_all = _time_filters(d_fl.data_ptr())
        f_ms = _time_filters(0)
        from oracle import fpo as _fpo
        # checked on a 96 x 96 corner taken as a map of its own, engine and oracle on identical inputs (cell positions of a
        # cut-out differ from the full map's in the last place, and membership at exactly one radius depends on them)
        fr, fc = min(rows, 96), min(cols, 96)
        corner = np.ascontiguousarray(elev[:fr, :fc])
        b = _fpo.traversability_filters(corner, res)["traversability"]
        a = planner.traversability_from_elevation(corner, res)
        f_okc = ~np.isnan(b)
        f_ok = bool(np.array_equal(np.isnan(a), ~f_okc) and np.abs(a[f_okc].view(np.int32).astype(np.int64) - b[f_okc].view(np.int32)).max(initial=0) <= 1)
        if not f_ok:
            verified = False
            line["config"]["verified"] = False
            line["config"]["verify_error"] = "filters: traversability layer differs from the oracle by more than one float ulp"
        n_cells = rows * cols
        line["filters"] = {"map": f"{rows}x{cols} @ {res} m elevation layer of this workload", "ms": f_ms, "cells_per_s": n_cells / (f_ms * 1e-3),
                           "mode": "traversability only (no layer buffer)", "bytes_per_cell_by_layers": 12,
                           "GB/s_by_layers": 12 * n_cells / (f_ms * 1e-3) / 1e9, "frac_of_hbm_peak_by_layers": 12 * n_cells / (f_ms * 1e-3) / 1e9 / peak,
                           "bytes_per_cell_moved": 20,
                           "all_layers": {"ms": f_ms_all, "bytes_per_cell_by_layers": 36, "GB/s_by_layers": 36 * n_cells / (f_ms_all * 1e-3) / 1e9,
                                          "frac_of_hbm_peak_by_layers": 36 * n_cells / (f_ms_all * 1e-3) / 1e9 / peak},
                           "verified": f_ok, "timing": "median of 9 blocks of 25 chains",
                           "note": "fpe_traversability_device, two launches: filter_step_runs_kernel (step heights) and filter_fused_kernel (normals / slope / "
                                   "roughness from row moments of the disc, Newton's iteration for the eigenvector, the step filter's second window by row "
                                   "runs, the weighted sum).  12 B/cell by layers = elevation read + step_height and traversability written; the two "
                                   "launches MOVE 20 B/cell (the second reads elevation and step_height again), measured fabric traffic 1.05 x that "
                                   "(profiles/round5_filters.txt).  Bound by the SIMDs' instruction issue (per 64 cells at 1 cm: 0.36 k + 1.5 k VALU "
                                   "instructions), not by HBM (DESIGN 4.5).  `verified`: engine against oracle on a 96 x 96 corner of the layer taken as a "
                                   "map of its own"}
        del d_fe, d_ft, d_fl
        # the opt track of the same batch (fpe_plan_opt_device; build-defined optimiser: DESIGN 4.6): the nominal
        # plan's cycle flags in, global_footholds_opt + the per-cycle problems out, device-resident
        from quadrupedal_foothold_planner_amd._capi import OPT_CYCLE_DTYPE, OPT_FOOTHOLD_DTYPE
        d_of = torch.zeros(B * n_cycles * 4 * OPT_FOOTHOLD_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_oc = torch.zeros(B * n_cycles * OPT_CYCLE_DTYPE.itemsize, dtype=torch.uint8, device=dev)
        d_og = torch.zeros(B, dtype=torch.uint8, device=dev)
        d_flags = d_ok  # (the timed launch's cycle flags)

        def opt_step():
            planner.plan_opt_device(d_poses.data_ptr(), B, n_cycles, d_flags.data_ptr(), d_of.data_ptr(), d_oc.data_ptr(), d_og.data_ptr(), stream=stream.cuda_stream)
        for _ in range(2):
            opt_step()
        o0, o1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        o0.record(stream)
        for _ in range(5):
            opt_step()
        o1.record(stream)
        torch.cuda.synchronize()
        o_ms = o0.elapsed_time(o1) / 5
        n_chk = min(B, 8)  # engine against oracle on the first poses: every integer of the problems and solutions, x / y bit-exact
        oc_host = d_oc.cpu().numpy().view(OPT_CYCLE_DTYPE).reshape(B, n_cycles)[:n_chk]
        from tests import util as _util
        _om = _fpo.OracleMap(trav, elev, res)
        _op, _opo = _util.to_oracle_params(planner.params), _util.to_oracle_poses(poses[:n_chk])
        _oplan = _om.plan(_op, _opo, n_cycles, threads=4)
        _oopt = _om.plan_opt(_op, _util.to_oracle_opt_params(planner.opt_params), _opo, n_cycles, _oplan["cycle_ok"])
        o_ok = all(np.array_equal(oc_host[f], _oopt["cycles"][f]) for f in ("x", "x_lower", "x_upper", "solver_status", "committed", "minf"))
        if not o_ok:
            verified = False
            line["config"]["verified"] = False
            line["config"]["verify_error"] = "opt track: problems / solutions differ from the oracle"
        line["opt_track"] = {"ms": o_ms, "poses_per_s": B / (o_ms * 1e-3), "solves_per_s": B * n_cycles / (o_ms * 1e-3), "verified": o_ok,
                             "note": "fpe_plan_opt_device on the step's batch: per pose and gait cycle the gait-cycle submap, centroid method, bounds and "
                                     "the build-defined lattice optimiser (the points that attain the smallest constraint violation listed, the objective "
                                     "evaluated on the list: DESIGN 4.6), positions, heights, commit; `verified`: first poses against the oracle"}
        del d_of, d_oc, d_og
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        line["cpu_baseline"] = cpu_baseline(trav, elev, res, params, poses, n_cycles, args.cpu_seconds)
    if rank == 0:
        # RCCL prints its version banner through C stdio, which is block-buffered on a pipe: flush it first so that the JSON
        # is the LAST line on stdout whatever the buffering
        try:
            import ctypes
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        sys.stdout.flush()
        print(json.dumps(line), flush=True)
    planner.close()
    if world > 1:
        dist.destroy_process_group()
    if not verified:
        print(f"bench.py: VERIFICATION FAILED on rank {rank}: {why or line['config'].get('verify_error')}", file=sys.stderr)
        sys.exit(1)


if __name__ == "__main__":
    main()
