"""N>1 path on CPU: world_size-2 gloo processes shard the batch, each plans its shard (the oracle
stands in for the per-rank compute here — the engine needs a GPU), and the all-gather reassembles
the global result byte-identically to the unsharded plan.  Also covers uneven shards and the map
broadcast."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _worker(rank, world, port, B, n_cycles, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import fpo
    from quadrupedal_foothold_planner_amd import dist as fdist
    from quadrupedal_foothold_planner_amd import synth
    from tests.conftest import yaml_params
    from tests.util import to_oracle_poses

    # rank 0 owns the map; everyone else starts from zeros and receives it by broadcast
    trav0, elev0 = synth.rough_map(160, 160, 0.02, seed=3)
    if rank != 0:
        trav0, elev0 = np.zeros_like(trav0), np.zeros_like(elev0)
    t, e = fdist.broadcast_map(torch.from_numpy(trav0), torch.from_numpy(elev0), torch.device("cpu"))
    trav, elev = t.numpy(), e.numpy()
    poses = synth.poses_in_map(B, 3.2, 3.2, n_cycles, 0.18, seed=4, margin=0.65)
    lo, hi = fdist.shard_range(B, rank, world)
    omap = fpo.OracleMap(trav, elev, 0.02)
    out = omap.plan(yaml_params(), to_oracle_poses(poses[lo:hi]), n_cycles)
    local = torch.from_numpy(np.ascontiguousarray(out["nominal"]).view(np.uint8).reshape(-1))
    gathered = fdist.all_gather_records(local, B, n_cycles * 4 * fpo.LEG_DTYPE.itemsize)
    if rank == 0:
        full = omap.plan(yaml_params(), to_oracle_poses(poses), n_cycles)
        ok = gathered.numpy().tobytes() == np.ascontiguousarray(full["nominal"]).tobytes()
        open(os.path.join(tmpdir, "result"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("B", [64, 37])
def test_sharded_plan_all_gather_equals_unsharded(tmp_path, B):
    port = 29500 + (os.getpid() + B) % 2000
    mp.spawn(_worker, args=(2, port, B, 4, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result").read() == "ok"


def _pipelined_worker(rank, world, port, B, n_cycles, steps, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import fpo
    from quadrupedal_foothold_planner_amd import dist as fdist
    from quadrupedal_foothold_planner_amd import synth
    from tests.conftest import yaml_params
    from tests.util import to_oracle_poses

    trav, elev = synth.rough_map(160, 160, 0.02, seed=3)
    omap = fpo.OracleMap(trav, elev, 0.02)
    lo, hi = fdist.shard_range(B, rank, world)
    local_bytes = (hi - lo) * n_cycles * 4 * fdist.SELECTED_DTYPE.itemsize  # the 16 B exchange records
    ex = fdist.FootholdExchange(local_bytes, torch.device("cpu"))

    def selected(nominal):  # the `selected` product of a plan (fpe_selected_foothold), field by field
        o = np.zeros(nominal.size, dtype=fdist.SELECTED_DTYPE)
        flat = nominal.reshape(-1)
        for f in ("row", "col", "z", "valid", "source"):
            o[f] = flat[f]
        return o.tobytes()

    ok = True
    expected = []
    for k in range(steps):  # a different pose list per step: a stale or overwritten block would show
        poses = synth.poses_in_map(B, 3.2, 3.2, n_cycles, 0.18, seed=40 + k, margin=0.65)
        buf = ex.acquire(k)
        out = omap.plan(yaml_params(), to_oracle_poses(poses[lo:hi]), n_cycles)
        buf.copy_(torch.frombuffer(bytearray(selected(out["nominal"])), dtype=torch.uint8))
        ex.gather(k)
        expected.append(selected(omap.plan(yaml_params(), to_oracle_poses(poses), n_cycles)["nominal"]))
        if k >= 1:  # read step k-1 while step k's exchange is in flight
            ok &= ex.result(k - 1).numpy().tobytes() == expected[k - 1]
    ok &= ex.result(steps - 1).numpy().tobytes() == expected[steps - 1]
    ex.drain()
    open(os.path.join(tmpdir, f"result{rank}"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


def _batched_worker(rank, world, port, B, n_cycles, steps, batch, tmpdir):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from oracle import fpo
    from quadrupedal_foothold_planner_amd import dist as fdist
    from quadrupedal_foothold_planner_amd import synth
    from tests.conftest import yaml_params
    from tests.util import to_oracle_poses

    trav, elev = synth.rough_map(160, 160, 0.02, seed=3)
    omap = fpo.OracleMap(trav, elev, 0.02)
    lo, hi = fdist.shard_range(B, rank, world)
    local_bytes = (hi - lo) * n_cycles * 4 * fdist.SELECTED_DTYPE.itemsize
    ex = fdist.BatchedFootholdExchange(local_bytes, torch.device("cpu"), batch=batch)

    def selected(nominal):
        o = np.zeros(nominal.size, dtype=fdist.SELECTED_DTYPE)
        flat = nominal.reshape(-1)
        for f in ("row", "col", "z", "valid", "source"):
            o[f] = flat[f]
        return o.tobytes()

    expected = []
    for k in range(steps):  # a different pose list per step: a stale, dropped or overwritten block would show
        poses = synth.poses_in_map(B, 3.2, 3.2, n_cycles, 0.18, seed=70 + k, margin=0.65)
        buf = ex.acquire(k)
        out = omap.plan(yaml_params(), to_oracle_poses(poses[lo:hi]), n_cycles)
        buf.copy_(torch.frombuffer(bytearray(selected(out["nominal"])), dtype=torch.uint8))
        ex.gather(k)
        expected.append(selected(omap.plan(yaml_params(), to_oracle_poses(poses), n_cycles)["nominal"]))
    ex.flush(steps - 1)
    # EVERY step's records reach every rank (the last `depth` batches are still held by the exchange)
    first_held = max(0, ((steps - 1) // batch - 1) * batch)
    ok = all(ex.result(k).numpy().tobytes() == expected[k] for k in range(first_held, steps))
    if first_held > 0:  # a batch whose buffer has been reused must be refused, not served from foreign records
        try:
            ex.result(0)
            ok = False
        except RuntimeError:
            pass
    ex.drain()
    open(os.path.join(tmpdir, f"result{rank}"), "w").write("ok" if ok else "mismatch")
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("steps,batch", [(7, 3), (8, 4), (5, 8)])
def test_batched_exchange_moves_every_step_in_fewer_collectives(tmp_path, steps, batch):
    port = 29500 + (os.getpid() + 131 * steps + batch) % 2000
    mp.spawn(_batched_worker, args=(2, port, 32, 2, steps, batch, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result0").read() == "ok" and open(tmp_path / "result1").read() == "ok"


def test_pipelined_exchange_overlaps_without_mixing_steps(tmp_path):
    port = 29500 + (os.getpid() + 977) % 2000
    mp.spawn(_pipelined_worker, args=(2, port, 32, 3, 5, str(tmp_path)), nprocs=2, join=True)
    assert open(tmp_path / "result0").read() == "ok" and open(tmp_path / "result1").read() == "ok"


def test_exchange_refuses_steps_it_does_not_hold_and_runs_a_forced_one_rank_collective():
    """BatchedFootholdExchange.result(k) before the batch of step k was gathered raises (it used to return whatever the
    staging buffer held); force_collective runs the all-gather in a process group of ONE rank (the functional check of the
    RCCL path that bench.py makes on a single-GPU box, here over gloo)."""
    from quadrupedal_foothold_planner_amd import dist as fdist

    port = 29500 + (os.getpid() + 555) % 2000
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        ex = fdist.BatchedFootholdExchange(64, torch.device("cpu"), batch=4, force_collective=True)
        assert ex.collective and ex.world == 1
        for k in range(6):
            ex.acquire(k).fill_(k + 1)
            ex.gather(k)
            if k == 1:
                with pytest.raises(RuntimeError):
                    ex.result(1)  # its batch (steps 0-3) has not been gathered yet
        with pytest.raises(RuntimeError):
            ex.result(5)  # trailing partial batch: needs flush()
        ex.flush(5)
        assert ex.collectives == 2
        for k in range(6):
            assert ex.result(k).eq(k + 1).all()
        ex.drain()
        plain = fdist.BatchedFootholdExchange(64, torch.device("cpu"), batch=2)  # one rank, no collective
        assert not plain.collective
        plain.acquire(0).fill_(9)
        with pytest.raises(RuntimeError):
            plain.result(0)
        plain.gather(0)
        plain.acquire(1).fill_(7)
        plain.gather(1)
        assert plain.result(0).eq(9).all() and plain.result(1).eq(7).all() and plain.collectives == 0
        # without a collective the result IS the staging buffer: once acquire() has handed it to a later batch (depth = 2
        # buffers, batches of 2 steps: step 4 reuses the buffer of steps 0-1) the old batch must be refused, not served
        # from a buffer the later plans are writing
        for k in (2, 3):
            plain.acquire(k).fill_(5)
            plain.gather(k)
        plain.acquire(4).fill_(3)  # the buffer of steps 0-1, no gather yet: `launched` still names step 1
        with pytest.raises(RuntimeError):
            plain.result(0)
        with pytest.raises(RuntimeError):
            plain.result(1)
        assert plain.result(2).eq(5).all()
    finally:
        dist.destroy_process_group()


def test_shard_ranges_partition_exactly():
    from quadrupedal_foothold_planner_amd import dist as fdist

    for total in (1, 7, 64, 4096, 262144):
        for world in (1, 2, 3, 4, 8):
            spans = [fdist.shard_range(total, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == total
            assert all(spans[i][1] == spans[i + 1][0] for i in range(world - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1 and sizes == fdist.shard_sizes(total, world)
