"""Multi-GPU sharding of the batch axis: one process per GPU, torch.distributed (backend "nccl" is
RCCL on ROCm; "gloo" for CPU tests).

Poses (candidate body trajectories) are independent, so the batch is split into contiguous blocks,
one per rank; the map is replicated (broadcast once per map from rank 0 over xGMI); the only
exchange on the path is one all-gather of the selected footholds per plan.
There is no collective inside the search.
"""
import numpy as np
import torch
import torch.distributed as dist

from ._capi import FOOTHOLD_DTYPE
from ._capi import SELECTED_DTYPE as _SELECTED_DTYPE


def shard_range(total, rank, world):
    """Contiguous block split: the first (total % world) ranks get one extra element."""
    base, rem = divmod(int(total), int(world))
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_sizes(total, world):
    return [shard_range(total, r, world)[1] - shard_range(total, r, world)[0] for r in range(world)]


def broadcast_map(trav, elev, device, src=0):
    """Replicate both layers from `src`.  trav/elev: float32 tensors of identical shape on every
    rank (contents only matter on src).  Returns device tensors."""
    t = trav.to(device).contiguous()
    e = elev.to(device).contiguous()
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(t, src=src)
        dist.broadcast(e, src=src)
    return t, e


def all_gather_records(local, total_poses, pose_bytes):
    """All-gather per-rank result blocks into the global pose order.  `local` is this rank's uint8
    tensor of shard_poses * pose_bytes (pose_bytes = n_cycles * 4 * record size); shards follow
    shard_range(total_poses, rank, world).  Uneven shards are padded to the largest shard for the
    collective and trimmed afterwards, so one all_gather_into_tensor moves everything."""
    if not dist.is_initialized() or dist.get_world_size() == 1:
        return local
    world = dist.get_world_size()
    sizes = shard_sizes(total_poses, world)
    mx = max(sizes) * pose_bytes
    buf = local
    if local.numel() != mx:
        buf = torch.zeros(mx, dtype=torch.uint8, device=local.device)
        buf[: local.numel()] = local
    out = torch.empty(mx * world, dtype=torch.uint8, device=local.device)
    dist.all_gather_into_tensor(out, buf)
    if all(s == sizes[0] for s in sizes):
        return out
    parts = [out[r * mx : r * mx + sizes[r] * pose_bytes] for r in range(world)]
    return torch.cat(parts)


# The exchange record of a selected foothold (16 B per foothold): what north_star asks every
# rank to end up with — the chosen grid index and the height — plus the flag bytes.  It is `fpe_selected_foothold`
# of include/fpe.h, written by the plan kernels themselves as the `selected` product of fpe_plan_out (x and y stay
# with the owning rank): the exchange needs no packing pass.
SELECTED_DTYPE = _SELECTED_DTYPE


class FootholdExchange:
    """Pipelined form of all_gather_records for a stream of plans with equal shards: the all-gather of
    plan k runs on the collective's own stream (async_op) while plan k+1 is computed, so a step costs
    max(plan, exchange) instead of their sum.  `depth` result blocks are cycled; a block is handed out
    again only after the all-gather that read it has completed (stream-ordered wait, no host sync on
    RCCL).  xGMI is point-to-point, so the 4 MB-per-rank exchange of the headline step is of the same
    order as the 60 us plan kernel: hiding it is what keeps weak scaling flat.

        ex = FootholdExchange(local_bytes, device)
        for k in range(steps):
            buf = ex.acquire(k)          # device block the plan of step k writes its selected records into
            plan(..., d_selected_ptr=buf.data_ptr())
            ex.gather(k)                 # asynchronous all-gather of that block
        all_footholds = ex.result(steps - 1)   # [world * local_bytes], waits for that step only
        ex.drain()
    """

    def __init__(self, local_bytes, device, depth=2):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        self.depth = depth
        self.local = [torch.zeros(local_bytes, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.out = [torch.empty(local_bytes * self.world, dtype=torch.uint8, device=device) if self.world > 1 else None
                    for _ in range(depth)]
        self.work = [None] * depth

    def acquire(self, k):
        slot = k % self.depth
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return self.local[slot]

    def gather(self, k):
        slot = k % self.depth
        if self.world > 1:
            self.work[slot] = dist.all_gather_into_tensor(self.out[slot], self.local[slot], async_op=True)

    def result(self, k):
        slot = k % self.depth
        if self.world == 1:
            return self.local[slot]
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None
        return self.out[slot]

    def drain(self):
        for slot in range(self.depth):
            if self.work[slot] is not None:
                self.work[slot].wait()
                self.work[slot] = None


class BatchedFootholdExchange:
    """The exchange of a stream of plans in collectives of `batch` steps: step k writes its selected records into
    sub-block k % batch of a staging buffer, and every `batch`-th step ONE all-gather moves the whole buffer — every
    step's footholds still reach every rank, in 1 / batch as many collectives of batch times the size.  xGMI is point
    to point and a collective has a fixed cost of tens of microseconds: the headline step is a 29 us kernel with 2 MB of
    records per rank, so per-step gathers are launch-bound long before they are link-bound ("fewer, larger
    collectives").  `depth` staging buffers are cycled so that the gather of one batch overlaps the plans of the next; a
    buffer is handed out again only after the all-gather that read it has completed (stream-ordered wait).

        ex = BatchedFootholdExchange(local_bytes, device, batch=8)
        for k in range(steps):
            buf = ex.acquire(k)          # device block the plan of step k writes its selected records into
            plan(..., d_selected_ptr=buf.data_ptr())
            ex.gather(k)                 # launches the batch's all-gather after its last step (asynchronous)
        ex.flush(steps - 1)              # a trailing partial batch
        all_footholds = ex.result(steps - 1)   # [world * local_bytes] of that step, rank order
        ex.drain()
    """

    def __init__(self, local_bytes, device, batch=8, depth=2, force_collective=False):
        self.world = dist.get_world_size() if dist.is_initialized() else 1
        # force_collective: run the all-gather even in a process group of ONE rank (functional check of the RCCL path —
        # init, the collective on device records, the work-handle waits — on a single-GPU box)
        self.collective = self.world > 1 or (bool(force_collective) and dist.is_initialized())
        self.batch = max(1, int(batch))
        self.depth = depth
        self.local_bytes = int(local_bytes)
        n = self.local_bytes * self.batch
        self.stage = [torch.zeros(n, dtype=torch.uint8, device=device) for _ in range(depth)]
        self.out = [torch.empty(n * self.world, dtype=torch.uint8, device=device) if self.collective else None for _ in range(depth)]
        self.work = [None] * depth
        self.launched = [-1] * depth  # last step whose batch was gathered from this buffer
        self.owner = [-1] * depth     # batch (step // batch) the buffer was last handed out for by acquire()
        self.collectives = 0          # all-gathers launched so far

    def _slot(self, k):
        return (k // self.batch) % self.depth

    def _wait(self, slot):
        if self.work[slot] is not None:
            self.work[slot].wait()
            self.work[slot] = None

    def acquire(self, k):
        slot = self._slot(k)
        if k % self.batch == 0:
            self._wait(slot)  # the gather that last read this buffer
        self.owner[slot] = k // self.batch  # from here on later plans overwrite whatever batch the buffer held
        lb = self.local_bytes
        return self.stage[slot][(k % self.batch) * lb:(k % self.batch + 1) * lb]

    def local_block(self, k):
        lb = self.local_bytes
        return self.stage[self._slot(k)][(k % self.batch) * lb:(k % self.batch + 1) * lb]

    def gather(self, k):
        if k % self.batch == self.batch - 1:
            self._launch(k)

    def flush(self, k_last):
        """After the last step: the trailing batch, if it was not complete."""
        if k_last >= 0 and k_last % self.batch != self.batch - 1:
            self._launch(k_last)

    def _launch(self, k):
        slot = self._slot(k)
        self.launched[slot] = k
        if self.collective:
            self.work[slot] = dist.all_gather_into_tensor(self.out[slot], self.stage[slot], async_op=True)
            self.collectives += 1

    def result(self, k):
        """Records of step k from every rank, rank order (waits for that batch's all-gather)."""
        slot = self._slot(k)
        lb = self.local_bytes
        sub = k % self.batch
        # the batch of step k must be the one this buffer was last gathered for: not yet gathered (no flush), or already
        # overwritten by a later batch (depth buffers are cycled), would silently hand out stale or foreign records
        if self.launched[slot] < k or self.launched[slot] // self.batch != k // self.batch:
            raise RuntimeError(f"result({k}): the batch of step {k} is not the one held by its buffer (last gathered step: "
                               f"{self.launched[slot]}; call gather()/flush() first, and read a batch before {self.depth} later ones start)")
        if not self.collective:
            # no collective copied the batch out: the result IS the staging buffer, valid only until acquire() hands the
            # buffer to a later batch (`launched` still names the old one then — ADVICE r4)
            if self.owner[slot] != k // self.batch:
                raise RuntimeError(f"result({k}): its buffer has been handed to batch {self.owner[slot]} since (read a batch before "
                                   f"{self.depth} later ones start)")
            return self.stage[slot][sub * lb:(sub + 1) * lb]
        self._wait(slot)
        n = lb * self.batch
        return torch.cat([self.out[slot][r * n + sub * lb: r * n + (sub + 1) * lb] for r in range(self.world)])

    def drain(self):
        for slot in range(self.depth):
            self._wait(slot)


def records_to_numpy(t, n_cycles, dtype=FOOTHOLD_DTYPE):
    a = t.detach().cpu().numpy().view(dtype)
    return a.reshape(-1, n_cycles, 4)
