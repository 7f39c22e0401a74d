"""Run by tests/test_gpu_device_api.py in a FRESH process with FPE_RCCL_LIB = tests/probe/_build/libcollective_shim.so (the
engine binds its collective library once per process): fpe_multi_plan_device with n > 1 ranks on ONE GPU.

Every rank's d_gathered must hold, byte for byte, the single-device plan's records of the WHOLE batch: that checks the slot
offsets of the padded in-place all-gather, the compaction copies, the staging buffer's reuse / growth and the stream order
between each rank's plan kernel and the exchange (csrc/fpe_multi.cpp, fpe_multi_plan_device) — the engine's side of
SURVEY 8(e), not RCCL's.  Prints one line per case and `ok <cases>` at the end; any mismatch raises."""
import sys

import numpy as np
import torch

from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner, FpeError, MultiFootholdPlanner


def main():
    dev = torch.device("cuda:0")
    trav, elev = synth.rough_map(300, 300, 0.02, seed=7, bad_frac=0.1)
    single = FootholdPlanner(0)
    single.gridmapCallback(trav, elev, 0.02)
    cases = 0
    kinds = ((_capi.EXCHANGE_SELECTED, "selected", _capi.SELECTED_DTYPE), (_capi.EXCHANGE_PACKED, "selected_packed", _capi.PACKED_DTYPE))
    for n_ranks in (2, 3, 8):
        mp = MultiFootholdPlanner([0] * n_ranks)
        try:
            mp.gridmapCallback(trav, elev, 0.02)
            own = [torch.cuda.Stream(device=dev) if k % 2 else None for k in range(n_ranks)]  # odd ranks: the caller's stream; even: the group's
            # batches: uneven small, uneven LARGER (the staging buffer grows), even (direct gather), even forced through the padded
            # path, uneven smaller again (a staging buffer larger than needed is reused), B == n (one pose per rank)
            plan = [(16 * n_ranks + 1, 5, 0), (40 * n_ranks + n_ranks - 1, 6, 0), (24 * n_ranks, 4, 0), (24 * n_ranks, 4, 1), (9 * n_ranks + 2, 3, 0), (n_ranks, 2, 0)]
            for B, n_cyc, forced in plan:
                poses = synth.poses_in_map(B, 6.0, 6.0, n_cyc, 0.18, seed=100 + B, margin=0.7)
                poses["gait"][::5] = 1
                want = single.plan(poses, n_cyc, products=("selected", "selected_packed", "cycle_ok"))
                raw = poses.view(np.uint8).reshape(B, -1)
                mp.set_tuning(gather_padded=forced)
                for kind, name, dt in kinds:
                    ios, keep = [], []
                    for k in range(n_ranks):
                        lo, hi = mp.shard_range(B, k)
                        d_poses = torch.from_numpy(raw[lo:hi].copy()).to(dev)
                        d_rec = torch.zeros((hi - lo) * n_cyc * 4 * dt.itemsize, dtype=torch.uint8, device=dev)
                        d_all = torch.full((B * n_cyc * 4 * dt.itemsize,), 0xA5, dtype=torch.uint8, device=dev)
                        d_ok = torch.zeros((hi - lo) * n_cyc, dtype=torch.uint8, device=dev)
                        keep.append((d_poses, d_rec, d_all, d_ok, lo, hi))
                        ios.append({"d_poses": d_poses.data_ptr(), name: d_rec.data_ptr(), "cycle_ok": d_ok.data_ptr(), "d_gathered": d_all.data_ptr(),
                                    "stream": own[k].cuda_stream if own[k] else 0})
                    torch.cuda.synchronize()
                    for rep in range(2):  # twice: the second call reuses the staging buffers behind their events
                        for _, _, d_all, _, _, _ in keep:
                            d_all.fill_(0xA5)
                        torch.cuda.synchronize()
                        mp.plan_device(B, n_cyc, ios, record_kind=kind)
                        mp.synchronize()
                        for s in own:
                            if s:
                                s.synchronize()
                        for k, (_, d_rec, d_all, d_ok, lo, hi) in enumerate(keep):
                            got = d_all.cpu().numpy().tobytes()
                            assert got == want[name].tobytes(), f"n={n_ranks} B={B} {name} rank {k} rep {rep}: gathered records differ"
                            assert d_rec.cpu().numpy().tobytes() == want[name][lo:hi].tobytes(), f"n={n_ranks} B={B} {name} rank {k}: own block"
                            assert np.array_equal(d_ok.cpu().numpy().reshape(hi - lo, n_cyc), want["cycle_ok"][lo:hi])
                    cases += 1
                    print(f"n_ranks {n_ranks} B {B} cycles {n_cyc} padded {int(B % n_ranks != 0 or forced)} {name}: every rank holds the whole batch", flush=True)
            try:  # fewer poses than ranks stays an argument error
                mp.plan_device(n_ranks - 1, 2, ios, record_kind=_capi.EXCHANGE_PACKED)
                raise AssertionError("B < n accepted")
            except FpeError as e:
                assert e.code == _capi.FPE_E_INVALID_ARG
        finally:
            mp.close()
    single.close()
    print("ok", cases)


if __name__ == "__main__":
    sys.exit(main())
