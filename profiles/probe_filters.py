"""The producer's filter chain (fpe_traversability_device) on larger layers: per-launch time by resolution.
usage on the GPU box: python3 profiles/probe_filters.py   (or under rocprofv3 --kernel-trace --stats for the per-kernel split)"""
import sys
import numpy as np
import torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
s = torch.cuda.current_stream()
import os
MAPS = ((1000, 0.02), (2000, 0.01), (2000, 0.005))
if os.environ.get('FPE_PROBE_MAP'): MAPS = tuple(MAPS[int(k)] for k in os.environ['FPE_PROBE_MAP'].split(','))  # e.g. FPE_PROBE_MAP=1: the 1 cm map alone
for rows, res in MAPS:
    _, elev = synth.rough_map(rows, rows, res, 5)
    d_e = torch.from_numpy(elev).cuda()
    d_t = torch.empty_like(d_e)
    d_l = torch.empty(8 * rows * rows, dtype=torch.float32, device='cuda')
    def timed(layers_ptr, reps=20):
        def run():
            pl.traversability_device(d_e.data_ptr(), d_t.data_ptr(), rows, rows, res, d_layers_ptr=layers_ptr, stream=s.cuda_stream)
        for _ in range(3): run()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(s)
        for _ in range(reps): run()
        e1.record(s); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / reps
    # FPE_PROBE_MODE=trav | layers: ONE chain only (counter passes: every launch of the run then belongs to that chain, so per-kernel
    # averages are per MODE — round 5's traffic file averaged the two chains' launches together)
    mode = os.environ.get('FPE_PROBE_MODE', 'both')
    if mode in ('trav', 'layers'):
        ms1 = timed(0 if mode == 'trav' else d_l.data_ptr())
        print(f"{rows}x{rows} @ {res} m: mode {mode}: {ms1:.3f} ms per chain")
        continue
    ms = timed(d_l.data_ptr())       # every layer stored (the caller passed a layer buffer): 4 B read + 32 B written per cell
    t = d_t.cpu().numpy()
    ms_t = timed(0)                   # traversability only: step_height and traversability stored (12 B per cell by layers)
    t2 = d_t.cpu().numpy()   # (the two chains route cells with a normal component at rounding level differently: a last bit may differ)
    okc = ~np.isnan(t)
    du = np.abs(t[okc].view(np.int32).astype(np.int64) - t2[okc].view(np.int32))
    if not np.array_equal(np.isnan(t), np.isnan(t2)) or du.max(initial=0) > 2: print("MISMATCH between the two modes")
    modes = f"the two chains' layers: {int((du != 0).sum())} of {int(okc.sum())} cells differ, by at most {int(du.max(initial=0))} float ulp"
    print(f"{rows}x{rows} @ {res} m: all layers {ms:.3f} ms per chain, {rows*rows/ms/1e6:.2f} Gcell/s, {36*rows*rows/ms/1e6:.1f} GB/s by layers (36 B/cell); "
          f"traversability only {ms_t:.3f} ms, {rows*rows/ms_t/1e6:.2f} Gcell/s, {12*rows*rows/ms_t/1e6:.1f} GB/s (12 B/cell); "
          f"traversability mean {np.nanmean(t):.3f}, below 0.7: {np.nanmean(t < 0.7):.3f}, holes {np.isnan(t).mean():.4f}; {modes}")
