import os, sys, numpy as np, torch
sys.path.insert(0, '.')
from quadrupedal_foothold_planner_amd import _capi, synth
from quadrupedal_foothold_planner_amd.planner import FootholdPlanner
pl = FootholdPlanner(0)
trav, elev, res, poses, n8, _ = synth.make_config("headline", B=4096)
pl.gridmapCallback(trav, elev, res)
d_poses = torch.from_numpy(poses.view(np.uint8).reshape(-1)).cuda()
s = torch.cuda.current_stream()
for n in (1, 8):
  for B in (2, 64, 512, 1024, 2048, 4096):
    nrec = B*n*4
    d_nom = torch.zeros(nrec*32, dtype=torch.uint8, device='cuda'); d_cen = torch.zeros(nrec*32, dtype=torch.uint8, device='cuda')
    d_def = torch.zeros(nrec*3, dtype=torch.float64, device='cuda'); d_ok = torch.zeros(B*n, dtype=torch.uint8, device='cuda'); d_st = torch.zeros(B*12, dtype=torch.float64, device='cuda')
    d_sel = torch.zeros(nrec*16, dtype=torch.uint8, device='cuda'); d_ps = torch.zeros(B, dtype=torch.uint8, device='cuda')
    def run():
        pl.plan_device(d_poses.data_ptr(), B, n, d_nom.data_ptr(), d_cen.data_ptr(), d_def.data_ptr(), d_ok.data_ptr(), d_st.data_ptr(), stream=s.cuda_stream, d_selected_ptr=d_sel.data_ptr(), d_pose_status_ptr=d_ps.data_ptr())
    # medians of blocks (one block of 50 launches per point gave 35 us outliers at B = 2048 / 4096 — a fresh
    # allocation's first touches and the clocks coming up; 15 blocks of 50 launches, the median block, min and max beside it)
    for _ in range(20): run()
    ts = []
    for _ in range(15):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize(); e0.record(s)
        for _ in range(50): run()
        e1.record(s); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1)/50*1e3)
    print("n_cycles", n, "B", B, "us per launch: median %.2f  min %.2f  max %.2f  (15 blocks of 50 launches)" % (float(np.median(ts)), min(ts), max(ts)))
