#!/bin/bash
# profiles/collect_opt_trace.sh: stage stamps of opt_track_kernel for one pose (a measurement build of the engine with
# -DFPE_OPT_TRACE, compiled on the box into scratch/; the shipped library carries no stamps).
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p scratch gpurun_out
( cd quadrupedal_foothold_planner_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
    -mllvm -amdgpu-kernarg-preload-count=16 -DFPE_OPT_TRACE -Wno-unused-function -x hip fpe_kernels.hip fpe_engine.cpp fpe_host.cpp fpe_multi.cpp \
    -o "$GRAFT_REPO_ROOT/scratch/libfpe_opttrace.so" ) 2>&1 | grep -E "error"
{
  echo "# opt_track_kernel<8>, one pose x 8 gait cycles, headline map and yaml parameters: wall_clock64 stamps (100 MHz) of wavefront 0's"
  echo "# stages and of the first helper's (measurement build, see the script); footholds requested, so heights are part of the last stage."
  FPE_LIB=$GRAFT_REPO_ROOT/scratch/libfpe_opttrace.so python3 profiles/probe_opt_trace.py 2>/dev/null | grep -v "Warn\|amdgpu.ids"
} > gpurun_out/r6p_opt_stage_trace.txt
cat gpurun_out/r6p_opt_stage_trace.txt | head -20
