#!/bin/bash
# profiles/collect_filters_timeline.sh: per-workgroup clock marks of filter_fused_kernel (a measurement build of the engine with
# -DFPE_FUSED_TIMELINE, compiled on the box into scratch/; the shipped library carries no marks) on the 2 cm and 1 cm maps.
set -u
cd "$GRAFT_REPO_ROOT"
mkdir -p scratch gpurun_out
( cd quadrupedal_foothold_planner_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math \
    -mllvm -amdgpu-kernarg-preload-count=16 -DFPE_FUSED_TIMELINE -Wno-unused-function -x hip fpe_kernels.hip fpe_engine.cpp fpe_host.cpp fpe_multi.cpp \
    -o "$GRAFT_REPO_ROOT/scratch/libfpe_tl.so" ) 2>&1 | grep -E "error" 
export FPE_LIB=$GRAFT_REPO_ROOT/scratch/libfpe_tl.so
{
  echo "# filter_fused_kernel, traversability-only chain: wall_clock64 marks by thread 0 of every workgroup (measurement build, see the script)."
  for m in 0 1; do FPE_PROBE_MAP=$m python3 profiles/probe_filters_timeline.py 2>/dev/null | grep -v "^ xcd\|start times"; done
  echo "# the same on a tilted plane with noise (no cell takes the literal walks):"
  TL_MAP=noise FPE_PROBE_MAP=0 python3 profiles/probe_filters_timeline.py 2>/dev/null | grep "workgroups\|step:\|resident"
} > gpurun_out/r6p_filters_timeline.txt
cat gpurun_out/r6p_filters_timeline.txt | cut -c1-200 | head -8
