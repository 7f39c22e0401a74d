#!/bin/bash
# Profiling build of the engine: -DFPE_TRACE (s_memtime stamps per stage; EXTRA_DEFS="-DFPE_TRACE_ALL_BLOCKS" adds the start /
# end / hardware id of every workgroup) -> scratch/libfpe_trace.so (git-ignored, travels with gpurun).  Run in the container
# before `gpurun -- 'bash profiles/run_round3_profiles.sh'`.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/scratch"
cd "$root/quadrupedal_foothold_planner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 \
  -DFPE_TRACE ${EXTRA_DEFS:-} -Wno-unused-function -x hip fpe_kernels.hip fpe_engine.cpp fpe_host.cpp fpe_multi.cpp -o "$root/scratch/libfpe_trace.so"
ls -la "$root/scratch/libfpe_trace.so"
