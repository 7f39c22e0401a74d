#!/bin/bash
# Variant build of the engine for A/B runs and measurement builds (no GPU needed):
#   bash profiles/build_variant.sh NAME "-DFOO -DBAR=1"   ->  scratch/libfpe_NAME.so   (scratch/ is git-ignored and travels with gpurun)
# e.g. NAME=ht  FLAGS=-DFPE_HOST_TIMING          host-side marks of the service call (probe_host_timing.py)
#      NAME=dbg FLAGS=-DFPE_DBG_COUNT_WALKS      filter cells sent to the literal walks, by reason (fpe_debug_walk_counts)
#      NAME=ch  FLAGS=-DFPE_FILTER_CHAIN_EXPERIMENT   the filter chain as one launch (taken with FPE_FILTER_CHAIN=1)
# Select a variant at run time with FPE_LIB=scratch/libfpe_NAME.so.
set -e
root="$(cd "$(dirname "$0")/.." && pwd)"
mkdir -p "$root/scratch"
cd "$root/quadrupedal_foothold_planner_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -mllvm -amdgpu-kernarg-preload-count=16 \
  $2 -Wno-unused-function -x hip fpe_kernels.hip fpe_engine.cpp fpe_host.cpp fpe_multi.cpp -o "$root/scratch/libfpe_$1.so"
ls -la "$root/scratch/libfpe_$1.so"
