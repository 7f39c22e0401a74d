"""Compile the engine (libfpe.so) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the contract: the f64 geometry
must keep the reference's expression order (no fused multiply-add) to reproduce its grid indices.
"""
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libfpe.so")
SOURCES = ["fpe_kernels.hip", "fpe_engine.cpp", "fpe_host.cpp"]
HEADERS = ["fpe_gridmath.hpp", "fpe_device.hpp", "fpe_host.hpp", os.path.join("..", "..", "include", "fpe.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-shared",
    "-ffp-contract=off",
    "-fno-fast-math",
    "-Wall",
    "-Wno-unused-function",
]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    files = [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]
    return any(os.path.getmtime(f) > t for f in files)


def build_engine(force=False, verbose=False):
    """Build quadrupedal_foothold_planner_amd/libfpe.so if missing or stale; returns its path."""
    if not force and not _stale():
        return LIB_PATH
    extra = os.environ.get("FPE_EXTRA_FLAGS", "").split()
    cmd = [HIPCC] + FLAGS + extra + ["-x", "hip"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", LIB_PATH]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB_PATH


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True))
