"""Compile the engine (libfpe.so) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU.  -ffp-contract=off is part of the contract: the f64 geometry
must keep the reference's expression order (no fused multiply-add) to reproduce its grid indices.

Concurrency: under torch.distributed every rank imports this module on a fresh checkout.  The build
runs under an exclusive file lock, compiles to a private temporary file and renames it into place,
so a rank can never dlopen a half-written library and only one rank pays for the compile (the
others find a fresh library once they get the lock).  A compile error raises: a stale library is
never loaded in its place.
"""
import fcntl
import os
import subprocess
import tempfile

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.path.join(_HERE, "libfpe.so")
LOCK_PATH = os.path.join(_HERE, ".libfpe.lock")
SOURCES = ["fpe_kernels.hip", "fpe_engine.cpp", "fpe_host.cpp", "fpe_multi.cpp"]
HEADERS = ["fpe_gridmath.hpp", "fpe_device.hpp", "fpe_host.hpp", "fpe_bits.hpp", "fpe_filters.hpp", "fpe_filters_fused.hpp", "fpe_opt.hpp", os.path.join("..", "..", "include", "fpe.h")]
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
    # leading scalar kernel arguments arrive in SGPRs at wave launch (gfx950 kernarg preload) instead of by s_load
    "-mllvm", "-amdgpu-kernarg-preload-count=16",
]

# what the last build_engine() call in this process did: "compiled" | "up-to-date"
LAST_ACTION = None


def _inputs():
    return [os.path.join(CSRC, f) for f in SOURCES + HEADERS] + [os.path.abspath(__file__)]


def _stale():
    if not os.path.exists(LIB_PATH):
        return True
    t = os.path.getmtime(LIB_PATH)
    # a listed input that is missing (renamed / deleted header) is an error of the tree, never "up to date"
    missing = [f for f in _inputs() if not os.path.exists(f)]
    if missing:
        raise FileNotFoundError(f"engine source listed in build.py is missing: {missing}")
    return any(os.path.getmtime(f) > t for f in _inputs())


def build_engine(force=False, verbose=False):
    """Build quadrupedal_foothold_planner_amd/libfpe.so if missing or stale; returns its path.
    Raises subprocess.CalledProcessError when hipcc fails (the old library is left untouched but the
    caller must not load it: _capi.lib() turns the error into EngineUnavailable)."""
    global LAST_ACTION
    if not force and not _stale():
        LAST_ACTION = "up-to-date"
        return LIB_PATH
    with open(LOCK_PATH, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and not _stale():  # another rank built it while this one waited for the lock
                LAST_ACTION = "up-to-date"
                return LIB_PATH
            fd, tmp = tempfile.mkstemp(prefix=".libfpe.", suffix=".so.tmp", dir=_HERE)
            os.close(fd)
            try:
                extra = os.environ.get("FPE_EXTRA_FLAGS", "").split()
                cmd = [HIPCC] + FLAGS + extra + ["-x", "hip"] + [os.path.join(CSRC, f) for f in SOURCES] + ["-o", tmp]
                if verbose:
                    print(" ".join(cmd))
                subprocess.check_call(cmd)
                os.replace(tmp, LIB_PATH)  # atomic: readers see the old or the new file, never a partial one
            finally:
                if os.path.exists(tmp):
                    os.unlink(tmp)
            LAST_ACTION = "compiled"
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB_PATH


if __name__ == "__main__":
    print(build_engine(force=True, verbose=True), LAST_ACTION)
