"""The oracle is the parity anchor, so it must itself be free of undefined behaviour: build its
sources with -fsanitize=address,undefined (no recovery) and run the stand-alone driver on hostile
maps / poses (oracle/selftest.cpp).  CPU only (GPU sanitizers are not available on the pool)."""
import os
import subprocess

import pytest

ORACLE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle")


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_oracle_is_clean_under_asan_ubsan(seed):
    subprocess.check_call(["make", "-s", "-C", ORACLE, "_build/selftest_asan"])
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=1:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1")
    r = subprocess.run([os.path.join(ORACLE, "_build", "selftest_asan"), str(seed)], env=env, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "selftest ok" in r.stdout
