import os
import sys

import numpy as np
import pytest

try:  # load torch's bundled HIP runtime BEFORE libfpe.so brings in /opt/rocm's: in the other order torch finds no GPU
    import torch  # noqa: F401  (only the tests that own device buffers and streams through torch use it)
except Exception:  # pragma: no cover
    pass

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (gfx950) — run with `pytest -m gpu`")


def yaml_params():
    """foothold_planner.yaml values with the reference's float typing (oracle record layout)."""
    from oracle import fpo

    p = np.zeros(1, dtype=fpo.PARAMS_DTYPE)
    p["footRadius"] = np.float32(0.02)
    p["defaultFootholdThreshold"] = np.float32(0.9)
    p["candidateFootholdThreshold"] = np.float32(0.7)
    p["searchRadius"] = np.float32(0.1)
    p["stepLength"] = np.float32(0.18)
    p["length"] = np.float32(0.4387)
    p["width"] = np.float32(0.175)
    p["l1"] = np.float32(0.037)
    p["skew"] = np.float32(0.04)
    p["RF_FIRST"] = 0
    p["h"] = 0.01
    p["lateralDrift"] = -0.007
    return p


def oracle_poses(xyz, gait=0, leg_radius=None, leg_poly=None):
    from oracle import fpo

    xyz = np.asarray(xyz, dtype=np.float64).reshape(-1, 3)
    p = np.zeros(xyz.shape[0], dtype=fpo.POSE_DTYPE)
    p["pose"] = xyz
    p["gait"] = gait
    if leg_radius is not None:
        p["legRadius"] = leg_radius
    if leg_poly is not None:
        p["legPoly"] = leg_poly
    return p


@pytest.fixture(scope="session")
def params():
    return yaml_params()
