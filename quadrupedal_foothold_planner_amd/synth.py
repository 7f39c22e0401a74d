"""Seeded synthetic inputs for tests and bench.  numpy only.

Map convention (grid_map): (rows, cols) row-major arrays; row i spans x (i grows toward -x),
col j spans y (j grows toward -y); cell (i, j) centre = position + length/2 - res/2 - res*(i, j).
"""
import numpy as np

from ._capi import POSE_DTYPE

GENERATOR_VERSION = "synth-1"


def cell_centres(rows, cols, res, position=(0.0, 0.0)):
    x = position[0] + 0.5 * rows * res - 0.5 * res - res * np.arange(rows, dtype=np.float64)
    y = position[1] + 0.5 * cols * res - 0.5 * res - res * np.arange(cols, dtype=np.float64)
    return x, y


def flat_map(rows, cols):
    """cfg-1: traversability == 1, elevation == 0."""
    return np.ones((rows, cols), np.float32), np.zeros((rows, cols), np.float32)


def rough_map(rows, cols, res, seed, position=(0.0, 0.0), nan_frac=0.005, bad_frac=0.02, stair_period=2.4):
    """cfg-2 recipe: elevation = 3 seeded sinusoids (amp <= 0.05 m) + stair groups (4 rises of
    0.15 m, 0.3 m tread, repeating every `stair_period` m along x); traversability = 1 except
    0.04 m stair-edge strips (0.2), `bad_frac` random cells ~ U[0, 0.6], `nan_frac` cells NaN
    (NaN in both layers, as unknown cells are in a real map)."""
    rng = np.random.default_rng(seed)
    x, y = cell_centres(rows, cols, res, position)
    X, Y = np.meshgrid(x, y, indexing="ij")
    elev = np.zeros((rows, cols), np.float64)
    for _ in range(3):
        amp = rng.uniform(0.01, 0.05)
        kx, ky = rng.uniform(0.5, 3.0, size=2)
        ph = rng.uniform(0, 2 * np.pi)
        elev += amp * np.sin(kx * X + ky * Y + ph)
    # stair groups along x
    u = np.mod(X - (position[0] - 0.5 * rows * res), stair_period)
    step = np.minimum(np.floor(u / 0.3), 4.0)
    elev += 0.15 * step
    trav = np.ones((rows, cols), np.float32)
    # 0.04 m wide strips centred on each rise (u = 0.3, 0.6, 0.9, 1.2) and on the drop (u = 0)
    for edge in (0.3, 0.6, 0.9, 1.2):
        trav[np.abs(u - edge) <= 0.02] = 0.2
    trav[(u <= 0.02) | (u >= stair_period - 0.02)] = 0.2
    n = rows * cols
    bad = rng.choice(n, size=int(round(bad_frac * n)), replace=False)
    trav.reshape(-1)[bad] = rng.uniform(0.0, 0.6, size=bad.size).astype(np.float32)
    nan = rng.choice(n, size=int(round(nan_frac * n)), replace=False)
    elev32 = elev.astype(np.float32)
    trav.reshape(-1)[nan] = np.nan
    elev32.reshape(-1)[nan] = np.nan
    return trav, elev32


def poses_uniform(B, x_range, y_range, seed, z=0.0):
    rng = np.random.default_rng(seed)
    p = np.zeros(B, dtype=POSE_DTYPE)
    p["position"][:, 0] = rng.uniform(x_range[0], x_range[1], size=B)
    p["position"][:, 1] = rng.uniform(y_range[0], y_range[1], size=B)
    p["position"][:, 2] = z
    return p


def poses_in_map(B, side_x, side_y, n_cycles, step, seed, margin=0.6, drift=0.007):
    """Pose rule of SURVEY §8(d): every trajectory stays inside the map."""
    return poses_uniform(
        B,
        (-0.5 * side_x + margin, 0.5 * side_x - margin - n_cycles * step),
        (-0.5 * side_y + margin + drift * n_cycles, 0.5 * side_y - margin),
        seed,
    )


# ---- named configurations (BASELINE.json `configs`) ---------------------------------
CONFIGS = {
    # name: rows, cols, res, terrain seed (None = flat), B, n_cycles, pose seed
    "cfg1": dict(rows=200, cols=200, res=0.02, terrain=None, B=1, n_cycles=8),
    "cfg2": dict(rows=400, cols=400, res=0.02, terrain=1, B=4096, n_cycles=8, pose_seed=1),
    "cfg3": dict(rows=1000, cols=1000, res=0.01, terrain=2, B=4096, n_cycles=32, pose_seed=2, gait=1, search_radius=0.15),
    "cfg4": dict(rows=2000, cols=2000, res=0.01, terrain=3, B=262144, n_cycles=16, pose_seed=3),
    "cfg5": dict(rows=4000, cols=4000, res=0.005, terrain=4, B=4096, n_cycles=8, pose_seed=4, mixed=5),
    "headline": dict(rows=1000, cols=1000, res=0.02, terrain=1, B=4096, n_cycles=8, pose_seed=6),
}


def make_config(name, B=None, n_cycles=None, step=0.18):
    """Return (trav, elev, res, poses, n_cycles, extra) for a named configuration."""
    c = dict(CONFIGS[name])
    if B is not None:
        c["B"] = B
    if n_cycles is not None:
        c["n_cycles"] = n_cycles
    rows, cols, res = c["rows"], c["cols"], c["res"]
    if c["terrain"] is None:
        trav, elev = flat_map(rows, cols)
    else:
        trav, elev = rough_map(rows, cols, res, c["terrain"])
    extra = {}
    if name == "cfg1":
        poses = np.zeros(c["B"], dtype=POSE_DTYPE)
        poses["position"][:] = (-1.0, 0.0, 0.0)
    elif name == "cfg2":
        poses = poses_uniform(c["B"], (-3.2, -2.0), (-3.0, 3.0), c["pose_seed"])
    else:
        poses = poses_in_map(c["B"], rows * res, cols * res, c["n_cycles"], step, c["pose_seed"])
    if c.get("gait"):
        poses["gait"] = c["gait"]
    if c.get("search_radius"):
        extra["search_radius"] = c["search_radius"]
    if c.get("mixed"):
        rng = np.random.default_rng(c["mixed"])
        poses["gait"] = rng.integers(0, 2, size=poses.shape[0])
        poses["leg_search_radius"] = rng.uniform(0.06, 0.15, size=(poses.shape[0], 4)).astype(np.float32)
        poses["leg_polygon_kind"] = rng.integers(0, 2, size=(poses.shape[0], 4))
        extra["max_leg_search_radius"] = 0.15
    return trav, elev, res, poses, c["n_cycles"], extra
