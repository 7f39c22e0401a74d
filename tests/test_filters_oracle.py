"""SURVEY §8(f) N3 — the producer's filter chain (elevation -> traversability): known answers of the oracle's
restatement (oracle/fpo_filters.cpp).  Expected values are written by hand from the published formulas."""
import math

import numpy as np

from oracle import fpo


def _grid(rows, cols, res, f):
    # cell (i, j) centre: x = (rows/2 - 0.5 - i) * res, y = (cols/2 - 0.5 - j) * res (map centred on the origin)
    x = (rows / 2 - 0.5 - np.arange(rows))[:, None] * res
    y = (cols / 2 - 0.5 - np.arange(cols))[None, :] * res
    return f(x + 0 * y, y + 0 * x).astype(np.float32)


def test_filter_defaults_are_the_published_chain():
    fp = fpo.filter_defaults()[0]
    assert (fp["normalRadius"], fp["slopeCritical"], fp["stepCritical"]) == (0.05, 1.0, 0.12)
    assert (fp["stepFirstRadius"], fp["stepSecondRadius"], fp["stepCriticalCells"]) == (0.08, 0.08, 4)
    assert (fp["roughnessCritical"], fp["roughnessRadius"]) == (0.05, 0.05)


def test_flat_ground_is_fully_traversable_with_a_vertical_normal():
    L = fpo.traversability_filters(np.zeros((40, 36), np.float32), 0.02)
    assert np.all(L["normal_z"] == 1.0) and np.all(L["normal_x"] == 0.0) and np.all(L["normal_y"] == 0.0)  # rank < 3: z axis
    assert np.all(L["slope"] == 1.0) and np.all(L["step_height"] == 0.0) and np.all(L["step"] == 1.0) and np.all(L["roughness"] == 1.0)
    np.testing.assert_allclose(L["traversability"], 1.0, rtol=0, atol=6e-8)  # float: (1/3) * 3


def test_tilted_plane_has_the_plane_normal_and_its_slope():
    # z = 0.5 x plus 0.1 mm of deterministic texture: an EXACT plane has a rank-2 covariance and the published filter then
    # answers with the z axis ("no noise in data"); see test_exact_plane_is_rank_deficient
    res, g = 0.02, 0.5
    ii, jj = np.meshgrid(np.arange(60), np.arange(50), indexing="ij")
    elev = _grid(60, 50, res, lambda x, y: g * x) + (1e-4 * np.sin(37.0 * ii + 17.0 * jj)).astype(np.float32)
    # windows of 0.09 m: four cells either way for certain (at exactly 0.08 m the rounding of the cell positions decides)
    L = fpo.traversability_filters(elev, res, params=_params(stepFirstRadius=0.09, stepSecondRadius=0.09))
    inner = (slice(10, 50), slice(10, 40))
    nz = 1.0 / math.sqrt(1.0 + g * g)
    np.testing.assert_allclose(L["normal_z"][inner], nz, atol=2e-3)
    np.testing.assert_allclose(L["normal_x"][inner], -g * nz, atol=2e-3)
    np.testing.assert_allclose(L["normal_y"][inner], 0.0, atol=2e-3)
    np.testing.assert_allclose(L["slope"][inner], 1.0 - math.atan(g) / 1.0, atol=3e-3)
    assert np.all(L["roughness"][inner] > 0.995)  # 0.1 mm against a critical value of 5 cm
    # step height inside the first window (4 cells either way along x): 8 cells * res * g
    np.testing.assert_allclose(L["step_height"][inner], 8 * res * g, atol=3e-4)
    # 0.08 < 0.12 -> no critical cell: step = min(stepMax, 0 / 4 * stepMax) = 0
    assert np.all(L["step"][inner] == 1.0)


def test_exact_plane_is_rank_deficient():
    # z = 0.5 x exactly (float-representable on this grid): the smallest eigenvalue is zero up to rounding, the rank test
    # fails for every cell whose sums cancel exactly and the normal falls back to the z axis there (published behaviour)
    elev = _grid(40, 40, 0.25, lambda x, y: 0.5 * x)  # x multiples of 0.125: exact in float and double
    L = fpo.traversability_filters(elev, 0.25, params=_params(normalRadius=0.6, roughnessRadius=0.6, stepFirstRadius=0.6, stepSecondRadius=0.6))
    inner = (slice(5, 35), slice(5, 35))
    assert np.all(L["normal_z"][inner] == 1.0) and np.all(L["normal_x"][inner] == 0.0)


def _params(**kw):
    fp = fpo.filter_defaults()
    for k, v in kw.items():
        fp[k] = v
    return fp


def test_a_stair_edge_fails_the_step_filter_next_to_the_riser():
    res = 0.02
    elev = _grid(60, 40, res, lambda x, y: np.where(x > 0, 0.15, 0.0))  # riser between rows 29 (x=+0.01) and 30 (x=-0.01)
    L = fpo.traversability_filters(elev, res, params=_params(stepFirstRadius=0.09, stepSecondRadius=0.09))
    # cells within four rows of the riser see both levels: step_height = 0.15 > 0.12
    assert np.allclose(L["step_height"][26:34, 10:30], 0.15)
    assert np.all(L["step_height"][:25, 10:30] == 0.0) and np.all(L["step_height"][35:, 10:30] == 0.0)
    # second window: far more than 4 critical cells around the riser -> step = stepMax = 0.15 >= critical -> 0
    assert np.all(L["step"][28:32, 10:30] == 0.0)
    assert np.all(L["step"][:20, 10:30] == 1.0)
    assert np.all(L["traversability"][29:31, 10:30] < 0.5)


def test_holes_stay_holes_and_their_neighbours_ignore_them():
    res = 0.02
    elev = np.zeros((30, 30), np.float32)
    elev[15, 15] = np.nan
    L = fpo.traversability_filters(elev, res)
    for name in ("normal_x", "normal_z", "slope", "step_height", "roughness", "traversability"):
        assert np.isnan(L[name][15, 15]), name
    assert L["step"][15, 15] == 1.0  # the second pass of the step filter runs on every cell
    mask = np.ones((30, 30), bool)
    mask[15, 15] = False
    assert np.all(L["traversability"][mask] > 0.9999)


def test_single_valid_cell_gets_the_degenerate_answers():
    elev = np.full((9, 9), np.nan, np.float32)
    elev[4, 4] = 0.3
    L = fpo.traversability_filters(elev, 0.02)
    assert L["normal_z"][4, 4] == 1.0 and L["slope"][4, 4] == 1.0 and L["step_height"][4, 4] == 0.0
    assert L["roughness"][4, 4] == 0.0  # sqrt(0 / (1 - 1)) is NaN: `NaN < critical` is false
