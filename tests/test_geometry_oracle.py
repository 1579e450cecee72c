"""The geometry-extraction oracle against the golden vectors made by the real reference (Trainer.save_preds + clean_pred with
networkx, tests/golden/make_golden_geometry.py), plus hand-made cases for the rules networkx's iteration order decides."""
import os

import numpy as np
import pytest

import helpers
from oracle import geometry_oracle as go


@pytest.mark.parametrize("name", list(helpers.GEOMETRY_CASES))
@pytest.mark.parametrize("frame", ["lagrangian", "eulerian"])
def test_oracle_matches_reference_golden(name, frame):
    seed, N, T, P, z = helpers.GEOMETRY_CASES[name]
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, f"geometry_{name}.npz"))[frame].astype(np.int64)
    got = go.extract_geometries(helpers.synth_geometry_videos(seed, N, T, P, z), z, frame)
    assert got.shape == gold.shape == (N, (P // 2) ** 2)
    assert np.array_equal(got, gold)


def _img(rows):
    return np.array([[1.0 if ch == "#" else 0.0 for ch in r] for r in rows], dtype=np.float32)[None]


def test_clean_pred_rules():
    # border pixels without neighbours survive the "individual pixel" pass but are not graph nodes -> dropped with the small parts
    a = _img(["#....", ".....", "..##.", "..#..", "....."])
    assert go.clean_pred(a.copy(), 5).reshape(5, 5).tolist() == [[0, 0, 0, 0, 0], [0, 0, 0, 0, 0], [0, 0, 1, 1, 0], [0, 0, 1, 0, 0], [0, 0, 0, 0, 0]]
    # tie between two parts of three pixels: the one holding the earliest axis-0 edge (row-major) comes first in networkx's order,
    # even though the purely horizontal part starts at an earlier pixel
    b = _img(["###...", "......", "....#.", "...##.", "......", "......"])
    assert go.clean_pred(b.copy(), 6).reshape(6, 6).tolist()[2:4] == [[0, 0, 0, 0, 1, 0], [0, 0, 0, 1, 1, 0]]
    # exactly 0.5 is not material
    c = np.full((1, 4, 4), 0.5, dtype=np.float32)
    c[0, 1, 1:3] = 0.51
    assert go.clean_pred(c.copy(), 4).sum() == 2
