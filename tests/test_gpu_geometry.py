"""Geometry extraction on the MI355X (vmm_extract_geometry through videometamaterials_amd.extract_geometries): bit-exact against the
golden vectors of the real reference and against the oracle on seeded / adversarial inputs (integer work: no tolerance)."""
import os

import numpy as np
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("name", list(helpers.GEOMETRY_CASES))
@pytest.mark.parametrize("frame", ["lagrangian", "eulerian"])
def test_matches_reference_golden(gpu, name, frame):
    import videometamaterials_amd as vm
    seed, N, T, P, z = helpers.GEOMETRY_CASES[name]
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, f"geometry_{name}.npz"))[frame].astype(np.int32)
    got = vm.extract_geometries(helpers.synth_geometry_videos(seed, N, T, P, z).to(gpu), z, frame)
    assert got.dtype == torch.int32 and tuple(got.shape) == gold.shape
    assert np.array_equal(got.cpu().numpy(), gold)


def _videos_from_masks(masks, T, z):
    """(N, Q, Q) bool geometry images (as clean_pred sees them) -> Lagrangian videos whose extraction yields exactly these masks."""
    N, Q, _ = masks.shape
    P = 2 * Q
    v = torch.zeros(N, 3, T, P, P) + z
    quarter = masks.permute(0, 2, 1).flip(-2)              # undo the transpose and the row mirror of the topology rule
    v[:, 1, T // 2, :Q, :Q] += quarter.float() * 0.05       # material leaves the 0.02 band in one frame
    return v


def test_random_and_adversarial_masks_match_oracle(gpu):
    import videometamaterials_amd as vm
    from oracle import geometry_oracle as go
    g = torch.Generator().manual_seed(77)
    Q = 24
    masks = [torch.rand(Q, Q, generator=g) < p for p in (0.3, 0.5, 0.6, 0.7, 0.45, 0.55, 0.2, 0.9)]
    snake = torch.zeros(Q, Q, dtype=torch.bool)             # one long winding component: worst case for label propagation
    for r in range(0, Q, 2):
        snake[r, :] = True
        snake[r + 1, -1 if (r // 2) % 2 == 0 else 0] = True
    masks.append(snake)
    empty = torch.zeros(Q, Q, dtype=torch.bool)
    empty[3, 3] = empty[10, 20] = True                      # only isolated pixels: the reference raises, we return an empty geometry
    masks.append(empty)
    ties = torch.zeros(Q, Q, dtype=torch.bool)              # four parts of equal size in different orientations
    ties[0, 0:4] = True
    ties[5:9, 2] = True
    ties[12:14, 10:12] = True
    ties[20, 18:22] = True
    masks.append(ties)
    masks = torch.stack(masks)
    for T in (1, 4):
        z = -0.37
        v = _videos_from_masks(masks, T, z)
        if T == 1:                                          # single frame: first-channel / first-frame rule on the bottom-left quarter
            v[:, 0, 0, Q:, :Q] = masks.permute(0, 2, 1).float() * 0.8 + 0.1
        want = go.extract_geometries(v, z, "lagrangian")
        got = vm.extract_geometries(v.to(gpu), z, "lagrangian").cpu().numpy()
        assert np.array_equal(got, want.astype(np.int32))
        if T > 1:
            assert np.array_equal(want[8].reshape(Q, Q), masks[8].numpy().astype(np.int64))  # the snake survives whole
            assert want[9].sum() == 0


def test_band_edges_follow_torch_isclose(gpu):
    """Values a few ulps around zero_u_2 +- (0.02 + 1e-5 |zero_u_2|): same side of the band as torch.isclose puts them."""
    import videometamaterials_amd as vm
    from oracle import geometry_oracle as go
    z = 0.731
    Q, T = 8, 3
    v = torch.full((1, 3, T, 2 * Q, 2 * Q), z)
    edge = torch.tensor(0.02 + 1e-5 * abs(z), dtype=torch.float32)
    vals = []
    for k in range(Q * Q):
        x = torch.tensor(z, dtype=torch.float32) + edge * (1 if k % 2 else -1)
        for _ in range(k // 2 % 5):
            x = torch.nextafter(x, torch.tensor(10.0 if k % 2 else -10.0))
        for _ in range(k // 10 % 5):
            x = torch.nextafter(x, torch.tensor(z))
        vals.append(float(x))
    v[0, 1, 1, :Q, :Q] = torch.tensor(vals).reshape(Q, Q)
    want = go.extract_geometries(v, z, "lagrangian")
    got = vm.extract_geometries(v.to(gpu), z, "lagrangian").cpu().numpy()
    assert np.array_equal(got, want.astype(np.int32))
