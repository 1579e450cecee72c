"""CPU restatement of the reference's geometry extraction -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module; the product path
(videometamaterials_amd/geometry.py -> vmm_extract_geometry) never does.

Follows, by behaviour (no code shared):
  * the topology rule of Trainer.save_preds, vddp.py:1890-1913  (quarter of the frame, "void iff u_2 stays within 0.02 of its
    zero value in every frame", transpose for Abaqus)
  * clean_pred, src/utils.py:32-82  (binarise, drop pixels whose four existing neighbours are all empty, keep the largest
    4-connected component; networkx's iteration order decides ties)

Pinned by tests/golden/geometry_*.npz, which tests/golden/make_golden_geometry.py produced by running the REAL
Trainer.save_preds / clean_pred (with networkx) in the build container.
"""
from __future__ import annotations

import numpy as np
import torch


def topologies(videos: torch.Tensor, zero_u_2: float, reference_frame: str = "lagrangian") -> np.ndarray:
    """(N, C, T, P, P) sampled videos -> (N, P/2, P/2) float topology, already transposed (vddp.py:1892-1913)."""
    n, _, t, p, _ = videos.shape
    h = p // 2
    if reference_frame == "eulerian" or (reference_frame == "lagrangian" and t == 1):
        topo = videos[:, 0, 0, h:, :h].clone()                      # bottom-left quarter, first channel, first frame (vddp.py:1895-1897)
    elif reference_frame == "lagrangian":
        red = videos[:, :, :, :h, :h].flip(-2)                      # upper-left quarter, mirrored along the pixel rows (vddp.py:1900-1902)
        z = torch.as_tensor([zero_u_2], dtype=videos.dtype)
        close = torch.isclose(red[:, 1], z, atol=0.02)              # channel 1 = u_2, every frame (vddp.py:1907)
        topo = torch.logical_not(close.all(dim=1)).to(videos.dtype)  # material where NOT close to zero in all frames (vddp.py:1909-1911)
    else:
        raise ValueError(reference_frame)
    return topo.permute(0, 2, 1).contiguous().numpy()               # "transpose topologies to ensure consistency with Abaqus" (vddp.py:1913)


def _component_order_key(pixels, p):
    """Position of a component in networkx's connected_components iteration = insertion time of its first node.
    create_graph (src/utils.py:12-30) inserts all axis-0 edges (x,y)-(x+1,y) in row-major order of their start, then all axis-1
    edges (x,y)-(x,y+1); a component is therefore ordered by its earliest axis-0 edge, or -- if it has none -- after all of those,
    by its earliest axis-1 edge."""
    s = set(pixels)
    h = [x * p + y for (x, y) in pixels if (x + 1, y) in s]
    if h:
        return min(h)
    v = [x * p + y for (x, y) in pixels if (x, y + 1) in s]
    return p * p + min(v)


def clean_pred(geom_raw: np.ndarray, pixels: int) -> np.ndarray:
    """(N, pixels, pixels) float -> (N, pixels**2) int (src/utils.py:32-82)."""
    g = np.asarray(geom_raw, dtype=np.float64).reshape(-1, pixels, pixels)
    out = np.zeros((g.shape[0], pixels, pixels), dtype=np.int64)
    for i in range(g.shape[0]):
        img = (g[i] > 0.5).astype(np.int64)  # < 0.5 -> 0, > 0.5 -> 1, exactly 0.5 truncates to 0 (src/utils.py:34-37)
        # "remove individual pixels" (src/utils.py:46-62): a pixel goes when all four neighbours are empty, a neighbour beyond the
        # border counting as present.  The reference scans in place in raster order; an occupied neighbour of an occupied pixel can
        # not have been removed earlier (it had this pixel as a neighbour), so the scan equals the simultaneous rule.
        pad = np.ones((pixels + 2, pixels + 2), dtype=np.int64)
        pad[1:-1, 1:-1] = img
        lonely = (pad[:-2, 1:-1] == 0) & (pad[2:, 1:-1] == 0) & (pad[1:-1, :-2] == 0) & (pad[1:-1, 2:] == 0)
        img = np.where(lonely, 0, img)
        # largest 4-connected component among pixels that have at least one occupied neighbour (src/utils.py:64-79)
        seen = np.zeros_like(img, dtype=bool)
        comps = []
        for x in range(pixels):
            for y in range(pixels):
                if not img[x, y] or seen[x, y]:
                    continue
                stack, comp = [(x, y)], []
                seen[x, y] = True
                while stack:
                    cx, cy = stack.pop()
                    comp.append((cx, cy))
                    for nx_, ny_ in ((cx - 1, cy), (cx + 1, cy), (cx, cy - 1), (cx, cy + 1)):
                        if 0 <= nx_ < pixels and 0 <= ny_ < pixels and img[nx_, ny_] and not seen[nx_, ny_]:
                            seen[nx_, ny_] = True
                            stack.append((nx_, ny_))
                if len(comp) > 1:  # single pixels are not nodes of the edge graph
                    comps.append(comp)
        if comps:
            comps.sort(key=lambda c: _component_order_key(c, pixels))
            best = max(comps, key=len)  # max() keeps the first of equal sizes, like the strict '<' of src/utils.py:69
            for x, y in best:
                out[i, x, y] = 1
        # (no component at all: the reference raises IndexError at src/utils.py:73; the restatement returns an empty geometry)
    return out.reshape(-1, pixels * pixels)


def extract_geometries(videos: torch.Tensor, zero_u_2: float, reference_frame: str = "lagrangian") -> np.ndarray:
    topo = topologies(videos, zero_u_2, reference_frame)
    return clean_pred(topo, topo.shape[1])
