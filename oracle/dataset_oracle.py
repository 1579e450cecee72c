"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the reference's training-sample assembly.

Only tests/ may import this module.  It restates, operation by operation (every intermediate rounded to fp32 where the reference's
torch expression rounds), what ``Dataset.__init__`` / ``Dataset.__getitem__`` of
denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py compute once the GIF frames are decoded:

* global ranges and ``zero_u_2`` from ``frame_range_data.csv``                        vddp.py:1196-1242
* labels: per-frame interpolation of the stress curve + 'global-min-max-2' scaling     vddp.py:1255-1281, src/normalization.py:35-37
* fields -> sample: ToTensor (u8 / 255), un-normalise with the sample's own range, zero where the topology is void,
  normalise with the global range, select channels, pad / crop frames               vddp.py:1304-1397, 1114-1124

Pinned by tests/golden/dataset_*.npz, which the REAL reference produced (tests/golden/make_golden_dataset.py).
"""
from __future__ import annotations

from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

# field order of the decoded-frame tensor handed to fields_to_sample: one u8 plane stack per GIF the reference opens
FIELDS = {"lagrangian": ("topo", "u_1", "u_2", "s_mises", "s_22"), "eulerian": ("topo", "s_mises", "s_22", "ener")}


def global_ranges(frame_ranges: torch.Tensor, reference_frame: str) -> Dict[str, torch.Tensor]:
    """vddp.py:1200-1226: float64 0-dim tensors, exactly the attributes the reference keeps on the Dataset."""
    fr = frame_ranges
    if reference_frame == "eulerian":
        return dict(max_s_mises=fr[:, 0].max(), min_s_22=fr[:, 1].min(), max_s_22=fr[:, 2].max(), max_strain_energy=fr[:, 3].max())
    g = dict(min_u_1=fr[:, 0].min(), max_u_1=fr[:, 1].max(), min_u_2=fr[:, 2].min(), max_u_2=fr[:, 3].max(), max_s_mises=fr[:, 4].max(),
             min_s_22=fr[:, 5].min(), max_s_22=fr[:, 6].max(), max_strain_energy=fr[:, 7].max())
    g["zero_u_2"] = (torch.zeros(1) - g["min_u_2"]) / (g["max_u_2"] - g["min_u_2"])  # normalize(torch.zeros(1), ...) vddp.py:1226
    return g


def labels_from_curves(curves: np.ndarray, num_frames: int, per_frame_cond: bool) -> torch.Tensor:
    """vddp.py:1255-1271 (before scaling): float32 (N, num_frames) or (N, n_points - 1)."""
    if per_frame_cond:
        strain = 0.2
        given = np.linspace(0., strain, num=curves.shape[1])
        ev = np.linspace(0., strain, num=num_frames)
        ev[0] = 0.01 * strain
        curves = np.array([np.interp(ev, given, curves[i, :]) for i in range(curves.shape[0])])
        return torch.tensor(curves).float()
    return torch.tensor(curves[:, 1:]).float()


def scale_labels(labels: torch.Tensor, gmin: torch.Tensor, gmax: torch.Tensor) -> torch.Tensor:
    """Normalization(..., 'global-min-max-2').normalize (src/normalization.py:35-37), column by column like the reference."""
    out = torch.zeros(labels.shape)
    for i in range(labels.shape[1]):
        out[:, i] = 2. * torch.div(labels[:, i] - gmin, gmax - gmin) - 1.
    return out


def _unnorm(arr, lo, hi):   # vddp.py:1298
    return arr * (hi - lo) + lo


def _normalize(arr, lo, hi):  # vddp.py:1295
    return (arr - lo) / (hi - lo)


def fields_to_sample(frames_u8: torch.Tensor, ranges_row: torch.Tensor, g: Dict[str, torch.Tensor], reference_frame: str,
                     selected_channels: Sequence[int], num_frames: int, force_num_frames: bool = True) -> torch.Tensor:
    """frames_u8: (n_fields, f, H, W) uint8 in FIELDS[reference_frame] order (what PIL hands to ToTensor); ranges_row: that sample's
    row of frame_range_data.csv (float64).  Returns what ``Dataset.__getitem__`` returns as its tensor."""
    fld = {name: (frames_u8[i].to(torch.float32) / 255)[None] for i, name in enumerate(FIELDS[reference_frame])}  # ToTensor: (1, f, H, W)
    topo = fld["topo"]
    r = ranges_row
    if reference_frame == "eulerian":
        t = torch.cat((fld["topo"], fld["s_mises"], fld["s_22"], fld["ener"]), 0)
        t[1] = _unnorm(t[1], 0., r[0])
        t[2] = _unnorm(t[2], r[1], r[2])
        t[3] = _unnorm(t[3], 0., r[3])
        for i in range(1, 4):
            t[i][topo[0] == 0.] = 0.
        t[1] = _normalize(t[1], 0., g["max_s_mises"])
        t[2] = _normalize(t[2], g["min_s_22"], g["max_s_22"])
        t[3] = _normalize(t[3], 0., g["max_strain_energy"])
    elif num_frames != 1:
        t = torch.cat((fld["u_1"], fld["u_2"], fld["s_mises"], fld["s_22"]), 0)
        t[0] = _unnorm(t[0], r[0], r[1])
        t[1] = _unnorm(t[1], r[2], r[3])
        t[2] = _unnorm(t[2], 0., r[4])
        t[3] = _unnorm(t[3], r[5], r[6])
        for i in range(4):
            t[i][topo[0] == 0.] = 0.
        t[0] = _normalize(t[0], g["min_u_1"], g["max_u_1"])
        t[1] = _normalize(t[1], g["min_u_2"], g["max_u_2"])
        t[2] = _normalize(t[2], 0., g["max_s_mises"])
        t[3] = _normalize(t[3], g["min_s_22"], g["max_s_22"])
    else:  # single-frame ablation: topology and sigma_22 (vddp.py:1370-1390); the reference overrides selected_channels here
        t = torch.cat((fld["topo"], fld["s_22"]), 0)
        t[1] = _unnorm(t[1], r[5], r[6])
        t[1][topo[0] == 0.] = 0.
        t[1] = _normalize(t[1], g["min_s_22"], g["max_s_22"])
        selected_channels = [0, 1]
    t = t[list(selected_channels)]
    if force_num_frames:  # cast_num_frames, vddp.py:1114-1124
        f = t.shape[1]
        if f > num_frames:
            t = t[:, :num_frames]
        elif f < num_frames:
            t = torch.nn.functional.pad(t, (0, 0, 0, 0, 0, num_frames - f))
    return t
