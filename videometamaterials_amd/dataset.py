"""Training-sample assembly on the GPU: the arithmetic of the reference's ``Dataset`` (vddp.py:1126-1397).

The reference opens four or five GIFs per sample on the host for every ``__getitem__``, converts them to fp32, rescales each field from
the sample's own range to the global range and zeroes the void pixels -- per sample, per epoch, on DataLoader workers.  Here the decoded
bytes of the whole dataset are uploaded ONCE (5 bytes per pixel and frame: the 53k-sample training set of the paper is 27 GB, a tenth of
one MI355X's HBM) and a minibatch is one kernel launch that gathers by index (``vmm_fields_to_samples``), bit-exact against the
reference's expressions (tests/test_gpu_dataset.py; fixtures made by the reference's own Dataset).

Same constructor keywords, attributes (``labels``, ``detached_labels``, ``labels_scaling``, ``zero_u_2``, ``min_u_1`` ...,
``selected_channels``) and ``__len__`` / ``__getitem__`` results as the reference class; ``batch(indices)`` is the device-side
replacement of DataLoader + collate.  Not rebuilt: ``horizontal_flip`` (the reference draws the flip per frame and per field, which
scrambles a sample; it is off in every configuration) -- raises NotImplementedError.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

from . import _native as N

# one decoded u8 plane stack per GIF the reference opens, in this order (field 0 must be the topology: it is the void mask)
FIELDS = {"lagrangian": ("topo", "u_1", "u_2", "s_mises", "s_22"), "eulerian": ("topo", "s_mises", "s_22", "ener")}


class LabelScaling:
    """``Normalization(labels, ['continuous'] * n, 'global-min-max-2')`` (src/normalization.py:4-96): to [-1, 1] by the global extrema."""

    def __init__(self, data: torch.Tensor):
        self.globalmin, self.globalmax = torch.min(data), torch.max(data)
        self.strategy = "global-min-max-2"

    def normalize(self, data: torch.Tensor) -> torch.Tensor:
        return 2. * torch.div(data - self.globalmin.to(data.device), (self.globalmax - self.globalmin).to(data.device)) - 1.

    def unnormalize(self, data: torch.Tensor) -> torch.Tensor:
        return torch.mul(0.5 * data + 0.5, (self.globalmax - self.globalmin).to(data.device)) + self.globalmin.to(data.device)


def interpolate_curves(curves: np.ndarray, num_frames: int, per_frame_cond: bool) -> torch.Tensor:
    """Stress-strain rows of stress_strain_data.csv -> conditioning labels before scaling (vddp.py:1255-1271)."""
    if not per_frame_cond:
        return torch.tensor(curves[:, 1:]).float()  # first column is all zeros
    strain = 0.2
    given = np.linspace(0., strain, num=curves.shape[1])
    at = np.linspace(0., strain, num=num_frames)
    at[0] = 0.01 * strain  # the first frame is taken at 1 % strain
    return torch.tensor(np.array([np.interp(at, given, row) for row in curves])).float()


class Dataset:
    """frames: uint8 (N, n_fields, f, H, W), FIELDS[reference_frame] order, decoded at image_size (``from_folder`` reads a reference
    dataset folder); frame_ranges: frame_range_data.csv (N, 8 | 4); curves: stress_strain_data.csv (N, n_points)."""

    def __init__(self, frames: torch.Tensor, frame_ranges, curves, *, labels_scaling: Optional[LabelScaling] = None,
                 selected_channels: Sequence[int] = (0, 1, 2, 3), num_frames: int = 16, horizontal_flip: bool = False, force_num_frames: bool = True,
                 per_frame_cond: bool = False, reference_frame: str = "eulerian", device: Optional[torch.device] = None):
        if reference_frame not in FIELDS:
            raise ValueError(f"reference_frame must be 'lagrangian' or 'eulerian', got {reference_frame!r}")
        if horizontal_flip:
            raise NotImplementedError("horizontal_flip: the reference flips every frame of every field independently (see module docstring)")
        if frames.dtype != torch.uint8 or frames.dim() != 5 or frames.shape[1] != len(FIELDS[reference_frame]):
            raise ValueError(f"frames must be uint8 (N, {len(FIELDS[reference_frame])}, f, H, W) in the order {FIELDS[reference_frame]}")
        self.reference_frame, self.num_frames, self.force_num_frames = reference_frame, int(num_frames), bool(force_num_frames)
        self.image_size = frames.shape[-1]
        fr = torch.as_tensor(np.asarray(frame_ranges, dtype=np.float64))
        if fr.shape[0] != frames.shape[0]:
            raise ValueError("frame_ranges and frames disagree on the number of samples")
        self.frame_ranges = fr
        z = torch.zeros((), dtype=torch.float64)
        # global ranges (vddp.py:1200-1226) and, per output channel of the un-selected 4-channel tensor: (source field, per-sample lo, hi, global lo, hi)
        if reference_frame == "eulerian":
            self.max_s_mises, self.min_s_22, self.max_s_22, self.max_strain_energy = fr[:, 0].max(), fr[:, 1].min(), fr[:, 2].max(), fr[:, 3].max()
            self.zero_u_2 = None
            zero = torch.zeros(len(fr), dtype=torch.float64)
            chans = [(0, None), (1, (zero, fr[:, 0], z, self.max_s_mises)), (2, (fr[:, 1], fr[:, 2], self.min_s_22, self.max_s_22)),
                     (3, (zero, fr[:, 3], z, self.max_strain_energy))]
        else:
            self.min_u_1, self.max_u_1, self.min_u_2, self.max_u_2 = fr[:, 0].min(), fr[:, 1].max(), fr[:, 2].min(), fr[:, 3].max()
            self.max_s_mises, self.min_s_22, self.max_s_22, self.max_strain_energy = fr[:, 4].max(), fr[:, 5].min(), fr[:, 6].max(), fr[:, 7].max()
            self.zero_u_2 = (torch.zeros(1) - self.min_u_2) / (self.max_u_2 - self.min_u_2)
            zero = torch.zeros(len(fr), dtype=torch.float64)
            if self.num_frames != 1:
                chans = [(1, (fr[:, 0], fr[:, 1], self.min_u_1, self.max_u_1)), (2, (fr[:, 2], fr[:, 3], self.min_u_2, self.max_u_2)),
                         (3, (zero, fr[:, 4], z, self.max_s_mises)), (4, (fr[:, 5], fr[:, 6], self.min_s_22, self.max_s_22))]
            else:  # one-frame ablation: topology and sigma_22, whatever was selected (vddp.py:1370-1390)
                chans = [(0, None), (4, (fr[:, 5], fr[:, 6], self.min_s_22, self.max_s_22))]
                selected_channels = [0, 1]
        self.selected_channels = list(selected_channels)
        chans = [chans[c] for c in self.selected_channels]
        src, coef = [], torch.zeros(len(fr), len(chans), 4, dtype=torch.float32)
        for c, (field, rng) in enumerate(chans):
            src.append(field | (0x100 if rng is None else 0))
            if rng is not None:  # differences in float64 like the reference's 0-dim tensors, then rounded once to the fp32 the kernels of torch see
                lo, hi, glo, ghi = rng
                coef[:, c, 0], coef[:, c, 1] = (hi - lo).float(), lo.float()
                coef[:, c, 2], coef[:, c, 3] = glo.float(), (ghi - glo).float()
        # labels (vddp.py:1253-1281)
        self.labels = interpolate_curves(np.asarray(curves, dtype=np.float64), self.num_frames, per_frame_cond)
        self.detached_labels = self.labels.clone().detach().numpy()
        self.labels_scaling = LabelScaling(self.labels) if labels_scaling is None else labels_scaling
        self.labels = self.labels_scaling.normalize(self.labels)
        # device residents
        dev = torch.device(device) if device is not None else (frames.device if frames.is_cuda else torch.device("cuda"))
        if dev.type != "cuda":
            raise N.NativeError("videometamaterials_amd.Dataset keeps the decoded frames in HBM (there is no CPU path)")
        self.device = dev
        self.frames = frames.to(dev).contiguous()
        self._src = torch.tensor(src, dtype=torch.int32, device=dev)
        self._coef = coef.to(dev)
        self._labels_dev = self.labels.to(dev)
        self._bad_index = torch.zeros((), dtype=torch.bool, device=dev)  # latched by batch() on device-resident indices out of range

    def __len__(self) -> int:
        return self.frames.shape[0]

    def batch(self, indices) -> Tuple[torch.Tensor, torch.Tensor]:
        """(B, channels, T, H, W) fp32 in [0, 1] and the (B, L) labels of dataset rows ``indices``, on the device, in one launch."""
        n, nf, f, H, W = self.frames.shape
        if isinstance(indices, torch.Tensor) and indices.is_cuda:
            # device-resident indices (a device sampler): wrapped and range-checked without a host round trip.  An index outside [-n, n) is
            # clamped into range for the launch (no out-of-bounds read) and latched in a device-side flag that `check_indices()` turns into
            # the IndexError of the host path at the caller's next natural synchronisation point (where the loop reads the loss);
            # VMM_DEBUG_SYNC checks on the spot.
            idx = indices.to(torch.int64).reshape(-1)
            if idx.numel() == 0:
                raise IndexError("dataset index out of range")
            self._bad_index |= ((idx < -n) | (idx >= n)).any()
            if os.environ.get("VMM_DEBUG_SYNC"):
                self.check_indices()
            idx = torch.where(idx < 0, idx + n, idx).clamp_(0, n - 1).to(torch.int32).contiguous()
        else:
            # the usual case, a sampler's list / numpy array: validated on the host (no device -> host synchronisation on the training
            # stream), negative indices count from the end like the reference's list indexing
            host = np.asarray(indices.cpu() if isinstance(indices, torch.Tensor) else indices, dtype=np.int64).reshape(-1)
            if host.size == 0 or host.min() < -n or host.max() >= n:
                raise IndexError("dataset index out of range")
            host = np.where(host < 0, host + n, host).astype(np.int32)
            idx = torch.from_numpy(host).to(self.device, non_blocking=True)
        T = self.num_frames if self.force_num_frames else f
        nch = len(self.selected_channels)
        out = torch.empty((idx.numel(), nch, T, H, W), dtype=torch.float32, device=self.device)
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        N.check(N.lib().vmm_fields_to_samples(self.frames.data_ptr(), nf, f, H * W, idx.data_ptr(), idx.numel(), self._src.data_ptr(), self._coef.data_ptr(),
                                              nch, T, out.data_ptr(), stream), "vmm_fields_to_samples")
        return out, self._labels_dev[idx.long()]

    def check_indices(self) -> None:
        """Raise the IndexError a batch() call with device-resident indices could not raise without a synchronisation (one device -> host read;
        call it where the training loop already waits for the device, e.g. when it logs the loss).  The flag is cleared."""
        if bool(self._bad_index):
            self._bad_index.zero_()
            raise IndexError("dataset index out of range (device-resident indices of an earlier batch() call; the batch was assembled from clamped rows)")

    def __getitem__(self, index: int) -> Tuple[torch.Tensor, torch.Tensor]:
        x, lab = self.batch([int(index)])
        return x[0], lab[0]

    @classmethod
    def from_folder(cls, folder: str, image_size: int, exts: Sequence[str] = ("gif",), **kw) -> "Dataset":
        """The reference's constructor signature (folder, image_size, ...): read_folder, then upload."""
        frames, fr, curves = read_folder(folder, image_size, exts, kw.get("reference_frame", "eulerian"))
        ds = cls(frames, fr, curves, **kw)
        ds.write_min_max_values(folder)
        return ds

    def write_min_max_values(self, folder: str) -> str:
        """<folder>/min_max_values.csv exactly as the reference's constructor leaves it (vddp.py:1210-1246; same rows, same order): the
        constants its README points to for rescaling predictions to physical units."""
        import csv
        names = (["min_u_1", "max_u_1", "min_u_2", "max_u_2"] if self.reference_frame == "lagrangian" else []) + \
                ["max_s_mises", "min_s_22", "max_s_22", "max_strain_energy"]
        path = os.path.join(folder, "min_max_values.csv")
        with open(path, "w", newline="") as f:
            csv.writer(f).writerows([[k, getattr(self, k).item()] for k in names])
        return path


def read_folder(folder: str, image_size: int, exts: Sequence[str] = ("gif",), reference_frame: str = "eulerian"):
    """Read a dataset folder laid out like the reference's (gifs/<field>/<i>.gif, frame_range_data.csv, stress_strain_data.csv;
    vddp.py:1143-1198, 1253) with PIL: (frames uint8 (N, n_fields, f, H, W), frame ranges, stress curves), all on the host.  The frames
    must already be image_size x image_size (the published dataset is); the reference's Resize / CenterCrop of other sizes is host
    image processing and not rebuilt."""
    from pathlib import Path
    from PIL import Image
    frame = reference_frame
    out = None  # (N, n_fields, f, H, W) uint8, allocated once the first GIF is decoded and filled in place (the 53 k-sample training set
    #             is 27 GB decoded: per-sample lists + two np.stack copies would double that on the host)
    n_fields = len(FIELDS[frame])
    for fi, field in enumerate(FIELDS[frame]):
        paths = sorted((p for ext in exts for p in Path(os.path.join(folder, "gifs", field)).glob(f"**/*.{ext}")), key=lambda p: int(p.name.split(".")[0]))
        assert all(int(p.stem) == i for i, p in enumerate(paths)), "file position is not equal to index"
        if out is not None and len(paths) != out.shape[0]:
            raise ValueError("number of files / frames in the field folders are not equal")
        for si, p in enumerate(paths):
            img, planes, i = Image.open(p), [], 0
            while True:  # seek_all_images, vddp.py:1077-1088
                try:
                    img.seek(i)
                except EOFError:
                    break
                planes.append(np.asarray(img.convert("L"), dtype=np.uint8))
                i += 1
            if out is None:
                out = np.empty((len(paths), n_fields, len(planes)) + planes[0].shape, dtype=np.uint8)
            if len(planes) != out.shape[2] or planes[0].shape != out.shape[3:]:
                raise ValueError("number of files / frames in the field folders are not equal")
            for k, pl in enumerate(planes):
                out[si, fi, k] = pl
    if out is None:
        raise ValueError(f"no {exts} files under {folder}/gifs")
    frames = torch.from_numpy(out)
    if frames.shape[-1] != image_size or frames.shape[-2] != image_size:
        raise NotImplementedError(f"GIFs are {frames.shape[-2]} x {frames.shape[-1]}, image_size is {image_size}: resize them offline")
    fr = np.genfromtxt(os.path.join(folder, "frame_range_data.csv"), delimiter=",")
    curves = np.genfromtxt(os.path.join(folder, "stress_strain_data.csv"), delimiter=",")
    return frames, fr, curves
