"""Geometry extraction from sampled videos on the GPU.

Replaces the host-side tail of ``Trainer.save_preds`` (vddp.py:1890-1918) and ``clean_pred`` (src/utils.py:32-82): the
reference moves every sampled video to the CPU, loops over pixels in Python and builds a networkx graph per sample; here
one workgroup per sample produces the same rows of ``geometries.csv`` (bit-exact, tests/test_gpu_geometry.py).
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _native as N


def extract_geometries(videos: torch.Tensor, zero_u_2: float = 0.0, reference_frame: str = "lagrangian") -> torch.Tensor:
    """videos: (N, C, T, P, P) fp32 on the GPU (``GaussianDiffusion.sample`` output).  Returns int32 (N, (P/2)**2) with entries
    0 / 1 -- row n is the flattened, cleaned topology of sample n exactly as the reference writes it to ``geometries.csv``.

    reference_frame 'lagrangian' with more than one frame: void iff channel 1 stays within 0.02 of ``zero_u_2`` (the dataset's
    normalised zero displacement, ``ds.zero_u_2``) in every frame of the mirrored upper-left quarter; 'eulerian' (or a single
    frame): channel 0 of frame 0 in the bottom-left quarter, binarised at 0.5."""
    if reference_frame not in ("lagrangian", "eulerian"):
        raise ValueError(f"reference_frame must be 'lagrangian' or 'eulerian', got {reference_frame!r}")
    if videos.dim() != 5 or videos.shape[-1] != videos.shape[-2]:
        raise ValueError("videos must be (N, C, T, P, P)")
    if not videos.is_cuda:
        raise N.NativeError("extract_geometries needs a GPU tensor (there is no CPU path)")
    v = videos.detach().to(torch.float32).contiguous()
    n, c, t, p, _ = v.shape
    out = torch.empty((n, (p // 2) ** 2), dtype=torch.int32, device=v.device)
    stream = C.c_void_p(torch.cuda.current_stream(v.device).cuda_stream)
    N.check(N.lib().vmm_extract_geometry(v.data_ptr(), n, c, t, p, 1 if reference_frame == "lagrangian" else 0, C.c_float(float(zero_u_2)),
                                         out.data_ptr(), stream), "vmm_extract_geometry")
    return out
