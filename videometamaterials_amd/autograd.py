"""Training glue: the noise-prediction loss and the denoiser's backward pass as single autograd nodes.

(Forward-only for now: the backward plan lands with the training milestone.)
"""
from __future__ import annotations

import torch

from . import _native as N
from .plan import _stream


def noise_loss(noise: torch.Tensor, pred: torch.Tensor, squared: bool) -> torch.Tensor:
    """F.l1_loss / F.mse_loss (vddp.py:1053-1056) through vmm_loss_reduce (fp64 accumulation)."""
    if torch.is_grad_enabled() and pred.requires_grad:
        raise NotImplementedError("backward pass of the HIP path is not built yet")
    noise, pred = noise.contiguous(), pred.contiguous()
    acc = torch.empty(1, dtype=torch.float64, device=pred.device)
    out = torch.empty((), dtype=torch.float32, device=pred.device)
    N.check(N.lib().vmm_loss_reduce(noise.data_ptr(), pred.data_ptr(), pred.numel(), 1 if squared else 0, acc.data_ptr(), out.data_ptr(), _stream()),
            "vmm_loss_reduce")
    return out


def unet_forward_with_grad(model, x, time, cond, mask):
    raise NotImplementedError("backward pass of the HIP path is not built yet; call under torch.no_grad()")
