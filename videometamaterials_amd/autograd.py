"""Autograd glue: the whole denoiser forward/backward is ONE autograd node (the backward is the hand-derived HIP launch
list of the training plan, plan.py), so ``loss.backward()`` from the reference ``Trainer`` (vddp.py:1629) works unchanged.

One forward -> one backward: the training plan keeps its intermediates in a static arena, so a second forward before
the backward of the first would overwrite them (the reference training loop never does that, vddp.py:1620-1633).
"""
from __future__ import annotations

import torch

from . import _native as N
from .plan import _stream


class _UnetFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, model, plan, names, x, time, cond, mask, focus, *params):
        ctx.plan, ctx.names, ctx.model = plan, names, model
        out = plan.run(x, time, cond, mask, focus)
        return out.clone()

    @staticmethod
    def backward(ctx, dout):
        pl = ctx.plan
        want_dx = bool(ctx.needs_input_grad[3])  # d loss / d x: one extra launch (the stem's data gradient), only when x requires grad
        pl.backward(dout.contiguous(), want_dx=want_dx)
        views = pl.grad_views(dict(ctx.model.named_parameters()))
        grads = tuple(views[n].clone() if n in views else None for n in ctx.names)
        return (None, None, None, pl.dx.clone() if want_dx else None, None, None, None, None) + grads


def unet_forward_with_grad(model, x, time, cond, mask, focus=None):
    B, _, T, H, W = x.shape
    pl = model.get_plan(B, T, H, W, cond.shape[-1], x.device, training=True, focus=focus is not None)
    named = [(n, p) for n, p in model.named_parameters() if p.requires_grad]
    names = tuple(n for n, _ in named)
    return _UnetFn.apply(model, pl, names, x.contiguous(), time, cond.contiguous(), mask, focus, *[p for _, p in named])


class _NoiseLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, noise, pred, squared):
        noise, pred = noise.contiguous(), pred.contiguous()
        acc = torch.empty(1, dtype=torch.float64, device=pred.device)
        out = torch.empty((), dtype=torch.float32, device=pred.device)
        N.check(N.lib().vmm_loss_reduce(noise.data_ptr(), pred.data_ptr(), pred.numel(), 1 if squared else 0, acc.data_ptr(), out.data_ptr(), _stream()),
                "vmm_loss_reduce")
        ctx.save_for_backward(noise, pred)
        ctx.squared = squared
        return out

    @staticmethod
    def backward(ctx, g):
        noise, pred = ctx.saved_tensors
        g = g.contiguous().float()
        dpred = torch.empty_like(pred)
        N.check(N.lib().vmm_loss_grad(noise.data_ptr(), pred.data_ptr(), pred.numel(), 1 if ctx.squared else 0, g.data_ptr(), dpred.data_ptr(), _stream()),
                "vmm_loss_grad")
        return None, dpred, None


def noise_loss(noise: torch.Tensor, pred: torch.Tensor, squared: bool) -> torch.Tensor:
    """F.l1_loss / F.mse_loss (vddp.py:1053-1056) through vmm_loss_reduce (fp64 accumulation) and vmm_loss_grad."""
    return _NoiseLossFn.apply(noise, pred, squared)
