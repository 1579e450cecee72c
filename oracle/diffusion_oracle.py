"""ORACLE (test infrastructure, not product code).

CPU restatement of `GaussianDiffusion` (vddp.py:824-1067) and of the integer
host helpers the sampling callers use (vddp.py:47-53, 1506-1532, 1848-1868).
All random draws are *injected* (noise tensors / timesteps passed in) so the
GPU path can be compared on identical inputs (SURVEY 8a "RNG" note).

Pinned by tests/golden/diffusion_*.npz (outputs of the real reference).
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import it.
"""
from __future__ import annotations

from typing import Callable, Dict, List, Sequence

import torch

Tensor = torch.Tensor

SCHEDULE_NAMES = (
    "betas",
    "alphas_cumprod",
    "alphas_cumprod_prev",
    "sqrt_alphas_cumprod",
    "sqrt_one_minus_alphas_cumprod",
    "log_one_minus_alphas_cumprod",
    "sqrt_recip_alphas_cumprod",
    "sqrt_recipm1_alphas_cumprod",
    "posterior_variance",
    "posterior_log_variance_clipped",
    "posterior_mean_coef1",
    "posterior_mean_coef2",
)


def cosine_betas(timesteps: int, s: float = 0.008) -> Tensor:
    """float64 cosine schedule, clipped to [0, 0.9999] (vddp.py:829-839)."""
    grid = torch.linspace(0, timesteps, timesteps + 1, dtype=torch.float64)
    acp = torch.cos(((grid / timesteps) + s) / (1 + s) * torch.pi * 0.5) ** 2
    acp = acp / acp[0]
    return torch.clip(1 - acp[1:] / acp[:-1], 0, 0.9999)


def schedule_buffers(timesteps: int) -> Dict[str, Tensor]:
    """The 12 registered buffers, computed in float64 and cast to fp32 (vddp.py:862-900)."""
    betas = cosine_betas(timesteps)
    alphas = 1.0 - betas
    acp = torch.cumprod(alphas, dim=0)
    acp_prev = torch.cat([torch.ones(1, dtype=torch.float64), acp[:-1]])
    post_var = betas * (1.0 - acp_prev) / (1.0 - acp)
    out64 = {
        "betas": betas,
        "alphas_cumprod": acp,
        "alphas_cumprod_prev": acp_prev,
        "sqrt_alphas_cumprod": acp.sqrt(),
        "sqrt_one_minus_alphas_cumprod": (1.0 - acp).sqrt(),
        "log_one_minus_alphas_cumprod": (1.0 - acp).log(),
        "sqrt_recip_alphas_cumprod": (1.0 / acp).sqrt(),
        "sqrt_recipm1_alphas_cumprod": (1.0 / acp - 1).sqrt(),
        "posterior_variance": post_var,
        "posterior_log_variance_clipped": post_var.clamp(min=1e-20).log(),
        "posterior_mean_coef1": betas * acp_prev.sqrt() / (1.0 - acp),
        "posterior_mean_coef2": (1.0 - acp_prev) * alphas.sqrt() / (1.0 - acp),
    }
    return {k: v.to(torch.float32) for k, v in out64.items()}


def _at(buf: Tensor, t: Tensor, ndim: int) -> Tensor:
    """extract (vddp.py:824-827)."""
    return buf.gather(-1, t).reshape(t.shape[0], *((1,) * (ndim - 1)))


def q_sample(sch, x0: Tensor, t: Tensor, noise: Tensor) -> Tensor:
    """vddp.py:1036-1042."""
    return _at(sch["sqrt_alphas_cumprod"], t, x0.ndim) * x0 + _at(sch["sqrt_one_minus_alphas_cumprod"], t, x0.ndim) * noise


def predict_start_from_noise(sch, x_t: Tensor, t: Tensor, eps: Tensor) -> Tensor:
    """vddp.py:920-924."""
    return _at(sch["sqrt_recip_alphas_cumprod"], t, x_t.ndim) * x_t - _at(sch["sqrt_recipm1_alphas_cumprod"], t, x_t.ndim) * eps


def dynamic_threshold(x0: Tensor, percentile: float = 0.9) -> Tensor:
    """vddp.py:939-951: s = max(quantile(|x0|, p), 1); clamp(-s, s) / s."""
    s = torch.quantile(x0.flatten(1).abs(), percentile, dim=-1).clamp(min=1.0)
    s = s.view(-1, *((1,) * (x0.ndim - 1)))
    return x0.clamp(-s, s) / s


def q_posterior_mean_logvar(sch, x0: Tensor, x_t: Tensor, t: Tensor):
    """vddp.py:926-933."""
    mean = _at(sch["posterior_mean_coef1"], t, x_t.ndim) * x0 + _at(sch["posterior_mean_coef2"], t, x_t.ndim) * x_t
    return mean, _at(sch["posterior_log_variance_clipped"], t, x_t.ndim)


def p_sample_step(sch, eps_fn: Callable, x: Tensor, t: Tensor, noise: Tensor, *, use_dynamic_thres: bool = True, percentile: float = 0.9) -> Tensor:
    """p_mean_variance + p_sample (vddp.py:935-963) with the Gaussian draw injected."""
    x0 = predict_start_from_noise(sch, x, t, eps_fn(x, t))
    x0 = dynamic_threshold(x0, percentile) if use_dynamic_thres else x0.clamp(-1.0, 1.0)
    mean, logvar = q_posterior_mean_logvar(sch, x0, x, t)
    keep = (1 - (t == 0).float()).reshape(-1, *((1,) * (x.ndim - 1)))
    return mean + keep * (0.5 * logvar).exp() * noise


def p_sample_loop(sch, eps_fn: Callable, x_T: Tensor, noises: Sequence[Tensor], *, timesteps: int, use_dynamic_thres: bool = True) -> Tensor:
    """vddp.py:965-975: i = T-1 ... 0 (one Gaussian per step, also at t = 0), then (x+1)/2 un-clamped."""
    img = x_T
    B = x_T.shape[0]
    for j, i in enumerate(reversed(range(timesteps))):
        img = p_sample_step(sch, eps_fn, img, torch.full((B,), i, dtype=torch.long), noises[j], use_dynamic_thres=use_dynamic_thres)
    return (img + 1) * 0.5


def ddim_times(total: int, sampling: int) -> List[int]:
    """vddp.py:990-991 (INT, bit-exact)."""
    return list(reversed(torch.linspace(-1, total - 1, steps=sampling + 1).int().tolist()))


def ddim_sample(sch, eps_fn: Callable, x_T: Tensor, noises: Sequence[Tensor], *, timesteps: int, sampling_timesteps: int, eta: float = 0.0) -> Tensor:
    """vddp.py:986-1018 (no clipping / thresholding on this path)."""
    times = ddim_times(timesteps, sampling_timesteps)
    img, B = x_T, x_T.shape[0]
    acp = sch["alphas_cumprod"]
    for j, (t_now, t_next) in enumerate(zip(times[:-1], times[1:])):
        tt = torch.full((B,), t_now, dtype=torch.long)
        eps = eps_fn(img, tt)
        x0 = predict_start_from_noise(sch, img, tt, eps)
        if t_next < 0:
            img = x0
            continue
        a, a_next = acp[t_now], acp[t_next]
        sigma = eta * ((1 - a / a_next) * (1 - a_next) / (1 - a)).sqrt()
        c = (1 - a_next - sigma**2).sqrt()
        img = x0 * a_next.sqrt() + c * eps + sigma * noises[j]
    return (img + 1) * 0.5


def p_losses(sch, eps_fn: Callable, x0: Tensor, t: Tensor, noise: Tensor, loss_type: str = "l1") -> Tensor:
    """vddp.py:1044-1060; x0 already mapped to [-1,1] (normalize_img, vddp.py:1066,1109)."""
    pred = eps_fn(q_sample(sch, x0, t, noise), t)
    if loss_type == "l1":
        return (noise - pred).abs().mean()
    if loss_type == "l2":
        return ((noise - pred) ** 2).mean()
    raise NotImplementedError(loss_type)


# --------------------------------------------------------------------------- integer host helpers
def num_to_groups(num: int, divisor: int) -> List[int]:
    """vddp.py:47-53."""
    full, rem = divmod(num, divisor)
    return [divisor] * full + ([rem] if rem else [])


def shard_rows(n_rows: int, rank: int, world: int, batch: int):
    """cond_to_gpu (vddp.py:1506-1532) as index ranges: contiguous block of floor(N/P) rows per
    rank (last rank takes the remainder), chunked to `batch`. Returns [(start, end), ...] global rows."""
    per = n_rows // world
    lo = rank * per
    hi = (rank + 1) * per if rank != world - 1 else n_rows
    out, cur = [], lo
    for g in num_to_groups(hi - lo, batch):
        out.append((cur, cur + g))
        cur += g
    return out


def strip_padding(gathered: Tensor, lengths: Sequence[int], max_len: int) -> Tensor:
    """remove_padding (vddp.py:1848-1868)."""
    parts, start = [], 0
    for n in lengths:
        parts.append(gathered[start : start + int(n)])
        start += max_len
    return torch.cat(parts, dim=0)
