"""Shared test utilities: deterministic synthetic weights and the named test configs.

The golden generator (tests/golden/make_golden.py) loads these exact weights into
the real reference; the tests regenerate them bit-identically from
(name, shape, seed), so no weight tensors need to be committed.
"""
from __future__ import annotations

import hashlib
import json
import os
from typing import Dict, Tuple

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(__file__), "golden")

# name -> (Unet3D kwargs, (B, T, H, W), cond_len)
CONFIGS = {
    # Lagrangian model.yaml wiring at dim=16 (SURVEY cfgL, reduced)
    "lagr16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16,
                    use_temporal_attention_cond=True, per_frame_cond=True, cond_bias=True), (2, 11, 32, 32), 11),
    # Lagrangian wiring at the real widths (64..512 channels), small frames
    "lagr64": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16,
                    use_temporal_attention_cond=True, per_frame_cond=True, cond_bias=True), (1, 11, 32, 32), 11),
    # BASELINE configs[0]: Unet3D(dim=16, channels=1), defaults otherwise (cfg1)
    "plumb16": (dict(dim=16, channels=1), (2, 4, 32, 32), 51),
    # cfg4 wiring reduced: per_frame_cond=False, self-stacked with 16 CNN tokens, temporal cond
    "hires16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16,
                     use_temporal_attention_cond=True, per_frame_cond=False), (2, 6, 16, 16), 51),
}


def _seed_for(name: str, seed: int) -> int:
    h = hashlib.sha256(f"{seed}:{name}".encode()).digest()
    return int.from_bytes(h[:7], "little")


def synth_tensor(name: str, shape: Tuple[int, ...], seed: int) -> torch.Tensor:
    """Deterministic fp32 value for a parameter, scaled so every branch of the net matters."""
    g = torch.Generator().manual_seed(_seed_for(name, seed))
    shape = tuple(shape)
    if name.endswith("rotary_emb.freqs"):
        d = shape[0] * 2
        return 1.0 / (10000 ** (torch.arange(0, d, 2).float() / d))
    if name.endswith("gamma") or name.endswith("norm.weight") or name == "cond_token_to_hidden.0.weight":
        return 1.0 + 0.2 * torch.randn(shape, generator=g)
    if name.endswith("norm.bias") or name == "cond_token_to_hidden.0.bias":
        return 0.1 * torch.randn(shape, generator=g)
    if name.startswith("null_text") or name.endswith("relative_attention_bias.weight"):
        return torch.randn(shape, generator=g)
    if name.endswith("bias"):
        return (torch.rand(shape, generator=g) * 2 - 1) * 0.1
    fan_in = int(np.prod(shape[1:])) if len(shape) > 1 else shape[0]
    if name.endswith(".4.weight") and name.startswith("ups.") and len(shape) == 5:
        fan_in = shape[0] * 4  # transposed conv: (Cin, Cout, 1, 4, 4); 4 taps hit each output
    bound = (3.0 / max(fan_in, 1)) ** 0.5
    return (torch.rand(shape, generator=g) * 2 - 1) * bound


def load_shapes(cfg_name: str) -> Dict[str, Tuple[int, ...]]:
    with open(os.path.join(GOLDEN_DIR, f"shapes_{cfg_name}.json")) as f:
        return {k: tuple(v) for k, v in json.load(f).items()}


def synth_state_dict(shapes: Dict[str, Tuple[int, ...]], seed: int = 0) -> Dict[str, torch.Tensor]:
    return {k: synth_tensor(k, s, seed) for k, s in shapes.items()}


def synth_inputs(cfg_name: str, seed: int = 1):
    """x ~ N(0,1) (a noisy sample), integer timesteps, cond ~ U[-1,1)."""
    _, (B, T, H, W), cond_len = CONFIGS[cfg_name]
    C = CONFIGS[cfg_name][0]["channels"]
    g = torch.Generator().manual_seed(seed)
    x = torch.randn((B, C, T, H, W), generator=g)
    t = torch.randint(0, 256, (B,), generator=g)
    cond = torch.rand((B, cond_len), generator=g) * 2 - 1
    return x, t, cond


def rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """Relative L2 error ||a-b|| / ||b|| (the north_star's 'relative fp32' measure)."""
    a, b = a.double(), b.double()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


def max_rel_err(a: torch.Tensor, b: torch.Tensor) -> float:
    """max |a-b| / max |b|."""
    return float((a.double() - b.double()).abs().max() / b.double().abs().max().clamp_min(1e-30))
