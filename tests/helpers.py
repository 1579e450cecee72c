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
    # BASELINE configs[3] wiring at the real widths with its 22 frames (more than the fused temporal kernels' 16 slots: the
    # two-frame-tile temporal attention path at every level), small frames
    "hires64t22": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16,
                        use_temporal_attention_cond=True, per_frame_cond=False), (2, 22, 32, 32), 51),  # (B = 1 crashes the reference: torch.squeeze in SignalEmbedding, vddp.py:571)
    # periodic padding variants (vddp.py:153-243): 'circular' at the real widths (2-D-tiled halo kernels at 32 x 32, wrapped implicit GEMMs
    # below; Upsample = CircularUpsample), 'circular_1d' at dim 16 (horizontal axis periodic; every convolution wrapped in a helper module)
    "circ64": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                    per_frame_cond=True, cond_bias=True, padding_mode="circular"), (1, 11, 32, 32), 11),
    # cond_attention = 'cross-attention' (vddp.py:354-363, 476-485; north_star's "cross-attention on the stress-strain conditioning"): queries from
    # to_q, keys / values = the conditioning tokens alone; the temporal sites add the (frames x frames) positional bias to the (frames x tokens)
    # scores, so tokens == frames.  dim 16, and the real widths (where self-stacked would take the fused kernels)
    "cross16": (dict(dim=16, channels=3, cond_attention="cross-attention", cond_attention_tokens=6, use_temporal_attention_cond=True,
                     per_frame_cond=False), (2, 6, 16, 16), 51),
    "cross64": (dict(dim=64, channels=3, cond_attention="cross-attention", cond_attention_tokens=11, use_temporal_attention_cond=True,
                     per_frame_cond=False), (2, 11, 16, 16), 40),
    "cross16s": (dict(dim=16, channels=3, cond_attention="cross-attention", cond_attention_tokens=9, use_temporal_attention_cond=False,
                      per_frame_cond=False), (2, 5, 16, 16), 51),  # spatial sites only: any number of tokens
    # cond_to_time = 'concat' (vddp.py:670, 788-789): the ResnetBlock mlps read cat(t, hidden); per-frame conditioning and the CNN embedding
    "concat16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                      per_frame_cond=True, cond_bias=True, cond_to_time="concat"), (2, 11, 16, 16), 11),
    "concat16c": (dict(dim=16, channels=1, cond_to_time="concat"), (2, 4, 16, 16), 51),
    # conditioning tokens at the spatial sites only: the temporal attentions see no tokens, so a non-trivial focus_present_mask is legal (vddp.py:514-524)
    "focus16s": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=9, use_temporal_attention_cond=False,
                      per_frame_cond=False), (3, 5, 16, 16), 51),
    # cond_att_GRU (vddp.py:546-549, 646-649, 769-770): the conditioning tokens are the 3-layer GRU's states over the signal, one per sample of it
    # (cond_attention_tokens == signal length); more tokens than the fused kernels' 16: the generic attention paths at every site
    "gru16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=34, use_temporal_attention_cond=True, per_frame_cond=False,
                   cond_att_GRU=True), (2, 5, 16, 16), 34),  # (the CNN embedding of the same signal needs 32 .. 63 samples)
    "circ1d16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                      per_frame_cond=True, cond_bias=True, padding_mode="circular_1d"), (2, 11, 32, 32), 11),
    # ---- the constructor keywords main.py:62-80 forwards from model.yaml, off their defaults (vddp.py:575-626, 669-710) ----
    # attn_heads (vddp.py:581, 615, 617, 679, 687): heads of all three attention families and of the relative-position bias; 4 (Lagrangian wiring)
    # and 3 (not a divisor of the 256-thread row kernels' block; CNN tokens)
    "heads4": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                    per_frame_cond=True, cond_bias=True, attn_heads=4), (2, 11, 16, 16), 11),
    "heads3": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=7, use_temporal_attention_cond=True,
                    per_frame_cond=False, attn_heads=3), (2, 6, 16, 16), 51),
    # attn_dim_head (vddp.py:582, 612, 615): the TEMPORAL attentions only (the linear and the mid spatial attention keep their default 32);
    # 16 = the rotary embedding shrinks with the head (RotaryEmbedding(16)), 64 = partial rotary (the leading 32 features of every head rotate)
    "dh16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                  per_frame_cond=True, cond_bias=True, attn_dim_head=16), (2, 11, 16, 16), 11),
    "dh64": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                  per_frame_cond=True, cond_bias=True, attn_dim_head=64), (2, 11, 16, 16), 11),
    # ... at the real widths (where attn_dim_head = 32 would take the fused attention blocks), and under 'cross-attention' (to_q rotated alone)
    "dh64w64": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                     per_frame_cond=True, cond_bias=True, attn_dim_head=64), (1, 11, 16, 16), 11),
    "dh16w64": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                     per_frame_cond=False, attn_dim_head=16), (2, 6, 16, 16), 51),
    "dh64cross": (dict(dim=16, channels=3, cond_attention="cross-attention", cond_attention_tokens=6, use_temporal_attention_cond=True,
                       per_frame_cond=False, attn_dim_head=64), (2, 6, 16, 16), 51),
    "dh24": (dict(dim=16, channels=1, attn_dim_head=24, attn_heads=2), (2, 4, 16, 16), 51),  # neither a power of two nor a multiple of 32
    # resnet_groups (vddp.py:586, 669): GroupNorm groups of every Block; 4 at dim 16 (4 ... 32 channels per group), 16 at the real widths
    "groups4": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                     per_frame_cond=True, cond_bias=True, resnet_groups=4), (2, 11, 16, 16), 11),
    "groups16w64": (dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                         per_frame_cond=True, cond_bias=True, resnet_groups=16), (1, 11, 16, 16), 11),
    # out_dim (vddp.py:576, 706-710): channels of final_conv.1; init_kernel_size (vddp.py:584, 621-626): the stem's (1, k, k) kernel
    "outdim5": (dict(dim=16, channels=3, out_dim=5, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                     per_frame_cond=True, cond_bias=True), (2, 11, 16, 16), 11),
    "k5": (dict(dim=16, channels=3, init_kernel_size=5, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                per_frame_cond=True, cond_bias=True), (2, 11, 16, 16), 11),
    "k3": (dict(dim=16, channels=1, init_kernel_size=3, padding_mode="circular"), (2, 4, 16, 16), 51),
    "k9w64": (dict(dim=64, channels=3, init_kernel_size=9, init_dim=64, cond_attention="self-stacked", cond_attention_tokens=16,
                   use_temporal_attention_cond=True, per_frame_cond=True, cond_bias=True), (1, 11, 16, 16), 11),
    # cross-attention on GRU tokens (vddp.py:546-549 + 354-363): one token per sample of the conditioning signal -- 40 here, 51 for the stress-strain curves --,
    # spatial sites only (the temporal sites' positional bias needs tokens == frames): more tokens than the softmax cross-attention kernel held until round 6
    "crossgru16": (dict(dim=16, channels=3, cond_attention="cross-attention", cond_attention_tokens=40, use_temporal_attention_cond=False, per_frame_cond=False,
                        cond_att_GRU=True), (2, 5, 16, 16), 40),
    # channels (vddp.py:576, 624; main.py:63 passes len(selected_channels)): more than the four of an RGBA GIF -- the stem kernel's K order holds four
    # channels per tap, wider inputs take the generic implicit GEMM over rows padded to a multiple of four; Lagrangian wiring at dim 16, defaults at the
    # real widths (where four or fewer channels would take the stem kernel)
    "chan6": (dict(dim=16, channels=6, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                   per_frame_cond=True, cond_bias=True), (2, 11, 16, 16), 11),
    "chan5w64": (dict(dim=64, channels=5), (1, 4, 16, 16), 51),
    # all of them at once (init_dim given explicitly: the reference's final_conv reads cat(x, r) as 2 * dim channels, so init_dim == dim is the only
    # value its forward accepts, vddp.py:706, 820): the configuration of the golden GRADIENT set (tests/golden/grads_ctor16.npz)
    "ctor16": (dict(dim=16, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                    per_frame_cond=True, cond_bias=True, attn_heads=4, attn_dim_head=16, resnet_groups=4, init_kernel_size=5, init_dim=16),
               (2, 11, 16, 16), 11),
}

# configurations whose golden file also holds the reference's parameter gradients of the l1 training loss (make_golden.py: gradient_goldens)
GRADIENT_CONFIGS = ("ctor16", "dh64")


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


# ---------------------------------------------------------------- geometry extraction
GEOMETRY_CASES = {  # name -> (seed, N, T, P, zero_u_2)
    "lagr96": (11, 3, 11, 96, -0.2137),
    "lagr32": (12, 8, 11, 32, 0.1),
    "lagr32b": (13, 8, 5, 32, -0.6),
    "single32": (14, 4, 1, 32, 0.0),  # num_frames == 1 -> the first-frame / first-channel rule even for 'lagrangian'
}


def synth_geometry_videos(seed: int, N: int, T: int, P: int, zero_u_2: float) -> torch.Tensor:
    """Sampler-like output (N, 3, T, P, P) whose u_2 channel encodes a blobby material / void pattern with speckle, thin bridges and
    equal-sized islands (ties), so that every rule of the extraction is exercised.  Deterministic in its arguments."""
    import torch.nn.functional as F
    g = torch.Generator().manual_seed(seed)
    v = torch.rand(N, 3, T, P, P, generator=g)
    cells = max(2, P // 8)
    mask = F.interpolate((torch.rand(N, 1, cells, cells, generator=g) > 0.45).float(), size=(P, P), mode="nearest")[:, 0].bool()
    mask ^= torch.rand(N, P, P, generator=g) > 0.93            # speckle: isolated pixels and pin holes
    mask[:, :: max(3, P // 6), :] |= torch.rand(N, 1, P, generator=g) > 0.5   # thin horizontal runs (components with axis-1 edges only)
    if N > 1:  # sample 1: islands of identical size (tie between components) in both quarters the two rules look at
        mask[1] = False
        for r0 in (0, P // 2):
            mask[1, r0 + 2:r0 + 4, 1:4] = True
            mask[1, r0 + 9:r0 + 12, 9:11] = True
            mask[1, r0 + 6, 13:15] = True
    dev = (torch.rand(N, T, P, P, generator=g) - 0.5) * 0.03    # void: |u_2 - zero| < 0.015 in every frame
    frame = torch.randint(0, T, (N, P, P), generator=g)
    bump = (0.03 + 0.1 * torch.rand(N, P, P, generator=g)) * torch.where(torch.rand(N, P, P, generator=g) > 0.5, 1.0, -1.0)
    onehot = F.one_hot(frame, T).permute(0, 3, 1, 2).float()
    v[:, 1] = zero_u_2 + dev + mask[:, None].float() * onehot * bump[:, None]   # material: one frame leaves the band
    v[:, 0, 0] = torch.where(mask, 0.55 + 0.4 * torch.rand(N, P, P, generator=g), 0.45 * torch.rand(N, P, P, generator=g))
    return v


# ---------------------------------------------------------------- training-sample assembly
DATASET_CASES = {  # name -> (seed, reference_frame, N samples, frames in the GIFs, H = W, num_frames, selected_channels, per_frame_cond)
    "lagr": (21, "lagrangian", 5, 11, 24, 11, [0, 1, 3], True),
    "lagr_pad": (22, "lagrangian", 3, 7, 16, 11, [0, 1, 2, 3], False),     # fewer frames than asked: zero-padded
    "lagr_crop": (23, "lagrangian", 3, 11, 20, 6, [1, 3], True),           # more frames than asked: cropped
    "euler": (24, "eulerian", 4, 11, 24, 11, [0, 1, 2, 3], True),
    "single": (25, "lagrangian", 4, 1, 32, 1, [0, 1, 2, 3], False),        # one-frame ablation: topology + sigma_22 whatever was selected
}


def synth_dataset(seed: int, frame: str, N: int, f: int, P: int):
    """Synthetic stand-in for a dataset folder: decoded GIF frames (N, n_fields, f, P, P) uint8 in oracle.dataset_oracle.FIELDS order
    (topology binary 0 / 255 with some grey pixels, fields using the whole byte range and exact zeros), frame_range_data.csv (N, 8)
    float64 with mixed signs, stress_strain_data.csv (N, 52) float64.  Deterministic in its arguments."""
    import numpy as np
    rng = np.random.default_rng(seed)
    nf = 5 if frame == "lagrangian" else 4
    frames = rng.integers(0, 256, size=(N, nf, f, P, P), dtype=np.uint8)
    topo = (rng.random((N, 1, P, P)) > 0.4).astype(np.uint8) * 255
    topo[rng.random((N, 1, P, P)) > 0.97] = 128
    frames[:, 0] = np.broadcast_to(topo, (N, f, P, P))
    frames[:, 1:][rng.random((N, nf - 1, f, P, P)) > 0.9] = 0
    fr = np.zeros((N, 8))
    if frame == "lagrangian":  # min_u_1, max_u_1, min_u_2, max_u_2, max_s_mises, min_s_22, max_s_22, max_strain_energy
        fr[:, 0], fr[:, 1] = -rng.random(N) * 0.3, rng.random(N) * 0.25
        fr[:, 2], fr[:, 3] = -rng.random(N) * 0.21, rng.random(N) * 0.013
        fr[:, 4] = rng.random(N) * 311.7 + 5
        fr[:, 5], fr[:, 6] = -rng.random(N) * 123.4 - 1, rng.random(N) * 88.8 + 1
        fr[:, 7] = rng.random(N) * 3.3
    else:                      # max_s_mises, min_s_22, max_s_22, max_strain_energy
        fr[:, 0] = rng.random(N) * 311.7 + 5
        fr[:, 1], fr[:, 2] = -rng.random(N) * 123.4 - 1, rng.random(N) * 88.8 + 1
        fr[:, 3] = rng.random(N) * 3.3
    curves = np.cumsum(rng.random((N, 52)) * 0.7 - 0.2, axis=1)
    curves[:, 0] = 0.0
    return frames, fr, curves
