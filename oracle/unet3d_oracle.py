"""ORACLE (test infrastructure, not product code).

CPU fp32 restatement of the reference denoiser `Unet3D`
(/root/reference/denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py,
abbreviated vddp.py below) as pure functions over a flat ``{name: tensor}``
state dict that uses the reference's parameter names.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this module; the shipped package never does.

Parity pinning: tests/golden/*.npz hold outputs of the *actual* reference
(imported in the build container through tools/ref_shims, script
tests/golden/make_golden.py) and tests/test_oracle_golden.py checks this file
against them.  The rotary embedding comes from the un-vendored third-party
package rotary_embedding_torch>=0.2.3 (reference README.md:75); it is restated
from the published RoFormer formula => "parity unpinned" at that one boundary
(pinned to the formula only).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- config
@dataclass
class UnetCfg:
    """Mirrors the keyword arguments of Unet3D.__init__ (vddp.py:575-595)."""

    dim: int = 64
    out_dim: Optional[int] = None
    dim_mults: Tuple[int, ...] = (1, 2, 4, 8)
    channels: int = 3
    attn_heads: int = 8
    attn_dim_head: int = 32
    init_dim: Optional[int] = None
    init_kernel_size: int = 7
    use_sparse_linear_attn: bool = True
    resnet_groups: int = 8
    cond_bias: bool = False
    cond_attention: str = "none"
    cond_attention_tokens: int = 6
    cond_att_GRU: bool = False
    use_temporal_attention_cond: bool = False
    cond_to_time: str = "add"
    per_frame_cond: bool = False
    padding_mode: str = "zeros"
    # derived (vddp.py:599-608)
    time_dim: int = field(init=False)
    cond_dim: int = field(init=False)

    def __post_init__(self):
        self.time_dim = self.dim * 4
        self.cond_dim = self.time_dim
        if self.per_frame_cond:  # vddp.py:602-603
            self.cond_attention = "self-stacked"
            self.cond_attention_tokens = 11
        if self.init_dim is None:
            self.init_dim = self.dim
        if self.out_dim is None:
            self.out_dim = self.channels

    @property
    def level_io(self):
        dims = [self.init_dim] + [self.dim * m for m in self.dim_mults]
        return list(zip(dims[:-1], dims[1:]))


# --------------------------------------------------------------------------- integer tables
def rel_pos_bucket_table(n: int, num_buckets: int = 32, max_distance: int = 32) -> Tensor:
    """T5 bidirectional bucket index for rel = k - q (vddp.py:82-106). INT64 (n,n)."""
    q = torch.arange(n, dtype=torch.long)[:, None]
    k = torch.arange(n, dtype=torch.long)[None, :]
    neg = -(k - q)  # vddp.py:85
    half = num_buckets // 2
    out = (neg < 0).long() * half
    dist = neg.abs()
    max_exact = half // 2
    large = max_exact + (
        torch.log(dist.float() / max_exact) / math.log(max_distance / max_exact) * (half - max_exact)
    ).long()
    large = torch.clamp(large, max=half - 1)
    return out + torch.where(dist < max_exact, dist, large)


def rel_pos_bias(sd: Dict[str, Tensor], n: int) -> Tensor:
    """(heads, n, n) bias, vddp.py:102-108 with max_distance=32 (vddp.py:617)."""
    emb = sd["time_rel_pos_bias.relative_attention_bias.weight"]  # (32, heads)
    return emb[rel_pos_bucket_table(n)].permute(2, 0, 1).contiguous()


# --------------------------------------------------------------------------- small pieces
def sinusoidal_embedding(time: Tensor, dim: int) -> Tensor:
    """vddp.py:139-151."""
    half = dim // 2
    step = math.log(10000) / (half - 1)
    freqs = torch.exp(torch.arange(half) * -step)
    arg = time[:, None] * freqs[None, :]
    return torch.cat((arg.sin(), arg.cos()), dim=-1)


def rotary_rotate(t: Tensor, freqs: Optional[Tensor] = None) -> Tensor:
    """Interleaved-pair RoPE along dim -2, positions 0..n-1.  The reference builds ONE RotaryEmbedding(min(32, attn_dim_head))
    (vddp.py:612): the leading rot = min(32, d) features of every head are rotated with frequencies theta^(-2i/rot), the features beyond them pass
    through unchanged (rotary_embedding_torch's apply_rotary_emb with start_index = 0)."""
    d = t.shape[-1]
    rot = min(32, d)
    if freqs is None:
        freqs = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
    pos = torch.arange(t.shape[-2]).float()
    ang = (pos[:, None] * freqs[None, :]).repeat_interleave(2, dim=-1)
    head, tail = t[..., :rot], t[..., rot:]
    pairs = head.reshape(*head.shape[:-1], rot // 2, 2)
    swapped = torch.stack((-pairs[..., 1], pairs[..., 0]), dim=-1).reshape(head.shape)
    head = head * ang.cos() + swapped * ang.sin()
    return torch.cat((head, tail), dim=-1) if tail.shape[-1] else head


def _pad_frames(x2: Tensor, pad: int, mode: str) -> Tensor:
    """The explicit padding of the periodic variants (vddp.py:163-237): 'circular' wraps both image axes (nn.Conv3d padding_mode='circular'
    / CircularUpsample), 'circular_1d' wraps the horizontal axis and zero-pads the vertical one (Circular_1d_Conv3d / Circular_1d_Upsample)."""
    if pad == 0:
        return x2
    if mode == "circular":
        return F.pad(x2, (pad, pad, pad, pad), mode="circular")
    x2 = F.pad(x2, (pad, pad, 0, 0), mode="circular")
    return F.pad(x2, (0, 0, pad, pad), mode="constant")


def frame_conv(x: Tensor, w: Tensor, b: Optional[Tensor], stride: int = 1, pad: int = 0, mode: str = "zeros") -> Tensor:
    """Conv3d with a (1,k,k) kernel == per-frame 2-D conv (vddp.py:241,271,297,626); mode: the layer's padding_mode."""
    B, C, T, H, W = x.shape
    x2 = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    if mode == "zeros":
        y = F.conv2d(x2, w[:, :, 0], b, stride=stride, padding=pad)
    else:
        y = F.conv2d(_pad_frames(x2, pad, mode), w[:, :, 0], b, stride=stride, padding=0)
    return y.reshape(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def frame_conv_transpose(x: Tensor, w: Tensor, b: Tensor, mode: str = "zeros") -> Tensor:
    """ConvTranspose3d (1,4,4)/(1,2,2)/(0,1,1) (vddp.py:155); periodic variants (vddp.py:163-213): the input is padded by
    true_padding = k - 1 - p = 2 explicitly and the transposed convolution crops removed_padding = (k - 1) + s + p - 1 = 5."""
    B, C, T, H, W = x.shape
    x2 = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H, W)
    if mode == "zeros":
        y = F.conv_transpose2d(x2, w[:, :, 0], b, stride=2, padding=1)
    else:
        y = F.conv_transpose2d(_pad_frames(x2, 2, mode), w[:, :, 0], b, stride=2, padding=5)
    return y.reshape(B, T, *y.shape[1:]).permute(0, 2, 1, 3, 4)


def ref_key(mode: str, k: str) -> str:
    """state_dict name of a convolution parameter under the layer's padding_mode: the periodic variants wrap their convolutions in
    helper modules (Circular_1d_Conv3d.conv, Circular*Upsample.conv_transpose; vddp.py:153-243, 268-273, 625-629)."""
    import re
    if mode in ("circular", "circular_1d"):
        k = re.sub(r"^(ups\.\d+\.4)\.(weight|bias)$", r"\1.conv_transpose.\2", k)
    if mode == "circular_1d":
        k = re.sub(r"^(.*\.block[12]\.proj|init_conv|downs\.\d+\.4)\.(weight|bias)$", r"\1.conv.\2", k)
    return k


def channel_layernorm(x: Tensor, gamma: Tensor, eps: float = 1e-5) -> Tensor:
    """vddp.py:245-254: biased variance over channel axis, eps inside sqrt, gamma only."""
    mean = x.mean(dim=1, keepdim=True)
    var = x.var(dim=1, unbiased=False, keepdim=True)
    return (x - mean) / torch.sqrt(var + eps) * gamma


def conv_gn_act(sd, p: str, x: Tensor, groups: int, scale_shift=None, mode: str = "zeros") -> Tensor:
    """Block (vddp.py:267-285): conv3x3 -> GroupNorm -> optional FiLM -> SiLU."""
    y = frame_conv(x, sd[ref_key(mode, p + ".proj.weight")], sd[ref_key(mode, p + ".proj.bias")], pad=1, mode=mode)
    y = F.group_norm(y, groups, sd[p + ".norm.weight"], sd[p + ".norm.bias"], eps=1e-5)
    if scale_shift is not None:
        s, sh = scale_shift
        y = y * (s + 1) + sh
    return F.silu(y)


def resnet_block(sd, p: str, x: Tensor, temb: Optional[Tensor], groups: int, mode: str = "zeros") -> Tensor:
    """ResnetBlock (vddp.py:287-311); only block1 receives scale/shift."""
    ss = None
    if (p + ".mlp.1.weight") in sd:
        e = F.linear(F.silu(temb), sd[p + ".mlp.1.weight"], sd[p + ".mlp.1.bias"])
        e = e[:, :, None, None, None]
        ss = e.chunk(2, dim=1)
    h = conv_gn_act(sd, p + ".block1", x, groups, ss, mode)
    h = conv_gn_act(sd, p + ".block2", h, groups, None, mode)
    if (p + ".res_conv.weight") in sd:
        x = frame_conv(x, sd[p + ".res_conv.weight"], sd[p + ".res_conv.bias"])
    return h + x


# --------------------------------------------------------------------------- attention
def linear_attention(sd, p: str, x: Tensor, tokens: Optional[Tensor], cfg: UnetCfg) -> Tensor:
    """SpatialLinearAttention (vddp.py:313-378); heads=cfg.attn_heads, dim_head=32 (default),
    per_frame_cond is *not* forwarded by Unet3D (vddp.py:679) -> every frame sees all tokens."""
    B, C, T, H, W = x.shape
    heads, dh = cfg.attn_heads, 32
    xf = x.permute(0, 2, 1, 3, 4).reshape(B * T, C, H * W)
    qkv = torch.einsum("oc,bcn->bon", sd[p + ".to_qkv.weight"][:, :, 0, 0], xf)
    q, k, v = (t.reshape(B * T, heads, dh, H * W) for t in qkv.chunk(3, dim=1))
    if cfg.cond_attention == "self-stacked" and tokens is not None:
        ek = F.linear(tokens, sd[p + ".to_k.weight"])  # (B, n, heads*dh)
        ev = F.linear(tokens, sd[p + ".to_v.weight"])
        n_tok = ek.shape[1]

        def per_frame(t):
            t = t.reshape(B, 1, n_tok, heads, dh).expand(B, T, n_tok, heads, dh)
            return t.permute(0, 1, 3, 4, 2).reshape(B * T, heads, dh, n_tok)

        k = torch.cat([per_frame(ek), k], dim=-1)
        v = torch.cat([per_frame(ev), v], dim=-1)
    elif cfg.cond_attention == "cross-attention" and tokens is not None:
        # vddp.py:354-363: queries from their own projection (to_q), keys / values are the conditioning tokens alone, the same for every frame
        q = torch.einsum("oc,bcn->bon", sd[p + ".to_q.weight"][:, :, 0, 0], xf).reshape(B * T, heads, dh, H * W)
        ek = F.linear(tokens, sd[p + ".to_k.weight"])
        ev = F.linear(tokens, sd[p + ".to_v.weight"])
        n_tok = ek.shape[1]
        k, v = (t.reshape(B, 1, n_tok, heads, dh).expand(B, T, n_tok, heads, dh).permute(0, 1, 3, 4, 2).reshape(B * T, heads, dh, n_tok) for t in (ek, ev))
    elif cfg.cond_attention not in ("none", "self-stacked", "cross-attention"):
        raise ValueError("cond_attention must be none, self-stacked or cross-attention")
    q = q.softmax(dim=-2) * dh ** -0.5
    k = k.softmax(dim=-1)
    v = v / (H * W)
    ctx = torch.einsum("bhdn,bhen->bhde", k, v)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q).reshape(B * T, heads * dh, H * W)
    out = torch.einsum("oc,bcn->bon", sd[p + ".to_out.weight"][:, :, 0, 0], out) + sd[p + ".to_out.bias"][None, :, None]
    return out.reshape(B, T, C, H, W).permute(0, 2, 1, 3, 4)


def softmax_attention(
    sd,
    p: str,
    x: Tensor,
    cfg: UnetCfg,
    *,
    pos_bias: Optional[Tensor],
    tokens: Optional[Tensor],
    rotary: bool,
    dim_head: int,
    focus: Optional[Tensor] = None,
) -> Tensor:
    """Attention (vddp.py:396-535) on x of shape (b, b2, n, c).

    temporal use: b2 = pixels, n = frames, rotary=True, pos_bias given.
    mid spatial use: b2 = frames, n = pixels, rotary=False, pos_bias None.
    focus: focus_present_mask (b,) bool or None (vddp.py:431, 438-443, 514-524): samples that "focus on the present" attend to their own
    position only.  Inert on every shipped config (prob_focus_present = 0)."""
    b, b2, n, c = x.shape
    heads, dh = cfg.attn_heads, dim_head
    qkv = F.linear(x, sd[p + ".to_qkv.weight"])
    if focus is not None and (cfg.cond_attention == "none" or tokens is None) and bool(focus.all()):  # vddp.py:438-443
        return F.linear(qkv.chunk(3, dim=-1)[-1], sd[p + ".to_out.weight"])
    q, k, v = (t.reshape(b, b2, n, heads, dh).transpose(2, 3) for t in qkv.chunk(3, dim=-1))
    stacked = cfg.cond_attention == "self-stacked" and tokens is not None
    if rotary:
        k = rotary_rotate(k)
    if stacked:
        ek = F.linear(tokens, sd[p + ".to_k.weight"])
        ev = F.linear(tokens, sd[p + ".to_v.weight"])
        if pos_bias is None and cfg.per_frame_cond:  # vddp.py:459-462: one token per frame
            ek, ev = (t[:, :, None, :] for t in (ek, ev))
        else:  # vddp.py:465: all tokens for every b2
            ek, ev = (t[:, None].expand(b, b2, *t.shape[1:]) for t in (ek, ev))
        ek, ev = (t.reshape(b, b2, t.shape[2], heads, dh).transpose(2, 3) for t in (ek, ev))
        if rotary and cfg.per_frame_cond:  # vddp.py:470-471
            ek = rotary_rotate(ek)
        k = torch.cat([ek, k], dim=-2)
        v = torch.cat([ev, v], dim=-2)
    elif cfg.cond_attention == "cross-attention" and tokens is not None:
        # vddp.py:476-485: q = to_q(x); k, v = the tokens alone (every b2 sees all of them), keys not rotated; the positional bias below is
        # added to the (n x tokens) scores as it is, which only broadcasts when tokens == n
        q = F.linear(x, sd[p + ".to_q.weight"]).reshape(b, b2, n, heads, dh).transpose(2, 3)
        ek = F.linear(tokens, sd[p + ".to_k.weight"])
        ev = F.linear(tokens, sd[p + ".to_v.weight"])
        k, v = (t[:, None].expand(b, b2, *t.shape[1:]).reshape(b, b2, t.shape[1], heads, dh).transpose(2, 3) for t in (ek, ev))
    elif cfg.cond_attention not in ("none", "self-stacked", "cross-attention"):
        raise ValueError("cond_attention must be none, self-stacked or cross-attention")
    q = q * dh ** -0.5
    if rotary:
        q = rotary_rotate(q)
    sim = torch.einsum("...id,...jd->...ij", q, k)
    if pos_bias is not None:
        if stacked:
            sim[..., -n:] = sim[..., -n:] + pos_bias
            if cfg.per_frame_cond:
                sim[..., :n] = sim[..., :n] + pos_bias
        else:
            sim = sim + pos_bias
    if focus is not None and bool(focus.any()):  # vddp.py:514-524 (an (n x n) mask: shape-errors against stacked token keys, like the reference)
        eye = torch.eye(n, dtype=torch.bool)
        keep = torch.where(focus.reshape(b, 1, 1, 1, 1), eye.reshape(1, 1, 1, n, n), torch.ones(1, 1, 1, n, n, dtype=torch.bool))
        sim = sim.masked_fill(~keep, -torch.finfo(sim.dtype).max)
    sim = sim - sim.amax(dim=-1, keepdim=True)
    attn = sim.softmax(dim=-1)
    out = torch.einsum("...ij,...jd->...id", attn, v)
    out = out.transpose(2, 3).reshape(b, b2, n, heads * dh)
    return F.linear(out, sd[p + ".to_out.weight"])


def temporal_attention_block(sd, p: str, x: Tensor, cfg, pos_bias, tokens, focus: Optional[Tensor] = None) -> Tensor:
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b (h w) f c', Attention))) (vddp.py:615,630,680)."""
    B, C, T, H, W = x.shape
    y = channel_layernorm(x, sd[p + ".fn.norm.gamma"])
    y = y.permute(0, 3, 4, 2, 1).reshape(B, H * W, T, C)
    y = softmax_attention(sd, p + ".fn.fn.fn", y, cfg, pos_bias=pos_bias, tokens=tokens, rotary=True, dim_head=cfg.attn_dim_head, focus=focus)
    y = y.reshape(B, H, W, T, C).permute(0, 4, 3, 1, 2)
    return y + x


def mid_spatial_attention_block(sd, p: str, x: Tensor, cfg, tokens) -> Tensor:
    """Residual(PreNorm(EinopsToAndFrom('b c f h w','b f (h w) c', Attention))) (vddp.py:687-689)."""
    B, C, T, H, W = x.shape
    y = channel_layernorm(x, sd[p + ".fn.norm.gamma"])
    y = y.permute(0, 2, 3, 4, 1).reshape(B, T, H * W, C)
    y = softmax_attention(sd, p + ".fn.fn.fn", y, cfg, pos_bias=None, tokens=tokens, rotary=False, dim_head=32)  # dim_head not forwarded at vddp.py:687
    y = y.reshape(B, T, H, W, C).permute(0, 4, 1, 2, 3)
    return y + x


def linear_attention_block(sd, p: str, x: Tensor, cfg, tokens) -> Tensor:
    y = channel_layernorm(x, sd[p + ".fn.norm.gamma"])
    return linear_attention(sd, p + ".fn.fn", y, tokens, cfg) + x


# --------------------------------------------------------------------------- conditioning
def signal_cnn(sd, cond: Tensor) -> Tensor:
    """SignalEmbedding('CNN') (vddp.py:538-572): 5x [Conv1d(k4,s2,p1) + SiLU], then squeeze."""
    h = cond[:, None, :]
    for i in range(0, 10, 2):
        h = F.silu(F.conv1d(h, sd[f"sign_emb_CNN.emb_model.{i}.weight"], sd[f"sign_emb_CNN.emb_model.{i}.bias"], stride=2, padding=1))
    return torch.squeeze(h)


def signal_gru(sd, cond: Tensor) -> Tensor:
    """SignalEmbedding('GRU') (vddp.py:546-549, 567-571): the (B, L) signal as a length-L sequence of scalars through nn.GRU(1, cond_dim, num_layers=3,
    batch_first=True); every time step's top-layer state is a token -> (B, L, cond_dim).  nn.GRU's cell (torch documentation), gates in the order
    (r, z, n) along the 3 H rows of weight_ih / weight_hh:
        r = sigmoid(W_ir x + b_ir + W_hr h + b_hr),  z = sigmoid(W_iz x + b_iz + W_hz h + b_hz),
        n = tanh(W_in x + b_in + r * (W_hn h + b_hn)),  h' = (1 - z) * n + z * h,   h_0 = 0."""
    x = cond[..., None]  # (B, L, 1)
    B, L, _ = x.shape
    for layer in range(3):
        pfx = f"sign_emb_GRU.emb_model."
        w_ih, w_hh = sd[pfx + f"weight_ih_l{layer}"], sd[pfx + f"weight_hh_l{layer}"]
        b_ih, b_hh = sd[pfx + f"bias_ih_l{layer}"], sd[pfx + f"bias_hh_l{layer}"]
        H = w_hh.shape[1]
        gi_all = F.linear(x, w_ih, b_ih)  # (B, L, 3H)
        h = torch.zeros(B, H, dtype=x.dtype)
        outs = []
        for t in range(L):
            gi, gh = gi_all[:, t], F.linear(h, w_hh, b_hh)
            r = torch.sigmoid(gi[:, :H] + gh[:, :H])
            z = torch.sigmoid(gi[:, H:2 * H] + gh[:, H:2 * H])
            n = torch.tanh(gi[:, 2 * H:] + r * gh[:, 2 * H:])
            h = (1 - z) * n + z * h
            outs.append(h)
        x = torch.stack(outs, dim=1)
    return x


def embed_condition(sd, cfg: UnetCfg, cond: Tensor, null_mask: Tensor):
    """vddp.py:751-795. Returns (tokens or None, hidden (B,time_dim))."""
    if cfg.per_frame_cond:
        tokens = F.linear(cond[..., None], sd["sign_emb.weight"], sd["sign_emb.bias"])
        pooled = tokens.mean(dim=-2)
        h = F.layer_norm(pooled, (cfg.cond_dim,), sd["cond_token_to_hidden.0.weight"], sd["cond_token_to_hidden.0.bias"])
        h = F.linear(h, sd["cond_token_to_hidden.1.weight"], sd["cond_token_to_hidden.1.bias"])
        hidden = F.linear(F.silu(h), sd["cond_token_to_hidden.3.weight"], sd["cond_token_to_hidden.3.bias"])
    else:
        hidden = signal_cnn(sd, cond)
        tokens = None
        if cfg.cond_attention != "none":
            if cfg.cond_att_GRU:
                tokens = signal_gru(sd, cond)  # vddp.py:769-770: one token per sample of the conditioning signal (cond_attention_tokens == its length)
            else:
                tokens = hidden[:, None, :].expand(-1, cfg.cond_attention_tokens, -1)
    if cfg.cond_attention != "none":
        tokens = torch.where(null_mask[:, None, None], sd["null_text_token"], tokens)
    hidden = torch.where(null_mask[:, None], sd["null_text_hidden"], hidden)
    return tokens, hidden


# --------------------------------------------------------------------------- the network
def unet3d_forward(sd: Dict[str, Tensor], cfg: UnetCfg, x: Tensor, time: Tensor, cond: Tensor, null_mask: Tensor,
                   taps: Optional[Dict[str, Tensor]] = None, focus: Optional[Tensor] = None) -> Tensor:
    """Unet3D.forward (vddp.py:730-821) with the CFG drop mask passed explicitly
    (null_cond_prob=0 -> all False, =1 -> all True; vddp.py:55-61).
    focus: focus_present_mask (b,) bool -- reaches every temporal attention except init_temporal_attn (vddp.py:743, 803, 809, 817).
    `taps`, if given, receives the output of every block (debug aid for the per-block GPU parity tests)."""

    def tap(name, v):
        if taps is not None:
            taps[name] = v.detach().clone()
        return v

    pm = cfg.padding_mode
    if pm not in ("zeros", "circular", "circular_1d"):
        raise ValueError(f"padding_mode {pm!r}")
    g = cfg.resnet_groups
    T = x.shape[2]
    bias = rel_pos_bias(sd, T)
    x = frame_conv(x, sd[ref_key(pm, "init_conv.weight")], sd[ref_key(pm, "init_conv.bias")], pad=cfg.init_kernel_size // 2, mode=pm)
    x = tap("init_temporal_attn", temporal_attention_block(sd, "init_temporal_attn", x, cfg, bias, None))
    r = x.clone()
    t = sinusoidal_embedding(time, cfg.dim)
    t = F.linear(t, sd["time_mlp.1.weight"], sd["time_mlp.1.bias"])
    t = F.linear(F.gelu(t), sd["time_mlp.3.weight"], sd["time_mlp.3.bias"])
    tokens, hidden = embed_condition(sd, cfg, cond, null_mask)
    t = t + hidden if cfg.cond_to_time == "add" else torch.cat((t, hidden), dim=-1)
    tokens_t = tokens if cfg.use_temporal_attention_cond else None

    skips = []
    n_lvl = len(cfg.level_io)
    for i in range(n_lvl):
        x = tap(f"downs.{i}.0", resnet_block(sd, f"downs.{i}.0", x, t, g, pm))
        x = tap(f"downs.{i}.1", resnet_block(sd, f"downs.{i}.1", x, t, g, pm))
        x = tap(f"downs.{i}.2", linear_attention_block(sd, f"downs.{i}.2", x, cfg, tokens))
        x = tap(f"downs.{i}.3", temporal_attention_block(sd, f"downs.{i}.3", x, cfg, bias, tokens_t, focus))
        skips.append(x)
        if i < n_lvl - 1:
            x = frame_conv(x, sd[ref_key(pm, f"downs.{i}.4.weight")], sd[ref_key(pm, f"downs.{i}.4.bias")], stride=2, pad=1, mode=pm)
    x = tap("mid_block1", resnet_block(sd, "mid_block1", x, t, g, pm))
    x = tap("mid_spatial_attn", mid_spatial_attention_block(sd, "mid_spatial_attn", x, cfg, tokens))
    x = tap("mid_temporal_attn", temporal_attention_block(sd, "mid_temporal_attn", x, cfg, bias, tokens_t, focus))
    x = tap("mid_block2", resnet_block(sd, "mid_block2", x, t, g, pm))
    for i in range(n_lvl):
        x = torch.cat((x, skips.pop()), dim=1)
        x = tap(f"ups.{i}.0", resnet_block(sd, f"ups.{i}.0", x, t, g, pm))
        x = tap(f"ups.{i}.1", resnet_block(sd, f"ups.{i}.1", x, t, g, pm))
        x = tap(f"ups.{i}.2", linear_attention_block(sd, f"ups.{i}.2", x, cfg, tokens))
        x = tap(f"ups.{i}.3", temporal_attention_block(sd, f"ups.{i}.3", x, cfg, bias, tokens_t, focus))
        if i < n_lvl - 1:
            x = frame_conv_transpose(x, sd[ref_key(pm, f"ups.{i}.4.weight")], sd[ref_key(pm, f"ups.{i}.4.bias")], mode=pm)
    x = torch.cat((x, r), dim=1)
    x = tap("final_conv.0", resnet_block(sd, "final_conv.0", x, None, g, pm))
    return frame_conv(x, sd["final_conv.1.weight"], sd["final_conv.1.bias"])


def unet3d_guided(sd, cfg, x, time, cond, guidance_scale: float = 5.0) -> Tensor:
    """forward_with_guidance_scale (vddp.py:715-728): w==1 -> conditional only; otherwise
    null + (cond - null) * w (two forwards, also for w == 0)."""
    B = x.shape[0]
    eps_c = unet3d_forward(sd, cfg, x, time, cond, torch.zeros(B, dtype=torch.bool))
    if guidance_scale == 1:
        return eps_c
    eps_n = unet3d_forward(sd, cfg, x, time, cond, torch.ones(B, dtype=torch.bool))
    return eps_n + (eps_c - eps_n) * guidance_scale
