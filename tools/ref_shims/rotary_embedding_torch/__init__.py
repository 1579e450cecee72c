"""Test-tooling stand-in for the public `rotary_embedding_torch` package
(>=0.2.3, absent offline).  Restates the published RoFormer formula used by
that package with its defaults (freqs_for='lang', theta=10000, interleaved
pairs).  Parity at this boundary is pinned to the formula only.

`RotaryEmbedding(dim)` holds dim/2 frequencies; `rotate_queries_or_keys` is the
package's `apply_rotary_emb(freqs, t, start_index=0)`: the LEADING `dim`
features of the last axis are rotated, features beyond `dim` pass through
unchanged (the partial-rotary case the reference reaches with
attn_dim_head > 32: `RotaryEmbedding(min(32, attn_dim_head))`, vddp.py:612).
"""
import torch
from torch import nn


def rotate_half(x):
    x = x.reshape(*x.shape[:-1], -1, 2)
    x1, x2 = x.unbind(-1)
    return torch.stack((-x2, x1), dim=-1).flatten(-2)


class RotaryEmbedding(nn.Module):
    def __init__(self, dim, theta=10000):
        super().__init__()
        freqs = 1.0 / (theta ** (torch.arange(0, dim, 2)[: dim // 2].float() / dim))
        self.freqs = nn.Parameter(freqs, requires_grad=False)

    def rotate_queries_or_keys(self, t, seq_dim=-2):
        n = t.shape[seq_dim]
        pos = torch.arange(n, device=t.device).type(self.freqs.dtype)
        ang = torch.einsum("n,f->nf", pos, self.freqs)
        ang = ang.repeat_interleave(2, dim=-1).to(t)
        rot_dim = ang.shape[-1]
        assert rot_dim <= t.shape[-1], "feature dimension is too small to rotate all the positions"
        t_rot, t_pass = t[..., :rot_dim], t[..., rot_dim:]
        t_rot = t_rot * ang.cos() + rotate_half(t_rot) * ang.sin()
        return torch.cat((t_rot, t_pass), dim=-1)
