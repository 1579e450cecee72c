"""Test-tooling stand-in for the public `rotary_embedding_torch` package
(>=0.2.3, absent offline).  Restates the published RoFormer formula used by
that package with its defaults (freqs_for='lang', theta=10000, interleaved
pairs).  Parity at this boundary is pinned to the formula only.
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
        return t * ang.cos() + rotate_half(t) * ang.sin()
