"""Test-tooling stand-in for the public `einops_exts` package (absent offline).

Only the two helpers the reference imports are provided; both are thin
wrappers over einops.rearrange, matching the published package's semantics.
"""
from einops import rearrange


def check_shape(tensor, pattern, **kwargs):
    return rearrange(tensor, f"{pattern} -> {pattern}", **kwargs)


def rearrange_many(tensors, pattern, **kwargs):
    return tuple(rearrange(t, pattern, **kwargs) for t in tensors)
