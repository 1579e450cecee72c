"""videometamaterials_amd -- MI355X (gfx950) native hot path of jhbastek/VideoMetamaterials.

Drop-in for ``from denoising_diffusion_pytorch import Unet3D, GaussianDiffusion``
(reference main.py:6): same constructor / forward signatures and state_dict keys,
arithmetic in hand-written HIP (libvmm_hip.so, C ABI in include/vmm_kernels.h).
"""
from .unet3d import Unet3D  # noqa: F401
from .diffusion import GaussianDiffusion  # noqa: F401
from . import hostmath  # noqa: F401
from .geometry import extract_geometries  # noqa: F401
from .dataset import Dataset  # noqa: F401

__all__ = ["Unet3D", "GaussianDiffusion", "hostmath", "extract_geometries", "Dataset"]
