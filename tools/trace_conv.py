"""Measurement aid: one eager denoiser forward at the bench shape, so that VMM_PW_TRACE=<k> prints the barrier time line of the k-th
persistent 3x3 launch (network order: 0 = downs.0.0.block1, 1 = downs.0.0.block2, ..., 16 = mid_block1.block1)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402

dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev).eval()
B = 2 * bench.B_PER_GPU
x = torch.randn(B, 3, bench.T, bench.HW, bench.HW, device=dev)
t = torch.randint(0, 256, (B,), device=dev)
cond = torch.rand(B, 11, device=dev) * 2 - 1
with torch.no_grad():
    model(x, t, cond=cond, null_cond_prob=0.0)
torch.cuda.synchronize()
