"""Soak of the public sampling entry point: n full 256-step guided samples through GaussianDiffusion.sample() (captured step reused
across calls), each checked for finiteness and for the gross statistics of a healthy sample.

    python tools/soak_sampler.py [n]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev).eval()
diff = vm.GaussianDiffusion(model, image_size=96, num_frames=11, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                            sampling_timesteps=256).to(dev)
bad = 0
for it in range(n):
    cond = torch.rand(4, 11, device=dev) * 2 - 1
    out = diff.sample(cond=cond, guidance_scale=5.0)
    torch.cuda.synchronize()
    mean = float(out.mean())
    ok = bool(torch.isfinite(out).all()) and 0.3 < mean < 0.8
    bad += not ok
    if not ok or it % 10 == 0:
        print(f"sample {it}: mean {mean:.4f} {'ok' if ok else 'SUSPICIOUS'}", flush=True)
print(f"{n - bad} of {n} samples healthy")
sys.exit(1 if bad else 0)
