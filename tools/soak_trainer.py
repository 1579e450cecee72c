"""Soak of the training step: n optimiser steps of the Lagrangian configuration on a fixed synthetic batch (the loss must stay finite and
fall), in either arithmetic mode.

    python tools/soak_trainer.py [fp32|bf16x3|bf16|fp16] [n]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402
from videometamaterials_amd.dp import DataParallelTrainer  # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else "fp32"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev)
model.train_precision = mode
model.train()
diff = vm.GaussianDiffusion(model, image_size=96, num_frames=11, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                            sampling_timesteps=256).to(dev)
tr = DataParallelTrainer(diff, train_lr=1e-4)
g = torch.Generator().manual_seed(7)
x = torch.rand(4, 3, 11, 96, 96, generator=g).to(dev)
cond = (torch.rand(4, 11, generator=g) * 2 - 1).to(dev)
losses = []
for i in range(n):
    losses.append(float(tr.train_step(x, cond)))
    if i % 10 == 0 or i == n - 1:
        print(f"step {i}: loss {losses[-1]:.4f}", flush=True)
first, last = sum(losses[:5]) / 5, sum(losses[-5:]) / 5
ok = all(map(lambda v: v == v and abs(v) < 1e6, losses)) and last < first
print(f"mean loss first 5 steps {first:.4f} -> last 5 steps {last:.4f}: {'ok' if ok else 'FAILED'}")
if mode == "fp16":
    print("loss scale state:", tr.loss_scale_state())
sys.exit(0 if ok else 1)
