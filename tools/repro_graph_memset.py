"""Stand-alone check of the captured sampling step across several full samples (the scenario in which a hipMemsetAsync NODE inside the
captured graph was observed to run unordered with its neighbouring kernels on this ROCm stack):

    python tools/repro_graph_memset.py [n_samples]

One _GraphedStep is reused for n full 256-step samples with new noise / conditioning written between the replays and no host sync
inside the loops.  Before the GroupNorm sums were zeroed by a kernel (norm.hip, vmm_groupnorm_stats) about three runs in four printed
`mean 0.0000` from the second sample on: a non-finite denoiser output at the first replay of that sample, which the x0 clamp turns
into -1 everywhere.  The failure was timing dependent (extra kernels or a host sync after the first replay of a sample hid it).
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402
from videometamaterials_amd.diffusion import _GraphedStep  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev).eval()
diff = vm.GaussianDiffusion(model, image_size=96, num_frames=11, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                            sampling_timesteps=256).to(dev)
shape = (4, 3, 11, 96, 96)
st = _GraphedStep(diff, shape, 11, 5.0)
bad = 0
with torch.inference_mode():
    for it in range(n):
        st.set_cond(torch.rand(4, 11, device=dev) * 2 - 1)
        img = torch.randn(shape, device=dev)
        for t in reversed(range(256)):
            img = st(img, t)
        torch.cuda.synchronize()
        mean = float(((img + 1) * 0.5).mean())
        bad += not (0.2 < mean < 0.8)
        print(f"sample {it}: finite {bool(torch.isfinite(img).all())} mean {mean:.4f}", flush=True)
print("FAILED" if bad else "ok")
sys.exit(1 if bad else 0)
