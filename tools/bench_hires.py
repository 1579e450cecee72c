"""configs[3] (BASELINE.json): 22 frames x 192 x 192, batch 8 per GPU -- timing of the denoiser forward and of one guided DDPM step on one MI355X
(random-init weights of the Lagrangian widths with per_frame_cond = False, i.e. tests/test_gpu_hires.py's model; fp32 storage, split-bf16 arithmetic).
   python tools/bench_hires.py [batch] [precision = bf16x3 | bf16 | fp32] [fp32store: the bf16 arithmetic on fp32-stored feature maps]"""
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import videometamaterials_amd as vm  # noqa: E402
from test_gpu_hires import KW_HIRES  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
    T, H = 22, 192
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    m = vm.Unet3D(**KW_HIRES).to(dev).eval()
    if len(sys.argv) > 2:
        m.precision = sys.argv[2]
    if len(sys.argv) > 3 and sys.argv[3] == "fp32store":
        m.bf16_storage = False
    diff = vm.GaussianDiffusion(m, image_size=H, num_frames=T, channels=3, timesteps=256, use_dynamic_thres=True, sampling_timesteps=256).to(dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 3, T, H, H, generator=g).to(dev)
    t = torch.randint(0, 256, (B,), generator=g).to(dev)
    cond = (torch.rand(B, 51, generator=g) * 2 - 1).to(dev)

    def timed(fn, n):
        fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n

    with torch.no_grad():
        fwd = timed(lambda: m(x, t, cond=cond, null_cond_prob=0.0), 5)
        step = timed(lambda: diff.p_sample(x, t, cond=cond, guidance_scale=5.0), 3)
    plan = m.get_plan(B, T, H, H, 51, dev)
    fam = {}
    for (name, flops, _), ms in zip(plan.meta, plan.launch_timed()):
        f = fam.setdefault(name, [0.0, 0.0, 0])
        f[0] += ms; f[1] += flops; f[2] += 1
    out = {"workload": f"configs[3]: 22x192x192, batch {B}, dim 64, random init", "precision": m.precision, "bf16_storage": bool(m.precision == "bf16" and getattr(m, "bf16_storage", False)), "denoiser_forward_ms": round(fwd, 2),
           "guided_step_ms": round(step, 2), "sampled_frames_per_sec": round(B * T / (step * 256 / 1000.0), 3),
           "launches_per_forward": len(plan.meta), "plan_GB": round(plan.arena_floats * 4 / 1e9, 1),
           "ms_by_family": {k: [round(v[0], 2), v[2], round(v[1] / v[0] / 1e9, 1) if v[0] and v[1] else None] for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:24]}}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
