"""How far the REFERENCE's own training gradients move under the mixed precision it trains with (main.py:34: Accelerator(mixed_precision='fp16')):
per-parameter relative deviation of the l1-loss gradients under torch.autocast(float16) from the fp32 ones, computed by running the real
reference on the CPU in the build container.  The numbers (not the gradients) are committed as tests/golden/autocast_lagr16.json;
tests/test_gpu_train.py::test_split_bf16_l1_gradients_inside_the_reference_autocast_deviation holds the split-bf16 training mode against them.

    PYTHONDONTWRITEBYTECODE=1 PYTHONPATH=tools/ref_shims:/root/reference:tests python tests/golden/make_golden_autocast.py [lagr16 | lagr64]
"""
import contextlib
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

from denoising_diffusion_pytorch import GaussianDiffusion, Unet3D  # noqa: E402  (the reference)

torch.set_num_threads(8)
CFG = sys.argv[1] if len(sys.argv) > 1 else "lagr16"  # (lagr64: the same yardstick at the real widths, for the reduced-precision training leg)


def main():
    kw, (B, T, H, W), _ = helpers.CONFIGS[CFG]
    model = Unet3D(**kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    model.load_state_dict(helpers.synth_state_dict(shapes, seed=0), strict=True)
    diff = GaussianDiffusion(model, image_size=H, num_frames=T, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True, sampling_timesteps=256)
    _, t, cond = helpers.synth_inputs(CFG)
    g = torch.Generator().manual_seed(7)  # the inputs of the gradient goldens in diffusion_lagr16.npz
    x0 = torch.rand((B, 3, T, H, W), generator=g) * 2 - 1
    noise = torch.randn((B, 3, T, H, W), generator=g)
    model.train()

    def grads(ctx, scale=1.0):
        model.zero_grad()
        with ctx:
            loss = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=0.0)
        (loss.float() * scale).backward()  # scale: what accelerator.backward does under mixed_precision='fp16' (GradScaler.scale(loss).backward(), vddp.py:1629)
        return float(loss.detach()), {k: p.grad.clone() / scale for k, p in model.named_parameters() if p.grad is not None}

    l32, g32 = grads(contextlib.nullcontext())
    out = {"config": CFG, "loss_fp32": l32, "note": "rel = ||g_autocast - g_fp32|| / ||g_fp32|| per parameter, l1 loss, reference on CPU"}
    # "fp16_scaled": fp16 autocast WITH GradScaler's initial loss scale 2^16 -- the recipe main.py:34 actually trains with (the l1 gradient 1 / N per output
    # element is a subnormal half without it); "fp16" / "bf16": the bare autocast figures of rounds 3-5
    for name, dt, scale in (("fp16", torch.float16, 1.0), ("bf16", torch.bfloat16, 1.0), ("fp16_scaled", torch.float16, 65536.0)):
        l16, g16 = grads(torch.autocast("cpu", dtype=dt), scale)
        if not all(bool(torch.isfinite(v).all()) for v in g16.values()):
            print(name, "non-finite gradients (GradScaler would skip the step and halve the scale)")
            continue
        rel = {k: float((g16[k].double() - v.double()).norm() / v.double().norm().clamp_min(1e-30)) for k, v in g32.items() if float(v.double().norm()) > 0}
        vals = np.array(list(rel.values()))
        out[name] = {"loss": l16, "median": float(np.median(vals)), "p10": float(np.percentile(vals, 10)), "p90": float(np.percentile(vals, 90)),
                     "max": float(vals.max()), "rel": rel}
        print(name, "loss", l16, "median", out[name]["median"], "p10", out[name]["p10"], "p90", out[name]["p90"], "max", out[name]["max"], "n", len(rel))
    with open(os.path.join(HERE, f"autocast_{CFG}.json"), "w") as f:
        json.dump(out, f)


if __name__ == "__main__":
    main()
