"""Key -> shape table of the REAL reference's `GaussianDiffusion.state_dict()` (what Trainer.save writes under 'model' / 'ema',
vddp.py:1534-1561), for the checkpoint-interop test.

    PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=tools/ref_shims:/root/reference:tests python tests/golden/make_golden_ckpt.py
"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

from denoising_diffusion_pytorch import GaussianDiffusion, Unet3D  # noqa: E402  (the reference)

cfg_name = "lagr16"
kw, (B, T, H, W), _ = helpers.CONFIGS[cfg_name]
diff = GaussianDiffusion(Unet3D(**kw), image_size=H, num_frames=T, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                         sampling_timesteps=256)
table = {k: list(v.shape) for k, v in diff.state_dict().items()}
with open(os.path.join(HERE, f"shapes_diffusion_{cfg_name}.json"), "w") as f:
    json.dump(table, f, indent=0)
print(len(table), "entries;", sum(1 for k in table if not k.startswith("denoise_fn.")), "schedule buffers")
