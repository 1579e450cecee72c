"""Golden vectors for the geometry extraction, produced by the REAL reference in the build container:
Trainer.save_preds (vddp.py:1870-1919) is run on synthetic sampler outputs with its file output captured, so the fixtures pin the
topology rule AND clean_pred (src/utils.py, networkx) exactly as the reference chains them.

    PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=tools/ref_shims:/root/reference:tests python tests/golden/make_golden_geometry.py

Only the expected outputs are committed; the inputs are regenerated from seeds by tests/helpers.py.
"""
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

from denoising_diffusion_pytorch import video_denoising_diffusion_pytorch as vddp  # noqa: E402  (the reference)

captured = {}
vddp.video_tensor_to_gif = lambda *a, **k: None                      # no GIF output
vddp.np.savetxt = lambda path, arr, **k: captured.__setitem__("geom", np.array(arr))  # instead of geometries.csv

for name, (seed, N, T, P, z) in helpers.GEOMETRY_CASES.items():
    videos = helpers.synth_geometry_videos(seed, N, T, P, z)
    out = {}
    for frame in ("lagrangian", "eulerian"):
        fake = types.SimpleNamespace(
            results_folder="unused", step=0, selected_channels=[0, 1, 2], reference_frame=frame, num_frames=T, device="cpu",
            ds=types.SimpleNamespace(zero_u_2=torch.tensor([z], dtype=torch.float32)), accelerator=types.SimpleNamespace(print=lambda *a, **k: None))
        fake.remove_padding = types.MethodType(vddp.Trainer.remove_padding, fake)
        captured.clear()
        try:
            vddp.Trainer.save_preds(fake, videos.clone(), torch.tensor([N]), N, 1, mode="eval")
            out[frame] = captured["geom"].astype(np.int8)
        except IndexError:  # clean_pred fails on an image without any pair of neighbouring pixels (src/utils.py:73)
            out[frame] = np.full((1, 1), -1, dtype=np.int8)
    np.savez_compressed(os.path.join(HERE, f"geometry_{name}.npz"), **out)
    print(name, {k: (v.shape, int(v.sum())) for k, v in out.items()})
