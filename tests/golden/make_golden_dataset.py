"""Golden vectors for the training-sample assembly, produced by the REAL reference in the build container: the
reference's Dataset (vddp.py:1126-1397) is constructed on a scratch folder holding the synthetic CSV files of tests/helpers.py and
one (empty) GIF file per sample and field; only the GIF decoding is replaced -- gif_to_tensor returns ToTensor of the synthetic u8
frames, which is what PIL + torchvision hand over for an image already at image_size -- so the fixtures pin the global ranges,
zero_u_2, the label interpolation / scaling and the per-field arithmetic exactly as the reference chains them.

    PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=tools/ref_shims:/root/reference:tests python tests/golden/make_golden_dataset.py

Only the expected outputs are committed; the inputs are regenerated from seeds by tests/helpers.py.
"""
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

from denoising_diffusion_pytorch import video_denoising_diffusion_pytorch as vddp  # noqa: E402  (the reference)

FOLDERS = {"lagrangian": ("topo", "u_1", "u_2", "s_mises", "s_22", "ener"), "eulerian": ("topo", "s_mises", "s_22", "ener")}
ORDER = {"lagrangian": ("topo", "u_1", "u_2", "s_mises", "s_22"), "eulerian": ("topo", "s_mises", "s_22", "ener")}

for name, (seed, frame, N, f, P, num_frames, sel, per_frame) in helpers.DATASET_CASES.items():
    frames, fr, curves = helpers.synth_dataset(seed, frame, N, f, P)
    with tempfile.TemporaryDirectory() as tmp:
        folder = tmp + "/"
        for sub in FOLDERS[frame]:
            os.makedirs(folder + "gifs/" + sub)
            for i in range(N):
                open(f"{folder}gifs/{sub}/{i}.gif", "wb").close()
        np.savetxt(folder + "frame_range_data.csv", fr, delimiter=",", fmt="%.17g")
        np.savetxt(folder + "stress_strain_data.csv", curves, delimiter=",", fmt="%.17g")

        def fake_gif_to_tensor(path, channels=1, transform=None):
            field, idx = path.parent.name, int(path.stem)
            u8 = torch.from_numpy(frames[idx, ORDER[frame].index(field)])
            return (u8.to(torch.float32) / 255)[None]  # torchvision's ToTensor on mode 'L' frames, stacked on dim 1

        vddp.gif_to_tensor = fake_gif_to_tensor
        ds = vddp.Dataset(folder, P, selected_channels=list(sel), num_frames=num_frames, per_frame_cond=per_frame, reference_frame=frame)
        out = {"labels": ds.labels.numpy()}
        for i in range(N):
            t, lab = ds[i]
            assert torch.equal(lab, ds.labels[i])
            out[f"sample{i}"] = t.numpy()
        if frame == "lagrangian":
            out["zero_u_2"] = ds.zero_u_2.numpy()
        for k in ("min_u_1", "max_u_1", "min_u_2", "max_u_2", "max_s_mises", "min_s_22", "max_s_22", "max_strain_energy"):
            if hasattr(ds, k):
                out["g_" + k] = np.float64(getattr(ds, k).item())
        np.savez_compressed(os.path.join(HERE, f"dataset_{name}.npz"), **out)
        print(name, {k: getattr(v, "shape", v) for k, v in out.items()})
