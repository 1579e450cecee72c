"""Generate the golden vectors by running the REAL reference in the build container.

    cd /root/repo && PYTHONDONTWRITEBYTECODE=1 \
        PYTHONPATH=tools/ref_shims:/root/reference:tests python tests/golden/make_golden.py

Needs /root/reference (absent on the GPU box) -> only the outputs are committed.
Weights are synthetic and regenerated from (name, shape, seed) by tests/helpers.py.
"""
import json
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import helpers  # noqa: E402

from denoising_diffusion_pytorch import GaussianDiffusion, Unet3D  # noqa: E402  (the reference)
from denoising_diffusion_pytorch.video_denoising_diffusion_pytorch import (  # noqa: E402
    RelativePositionBias,
    Trainer,
    num_to_groups,
)

torch.set_num_threads(8)


def build(cfg_name):
    kw, _, _ = helpers.CONFIGS[cfg_name]
    model = Unet3D(**kw)
    shapes = {k: tuple(v.shape) for k, v in model.state_dict().items()}
    with open(os.path.join(HERE, f"shapes_{cfg_name}.json"), "w") as f:
        json.dump({k: list(v) for k, v in shapes.items()}, f, indent=0)
    sd = helpers.synth_state_dict(shapes, seed=0)
    model.load_state_dict(sd, strict=True)
    return model.eval()


def unet_goldens(only=None):
    for cfg_name in helpers.CONFIGS:
        if only and cfg_name not in only:
            continue
        model = build(cfg_name)
        x, t, cond = helpers.synth_inputs(cfg_name)
        out = {}
        with torch.no_grad():
            out["eps_cond"] = model(x, t, cond=cond, null_cond_prob=0.0).numpy()
            out["eps_null"] = model(x, t, cond=cond, null_cond_prob=1.0).numpy()
            out["eps_w5"] = model.forward_with_guidance_scale(x, t, cond=cond).numpy()
            if cfg_name == "lagr16":
                out["eps_w3"] = model.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=3.0).numpy()
                out["eps_w0"] = model.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=0.0).numpy()
                out["eps_w1"] = model.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=1.0).numpy()
        np.savez(os.path.join(HERE, f"unet_{cfg_name}.npz"), **out)
        print(cfg_name, {k: (v.shape, float(np.abs(v).mean())) for k, v in out.items()})


def gradient_goldens(only=None):
    """Parameter gradients of the l1 training loss (vddp.py:1044-1060, 1622-1629) as the REAL reference's autograd computes them, for the
    configurations of helpers.GRADIENT_CONFIGS (constructor keywords off their defaults): every parameter whose gradient has fewer than 6000
    elements plus a fixed list of large ones -> tests/golden/grads_<config>.npz."""
    for cfg_name in helpers.GRADIENT_CONFIGS:
        if only and cfg_name not in only:
            continue
        model = build(cfg_name)
        kw, (B, T, H, W), _ = helpers.CONFIGS[cfg_name]
        x, t, cond = helpers.synth_inputs(cfg_name)
        diff = GaussianDiffusion(model, image_size=H, num_frames=T, channels=kw["channels"], timesteps=256, loss_type="l1", use_dynamic_thres=True,
                                 sampling_timesteps=256)
        g = torch.Generator().manual_seed(7)
        x0 = torch.rand((B, kw["channels"], T, H, W), generator=g) * 2 - 1
        noise = torch.randn((B, kw["channels"], T, H, W), generator=g)
        out = {"x0": x0.numpy(), "noise": noise.numpy(), "t": t.numpy()}
        model.train()
        model.zero_grad()
        loss = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=0.0)
        loss.backward()
        out["loss_train"] = loss.detach().numpy()
        big = {"downs.0.0.block1.proj.weight", "downs.1.2.fn.fn.to_qkv.weight", "downs.0.3.fn.fn.fn.to_qkv.weight", "downs.0.3.fn.fn.fn.to_out.weight",
               "downs.0.3.fn.fn.fn.to_k.weight", "mid_temporal_attn.fn.fn.fn.to_qkv.weight", "mid_spatial_attn.fn.fn.fn.to_qkv.weight", "ups.3.0.block1.proj.weight",
               "init_temporal_attn.fn.fn.fn.to_qkv.weight", "init_temporal_attn.fn.fn.fn.to_out.weight", "ups.3.3.fn.fn.fn.to_v.weight", "downs.0.4.weight"}
        nograd = []
        for k, p in model.named_parameters():
            if p.grad is None:
                nograd.append(k)
            elif p.numel() < 6000 or k in big:
                out["grad/" + k] = p.grad.numpy()
        out["nograd"] = np.array(sorted(nograd))
        np.savez_compressed(os.path.join(HERE, f"grads_{cfg_name}.npz"), **out)
        print("gradients", cfg_name, float(loss), len([k for k in out if k.startswith("grad/")]), "tensors;", len(nograd), "without gradient")
        model.eval()


FOCUS_CASES = {"plumb16": ([1, 0], [1, 1]), "focus16s": ([1, 0, 1], [1, 1, 1], [0, 1, 0])}


def focus_goldens():
    """Non-trivial focus_present_mask (vddp.py:431, 438-443, 514-524) where the reference accepts one: temporal attentions without tokens."""
    out = {}
    for cfg_name, masks in FOCUS_CASES.items():
        model = build(cfg_name)
        x, t, cond = helpers.synth_inputs(cfg_name)
        with torch.no_grad():
            for m in masks:
                fm = torch.tensor(m, dtype=torch.bool)
                tag = "".join(str(v) for v in m)
                out[f"{cfg_name}/{tag}"] = model(x, t, cond=cond, null_cond_prob=0.0, focus_present_mask=fm).numpy()
                out[f"{cfg_name}/w5_{tag}"] = model.forward_with_guidance_scale(x, t, cond=cond, focus_present_mask=fm).numpy()
            out[f"{cfg_name}/prob1"] = model(x, t, cond=cond, null_cond_prob=0.0, prob_focus_present=1.0).numpy()
    np.savez(os.path.join(HERE, "unet_focus.npz"), **out)
    print({k: float(np.abs(v).mean()) for k, v in out.items()})


def diffusion_goldens():
    cfg_name = "lagr16"
    model = build(cfg_name)
    _, (B, T, H, W), _ = helpers.CONFIGS[cfg_name]
    x, t, cond = helpers.synth_inputs(cfg_name)
    out = {}
    diff = GaussianDiffusion(model, image_size=H, num_frames=T, channels=3, timesteps=256, loss_type="l1",
                             use_dynamic_thres=True, sampling_timesteps=256)
    for k, v in diff.state_dict().items():
        if not k.startswith("denoise_fn."):
            out["sched256_" + k] = v.numpy()
    g = torch.Generator().manual_seed(7)
    x0 = torch.rand((B, 3, T, H, W), generator=g) * 2 - 1
    noise = torch.randn((B, 3, T, H, W), generator=g)
    out["x0"], out["noise"], out["t"] = x0.numpy(), noise.numpy(), t.numpy()
    with torch.no_grad():
        out["q_sample"] = diff.q_sample(x0, t, noise).numpy()
        out["loss_l1_cond"] = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=0.0).numpy()
        out["loss_l1_null"] = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=1.0).numpy()
        diff.loss_type = "l2"
        out["loss_l2_cond"] = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=0.0).numpy()
        diff.loss_type = "l1"
        # one ancestral step with dynamic thresholding; Gaussian = first draw after manual_seed(11)
        tt = torch.tensor([200, 0])
        torch.manual_seed(11)
        out["p_sample_t"] = tt.numpy()
        out["p_sample_w5"] = diff.p_sample(x, tt, cond=cond, guidance_scale=5.0).numpy()
        mean, _, logvar = diff.p_mean_variance(x=x, t=tt, clip_denoised=True, cond=cond, guidance_scale=5.0)
        out["p_mean_w5"], out["p_logvar"] = mean.numpy(), logvar.numpy()
        diff.use_dynamic_thres = False
        mean, _, _ = diff.p_mean_variance(x=x, t=tt, clip_denoised=True, cond=cond, guidance_scale=1.0)
        out["p_mean_static_w1"] = mean.numpy()
        diff.use_dynamic_thres = True

    # gradients of the training loss wrt representative parameters (vddp.py:1622-1629)
    model.train()
    model.zero_grad()
    loss = diff.p_losses(x0, t, cond=cond, noise=noise, null_cond_prob=0.0)
    loss.backward()
    keep = ["init_conv.weight", "init_conv.bias", "downs.0.0.block1.proj.weight", "downs.0.0.block1.norm.weight",
            "downs.0.0.block1.norm.bias", "downs.0.0.mlp.1.weight", "downs.1.0.res_conv.weight",
            "downs.1.2.fn.fn.to_qkv.weight", "downs.1.2.fn.fn.to_k.weight", "downs.1.2.fn.fn.to_out.bias",
            "downs.0.2.fn.norm.gamma", "downs.0.3.fn.fn.fn.to_qkv.weight", "downs.0.3.fn.fn.fn.to_k.weight",
            "downs.0.3.fn.fn.fn.to_out.weight", "downs.0.4.weight", "mid_spatial_attn.fn.fn.fn.to_qkv.weight",
            "mid_spatial_attn.fn.fn.fn.to_v.weight", "mid_temporal_attn.fn.norm.gamma", "ups.0.4.weight", "ups.0.4.bias",
            "ups.3.0.block1.proj.weight", "final_conv.1.weight", "final_conv.1.bias", "null_text_token",
            "null_text_hidden", "time_mlp.1.weight", "time_mlp.3.bias", "sign_emb.weight", "cond_token_to_hidden.1.weight",
            "cond_token_to_hidden.0.weight", "time_rel_pos_bias.relative_attention_bias.weight",
            "init_temporal_attn.fn.fn.fn.to_qkv.weight"]
    named = dict(model.named_parameters())
    for k in keep:
        gk = named[k].grad
        out["grad/" + k] = (gk if gk is not None else torch.zeros_like(named[k])).numpy()
    nograd = sorted(k for k, p in named.items() if p.grad is None)
    out["loss_train"] = loss.detach().numpy()
    model.eval()

    # short ancestral and DDIM loops (timesteps=8); RNG sequence = randn(shape) then one randn_like per step
    diff8 = GaussianDiffusion(model, image_size=H, num_frames=T, channels=3, timesteps=8, use_dynamic_thres=True, sampling_timesteps=8)
    torch.manual_seed(21)
    out["loop8_w5"] = diff8.sample(cond=cond, guidance_scale=5.0).numpy()
    for k, v in diff8.state_dict().items():
        if not k.startswith("denoise_fn."):
            out["sched8_" + k] = v.numpy()
    diffd = GaussianDiffusion(model, image_size=H, num_frames=T, channels=3, timesteps=8, use_dynamic_thres=True, sampling_timesteps=4,
                              ddim_sampling_eta=0.5)
    torch.manual_seed(22)
    out["ddim4_w3"] = diffd.sample(cond=cond, guidance_scale=3.0).numpy()
    np.savez(os.path.join(HERE, "diffusion_lagr16.npz"), **out)
    print("diffusion", {k: getattr(v, "shape", None) for k, v in out.items()})
    return nograd


def relpos_rows():
    """The bucket of every signed frame distance -40 .. 40 (the table of n frames is bucket[j - i]): pins the whole envelope of the fused kernels
    (T <= 32) and beyond, including the logarithmic branch whose fp32 log decides the bucket boundaries."""
    rel = torch.arange(-40, 41)[None, :]
    return RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32)[0].tolist()


def table_goldens(nograd):
    tabs = {"nograd_params_lagr16": nograd}
    for n in (4, 11, 22):
        q = torch.arange(n)
        rel = q[None, :] - q[:, None]
        tabs[f"bucket_{n}"] = RelativePositionBias._relative_position_bucket(rel, num_buckets=32, max_distance=32).tolist()
    tabs["bucket_by_distance_m40_40"] = relpos_rows()
    tabs["ddim_times_256_10"] = list(reversed(torch.linspace(-1, 255, steps=11).int().tolist()))
    tabs["ddim_times_256_256_head"] = list(reversed(torch.linspace(-1, 255, steps=257).int().tolist()))[:5]
    tabs["ddim_times_8_4"] = list(reversed(torch.linspace(-1, 7, steps=5).int().tolist()))
    tabs["num_to_groups"] = {f"{a},{b}": num_to_groups(a, b) for a, b in [(50, 2), (7, 2), (1, 4), (0, 4), (400, 16)]}
    splits = {}
    for N in (4, 7, 400):
        for P in (1, 2, 8):
            cond = torch.arange(N * 11, dtype=torch.float32).reshape(N, 11)
            per_rank = []
            for r in range(P):
                fake = types.SimpleNamespace(accelerator=types.SimpleNamespace(process_index=r, num_processes=P), test_batch_size=2)
                chunks = Trainer.cond_to_gpu(fake, cond)
                per_rank.append([[int(c[0, 0].item()) // 11, int(c[-1, 0].item()) // 11 + 1] if c.shape[0] else [] for c in chunks])
            splits[f"{N},{P}"] = per_rank
    tabs["cond_to_gpu_batch2"] = splits
    # remove_padding: 3 ranks with lengths 2,1,3 padded to 3
    gathered = torch.arange(9, dtype=torch.float32)[:, None].repeat(1, 2)
    kept = Trainer.remove_padding(None, gathered, torch.tensor([2, 1, 3]), 3)
    tabs["remove_padding_2_1_3"] = kept[:, 0].int().tolist()
    with open(os.path.join(HERE, "tables.json"), "w") as f:
        json.dump(tabs, f)
    print("tables ok; nograd params:", len(nograd))


if __name__ == "__main__":
    if sys.argv[1:] == ["--relpos"]:  # only (re)write the all-distances bucket row into tables.json
        path = os.path.join(HERE, "tables.json")
        tabs = json.load(open(path))
        tabs["bucket_by_distance_m40_40"] = relpos_rows()
        json.dump(tabs, open(path, "w"))
        print("relpos row written")
    elif sys.argv[1:] == ["--focus"]:
        focus_goldens()
    elif sys.argv[1:2] == ["--grads"]:  # python make_golden.py --grads [<config> ...]
        gradient_goldens(only=sys.argv[2:])
    elif len(sys.argv) > 1:  # python make_golden.py <config> [...]: only the Unet3D goldens of the named configs (adding one leaves the rest untouched)
        unet_goldens(only=sys.argv[1:])
    else:
        unet_goldens()
        ng = diffusion_goldens()
        table_goldens(ng)
        gradient_goldens()
