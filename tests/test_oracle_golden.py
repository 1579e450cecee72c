"""The oracle (oracle/*.py) against outputs of the real reference (tests/golden/*.npz)."""
import json
import os

import numpy as np
import pytest
import torch

import helpers
from oracle import diffusion_oracle as do
from oracle import unet3d_oracle as uo

RTOL = 2e-5  # oracle and reference are both CPU fp32; differences are summation-order only


def _load(cfg_name):
    kw, _, _ = helpers.CONFIGS[cfg_name]
    cfg = uo.UnetCfg(**kw)
    sd = helpers.synth_state_dict(helpers.load_shapes(cfg_name))
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, f"unet_{cfg_name}.npz"))
    return cfg, sd, gold


@pytest.mark.parametrize("cfg_name", list(helpers.CONFIGS))
def test_unet_forward_matches_reference(cfg_name):
    cfg, sd, gold = _load(cfg_name)
    x, t, cond = helpers.synth_inputs(cfg_name)
    B = x.shape[0]
    with torch.no_grad():
        e_c = uo.unet3d_forward(sd, cfg, x, t, cond, torch.zeros(B, dtype=torch.bool))
        e_n = uo.unet3d_forward(sd, cfg, x, t, cond, torch.ones(B, dtype=torch.bool))
        e_5 = uo.unet3d_guided(sd, cfg, x, t, cond, 5.0)
    assert helpers.rel_err(e_c, torch.from_numpy(gold["eps_cond"])) < RTOL
    assert helpers.rel_err(e_n, torch.from_numpy(gold["eps_null"])) < RTOL
    assert helpers.rel_err(e_5, torch.from_numpy(gold["eps_w5"])) < 5 * RTOL


def test_guidance_scales():
    cfg, sd, gold = _load("lagr16")
    x, t, cond = helpers.synth_inputs("lagr16")
    with torch.no_grad():
        for w, key in ((3.0, "eps_w3"), (0.0, "eps_w0"), (1.0, "eps_w1")):
            got = uo.unet3d_guided(sd, cfg, x, t, cond, w)
            assert helpers.rel_err(got, torch.from_numpy(gold[key])) < 5 * RTOL


@pytest.mark.parametrize("cfg_name", ["plumb16", "focus16s"])
def test_focus_present_mask_matches_reference(cfg_name):
    """A non-trivial focus_present_mask (vddp.py:431, 438-443, 514-524) on the configs where the reference accepts one (no tokens at the temporal
    sites): masked samples attend to their own frame only; all-masked takes the values-only shortcut."""
    cfg, sd, _ = _load(cfg_name)
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, "unet_focus.npz"))
    x, t, cond = helpers.synth_inputs(cfg_name)
    B = x.shape[0]
    tags = sorted(k.split("/")[1] for k in gold.files if k.startswith(cfg_name + "/") and k.split("/")[1][0] in "01")
    assert len(tags) >= 2
    with torch.no_grad():
        for tag in tags:
            fm = torch.tensor([c == "1" for c in tag])
            e_c = uo.unet3d_forward(sd, cfg, x, t, cond, torch.zeros(B, dtype=torch.bool), focus=fm)
            assert helpers.rel_err(e_c, torch.from_numpy(gold[f"{cfg_name}/{tag}"])) < RTOL, tag
            e_n = uo.unet3d_forward(sd, cfg, x, t, cond, torch.ones(B, dtype=torch.bool), focus=fm)
            assert helpers.rel_err(e_n + (e_c - e_n) * 5.0, torch.from_numpy(gold[f"{cfg_name}/w5_{tag}"])) < 5 * RTOL, tag
        ones = uo.unet3d_forward(sd, cfg, x, t, cond, torch.zeros(B, dtype=torch.bool), focus=torch.ones(B, dtype=torch.bool))
        assert helpers.rel_err(ones, torch.from_numpy(gold[f"{cfg_name}/prob1"])) < RTOL
        # ... and it differs from the unmasked output (the mask is not inert here)
        plain = uo.unet3d_forward(sd, cfg, x, t, cond, torch.zeros(B, dtype=torch.bool))
        assert helpers.rel_err(ones, plain) > 1e-2


def test_integer_tables_bit_exact():
    with open(os.path.join(helpers.GOLDEN_DIR, "tables.json")) as f:
        tabs = json.load(f)
    for n in (4, 11, 22):
        assert uo.rel_pos_bucket_table(n).tolist() == tabs[f"bucket_{n}"]
    by_dist = tabs["bucket_by_distance_m40_40"]
    for n in range(1, 41):
        assert uo.rel_pos_bucket_table(n).tolist() == [[by_dist[40 + j - i] for j in range(n)] for i in range(n)], n
    # literal rows quoted in SURVEY 8a row a4
    assert uo.rel_pos_bucket_table(11)[0].tolist() == [0, 17, 18, 19, 20, 21, 22, 23, 24, 24, 25]
    assert uo.rel_pos_bucket_table(11)[:, 0].tolist() == [0, 1, 2, 3, 4, 5, 6, 7, 8, 8, 9]
    assert do.ddim_times(256, 10) == tabs["ddim_times_256_10"] == [255, 229, 203, 178, 152, 127, 101, 75, 50, 24, -1]
    assert do.ddim_times(256, 256)[:5] == tabs["ddim_times_256_256_head"]
    assert do.ddim_times(8, 4) == tabs["ddim_times_8_4"]
    for key, want in tabs["num_to_groups"].items():
        a, b = map(int, key.split(","))
        assert do.num_to_groups(a, b) == want
    for key, per_rank in tabs["cond_to_gpu_batch2"].items():
        N, P = map(int, key.split(","))
        for r in range(P):
            got = [list(p) for p in do.shard_rows(N, r, P, 2)]
            assert got == [p for p in per_rank[r] if p], (key, r)
    gathered = torch.arange(9, dtype=torch.float32)[:, None].repeat(1, 2)
    assert do.strip_padding(gathered, [2, 1, 3], 3)[:, 0].int().tolist() == tabs["remove_padding_2_1_3"]


def test_schedule_and_elementwise():
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, "diffusion_lagr16.npz"))
    for T in (256, 8):
        sch = do.schedule_buffers(T)
        for name in do.SCHEDULE_NAMES:
            assert np.array_equal(sch[name].numpy(), gold[f"sched{T}_{name}"]), name  # bit-exact (same fp64 formulas)
    sch = do.schedule_buffers(256)
    # literal values quoted in SURVEY 8a row a15
    assert abs(float(do.cosine_betas(256)[0]) - 1.888266729946908e-4) < 1e-15
    assert float(do.cosine_betas(256)[-1]) == 0.9999
    x0, noise, t = (torch.from_numpy(gold[k]) for k in ("x0", "noise", "t"))
    assert torch.equal(do.q_sample(sch, x0, t, noise), torch.from_numpy(gold["q_sample"]))


def test_losses_and_sampling_steps():
    cfg, sd, _ = _load("lagr16")
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, "diffusion_lagr16.npz"))
    x, t, cond = helpers.synth_inputs("lagr16")
    B = x.shape[0]
    sch = do.schedule_buffers(256)
    x0, noise = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["noise"])
    F_, T_ = torch.zeros(B, dtype=torch.bool), torch.ones(B, dtype=torch.bool)
    with torch.no_grad():
        l1c = do.p_losses(sch, lambda a, b: uo.unet3d_forward(sd, cfg, a, b, cond, F_), x0, t, noise, "l1")
        l1n = do.p_losses(sch, lambda a, b: uo.unet3d_forward(sd, cfg, a, b, cond, T_), x0, t, noise, "l1")
        l2c = do.p_losses(sch, lambda a, b: uo.unet3d_forward(sd, cfg, a, b, cond, F_), x0, t, noise, "l2")
        assert abs(float(l1c) - float(gold["loss_l1_cond"])) < 1e-5 * float(gold["loss_l1_cond"])
        assert abs(float(l1n) - float(gold["loss_l1_null"])) < 1e-5 * float(gold["loss_l1_null"])
        assert abs(float(l2c) - float(gold["loss_l2_cond"])) < 1e-5 * float(gold["loss_l2_cond"])

        tt = torch.from_numpy(gold["p_sample_t"])
        torch.manual_seed(11)
        z = torch.randn_like(x)
        got = do.p_sample_step(sch, lambda a, b: uo.unet3d_guided(sd, cfg, a, b, cond, 5.0), x, tt, z)
        assert helpers.rel_err(got, torch.from_numpy(gold["p_sample_w5"])) < 1e-4

        # 8-step ancestral loop and 4-step DDIM, RNG order of vddp.py:970,960 / 994,1012
        sch8 = do.schedule_buffers(8)
        torch.manual_seed(21)
        xT = torch.randn(x.shape)
        zs = [torch.randn_like(x) for _ in range(8)]
        got = do.p_sample_loop(sch8, lambda a, b: uo.unet3d_guided(sd, cfg, a, b, cond, 5.0), xT, zs, timesteps=8)
        assert helpers.rel_err(got, torch.from_numpy(gold["loop8_w5"])) < 1e-3
        torch.manual_seed(22)
        xT = torch.randn(x.shape)
        zs = [torch.randn_like(x) for _ in range(4)]
        got = do.ddim_sample(sch8, lambda a, b: uo.unet3d_guided(sd, cfg, a, b, cond, 3.0), xT, zs, timesteps=8, sampling_timesteps=4, eta=0.5)
        assert helpers.rel_err(got, torch.from_numpy(gold["ddim4_w3"])) < 1e-3


def test_training_gradients():
    """Autograd through the oracle reproduces the reference's parameter gradients (SURVEY 8c iii)."""
    cfg, sd, _ = _load("lagr16")
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, "diffusion_lagr16.npz"))
    with open(os.path.join(helpers.GOLDEN_DIR, "tables.json")) as f:
        nograd = {k for k in json.load(f)["nograd_params_lagr16"] if not k.endswith("freqs")}
    x, t, cond = helpers.synth_inputs("lagr16")
    B = x.shape[0]
    sd = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sd.items()}
    sch = do.schedule_buffers(256)
    x0, noise = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["noise"])
    loss = do.p_losses(sch, lambda a, b: uo.unet3d_forward(sd, cfg, a, b, cond, torch.zeros(B, dtype=torch.bool)), x0, t, noise)
    loss.backward()
    assert abs(float(loss.detach()) - float(gold["loss_train"])) < 1e-5
    for key in gold.files:
        if key.startswith("grad/"):
            name = key[5:]
            want = torch.from_numpy(gold[key])
            got = sd[name].grad if sd[name].grad is not None else torch.zeros_like(want)
            if float(want.abs().max()) == 0:
                assert float(got.abs().max()) == 0, name
            else:
                assert helpers.rel_err(got, want) < 2e-4, name
    got_nograd = {k for k, v in sd.items() if v.requires_grad and (v.grad is None)}
    assert got_nograd == nograd


@pytest.mark.parametrize("cfg_name", helpers.GRADIENT_CONFIGS)
def test_training_gradients_off_default_constructor_keywords(cfg_name):
    """Autograd through the oracle against the reference's own gradients with the constructor keywords off their defaults (attn_heads, attn_dim_head
    incl. the partial rotary of widths above 32, resnet_groups, init_kernel_size; vddp.py:575-626): tests/golden/grads_<config>.npz."""
    cfg, sd, _ = _load(cfg_name)
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, f"grads_{cfg_name}.npz"))
    x, t, cond = helpers.synth_inputs(cfg_name)
    B = x.shape[0]
    sd = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sd.items()}
    sch = do.schedule_buffers(256)
    x0, noise = torch.from_numpy(gold["x0"]), torch.from_numpy(gold["noise"])
    loss = do.p_losses(sch, lambda a, b: uo.unet3d_forward(sd, cfg, a, b, cond, torch.zeros(B, dtype=torch.bool)), x0, t, noise)
    loss.backward()
    assert abs(float(loss.detach()) - float(gold["loss_train"])) < 1e-5
    n = 0
    for key in gold.files:
        if key.startswith("grad/"):
            name, want = key[5:], torch.from_numpy(gold[key])
            got = sd[name].grad if sd[name].grad is not None else torch.zeros_like(want)
            n += 1
            if float(want.abs().max()) == 0:
                assert float(got.abs().max()) == 0, name
            else:
                assert helpers.rel_err(got, want) < 2e-4, name
    assert n > 200
    nograd = {str(k) for k in gold["nograd"] if not str(k).endswith("freqs")}
    assert {k for k, v in sd.items() if v.requires_grad and (v.grad is None)} == nograd
