"""CPU-only checks of the host side: C-ABI surface, integer/float64 host mathematics, state_dict drop-in, plan construction."""
import copy
import json
import os
import pickle
import re

import numpy as np
import pytest
import torch

import helpers

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _gold_tables():
    with open(os.path.join(helpers.GOLDEN_DIR, "tables.json")) as f:
        return json.load(f)


def test_c_abi_exports_every_declared_symbol():
    from videometamaterials_amd import _native as N
    hdr = open(os.path.join(ROOT, "include", "vmm_kernels.h")).read()
    declared = set(re.findall(r"^(?:int|int64_t) (vmm_\w+)\(", hdr, flags=re.M))
    assert declared == set(N.SIGNATURES), declared ^ set(N.SIGNATURES)
    lib = N.lib()  # raises if the library is missing or a symbol is not exported (no compute calls here)
    for name in declared:
        assert getattr(lib, name) is not None


def test_missing_library_fails_loudly(monkeypatch):
    from videometamaterials_amd import _native as N
    monkeypatch.setattr(N, "_lib", None)
    monkeypatch.setattr(N, "LIB_PATH", "/nonexistent/libvmm_hip.so")
    with pytest.raises(N.NativeError):
        N.lib()


def test_hostmath_integer_tables_bit_exact():
    from videometamaterials_amd import hostmath as hm
    tabs = _gold_tables()
    for n in (4, 11, 22):
        assert hm.relpos_buckets(n).tolist() == tabs[f"bucket_{n}"]
    assert [p[0] for p in hm.ddim_time_pairs(256, 10)] + [-1] == tabs["ddim_times_256_10"]
    assert [p[0] for p in hm.ddim_time_pairs(8, 4)] + [-1] == tabs["ddim_times_8_4"]
    for key, want in tabs["num_to_groups"].items():
        a, b = map(int, key.split(","))
        assert hm.num_to_groups(a, b) == want
    for key, per_rank in tabs["cond_to_gpu_batch2"].items():
        n, p = map(int, key.split(","))
        for r in range(p):
            assert [list(x) for x in hm.shard_rows(n, r, p, 2)] == [x for x in per_rank[r] if x]
    gathered = torch.arange(9, dtype=torch.float32)[:, None].repeat(1, 2)
    assert hm.strip_padding(gathered, [2, 1, 3], 3)[:, 0].int().tolist() == tabs["remove_padding_2_1_3"]


def test_hostmath_schedule_and_quantile_rank():
    from videometamaterials_amd import hostmath as hm
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, "diffusion_lagr16.npz"))
    for T in (256, 8):
        sch = hm.schedule_buffers(T)
        for name in hm.SCHEDULE_NAMES:
            assert np.array_equal(sch[name].numpy(), gold[f"sched{T}_{name}"]), name
    # torch.quantile's float32 rank arithmetic
    for n in (304128, 1000, 33792, 7):
        x = torch.rand(n)
        k, frac = hm.quantile_rank(n, 0.9)
        srt = x.sort().values
        lo, hi = srt[k], srt[min(k + 1, n - 1)]
        assert torch.quantile(x, 0.9) == torch.lerp(lo, hi, torch.tensor(frac))


@pytest.mark.parametrize("cfg_name", list(helpers.CONFIGS))
def test_state_dict_is_a_drop_in(cfg_name):
    import videometamaterials_amd as vm
    kw, _, _ = helpers.CONFIGS[cfg_name]
    m = vm.Unet3D(**kw)
    shapes = helpers.load_shapes(cfg_name)
    sd = m.state_dict()
    assert set(sd) == set(shapes)
    assert all(tuple(sd[k].shape) == shapes[k] for k in shapes)
    ref_sd = helpers.synth_state_dict(shapes)
    m.load_state_dict(ref_sd, strict=True)
    no_freqs = {("module." + k): v for k, v in ref_sd.items() if not k.endswith("freqs")}  # DDP prefix, older rotary package
    m.load_state_dict(no_freqs, strict=True)
    m2 = copy.deepcopy(m)
    m3 = pickle.loads(pickle.dumps(m))
    assert torch.equal(m2.state_dict()["init_conv.weight"], m3.state_dict()["init_conv.weight"])


def test_constructor_rejections():
    import videometamaterials_amd as vm
    with pytest.raises(ValueError):
        vm.Unet3D(dim=16, cond_attention="bogus")
    with pytest.raises(AssertionError):
        vm.Unet3D(dim=16, init_kernel_size=6)
    with pytest.raises(AssertionError):
        vm.GaussianDiffusion(vm.Unet3D(dim=16), image_size=8, num_frames=2, timesteps=10, sampling_timesteps=20)
    d = vm.GaussianDiffusion(vm.Unet3D(dim=16), image_size=8, num_frames=2, channels=3, timesteps=10, sampling_timesteps=5)
    assert d.is_ddim_sampling and d.num_timesteps == 10
    with pytest.raises(ValueError):
        d.forward(torch.zeros(1, 3, 2, 9, 8))  # check_shape (vddp.py:1064)


def test_plan_construction_without_gpu():
    """Plans are static launch lists over a static arena: build them on CPU buffers and check the bookkeeping."""
    import videometamaterials_amd as vm
    from videometamaterials_amd import plan
    kw, (B, T, H, W), cl = helpers.CONFIGS["lagr16"]
    m = vm.Unet3D(**kw)
    pl = plan.build_plan(m, B, T, H, W, cl, "cpu")
    assert len(pl.steps) == len(pl.meta) > 150  # (fusions keep shrinking the launch list)
    conv_flops = sum(f for k, f, _ in pl.meta if k.startswith("vmm_conv_igemm"))
    assert conv_flops > 1e9
    assert not pl.bwd_steps
    keep = plan.build_plan(m, B, T, H, W, cl, "cpu", training=True)
    assert keep.arena_floats > pl.arena_floats  # inference plans reuse dead buffers
    assert pl.out.shape == (B, 3, T, H, W) and pl.x_in.shape == (B, 3, T, H, W)
    # training plan: a backward launch list and one flat gradient buffer covering exactly the parameters that receive gradients
    with open(os.path.join(helpers.GOLDEN_DIR, "tables.json")) as f:
        nograd = {k for k in json.load(f)["nograd_params_lagr16"] if not k.endswith("freqs")}
    trainable = {k for k, _ in m.named_parameters()}
    assert set(keep.param_slices) == trainable - nograd
    assert len(keep.bwd_steps) > len(keep.steps)
    # marks are monotone: later in the backward, a longer tail of the flat buffer is final
    offs = [x for _, x in keep.bwd_marks]
    assert offs == sorted(offs, reverse=True) and offs[-1] == 0
    spans = sorted(keep.param_slices.values())
    assert all(a[0] + a[1] <= b[0] for a, b in zip(spans, spans[1:]))
