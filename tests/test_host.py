"""CPU-only checks of the host side: C-ABI surface, integer/float64 host mathematics, state_dict drop-in, plan construction."""
import copy
import json
import os
import pickle
import re
import sys

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
    # the data-parallel engine's surface (include/vmm_dp.h) lives in the same library
    dp_hdr = open(os.path.join(ROOT, "include", "vmm_dp.h")).read()
    dp_declared = set(re.findall(r"^(?:int|const char\*) (vmm_dp_\w+)\(", dp_hdr, flags=re.M))
    assert dp_declared == set(N.DP_SIGNATURES), dp_declared ^ set(N.DP_SIGNATURES)
    for name in dp_declared:
        assert getattr(lib, name) is not None
    # experiments (include/vmm_experiments.h) are declared apart and are NOT in the product library
    ex_hdr = open(os.path.join(ROOT, "include", "vmm_experiments.h")).read()
    ex_declared = set(re.findall(r"^(?:int|int64_t) (vmm_\w+)\(", ex_hdr, flags=re.M))
    assert ex_declared == set(N.EXPERIMENT_SIGNATURES) and not (ex_declared & declared)
    if "VMM_LIB_PATH" not in os.environ:
        import subprocess
        syms = subprocess.run(["nm", "-D", "--defined-only", N.LIB_PATH], capture_output=True, text=True, check=True).stdout
        exported = set(re.findall(r" T (vmm_\w+)", syms))
        assert exported == declared | dp_declared, exported ^ (declared | dp_declared)
        assert not [s_ for s_ in exported if "wino" in s_ or "persistent" in s_ or "_pw" in s_]


def test_dp_engine_rejects_bad_arguments_without_a_gpu():
    """Argument checks of the engine that need neither a GPU nor RCCL (no collective runs here)."""
    import ctypes as C
    from videometamaterials_amd import _native as N
    lib = N.lib()
    assert lib.vmm_dp_get_unique_id(None, None) == -1
    handle = C.c_void_p()
    ident = (C.c_char * 128)()
    assert lib.vmm_dp_init(C.byref(handle), None, 3, 2, ident, 0) == -1  # rank outside the world
    assert lib.vmm_dp_init(C.byref(handle), b"/nonexistent/librccl.so", 0, 1, ident, 0) == -2  # RCCL cannot be loaded: loud, no fallback
    assert not handle.value
    assert lib.vmm_dp_wait_all(None, None) == -1 and lib.vmm_dp_finalize(None) == -1
    assert lib.vmm_dp_world(None) == -1 and lib.vmm_dp_last_error(None) == b"null engine"


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
    # every frame count up to 40 (the fused kernels' envelope is T <= 32): the reference's bucket of every signed distance -40 .. 40, the
    # logarithmic branch included (hostmath evaluates it with numpy's fp32 log, the reference with torch's)
    by_dist = tabs["bucket_by_distance_m40_40"]
    for n in range(1, 41):
        assert hm.relpos_buckets(n).tolist() == [[by_dist[40 + j - i] for j in range(n)] for i in range(n)], n
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


def test_hostmath_rotary_table_spans_min_32_dim_head():
    """RotaryEmbedding(min(32, attn_dim_head)) (vddp.py:612): the table's leading min(32, dh) / 2 pairs are the oracle's angles (pinned by the
    reference goldens dh16 / dh64 / dh24), the pairs beyond them the exact identity."""
    from oracle import unet3d_oracle as uo
    from videometamaterials_amd import hostmath
    for dh in (4, 8, 16, 24, 32, 48, 64, 128):
        tab = hostmath.rotary_table(7, dh)
        assert tab.shape == (7, dh // 2, 2)
        span = min(32, dh) // 2
        assert torch.equal(tab[:, span:, 0], torch.ones(7, dh // 2 - span)) and torch.equal(tab[:, span:, 1], torch.zeros(7, dh // 2 - span))
        x = torch.randn(7, dh)
        e, o = x[:, 0::2], x[:, 1::2]
        rot = torch.stack([e * tab[..., 0] - o * tab[..., 1], o * tab[..., 0] + e * tab[..., 1]], -1).reshape(7, dh)
        assert torch.allclose(rot, uo.rotary_rotate(x), atol=1e-6)
        assert torch.equal(rot[:, 2 * span:], x[:, 2 * span:])
    assert torch.equal(hostmath.rotary_table(11, 32)[..., 0], (torch.arange(11).float()[:, None] * (1.0 / (10000 ** (torch.arange(0, 32, 2).float() / 32)))[None]).cos())
    with pytest.raises(ValueError):
        hostmath.rotary_table(4, 7)


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
    stem = next(k for k in shapes if k.startswith("init_conv.") and k.endswith("weight"))  # ('init_conv.conv.weight' under circular_1d)
    assert torch.equal(m2.state_dict()[stem], m3.state_dict()[stem]) and torch.equal(m2.state_dict()[stem], ref_sd[stem])


def test_constructor_rejections():
    import videometamaterials_amd as vm
    with pytest.raises(ValueError):
        vm.Unet3D(dim=16, cond_attention="bogus")
    with pytest.raises(AssertionError):
        vm.Unet3D(dim=16, init_kernel_size=6)
    for dh in (6, 2, 132, 30):  # (head widths that do not move in 16-byte pieces / beyond the kernels' register budget: refused at construction, loudly)
        with pytest.raises(NotImplementedError, match="attn_dim_head"):
            vm.Unet3D(dim=16, attn_dim_head=dh)
    for dh in (4, 8, 16, 24, 32, 48, 64, 96, 128):
        m = vm.Unet3D(dim=16, attn_dim_head=dh, attn_heads=2)
        assert m.state_dict()["downs.0.3.fn.fn.fn.to_qkv.weight"].shape == (3 * 2 * dh, 16)
        assert m.state_dict()["downs.0.2.fn.fn.to_qkv.weight"].shape == (3 * 2 * 32, 16, 1, 1)       # the linear attention keeps dim_head = 32 (vddp.py:679)
        assert m.state_dict()["mid_spatial_attn.fn.fn.fn.to_qkv.weight"].shape == (3 * 2 * 32, 128)  # ... and so does the mid spatial one (vddp.py:687)
        assert m.state_dict()["downs.0.3.fn.fn.fn.rotary_emb.freqs"].shape == (min(32, dh) // 2,)      # RotaryEmbedding(min(32, attn_dim_head)), vddp.py:612
    # init_dim != dim constructs (state_dict-compatible) and fails in forward like the reference (vddp.py:706, 820)
    m = vm.Unet3D(dim=16, init_dim=24)
    assert m.state_dict()["init_conv.weight"].shape[0] == 24 and m.state_dict()["final_conv.0.block1.proj.weight"].shape[1] == 32
    with pytest.raises(RuntimeError, match="init_dim"):
        m._check_inputs(torch.zeros(1, 3, 2, 8, 8), torch.zeros(1, 51), None, 0.0)
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


def _diffusion_ckpt_variants(ref_sd):
    """The key variants a reference checkpoint can come in: rotary freqs present or not (package version), keys
    bare or behind DDP's `module.` -- at the wrapped GaussianDiffusion or at the denoiser."""
    no_freqs = {k: v for k, v in ref_sd.items() if ".rotary_emb." not in k}
    yield "as saved", dict(ref_sd)
    yield "no rotary freqs", no_freqs
    yield "module. prefix", {"module." + k: v for k, v in ref_sd.items()}
    yield "module. prefix, no rotary freqs", {"module." + k: v for k, v in no_freqs.items()}
    yield "denoise_fn.module.", {(k.replace("denoise_fn.", "denoise_fn.module.", 1)): v for k, v in ref_sd.items()}
    newer = dict(ref_sd)
    newer["denoise_fn.downs.0.3.fn.fn.fn.rotary_emb.cached_freqs"] = torch.zeros(11, 32)  # tables newer rotary packages may persist
    yield "extra rotary tables", newer


def test_reference_checkpoint_loads_at_the_gaussian_diffusion_level():
    """Trainer.load (vddp.py:1563-1592) calls model.load_state_dict(data['model']) and ema_model.load_state_dict(data['ema']) on the
    GaussianDiffusion objects, strict.  Key list = the real reference's GaussianDiffusion.state_dict() (make_golden_ckpt.py)."""
    import videometamaterials_amd as vm
    kw, (B, T, H, W), _ = helpers.CONFIGS["lagr16"]
    with open(os.path.join(helpers.GOLDEN_DIR, "shapes_diffusion_lagr16.json")) as f:
        shapes = {k: tuple(v) for k, v in json.load(f).items()}
    diff = vm.GaussianDiffusion(vm.Unet3D(**kw), image_size=H, num_frames=T, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=256)
    own = diff.state_dict()
    assert set(own) == set(shapes) and all(tuple(own[k].shape) == shapes[k] for k in shapes)
    ref_sd = {k: (helpers.synth_tensor(k[len("denoise_fn."):], s, 3) if k.startswith("denoise_fn.") else own[k].clone()) for k, s in shapes.items()}
    ema = copy.deepcopy(diff)  # vddp.py:1453
    for what, sd in _diffusion_ckpt_variants(ref_sd):
        for target in (diff, ema):
            for p in target.parameters():
                p.data.zero_()
            gen = target.denoise_fn._generation
            res = target.load_state_dict(sd)  # strict
            assert not res.missing_keys and not res.unexpected_keys, what
            assert target.denoise_fn._generation > gen, what  # cached plans will re-pack
            got = target.state_dict()
            for k in ("denoise_fn.init_conv.weight", "denoise_fn.ups.3.3.fn.fn.fn.to_out.weight", "denoise_fn.null_text_token"):
                assert torch.equal(got[k], ref_sd[k]), (what, k)
            assert torch.equal(got["betas"], own["betas"])
    with pytest.raises(RuntimeError):  # strictness is kept for everything that is not one of the tolerated variants
        diff.load_state_dict({k: v for k, v in ref_sd.items() if k != "denoise_fn.init_conv.weight"})
    with pytest.raises(RuntimeError):
        diff.load_state_dict(dict(ref_sd, **{"denoise_fn.bogus.weight": torch.zeros(1)}))
    pickle.loads(pickle.dumps(diff)).load_state_dict(ref_sd)


def test_ctypes_structs_match_the_c_header(tmp_path):
    """sizeof / offsetof of every struct in include/vmm_kernels.h, as gcc lays them out, against the ctypes Structures the host binds
    them with (and against the listing in INTEGRATION.md)."""
    import ctypes
    import shutil
    import subprocess
    from videometamaterials_amd import _native as N
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    pairs = {"vmm_conv_desc": N.ConvDesc, "vmm_dense_job": N.DenseJob, "vmm_pack_job": N.PackJob, "vmm_dense_bwd_job": N.DenseBwdJob,
             "vmm_optim_job": N.OptimJob, "vmm_attn_block_bwd": N.AttnBlockBwd, "vmm_reduce_job": N.ReduceJob}
    hdr = open(os.path.join(ROOT, "include", "vmm_kernels.h")).read()
    assert set(re.findall(r"typedef struct (\w+)", hdr)) == set(pairs)
    lines = ['#include <stdio.h>', '#include <stddef.h>', '#include "vmm_kernels.h"', "int main(void) {"]
    for cname, cls in pairs.items():
        lines.append(f'  printf("{cname} %zu\\n", sizeof({cname}));')
        for fname, _ in cls._fields_:
            lines.append(f'  printf("{cname}.{fname} %zu\\n", offsetof({cname}, {fname}));')
    lines += ["  return 0;", "}"]
    src = tmp_path / "layout.c"
    src.write_text("\n".join(lines))
    exe = tmp_path / "layout"
    subprocess.run(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)], check=True)
    got = dict(line.split() for line in subprocess.run([str(exe)], check=True, capture_output=True, text=True).stdout.splitlines())
    for cname, cls in pairs.items():
        assert int(got[cname]) == ctypes.sizeof(cls), cname
        for fname, _ in cls._fields_:
            assert int(got[f"{cname}.{fname}"]) == getattr(cls, fname).offset, f"{cname}.{fname}"
        # every member of the C struct is bound: the field count of the header's struct body equals the ctypes one
        body = re.search(r"typedef struct %s \{(.*?)\} %s;" % (cname, cname), hdr, flags=re.S).group(1)
        body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
        members = [m for decl in body.split(";") for m in decl.split(",") if m.strip()]
        assert len(members) == len(cls._fields_), (cname, len(members), len(cls._fields_))
    # INTEGRATION.md shows the same vmm_conv_desc binding
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    for fname, _ in N.ConvDesc._fields_:
        assert re.search(r'\("%s"' % fname, doc), f"INTEGRATION.md's vmm_conv_desc listing lacks {fname}"


@pytest.mark.parametrize("cfg_name", sorted(helpers.CONFIGS))
def test_optimizer_state_indices_follow_the_reference_parameter_order(cfg_name):
    """torch.optim.Adam's state_dict is indexed by position in GaussianDiffusion.parameters() (vddp.py:1455, Trainer.save / load 1548-1585).
    The key lists tests/golden/shapes_*.json were dumped from the real reference's state_dict (module registration order, the shared rotary table
    repeated under every temporal attention): parameters() = that order with the repeats dropped.  The checkpoint index list of the trainer must
    be exactly that -- PreNorm's fn before its norm, `ups` registered before the middle blocks, one slot for the frozen rotary table."""
    import videometamaterials_amd as vm
    from videometamaterials_amd.dp import DataParallelTrainer
    kw, (B, T, H, W), _ = helpers.CONFIGS[cfg_name]
    ref, seen = [], False
    for k in helpers.load_shapes(cfg_name):
        if k.endswith("rotary_emb.freqs"):
            if seen:
                continue
            seen = True
        ref.append(k)
    model = vm.Unet3D(**kw)
    diff = vm.GaussianDiffusion(model, image_size=H, num_frames=T, channels=kw["channels"], timesteps=8, sampling_timesteps=8)
    tr = DataParallelTrainer(diff)
    names = tr._optimizer_param_names()
    assert [model._ref_key(n) if n is not None else "init_temporal_attn.fn.fn.fn.rotary_emb.freqs" for n in names] == ref
    assert [model._ref_key(n) for n, _ in model.named_parameters()] == [k for k in ref if not k.endswith("rotary_emb.freqs")]
    sd = tr.state_dict()
    assert sd["optimizer"]["param_groups"][0]["params"] == list(range(len(ref)))


def test_optimizer_state_of_the_wrong_size_is_rejected():
    import videometamaterials_amd as vm
    from videometamaterials_amd.dp import DataParallelTrainer
    kw, (B, T, H, W), _ = helpers.CONFIGS["plumb16"]
    model = vm.Unet3D(**kw)
    diff = vm.GaussianDiffusion(model, image_size=H, num_frames=T, channels=kw["channels"], timesteps=8, sampling_timesteps=8)
    tr = DataParallelTrainer(diff)
    names = tr._optimizer_param_names()
    params = dict(model.named_parameters())
    obj = tr.state_dict()
    i = names.index("init_conv.weight")
    good = {"step": torch.tensor(3.0), "exp_avg": torch.ones_like(params["init_conv.weight"]).cpu(), "exp_avg_sq": torch.ones_like(params["init_conv.weight"]).cpu()}
    obj["optimizer"]["state"] = {i: good}
    tr.load_state_dict(obj)  # fits
    assert float(tr._moments["init_conv.weight"][0].sum()) == params["init_conv.weight"].numel()
    obj["optimizer"]["state"] = {i + 1: good}  # the same moments one index off (init_conv.bias): must raise, not bind 3 k floats to a 16-float parameter
    with pytest.raises(ValueError, match="does not fit"):
        tr.load_state_dict(obj)
    # equal element counts are not enough: a moment of another SHAPE (a checkpoint written in another parameter order) must not bind
    w = params["init_conv.weight"]
    other = {"step": torch.tensor(3.0), "exp_avg": torch.ones(w.shape[1], w.shape[0], *w.shape[2:]), "exp_avg_sq": torch.ones(w.shape[1], w.shape[0], *w.shape[2:])}
    if other["exp_avg"].shape != w.shape:
        obj["optimizer"]["state"] = {i: other}
        with pytest.raises(ValueError, match="does not fit"):
            tr.load_state_dict(obj)
    # a reference checkpoint whose rotary table is a buffer (one slot fewer, same order): accepted, indices after the slot shift by one
    obj["optimizer"]["param_groups"][0]["params"] = list(range(len(names) - 1))
    last = len(names) - 1
    pl = params[names[last]]
    obj["optimizer"]["state"] = {i: good, last - 1: {"step": torch.tensor(3.0), "exp_avg": torch.full_like(pl, 2.0).cpu(), "exp_avg_sq": torch.ones_like(pl).cpu()}}
    tr.load_state_dict(obj)
    assert float(tr._moments[names[last]][0].sum()) == 2.0 * pl.numel()
    # ... but only when the checkpoint's own `model` entries follow this tree's parameter order: the same one-slot-short count from a checkpoint written
    # in ANOTHER order (two equal-shaped parameters swapped: the per-index shape check cannot see it) is refused
    import collections
    keys = list(obj["model"].keys())
    a, b = keys.index("denoise_fn.downs.0.3.fn.fn.fn.to_k.weight"), keys.index("denoise_fn.downs.0.3.fn.fn.fn.to_v.weight")
    keys[a], keys[b] = keys[b], keys[a]
    swapped = dict(obj, model=collections.OrderedDict((k, obj["model"][k]) for k in keys))
    with pytest.raises(ValueError, match="parameter order"):
        tr.load_state_dict(swapped)
    tr.load_state_dict(dict(obj, model=collections.OrderedDict(("module." + k, v) for k, v in obj["model"].items())))  # (a DDP-wrapped save: same order)
    obj["optimizer"]["param_groups"][0]["params"] = list(range(len(names) - 2))  # any other count is another model
    with pytest.raises(ValueError, match="optimizer state covers"):
        tr.load_state_dict(obj)


@pytest.mark.parametrize("ops", ["all", "conv3x3", "proj", "conv3x3,proj"])
def test_storage_conversions_fit_their_arena_slots(monkeypatch, ops):
    """Sizing-only build (no GPU) of the mirrored "bf16" plan with the unfused temporal attention and kernel families switched back to fp32 I/O
    (VMM_A16_OPS): every vmm_convert_act writes into an allocation that holds all the elements it converts.  (Round-4 advisor: resnet_block's
    mirror_in path cast x1 while self.B was half the batch; the copy was half the size the conversion wrote.)"""
    import videometamaterials_amd as vm
    from videometamaterials_amd import plan as P
    kw, _, _ = helpers.CONFIGS["lagr64"]
    model = vm.Unet3D(**kw)
    model.precision = "bf16"
    model.use_fused_temporal = False
    monkeypatch.setenv("VMM_A16_OPS", ops)
    seen = []

    class Checked(P._Builder):
        def step(self, fn, args, what, flops=0.0, nbytes=0.0):
            if fn.__name__ == "vmm_convert_act":
                _, _, dst, dst_bf, n = args
                off, size = self.arena.log[-1]  # cast() allocates the destination right before the launch
                assert self.ptr(off) == dst, what
                assert size >= (n // 2 if dst_bf else n), (what, size, n, dst_bf)
                seen.append(n)
            return super().step(fn, args, what, flops, nbytes)

    with torch.inference_mode(False), torch.no_grad():
        Checked(model, 2, 11, 96, 96, 11, torch.device("cpu"), (1 << 40, 1 << 41, 1 << 42, 1 << 43), False, True, False).build()
    assert seen  # (the unfused temporal attention has no 2-byte path: the plan does convert)


def test_agent_release_fences_keep_their_wait(tmp_path):
    """ISA check of the ticket hand-over in the 3 x 3 kernel's split epilogue (LABNOTES 9.8): every agent-scope release (buffer_wbl2 sc1) is followed by
    s_waitcnt vmcnt(0) before the barrier / ticket store.  ROCm 7.2 may drop the wait the fence itself implies (MI355X_MICROARCH.md, "Compiler hazard");
    the source spells it out in inline asm, and this compiles the file (hipcc cross-compiles without a GPU) and looks."""
    import shutil
    import subprocess
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import scan_isa
    from videometamaterials_amd import build as b
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    if shutil.which(hipcc) is None and not os.path.exists(hipcc):
        pytest.skip("no hipcc")
    out = tmp_path / "c3.s"
    subprocess.run([hipcc, *b.FLAGS, "-w", "-S", "--cuda-device-only", "-o", str(out), os.path.join(b.CSRC, "conv3x3_bf16x3.hip")], check=True)
    ks = scan_isa.kernels(str(out))
    assert sum(k["wbl2"] for k in ks) > 0  # the split instances are there
    assert sum(k["wbl2_bad"] for k in ks) == 0
    assert scan_isa.unguarded_release_fences(["buffer_wbl2 sc1", "s_barrier"]) == 1  # (the detector itself)
