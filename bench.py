#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

N > 1: one rank per GPU over RCCL.  When the launcher's environment (WORLD_SIZE) is absent, bench.py starts the N ranks itself by
re-executing under `python -m torch.distributed.run --nnodes=1 --nproc-per-node N`; every rank asserts that the process group
really has N members and rank 0 puts `rccl_ranks` on the JSON line.

Workload (BASELINE.json configs[1]): Lagrangian model.yaml denoiser (dim 64, mults 1-2-4-8, 39.5 M parameters, random-init),
3 x 11 x 96 x 96 video, batch 4 per GPU, classifier-free guidance w = 5, dynamic thresholding.
One "step" = one guided ancestral DDPM step p_sample(x_t, t) for the whole batch: the denoiser at batch 2B (conditional +
unconditional branch), x0 prediction, exact 0.9-quantile, posterior update -- what the reference executes 256 times per sample()
call (vddp.py:956-975).  Inputs are resident in HBM.  The default K = 256 steps t = 255 .. 0 is one full sample(); value = sampled
frames/s = n_gpus * B * 11 frames * (K / 256) / elapsed.  `full_sample` repeats the measurement through the public
GaussianDiffusion.sample() call (SURVEY 8(d)'s definition).

Extra objects on the JSON line: `roofline` for the dominant kernel family of the sampling step, `attention` (the attention kernel
families against the matrix-core roof), `fp32_exact` (the same sampler in exact-fp32 arithmetic), `training` = the data-parallel
optimisation step (the other half of BASELINE.json's metric) with its own `roofline_training`, and `cpu_baseline` = the oracle (CPU
restatement of the reference) timed on the host cores.
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

LAGRANGIAN = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3, attn_heads=8, attn_dim_head=32, init_dim=None, init_kernel_size=7,
                  use_sparse_linear_attn=True, resnet_groups=8, cond_bias=True, cond_attention="self-stacked", cond_attention_tokens=16,
                  cond_att_GRU=False, use_temporal_attention_cond=True, cond_to_time="add", per_frame_cond=True, padding_mode="zeros")
B_PER_GPU, T, HW, TIMESTEPS, W_GUIDE = 4, 11, 96, 256, 5.0
PEAK_FP32_MFMA_TFLOPS = 157.3  # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 peak
PEAK_BF16_MFMA_TFLOPS = 2500.0  # dense (AMD's 5 PF headline includes 2:1 sparsity)
PEAK_HBM_GBS = 8000.0
# launcher (family) -> device kernel name prefix in rocprofv3 output
KERNEL_OF = {"vmm_conv3x3_bf16x3": ("conv3x3_x3_kernel", "conv3x3_pw_kernel"), "vmm_conv_igemm_bf16x3": ("igemm_bf16x3_kernel",), "vmm_conv_s2_bf16x3": ("conv_s2_kernel",),
             "vmm_conv_igemm_f32": ("igemm_f32_kernel",), "vmm_temporal_block_bf16x3": ("temporal_block_kernel", "temporal_block2_kernel", "temporal_block128_kernel"),
             "vmm_linattn_block_bf16x3": ("linattn_ctx_kernel", "linattn_combine_kernel", "linattn_apply_kernel"),
             "vmm_temporal_core_bf16x3": ("temporal_core_kernel",), "vmm_proj_bf16x3": ("proj_x3_kernel",), "vmm_conv_wgrad_f32": ("wgrad_f32_kernel",),
             "vmm_conv3x3_f32": ("conv3x3_x3_kernel", "conv3x3_pw_kernel"), "vmm_proj_f32": ("proj_x3_kernel",),
             "vmm_linattn_context": ("linattn_partial_kernel", "linattn_merge_kernel"), "vmm_linattn_apply": ("linattn_apply_mfma_kernel",),
             "vmm_spatial_attention": ("spatial_attn_kernel",), "vmm_spatial_attention_bf16x3": ("spatial_attn_mfma_kernel",),
             "vmm_linattn_context_bf16x3": ("linattn_partial_mfma_kernel", "linattn_merge_kernel")}
ATTENTION_FAMILIES = ("vmm_temporal_block_bf16x3", "vmm_temporal_core_bf16x3", "vmm_linattn_block_bf16x3", "vmm_spatial_attention", "vmm_spatial_attention_bf16x3",
                      "vmm_linattn_context", "vmm_linattn_context_bf16x3", "vmm_linattn_apply", "vmm_temporal_attention")


def _committed(name: str):
    path = os.path.join(ROOT, "profiles", name)
    return json.load(open(path)) if os.path.exists(path) else None


def profile_id(which: str):
    """Which committed rocprofv3 pass the replayed counter fields describe (profiles/latest_id.json: {"traffic": "r05_c", "sq": "r05_c"}); those fields are
    properties of the BUILDER's profiled run of this command, not of the run that prints the line."""
    ids = _committed("latest_id.json") or {}
    return ids.get(which)


def pmc_traffic(family: str):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/traffic_latest.json, made by
    tools/profile_bench.sh on this same bench command: separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950's 128-byte requests).  None when the file or the kernel is missing."""
    prefix, tab = KERNEL_OF.get(family), _committed("traffic_latest.json")
    if not prefix or tab is None:
        return None
    tot, n = 0.0, 0
    for k, v in tab.items():
        if k.startswith(prefix) and "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            ln = v.get("launches_FETCH_SIZE", 1)
            tot += ln * (2.0 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024.0
            n += ln
    return round(tot / n) if n else None


def pmc_mfma_util(family: str):
    """Matrix-pipe utilisation of a kernel family from the committed SQ counter pass (profiles/sq_latest.json, tools/pmc_kernels.sh +
    tools/summarize_sq.py on this bench command): SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x un-profiled kernel duration x 2.4 GHz),
    launch-weighted -- the share of the matrix pipes' nominal issue capacity the family keeps busy, padded products included."""
    prefix, tab = KERNEL_OF.get(family), _committed("sq_latest.json")
    if not prefix or tab is None:
        return None
    busy = cap = 0.0
    for k, v in tab.items():
        if k.startswith(prefix) and v.get("avg_ns_unprofiled"):
            busy += v["launches"] * v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0)
            cap += v["launches"] * 1024.0 * v["avg_ns_unprofiled"] * 2.4
    return round(busy / cap, 4) if cap else None


def usable_cores() -> int:
    """Cores this process may really use: affinity mask and cgroup CPU quota, not the host's core count."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            q = int(open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us").read())
            p = int(open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read())
            if q > 0:
                n = min(n, max(1, q // p))
        except (OSError, ValueError):
            pass
    return max(1, min(n, 64))


def _cpu_model_name() -> str:
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(quick: bool = False):
    """Oracle (oracle/unet3d_oracle.py + diffusion_oracle.py, the parity-pinned CPU restatement of the reference) on the host cores, fp32,
    torch CPU with every usable core, by BASELINE.md section 3.2's protocol: batch 4, one warm-up, then the median of 3 guided p_sample
    steps (w = 5: two denoiser passes + x0 / quantile / posterior) and the median of 3 optimisation steps (forward + backward + Adam), about
    five minutes.  quick = the same at batch 1 scaled x4 (every op of the path is per sample), about a minute: profiling passes only."""
    from oracle import diffusion_oracle as do
    from oracle import unet3d_oracle as uo
    import videometamaterials_amd as vm
    torch.manual_seed(0)
    model = vm.Unet3D(**LAGRANGIAN)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = uo.UnetCfg(**{k: v for k, v in LAGRANGIAN.items()})
    nthreads = usable_cores()
    torch.set_num_threads(nthreads)
    b = 1 if quick else B_PER_GPU
    g = torch.Generator().manual_seed(1)
    x = torch.randn(b, 3, T, HW, HW, generator=g)
    t = torch.randint(0, TIMESTEPS, (b,), generator=g)
    cond = torch.rand(b, 11, generator=g) * 2 - 1
    noise = torch.randn(b, 3, T, HW, HW, generator=g)
    sch = do.schedule_buffers(TIMESTEPS)
    zeros = torch.zeros(b, dtype=torch.bool)

    def timed(fn, n):
        out = []
        for _ in range(n):
            t0 = time.perf_counter()
            fn()
            out.append(time.perf_counter() - t0)
        return out

    with torch.no_grad():
        fwd = timed(lambda: uo.unet3d_forward(sd, cfg, x, t, cond, zeros), 1)  # warm-up (thread pool, allocator); also the forward time
        step = timed(lambda: do.p_sample_step(sch, lambda a, c: uo.unet3d_guided(sd, cfg, a, c, cond, W_GUIDE), x, t, noise), 3)
    sdg = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sd.items()}
    params = [v for v in sdg.values() if v.requires_grad]
    opt = torch.optim.Adam(params, lr=1e-4)

    def train_step():
        opt.zero_grad(set_to_none=True)
        loss = do.p_losses(sch, lambda a, c: uo.unet3d_forward(sdg, cfg, a, c, cond, zeros), x.clamp(-1, 1), t, noise)
        loss.backward()
        opt.step()

    train = timed(train_step, 1 if quick else 4)[0 if quick else 1:]  # (first optimisation step = warm-up of the autograd graph / Adam state)
    step_s, train_s = statistics.median(step), statistics.median(train)
    # BASELINE.json configs[0] -- Unet3D(dim=16, channels=1), 4 frames of 32 x 32, batch 2 -- one optimisation step
    cfg1 = dict(dim=16, channels=1)
    m1 = vm.Unet3D(**cfg1)
    sd1 = {k: v.detach().clone().requires_grad_(not k.endswith("freqs")) for k, v in m1.state_dict().items()}
    c1 = uo.UnetCfg(**cfg1)
    x1, t1, cond1 = torch.rand(2, 1, 4, 32, 32, generator=g) * 2 - 1, torch.randint(0, TIMESTEPS, (2,), generator=g), torch.randn(2, 51, generator=g)
    n1 = torch.randn(2, 1, 4, 32, 32, generator=g)
    opt1 = torch.optim.Adam([v for v in sd1.values() if v.requires_grad], lr=1e-4)
    z1 = torch.zeros(2, dtype=torch.bool)

    def cfg1_step():
        opt1.zero_grad(set_to_none=True)
        do.p_losses(sch, lambda a, c: uo.unet3d_forward(sd1, c1, a, c, cond1, z1), x1, t1, n1).backward()
        opt1.step()

    cfg1_s = statistics.median(timed(cfg1_step, 6)[1:])
    scale = B_PER_GPU / b  # per-sample cost is batch independent
    return {"value": round(B_PER_GPU * T / (TIMESTEPS * step_s * scale), 6), "unit": "frames/s", "cores": nthreads, "cpu_model": _cpu_model_name(),
            "kind": "port", "kind_detail": "oracle (parity-pinned CPU restatement of the reference; the reference itself cannot travel to the GPU box)",
            "sample": (f"oracle (torch CPU fp32, {nthreads} threads), Lagrangian widths, 11x96x96, batch {b}: 1 warm-up forward ({fwd[0]:.1f} s), median of "
                       f"{len(step)} guided p_sample steps (w=5, two denoiser passes + x0 / quantile / posterior) = {step_s:.2f} s, median of {len(train)} "
                       f"optimisation step(s) (forward + backward + Adam) = {train_s:.2f} s" + (f"; scaled x{int(scale)} to batch {B_PER_GPU}" if quick else "")
                       + "; x256 steps for a full sample"),
            "guided_step_s": round(step_s * scale, 3), "guided_step_s_samples": [round(v, 3) for v in step], "batch_measured": b,
            "train_step_s": round(train_s * scale, 3), "train_step_s_samples": [round(v, 3) for v in train],
            "train_denoising_steps_per_sec": round(B_PER_GPU / (train_s * scale), 5), "forward_s": round(fwd[0], 3),
            "cfg1_train_step_s": round(cfg1_s, 4), "cfg1_train_denoising_steps_per_sec": round(2 / cfg1_s, 3),
            "cfg1_note": "BASELINE.md 3.2(d): configs[0] = Unet3D(dim=16, channels=1), 4x32x32, batch 2, one optimisation step (fwd + bwd + Adam), median of 5"}


def _family_times(meta, ms_lists):
    fam = {}
    for ms in ms_lists:
        for (name, fl, by), t_ms in zip(meta, ms):
            f = fam.setdefault(name, [0.0, 0.0, 0.0, 0])
            f[0] += t_ms
            f[1] += fl
            f[2] += by
            f[3] += 1
    return fam


def bench_training(vm, model, diff, dev, dist, world, rank, steps: int, precision: str = "bf16x3", want_roofline: bool = False):
    """One data-parallel optimisation step = q_sample -> denoiser forward -> L1 loss -> hand-written backward -> bucketed RCCL
    all-reduce overlapped with the backward -> multi-tensor Adam (+ EMA every 10 steps); per-GPU batch 4 (model.yaml:2).
    precision "bf16x3" (the measured training arithmetic; the reference trains under fp16 autocast, main.py:34): forward, data gradients and
    the 3 x 3 weight gradients on the split-bf16 matrix cores (fp32-class products, 1.5e-5), the remaining weight gradients exact fp32;
    "fp32": exact-fp32 MFMA everywhere (the parity mode, gradients within 1e-3 of the reference)."""
    from videometamaterials_amd.dp import DataParallelTrainer
    model.static_weights = False
    model.train_precision = precision
    model.train()
    tr = DataParallelTrainer(diff, train_lr=1e-4)
    assert tr.world == world
    rehearse = tr.engine is not None or os.environ.get("VMM_DP_FORCE") == "1"  # one rank through the collective path (first contact with RCCL)
    selfcheck = tr.rccl_selfcheck() if (world > 1 or rehearse) else None
    g = torch.Generator().manual_seed(100 + rank)
    x = torch.rand(B_PER_GPU, 3, T, HW, HW, generator=g).to(dev)
    cond = (torch.rand(B_PER_GPU, 11, generator=g) * 2 - 1).to(dev)
    for _ in range(2):
        tr.train_step(x, cond)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(x, cond)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    el = time.perf_counter() - t0
    if dist is not None:
        et = torch.tensor([el], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        el = float(et.item())
    ms = el / steps * 1e3
    pl = tr._plan
    comm = None
    if world > 1 or rehearse:
        # one more (untimed) step with events on the side stream: how long the buckets keep it busy and how much of that lies inside the
        # backward window (bucket i starts when the backward marks its tail slice final)
        tr._reducer.timing = True
        tr.train_step(x, cond)
        torch.cuda.synchronize()
        comm = tr._reducer.last_step_timing()
        tr._reducer.timing = False
    fl = sum(f for _, f, _ in pl.meta) + sum(f for _, f, _ in pl.bwd_meta)
    out = {"optimizer_steps_per_sec": round(1e3 / ms, 4), "denoising_steps_per_sec": round(world * B_PER_GPU * 1e3 / ms, 3), "ms_per_step": round(ms, 2),
           "batch_per_gpu": B_PER_GPU, "global_batch": world * B_PER_GPU,
           "arithmetic": ("fp32 (exact fp32 MFMA)" if precision == "fp32"
                          else "fp16: ONE MFMA pass on fp16-rounded operands (the reference's autocast dtype, main.py:34) in forward, data gradients, 3x3 / 1x1 / to_qkv weight "
                               "gradients, the generic implicit GEMM and the recomputing attention backward (whose qkv-row gradient travels as fp16 operands: same operand "
                               "bits, half the bytes); device-side GradScaler (2^16, backoff 0.5 on a non-finite gradient, growth 2 per 2000 clean steps: vddp.py:1629-1633); "
                               "fp32 master weights, activations, accumulation, norms, softmax, Adam; the 4x4 / 7x7 weight gradients on the same operands (tap-decoding 1x1 kernel)" if precision == "fp16"
                          else "bf16: ONE MFMA pass on bf16-rounded operands in forward, data gradients, 3x3 / 1x1 / to_qkv weight gradients and the recomputing attention "
                               "backward; fp32 master weights, activations, accumulation, norms, softmax, Adam; the 4x4 / 7x7 weight gradients on the same operands" if precision == "bf16"
                          else "bf16x3: forward, data gradients and all convolution / projection weight gradients split-bf16 MFMA (fp32-class)"),
           "loss": float(loss), "gemm_TFLOP_per_step": round(fl / 1e12, 3), "achieved_gemm_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1),
           "arena_GB": round(pl.arena_floats * 4 / 1e9, 2), "launches_per_step": len(pl.steps) + len(pl.bwd_steps),
           "launches_note": "entry-point calls of the plan; kernel launches per step by rocprofv3: profiles/r06_c_train_kernels_*.txt (735)",
           "attention": ("fused blocks forward (no qkv rows / attention outputs / softmax statistics stored), recomputing backward kernels at the C = 64 sites"
                         if any("_block_bwd_" in fn.__name__ for fn, _, _ in pl.bwd_steps) else "unfused (qkv rows through HBM)"),
           "grad_allreduce_MB": round(pl.pgrad_floats * 4 / 1e6, 1), "allreduce_buckets": len(tr._reducer.launched) if (world > 1 or rehearse) else 0,
           "dp_engine": (f"native: vmm_dp C ABI over RCCL {tr.engine.rccl_version} (include/vmm_dp.h)" if tr.engine is not None
                         else f"torch.distributed/{dist.get_backend()}" if dist is not None else "none (single rank)")}
    if precision == "fp16":
        out["loss_scale"] = tr.loss_scale_state()  # the device-side GradScaler after the timed steps (scale, steps skipped for a non-finite gradient)
    if comm:
        out.update(allreduce_ms=comm["allreduce_ms"], overlap_frac=comm["overlap_frac"], comm_detail=comm)
    if selfcheck:
        out["rccl_selfcheck"] = selfcheck
    if want_roofline and rank == 0:
        # per-launch HIP events over one forward + backward of the training plan (same stream as the launches)
        fwd_ms, bwd_ms = pl.launch_timed(), pl.backward_timed()
        fam = _family_times(pl.meta, [fwd_ms])
        for k, v in _family_times(pl.bwd_meta, [bwd_ms]).items():
            a = fam.setdefault(k, [0.0, 0.0, 0.0, 0])
            for i in range(4):
                a[i] += v[i]
        dom = max((k for k in fam if fam[k][1] > 0), key=lambda k: fam[k][0])
        objs = {}
        for k in {dom, "vmm_conv_wgrad_f32", "vmm_conv3x3_wgrad_bf16x3", "vmm_conv3x3_f32" if precision == "fp32" else "vmm_conv3x3_bf16x3"} & set(fam):
            t_ms, flops, nbytes, n = fam[k]
            peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if k.endswith("bf16x3") else PEAK_FP32_MFMA_TFLOPS
            ach = flops / (t_ms * 1e-3) / 1e12
            objs[k] = {"bound": "mfma", "achieved": round(ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                       "launches_per_step": n, "ms_per_step": round(t_ms, 3), "avg_launch_ms": round(t_ms / n, 4), "algorithmic_GFLOP_per_step": round(flops / 1e9, 1),
                       "algorithmic_GB_per_step": round(nbytes / 1e9, 2)}
        out["roofline_training"] = {"dominant": dom, "kernels": objs, "event_ms_forward": round(sum(fwd_ms), 2), "event_ms_backward": round(sum(bwd_ms), 2),
                                    "ms_by_kernel_family": {k: round(v[0], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])[:20]},
                                    "peak_note": "fp32 MFMA 157.3 TFLOP/s (v_mfma_f32_32x32x2_f32); split-bf16 kernels: 2500 / 3"}
    if tr.engine is not None:
        tr.engine.close()  # the next leg builds its own communicator
    return out


def bench_cfg1(vm, dev, steps: int = 20):
    """BASELINE.json configs[0] (the reference's own CPU-runnable plumbing case: Unet3D(dim=16, channels=1), 4 frames of 32 x 32, batch 2) on the GPU: one
    optimisation step (forward + backward + Adam) through the same trainer, in the drop-in's default training arithmetic.  Launch-bound at this size."""
    from videometamaterials_amd.dp import DataParallelTrainer
    torch.manual_seed(0)
    m = vm.Unet3D(dim=16, channels=1).to(dev)
    d = vm.GaussianDiffusion(m, image_size=32, num_frames=4, channels=1, timesteps=TIMESTEPS, loss_type="l1", sampling_timesteps=TIMESTEPS).to(dev)
    tr = DataParallelTrainer(d, train_lr=1e-4, engine="torch")
    g = torch.Generator().manual_seed(7)
    x, c = torch.rand(2, 1, 4, 32, 32, generator=g).to(dev), torch.randn(2, 51, generator=g).to(dev)
    for _ in range(3):
        tr.train_step(x, c)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = tr.train_step(x, c)
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / steps * 1e3
    return {"workload": "configs[0]: Unet3D(dim=16, channels=1), 4x32x32, batch 2, one optimisation step (fwd + bwd + Adam)", "ms_per_step": round(ms, 3),
            "denoising_steps_per_sec": round(2e3 / ms, 1), "arithmetic": m.train_precision, "loss": float(loss),
            "launches_per_step": len(tr._plan.steps) + len(tr._plan.bwd_steps)}


HIRES = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
             per_frame_cond=False)


def bench_config4(vm, dev, timed_region, world, B: int = 8, frames: int = 22, size: int = 192):
    """BASELINE.json configs[3]: 22 frames of 192 x 192, batch 8 per GPU, Lagrangian widths with the CNN signal embedding (random init): the
    denoiser forward and one guided DDPM step (denoiser at batch 16), in the parity arithmetic (fp32 activations, split-bf16 MFMA) and in
    the bf16 throughput mode when the model has one.  Skipped (None) when less than 40 GB of HBM is free."""
    free, _ = torch.cuda.mem_get_info(dev)
    if free < 40e9:
        return {"skipped": f"only {free / 1e9:.0f} GB of HBM free"}
    torch.manual_seed(0)
    m = vm.Unet3D(**HIRES).to(dev).eval()
    diff = vm.GaussianDiffusion(m, image_size=size, num_frames=frames, channels=3, timesteps=TIMESTEPS, use_dynamic_thres=True, sampling_timesteps=TIMESTEPS).to(dev)
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 3, frames, size, size, generator=g).to(dev)
    t = torch.randint(0, TIMESTEPS, (B,), generator=g).to(dev)
    cond = (torch.rand(B, 51, generator=g) * 2 - 1).to(dev)
    out = {"workload": f"configs[3]: {frames}x{size}x{size}, batch {B} per GPU, dim 64 (random init), CNN signal embedding + 16 tokens", "batch_per_gpu": B}
    modes = [("bf16x3", "fp32 activations in HBM, split-bf16 MFMA (parity mode, 2e-4 vs the oracle at this size)")]
    if "bf16" in getattr(vm.Unet3D, "PRECISIONS", ()):
        modes.append(("bf16", "bf16 activations in HBM at the two upper levels (192 x 192 and 96 x 96: one rounding per stored element), ONE MFMA pass on "
                              "bf16-rounded operands, fp32 accumulation / norms / softmax (throughput mode, 2e-2 vs the oracle at this size)"))
        modes.append(("bf16_fp32_storage", "the same single-pass arithmetic with fp32 activations in HBM (Unet3D.bf16_storage = False): what the storage type alone buys"))
    with torch.no_grad():
        for key, note in modes:
            prec = "bf16" if key.startswith("bf16") and key != "bf16x3" else key
            m.precision = prec
            m.bf16_storage = key != "bf16_fp32_storage"
            m._plans.clear()
            n_f, n_s = 4, 3
            m(x, t, cond=cond, null_cond_prob=0.0)
            diff.p_sample(x, t, cond=cond, guidance_scale=W_GUIDE)
            fwd = timed_region(lambda: [m(x, t, cond=cond, null_cond_prob=0.0) for _ in range(n_f)]) / n_f
            stp = timed_region(lambda: [diff.p_sample(x, t, cond=cond, guidance_scale=W_GUIDE) for _ in range(n_s)]) / n_s
            plan = m.get_plan(B, frames, size, size, 51, dev)
            fl = sum(f for _, f, _ in plan.meta)
            out[key] = {"arithmetic": note, "denoiser_forward_ms": round(fwd * 1e3, 2), "forward_TFLOPs": round(fl / fwd / 1e12, 1),
                         "guided_step_ms": round(stp * 1e3, 2), "sampled_frames_per_sec": round(world * B * frames / (stp * TIMESTEPS), 3),
                         "launches_per_forward": len(plan.meta), "plan_GB": round(plan.arena_floats * 4 / 1e9, 1),
                         "storage_conversions": sum(1 for fn, _, _ in plan.steps if fn.__name__ == "vmm_convert_act")}
    m.bf16_storage = True
    m._plans.clear()
    del m, diff
    torch.cuda.empty_cache()
    return out


def preflight(vm, dev, dist, world, rank, local_rank, backend) -> dict:
    """`bench.py --gpus N --preflight`: everything the N-rank run touches before its first timed step, on a small model, with a verdict per item --
    so that a first contact with a multi-GPU node that goes wrong says WHERE (launcher / binding / rendezvous / collective / engine / step).
    Under VMM_DIST_BACKEND=gloo several ranks may share one GPU (RCCL refuses duplicate devices): the native engine's communicator is then
    skipped, its rendezvous-id exchange is not."""
    from videometamaterials_amd import dp as _dp
    checks = {"launcher": {"RANK": rank, "LOCAL_RANK": local_rank, "WORLD_SIZE": world, "MASTER_ADDR": os.environ.get("MASTER_ADDR"),
                           "HSA_ENABLE_IPC_MODE_LEGACY": os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"), "backend": backend if world > 1 else None}}
    ok = True
    ndev = torch.cuda.device_count()
    checks["device_binding"] = {"visible_gpus": ndev, "this_rank": torch.cuda.current_device(), "name": torch.cuda.get_device_name(dev),
                                "one_gpu_per_rank": backend != "nccl" or ndev >= world}
    ok &= checks["device_binding"]["one_gpu_per_rank"]
    if dist is not None:
        # every rank's (rank, device) through the backend: the all-gathered table must list each rank once
        t = torch.tensor([rank, torch.cuda.current_device()], device=dev if backend == "nccl" else "cpu", dtype=torch.int64)
        got = [torch.zeros_like(t) for _ in range(world)]
        dist.all_gather(got, t)
        table = [tuple(int(v) for v in g.tolist()) for g in got]
        checks["all_gather"] = {"rank_device_table": table, "ok": [r for r, _ in table] == list(range(world))}
        ok &= checks["all_gather"]["ok"]
        # the native engine's rendezvous: rank 0 asks RCCL for a unique id (if RCCL loads), every rank receives the same 128 bytes
        try:
            box = [_dp.RcclEngine.new_unique_id() if rank == 0 else None]
            err = None
        except Exception as e:  # noqa: BLE001 -- the report carries the reason
            box, err = [None], repr(e)
        dist.broadcast_object_list(box, src=0)
        same = None
        if box[0] is not None:
            import hashlib
            digest = hashlib.sha256(box[0]).hexdigest()[:16]
            all_d = [None] * world
            dist.all_gather_object(all_d, digest)
            same = len(set(all_d)) == 1
            ok &= same
        checks["rccl_unique_id"] = {"obtained_on_rank0": box[0] is not None, "error": err, "same_on_every_rank": same}
    # a small data-parallel step through the trainer (torch.distributed collectives; the native engine when every rank owns a GPU)
    torch.manual_seed(0)
    small = vm.Unet3D(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True, per_frame_cond=True,
                      cond_bias=True).to(dev)
    small.train_precision = "bf16x3"
    d_small = vm.GaussianDiffusion(small, image_size=32, num_frames=T, channels=3, timesteps=TIMESTEPS, loss_type="l1", sampling_timesteps=TIMESTEPS).to(dev)
    engines = ["torch"] + (["native"] if (dist is None or backend == "nccl") else [])
    checks["train_step"] = {}
    for eng in engines:
        try:
            tr = _dp.DataParallelTrainer(d_small, train_lr=1e-4, engine=eng)
            g = torch.Generator().manual_seed(10 + rank)
            x = torch.rand(2, 3, T, 32, 32, generator=g).to(dev)
            c = (torch.rand(2, 11, generator=g) * 2 - 1).to(dev)
            sc = tr.rccl_selfcheck() if (world > 1 or tr.engine is not None) else None
            l0 = float(tr.train_step(x, c))
            tr._reducer.timing = world > 1 or tr.engine is not None
            l1 = float(tr.train_step(x, c))
            torch.cuda.synchronize()
            comm = tr._reducer.last_step_timing() if tr._reducer.timing else None
            # the replicas must still agree after two steps: checksum of the parameters, max - min over ranks
            cs = torch.stack([p.detach().double().sum() for p in small.parameters()]).sum().reshape(1)
            spread = 0.0
            if dist is not None:
                lo, hi = cs.clone(), cs.clone()
                if backend != "nccl":
                    lo, hi = lo.cpu(), hi.cpu()
                dist.all_reduce(lo, op=dist.ReduceOp.MIN)
                dist.all_reduce(hi, op=dist.ReduceOp.MAX)
                spread = float((hi - lo).abs().item())
            good = l0 == l0 and l1 == l1 and spread == 0.0 and (sc is None or sc["ok"])
            checks["train_step"][eng] = {"loss": [round(l0, 5), round(l1, 5)], "selfcheck": sc, "replica_checksum_spread": spread,
                                         "buckets": len(tr._reducer.launched) if comm else 0,
                                         "bucket_spans_ms": comm.get("bucket_spans_ms") if comm else None, "overlap_frac": comm.get("overlap_frac") if comm else None, "ok": bool(good)}
            ok &= bool(good)
            del tr
        except Exception as e:  # noqa: BLE001
            checks["train_step"][eng] = {"ok": False, "error": repr(e)}
            ok = False
    return {"preflight": True, "n_gpus": world, "ok": bool(ok), "checks": checks}


def clock_bound_probe(dev, rep: int = 40):
    """Is the dominant kernel bound by its instruction stream or by the chip's clock / power management?  ONE representative launch of the 3 x 3 kernel
    (512 -> 512 at 12 x 12, the sampler's batch of 8 x 11 frames) on random operands and on all-zero operands: identical code object, grid and
    instruction stream, only the bits toggling in the matrix pipe differ (MI355X_MICROARCH.md "DVFS give-back"; tools/bench_c3_data.py, LABNOTES 10.5).
    HIP events on the launch stream; a measurement of the kernel's environment, not part of any throughput number."""
    import ctypes as C
    import math
    from videometamaterials_amd import _native as N
    lib = N.lib()
    s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)  # noqa: E731
    nimg, H, Cin, Cout = 88, 12, 512, 512
    g = torch.Generator(device=dev).manual_seed(5)
    out = torch.zeros(nimg * H * H, Cout, device=dev)
    res = {}
    for kind in ("random", "zero", "random"):
        x = torch.randn(nimg * H * H, Cin, generator=g, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) / math.sqrt(Cin * 9)).contiguous()
        if kind == "zero":
            x.zero_()
            w.zero_()
        nfl = Cout * 9 * Cin
        packed = torch.zeros(nfl, device=dev)
        job = (N.PackJob * 1)()
        j = job[0]
        j.torch_w, j.packed = w.data_ptr(), packed.data_ptr()
        j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
        j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 2
        tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
        N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, nfl, 0, s()), "pack")
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1, d.w, d.out, d.ldo = x.data_ptr(), Cin, Cin, packed.data_ptr(), out.data_ptr(), Cout
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, H, H, H, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, H, 1, Cout, 32, 1.0
        d.a_imgs_per_sample = 11
        for _ in range(5):
            N.check(lib.vmm_conv3x3_bf16x3(C.byref(d), s()), "conv3x3 probe")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(rep):
            lib.vmm_conv3x3_bf16x3(C.byref(d), s())
        e1.record()
        torch.cuda.synchronize()
        res.setdefault(kind, []).append(e0.elapsed_time(e1) / rep * 1e3)
    fl = 2.0 * 9 * Cin * Cout * nimg * H * H
    r_us, z_us = min(res["random"]), res["zero"][0]
    return {"layer": "3x3 512->512, 12x12, 88 frames (one launch of the dominant family)", "random_operands_us": round(r_us, 1), "zero_operands_us": round(z_us, 1),
            "random_TFLOPs": round(fl / r_us / 1e6, 1), "zero_TFLOPs": round(fl / z_us / 1e6, 1), "zero_over_random_speed": round(r_us / z_us, 3),
            "frac_of_roof_on_zero_operands": round(fl / z_us / 1e6 / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4),
            "note": "same code object and grid; what real split-bf16 operand data costs is the clock governor's answer to the matrix pipe's switching activity"}


def _respawn_under_torchrun(n: int) -> None:
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU)."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1", "--master-port", str(port),
           os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL across processes needs it on this driver
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=TIMESTEPS, help="guided p_sample steps in the timed region (256 = one full sample)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline-quick", action="store_true", help="CPU baseline at batch 1 scaled x4 (a minute) instead of BASELINE.md 3.2's batch-4 protocol")
    ap.add_argument("--no-config4", action="store_true", help="skip the configs[3] leg (22 x 192 x 192, batch 8)")
    ap.add_argument("--no-graph", action="store_true", help="eager launches instead of the captured hipGraph step (profiling passes)")
    ap.add_argument("--detail", action="store_true", help="per-launch timing table on stderr")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement")
    ap.add_argument("--no-extras", action="store_true", help="skip the sample() call and the exact-fp32 sampler (profiling passes)")
    ap.add_argument("--preflight", action="store_true", help="multi-rank contact check: respawn, device binding, rendezvous-id exchange, one all-reduce, "
                                                             "one small data-parallel training step; prints a JSON report and exits before any timing")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        _respawn_under_torchrun(args.gpus)
    # stdout carries ONE JSON line and nothing else: RCCL prints a version banner with printf when a communicator is created, other
    # libraries may do the like.  File descriptor 1 points at stderr for the whole run; the line goes to the saved descriptor at the end.
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        raise SystemExit(f"bench.py --gpus {args.gpus} was launched with WORLD_SIZE={world}: one rank per GPU, both must agree")
    backend = os.environ.get("VMM_DIST_BACKEND", "nccl")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        ndev = torch.cuda.device_count()
        # one process per GPU over RCCL.  (VMM_DIST_BACKEND=gloo lets a single-GPU box rehearse the multi-rank control flow with
        # several ranks sharing cuda:0 -- RCCL itself refuses duplicate devices; numbers from such a run are meaningless.)
        if backend == "nccl" and ndev < world:
            raise SystemExit(f"--gpus {world} needs {world} visible GPUs (found {ndev}): one RCCL rank per GPU")
        dev_index = local_rank % ndev
        torch.cuda.set_device(dev_index)
        # (device_id: the communicator is bound to this rank's GPU at once -- no guessing from the global rank at the first barrier)
        dist.init_process_group(backend, rank=rank, world_size=world, **({"device_id": torch.device("cuda", dev_index)} if backend == "nccl" else {}))
        assert dist.get_world_size() == args.gpus
    elif os.environ.get("VMM_DP_FORCE") == "1":
        # one-GPU rehearsal of the N-rank control flow THROUGH RCCL: a one-rank process group, every collective of the training leg issued
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        dev_index = 0
        torch.cuda.set_device(0)
        dist.init_process_group(backend, rank=0, world_size=1, **({"device_id": torch.device("cuda", 0)} if backend == "nccl" else {}))
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    import videometamaterials_amd as vm
    if args.preflight:
        report = preflight(vm, dev, dist, world, rank, local_rank, backend)
        if rank == 0:
            os.write(json_fd, (json.dumps(report) + "\n").encode())
        if dist is not None:
            dist.barrier()
            dist.destroy_process_group()
        if not report["ok"]:
            raise SystemExit(1)
        return
    torch.manual_seed(0)  # identical random-init weights on every rank
    model = vm.Unet3D(**LAGRANGIAN).to(dev).eval()
    diff = vm.GaussianDiffusion(model, image_size=HW, num_frames=T, channels=3, timesteps=TIMESTEPS, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=TIMESTEPS).to(dev)
    g = torch.Generator().manual_seed(2 + rank)
    # rows of data/target_responses.csv are min-max normalised to [-1, 1] by the reference; synthetic stand-in of that range
    cond = (torch.rand(B_PER_GPU, 11, generator=g) * 2 - 1).to(dev)
    shape = (B_PER_GPU, 3, T, HW, HW)
    x_T = torch.randn(shape, generator=g).to(dev)

    def timed_region(fn):
        """barrier + synchronize on both sides; MAX over ranks."""
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        fn()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        if dist is not None:
            et = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(et, op=dist.ReduceOp.MAX)
            el = float(et.item())
        return el

    def sampler_leg(steps, warmup, w=W_GUIDE, ddim=False, d_=None):
        """K guided p_sample steps t = 255, 254, ... on the sampler's own captured step (GaussianDiffusion._graphed_step)."""
        d_ = diff if d_ is None else d_
        model.refresh_plans()
        model.static_weights = True
        d_.use_graph = not args.no_graph
        stepper = d_._graphed_step(shape, cond, float(w), ddim=ddim)
        if args.no_graph:
            stepper.captured = True  # stay eager
        ts = list(reversed(range(TIMESTEPS)))
        state = {"img": x_T.clone()}

        if ddim:  # the sampler's own time list (vddp.py:990-991), cycled
            from videometamaterials_amd import hostmath
            pairs = hostmath.ddim_time_pairs(d_.num_timesteps, d_.sampling_timesteps)

        def run(n, offset):
            for j in range(n):
                if ddim:
                    tm, tn = pairs[(offset + j) % len(pairs)]
                    state["img"] = stepper(state["img"], tm, nxt=tn)
                else:
                    state["img"] = stepper(state["img"], ts[(offset + j) % TIMESTEPS])
        with torch.inference_mode():
            run(warmup, 0)
            state["img"] = x_T.clone()  # the timed steps start from pure noise at t = 255, like sample()
            el = timed_region(lambda: run(steps, 0))
        if not args.no_graph and stepper.graph is None:
            raise SystemExit("hipGraph capture of the sampling step failed: refusing to report an eager number as the graphed one")
        model.static_weights = False
        return el, stepper, state["img"]

    elapsed, stepper, img = sampler_leg(args.steps, args.warmup)
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * B_PER_GPU * T * (args.steps / TIMESTEPS) / elapsed
    finite = bool(torch.isfinite(img).all().item())

    full_sample = fp32_exact = bf16_mode = None
    if not args.no_extras:
        # the public call: GaussianDiffusion.sample() = 256 steps + unnormalise, draws its own x_T (vddp.py:965-984)
        diff.use_graph = not args.no_graph
        out_holder = {}
        el = timed_region(lambda: out_holder.__setitem__("v", diff.sample(cond=cond, guidance_scale=W_GUIDE)))
        v = out_holder["v"]
        full_sample = {"seconds": round(el, 3), "frames_per_sec": round(world * B_PER_GPU * T / el, 4), "output_shape": list(v.shape),
                       "output_finite": bool(torch.isfinite(v).all().item()), "output_mean": round(float(v.mean()), 4)}
        # the same sampler in exact-fp32 arithmetic (v_mfma_f32_32x32x2_f32 everywhere), a shorter run
        model.precision = "fp32"
        n32 = max(2, min(args.steps, 16))
        el32, st32, _ = sampler_leg(n32, 2)
        model.precision = "bf16x3"
        fp32_exact = {"ms_per_step": round(el32 / n32 * 1e3, 3), "frames_per_sec": round(world * B_PER_GPU * T / (TIMESTEPS * el32 / n32), 4), "steps": n32,
                      "arithmetic": "exact fp32 MFMA (v_mfma_f32_32x32x2_f32), 1e-6 parity", "hipgraph": st32.graph is not None}
        # ... and in the single-pass bf16 THROUGHPUT mode (the dtype BASELINE.json states for configs[3]; 2e-2 on the denoiser output, NOT a parity number:
        # the headline `value` above stays the fp32-class split-bf16 path)
        bf16_mode = None
        if "bf16" in getattr(vm.Unet3D, "PRECISIONS", ()):
            model.precision = "bf16"
            n16 = max(2, min(args.steps, 32))
            el16, st16, img16 = sampler_leg(n16, 2)
            model.precision = "bf16x3"
            bf16_mode = {"ms_per_step": round(el16 / n16 * 1e3, 3), "frames_per_sec": round(world * B_PER_GPU * T / (TIMESTEPS * el16 / n16), 4), "steps": n16,
                         "arithmetic": "one MFMA pass on bf16-rounded operands in the 3x3 / stride-2 convolutions, projections and fused attention blocks; bf16-stored feature maps at the two upper levels; "
                                       "fp32 accumulation, norms, softmax (2e-2 relative on the denoiser output)",
                         "hipgraph": st16.graph is not None, "output_finite": bool(torch.isfinite(img16).all().item())}

    # ---- BASELINE.json configs[4]: the guidance sweep w in {0, 1, 3, 5} (vddp.py:715-728: w == 1 runs the conditional branch alone -- a B-row
    # captured step --, every other w both branches as one 2B-row batch), and the DDIM sampler's captured step (vddp.py:986-1018)
    guidance_sweep = None
    if not args.no_extras:
        guidance_sweep = {}
        nsw = max(2, min(args.steps, 16))
        for w_ in (0.0, 1.0, 3.0, 5.0):
            el_, st_, img_ = sampler_leg(nsw, 2, w=w_)
            guidance_sweep[f"w={w_:g}"] = {"ms_per_step": round(el_ / nsw * 1e3, 3), "frames_per_sec": round(world * B_PER_GPU * T / (TIMESTEPS * el_ / nsw), 4),
                                          "denoiser_rows": st_.plan.shape[0], "launches_per_step": len(st_.plan.steps) + 4, "hipgraph": st_.graph is not None,
                                          "output_finite": bool(torch.isfinite(img_).all().item())}
        guidance_sweep["w1_over_w5"] = round(guidance_sweep["w=1"]["ms_per_step"] / guidance_sweep["w=5"]["ms_per_step"], 3)
        diff_ddim = vm.GaussianDiffusion(model, image_size=HW, num_frames=T, channels=3, timesteps=TIMESTEPS, loss_type="l1", use_dynamic_thres=True,
                                         sampling_timesteps=50).to(dev)
        el_, st_, img_ = sampler_leg(nsw, 2, ddim=True, d_=diff_ddim)
        guidance_sweep["ddim50_w=5"] = {"ms_per_step": round(el_ / nsw * 1e3, 3), "frames_per_sec": round(world * B_PER_GPU * T / (50 * el_ / nsw), 4),
                                        "hipgraph": st_.graph is not None, "output_finite": bool(torch.isfinite(img_).all().item())}
        del diff_ddim

    # ---- second half of BASELINE.json's metric: training denoising steps/s (configs[2]: per-GPU batch 4, fp32, Adam, RCCL all-reduce)
    train = None
    if not args.no_train:
        nst = max(2, min(args.steps, 6))
        train = bench_training(vm, model, diff, dev, dist, world, rank, steps=nst, want_roofline=True)
        # the same step in exact-fp32 arithmetic (the parity mode: every gradient within 1e-3 of the reference's fp32 autograd)
        train["fp32_parity_variant"] = bench_training(vm, model, diff, dev, dist, world, rank, steps=nst, precision="fp32")
        # the reduced-precision leg, BESIDE the headline: one matrix pass on bf16-rounded operands in every contraction that has such an instance
        # (forward, data and weight gradients; fp32 master weights, activations, accumulation, norms, softmax, Adam) -- the counterpart of the
        # reference's own recipe, fp16 autocast (main.py:34).  Gradient deviation against fp32 autograd: inside the reference's measured bf16-autocast
        # deviation at both widths, inside its fp16-autocast deviation at dim 16 (tests/test_gpu_train.py, tests/golden/autocast_lagr*.json)
        train["reduced_precision_variant"] = bench_training(vm, model, diff, dev, dist, world, rank, steps=nst, precision="bf16")
        # the reference's OWN training arithmetic (main.py:34 mixed_precision='fp16', vddp.py:1619-1633 GradScaler): fp16 operands on the matrix cores in one
        # pass, fp32 accumulation / master weights / activations, static loss scale with the non-finite check that skips the step
        if "fp16" in getattr(vm.Unet3D, "PRECISIONS", ()):
            train["reference_precision_variant"] = bench_training(vm, model, diff, dev, dist, world, rank, steps=nst, precision="fp16")
        model.train_precision = "bf16x3"
        model.eval()
        for k in [k for k, v in model._plans.items() if v.training]:  # the training plans keep every intermediate (19 GB at batch 4):
            del model._plans[k]                                       # released before the configs[3] leg
        torch.cuda.empty_cache()

    cfg1_gpu = bench_cfg1(vm, dev) if (not args.no_train and not args.no_extras and world == 1 and dist is None) else None
    # ---- BASELINE.json configs[3]: 22 frames x 192 x 192, batch 8 per GPU (the HBM stress configuration)
    config4 = None
    if not args.no_config4 and not args.no_extras:
        config4 = bench_config4(vm, dev, timed_region, world)

    if rank == 0:
        # ---- per-kernel-family timing with HIP events on the launch stream (separate, un-timed pass)
        pl = stepper.plan
        model.refresh_plans()
        reps = 3
        fam = _family_times(pl.meta, [pl.launch_timed() for _ in range(reps)])
        if args.detail:
            ms = pl.launch_timed()
            for t_ms, (_, _, what), (f_, fl, by) in sorted(zip(ms, pl.steps, pl.meta), key=lambda r: -r[0]):
                print(f"  {t_ms:8.3f} ms  {fl / max(t_ms, 1e-9) / 1e9:8.1f} TFLOP/s  {by / max(t_ms, 1e-9) / 1e6:8.1f} GB/s  {what}", file=sys.stderr)
        total_ms = sum(v[0] for v in fam.values())
        fwd_ms = total_ms / reps
        dom = max(fam, key=lambda k: fam[k][0])
        d_ms, d_fl, d_by, d_n = fam[dom]
        ach_tflops = d_fl / (d_ms * 1e-3) / 1e12
        ach_gbs = d_by / (d_ms * 1e-3) / 1e9
        # fp32 MFMA peak for the exact kernel; for the split-bf16 kernel every algorithmic flop costs three bf16 MFMA flops,
        # so its roof for ALGORITHMIC flops is the dense bf16 peak / 3
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if dom.endswith("bf16x3") else PEAK_FP32_MFMA_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach_tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach_tflops / peak, 4), "traffic": pmc_traffic(dom), "traffic_profile": profile_id("traffic"), "mfma_pipe_util_profile": profile_id("sq"),
                    "traffic_note": "HBM bytes per launch, mean over the family, REPLAYED from the committed rocprofv3 PMC passes of this command on the builder's box "
                                    "(profiles/<traffic_profile>_traffic.json = traffic_latest.json; mfma_pipe_util likewise from profiles/<mfma_pipe_util_profile>_sq.json); "
                                    "achieved / frac / avg_launch_ms are measured live in this run",
                    "peak_note": "dense bf16 MFMA 2500 TFLOP/s / 3 passes (split-bf16 operands, fp32-class result)" if dom.endswith("bf16x3")
                    else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                    "launches_per_step": d_n // reps, "avg_launch_ms": round(d_ms / d_n, 4),
                    "algorithmic_GFLOP_per_step": round(d_fl / reps / 1e9, 1), "algorithmic_GB_per_step": round(d_by / reps / 1e9, 2),
                    "algorithmic_bytes_per_launch": round(d_by / d_n), "hbm_frac_of_8TBs": round(ach_gbs / PEAK_HBM_GBS, 4),
                    "mfma_pipe_util": pmc_mfma_util(dom), "share_of_denoiser_time": round(d_ms / total_ms, 3)}
        if dom == "vmm_conv3x3_bf16x3" and not args.no_extras:
            roofline["clock_bound_probe"] = clock_bound_probe(dev)
        attention = {}
        for k in ATTENTION_FAMILIES:
            if k in fam and fam[k][0] > 0:
                a_ms, a_fl, a_by, a_n = fam[k]
                attention[k] = {"ms_per_step": round(a_ms / reps, 3), "launches_per_step": a_n // reps,
                                "achieved_TFLOPs": round(a_fl / (a_ms * 1e-3) / 1e12, 1) if a_fl else None,
                                "frac_of_bf16x3_roof": round(a_fl / (a_ms * 1e-3) / 1e12 / (PEAK_BF16_MFMA_TFLOPS / 3.0), 4) if a_fl else None,
                                "achieved_GBs": round(a_by / (a_ms * 1e-3) / 1e9, 1), "mfma_pipe_util": pmc_mfma_util(k)}
        families = {k: round(v[0] / reps, 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])}
        cpu = None
        if not args.no_cpu_baseline and world == 1:
            cpu = cpu_baseline(quick=args.cpu_baseline_quick)
        # the training half of the metric is quoted on the reference's OWN arithmetic (main.py:34: fp16 autocast + loss scaling) when the library has that
        # leg: fp16 operands, fp32 accumulation / master weights, GradScaler on the device; per-parameter gradient deviation inside the reference's own
        # fp16-autocast figures at both widths (tests/test_gpu_train.py).  The split-bf16 step (fp32-class, the drop-in's default) stays in `training`.
        ref_leg = (train or {}).get("reference_precision_variant")
        tr_var = ("fp16 operands + loss scaling = the reference's own training arithmetic (main.py:34); split-bf16 (fp32-class, the drop-in's default) = train_fp32class_ms_per_step"
                  if ref_leg else "split-bf16 (fp32-class; the drop-in's default train_precision)")
        head = ref_leg or train
        c4 = (config4 or {}).get("bf16") or {}
        # Key order = what survives a consumer that keeps only the END of the line: the contract's scalars first (parsed by name), the bulky per-leg detail in
        # the middle, and roofline / attention / cpu_baseline / summary -- the objects a review reads -- LAST.
        out = {
            "metric": "sampled frames/sec (guided DDPM sampling, 11x96x96 video)", "value": round(frames_per_s, 4), "unit": "frames/s",
            "train_ms_per_step": head["ms_per_step"] if train else None, "train_variant": tr_var if train else None,
            "train_denoising_steps_per_sec": head["denoising_steps_per_sec"] if train else None,
            "train_fp32class_ms_per_step": train["ms_per_step"] if train else None,
            "cfg4_bf16_forward_ms": c4.get("denoiser_forward_ms"),
            "n_gpus": world, "rccl_ranks": world if (world > 1 and backend == "nccl") else (1 if world == 1 else 0), "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "bf16x3",
            "arithmetic": "split-bf16 MFMA: fp32 operands as bf16 hi + lo, products hi*hi + hi*lo + lo*hi accumulated in fp32 (1.5e-5 relative to the "
                          "reference, inside north_star's 1e-3); activations and weights fp32 in HBM; `fp32_exact` = the exact-fp32 sampler",
            "data": "synthetic",
            "config": {"workload": "configs[1]: Lagrangian model.yaml Unet3D (dim 64, 39.5M params, random init), 3x11x96x96, batch 4 per GPU, "
                                   "guidance w=5, 256-step ancestral DDPM with dynamic thresholding; step = one guided p_sample over the batch "
                                   "(denoiser at batch 8 + x0/quantile/posterior); 256 steps = one sample()",
                       "batch_per_gpu": B_PER_GPU, "frames": T, "image": HW, "timesteps": TIMESTEPS, "guidance_scale": W_GUIDE,
                       "hipgraph": stepper.graph is not None, "launches_per_step": len(pl.steps),
                       "parallelism": f"independent sampling shards x{world} (no data-path collective)" + ("" if backend == "nccl" or world == 1
                                                                                                           else f" [{backend} rehearsal, not RCCL]"),
                       "train_ms_per_step": head["ms_per_step"] if train else None, "train_variant": tr_var if train else None,
                       "train_fp32class_ms_per_step": train["ms_per_step"] if train else None, "cfg4_bf16_forward_ms": c4.get("denoiser_forward_ms")},
            "denoising_sample_steps_per_sec": round(world * B_PER_GPU / (ms_per_step * 1e-3), 3),
            "full_sample": full_sample, "fp32_exact": fp32_exact, "bf16_throughput_mode": bf16_mode,
            "denoiser_ms_by_kernel_family": families, "denoiser_event_ms": round(fwd_ms, 3), "output_finite": finite,
            "training": train, "config4": config4, "guidance_sweep": guidance_sweep, "cfg1_gpu": cfg1_gpu,
            "attention": attention, "roofline": roofline, "cpu_baseline": cpu,
        }
        tv = lambda key: ({k: (train.get(key) or {}).get(k) for k in ("ms_per_step", "denoising_steps_per_sec", "launches_per_step", "arena_GB")}  # noqa: E731
                          if train and train.get(key) else None)
        rt = (train or {}).get("roofline_training") or {}
        out["summary"] = {
            "sampling": {"ms_per_guided_step": round(ms_per_step, 3), "frames_per_sec": round(frames_per_s, 4), "dtype": "bf16x3 (split-bf16, fp32-class)",
                         "fp32_exact_ms": (fp32_exact or {}).get("ms_per_step"), "bf16_mode_ms": (bf16_mode or {}).get("ms_per_step"),
                         "dominant_kernel": dom, "roofline_frac": roofline["frac"], "launches_per_step": len(pl.steps)},
            "training": {"headline_variant": tr_var, "ms_per_step": head["ms_per_step"], "denoising_steps_per_sec": head["denoising_steps_per_sec"],
                         "reference_precision_variant (fp16 operands, loss scaling: main.py:34)": tv("reference_precision_variant"),
                         "split_bf16_fp32class_variant (Unet3D default)": {"ms_per_step": train["ms_per_step"], "denoising_steps_per_sec": train["denoising_steps_per_sec"],
                                                                             "launches_per_step": train["launches_per_step"], "event_ms_forward": rt.get("event_ms_forward"),
                                                                             "event_ms_backward": rt.get("event_ms_backward")},
                         "fp32_parity_variant": tv("fp32_parity_variant"),
                         "bf16_single_pass_variant": tv("reduced_precision_variant")} if train else None,
            "config4": {k: {"denoiser_forward_ms": v.get("denoiser_forward_ms"), "guided_step_ms": v.get("guided_step_ms")} for k, v in (config4 or {}).items()
                        if isinstance(v, dict)} or config4,
            "attention_ms": {k: v["ms_per_step"] for k, v in attention.items()},
            "attention_mfma_pipe_util": {k: v["mfma_pipe_util"] for k, v in attention.items() if v.get("mfma_pipe_util") is not None},
            "counter_profiles": {"traffic": profile_id("traffic"), "sq": profile_id("sq")},
            "cpu_baseline": {k: cpu.get(k) for k in ("value", "guided_step_s", "train_step_s", "cfg1_train_step_s", "cores", "cpu_model")} if cpu else None,
            "cfg1_gpu_train_ms": (cfg1_gpu or {}).get("ms_per_step"),
        }
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + "\n").encode())
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
