#!/usr/bin/env python
"""Benchmark of the hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W        (N > 1: launched by torch.distributed.run, one rank per GPU)

Workload (BASELINE.json configs[1]): Lagrangian model.yaml denoiser (dim 64, mults 1-2-4-8, 39.5 M parameters,
random-init), 3 x 11 x 96 x 96 video, batch 4 per GPU, classifier-free guidance w = 5, dynamic thresholding.
One "step" = one guided ancestral DDPM step p_sample(x_t, t) for the whole batch: the denoiser at batch 2B
(conditional + unconditional branch), x0 prediction, exact 0.9-quantile, posterior update -- exactly what the
reference executes 256 times per sample() call (vddp.py:956-975).  Inputs are resident in HBM.
value = sampled frames/s = n_gpus * B * 11 frames / (256 steps * step time).

Extra objects on the JSON line: `roofline` for the dominant kernel family (the 3x3 convolutions, split-bf16 MFMA bound) from
HIP-event timing of every launch, `cpu_baseline` = the oracle (CPU restatement) timed on the host cores, and `training` = the
data-parallel optimisation step (the other half of BASELINE.json's metric).
"""
from __future__ import annotations

import argparse
import json
import os
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
KERNEL_OF = {"vmm_conv3x3_bf16x3": "conv3x3_x3_kernel", "vmm_conv_igemm_bf16x3": "igemm_bf16x3_kernel", "vmm_conv_igemm_f32": "igemm_f32_kernel",
             "vmm_temporal_block_bf16x3": "temporal_block_kernel", "vmm_linattn_block_bf16x3": "linattn_"}


def pmc_traffic(family: str):
    """HBM bytes per launch of a kernel family from the committed rocprofv3 PMC passes (profiles/traffic_latest.json, made by
    tools/profile_bench.sh on this same bench command: separate FETCH_SIZE / WRITE_SIZE passes; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950's 128-byte requests).  None when the file or the kernel is missing."""
    path = os.path.join(ROOT, "profiles", "traffic_latest.json")
    prefix = KERNEL_OF.get(family)
    if not prefix or not os.path.exists(path):
        return None
    tab = json.load(open(path))
    tot, n = 0.0, 0
    for k, v in tab.items():
        if k.startswith(prefix) and "FETCH_SIZE_KiB_per_launch" in v and "WRITE_SIZE_KiB_per_launch" in v:
            ln = v.get("launches_FETCH_SIZE", 1)
            tot += ln * (2.0 * v["FETCH_SIZE_KiB_per_launch"] + v["WRITE_SIZE_KiB_per_launch"]) * 1024.0
            n += ln
    return round(tot / n) if n else None



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


def cpu_baseline(max_seconds: float = 30.0):
    """Oracle (oracle/unet3d_oracle.py, parity-pinned CPU restatement of the reference) on the host cores."""
    from oracle import unet3d_oracle as uo
    torch.manual_seed(0)
    import videometamaterials_amd as vm
    model = vm.Unet3D(**LAGRANGIAN)
    sd = {k: v.detach().clone() for k, v in model.state_dict().items()}
    cfg = uo.UnetCfg(**{k: v for k, v in LAGRANGIAN.items()})
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 3, T, HW, HW, generator=g)
    t = torch.randint(0, TIMESTEPS, (1,), generator=g)
    cond = torch.rand(1, 11, generator=g) * 2 - 1
    nthreads = usable_cores()
    torch.set_num_threads(nthreads)
    times = []
    with torch.no_grad():
        t_begin = time.perf_counter()
        for i in range(4):
            t0 = time.perf_counter()
            uo.unet3d_forward(sd, cfg, x, t, cond, torch.zeros(1, dtype=torch.bool))
            times.append(time.perf_counter() - t0)
            if time.perf_counter() - t_begin > max_seconds:
                break
    fwd = min(times[1:]) if len(times) > 1 else times[0]
    step_s = fwd * 2 * B_PER_GPU  # guided step at batch 4 = 8 single-sample forwards
    return {"value": B_PER_GPU * T / (TIMESTEPS * step_s), "unit": "frames/s", "cores": nthreads, "kind": "port",
            "sample": f"{len(times)} oracle Unet3D forwards (B=1, 11x96x96, Lagrangian widths, fp32, torch CPU {nthreads} threads); "
                      f"best {fwd:.2f} s/forward, extrapolated x8 forwards per guided step (B=4) x 256 steps",
            "forward_s_b1": fwd}


def bench_training(vm, model, diff, dev, dist, world, rank, steps: int, precision: str = "fp32"):
    """One data-parallel optimisation step = q_sample -> denoiser forward -> L1 loss -> hand-written backward -> bucketed RCCL
    all-reduce overlapped with the backward -> multi-tensor Adam (+ EMA every 10 steps); per-GPU batch 4 (model.yaml:2).
    precision "fp32": exact-fp32 MFMA everywhere (the parity mode, gradients within 1e-3 of the reference);
    "bf16x3": forward + data gradients on the split-bf16 matrix cores, weight gradients fp32."""
    from videometamaterials_amd.dp import DataParallelTrainer
    model.static_weights = False
    model.train_precision = precision
    model.train()
    tr = DataParallelTrainer(diff, train_lr=1e-4)
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
    fl = sum(f for _, f, _ in pl.meta) + sum(f for _, f, _ in pl.bwd_meta)
    return {"optimizer_steps_per_sec": round(1e3 / ms, 4), "denoising_steps_per_sec": round(world * B_PER_GPU * 1e3 / ms, 3), "ms_per_step": round(ms, 2),
            "batch_per_gpu": B_PER_GPU, "dtype": "f32", "matrix_core_mode": precision, "loss": float(loss), "gemm_TFLOP_per_step": round(fl / 1e12, 3),
            "achieved_gemm_TFLOPs": round(fl / (ms * 1e-3) / 1e12, 1), "grad_allreduce_MB": round(pl.pgrad_floats * 4 / 1e6, 1)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=8)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--detail", action="store_true", help="per-launch timing table on stderr")
    ap.add_argument("--no-train", action="store_true", help="skip the training-step measurement")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 or world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        # one process per GPU over RCCL.  (VMM_DIST_BACKEND=gloo lets a single-GPU box rehearse the multi-rank control flow with
        # several ranks sharing cuda:0 -- RCCL itself refuses duplicate devices; numbers from such a run are meaningless.)
        dev_index = local_rank % torch.cuda.device_count()
        torch.cuda.set_device(dev_index)
        dist.init_process_group(os.environ.get("VMM_DIST_BACKEND", "nccl"), rank=rank, world_size=world)
    else:
        dist = None
        dev_index = 0
        torch.cuda.set_device(0)
    dev = torch.device("cuda", dev_index)

    import videometamaterials_amd as vm
    torch.manual_seed(0)  # identical random-init weights on every rank
    model = vm.Unet3D(**LAGRANGIAN).to(dev).eval()
    diff = vm.GaussianDiffusion(model, image_size=HW, num_frames=T, channels=3, timesteps=TIMESTEPS, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=TIMESTEPS).to(dev)
    g = torch.Generator().manual_seed(2 + rank)
    # rows of data/target_responses.csv are min-max normalised to [-1, 1] by the reference; synthetic stand-in of that range
    cond = (torch.rand(B_PER_GPU, 11, generator=g) * 2 - 1).to(dev)
    shape = (B_PER_GPU, 3, T, HW, HW)
    img = torch.randn(shape, generator=g).to(dev)

    model.static_weights = True
    from videometamaterials_amd.diffusion import _GraphedStep
    stepper = _GraphedStep(diff, shape, 11, W_GUIDE)
    stepper.set_cond(cond)
    if args.no_graph:
        stepper.captured = True  # stay eager
    ts = list(reversed(range(TIMESTEPS)))

    def run_steps(n, offset=0):
        nonlocal img
        for j in range(n):
            img = stepper(img, ts[(offset + j) % TIMESTEPS])

    run_steps(args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run_steps(args.steps, args.warmup)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        et = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(et, op=dist.ReduceOp.MAX)
        elapsed = float(et.item())
    ms_per_step = elapsed / args.steps * 1e3
    frames_per_s = world * B_PER_GPU * T / (TIMESTEPS * ms_per_step * 1e-3)
    finite = bool(torch.isfinite(img).all().item())

    # ---- second half of BASELINE.json's metric: training denoising steps/s (configs[2]: per-GPU batch 4, fp32, Adam, RCCL all-reduce)
    train = None
    if not args.no_train:
        try:
            train = bench_training(vm, model, diff, dev, dist, world, rank, steps=max(2, min(args.steps, 6)))
            # same step with the forward and the data gradients on the split-bf16 matrix cores (extra information, not the parity mode)
            train["split_bf16_variant"] = bench_training(vm, model, diff, dev, dist, world, rank, steps=max(2, min(args.steps, 6)), precision="bf16x3")
        except Exception as e:  # the sampling metric above stays valid; report the failure instead of hiding it
            train = {"error": f"{type(e).__name__}: {e}"}

    out = None
    if rank == 0:
        # ---- per-kernel-family timing with HIP events on the launch stream (separate, un-timed pass)
        pl = stepper.plan
        fam_ms, fam_fl, fam_by, fam_n = {}, {}, {}, {}
        reps = 3
        for _ in range(reps):
            for (name, fl, by), ms in zip(pl.meta, pl.launch_timed()):
                fam_ms[name] = fam_ms.get(name, 0.0) + ms
                fam_fl[name] = fam_fl.get(name, 0.0) + fl
                fam_by[name] = fam_by.get(name, 0.0) + by
                fam_n[name] = fam_n.get(name, 0) + 1
        if args.detail:
            ms = pl.launch_timed()
            rows_ = sorted(zip(ms, pl.steps, pl.meta), key=lambda r: -r[0])
            for t_ms, (_, _, what), (fam, fl, by) in rows_:
                print(f"  {t_ms:8.3f} ms  {fl / max(t_ms, 1e-9) / 1e9:8.1f} TFLOP/s  {by / max(t_ms, 1e-9) / 1e6:8.1f} GB/s  {what}", file=sys.stderr)
        fwd_ms = sum(fam_ms.values()) / reps
        dom = max(fam_ms, key=fam_ms.get)
        ach_tflops = fam_fl[dom] / (fam_ms[dom] * 1e-3) / 1e12
        ach_gbs = fam_by[dom] / (fam_ms[dom] * 1e-3) / 1e9
        # fp32 MFMA peak for the exact kernel; for the split-bf16 kernel every algorithmic flop costs three bf16 MFMA flops,
        # so its roof for ALGORITHMIC flops is the dense bf16 peak / 3
        peak = PEAK_BF16_MFMA_TFLOPS / 3.0 if dom.endswith("bf16x3") else PEAK_FP32_MFMA_TFLOPS
        roofline = {"kernel": dom, "bound": "mfma", "achieved": round(ach_tflops, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                    "frac": round(ach_tflops / peak, 4), "traffic": pmc_traffic(dom),
                    "traffic_note": "HBM bytes per launch, mean over the family, from profiles/traffic_latest.json (rocprofv3 PMC passes of this command)",
                    "peak_note": "dense bf16 MFMA 2500 TFLOP/s / 3 passes (split-bf16 operands, fp32-class result)" if dom.endswith("bf16x3")
                    else "fp32 MFMA (v_mfma_f32_32x32x2_f32)",
                    "launches_per_step": fam_n[dom] // reps, "avg_launch_ms": round(fam_ms[dom] / fam_n[dom], 4),
                    "algorithmic_GFLOP_per_step": round(fam_fl[dom] / reps / 1e9, 1), "algorithmic_GB_per_step": round(fam_by[dom] / reps / 1e9, 2),
                    "hbm_frac_of_8TBs": round(ach_gbs / PEAK_HBM_GBS, 4),
                    "share_of_denoiser_time": round(fam_ms[dom] / sum(fam_ms.values()), 3)}
        families = {k: round(v / reps, 3) for k, v in sorted(fam_ms.items(), key=lambda kv: -kv[1])}
        cpu = None
        if not args.no_cpu_baseline:
            cpu = cpu_baseline()
        out = {
            "metric": "sampled frames/sec (guided DDPM sampling, 11x96x96 video)", "value": round(frames_per_s, 4), "unit": "frames/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 3), "higher_is_better": True,
            "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "configs[1]: Lagrangian model.yaml Unet3D (dim 64, 39.5M params, random init), 3x11x96x96, batch 4 per GPU, "
                                   "guidance w=5, 256-step ancestral DDPM with dynamic thresholding; step = one guided p_sample over the batch "
                                   "(denoiser at batch 8 + x0/quantile/posterior)",
                       "batch_per_gpu": B_PER_GPU, "frames": T, "image": HW, "timesteps": TIMESTEPS, "guidance_scale": W_GUIDE,
                       "hipgraph": stepper.graph is not None, "parallelism": f"independent sampling shards x{world} (no data-path collective)"},
            "denoising_sample_steps_per_sec": round(world * B_PER_GPU / (ms_per_step * 1e-3), 3),
            "full_sample_seconds": round(TIMESTEPS * ms_per_step * 1e-3, 2),
            "denoiser_ms_by_kernel_family": families, "denoiser_event_ms": round(fwd_ms, 3), "output_finite": finite,
            "training": train, "roofline": roofline, "cpu_baseline": cpu,
        }
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
