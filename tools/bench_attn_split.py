"""The sampler's fused attention blocks on their two fp32-class operand forms -- split-bf16 (`_bf16x3`) and IEEE-half hi | lo (`_f16x3`, csrc/vmm_common.h VMM_SPLIT_F16):
time at the Lagrangian full-resolution shape (batch 8 = guidance-doubled batch 4), interleaved, and the deviation of each from the same block in fp64 on a small shape.
   python tools/bench_attn_split.py [reps]"""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videometamaterials_amd import _native as N, hostmath  # noqa: E402
import test_gpu_kernels as tk  # noqa: E402

lib = N.lib()
gpu = torch.device("cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
FMT = {"bf16x3": 0, "f16x3": 32}


def temporal(B, T, HW, ntok, reps, check):
    Cc, heads, hid = 64, 8, 256
    g = torch.Generator().manual_seed(1)
    x = (torch.randn(B * T * HW, Cc, generator=g) * 1.5 + 0.3).to(gpu)
    wqkv, wout = torch.randn(3 * hid, Cc, generator=g) / 8, torch.randn(Cc, hid, generator=g) / 16
    gam, bias, rot = (1 + 0.2 * torch.randn(Cc, generator=g)).to(gpu), torch.randn(heads, T, T, generator=g).to(gpu), hostmath.rotary_table(T, 32).to(gpu)
    ek = torch.randn(B, ntok, hid, generator=g).to(gpu) if ntok else None
    ev = torch.randn(B, ntok, hid, generator=g).to(gpu) if ntok else None
    outs, fns = {}, {}
    for v in FMT:
        wq, wo = tk._pack_frag(N, lib, gpu, wqkv, 2 | FMT[v]), tk._pack_frag(N, lib, gpu, wout, 3 | FMT[v])
        out = torch.empty_like(x)
        fn = getattr(lib, "vmm_temporal_block_" + v)

        def run(fn=fn, wq=wq, wo=wo, out=out):
            N.check(fn(x.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), ek.data_ptr() if ntok else None, ev.data_ptr() if ntok else None, ntok,
                       bias.data_ptr(), 1 if ntok == T else 0, rot.data_ptr(), out.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), s), "tb")
        fns[v], outs[v] = run, out
    times = {v: [] for v in FMT}
    for r in range(reps):
        for v in FMT:
            for _ in range(2):
                fns[v]()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                fns[v]()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / 10)
    print(f"temporal block T={T} HW={HW} B={B} ntok={ntok}: " + "  ".join(f"{v} {min(t):.4f} ms (median {sorted(t)[len(t) // 2]:.4f})" for v, t in times.items()))
    if check:  # fp64 reference of the block on the CPU (the kernel test's arithmetic)
        from oracle import unet3d_oracle as uo  # noqa: F401  (only to make sure the oracle package is importable next to the tests' helpers)
        xd = x.double().cpu().reshape(B, T, HW, Cc)
        mean, var = xd.mean(-1, keepdim=True), xd.var(-1, unbiased=False, keepdim=True)
        y = (xd - mean) / (var + 1e-5).sqrt() * gam.double().cpu()
        qkv = (y @ wqkv.double().t()).reshape(B, T, HW, 3, heads, 32)
        rt = rot.double().cpu()
        cos, sin = rt[:, :, 0].repeat_interleave(2, -1)[None, :, None, None], rt[:, :, 1].repeat_interleave(2, -1)[None, :, None, None]

        def rotate(t):
            pr = t.reshape(*t.shape[:-1], 16, 2)
            return t * cos + torch.stack((-pr[..., 1], pr[..., 0]), -1).reshape(t.shape) * sin
        q, k, vv = rotate(qkv[:, :, :, 0] * 32 ** -0.5), rotate(qkv[:, :, :, 1]), qkv[:, :, :, 2]
        q, k, vv = (t.permute(0, 2, 3, 1, 4) for t in (q, k, vv))
        bf = bias.double().cpu()[None, None]
        if ntok:
            ekd, evd = ek.double().cpu().reshape(B, ntok, heads, 32), ev.double().cpu().reshape(B, ntok, heads, 32)
            k = torch.cat([ekd.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), k], dim=-2)
            vv = torch.cat([evd.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), vv], dim=-2)
            bf = torch.cat([bias.double().cpu() if ntok == T else torch.zeros(heads, T, ntok, dtype=torch.float64), bias.double().cpu()], dim=-1)[None, None]
        att = (q @ k.transpose(-1, -2) + bf).softmax(-1) @ vv
        o = att.permute(0, 3, 1, 2, 4).reshape(B * T * HW, hid)
        want = o @ wout.double().t() + x.double().cpu()
        for v in FMT:
            print(f"   {v}: relative deviation from the fp64 block {tk.relerr(outs[v].cpu().double(), want):.2e}")


reps = int(sys.argv[1]) if len(sys.argv) > 1 else 3
temporal(2, 11, 256, 11, 1, True)
temporal(8, 11, 96 * 96, 11, reps, False)
temporal(8, 11, 96 * 96, 0, reps, False)
