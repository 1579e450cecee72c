"""Kernel-level proof that the 16-bit hand-over of the qkv-row gradient (csrc/vmm_common.h VMM_DQKV16) changes no bit: the product library and a library
whose single-pass objects were built with -DVMM_DQKV16=0 (fp32 rows) in ONE process, on the same inputs:

    python tools/build_ab.py temporal_block_bwd linattn_block_bwd qkv_bwd -DVMM_DQKV16=0
    python tools/check_dqkv16_kernels.py [fp16|bf16]

 1. vmm_temporal_block_bwd_<p>: the 16-bit rows == the fp32 rows of the other library rounded to the operand type (round-to-nearest-even), every element;
    the other outputs (dW_out, dbias, d ek, d ev, LayerNorm statistics) identical.
 2. vmm_qkv_bwd_ln_<p> fed with each library's own rows: dx, dgamma and dW identical bit for bit (the kernel's sums run in a fixed order).
Whole-step gradients cannot show this: the training backward contains order-dependent fp32 atomics by design (tests/test_gpu_concurrency.py)."""
import ctypes as C
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from videometamaterials_amd import _native as N, hostmath  # noqa: E402
import test_gpu_kernels as tk  # noqa: E402
from test_gpu_block_bwd import _pack_frag_t  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
qdtype = torch.float16 if prec == "fp16" else torch.bfloat16
half = 16 if prec == "fp16" else 0
lib_a = N.lib()
lib_b = C.CDLL(os.path.join(ROOT, "videometamaterials_amd", "libvmm_hip_ab.so"))
for name, argtypes in N.SIGNATURES.items():
    fn = getattr(lib_b, name)
    fn.argtypes, fn.restype = argtypes, C.c_int
gpu = torch.device("cuda:0")
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
B, T, HW, ntok, Cc, heads, hid = 2, 11, 48 * 48, 11, 64, 8, 256
rows = B * T * HW
g = torch.Generator().manual_seed(3)
x = (torch.randn(rows, Cc, generator=g) * 1.5 + 0.3).to(gpu)
gamma = (1 + 0.2 * torch.randn(Cc, generator=g)).to(gpu)
wqkv, wout = torch.randn(3 * hid, Cc, generator=g) / 8, torch.randn(Cc, hid, generator=g) / 16
bias, rot = torch.randn(heads, T, T, generator=g).to(gpu), hostmath.rotary_table(T, 32).to(gpu)
ek, ev = torch.randn(B, ntok, hid, generator=g).to(gpu), torch.randn(B, ntok, hid, generator=g).to(gpu)
dout = (torch.randn(rows, Cc, generator=g) * 64).to(gpu)  # (a loss-scaled gradient)
wq, woT = tk._pack_frag(N, lib_a, gpu, wqkv, 2 | half), _pack_frag_t(N, lib_a, gpu, wout, 2 | half)
wd = tk._pack_frag(N, lib_a, gpu, wqkv.t().contiguous(), 2 | half)  # the (K = 768, N = 64) operand of the to_qkv backward


def run(lib, dq_dtype):
    dqkv = torch.zeros(rows, 3 * hid, device=gpu, dtype=dq_dtype)
    stats, dwo, dbias = torch.zeros(rows, 2, device=gpu), torch.zeros(hid, Cc, device=gpu), torch.zeros(heads, T, T, device=gpu)
    dek, dev_ = torch.zeros(B, ntok, hid, device=gpu), torch.zeros(B, ntok, hid, device=gpu)
    ws = torch.empty(lib_a.vmm_temporal_block_bwd_workspace(B, T, HW, Cc, heads, ntok), device=gpu)
    d = N.AttnBlockBwd()
    d.x, d.ldx, d.gamma, d.wqkv_frag, d.wout_t_frag = x.data_ptr(), Cc, gamma.data_ptr(), wq.data_ptr(), woT.data_ptr()
    d.ek, d.ev, d.ntok = ek.data_ptr(), ev.data_ptr(), ntok
    d.bias, d.bias_on_cond, d.rot_tab = bias.data_ptr(), 1, rot.data_ptr()
    d.dout, d.lddo, d.dqkv, d.lddqkv, d.ln_stats = dout.data_ptr(), Cc, dqkv.data_ptr(), 3 * hid, stats.data_ptr()
    d.dwout_packed, d.dbias, d.dek, d.dev, d.workspace = dwo.data_ptr(), dbias.data_ptr(), dek.data_ptr(), dev_.data_ptr(), ws.data_ptr()
    d.B, d.T, d.HW, d.C, d.heads, d.q_scale, d.eps = B, T, HW, Cc, heads, 32 ** -0.5, 1e-5
    N.check(getattr(lib, "vmm_temporal_block_bwd_" + prec)(C.byref(d), s), "block bwd")
    dx, dgam, dw = torch.zeros(rows, Cc, device=gpu), torch.zeros(Cc, device=gpu), torch.zeros(Cc, 3 * hid, device=gpu)
    ws2 = torch.empty(int(lib_a.vmm_qkv_bwd_workspace(rows, Cc, 3 * hid)), device=gpu)
    N.check(getattr(lib, "vmm_qkv_bwd_ln_" + prec)(x.data_ptr(), Cc, stats.data_ptr(), gamma.data_ptr(), dqkv.data_ptr(), 3 * hid, wd.data_ptr(), dx.data_ptr(), Cc, 0,
                                                    dgam.data_ptr(), dw.data_ptr(), ws2.data_ptr(), rows, Cc, 3 * hid, s), "qkv_bwd_ln")
    torch.cuda.synchronize()
    return dict(dqkv=dqkv, stats=stats, dwo=dwo, dbias=dbias, dek=dek, dev=dev_, dx=dx, dgamma=dgam, dw=dw)


a, b = run(lib_a, qdtype), run(lib_b, torch.float32)
ok = True
same = torch.equal(a["dqkv"], b["dqkv"].to(qdtype))
print(prec, "16-bit rows == fp32 rows of the -DVMM_DQKV16=0 library rounded to the operand type:", same, "| nonzero elements", int((a["dqkv"] != 0).sum()), "of", a["dqkv"].numel())
ok &= same
for k in ("stats", "dwo", "dbias", "dek", "dev", "dx", "dgamma", "dw"):
    same = torch.equal(a[k], b[k])
    print(f"{prec} {k:7s} identical: {same}  (norm {float(a[k].double().norm()):.6e})")
    ok &= same
sys.exit(0 if ok else 1)
