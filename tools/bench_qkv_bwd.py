"""Stand-alone timing of the fused to_qkv backward (qkv_bwd.hip) on the Lagrangian 96 x 96 shape (batch 4, 11 frames: 405 504 rows x 768):

    python tools/bench_qkv_bwd.py [rows] [reps]

(A/B of two builds on one box: VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so, see tools/build_ab.py; -DVMM_QB_SKIP=<bits> knock-outs;
QKV_VARIANT=fp16 | bf16: the single-pass instances, g as rows of 16-bit operands.)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import _native as N  # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4 * 11 * 96 * 96
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
dev = torch.device("cuda", 0)
lib = N.lib()
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
g_ = torch.Generator().manual_seed(1)
Cc, Nq = 64, 768
x = (torch.randn(rows, Cc, generator=g_) * 1.5 + 0.3).to(dev)
gamma = (1 + 0.2 * torch.randn(Cc, generator=g_)).to(dev)
w = (torch.randn(Nq, Cc, generator=g_) / 8).to(dev)
g = torch.randn(rows, Nq, generator=g_).to(dev)
VAR = os.environ.get("QKV_VARIANT", "bf16x3")
gbytes = 4
if VAR != "bf16x3":
    g = g.to(torch.float16 if VAR == "fp16" else torch.bfloat16)
    gbytes = 2
mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
stats = torch.cat([mean, 1 / (var + 1e-5).sqrt()], 1).contiguous()
# fmt-2 fragments of the (K = 768, N = 64) operand
wt = w.t().contiguous()  # "weight (out = 64, in = 768)"
packed = torch.zeros(64 * 768, device=dev)
job = (N.PackJob * 1)()
j = job[0]
j.torch_w, j.packed = wt.data_ptr(), packed.data_ptr()
j.TH, j.TW, j.C, j.Cp, j.N, j.sn, j.sc = 1, 1, 768, 768, 64, 768, 1
j.fmt = 2 | (16 if VAR == "fp16" else 0)
tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, packed.numel(), 0, s), "pack")
ws = torch.empty(int(lib.vmm_qkv_bwd_workspace(rows, Cc, Nq)), device=dev)
gy = torch.empty(rows, Cc, device=dev)
dw = torch.zeros(Cc, Nq, device=dev)


dx = torch.randn(rows, Cc, device=dev)
dgam = torch.zeros(Cc, device=dev)
LN = os.environ.get("QKV_LN") == "1"  # the variant with the LayerNorm backward as its epilogue


def run():
    if LN:
        N.check(getattr(lib, "vmm_qkv_bwd_ln_" + VAR)(x.data_ptr(), Cc, stats.data_ptr(), gamma.data_ptr(), g.data_ptr(), Nq, packed.data_ptr(), dx.data_ptr(), Cc, 1,
                                          dgam.data_ptr(), dw.data_ptr(), ws.data_ptr(), rows, Cc, Nq, s), "qkv_bwd_ln")
        return
    N.check(getattr(lib, "vmm_qkv_bwd_" + VAR)(x.data_ptr(), Cc, stats.data_ptr(), gamma.data_ptr(), g.data_ptr(), Nq, packed.data_ptr(), gy.data_ptr(), Cc,
                                   dw.data_ptr(), ws.data_ptr(), rows, Cc, Nq, s), "qkv_bwd")


for _ in range(3):
    run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(reps):
    run()
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
want = g[:4096].double() @ w.double()
err = float((gy[:4096].double() - want).abs().max() / want.abs().max())
print(f"qkv_bwd {VAR}{' +ln' if LN else ''} rows {rows}: {ms:.3f} ms  ({rows * Nq * gbytes / ms / 1e6:.0f} GB/s of g, {4.0 * rows * Nq * Cc / ms / 1e9:.0f} TFLOP/s)  gy err {err:.1e}  lib {os.environ.get('VMM_LIB_PATH', 'default')}")
