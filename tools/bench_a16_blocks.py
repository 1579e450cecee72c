"""fp32-stored vs bf16-stored feature maps through the single-pass fused blocks, same shape, one process:
   python tools/bench_a16_blocks.py [T HW ntok B]      (configs[3]: 22 36864 16 8)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from videometamaterials_amd import _native as N, hostmath  # noqa: E402
import test_gpu_kernels as tk  # noqa: E402

T, HW, ntok, B = (int(v) for v in sys.argv[1:5]) if len(sys.argv) >= 5 else (22, 192 * 192, 16, 8)
Cc, heads, hid = 64, 8, 256
lib = N.lib()
gpu = torch.device("cuda:0")
g = torch.Generator().manual_seed(1)
x = torch.randn(B * T * HW, Cc, generator=g).to(gpu)
x16 = x.to(torch.bfloat16)
wq = tk._pack_frag(N, lib, gpu, torch.randn(3 * hid, Cc, generator=g) / 8, 2)
wo = tk._pack_frag(N, lib, gpu, torch.randn(Cc, hid, generator=g) / 16, 3)
gam, bias, rot = torch.ones(Cc, device=gpu), torch.randn(heads, T, T, generator=g).to(gpu), hostmath.rotary_table(T, 32).to(gpu)
bo = torch.zeros(Cc, device=gpu)
ek, ev = torch.randn(B, ntok, hid, generator=g).to(gpu), torch.randn(B, ntok, hid, generator=g).to(gpu)
out, out16 = torch.empty_like(x), torch.empty_like(x16)
ws = torch.empty(lib.vmm_linattn_block_workspace(B, T, HW), device=gpu)
s = C.c_void_p(torch.cuda.current_stream().cuda_stream)


def timed(fn, n=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n


for name, f32fn, a16fn, args in (
        ("temporal block", lib.vmm_temporal_block_bf16, lib.vmm_temporal_block_bf16_a16,
         lambda xi, oi: (xi.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), ek.data_ptr(), ev.data_ptr(), ntok, bias.data_ptr(), 0, rot.data_ptr(),
                         oi.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), s)),
        ("linear attention block", lib.vmm_linattn_block_bf16, lib.vmm_linattn_block_bf16_a16,
         lambda xi, oi: (xi.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), bo.data_ptr(), ek.data_ptr(), ev.data_ptr(), ntok, ws.data_ptr(),
                         oi.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(1e-5), s))):
    t32 = timed(lambda: N.check(f32fn(*args(x, out)), name))
    t16 = timed(lambda: N.check(a16fn(*args(x16, out16)), name))
    err = float((out16.float() - out).norm() / out.norm())
    print(f"{name}: fp32-stored {t32:.3f} ms, bf16-stored {t16:.3f} ms, rel diff of the outputs {err:.2e}  (B={B} T={T} HW={HW} ntok={ntok})")
