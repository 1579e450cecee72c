"""Time vmm_temporal_block_bf16x3 at the full-resolution shapes (batch 8 = guidance-doubled batch 4).
   python tools/bench_temporal_block.py [T HW ntok]      env: VMM_TB_VERSION=1|2, VMM_TB_GROUP=0|1"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from videometamaterials_amd import _native as N, hostmath  # noqa: E402
import test_gpu_kernels as tk  # noqa: E402


def main():
    T, HW, ntok = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (11, 96 * 96, 11)
    B, Cc, heads, hid = 8, 64, 8, 256
    lib = N.lib()
    gpu = torch.device("cuda:0")
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B * T * HW, Cc, generator=g).to(gpu)
    wq = tk._pack_frag(N, lib, gpu, torch.randn(3 * hid, Cc, generator=g) / 8, 2)
    wo = tk._pack_frag(N, lib, gpu, torch.randn(Cc, hid, generator=g) / 16, 3)
    gam, bias, rot = torch.ones(Cc, device=gpu), torch.randn(heads, T, T, generator=g).to(gpu), hostmath.rotary_table(T, 32).to(gpu)
    ek = torch.randn(B, ntok, hid, generator=g).to(gpu) if ntok else None
    ev = torch.randn(B, ntok, hid, generator=g).to(gpu) if ntok else None
    out = torch.empty_like(x)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    flops = 2.0 * B * T * HW * Cc * 3 * hid + 2.0 * B * T * HW * hid * Cc + 4.0 * B * T * HW * heads * 32 * (T + ntok)
    res = {}
    for ver, grp in ((1, 0), (2, 0), (2, 1)):
        os.environ["VMM_TB_VERSION"], os.environ["VMM_TB_GROUP"] = str(ver), str(grp)
        if lib.vmm_temporal_block_supported(T, ntok, HW, Cc, heads) != ver:
            continue

        def run():
            N.check(lib.vmm_temporal_block_bf16x3(x.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), ek.data_ptr() if ntok else None,
                                                  ev.data_ptr() if ntok else None, ntok, bias.data_ptr(), 1 if ntok == T else 0, rot.data_ptr(), out.data_ptr(),
                                                  Cc, B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), s), "tb")
        for _ in range(3):
            run()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            run()
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        res[(ver, grp)] = out.clone()
        print(f"version {ver} group {grp}: {ms:.3f} ms  {flops / ms / 1e9:.1f} TFLOP/s (algorithmic)  T={T} HW={HW} ntok={ntok}")
    if (1, 0) in res and (2, 0) in res:
        print("v2 == v1 bit for bit:", bool(torch.equal(res[(1, 0)], res[(2, 0)])), " max |diff|", float((res[(1, 0)] - res[(2, 0)]).abs().max()))


if __name__ == "__main__":
    main()
