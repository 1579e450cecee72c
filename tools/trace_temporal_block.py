"""VMM_TB_TRACE time line of one fused temporal block launch (workgroup 0): python tools/trace_temporal_block.py [T HW ntok]
(needs a library built with the stamps: python tools/build_ab.py temporal_block -DVMM_TB_TRACE_BUILD=1, then VMM_LIB_PATH=.../libvmm_hip_ab.so)"""
import os
import sys

os.environ["VMM_TB_TRACE"] = "1"
os.environ.setdefault("VMM_TB_VERSION", "2")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import bench_temporal_block as b  # noqa: E402

if __name__ == "__main__":
    import ctypes as C
    import torch
    from videometamaterials_amd import _native as N, hostmath
    import test_gpu_kernels as tk
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
    os.environ.pop("VMM_TB_TRACE")
    for i in range(3):
        if i == 2:
            os.environ["VMM_TB_TRACE"] = "1"
        N.check(lib.vmm_temporal_block_bf16x3(x.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), ek.data_ptr() if ntok else None,
                                              ev.data_ptr() if ntok else None, ntok, bias.data_ptr(), 1 if ntok == T else 0, rot.data_ptr(), out.data_ptr(),
                                              Cc, B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), s), "tb")
        torch.cuda.synchronize()
