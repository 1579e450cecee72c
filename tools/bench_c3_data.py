"""Measurement aid: is the 3 x 3 kernel bound by the chip's clock / power management rather than by its own instruction stream?  The same launch on random
operands, on all-zero operands (MI355X_MICROARCH.md "DVFS give-back": zero-filled inputs run the same instruction stream at a higher sustained clock) and
on operands whose LOW split planes are zero (bf16-representable values: the lo * hi / hi * lo passes multiply zeros), back to back on one box.
    python tools/bench_c3_data.py [nimg = 88] [rep = 50]
Prints us per launch and TFLOP/s (algorithmic) per shape and data kind."""
import ctypes as C, math, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import _native as N

NIMG = int(sys.argv[1]) if len(sys.argv) > 1 else 88
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 50
lib = N.lib()
dev = torch.device("cuda")
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)


def pack(w, nfloats):
    Cout, Cin = w.shape[:2]
    packed = torch.zeros(nfloats, device=dev)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = w.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 2
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, nfloats, 0, s()), "pack")
    return packed


def run(H, Cin, Cout):
    g = torch.Generator(device="cuda").manual_seed(5)
    line = f"{H:3d}x{H:<3d} {Cin:4d}->{Cout:4d}  "
    for kind in ("random", "bf16-exact", "zero", "random"):
        x = torch.randn(NIMG * H * H, Cin, generator=g, device=dev)
        w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) / math.sqrt(Cin * 9)).contiguous()
        if kind == "bf16-exact":
            x, w = x.bfloat16().float(), w.bfloat16().float()
        if kind == "zero":
            x.zero_()
            w.zero_()
        packed = pack(w, (Cout + 31) // 32 * 32 * 9 * Cin)
        out = torch.zeros(NIMG * H * H, Cout, device=dev)
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1, d.w, d.out, d.ldo = x.data_ptr(), Cin, Cin, packed.data_ptr(), out.data_ptr(), Cout
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = NIMG, H, H, H, H, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, H, 1, Cout, 32, 1.0
        d.a_imgs_per_sample = 11
        for _ in range(5):
            N.check(lib.vmm_conv3x3_bf16x3(C.byref(d), s()), "conv")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            lib.vmm_conv3x3_bf16x3(C.byref(d), s())
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / REP * 1e3
        line += f"{kind}: {us:6.1f} us {2 * 9 * Cin * Cout * NIMG * H * H / us / 1e6:6.1f} TF/s   "
    print(line, flush=True)


if __name__ == "__main__":
    for H, Cin, Cout in ((12, 512, 512), (24, 256, 256), (48, 128, 128), (96, 64, 64)):
        run(H, Cin, Cout)
