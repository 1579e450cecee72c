"""Measurement aid: the Winograd 3x3 kernel (vmm_conv3x3_wino_bf16x3) against the direct one (vmm_conv3x3_bf16x3) on the denoiser's layer shapes --
time per launch (hipEvents over REP launches), TFLOP/s on the direct-convolution flop count, error against torch's fp64 conv2d.
    python tools/bench_wino.py [nimg] [rep]"""
import ctypes as C, math, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import _native as N

NIMG = int(sys.argv[1]) if len(sys.argv) > 1 else 22
REP = int(sys.argv[2]) if len(sys.argv) > 2 else 20
lib = N.lib()
dev = torch.device("cuda")
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
rows_of = lambda t: t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()


def pack(w, fmt, nfloats):
    Cout, Cin = w.shape[:2]
    packed = torch.zeros(nfloats, device=dev)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = w.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, fmt
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, nfloats, 0, s()), "pack")
    return packed


def run(H, C1, C2, Cout, fused, gn=True):
    g = torch.Generator(device="cuda").manual_seed(5)
    Cin = C1 + C2
    x1 = torch.randn(NIMG, C1, H, H, generator=g, device=dev)
    x2 = torch.randn(NIMG, C2, H, H, generator=g, device=dev) if C2 else None
    w = (torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) / math.sqrt(Cin * 9)).contiguous()
    b = torch.randn(Cout, generator=g, device=dev)
    xa = x1
    coef = None
    if fused:
        coef = torch.randn(1, C1, 2, generator=g, device=dev)
        xa = F.silu(x1 * coef[:, :, 0][:, :, None, None] + coef[:, :, 1][:, :, None, None])
    xin = torch.cat([xa, x2], 1) if C2 else xa
    ref = F.conv2d(xin.double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    x1r = rows_of(x1)
    x2r = rows_of(x2) if C2 else None
    out = torch.zeros(NIMG * H * H, Cout, device=dev)
    part = torch.zeros(1 << 22, device=dev)
    tickets = torch.zeros(4096, dtype=torch.int32, device=dev)
    res = {}
    for name, kernel, fmt, nfl, query in (("direct", lib.vmm_conv3x3_bf16x3, 2, (Cout + 31) // 32 * 32 * 9 * Cin, lib.vmm_conv3x3_fuses_gn),
                                          ("wino", lib.vmm_conv3x3_wino_bf16x3, 8, 16 * Cin * Cout, lib.vmm_conv3x3_wino_fuses_gn)):
        packed = pack(w, fmt, nfl)
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), b.data_ptr(), out.data_ptr(), Cout
        if C2:
            d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = NIMG, H, H, H, H, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, H, 1, Cout, 32, 1.0
        d.a_imgs_per_sample = NIMG
        if name == "direct" and os.environ.get("WINO_TICKETS"):
            d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
        if fused:
            d.a_mode, d.a_coef = 1, coef.data_ptr()
        if gn:
            d.gn_part, d.gn_groups = part.data_ptr(), 8
            if not query(C.byref(d)):
                d.gn_part, d.gn_groups = None, 0
        out.zero_()
        rc = kernel(C.byref(d), s())
        if rc != 0:
            res[name] = None
            continue
        torch.cuda.synchronize()
        rel = float((out.double() - ref).norm() / ref.norm())
        for _ in range(3):
            kernel(C.byref(d), s())
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(REP):
            kernel(C.byref(d), s())
        e1.record()
        torch.cuda.synchronize()
        res[name] = (e0.elapsed_time(e1) / REP * 1e3, rel)
    fl = 2 * 9 * Cin * Cout * NIMG * H * H
    byt = 4 * NIMG * H * H * (Cin + Cout)
    line = f"{H:3d}x{H:<3d} {C1:4d}+{C2:<4d}->{Cout:4d} fused={int(fused)}  "
    for name in ("direct", "wino"):
        r = res[name]
        line += f"{name}: " + (f"{r[0]:7.1f} us {fl / r[0] / 1e6:6.1f} TF/s {byt / r[0] / 1e6:5.2f} TB/s err {r[1]:.1e}   " if r else "   --   ")
    if res["direct"] and res["wino"]:
        line += f"x{res['direct'][0] / res['wino'][0]:.2f}"
    print(line, flush=True)


if __name__ == "__main__" and os.environ.get("WINO_SHAPES"):
    for spec in os.environ["WINO_SHAPES"].split(";"):
        H, C1, C2, Cout, fused = [int(v) for v in spec.split(",")]
        run(H, C1, C2, Cout, bool(fused))
elif __name__ == "__main__":
    for H, C1, C2, Cout in ((96, 64, 0, 64), (96, 64, 64, 64), (48, 64, 0, 128), (48, 128, 0, 128), (48, 128, 128, 128), (24, 128, 0, 256), (24, 256, 0, 256),
                            (24, 256, 256, 256), (12, 256, 0, 512), (12, 512, 0, 512), (12, 512, 512, 512)):
        for fused in (False, True):
            if fused and C2:
                continue
            run(H, C1, C2, Cout, fused)
