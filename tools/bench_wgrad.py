"""Time the weight-gradient kernels on the Lagrangian training shapes (batch 4, 11 frames): exact-fp32 nine-tap kernel, the generic split-bf16
kernel and the nine-tap split-bf16 kernel.   python tools/bench_wgrad.py [--batch 4]"""
import argparse
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import _native as N  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--reps", type=int, default=20)
    args = ap.parse_args()
    lib = N.lib()
    dev = torch.device("cuda:0")
    nimg = args.batch * 11
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [(96, 64, 0, 64), (96, 64, 64, 64), (48, 128, 0, 128), (48, 64, 0, 128), (48, 128, 128, 128), (24, 256, 0, 256), (24, 256, 256, 256),
              (12, 512, 0, 512), (12, 512, 512, 512)]
    def f32(d, dy, Cout, dw, db, sc, nsplit, ws):
        return lib.vmm_conv3x3_wgrad_f32(C.byref(d), dy.data_ptr(), Cout, dw.data_ptr(), nsplit, db.data_ptr(), sc.data_ptr(), s)

    def x3_atomic(d, dy, Cout, dw, db, sc, nsplit, ws):
        return lib.vmm_conv3x3_wgrad_bf16x3(C.byref(d), dy.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(), None, s)

    def x3(d, dy, Cout, dw, db, sc, nsplit, ws):
        return lib.vmm_conv3x3_wgrad_bf16x3(C.byref(d), dy.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), s)

    kernels = [("f32 3x3", f32), ("x3 atomics", x3_atomic), ("x3 3x3", x3)]
    os.environ.setdefault("VMM_WGRAD3X3", "1")
    tot = {k: 0.0 for k, _ in kernels}
    for HW, C1, C2, Cout in shapes:
        Cin = C1 + C2
        rows = nimg * HW * HW
        x1 = torch.randn(rows, C1, device=dev)
        x2 = torch.randn(rows, C2, device=dev) if C2 else None
        dy = torch.randn(rows, Cout, device=dev)
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1 = x1.data_ptr(), C1, C1
        if C2:
            d.a2, d.C2, d.lda2 = x2.data_ptr(), C2, C2
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, HW, HW, HW, HW, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale, d.Cout = HW, HW, 1, Cout
        flops = 2.0 * rows * 9 * Cin * Cout
        line = f"{HW:3d}x{HW:<3d} {Cin:4d}->{Cout:<4d} {flops / 1e9:7.1f} GF:"
        ref = None
        for name, fn in kernels:
            dw = torch.zeros(9 * Cin, Cout, device=dev)
            db = torch.zeros(Cout, device=dev)
            sc = torch.zeros(4096, Cout, device=dev)
            nsplit = max(1, min(-(-2048 // ((9 * Cin // 64) * (Cout // 64))), max(1, rows // 256)))
            ws = torch.empty(max(1, int(lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), Cout))), device=dev)
            rc = fn(d, dy, Cout, dw, db, sc, nsplit, ws)
            torch.cuda.synchronize()
            if rc != 0:
                line += f"  {name}: rc={rc}"
                continue
            if ref is None:
                ref = dw.clone()
            else:
                line += f" (rel {float((dw - ref).norm() / ref.norm()):.1e})"
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                fn(d, dy, Cout, dw, db, sc, nsplit, ws)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / args.reps
            tot[name] += ms
            line += f"  {name}: {ms * 1e3:7.1f} us {flops / ms / 1e9:6.1f} TF/s"
        print(line, flush=True)
    print("sum over shapes (ms):", {k: round(v, 3) for k, v in tot.items()})


if __name__ == "__main__":
    main()
