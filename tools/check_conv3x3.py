"""Measurement aid: vmm_conv3x3_bf16x3 / _f32 at an arbitrary shape against torch's conv2d on the GPU (fp64 reference).
    python tools/check_conv3x3.py B T H W C1 C2 Cout fused exact"""
import ctypes as C, math, os, sys
import torch, torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from videometamaterials_amd import _native as N
B, T, H, W, C1, C2, Cout, fused, exact = [int(v) for v in sys.argv[1:10]]
lib = N.lib()
dev = torch.device("cuda")
s = lambda: C.c_void_p(torch.cuda.current_stream().cuda_stream)
g = torch.Generator(device="cuda").manual_seed(12)
Cin = C1 + C2
x1 = torch.randn(B, C1, T, H, W, generator=g, device=dev)
x2 = torch.randn(B, C2, T, H, W, generator=g, device=dev) if C2 else None
w = torch.randn(Cout, Cin, 3, 3, generator=g, device=dev) / math.sqrt(Cin * 9)
b = torch.randn(Cout, generator=g, device=dev)
xa = x1
coef = None
if fused:
    coef = torch.randn(B, C1, 2, generator=g, device=dev)
    xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
xin = torch.cat([xa, x2], 1) if C2 else xa
ref = F.conv2d(xin.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W).double(), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
rows_of = lambda t: t.permute(0, 2, 3, 4, 1).reshape(-1, t.shape[1]).contiguous()
K = 9 * Cin
Kpad = (K + 31) // 32 * 32
packed = torch.zeros((Cout + 31) // 32 * 32 * Kpad, device=dev)
job = (N.PackJob * 1)()
j = job[0]
wg = w.contiguous()
j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 4 if exact else 2
tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(dev)
N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Kpad, 0, s()), "pack")
d = N.ConvDesc()
x1r = rows_of(x1)
x2r = rows_of(x2) if C2 else None
out = torch.zeros(B * T * H * W, Cout, device=dev)
d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), b.data_ptr(), out.data_ptr(), Cout
if C2:
    d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, H, W, 1
d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cout, 32, 1.0
if fused:
    d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, coef.data_ptr(), T
kernel = lib.vmm_conv3x3_f32 if exact else lib.vmm_conv3x3_bf16x3
N.check(kernel(C.byref(d), s()), "conv3x3")
torch.cuda.synchronize()
err = (out.double() - ref)
rel = float(err.norm() / ref.norm())
bad = (err.abs().amax(dim=1) > 1e-2 * float(ref.abs().max())).nonzero().flatten()
print(f"shape B{B} T{T} {H}x{W} {C1}+{C2}->{Cout} fused={fused} exact={exact}: rel {rel:.3e}; bad rows {bad.numel()} of {out.shape[0]}", "first bad rows:", bad[:8].tolist(), "last:", bad[-4:].tolist())
# GroupNorm partial sums from the epilogue
G = 8
d.gn_part, d.gn_groups = 1, G
if not fused:
    d.a_imgs_per_sample = T
n = int(lib.vmm_conv3x3_fuses_gn(C.byref(d)))
if n:
    part = torch.full((B * G * n * 2,), float("nan"), device=dev)
    d.gn_part = part.data_ptr()
    out.zero_()
    N.check(kernel(C.byref(d), s()), "conv3x3 gn")
    torch.cuda.synchronize()
    pp = part.view(B, G, n, 2).double()
    got = pp.sum(2)
    r = ref.view(B, T * H * W, G, Cout // G)
    want = torch.stack([r.sum((1, 3)), (r * r).sum((1, 3))], -1)
    print("gn slots", n, "nan slots", int(torch.isnan(part).sum()), "rel err of sums", float((got - want).norm() / want.norm()))
