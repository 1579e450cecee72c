"""Per-kernel parity on the MI355X: each C-ABI entry point against a plain PyTorch fp32 CPU reference
of the same op, on seeded inputs.  Tolerances are fp32-roundoff class (the kernels compute in fp32)."""
import ctypes as C
import math
import os

import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu


def _lib():
    from videometamaterials_amd import _native as N
    return N, N.lib()


def _s():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def rows_of(x):  # (B,C,T,H,W) -> (B*T*H*W, C)
    return x.permute(0, 2, 3, 4, 1).reshape(-1, x.shape[1]).contiguous()


def unrows(r, B, T, H, W):  # (rows, C) -> (B,C,T,H,W)
    return r.reshape(B, T, H, W, -1).permute(0, 4, 1, 2, 3).contiguous()


def relerr(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm().clamp_min(1e-30))


def run_conv(N, lib, dev, a1, w_packed, Cout, *, a2=None, bias=None, KH=1, KW=1, stride=1, off=(0, 0), sgn=(1, 1), nimg, Hin, Win, Hv, Wv,
             Hout=None, Wout=None, oscale=1, oo=(0, 0), out=None, res=None, rot=None, rot_T=1, rot_ncols=0, q_scale=1.0, q_ncols=0, a_coef=None, T=1, rot_dh=32):
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = a1.data_ptr(), a1.shape[1], a1.shape[1]
    if a2 is not None:
        d.a2, d.C2, d.lda2 = a2.data_ptr(), a2.shape[1], a2.shape[1]
    d.w = w_packed.data_ptr()
    d.bias = bias.data_ptr() if bias is not None else None
    Hout, Wout = Hout or Hv, Wout or Wv
    if out is None:
        out = torch.zeros(nimg * Hout * Wout, Cout, device=dev)
    d.out, d.ldo = out.data_ptr(), Cout
    if res is not None:
        d.res, d.ldres = res.data_ptr(), res.shape[1]
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, Hin, Win, Hv, Wv, stride
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = KH, KW, off[0], off[1], sgn[0], sgn[1]
    d.Hout, d.Wout, d.oscale, d.ooh, d.oow = Hout, Wout, oscale, oo[0], oo[1]
    d.Cout = Cout
    if rot is not None:
        d.rot_tab = rot.data_ptr()
    d.rot_T, d.rot_HW, d.rot_ncols, d.rot_dh = rot_T, Hin * Win, rot_ncols, rot_dh
    d.q_scale, d.q_ncols = q_scale, q_ncols
    if a_coef is not None:
        d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, a_coef.data_ptr(), T
    N.check(lib.vmm_conv_igemm_f32(C.byref(d), _s()), "conv")
    torch.cuda.synchronize()
    return out


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,k,stride", [
    (2, 3, 12, 12, 16, 32, 3, 1), (1, 2, 24, 20, 64, 64, 3, 1), (1, 1, 12, 12, 256, 128, 3, 1), (2, 2, 16, 16, 32, 48, 1, 1),
    (1, 2, 16, 16, 4, 16, 7, 1), (2, 2, 16, 16, 32, 32, 4, 2), (1, 11, 12, 12, 128, 512, 3, 1), (1, 2, 96, 96, 64, 64, 3, 1),
    (1, 2, 16, 16, 4, 64, 9, 1), (2, 1, 20, 12, 4, 16, 11, 1), (1, 1, 16, 16, 4, 32, 5, 1)])  # (init_kernel_size off its default, vddp.py:584, 621-626)
def test_conv_igemm(gpu, B, T, H, W, Cin, Cout, k, stride):
    N, lib = _lib()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    pad = {4: 1}.get(k, k // 2)
    ref = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w, b, stride=stride, padding=pad)
    Ho, Wo = ref.shape[-2:]
    ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    wp = w.permute(2, 3, 1, 0).reshape(k * k * Cin, Cout).contiguous().to(gpu)
    out = run_conv(N, lib, gpu, rows_of(x).to(gpu), wp, Cout, bias=b.to(gpu), KH=k, KW=k, stride=stride, off=(-pad, -pad), nimg=B * T, Hin=H, Win=W,
                   Hv=Ho, Wv=Wo)
    assert relerr(out.cpu(), ref_rows) < 2e-6


@pytest.mark.parametrize("B,T,H,W,Cin,Cout,k,stride", [
    (2, 3, 12, 12, 16, 32, 3, 1), (1, 2, 24, 20, 64, 64, 3, 1), (1, 1, 12, 12, 256, 128, 3, 1), (1, 2, 16, 16, 4, 16, 7, 1), (2, 2, 16, 16, 32, 32, 4, 2),
    (1, 11, 12, 12, 128, 512, 3, 1), (1, 2, 96, 96, 64, 64, 3, 1), (1, 2, 20, 20, 36, 768, 1, 1), (1, 2, 16, 16, 4, 64, 9, 1), (2, 1, 20, 12, 4, 16, 11, 1)])
@pytest.mark.parametrize("variant", ["bf16x3", "bf16", "fp16"])
def test_conv_igemm_bf16x3(gpu, B, T, H, W, Cin, Cout, k, stride, variant):
    """Split-bf16 matrix-core path: weights through vmm_pack_weights fmt 1, result within 5e-5 of the fp32 convolution.  `_bf16` / `_fp16` (the
    single-pass instances of the reduced-precision training legs; fmt 1 | 16 planes for fp16): what they must compute is known exactly -- the
    fp32-accumulated convolution of the operands rounded to the 16-bit type -- and is checked at 1e-5."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, Cin, T, H, W, generator=g)
    w = torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)
    b = torch.randn(Cout, generator=g)
    pad = {4: 1}.get(k, k // 2)
    ref = F.conv2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w, b, stride=stride, padding=pad)
    Ho, Wo = ref.shape[-2:]
    ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    K = k * k * Cin
    Kpad = (K + 31) // 32 * 32
    wg = w.contiguous().to(gpu)
    packed = torch.zeros(Cout * Kpad, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = k, k, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * k * k, k * k, k, 1, 0, 1, 0, 1, 0, 1 | (16 if variant == "fp16" else 0)
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Kpad, 0, _s()), "pack")
    tol = 5e-5
    if variant != "bf16x3":
        rd = torch.float16 if variant == "fp16" else torch.bfloat16
        ref = F.conv2d(x.to(rd).double().permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w.to(rd).double(), b.double(), stride=stride, padding=pad)
        ref_rows, tol = ref.permute(0, 2, 3, 1).reshape(-1, Cout).float(), 1e-5
    d = N.ConvDesc()
    xr, bg = rows_of(x).to(gpu), b.to(gpu)
    out = torch.zeros(B * T * Ho * Wo, Cout, device=gpu)
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = xr.data_ptr(), Cin, Cin, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, Ho, Wo, stride
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = k, k, -pad, -pad, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = Ho, Wo, 1, Cout, 32, 1.0
    N.check(getattr(lib, "vmm_conv_igemm_" + variant)(C.byref(d), _s()), "conv " + variant)
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref_rows) < tol


@pytest.mark.parametrize("B,T,H,W,C1,C2,Cout,fused", [(1, 2, 96, 96, 64, 0, 64, True), (2, 3, 48, 48, 64, 64, 128, False), (2, 2, 24, 24, 256, 0, 256, True),
                                                      (1, 2, 12, 12, 512, 0, 512, False), (2, 3, 10, 20, 64, 64, 128, True)])
def test_conv3x3_bf16_single_pass(gpu, B, T, H, W, C1, C2, Cout, fused):
    """vmm_conv3x3_bf16, the "bf16" throughput mode of the 3x3 kernel (BASELINE.json configs[3]): ONE matrix pass on the operands' bf16 roundings.  What it must
    compute is therefore known exactly: the fp32-accumulated convolution of bf16(operand) x bf16(weight) -- checked at 1e-5 --, which is 2^-9-class (here
    < 1e-2) away from the fp32 convolution.  2-D and flat tiles, two sources, fused GN+SiLU operand, bias, residual, split channel chunks."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(12)
    Cin = C1 + C2
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    x2 = torch.randn(B, C2, T, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B * T * H * W, Cout, generator=g)
    xa, coef = x1, None
    if fused:
        coef = torch.randn(B, C1, 2, generator=g)
        xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    xin = torch.cat([xa, x2], 1) if C2 else xa
    rnd = lambda t: t.to(torch.bfloat16).double()
    as_rows = lambda t: t.permute(0, 2, 3, 1).reshape(-1, Cout)
    x4 = xin.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W)
    want = as_rows(F.conv2d(rnd(x4), rnd(w), b.double(), padding=1)).float() + res
    full = as_rows(F.conv2d(x4.double(), w.double(), b.double(), padding=1)).float() + res
    K = 9 * Cin
    Kpad = (K + 31) // 32 * 32
    wg = w.contiguous().to(gpu)
    packed = torch.zeros((Cout + 31) // 32 * 32 * Kpad, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 2
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Kpad, 0, _s()), "pack")
    d = N.ConvDesc()
    x1r, bg, rg = rows_of(x1).to(gpu), b.to(gpu), res.to(gpu)
    x2r = rows_of(x2).to(gpu) if C2 else None
    cg = coef.to(gpu) if fused else None
    out = torch.zeros(B * T * H * W, Cout, device=gpu)
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    if C2:
        d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
    d.res, d.ldres = rg.data_ptr(), Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cout, 32, 1.0
    if fused:
        d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, cg.data_ptr(), T
    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)
    d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
    N.check(lib.vmm_conv3x3_bf16(C.byref(d), _s()), "conv3x3 single pass")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), want) < (1e-5 if not fused else 2e-3)  # (fused: the operand is rounded AFTER silu(x a + b), whose last fp32 bits differ from torch's)
    assert 1e-4 < relerr(out.cpu(), full) < 1e-2
    assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("B,T,H,W,C1,C2,Cout,fused", [(1, 2, 96, 96, 64, 0, 64, False), (2, 1, 96, 96, 64, 0, 64, True), (2, 3, 48, 48, 64, 64, 128, True),
                                                      (1, 2, 48, 48, 128, 0, 128, False), (2, 2, 24, 24, 256, 0, 256, True), (1, 3, 24, 24, 256, 256, 128, False),
                                                      (2, 1, 12, 12, 512, 0, 512, True), (2, 3, 10, 20, 64, 16, 64, True), (1, 1, 16, 16, 16, 0, 64, False),
                                                      (3, 11, 48, 48, 64, 0, 128, True)])
def test_conv3x3_winograd_bf16x3(gpu, B, T, H, W, C1, C2, Cout, fused):
    """Winograd F(2x2, 3x3) form of the 3x3 convolution (weights G g G^T in fragment order, fmt 8): image borders, partial tile blocks (48 / 36 / 50 of 64
    tiles), two sources, fused per-sample GN+SiLU operand, bias, residual, the GroupNorm partial sums of the epilogue; same tolerance as the direct kernel.
    An experiment (include/vmm_experiments.h): runs in the process test_experiments_library starts on libvmm_hip_exp.so, skipped on the product library."""
    N, lib = _lib()
    if not N.experiments_built():
        pytest.skip("product library: the Winograd kernel lives in libvmm_hip_exp.so")
    g = torch.Generator().manual_seed(21)
    Cin = C1 + C2
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    x2 = torch.randn(B, C2, T, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B * T * H * W, Cout, generator=g)
    xa, coef = x1, None
    if fused:
        coef = torch.randn(B, C1, 2, generator=g)
        xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    xin = torch.cat([xa, x2], 1) if C2 else xa
    ref = F.conv2d(xin.double().permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w.double(), b.double(), padding=1).permute(0, 2, 3, 1).reshape(-1, Cout).float() + res
    wg = w.contiguous().to(gpu)
    packed = torch.zeros(16 * Cin * Cout, device=gpu)  # 64 bytes per (channel, column)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 8
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Cin * 16, 0, _s()), "pack")
    d = N.ConvDesc()
    x1r, bg, rg = rows_of(x1).to(gpu), b.to(gpu), res.to(gpu)
    x2r = rows_of(x2).to(gpu) if C2 else None
    cg = coef.to(gpu) if fused else None
    out = torch.zeros(B * T * H * W, Cout, device=gpu)
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    if C2:
        d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
    d.res, d.ldres = rg.data_ptr(), Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cout, 32, 1.0
    if fused:
        d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, cg.data_ptr(), T
    assert lib.vmm_conv3x3_wino_accepts(C.byref(d)) == 1
    N.check(lib.vmm_conv3x3_wino_bf16x3(C.byref(d), _s()), "conv3x3 winograd")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 5e-5
    first = out.clone()
    out.fill_(7.0)
    N.check(lib.vmm_conv3x3_wino_bf16x3(C.byref(d), _s()), "conv3x3 winograd")
    torch.cuda.synchronize()
    assert torch.equal(out, first)
    # GroupNorm sums of the output (+ bias) from the epilogue
    G = 8
    d.res, d.ldres = None, 0
    d.gn_part, d.gn_groups, d.a_imgs_per_sample = 1, G, T
    n_part = lib.vmm_conv3x3_wino_fuses_gn(C.byref(d))
    assert n_part > 0
    part = torch.full((B * G, n_part, 2), float("nan"), device=gpu)
    d.gn_part = part.data_ptr()
    N.check(lib.vmm_conv3x3_wino_bf16x3(C.byref(d), _s()), "conv3x3 winograd + gn sums")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref - res) < 5e-5
    y = (ref - res).reshape(B, T * H * W, G, Cout // G).double()
    want = torch.stack([y.sum((1, 3)), (y * y).sum((1, 3))], -1).reshape(B * G, 2)
    got = part.double().sum(1).cpu()
    assert not torch.isnan(got).any()
    assert (got - want).abs().max() <= 2e-5 * want.abs().max()


@pytest.mark.parametrize("B,T,H,W,C1,C2,Cout,fused", [(1, 2, 12, 12, 64, 0, 64, False), (2, 1, 24, 24, 32, 32, 128, False), (1, 2, 96, 96, 64, 0, 64, True),
                                                      (1, 1, 12, 12, 512, 512, 256, False), (2, 3, 10, 20, 64, 64, 128, True), (2, 1, 32, 32, 64, 0, 64, True),
                                                      (2, 1, 48, 48, 32, 32, 128, True), (1, 11, 12, 12, 256, 0, 512, False),
                                                      # more tiles than resident workgroups: the multi-tile kernels (a workgroup walks several tiles)
                                                      (2, 9, 96, 96, 64, 0, 64, True), (3, 11, 48, 48, 64, 0, 128, True), (1, 17, 96, 96, 32, 32, 64, False)])
@pytest.mark.parametrize("exact", [False, True])
def test_conv3x3_halo_bf16x3(gpu, B, T, H, W, C1, C2, Cout, fused, exact):
    """LDS halo-patch 3x3 kernel (weights in fragment order, fmt 2; exact: its fp32-MFMA variant vmm_conv3x3_f32, fmt 4): image borders, flat row tiles running across frames and samples with a
    partial last tile, 2-D pixel tiles, two sources, fused per-sample GN+SiLU operand, residual, split channel chunks (atomics)."""
    N, lib = _lib()
    kernel, tol = (lib.vmm_conv3x3_f32, 3e-6) if exact else (lib.vmm_conv3x3_bf16x3, 5e-5)
    g = torch.Generator().manual_seed(12)
    Cin = C1 + C2
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    x2 = torch.randn(B, C2, T, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    res = torch.randn(B * T * H * W, Cout, generator=g)
    xa = x1
    coef = None
    if fused:
        coef = torch.randn(B, C1, 2, generator=g)
        xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    xin = torch.cat([xa, x2], 1) if C2 else xa
    ref = F.conv2d(xin.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout) + res
    K = 9 * Cin
    Kpad = (K + 31) // 32 * 32
    wg = w.contiguous().to(gpu)
    packed = torch.zeros((Cout + 31) // 32 * 32 * Kpad, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, 4 if exact else 2
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Kpad, 0, _s()), "pack")
    d = N.ConvDesc()
    x1r, bg, rg = rows_of(x1).to(gpu), b.to(gpu), res.to(gpu)
    x2r = rows_of(x2).to(gpu) if C2 else None
    cg = coef.to(gpu) if fused else None
    out = torch.zeros(B * T * H * W, Cout, device=gpu)
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    if C2:
        d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
    d.res, d.ldres = rg.data_ptr(), Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cout, 32, 1.0
    if fused:
        d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, cg.data_ptr(), T
    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)
    d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
    N.check(kernel(C.byref(d), _s()), "conv3x3 halo")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < tol
    # the split channel reduction of the few-row layers adds its partial sums in a fixed order: bit-reproducible, tickets left at zero
    first = out.clone()
    for _ in range(3):
        out.fill_(7.0)
        N.check(kernel(C.byref(d), _s()), "conv3x3 halo")
        torch.cuda.synchronize()
        assert torch.equal(out, first)
    assert int(tickets.abs().sum()) == 0
    # GroupNorm sums of the output from the epilogue (unsplit layers without a residual; flat row tiles keep two sets of sums where they cross a sample boundary)
    G = 8
    d.res, d.ldres = None, 0
    d.split_tickets, d.n_tickets = None, 0  # (these small cases would otherwise split the channel reduction)
    d.gn_part, d.gn_groups, d.a_imgs_per_sample = 1, G, T
    n_part = lib.vmm_conv3x3_fuses_gn(C.byref(d))
    if n_part:
        part = torch.full((B * G, n_part, 2), float("nan"), device=gpu)
        d.gn_part = part.data_ptr()
        N.check(kernel(C.byref(d), _s()), "conv3x3 halo + gn sums")
        torch.cuda.synchronize()
        y = (ref - res).reshape(B, T * H * W, G, Cout // G).double()
        want = torch.stack([y.sum((1, 3)), (y * y).sum((1, 3))], -1)
        assert relerr(out.cpu(), ref - res) < tol
        assert relerr(part.cpu().double().sum(1).reshape(B, G, 2), want) < 2e-5  # every slot written exactly once (no NaN left)
    else:  # only flat row tiles longer than a sample (a tile would touch three samples) go without the fused sums
        assert not (W >= 32 and W % 16 == 0 and H % 16 == 0)
        if os.environ.get("VMM_C3_PERSISTENT") != "2":  # (2 = the persistent kernel for every shape: it fuses the sums for 2-D tiles only)
            assert T * H * W < (128 if Cout >= 128 else 256)


@pytest.mark.parametrize("B,T,H,W,C1,C2,Cout,fused,slots", [(1, 11, 12, 12, 256, 0, 512, False, 64), (1, 11, 12, 12, 256, 0, 512, True, 40), (1, 11, 12, 12, 256, 0, 512, False, 200),
                                                            (2, 3, 12, 12, 128, 128, 256, True, 24), (1, 2, 48, 48, 128, 0, 128, True, 24), (1, 3, 24, 24, 256, 0, 128, False, 16),
                                                            
                                                            # a full grid (two workgroups per CU) by the library's own rule: the sampler's 24 x 24 level at batch 8
                                                            (8, 11, 24, 24, 128, 0, 256, True, 512)])
@pytest.mark.parametrize("arith", ["x3", "f32", "one"])
def test_conv3x3_balanced_launch(gpu, B, T, H, W, C1, C2, Cout, fused, slots, arith):
    """The balanced ("stream-K") launch of the few-tile 3 x 3 layers (conv3x3_sk_kernel: a grid of `slots` workgroups shares the (tile, channel chunk)
    iterations evenly; tails / middles of tiles travel as partial tiles, heads collect them in iteration order): against F.conv2d, against the
    one-workgroup-per-tile launch of the same descriptor, bit-reproducible over repeated launches, flags left zero, fused GroupNorm sums."""
    N, lib = _lib()
    if not N.experiments_built():
        pytest.skip("product library: the balanced launch lives in libvmm_hip_exp.so (test_experiments_library runs this test there)")
    kernel, tol, fmt = {"x3": (lib.vmm_conv3x3_bf16x3, 5e-5, 2), "f32": (lib.vmm_conv3x3_f32, 3e-6, 4), "one": (lib.vmm_conv3x3_bf16, 2e-2, 2)}[arith]
    g = torch.Generator().manual_seed(31)
    Cin = C1 + C2
    x1 = torch.randn(B, C1, T, H, W, generator=g)
    x2 = torch.randn(B, C2, T, H, W, generator=g) if C2 else None
    w = torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)
    b = torch.randn(Cout, generator=g)
    xa, coef = x1, None
    if fused:
        coef = torch.randn(B, C1, 2, generator=g)
        xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    xin = torch.cat([xa, x2], 1) if C2 else xa
    ref = F.conv2d(xin.permute(0, 2, 1, 3, 4).reshape(B * T, Cin, H, W), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    K = 9 * Cin
    Kpad = (K + 31) // 32 * 32
    wg = w.contiguous().to(gpu)
    packed = torch.zeros((Cout + 31) // 32 * 32 * Kpad, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cin * 9, 9, 3, 1, 0, 1, 0, 1, 0, fmt
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cout * Kpad, 0, _s()), "pack")
    d = N.ConvDesc()
    x1r, bg = rows_of(x1).to(gpu), b.to(gpu)
    x2r = rows_of(x2).to(gpu) if C2 else None
    cg = coef.to(gpu) if fused else None
    out = torch.zeros(B * T * H * W, Cout, device=gpu)
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1r.data_ptr(), C1, C1, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    if C2:
        d.a2, d.C2, d.lda2 = x2r.data_ptr(), C2, C2
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cout, 32, 1.0
    d.a_imgs_per_sample = T
    if fused:
        d.a_mode, d.a_coef = 1, cg.data_ptr()
    # reference launch: one workgroup per tile (no workspace, no tickets: neither a split channel reduction nor the balanced grid)
    N.check(kernel(C.byref(d), _s()), "conv3x3 plain")
    torch.cuda.synchronize()
    plain = out.clone()
    assert relerr(plain.cpu(), ref) < tol
    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)
    work = torch.full((slots * 128 * 128,), float("nan"), device=gpu)
    d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
    d.sk_work, d.sk_slots = work.data_ptr(), slots
    G = 8
    d.gn_part, d.gn_groups = 1, G
    n_part = lib.vmm_conv3x3_fuses_gn(C.byref(d))
    assert n_part > 0
    part = torch.full((B * G, n_part, 2), float("nan"), device=gpu)
    d.gn_part = part.data_ptr()
    out.fill_(7.0)
    N.check(kernel(C.byref(d), _s()), "conv3x3 balanced")
    torch.cuda.synchronize()
    first = out.clone()
    assert relerr(first.cpu(), ref) < tol
    assert relerr(first.cpu(), plain.cpu()) < (1e-2 if arith == "one" else 2e-6)  # the same products, summed piecewise
    assert not torch.isnan(work).all()  # the balanced grid ran (partial tiles were published) ...
    assert int(tickets.abs().sum()) == 0  # ... and every flag is back to zero
    y = ref.reshape(B, T * H * W, G, Cout // G).double()
    want = torch.stack([y.sum((1, 3)), (y * y).sum((1, 3))], -1)
    assert relerr(part.cpu().double().sum(1).reshape(B, G, 2), want) < (5e-3 if arith == "one" else 2e-5)
    for _ in range(3):
        out.fill_(7.0)
        N.check(kernel(C.byref(d), _s()), "conv3x3 balanced")
        torch.cuda.synchronize()
        assert torch.equal(out, first)
    assert int(tickets.abs().sum()) == 0


@pytest.mark.parametrize("exact", [False, True])
def test_conv3x3_shared_source_frames(gpu, exact):
    """a_img_mod of the descriptor (mirrored plans): the frames of the batch's second half read the first half's rows of a1 -- one
    pre-norm tensor shared by both guidance branches -- under each sample's own GroupNorm * FiLM coefficients; against torch on the duplicated
    tensor.  Kernels that do not honour the field must refuse the descriptor."""
    N, lib = _lib()
    kernel, tol = (lib.vmm_conv3x3_f32, 3e-6) if exact else (lib.vmm_conv3x3_bf16x3, 5e-5)
    g = torch.Generator().manual_seed(21)
    Bh, T, H, W, Cc = 2, 9, 96, 96, 64  # 2 x Bh samples; enough tiles for the unsplit 2-D instance
    h1 = torch.randn(Bh, Cc, T, H, W, generator=g)
    coef = torch.randn(2 * Bh, Cc, 2, generator=g)
    w = torch.randn(Cc, Cc, 3, 3, generator=g) / math.sqrt(Cc * 9)
    b = torch.randn(Cc, generator=g)
    both = torch.cat([h1, h1], 0)
    xa = F.silu(both * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    ref = F.conv2d(xa.permute(0, 2, 1, 3, 4).reshape(2 * Bh * T, Cc, H, W), w, b, padding=1).permute(0, 2, 3, 1).reshape(-1, Cc)
    K = 9 * Cc
    wg = w.contiguous().to(gpu)
    packed = torch.zeros(Cc * K, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 3, 3, Cc, Cc, Cc
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cc * 9, 9, 3, 1, 0, 1, 0, 1, 0, 4 if exact else 2
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, Cc * K, 0, _s()), "pack")
    hr, bg, cg = rows_of(h1).to(gpu), b.to(gpu), coef.to(gpu)
    out = torch.full((2 * Bh * T * H * W, Cc), 7.0, device=gpu)
    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = hr.data_ptr(), Cc, Cc, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cc
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = 2 * Bh * T, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = H, W, 1, Cc, 32, 1.0
    d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, cg.data_ptr(), T
    d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
    d.a_img_mod = Bh * T
    N.check(kernel(C.byref(d), _s()), "conv3x3 shared source")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < tol
    assert lib.vmm_conv_igemm_f32(C.byref(d), _s()) < 0 and lib.vmm_conv_igemm_bf16x3(C.byref(d), _s()) < 0
    d.Hin = d.Hv = d.Hout = 12  # (flat row tiles: no shared source frames)
    d.Win = d.Wv = d.Wout = 12
    assert kernel(C.byref(d), _s()) == 1


def test_conv_concat_residual_and_fused_gn(gpu):
    """two-source input (torch.cat skip), residual epilogue, and the fused GroupNorm+FiLM+SiLU operand transform."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(1)
    B, T, H, W, C1, C2, Cout = 2, 2, 10, 14, 32, 16, 64
    x1, x2 = torch.randn(B, C1, T, H, W, generator=g), torch.randn(B, C2, T, H, W, generator=g)
    w = torch.randn(Cout, C1 + C2, 3, 3, generator=g) / math.sqrt((C1 + C2) * 9)
    res = torch.randn(B * T * H * W, Cout, generator=g)
    ref = F.conv2d(torch.cat([x1, x2], 1).permute(0, 2, 1, 3, 4).reshape(B * T, C1 + C2, H, W), w, None, padding=1)
    ref_rows = ref.permute(0, 2, 3, 1).reshape(-1, Cout) + res
    wp = w.permute(2, 3, 1, 0).reshape(9 * (C1 + C2), Cout).contiguous().to(gpu)
    out = run_conv(N, lib, gpu, rows_of(x1).to(gpu), wp, Cout, a2=rows_of(x2).to(gpu), KH=3, KW=3, off=(-1, -1), nimg=B * T, Hin=H, Win=W, Hv=H, Wv=W,
                   res=res.to(gpu))
    assert relerr(out.cpu(), ref_rows) < 2e-6
    # fused transform: conv(silu(x*a+b)) with per-(sample, channel) coefficients, zero padding applied AFTER the activation
    coef = torch.randn(B, C1, 2, generator=g)
    xa = F.silu(x1 * coef[:, :, 0][:, :, None, None, None] + coef[:, :, 1][:, :, None, None, None])
    w1 = torch.randn(Cout, C1, 3, 3, generator=g) / math.sqrt(C1 * 9)
    ref = F.conv2d(xa.permute(0, 2, 1, 3, 4).reshape(B * T, C1, H, W), w1, None, padding=1).permute(0, 2, 3, 1).reshape(-1, Cout)
    wp = w1.permute(2, 3, 1, 0).reshape(9 * C1, Cout).contiguous().to(gpu)
    out = run_conv(N, lib, gpu, rows_of(x1).to(gpu), wp, Cout, KH=3, KW=3, off=(-1, -1), nimg=B * T, Hin=H, Win=W, Hv=H, Wv=W, a_coef=coef.to(gpu), T=T)
    assert relerr(out.cpu(), ref) < 5e-6


def test_conv_transpose_as_four_phases(gpu):
    N, lib = _lib()
    g = torch.Generator().manual_seed(2)
    B, T, H, W, Cc = 1, 2, 6, 8, 32
    x = torch.randn(B, Cc, T, H, W, generator=g)
    w = torch.randn(Cc, Cc, 4, 4, generator=g) / math.sqrt(Cc * 4)
    b = torch.randn(Cc, generator=g)
    ref = F.conv_transpose2d(x.permute(0, 2, 1, 3, 4).reshape(B * T, Cc, H, W), w, b, stride=2, padding=1).permute(0, 2, 3, 1).reshape(-1, Cc)
    out = torch.zeros(B * T * 4 * H * W, Cc, device=gpu)
    for ph in range(2):
        for pw in range(2):
            wp = w[:, :, [1 - ph, 3 - ph], :][:, :, :, [1 - pw, 3 - pw]].permute(2, 3, 0, 1).reshape(4 * Cc, Cc).contiguous().to(gpu)
            run_conv(N, lib, gpu, rows_of(x).to(gpu), wp, Cc, bias=b.to(gpu), KH=2, KW=2, off=(ph, pw), sgn=(-1, -1), nimg=B * T, Hin=H, Win=W, Hv=H, Wv=W,
                     Hout=2 * H, Wout=2 * W, oscale=2, oo=(ph, pw), out=out)
    assert relerr(out.cpu(), ref) < 2e-6


@pytest.mark.parametrize("up,nimg,H,W,Cin,Cout", [
    (1, 3, 48, 48, 64, 64),     # 2-D tiles, four phases = 256 columns (ups.2.4)
    (1, 5, 24, 24, 128, 128),   # flat row tiles across frames, partial last tile
    (1, 2, 12, 12, 256, 256),   # 12 x 12 level
    (1, 1, 8, 32, 32, 64),      # non-square 2-D tiles, one channel chunk
    (0, 3, 96, 96, 64, 64),     # cells 48 x 48: 2-D tiles of 256 cells x 64 columns (downs.0.4)
    (0, 5, 48, 48, 128, 128),   # cells 24 x 24: flat tiles
    (0, 2, 24, 24, 256, 256),   # cells 12 x 12
    (0, 1, 64, 32, 32, 128),    # cells 32 x 16: flat (16 wide), one chunk per sub-pixel
])
def test_stride2_resampling_as_tap_subset_convolutions(gpu, up, nimg, H, W, Cin, Cout):
    """vmm_conv_s2_bf16x3: Downsample = Conv3d (1,4,4)/2 pad 1 (vddp.py:158) and Upsample = ConvTranspose3d (1,4,4)/2 pad 1 (vddp.py:155) as 3 x 3
    neighbourhood convolutions whose units use a 2 x 2 corner of the taps (pack formats 5 / 6), against torch's convolutions."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(31 + H + Cin + up)
    x = torch.randn(nimg, Cin, H, W, generator=g)
    b = torch.randn(Cout, generator=g)
    if up:
        w = torch.randn(Cin, Cout, 4, 4, generator=g) / math.sqrt(Cin * 4)
        ref = F.conv_transpose2d(x, w, b, stride=2, padding=1)
        sn, sc = 16, Cout * 16
    else:
        w = torch.randn(Cout, Cin, 4, 4, generator=g) / math.sqrt(Cin * 16)
        ref = F.conv2d(x, w, b, stride=2, padding=1)
        sn, sc = Cin * 16, 16
    Ho, Wo = ref.shape[2], ref.shape[3]
    ref = ref.permute(0, 2, 3, 1).reshape(-1, Cout)
    wg = w.contiguous().to(gpu)
    n_el = 9 * (Cin if up else 4 * Cin) * (4 * Cout if up else Cout)
    packed = torch.zeros(n_el, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 4, 4, Cin, Cin, Cout
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = sn, sc, 4, 1, 0, 1, 0, 1, 0, 6 if up else 5
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, n_el, 0, _s()), "pack")
    xg = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(gpu)
    out = torch.full((nimg * Ho * Wo, Cout), 7.0, device=gpu)
    bg = b.to(gpu)
    rc = lib.vmm_conv_s2_bf16x3(xg.data_ptr(), Cin, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout, nimg, H, W, Cin, Cout, up, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 2e-5
    # ... accumulating into its output (the layers' data gradients add to a gradient buffer that already holds the skip connection's share), and with the
    # channel reduction of few-tile Downsample layers split over several workgroups per tile (fixed-order tickets: bit-reproducible, tickets left at zero)
    res = torch.randn(nimg * Ho * Wo, Cout, generator=g)
    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)
    first = None
    for _ in range(3):
        out.copy_(res.to(gpu))
        rc = lib.vmm_conv_s2_acc_bf16x3(xg.data_ptr(), Cin, packed.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout, out.data_ptr(), Cout, nimg, H, W, Cin, Cout, up,
                                        tickets.data_ptr(), tickets.numel(), _s())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert relerr(out.cpu(), ref + res) < 2e-5
        first = out.clone() if first is None else first
        assert torch.equal(out, first)
    assert int(tickets.abs().sum()) == 0
    # the "bf16" throughput mode of the same layers: exactly the fp32-accumulated result on bf16-rounded operands
    rnd = lambda t_: t_.to(torch.bfloat16).double()
    if up:
        want1 = F.conv_transpose2d(rnd(x), rnd(w), b.double(), stride=2, padding=1)
    else:
        want1 = F.conv2d(rnd(x), rnd(w), b.double(), stride=2, padding=1)
    want1 = want1.permute(0, 2, 3, 1).reshape(-1, Cout).float()
    out.fill_(7.0)
    rc = lib.vmm_conv_s2_acc_bf16(xg.data_ptr(), Cin, packed.data_ptr(), bg.data_ptr(), None, 0, out.data_ptr(), Cout, nimg, H, W, Cin, Cout, up,
                                  tickets.data_ptr(), tickets.numel(), _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(out.cpu(), want1) < 1e-5
    assert 1e-4 < relerr(out.cpu(), ref) < 1e-2


@pytest.mark.parametrize("nimg,H,W,Cc,k", [(3, 96, 96, 3, 7), (2, 20, 37, 4, 5), (1, 8, 8, 1, 3), (5, 16, 48, 2, 7)])
def test_stem_convolution(gpu, nimg, H, W, Cc, k):
    """vmm_stem_conv_bf16x3 (init_conv, vddp.py:600: Conv3d(channels, 64, (1,k,k)) pad k/2) against torch; whole and partial 16 x 16 tiles,
    fewer than four channels, kernel sizes below 7 (zero-weight taps)."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(40 + k + Cc)
    x = torch.randn(nimg, Cc, H, W, generator=g)
    w = torch.randn(64, Cc, k, k, generator=g) / math.sqrt(Cc * k * k)
    b = torch.randn(64, generator=g)
    ref = F.conv2d(x, w, b, padding=k // 2).permute(0, 2, 3, 1).reshape(-1, 64)
    xr = torch.zeros(nimg * H * W, 4)
    xr[:, :Cc] = x.permute(0, 2, 3, 1).reshape(-1, Cc)
    wg = w.contiguous().to(gpu)
    packed = torch.zeros(2048 * k, device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = k, k, Cc, Cc, 64
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = Cc * k * k, k * k, k, 1, 0, 1, 0, 1, 0, 7
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, 2048 * k, 0, _s()), "pack")
    xg, bg = xr.to(gpu), b.to(gpu)
    out = torch.full((nimg * H * W, 64), 7.0, device=gpu)
    rc = lib.vmm_stem_conv_bf16x3(xg.data_ptr(), packed.data_ptr(), bg.data_ptr(), out.data_ptr(), 64, nimg, H, W, 64, k, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 2e-5


@pytest.mark.parametrize("B,T,HW,Cout,with_res", [(2, 3, 100, 3, True), (1, 1, 37, 1, False), (3, 2, 2304, 4, True)])
def test_output_pass_with_final_pointwise_conv(gpu, B, T, HW, Cout, with_res):
    """vmm_affine_silu_pointwise_to_ncthw: silu(x * a + b') + res (the last ResnetBlock's output pass, vddp.py:311) and final_conv.1
    (1x1 convolution to (B, Cout, T, HW), vddp.py:729) in one kernel; partial 64-row groups."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(50 + Cout)
    rows = B * T * HW
    x = torch.randn(rows, 64, generator=g)
    coef = torch.randn(B, 64, 2, generator=g)
    res = torch.randn(rows, 64, generator=g) if with_res else None
    w = torch.randn(Cout, 64, generator=g) / 8
    b = torch.randn(Cout, generator=g)
    a_, b_ = coef[:, :, 0].repeat_interleave(T * HW, 0), coef[:, :, 1].repeat_interleave(T * HW, 0)
    y = F.silu(x * a_ + b_) + (res if with_res else 0)
    ref = (y @ w.t() + b).reshape(B, T, HW, Cout).permute(0, 3, 1, 2).contiguous()
    out = torch.full((B, Cout, T, HW), 7.0, device=gpu)
    xg, cg, wg, bg = x.to(gpu), coef.to(gpu), w.to(gpu), b.to(gpu)
    rg = res.to(gpu) if with_res else None
    rc = lib.vmm_affine_silu_pointwise_to_ncthw(xg.data_ptr(), 64, cg.data_ptr(), rg.data_ptr() if with_res else None, 64, 64, wg.data_ptr(), bg.data_ptr(),
                                                B, Cout, T, HW, out.data_ptr(), _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 3e-6


@pytest.mark.parametrize("wrap", [(1, 1), (0, 1)])
@pytest.mark.parametrize("nimg,H,W,Cin,Cout,k,stride", [(2, 32, 32, 64, 64, 3, 1), (1, 96, 96, 64, 128, 3, 1), (2, 16, 48, 32, 64, 3, 1), (3, 12, 12, 128, 128, 3, 1),
                                                     (2, 16, 16, 32, 32, 4, 2), (1, 24, 24, 4, 64, 7, 1), (2, 10, 6, 20, 24, 3, 1)])
def test_periodic_padding_convolutions(gpu, wrap, nimg, H, W, Cin, Cout, k, stride):
    """wrap_h / wrap_w of the descriptor (padding_mode 'circular' = both axes, 'circular_1d' = w only, vddp.py:163-243): taps that leave the frame
    read the opposite border.  Every kernel that accepts the flags against torch on a circularly padded input: exact-fp32 and split-bf16
    implicit GEMM, both 3 x 3 halo kernels where the shape gets 2-D tiles (they must refuse, not mis-compute, elsewhere), the data gradient
    (reversed taps) and the weight gradient."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(70 + H + k)
    x = torch.randn(nimg, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, k, k, generator=g) / math.sqrt(Cin * k * k)).requires_grad_(True)
    b = torch.randn(Cout, generator=g).requires_grad_(True)
    pad = {3: 1, 7: 3, 4: 1}[k]
    xl = x.clone().requires_grad_(True)
    xp = F.pad(xl, (pad, pad, 0, 0), mode="circular")                       # w always wraps
    xp = F.pad(xp, (0, 0, pad, pad), mode="circular") if wrap[0] else F.pad(xp, (0, 0, pad, pad))
    ref4 = F.conv2d(xp, w, b, stride=stride)
    Ho, Wo = ref4.shape[-2:]
    dy = torch.randn(ref4.shape, generator=g)
    ref4.backward(dy)
    ref = ref4.detach().permute(0, 2, 3, 1).reshape(-1, Cout)
    xr = x.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(gpu)
    bg = b.detach().to(gpu)
    wg = w.detach().contiguous().to(gpu)
    K = k * k * Cin
    Kpad = (K + 31) // 32 * 32

    def packed(fmt, flip=False, transpose=False):
        """fmt 0-like plain [tap][c][n] built on the host, or a vmm_pack_weights format; flip / transpose: the data-gradient operand."""
        src = wg
        ci, co = Cin, Cout
        if transpose:
            src = wg.permute(1, 0, 2, 3).contiguous()
            ci, co = Cout, Cin
        if flip:
            src = src.flip(2, 3).contiguous()
        if fmt == 0:
            return src.permute(2, 3, 1, 0).reshape(k * k * ci, co).contiguous(), ci, co
        kp = (k * k * ci + 31) // 32 * 32
        buf = torch.zeros((co + 31) // 32 * 32 * kp, device=gpu)
        job = (N.PackJob * 1)()
        j = job[0]
        j.torch_w, j.packed = src.data_ptr(), buf.data_ptr()
        j.TH, j.TW, j.C, j.Cp, j.N = k, k, ci, ci, co
        j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = ci * k * k, k * k, k, 1, 0, 1, 0, 1, 0, fmt
        tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
        N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, co * kp, 0, _s()), "pack")
        buf._keep = (src, tab)
        return buf, ci, co

    tickets = torch.zeros(4096, dtype=torch.int32, device=gpu)

    def desc(a, wbuf, ci, co, out, hin, win, hv, wv, bias=None, st=1):
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1, d.w, d.out, d.ldo = a.data_ptr(), ci, ci, wbuf.data_ptr(), out.data_ptr(), co
        d.bias = bias.data_ptr() if bias is not None else None
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, hin, win, hv, wv, st
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = k, k, -pad, -pad, 1, 1
        d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = hv, wv, 1, co, 32, 1.0
        d.split_tickets, d.n_tickets = tickets.data_ptr(), tickets.numel()
        d.wrap_h, d.wrap_w = wrap
        return d

    # forward: the two implicit GEMMs, and the halo kernels where they take the shape
    ran = set()
    for name, fn, fmt, tol in (("igemm_f32", lib.vmm_conv_igemm_f32, 0, 3e-6), ("igemm_x3", lib.vmm_conv_igemm_bf16x3, 1, 5e-5),
                               ("halo_x3", lib.vmm_conv3x3_bf16x3, 2, 5e-5), ("halo_f32", lib.vmm_conv3x3_f32, 4, 3e-6)):
        if name.startswith("halo") and (k != 3 or stride != 1 or Cin % 32 or Cout % 32):
            continue
        wbuf, ci, co = packed(fmt)
        out = torch.full((nimg * Ho * Wo, Cout), 7.0, device=gpu)
        rc = fn(C.byref(desc(xr, wbuf, ci, co, out, H, W, Ho, Wo, bias=bg, st=stride)), _s())
        torch.cuda.synchronize()
        if name.startswith("halo") and rc == 1:  # flat row tiles: no periodic neighbourhood, nothing launched
            assert torch.all(out == 7.0)
            continue
        assert rc == 0, (name, rc)
        assert relerr(out.cpu(), ref) < tol, name
        ran.add(name)
    assert {"igemm_f32", "igemm_x3"} <= ran
    if k == 3 and W >= 32 and W % 16 == 0 and H % 16 == 0 and Cin % 32 == 0:
        assert {"halo_x3", "halo_f32"} <= ran  # the 2-D-tiled instances do wrap
    if k == 3:
        # data gradient = the periodic convolution of dY with the reversed, transposed taps
        dyr = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(gpu)
        want_dx = xl.grad.permute(0, 2, 3, 1).reshape(-1, Cin)
        for name, fn, fmt, tol in (("igemm_f32", lib.vmm_conv_igemm_f32, 0, 3e-6), ("halo_x3", lib.vmm_conv3x3_bf16x3, 2, 5e-5)):
            if name == "halo_x3" and (Cin % 32 or Cout % 32):
                continue
            wbuf, ci, co = packed(fmt, flip=True, transpose=True)
            dx = torch.full((nimg * H * W, Cin), 7.0, device=gpu)
            rc = fn(C.byref(desc(dyr, wbuf, ci, co, dx, H, W, H, W)), _s())
            torch.cuda.synchronize()
            if rc == 1 and name == "halo_x3":
                continue
            assert rc == 0, (name, rc)
            assert relerr(dx.cpu(), want_dx) < tol, name
    # weight and bias gradient (generic kernel: the nine-tap kernel declines periodic layers)
    if Cin % 4 == 0 and Cout % 4 == 0:
        dyr = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(gpu)
        nsplit = 5
        dw = torch.zeros(K, Cout, device=gpu)
        db = torch.zeros(Cout, device=gpu)
        scratch = torch.zeros(nsplit, Cout, device=gpu)
        dummy = torch.zeros(1, device=gpu)
        d = desc(xr, dummy, Cin, Cout, dummy, H, W, Ho, Wo, st=stride)
        assert lib.vmm_conv3x3_wgrad_f32(C.byref(d), dyr.data_ptr(), Cout, dw.data_ptr(), nsplit, db.data_ptr(), scratch.data_ptr(), _s()) == 1
        rc = lib.vmm_conv_wgrad_f32(C.byref(d), dyr.data_ptr(), Cout, dw.data_ptr(), nsplit, db.data_ptr(), scratch.data_ptr(), _s())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert relerr(dw.cpu(), w.grad.permute(2, 3, 1, 0).reshape(K, Cout)) < 3e-6
        assert relerr(db.cpu(), b.grad) < 3e-6


@pytest.mark.parametrize("nimg,H,W,C1,C2,Cout,nsplit", [(2, 48, 48, 64, 0, 64, 37), (1, 12, 12, 128, 64, 128, 3), (3, 24, 24, 64, 0, 128, 500), (1, 5, 96, 64, 64, 64, 4), (2, 6, 12, 64, 0, 64, 2)])
def test_conv3x3_weight_gradient_nine_taps(gpu, nimg, H, W, C1, C2, Cout, nsplit):
    """vmm_conv3x3_wgrad_f32 (all nine taps of a 64 x 64 channel block from one LDS patch, exact fp32 MFMA) against torch autograd's weight and
    bias gradients of the same 3 x 3 'same' convolution: image borders, two concatenated sources, several rows per segment (W = 12, 24),
    two segments per row (W = 96), more slices than segments."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(60 + W)
    Cin = C1 + C2
    x = torch.randn(nimg, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    dy = torch.randn(nimg, Cout, H, W, generator=g)
    F.conv2d(x, w, b, padding=1).backward(dy)
    want_w = w.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout)
    xr = x.permute(0, 2, 3, 1).reshape(-1, Cin)
    x1g = xr[:, :C1].contiguous().to(gpu)
    x2g = xr[:, C1:].contiguous().to(gpu) if C2 else None
    dyg = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = x1g.data_ptr(), C1, C1
    if C2:
        d.a2, d.C2, d.lda2 = x2g.data_ptr(), C2, C2
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout = H, W, 1, Cout
    dw = torch.zeros(9 * Cin, Cout, device=gpu)
    db = torch.zeros(Cout, device=gpu)
    scratch = torch.full((nsplit, Cout), float("nan"), device=gpu)
    rc = lib.vmm_conv3x3_wgrad_f32(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), nsplit, db.data_ptr(), scratch.data_ptr(), _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), want_w) < 3e-6
    assert relerr(db.cpu(), b.grad) < 3e-6


@pytest.mark.parametrize("nimg,H,W,C1,C2,Cout", [(2, 48, 48, 64, 0, 64), (1, 12, 12, 128, 64, 128), (3, 24, 24, 64, 0, 128), (1, 5, 96, 64, 64, 64),
                                                 (2, 6, 12, 64, 0, 64), (5, 96, 96, 64, 0, 64), (2, 7, 20, 64, 0, 64), (3, 3, 8, 64, 64, 64),
                                                 (44, 12, 12, 128, 0, 64)])
def test_conv3x3_weight_gradient_nine_taps_bf16x3(gpu, nimg, H, W, C1, C2, Cout):
    """vmm_conv3x3_wgrad_bf16x3 (position-space ring of x fragments, the horizontal taps as register shifts of the dY fragment, split-bf16
    MFMA) against torch autograd's weight and bias gradients of the same 3 x 3 'same' convolution: frame borders in both directions, widths
    that are not a multiple of eight (zero columns in position space), frames shorter than the look-ahead, two concatenated sources, more
    workgroups than chunks, many frames per workgroup."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(160 + W + nimg)
    Cin = C1 + C2
    x = torch.randn(nimg, Cin, H, W, generator=g)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    b = torch.zeros(Cout, requires_grad=True)
    dy = torch.randn(nimg, Cout, H, W, generator=g)
    F.conv2d(x, w, b, padding=1).backward(dy)
    want_w = w.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout)
    xr = x.permute(0, 2, 3, 1).reshape(-1, Cin)
    x1g = xr[:, :C1].contiguous().to(gpu)
    x2g = xr[:, C1:].contiguous().to(gpu) if C2 else None
    dyg = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = x1g.data_ptr(), C1, C1
    if C2:
        d.a2, d.C2, d.lda2 = x2g.data_ptr(), C2, C2
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout = H, W, 1, Cout
    n_ws = int(lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), Cout))
    assert n_ws > 0
    results = []
    for use_ws in (True, True, False):  # partial blocks + fixed-order reduction (twice: bit-reproducible), then the atomics path
        dw = torch.zeros(9 * Cin, Cout, device=gpu)
        db = torch.zeros(Cout, device=gpu)
        ws = torch.full((n_ws,), float("nan"), device=gpu) if use_ws else None
        rc = lib.vmm_conv3x3_wgrad_bf16x3(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr() if use_ws else None, _s())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert relerr(dw.cpu(), want_w) < 5e-5
        assert relerr(db.cpu(), b.grad) < 3e-6
        for t in range(9):  # every tap on its own (a wrong shift at a border hides in the norm of the whole tensor)
            assert relerr(dw[t * Cin:(t + 1) * Cin].cpu(), want_w[t * Cin:(t + 1) * Cin]) < 1e-4, t
        results.append((dw, db))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1])
    # through the generic entry point too (forwards here), and a periodic layer is declined
    dw2 = torch.zeros_like(dw)
    assert lib.vmm_conv_wgrad_bf16x3(C.byref(d), dyg.data_ptr(), Cout, dw2.data_ptr(), 4, None, None, _s()) == 0
    torch.cuda.synchronize()
    assert relerr(dw2.cpu(), want_w) < 5e-5
    d.wrap_w = 1
    assert lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), Cout) == 0
    assert lib.vmm_conv3x3_wgrad_bf16x3(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), None, None, _s()) == 1


def test_weight_gradient_scatter_batched(gpu):
    """vmm_pack_weights direction 1: packed [(th, tw, c)][n] weight gradients -> torch layout (N, C, TH, TW), = or +=, several jobs per launch.  The plain
    conv / linear layouts leave through the LDS-tiled kernel, everything else (N not a multiple of 64, a strided tap subset) element-wise; both paths
    in one table."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(77)
    specs = [(3, 3, 32, 64, 1, None), (1, 1, 64, 128, 0, None), (4, 4, 16, 64, 1, None), (3, 3, 128, 192, 1, None), (1, 1, 32, 48, 1, None),
             (2, 2, 16, 64, 1, (1, 2, 0, 2))]  # last: the (h0, hs, w0, ws) tap subset of a 4 x 4 kernel (transposed-convolution phase)
    jobs = (N.PackJob * len(specs))()
    keep, want = [], []
    mx = 0
    for j, (th, tw, c, n, acc, sub) in zip(jobs, specs):
        kh, kw = (4, 4) if sub else (th, tw)
        torch_w = torch.randn(n, c, kh, kw, generator=g).to(gpu)
        packed = torch.randn(th * tw * c, n, generator=g).to(gpu)
        ref = torch_w.clone().cpu() if acc else torch.zeros(n, c, kh, kw)
        upd = packed.cpu().reshape(th, tw, c, n).permute(3, 2, 0, 1)
        if sub:
            h0, hs, w0, ws = sub
            ref = torch_w.clone().cpu()
            if acc:
                ref[:, :, h0::hs, w0::ws] += upd
            else:
                ref[:, :, h0::hs, w0::ws] = upd
        else:
            h0, hs, w0, ws = 0, 1, 0, 1
            ref = ref + upd
        j.torch_w, j.packed = torch_w.data_ptr(), packed.data_ptr()
        j.TH, j.TW, j.C, j.Cp, j.N = th, tw, c, c, n
        j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = c * kh * kw, kh * kw, kw, 1, h0, hs, w0, ws, acc, 0
        keep.append((torch_w, packed))
        want.append(ref)
        mx = max(mx, th * tw * c * n)
    tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), len(specs), mx, 1, _s()), "unpack")
    torch.cuda.synchronize()
    for (tw_, _), ref, spec in zip(keep, want, specs):
        assert torch.equal(tw_.cpu(), ref), spec


@pytest.mark.parametrize("B,T,HW,ntok,with_bias", [(2, 6, 20, 6, True), (3, 4, 33, 9, False), (1, 11, 64, 11, True), (2, 3, 7, 32, False), (2, 5, 16, 1, False),
                                                   (2, 2, 40, 51, False), (1, 3, 9, 64, False)])  # (51: the stress-strain signal as GRU tokens; 64: the kernel's limit)
@pytest.mark.parametrize("heads,dh", [(8, 32), (4, 32), (3, 32), (8, 64), (2, 16), (5, 24), (1, 128)])
def test_cross_attention_kernels(gpu, B, T, HW, ntok, with_bias, heads, dh):
    """cond_attention = 'cross-attention' (vddp.py:354-363, 476-485): softmax attention of every (row, head) over the sample's conditioning tokens
    (+ the relative-position bias on the temporal sites, tokens == frames), and the linear-attention context of the tokens alone followed by
    the unchanged apply pass; both against the einsum restatement of the reference lines."""
    N, lib = _lib()
    if 8 * ntok * heads * (dh + 1) > 160 * 1024:
        pytest.skip("the sample's token keys / values do not fit the CU's 160 KB of LDS at this head width (the entry point returns -1)")
    hid = heads * dh  # (dh: the temporal sites follow attn_dim_head, vddp.py:615; the linear flavour below is always 32 wide)
    g = torch.Generator().manual_seed(31 + ntok)
    rows = B * T * HW
    q = torch.randn(rows, hid, generator=g)
    ek = torch.randn(B, ntok, hid, generator=g)
    ev = torch.randn(B, ntok, hid, generator=g)
    bias = torch.randn(heads, T, T, generator=g) if with_bias else None
    qg, ekg, evg = q.to(gpu), ek.to(gpu), ev.to(gpu)
    out = torch.full((rows, hid), 7.0, device=gpu)
    bg = bias.to(gpu) if with_bias else None
    rc = lib.vmm_cross_attention(qg.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), ntok, bg.data_ptr() if with_bias else None, out.data_ptr(), hid, B, T, HW,
                                 heads, dh, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    q5 = q.reshape(B, T, HW, heads, dh)
    k5, v5 = ek.reshape(B, ntok, heads, dh), ev.reshape(B, ntok, heads, dh)
    sim = torch.einsum("btphd,bjhd->btphj", q5, k5)
    if with_bias:
        sim = sim + bias.permute(1, 0, 2)[None, :, None]  # bias[h][t][j] -> (1, t, 1, h, j)
    want = torch.einsum("btphj,bjhd->btphd", sim.softmax(-1), v5).reshape(rows, hid)
    assert relerr(out.cpu(), want) < 3e-6
    if with_bias:  # a bias that cannot broadcast is rejected like the reference's addition would fail
        assert lib.vmm_cross_attention(qg.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), ntok, bg.data_ptr(), out.data_ptr(), hid, B, T + 1, HW, heads, dh, _s()) == -1
    if dh != 32:
        return
    # linear flavour: q.softmax(d) * scale, k.softmax(tokens), v / HW, ctx = k v^T, out = ctx^T q
    ctx = torch.empty(B * T * heads, dh, dh, device=gpu)
    assert lib.vmm_linattn_cross_context(ekg.data_ptr(), evg.data_ptr(), ntok, B, T, HW, heads, dh, ctx.data_ptr(), None, _s()) == 0
    o2 = torch.full((rows, hid), 7.0, device=gpu)
    assert lib.vmm_linattn_apply(qg.data_ptr(), hid, ctx.data_ptr(), o2.data_ptr(), hid, B, T, HW, heads, dh, _s()) == 0
    torch.cuda.synchronize()
    ks = k5.softmax(dim=1)
    cw = torch.einsum("bjhd,bjhe->bhde", ks, v5 / HW)
    assert relerr(ctx.cpu().reshape(B, T, heads, dh, dh), cw[:, None].expand(B, T, heads, dh, dh)) < 3e-6
    qs = q5.softmax(-1) * dh ** -0.5
    want2 = torch.einsum("bhde,btphd->btphe", cw, qs).reshape(rows, hid)
    assert relerr(o2.cpu(), want2) < 3e-6


@pytest.mark.parametrize("B,T,HW,ntok,with_bias,rotate", [(2, 6, 20, 6, True, True), (3, 4, 33, 9, False, False), (1, 11, 150, 11, True, True),
                                                         (2, 5, 3400, 16, False, False), (2, 5, 16, 1, False, False), (2, 2, 40, 51, False, False), (1, 3, 9, 64, False, True)])
@pytest.mark.parametrize("heads,dh", [(8, 32), (4, 32), (3, 32), (8, 64), (2, 16), (5, 24), (1, 128)])
def test_cross_attention_backward_kernels(gpu, B, T, HW, ntok, with_bias, rotate, heads, dh):
    """Backward of the two cross-attention cores against torch autograd through the einsum restatement of vddp.py:354-363 / 476-485, including what
    the projection epilogue did to q (scale, rotary rotation by the row's frame): dq is the gradient of the RAW to_q output; token and bias
    gradients are ADDED to what the buffers hold."""
    N, lib = _lib()
    from videometamaterials_amd import hostmath
    if 8 * ntok * heads * (dh + 1) > 160 * 1024:
        pytest.skip("the sample's token keys / values do not fit the CU's 160 KB of LDS at this head width (the entry point returns -1)")
    if (heads, dh) != (8, 32) and HW > 1000:
        pytest.skip("the two-pass form: the small shapes")
    hid = heads * dh
    scale = dh ** -0.5
    g = torch.Generator().manual_seed(77 + ntok)
    rows = B * T * HW
    q_raw = torch.randn(rows, hid, generator=g, requires_grad=True)
    ek = torch.randn(B, ntok, hid, generator=g, requires_grad=True)
    ev = torch.randn(B, ntok, hid, generator=g, requires_grad=True)
    bias = torch.randn(heads, T, T, generator=g, requires_grad=True) if with_bias else None
    go = torch.randn(rows, hid, generator=g)
    rot = hostmath.rotary_table(T, dh)  # [T][dh/2][2] (cos, sin)
    q5 = q_raw.reshape(B, T, HW, heads, dh) * scale
    if rotate:
        c, s_ = rot[:, :, 0][None, :, None, None, :], rot[:, :, 1][None, :, None, None, :]
        x0, x1 = q5[..., 0::2], q5[..., 1::2]
        q5 = torch.stack([x0 * c - x1 * s_, x1 * c + x0 * s_], -1).reshape(B, T, HW, heads, dh)
    k5, v5 = ek.reshape(B, ntok, heads, dh), ev.reshape(B, ntok, heads, dh)
    sim = torch.einsum("btphd,bjhd->btphj", q5, k5)
    if with_bias:
        sim = sim + bias.permute(1, 0, 2)[None, :, None]
    out = torch.einsum("btphj,bjhd->btphd", sim.softmax(-1), v5).reshape(rows, hid)
    out.backward(go)
    q_used = q5.detach().reshape(rows, hid).contiguous().to(gpu)
    ekg, evg, gog = ek.detach().to(gpu), ev.detach().to(gpu), go.to(gpu)
    bg = bias.detach().to(gpu) if with_bias else None
    rotg = rot.to(gpu).contiguous()
    dq = torch.full((rows, hid), 7.0, device=gpu)
    base = 0.5
    dek, dev_ = torch.full((B, ntok, hid), base, device=gpu), torch.full((B, ntok, hid), base, device=gpu)
    dbias = torch.full((heads, T, T), base, device=gpu) if with_bias else None
    nsc = int(lib.vmm_cross_attention_bwd_scratch(B, T, HW, heads, dh, ntok))
    assert (nsc == 0) == ((heads, dh) == (8, 32) and ntok <= 16)  # (the one-pass kernel: 8 heads of 32, at most 16 tokens)
    sc = torch.full((max(nsc, 1),), float("nan"), device=gpu)
    rc = lib.vmm_cross_attention_bwd(q_used.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), ntok, bg.data_ptr() if with_bias else None, gog.data_ptr(), hid,
                                     rotg.data_ptr() if rotate else None, scale, dq.data_ptr(), hid, dek.data_ptr(), dev_.data_ptr(),
                                     dbias.data_ptr() if with_bias else None, sc.data_ptr() if nsc else None, B, T, HW, heads, dh, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    if ntok == 1:  # one key: p = 1, ds = 0 -- the scores do not matter
        assert float(dq.abs().max()) == 0 and float((dek - base).abs().max()) == 0 and float(q_raw.grad.abs().max()) == 0
    else:
        assert relerr(dq.cpu(), q_raw.grad) < 2e-5 and relerr(dek.cpu() - base, ek.grad) < 2e-5
    assert relerr(dev_.cpu() - base, ev.grad) < 2e-5
    if with_bias:
        assert relerr(dbias.cpu() - base, bias.grad) < 2e-5
    assert lib.vmm_cross_attention_bwd(q_used.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), 65, None, gog.data_ptr(), hid, None, scale, dq.data_ptr(), hid,
                                       dek.data_ptr(), dev_.data_ptr(), None, sc.data_ptr(), B, T, HW, heads, dh, _s()) == -1  # more than 64 tokens
    assert lib.vmm_cross_attention_bwd(q_used.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), 17, None, gog.data_ptr(), hid, None, scale, dq.data_ptr(), hid,
                                       dek.data_ptr(), dev_.data_ptr(), None, None, B, T, HW, heads, dh, _s()) == -1  # the two-pass form without its scratch
    if dh != 32:
        return

    # linear flavour: forward context + statistics from the library, backward against autograd
    q2 = torch.randn(rows, hid, generator=g, requires_grad=True)
    ek2, ev2 = ek.detach().clone().requires_grad_(True), ev.detach().clone().requires_grad_(True)
    ks = ek2.reshape(B, ntok, heads, dh).softmax(dim=1)
    cw = torch.einsum("bjhd,bjhe->bhde", ks, ev2.reshape(B, ntok, heads, dh) / HW)
    qs = q2.reshape(B, T, HW, heads, dh).softmax(-1) * scale
    o2 = torch.einsum("bhde,btphd->btphe", cw, qs).reshape(rows, hid)
    o2.backward(go)
    ctx = torch.empty(B * T * heads, dh, dh, device=gpu)
    kstat = torch.empty(B * T * heads, 2, dh, device=gpu)
    dctx = torch.empty_like(ctx)
    q2g = q2.detach().to(gpu)
    assert lib.vmm_linattn_cross_context(ekg.data_ptr(), evg.data_ptr(), ntok, B, T, HW, heads, dh, ctx.data_ptr(), kstat.data_ptr(), _s()) == 0
    dq2 = torch.full((rows, hid), 7.0, device=gpu)
    base = 0.0  # (these gradients carry a 1 / HW: next to an offset of 0.5 the comparison would measure the offset's rounding)
    dek2, dev2 = torch.full((B, ntok, hid), base, device=gpu), torch.full((B, ntok, hid), base, device=gpu)
    rc = lib.vmm_linattn_cross_bwd(q2g.data_ptr(), hid, ekg.data_ptr(), evg.data_ptr(), ntok, ctx.data_ptr(), kstat.data_ptr(), gog.data_ptr(), hid,
                                   dctx.data_ptr(), dq2.data_ptr(), hid, dek2.data_ptr(), dev2.data_ptr(), B, T, HW, heads, dh, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    if ntok == 1:  # one key: softmax over the tokens is 1, every context row equals v / HW, sum_d softmax_d(q) = 1 -- q and k do not matter
        assert float(dq2.abs().max()) < 1e-6 * float(ev2.grad.abs().max()) and float(dek2.abs().max()) < 1e-6 * float(ev2.grad.abs().max())
    else:
        assert relerr(dq2.cpu(), q2.grad) < 2e-5 and relerr(dek2.cpu() - base, ek2.grad) < 5e-5
    assert relerr(dev2.cpu() - base, ev2.grad) < 5e-5


@pytest.mark.parametrize("B,L,H,I", [(2, 34, 64, 1), (3, 7, 300, 300), (1, 12, 512, 512), (2, 5, 256, 1)])
def test_gru_recurrence_against_torch_gru(gpu, B, L, H, I):
    """vmm_gru_recurrent / _bwd (cond_att_GRU, vddp.py:546-549: nn.GRU) against torch.nn.GRU itself, one layer: states, and through autograd the
    gradients of the pre-activations' producers (input, W_ih, W_hh, both biases) assembled the way the plan does -- input side as a dense product,
    recurrence in the kernel, weight gradients from dgi / dgh."""
    N, lib = _lib()
    torch.manual_seed(5 + H)
    gru = torch.nn.GRU(I, H, num_layers=1, batch_first=True)
    x = torch.randn(B, L, I, requires_grad=True)
    dy = torch.randn(B, L, H)
    y_ref, _ = gru(x)
    y_ref.backward(dy)
    w_ih, w_hh, b_ih, b_hh = (p.detach() for p in (gru.weight_ih_l0, gru.weight_hh_l0, gru.bias_ih_l0, gru.bias_hh_l0))
    gi = (x.detach() @ w_ih.t() + b_ih).to(gpu).contiguous()
    whh_t = w_hh.t().contiguous().to(gpu)
    bhh, whh = b_hh.to(gpu), w_hh.contiguous().to(gpu)
    y = torch.empty(B, L, H, device=gpu)
    hprev, gates = torch.empty(B, L, H, device=gpu), torch.empty(B, L, 4, H, device=gpu)
    assert lib.vmm_gru_recurrent(gi.data_ptr(), whh_t.data_ptr(), bhh.data_ptr(), y.data_ptr(), hprev.data_ptr(), gates.data_ptr(), B, L, H, _s()) == 0
    torch.cuda.synchronize()
    assert relerr(y.cpu(), y_ref.detach()) < 2e-6
    assert torch.equal(hprev[:, 1:].cpu(), y[:, :-1].cpu()) and float(hprev[:, 0].abs().max()) == 0
    dgi, dgh = torch.empty(B, L, 3 * H, device=gpu), torch.empty(B, L, 3 * H, device=gpu)
    dyg = dy.to(gpu)
    assert lib.vmm_gru_recurrent_bwd(dyg.data_ptr(), gates.data_ptr(), hprev.data_ptr(), whh.data_ptr(), dgi.data_ptr(), dgh.data_ptr(), B, L, H, _s()) == 0
    torch.cuda.synchronize()
    dgi_c, dgh_c, hp_c = dgi.cpu().reshape(B * L, 3 * H), dgh.cpu().reshape(B * L, 3 * H), hprev.cpu().reshape(B * L, H)
    assert relerr(dgi_c.t() @ x.detach().reshape(B * L, I), gru.weight_ih_l0.grad) < 2e-5
    assert relerr(dgh_c.t() @ hp_c, gru.weight_hh_l0.grad) < 2e-5
    assert relerr(dgi_c.sum(0), gru.bias_ih_l0.grad) < 2e-5 and relerr(dgh_c.sum(0), gru.bias_hh_l0.grad) < 2e-5
    assert relerr((dgi_c @ w_ih).reshape(B, L, I), x.grad) < 2e-5
    # token select: forward and backward
    D, Nn = 32, 5
    g0 = torch.randn(B, Nn, D)
    null = torch.randn(Nn, D)
    mask = torch.tensor([b % 2 for b in range(B)], dtype=torch.uint8)
    out = torch.empty(B, Nn, D, device=gpu)
    g0g, nullg, maskg = g0.to(gpu), null.to(gpu), mask.to(gpu)
    assert lib.vmm_tokens_select(g0g.data_ptr(), nullg.data_ptr(), maskg.data_ptr(), B, Nn, D, out.data_ptr(), _s()) == 0
    want = torch.where(mask.bool()[:, None, None], null[None], g0)
    dtok = torch.randn(B, Nn, D)
    dg, dnull = torch.full((B, Nn, D), 7.0, device=gpu), torch.full((Nn, D), 0.25, device=gpu)
    dtokg = dtok.to(gpu)
    assert lib.vmm_tokens_select_bwd(dtokg.data_ptr(), maskg.data_ptr(), B, Nn, D, dg.data_ptr(), dnull.data_ptr(), _s()) == 0
    torch.cuda.synchronize()
    assert torch.equal(out.cpu(), want)
    assert torch.equal(dg.cpu(), torch.where(mask.bool()[:, None, None], torch.zeros(()), dtok))
    assert relerr(dnull.cpu() - 0.25, (dtok * mask.float()[:, None, None]).sum(0)) < 1e-5 or float((dtok * mask.float()[:, None, None]).abs().max()) == 0


@pytest.mark.parametrize("B,T,H,W,Cin,Cout", [(2, 3, 24, 24, 64, 128), (3, 2, 6, 12, 128, 64), (1, 5, 16, 96, 64, 64)])
def test_conv3x3_weight_gradient_fused_operand_bf16x3(gpu, B, T, H, W, Cin, Cout):
    """a_mode 1 of the nine-tap split-bf16 weight-gradient kernel: the layer's input is silu(h * ga[b, c] + gb[b, c]) -- GroupNorm * FiLM -> SiLU of the
    producing block with per-(sample, channel) coefficients (vddp.py:279-285) -- formed in the loader; against torch autograd on the
    materialised operand.  Samples = runs of T frames; zero padding stays zero (it is applied after the activation)."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(400 + W)
    nimg = B * T
    h = torch.randn(nimg, Cin, H, W, generator=g)
    coef = torch.stack([1 + 0.3 * torch.randn(B, Cin, generator=g), 0.3 * torch.randn(B, Cin, generator=g)], -1)  # (B, C, 2) = (ga, gb)
    ga = coef[:, :, 0].repeat_interleave(T, 0)[:, :, None, None]
    gb = coef[:, :, 1].repeat_interleave(T, 0)[:, :, None, None]
    x = F.silu(h * ga + gb)
    w = (torch.randn(Cout, Cin, 3, 3, generator=g) / math.sqrt(Cin * 9)).requires_grad_(True)
    dy = torch.randn(nimg, Cout, H, W, generator=g)
    F.conv2d(x, w, None, padding=1).backward(dy)
    want_w = w.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout)
    hg = h.permute(0, 2, 3, 1).reshape(-1, Cin).contiguous().to(gpu)
    dyg = dy.permute(0, 2, 3, 1).reshape(-1, Cout).contiguous().to(gpu)
    cg = coef.contiguous().to(gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = hg.data_ptr(), Cin, Cin
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
    d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout = H, W, 1, Cout
    d.a_mode, d.a_coef, d.a_imgs_per_sample = 1, cg.data_ptr(), T
    n_ws = int(lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), Cout))
    assert n_ws > 0
    ws = torch.empty(n_ws, device=gpu)
    dw = torch.zeros(9 * Cin, Cout, device=gpu)
    assert lib.vmm_conv3x3_wgrad_bf16x3(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), None, ws.data_ptr(), _s()) == 0
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), want_w) < 5e-5
    d.a_coef = None
    assert lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), Cout) == 0  # a fused transform without coefficients is refused


@pytest.mark.parametrize("rows,C1,C2,Cout,bias", [(5000, 64, 0, 768, False), (777, 256, 0, 64, True), (64, 128, 128, 128, True), (20000, 128, 0, 768, False),
                                                  (3, 512, 0, 192, True), (130000, 64, 64, 64, False)])
def test_conv1x1_weight_gradient_bf16x3(gpu, rows, C1, C2, Cout, bias):
    """vmm_conv1x1_wgrad_bf16x3 (128 x 128 channel blocks, [channel][row] fragment images, partial blocks + fixed-order reduction) against x^T dY and
    the column sums of dY: channel counts that fill half a block, two concatenated sources, row counts that are no multiple of the 64-row chunk,
    more workgroups than chunks; += semantics and bit-reproducibility."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(rows + Cout)
    Cin = C1 + C2
    x = torch.randn(rows, Cin, generator=g)
    dy = torch.randn(rows, Cout, generator=g)
    want_w = x.double().t() @ dy.double()
    want_b = dy.double().sum(0)
    x1 = x[:, :C1].contiguous().to(gpu)
    x2 = x[:, C1:].contiguous().to(gpu) if C2 else None
    dyg = dy.to(gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = x1.data_ptr(), C1, C1
    if C2:
        d.a2, d.C2, d.lda2 = x2.data_ptr(), C2, C2
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = 1, 1, rows, 1, rows, 1
    d.KH, d.KW, d.sgn_h, d.sgn_w, d.Hout, d.Wout, d.oscale, d.Cout = 1, 1, 1, 1, 1, rows, 1, Cout
    n_ws = int(lib.vmm_conv1x1_wgrad_bf16x3_workspace(C.byref(d), Cout))
    assert n_ws > 0
    outs = []
    for _ in range(2):
        ws = torch.full((n_ws,), float("nan"), device=gpu)
        dw = torch.ones(Cin, Cout, device=gpu)  # += : starts from one
        db = torch.ones(Cout, device=gpu)
        assert lib.vmm_conv1x1_wgrad_bf16x3(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), db.data_ptr() if bias else None, ws.data_ptr(), _s()) == 0
        torch.cuda.synchronize()
        assert relerr(dw.cpu().double() - 1, want_w) < 5e-5
        if bias:
            assert relerr(db.cpu().double() - 1, want_b) < 3e-6
        outs.append(dw)
    assert torch.equal(outs[0], outs[1])
    d.KH = 3
    assert lib.vmm_conv1x1_wgrad_bf16x3_workspace(C.byref(d), Cout) == 0


@pytest.mark.parametrize("variant,tol", [("bf16x3", 5e-5), ("fp16", 2e-3), ("bf16", 1.5e-2)])
@pytest.mark.parametrize("geom", ["down", "down_two_sources", "up", "stem7", "circ3", "down_small"])
def test_conv_weight_gradient_any_geometry_on_the_matrix_cores(gpu, geom, variant, tol):
    """vmm_conv_wgrad_tap_*: the weight (and bias) gradient of whatever convolution the descriptor states, against torch autograd in fp64 -- the (1,4,4) stride-2
    Downsample (vddp.py:139-151), the four 2 x 2 phases of the transposed Upsample (their packed slots are the taps kh = 1 - ph + 2 kh'), the 7 x 7 stem over rows
    padded to four channels, a periodic 3 x 3 layer (the nine-tap kernel declines those); += semantics, partial last chunk, more workgroups than chunks."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(11)
    fn = getattr(lib, "vmm_conv_wgrad_tap_" + variant)

    def rows(t):  # (n, c, h, w) -> [(n, h, w)][c]
        return t.permute(0, 2, 3, 1).reshape(-1, t.shape[1]).contiguous()

    def run(d, xs, dy, K, Cout, want_w, want_b):
        x1 = rows(xs[0]).float().to(gpu)
        x2 = rows(xs[1]).float().to(gpu) if len(xs) > 1 else None
        dyg = rows(dy).float().to(gpu)
        d.a1, d.C1, d.lda1 = x1.data_ptr(), x1.shape[1], x1.shape[1]
        if x2 is not None:
            d.a2, d.C2, d.lda2 = x2.data_ptr(), x2.shape[1], x2.shape[1]
        d.Cout = Cout
        n_ws = int(lib.vmm_conv_wgrad_tap_workspace(C.byref(d), Cout))
        assert n_ws > 0
        outs = []
        for _ in range(2):
            ws = torch.full((n_ws,), float("nan"), device=gpu)
            dw = torch.full((K, Cout), 0.5, device=gpu)
            db = torch.full((Cout,), 0.25, device=gpu)
            assert fn(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), _s()) == 0
            torch.cuda.synchronize()
            assert relerr(dw.cpu().double() - 0.5, want_w) < tol, geom
            assert relerr(db.cpu().double() - 0.25, want_b) < max(tol / 10, 3e-6), geom
            outs.append(dw)
        assert torch.equal(outs[0], outs[1])

    if geom in ("down", "down_two_sources", "down_small"):
        nimg, H, W, C1, C2, Cout = {"down": (3, 16, 24, 64, 0, 64), "down_two_sources": (2, 8, 8, 64, 64, 128), "down_small": (1, 4, 4, 8, 0, 12)}[geom]
        Cin = C1 + C2
        x = torch.randn(nimg, Cin, H, W, generator=g).double()
        w = (torch.randn(Cout, Cin, 4, 4, generator=g) / 16).double().requires_grad_()
        b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w, b, stride=2, padding=1)
        dy = torch.randn(y.shape, generator=g).double()
        y.backward(dy)
        d = N.ConvDesc()
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H // 2, W // 2, 2
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 4, 4, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale = H // 2, W // 2, 1
        run(d, [x[:, :C1]] + ([x[:, C1:]] if C2 else []), dy, 16 * Cin, Cout, w.grad.permute(2, 3, 1, 0).reshape(16 * Cin, Cout), b.grad)
    elif geom == "up":
        nimg, H, W, Cin, Cout = 2, 8, 12, 64, 64
        x = torch.randn(nimg, Cin, H, W, generator=g).double()
        w = (torch.randn(Cin, Cout, 4, 4, generator=g) / 16).double().requires_grad_()
        b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = F.conv_transpose2d(x, w, b, stride=2, padding=1)
        dy = torch.randn(y.shape, generator=g).double()
        y.backward(dy)
        for ph in range(2):
            for pw in range(2):
                d = N.ConvDesc()
                d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
                d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 2, 2, ph, pw, -1, -1
                d.Hout, d.Wout, d.oscale, d.ooh, d.oow = 2 * H, 2 * W, 2, ph, pw
                want = w.grad[:, :, 1 - ph::2, 1 - pw::2].permute(2, 3, 0, 1).reshape(4 * Cin, Cout)  # [(kh', kw', ci)][co], kh = 1 - ph + 2 kh'
                want_b = dy[:, :, ph::2, pw::2].sum((0, 2, 3))  # the bias gradient of the rows this phase writes
                run(d, [x], dy, 4 * Cin, Cout, want, want_b)
    elif geom == "stem7":
        nimg, H, W, Cout = 3, 16, 16, 64
        x = torch.randn(nimg, 3, H, W, generator=g).double()
        w = (torch.randn(Cout, 3, 7, 7, generator=g) / 12).double().requires_grad_()
        b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(x, w, b, padding=3)
        dy = torch.randn(y.shape, generator=g).double()
        y.backward(dy)
        xp = torch.cat([x, torch.zeros(nimg, 1, H, W, dtype=torch.float64)], 1)  # rows padded to four channels, as the plan stages the network input
        d = N.ConvDesc()
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 7, 7, -3, -3, 1, 1
        d.Hout, d.Wout, d.oscale = H, W, 1
        want = torch.cat([w.grad, torch.zeros(Cout, 1, 7, 7, dtype=torch.float64)], 1).permute(2, 3, 1, 0).reshape(49 * 4, Cout)
        run(d, [xp], dy, 49 * 4, Cout, want, b.grad)
    else:  # a 3 x 3 layer under periodic padding (vddp.py:163-243)
        nimg, H, W, Cin, Cout = 2, 8, 8, 64, 64
        x = torch.randn(nimg, Cin, H, W, generator=g).double()
        w = (torch.randn(Cout, Cin, 3, 3, generator=g) / 24).double().requires_grad_()
        b = torch.zeros(Cout, dtype=torch.float64, requires_grad=True)
        y = F.conv2d(F.pad(x, (1, 1, 1, 1), mode="circular"), w, b)
        dy = torch.randn(y.shape, generator=g).double()
        y.backward(dy)
        d = N.ConvDesc()
        d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
        d.Hout, d.Wout, d.oscale, d.wrap_h, d.wrap_w = H, W, 1, 1, 1
        run(d, [x], dy, 9 * Cin, Cout, w.grad.permute(2, 3, 1, 0).reshape(9 * Cin, Cout), b.grad)


@pytest.mark.parametrize("variant", ["bf16x3", "fp16"])
def test_deferred_weight_gradient_totals_are_the_per_layer_totals(gpu, variant):
    """vmm_conv_desc.defer_reduce + vmm_reduce_batch: three 3 x 3 layers and three 1 x 1 layers (with / without bias, two sources, several channel blocks)
    leave their partial blocks in their workspaces, ONE launch totals all of them (plus a plain partial-rows job) -- bit for bit what the per-layer second
    stages give, += semantics included."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(41)
    layers = []
    for kind, shape in [(1, (2, 24, 24, 64, 0, 64, True)), (1, (1, 12, 12, 128, 64, 128, False)), (1, (3, 8, 16, 64, 0, 128, True)),
                        (2, (5000, 64, 0, 768, False)), (2, (777, 256, 0, 64, True)), (2, (640, 128, 128, 128, True))]:
        d = N.ConvDesc()
        if kind == 1:
            nimg, H, W, C1, C2, Cout, bias = shape
            rows = nimg * H * W
            d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = nimg, H, W, H, W, 1
            d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = 3, 3, -1, -1, 1, 1
            d.Hout, d.Wout, d.oscale, d.Cout = H, W, 1, Cout
        else:
            rows, C1, C2, Cout, bias = shape
            d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = 1, 1, rows, 1, rows, 1
            d.KH, d.KW, d.sgn_h, d.sgn_w, d.Hout, d.Wout, d.oscale, d.Cout = 1, 1, 1, 1, 1, rows, 1, Cout
        x1 = torch.randn(rows, C1, generator=g).to(gpu)
        x2 = torch.randn(rows, C2, generator=g).to(gpu) if C2 else None
        dy = torch.randn(rows, Cout, generator=g).to(gpu)
        d.a1, d.C1, d.lda1 = x1.data_ptr(), C1, C1
        if C2:
            d.a2, d.C2, d.lda2 = x2.data_ptr(), C2, C2
        stem = "vmm_conv3x3_wgrad_" if kind == 1 else "vmm_conv1x1_wgrad_"
        n_ws = int(getattr(lib, stem + "bf16x3_workspace")(C.byref(d), Cout))
        assert n_ws > 0
        layers.append((kind, d, getattr(lib, stem + variant), getattr(lib, stem + "reduce_job"), (x1, x2, dy), (9 if kind == 1 else 1) * (C1 + C2), Cout, bias, n_ws))

    part = torch.randn(37, 200, generator=g).to(gpu)

    def run(defer):
        outs, keep, jobs = [], [], (N.ReduceJob * (len(layers) + 1))()
        wg = 0
        for i, (kind, d, fn, jobfn, (x1, x2, dy), K, Cout, bias, n_ws) in enumerate(layers):
            dw = torch.full((K, Cout), 0.25, device=gpu)  # += : starts from a quarter
            db = torch.full((Cout,), 0.5, device=gpu)
            ws = torch.full((n_ws,), float("nan"), device=gpu)
            d.defer_reduce = 1 if defer else 0
            assert fn(C.byref(d), dy.data_ptr(), Cout, dw.data_ptr(), db.data_ptr() if bias else None, ws.data_ptr(), _s()) == 0
            if defer:
                assert jobfn(C.byref(d), Cout, dw.data_ptr(), db.data_ptr() if bias else None, ws.data_ptr(), C.byref(jobs[i])) == 0
                assert jobs[i].kind == kind and jobs[i].wgs > 0
                jobs[i].wg0 = wg
                wg += jobs[i].wgs
            d.defer_reduce = 0
            outs += [dw, db]
            keep.append(ws)
        # a plain partial-rows job (kind 0: what vmm_sum_partials does)
        tot = torch.full((150,), 2.0, device=gpu)
        if defer:
            j = jobs[len(layers)]
            j.part, j.out, j.kind, j.nz, j.ld, j.Cout, j.wgs, j.wg0 = part.data_ptr(), tot.data_ptr(), 0, 37, 200, 150, (150 + 15) // 16, wg
            wg += j.wgs
            torch.cuda.synchronize()
            for k in range(0, len(outs), 2):  # nothing was totalled yet
                assert bool((outs[k] == 0.25).all())
            tab = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(gpu)
            assert lib.vmm_reduce_batch(tab.data_ptr(), len(layers) + 1, wg, _s()) == 0
        else:
            assert lib.vmm_sum_partials(part.data_ptr(), 37, 200, 150, tot.data_ptr(), _s()) == 0
        torch.cuda.synchronize()
        return outs + [tot]

    ref, got = run(False), run(True)
    for i, (a, b) in enumerate(zip(ref, got)):
        assert torch.isfinite(a).all() and torch.equal(a, b), i
    assert float((ref[0] - 0.25).abs().max()) > 1.0  # (the layers did produce gradients)


@pytest.mark.parametrize("rows,Cc,Cout", [(3000, 64, 768), (515, 128, 256), (64, 256, 64)])
def test_layernorm_fused_projection_statistics_and_weight_gradient(gpu, rows, Cc, Cout):
    """Training forward of PreNorm(to_qkv): vmm_proj_bf16x3_ln_stats = the LayerNorm-fused projection that also leaves (mean, rstd) per row, and
    vmm_conv1x1_wgrad_bf16x3_ln = the weight gradient that re-normalises x from those statistics while it stages it; against LayerNorm -> Linear
    under torch autograd (vddp.py:245-254: biased variance, eps inside the sqrt, gamma only)."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(rows)
    x = torch.randn(rows, Cc, generator=g) * 2 + 0.5
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    w = (torch.randn(Cout, Cc, generator=g) / math.sqrt(Cc)).requires_grad_(True)
    dy = torch.randn(rows, Cout, generator=g)
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    y = (x - mean) / (var + 1e-5).sqrt() * gamma
    out = y @ w.t()
    out.backward(dy)
    xg, gg, dyg = x.to(gpu), gamma.to(gpu), dy.to(gpu)
    wp = _pack_frag(N, lib, gpu, w.detach(), 2)
    og = torch.full((rows, Cout), 7.0, device=gpu)
    stats = torch.full((rows, 2), float("nan"), device=gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1, d.w, d.out, d.ldo, d.Cout = xg.data_ptr(), Cc, Cc, wp.data_ptr(), og.data_ptr(), Cout, Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = 1, 1, rows, 1, rows, 1
    d.KH, d.KW, d.sgn_h, d.sgn_w, d.Hout, d.Wout, d.oscale, d.rot_dh, d.q_scale = 1, 1, 1, 1, 1, rows, 1, 32, 1.0
    assert lib.vmm_proj_bf16x3_ln_stats(C.byref(d), gg.data_ptr(), C.c_float(1e-5), stats.data_ptr(), _s()) == 0
    torch.cuda.synchronize()
    assert relerr(og.cpu(), out.detach()) < 5e-5
    assert relerr(stats[:, 0].cpu(), mean[:, 0]) < 1e-5 and relerr(stats[:, 1].cpu(), 1 / (var[:, 0] + 1e-5).sqrt()) < 1e-5
    n_ws = int(lib.vmm_conv1x1_wgrad_bf16x3_workspace(C.byref(d), Cout))
    assert n_ws > 0
    ws = torch.empty(n_ws, device=gpu)
    dw = torch.zeros(Cc, Cout, device=gpu)
    assert lib.vmm_conv1x1_wgrad_bf16x3_ln(C.byref(d), dyg.data_ptr(), Cout, dw.data_ptr(), ws.data_ptr(), stats.data_ptr(), gg.data_ptr(), _s()) == 0
    torch.cuda.synchronize()
    assert relerr(dw.cpu(), w.grad.t()) < 5e-5


# (variant, element type of the rows of g, tolerance): the single-pass instances take g in their operand's own 16-bit type (csrc/vmm_common.h VMM_DQKV16)
QKV_BWD_VARIANTS = [("bf16x3", torch.float32, 5e-5), ("fp16", torch.float16, 2e-3), ("bf16", torch.bfloat16, 1.5e-2)]


@pytest.mark.parametrize("variant,gdtype,tol", QKV_BWD_VARIANTS)
@pytest.mark.parametrize("rows,with_ln", [(64, False), (1920, True), (70400, True), (20032, False)])
def test_fused_to_qkv_backward(gpu, rows, with_ln, variant, gdtype, tol):
    """vmm_qkv_bwd_bf16x3 / _fp16 / _bf16: data gradient gy = g W and weight gradient dW += g^T y of to_qkv (768 x 64) from ONE pass over g, y = x or LayerNorm(x)
    re-formed from the forward's row statistics; against the two matrix products in fp64.  One chunk, fewer chunks than workgroups, many chunks per
    workgroup; += semantics of dW, plain store of gy, bit-reproducible.  The single-pass instances read g as rows of 16-bit operands (what the recomputing
    attention backward of the same build stores): the reference multiplies exactly those values, the tolerance is the 16-bit rounding of W and y."""
    N, lib = _lib()
    g_ = torch.Generator().manual_seed(rows)
    Cc, Nq = 64, 768
    x = torch.randn(rows, Cc, generator=g_) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g_)
    w = torch.randn(Nq, Cc, generator=g_) / 8          # torch layout of to_qkv.weight: (out = 768, in = 64)
    g = torch.randn(rows, Nq, generator=g_)
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = 1 / (var + 1e-5).sqrt()
    y = (x - mean) * rstd * gamma if with_ln else x
    want_gy = g.double() @ w.double()
    want_dw = y.double().t() @ g.double()             # packed k-major [c][n]
    g = g.to(gdtype).float()                           # (the values the kernel is handed, exactly)
    want_gy = g.double() @ w.double()
    want_dw = y.double().t() @ g.double()
    wp = _pack_frag(N, lib, gpu, w.t().contiguous(), 2 | (16 if variant == "fp16" else 0))  # the (K = 768, N = 64) operand: "weight (out = 64, in = 768)" = W^T
    xg, gg = x.to(gpu), g.to(gdtype).to(gpu)
    fn = getattr(lib, "vmm_qkv_bwd_" + variant)
    stats = torch.cat([mean, rstd], 1).contiguous().to(gpu)
    gam = gamma.to(gpu)
    n_ws = int(lib.vmm_qkv_bwd_workspace(rows, Cc, Nq))
    assert n_ws > 0 and lib.vmm_qkv_bwd_workspace(rows + 1, Cc, Nq) == 0 and lib.vmm_qkv_bwd_workspace(rows, 128, Nq) == 0
    outs = []
    for _ in range(2):
        ws = torch.full((n_ws,), float("nan"), device=gpu)
        gy = torch.full((rows, Cc), 7.0, device=gpu)
        dw = torch.ones(Cc, Nq, device=gpu)
        rc = fn(xg.data_ptr(), Cc, stats.data_ptr() if with_ln else None, gam.data_ptr() if with_ln else None, gg.data_ptr(), Nq,
                wp.data_ptr(), gy.data_ptr(), Cc, dw.data_ptr(), ws.data_ptr(), rows, Cc, Nq, _s())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert relerr(gy.cpu().double(), want_gy) < tol
        assert relerr(dw.cpu().double() - 1, want_dw) < tol
        outs.append((gy, dw))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


@pytest.mark.parametrize("rows,C1,C2,bias,res", [(5000, 256, 0, True, True), (333, 768, 0, False, True), (64, 64, 64, True, False), (1, 256, 0, False, False),
                                                 (40000, 768, 0, False, False), (257, 128, 128, True, True)])
def test_narrow_projection_streaming(gpu, rows, C1, C2, bias, res):
    """vmm_proj_narrow_bf16x3 (wide contraction -> 64 columns: activations and fmt-2 weights straight into registers, no LDS): to_out K = 256 -> 64
    and the to_qkv data gradient K = 768 -> 64, against x @ W (+ bias + residual, in place); row counts that are no multiple of the 64-row wave
    tile, two concatenated sources."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(rows + C1)
    K = C1 + C2
    x = torch.randn(rows, K, generator=g)
    w = torch.randn(64, K, generator=g) / math.sqrt(K)  # (out, in)
    b = torch.randn(64, generator=g) if bias else None
    r = torch.randn(rows, 64, generator=g) if res else None
    want = x @ w.t() + (b if bias else 0) + (r if res else 0)
    wp = _pack_frag(N, lib, gpu, w, 2)
    x1 = x[:, :C1].contiguous().to(gpu)
    x2 = x[:, C1:].contiguous().to(gpu) if C2 else None
    out = r.clone().to(gpu) if res else torch.full((rows, 64), 7.0, device=gpu)  # the residual is read from the output buffer (in place)
    bg = b.to(gpu) if bias else None
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1 = x1.data_ptr(), C1, C1
    if C2:
        d.a2, d.C2, d.lda2 = x2.data_ptr(), C2, C2
    d.w, d.bias = wp.data_ptr(), bg.data_ptr() if bias else None
    d.out, d.ldo, d.Cout = out.data_ptr(), 64, 64
    if res:
        d.res, d.ldres = out.data_ptr(), 64
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = 1, 1, rows, 1, rows, 1
    d.KH, d.KW, d.sgn_h, d.sgn_w, d.Hout, d.Wout, d.oscale = 1, 1, 1, 1, 1, rows, 1
    assert lib.vmm_proj_narrow_bf16x3(C.byref(d), _s()) == 0
    torch.cuda.synchronize()
    assert relerr(out.cpu(), want) < 5e-5
    if res:  # the ResnetBlock tail on the same kernel: out = silu(res * a + b') + x W (+ bias), per-sample coefficients (vddp.py:311)
        nsmp = 3
        rps = (rows + nsmp - 1) // nsmp
        coef = torch.randn(nsmp, 64, 2, generator=g)
        smp = torch.arange(rows) // rps
        want2 = x @ w.t() + (b if bias else 0) + F.silu(r * coef[smp, :, 0] + coef[smp, :, 1])
        out.copy_(r.to(gpu))
        cg = coef.to(gpu)
        assert lib.vmm_proj_narrow_bf16x3_res_silu(C.byref(d), cg.data_ptr(), rps, _s()) == 0
        torch.cuda.synchronize()
        assert relerr(out.cpu(), want2) < 5e-5
    d.Cout = 128
    assert lib.vmm_proj_narrow_bf16x3(C.byref(d), _s()) == 1  # outside the envelope: nothing launched


@pytest.mark.parametrize("dh,heads", [(32, 2), (16, 4), (64, 2), (24, 2), (8, 3), (128, 1), (48, 1)])
def test_projection_rotary_epilogue(gpu, dh, heads):
    """q*scale then interleaved-pair rotation of q,k by the frame index (vddp.py:449,491-496), generic implicit-GEMM epilogue, for every head width
    of the temporal attentions (attn_dim_head, vddp.py:582, 615): the leading min(32, dh) features of a head rotate (RotaryEmbedding(min(32, dh)),
    vddp.py:612), the rest pass through BIT-exactly (identity pairs in hostmath.rotary_table)."""
    from videometamaterials_amd import hostmath
    N, lib = _lib()
    g = torch.Generator().manual_seed(3)
    B, T, H, W, Cc = 2, 5, 4, 4, 32
    hid = heads * dh
    x = torch.randn(B * T * H * W, Cc, generator=g)
    w = torch.randn(3 * hid, Cc, generator=g) / math.sqrt(Cc)
    rot = hostmath.rotary_table(T, dh)
    assert rot.shape == (T, dh // 2, 2)
    out = run_conv(N, lib, gpu, x.to(gpu), w.t().contiguous().to(gpu), 3 * hid, nimg=B * T, Hin=H, Win=W, Hv=H, Wv=W, rot=rot.to(gpu), rot_T=T,
                   rot_ncols=2 * hid, q_scale=dh ** -0.5, q_ncols=hid, rot_dh=dh)
    qkv = (x @ w.t()).reshape(B, T, H * W, 3, heads, dh)
    q, k, v = qkv[:, :, :, 0] * dh ** -0.5, qkv[:, :, :, 1], qkv[:, :, :, 2]
    from oracle import unet3d_oracle as uo

    def rotate(t):  # the oracle's restatement (pinned by the reference goldens) wants the position on axis -2
        return uo.rotary_rotate(t.permute(0, 2, 3, 1, 4)).permute(0, 3, 1, 2, 4)

    ref = torch.stack((rotate(q), rotate(k), v), dim=3).reshape(B * T * H * W, 3 * hid)
    assert relerr(out.cpu(), ref) < 2e-6
    if dh > 32:  # pass-through features: untouched by the rotation
        got = out.cpu().reshape(B, T, H * W, 3, heads, dh)
        plain = run_conv(N, lib, gpu, x.to(gpu), w.t().contiguous().to(gpu), 3 * hid, nimg=B * T, Hin=H, Win=W, Hv=H, Wv=W, q_scale=dh ** -0.5,
                         q_ncols=hid).cpu().reshape(B, T, H * W, 3, heads, dh)
        assert torch.equal(got[..., 32:], plain[..., 32:])


@pytest.mark.parametrize("C_,G", [(16, 8), (64, 8), (512, 8)])
def test_groupnorm_film_silu(gpu, C_, G):
    N, lib = _lib()
    g = torch.Generator().manual_seed(4)
    B, T, H, W = 2, 3, 12, 12
    x = torch.randn(B, C_, T, H, W, generator=g) * 2 + 0.5
    gamma, beta = torch.randn(C_, generator=g), torch.randn(C_, generator=g)
    film = torch.randn(B, 2 * C_, generator=g)
    res = torch.randn(B * T * H * W, C_, generator=g)
    y = F.group_norm(x, G, gamma, beta, eps=1e-5) * (film[:, :C_, None, None, None] + 1) + film[:, C_:, None, None, None]
    ref = rows_of(F.silu(y)) + res
    xr = rows_of(x).to(gpu)
    rps = T * H * W
    sums = torch.empty(B * G * 2, dtype=torch.float64, device=gpu)
    coef = torch.empty(B, C_, 2, device=gpu)
    stats = torch.empty(B * G * 2, device=gpu)
    gg, bg, fg, rg = gamma.to(gpu), beta.to(gpu), film.to(gpu), res.to(gpu)  # keep the device copies alive across the launches
    N.check(lib.vmm_groupnorm_stats(xr.data_ptr(), C_, B, rps, C_, G, sums.data_ptr(), _s()), "stats")
    N.check(lib.vmm_groupnorm_coef(sums.data_ptr(), rps * (C_ // G), 1e-5, gg.data_ptr(), bg.data_ptr(), fg.data_ptr(),
                                   2 * C_, B, C_, G, coef.data_ptr(), stats.data_ptr(), None, 0, None, 0, _s()), "coef")
    out = torch.empty_like(xr)
    N.check(lib.vmm_affine_silu(xr.data_ptr(), C_, coef.data_ptr(), rg.data_ptr(), C_, out.data_ptr(), C_, xr.shape[0], rps, C_, _s()), "apply")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 5e-6
    xs = x.reshape(B, G, -1)
    assert torch.allclose(stats.cpu().reshape(B, G, 2)[..., 0], xs.mean(-1), atol=1e-5)
    assert torch.allclose(stats.cpu().reshape(B, G, 2)[..., 1], 1 / torch.sqrt(xs.var(-1, unbiased=False) + 1e-5), rtol=1e-5)
    # slot mode: the statistics pass leaves one (sum, sum of squares) pair per workgroup, the coefficient kernel adds them in fixed order
    nslots = int(lib.vmm_groupnorm_stats_slots(B, rps, C_))
    part = torch.full((B * G * nslots * 2,), float("nan"), device=gpu)
    coef3 = torch.full_like(coef, float("nan"))
    for _ in range(2):  # twice: bit-reproducible
        N.check(lib.vmm_groupnorm_stats_partials(xr.data_ptr(), C_, B, rps, C_, G, part.data_ptr(), _s()), "stats slots")
        N.check(lib.vmm_groupnorm_coef(None, rps * (C_ // G), 1e-5, gg.data_ptr(), bg.data_ptr(), fg.data_ptr(), 2 * C_, B, C_, G, coef3.data_ptr(), None,
                                       part.data_ptr(), nslots, None, 0, _s()), "coef from slots")
        torch.cuda.synchronize()
        assert relerr(coef3.cpu(), coef.cpu()) < 1e-6
        again = coef3.clone() if _ == 0 else again
    assert torch.equal(coef3, again)
    if (C_ // G) % 4 == 0:  # direct mode: the coefficient kernel reduces each (sample, group) slice of x itself, no statistics launch
        coef2 = torch.full_like(coef, float("nan"))
        N.check(lib.vmm_groupnorm_coef(None, rps * (C_ // G), 1e-5, gg.data_ptr(), bg.data_ptr(), fg.data_ptr(), 2 * C_, B, C_, G, coef2.data_ptr(), None,
                                       None, 0, xr.data_ptr(), C_, _s()), "coef direct")
        torch.cuda.synchronize()
        assert relerr(coef2.cpu(), coef.cpu()) < 1e-6


@pytest.mark.parametrize("C_,G,B,T,H,W,with_film", [(16, 8, 2, 3, 12, 12, True), (64, 8, 3, 5, 24, 20, True), (64, 8, 2, 3, 12, 12, False),
                                                    (128, 8, 2, 2, 12, 12, True), (512, 8, 2, 3, 4, 4, True), (96, 8, 1, 2, 8, 8, False),
                                                    (1024, 8, 2, 1, 4, 4, True)])
def test_groupnorm_film_silu_backward(gpu, C_, G, B, T, H, W, with_film):
    """vmm_groupnorm_bwd (vddp.py:273-285 under autograd): dh, dgamma, dbeta, dscale / dshift against torch autograd of
    silu(group_norm(h) * (scale + 1) + shift); the per-workgroup partial rows are summed in a fixed order, so dh repeats bit for bit."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(41)
    x = (torch.randn(B, C_, T, H, W, generator=g) * 2 + 0.5).requires_grad_(True)
    gamma = torch.randn(C_, generator=g).requires_grad_(True)
    beta = torch.randn(C_, generator=g).requires_grad_(True)
    film = torch.randn(B, 2 * C_, generator=g).requires_grad_(True)
    y = F.group_norm(x, G, gamma, beta, eps=1e-5)
    if with_film:
        y = y * (film[:, :C_, None, None, None] + 1) + film[:, C_:, None, None, None]
    z = F.silu(y)
    dz = torch.randn(z.shape, generator=g)
    z.backward(dz)
    xr, dzr = rows_of(x.detach()).to(gpu), rows_of(dz).to(gpu)
    rps = T * H * W
    sums = torch.empty(B * G * 2, dtype=torch.float64, device=gpu)
    coef = torch.empty(B, C_, 2, device=gpu)
    stats = torch.empty(B * G * 2, device=gpu)
    gg, bg, fg = gamma.detach().to(gpu), beta.detach().to(gpu), film.detach().to(gpu)
    fptr, ldf = (fg.data_ptr(), 2 * C_) if with_film else (None, 0)
    N.check(lib.vmm_groupnorm_stats(xr.data_ptr(), C_, B, rps, C_, G, sums.data_ptr(), _s()), "stats")
    N.check(lib.vmm_groupnorm_coef(sums.data_ptr(), rps * (C_ // G), 1e-5, gg.data_ptr(), bg.data_ptr(), fptr, ldf, B, C_, G, coef.data_ptr(),
                                   stats.data_ptr(), None, 0, None, 0, _s()), "coef")
    nsc = int(lib.vmm_groupnorm_bwd_scratch(B, rps, C_, G))
    assert nsc >= B * C_ * 2 + B * G * 2
    scratch = torch.full((nsc,), float("nan"), device=gpu)
    outs = []
    for rep in range(2):
        dh = torch.full_like(xr, float("nan"))
        dgam, dbet = torch.zeros(C_, device=gpu), torch.zeros(C_, device=gpu)
        dfilm = torch.full((B, 2 * C_), float("nan"), device=gpu)
        N.check(lib.vmm_groupnorm_bwd(dzr.data_ptr(), C_, xr.data_ptr(), C_, coef.data_ptr(), stats.data_ptr(), gg.data_ptr(), bg.data_ptr(), fptr, ldf,
                                      B, rps, C_, G, scratch.data_ptr(), dh.data_ptr(), C_, 0, dgam.data_ptr(), dbet.data_ptr(),
                                      dfilm.data_ptr() if with_film else None, _s()), "gn bwd")
        torch.cuda.synchronize()
        outs.append(dh)
    assert torch.equal(outs[0], outs[1])
    assert relerr(dh.cpu(), rows_of(x.grad)) < 2e-5
    assert relerr(dgam.cpu(), gamma.grad) < 2e-5
    assert relerr(dbet.cpu(), beta.grad) < 2e-5
    if with_film:
        assert relerr(dfilm.cpu(), film.grad) < 2e-5
    # accumulate = 1: dh += the same gradient
    N.check(lib.vmm_groupnorm_bwd(dzr.data_ptr(), C_, xr.data_ptr(), C_, coef.data_ptr(), stats.data_ptr(), gg.data_ptr(), bg.data_ptr(), fptr, ldf,
                                  B, rps, C_, G, scratch.data_ptr(), dh.data_ptr(), C_, 1, dgam.data_ptr(), dbet.data_ptr(), None, _s()), "gn bwd acc")
    torch.cuda.synchronize()
    assert relerr(dh.cpu(), 2 * rows_of(x.grad)) < 2e-5


@pytest.mark.parametrize("C_", [16, 32, 64, 128, 512])
def test_channel_layernorm(gpu, C_):
    N, lib = _lib()
    g = torch.Generator().manual_seed(5)
    x = torch.randn(1000, C_, generator=g) * 3 + 1
    gamma = torch.randn(C_, generator=g)
    ref = (x - x.mean(1, keepdim=True)) / torch.sqrt(x.var(1, unbiased=False, keepdim=True) + 1e-5) * gamma
    xg, gg, out = x.to(gpu), gamma.to(gpu), torch.empty(1000, C_, device=gpu)
    N.check(lib.vmm_channel_layernorm(xg.data_ptr(), C_, gg.data_ptr(), out.data_ptr(), C_, 1000, C_, 1e-5, _s()), "ln")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 2e-6


def _attn_ref(q, k, v, bias=None):
    sim = torch.einsum("...id,...jd->...ij", q, k)
    if bias is not None:
        sim = sim + bias
    return torch.einsum("...ij,...jd->...id", sim.softmax(-1), v)


@pytest.mark.parametrize("heads,T,ntok,bias_on_cond", [(4, 7, 0, 0), (4, 7, 7, 1), (4, 7, 16, 0),              # thread-per-query kernel
                                                       (8, 11, 11, 1), (8, 16, 16, 0), (8, 5, 0, 0),           # fp32 matrix-core kernel, one frame tile
                                                       (8, 22, 16, 0), (8, 32, 5, 0), (8, 17, 0, 0)])  # two frame tiles
@pytest.mark.parametrize("dh", [32, 16, 64, 24, 8, 128, 100])  # (attn_dim_head of the temporal attentions, vddp.py:582, 615; 32 = every shipped config)
def test_temporal_attention_core(gpu, heads, T, ntok, bias_on_cond, dh):
    N, lib = _lib()
    if dh != 32 and (heads, T) not in ((4, 7), (8, 11), (8, 22)):
        pytest.skip("the other head widths share one kernel: a subset of the shapes")
    g = torch.Generator().manual_seed(6)
    B, HW = 2, 10
    hid = heads * dh
    qkv = torch.randn(B, T, HW, 3, heads, dh, generator=g) * (32 / dh) ** 0.25
    bias = torch.randn(heads, T, T, generator=g)
    q, k, v = (qkv[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))  # b hw h t d
    bfull = bias[None, None]
    ek = ev = None
    if ntok:
        ek, ev = torch.randn(B, ntok, heads, dh, generator=g) * (32 / dh) ** 0.25, torch.randn(B, ntok, heads, dh, generator=g)
        k = torch.cat([ek.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, dh), k], dim=-2)
        v = torch.cat([ev.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, dh), v], dim=-2)
        bfull = torch.cat([bias if bias_on_cond else torch.zeros(heads, T, ntok), bias], dim=-1)[None, None]
    ref = _attn_ref(q, k, v, bfull).permute(0, 3, 1, 2, 4).reshape(B * T * HW, hid)
    sim = torch.einsum("...id,...jd->...ij", q, k) + bfull
    lse_ref = torch.logsumexp(sim, -1).permute(0, 3, 1, 2).reshape(B * T * HW, heads)  # rows (b, t, pix) x heads
    qg = qkv.reshape(B * T * HW, 3 * hid).to(gpu)
    out = torch.full((B * T * HW, hid), float("nan"), device=gpu)
    lse = torch.full((B * T * HW, heads), float("nan"), device=gpu)
    ekg = ek.reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.reshape(B, ntok, hid).to(gpu) if ntok else None
    bg = bias.to(gpu)
    N.check(lib.vmm_temporal_attention(qg.data_ptr(), 3 * hid, ekg.data_ptr() if ntok else None, evg.data_ptr() if ntok else None, ntok,
                                       bg.data_ptr(), bias_on_cond, out.data_ptr(), hid, B, T, HW, heads, dh, lse.data_ptr(), _s()), "temporal")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 5e-6
    assert relerr(lse.cpu(), lse_ref) < 5e-6
    if dh == 32:  # a head width that does not move in 16-byte pieces is refused, not mangled
        assert lib.vmm_temporal_attention(qg.data_ptr(), 3 * hid, None, None, 0, bg.data_ptr(), 0, out.data_ptr(), hid, B, T, HW, heads, 30, None, _s()) == -1


@pytest.mark.parametrize("Cc,T,HW,ntok,bias_on_cond", [(128, 11, 36, 11, 1), (256, 7, 10, 16, 0), (128, 16, 6, 0, 0), (512, 3, 144, 5, 0)])
def test_temporal_core_with_to_out(gpu, Cc, T, HW, ntok, bias_on_cond):
    """vmm_temporal_core_bf16x3: attention over frames (+ stacked tokens, relative-position bias) on the matrix cores, to_out and the
    residual in the same kernel; several pixel splits, several 128-channel head-sum passes, padded frame slots."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(16)
    B, heads, hid = 2, 8, 256
    qkv = torch.randn(B, T, HW, 3, heads, 32, generator=g) * 0.7
    bias = torch.randn(heads, T, T, generator=g)
    wout = torch.randn(Cc, hid, generator=g) / 16
    x = torch.randn(B * T * HW, Cc, generator=g)
    q, k, v = (qkv[:, :, :, i].permute(0, 2, 3, 1, 4) for i in range(3))  # b hw h t d
    bfull = bias[None, None]
    ek = ev = None
    if ntok:
        ek, ev = torch.randn(B, ntok, heads, 32, generator=g), torch.randn(B, ntok, heads, 32, generator=g)
        k = torch.cat([ek.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), k], dim=-2)
        v = torch.cat([ev.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), v], dim=-2)
        bfull = torch.cat([bias if bias_on_cond else torch.zeros(heads, T, ntok), bias], dim=-1)[None, None]
    o = _attn_ref(q, k, v, bfull).permute(0, 3, 1, 2, 4).reshape(B * T * HW, hid)
    branch = o @ wout.t()
    wp = _pack_frag(N, lib, gpu, wout, 3)
    qg, xg, bg = qkv.reshape(B * T * HW, 3 * hid).to(gpu), x.to(gpu), bias.to(gpu)
    ekg = ek.reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.reshape(B, ntok, hid).to(gpu) if ntok else None
    out = torch.full_like(xg, 7.0)
    N.check(lib.vmm_temporal_core_bf16x3(qg.data_ptr(), 3 * hid, xg.data_ptr(), Cc, wp.data_ptr(), ekg.data_ptr() if ntok else None,
                                         evg.data_ptr() if ntok else None, ntok, bg.data_ptr(), bias_on_cond, out.data_ptr(), Cc, B, T, HW, Cc, heads,
                                         _s()), "temporal core + to_out")
    torch.cuda.synchronize()
    assert relerr(out.cpu() - x, branch) < 5e-5  # on the attention branch alone (the residual would mask errors)


@pytest.mark.parametrize("version", [1, 2, 3])
@pytest.mark.parametrize("B,T,HW,ntok,bias_on_cond", [(2, 11, 36, 11, 1), (1, 16, 10, 0, 0), (3, 5, 130, 7, 0), (1, 11, 2304, 16, 0), (2, 1, 4, 3, 0),
                                                      (2, 22, 37, 16, 0), (1, 32, 5, 0, 0), (1, 17, 64, 2, 0)])
def test_fused_temporal_block(gpu, monkeypatch, version, B, T, HW, ntok, bias_on_cond):
    """vmm_temporal_block_bf16x3 = x + to_out(attention over frames(rotary(to_qkv(LayerNorm(x))))) (vddp.py:615,630,680) in one kernel, against the
    same block in fp32 torch: both kernels (lock-step heads / two tiles in flight with the head groups half a tile apart), two pixels or one pixel
    per tile (T <= 16 / T <= 32), several tiles per workgroup, odd tile counts, padded frame slots, tokens with and without the bias."""
    from videometamaterials_amd import hostmath
    N, lib = _lib()
    monkeypatch.setenv("VMM_TB_VERSION", str(min(version, 2)))
    Cc, heads, hid = (128 if version == 3 else 64), 8, 256  # version 3: the C = 128 kernel (weights streamed from L2)
    kind = lib.vmm_temporal_block_supported(T, ntok, HW, Cc, heads)
    if (T > 16 and version == 1) or (version == 3 and (T > 16 or HW % 2)):
        assert kind == 0
        return
    assert kind == version
    g = torch.Generator().manual_seed(23 + T)
    x = torch.randn(B, T, HW, Cc, generator=g)
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    wqkv = torch.randn(3 * hid, Cc, generator=g) / 8
    wout = torch.randn(Cc, hid, generator=g) / 16
    bias = torch.randn(heads, T, T, generator=g)
    rot = hostmath.rotary_table(T, 32)
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    y = (x - mean) / (var + 1e-5).sqrt() * gamma
    qkv = (y @ wqkv.t()).reshape(B, T, HW, 3, heads, 32)
    cos, sin = rot[:, :, 0].repeat_interleave(2, -1)[None, :, None, None], rot[:, :, 1].repeat_interleave(2, -1)[None, :, None, None]

    def rotate(t):
        pr = t.reshape(*t.shape[:-1], 16, 2)
        return t * cos + torch.stack((-pr[..., 1], pr[..., 0]), -1).reshape(t.shape) * sin

    q, k, v = rotate(qkv[:, :, :, 0] * 32 ** -0.5), rotate(qkv[:, :, :, 1]), qkv[:, :, :, 2]
    q, k, v = (t.permute(0, 2, 3, 1, 4) for t in (q, k, v))  # b hw h t d
    bfull = bias[None, None]
    ek = ev = None
    if ntok:
        ek, ev = torch.randn(B, ntok, heads, 32, generator=g), torch.randn(B, ntok, heads, 32, generator=g)
        k = torch.cat([ek.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), k], dim=-2)
        v = torch.cat([ev.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), v], dim=-2)
        bfull = torch.cat([bias if bias_on_cond else torch.zeros(heads, T, ntok), bias], dim=-1)[None, None]
    o = _attn_ref(q, k, v, bfull).permute(0, 3, 1, 2, 4).reshape(B * T * HW, hid)
    branch = o @ wout.t()
    wq, wo = _pack_frag(N, lib, gpu, wqkv, 2), _pack_frag(N, lib, gpu, wout, 3)
    xg, gg, bg, rg = x.reshape(B * T * HW, Cc).to(gpu), gamma.to(gpu), bias.to(gpu), rot.to(gpu)
    ekg = ek.reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.reshape(B, ntok, hid).to(gpu) if ntok else None
    out = torch.full_like(xg, 7.0)
    N.check(lib.vmm_temporal_block_bf16x3(xg.data_ptr(), Cc, gg.data_ptr(), wq.data_ptr(), wo.data_ptr(), ekg.data_ptr() if ntok else None,
                                          evg.data_ptr() if ntok else None, ntok, bg.data_ptr(), bias_on_cond, rg.data_ptr(), out.data_ptr(), Cc,
                                          B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), _s()), "temporal block")
    torch.cuda.synchronize()
    assert relerr(out.cpu() - x.reshape(B * T * HW, Cc), branch) < 5e-5  # on the attention branch alone (the residual would mask errors)
    # the "bf16" throughput mode of the same block: one matrix pass per product on bf16-rounded operands
    out.fill_(7.0)
    N.check(lib.vmm_temporal_block_bf16(xg.data_ptr(), Cc, gg.data_ptr(), wq.data_ptr(), wo.data_ptr(), ekg.data_ptr() if ntok else None,
                                        evg.data_ptr() if ntok else None, ntok, bg.data_ptr(), bias_on_cond, rg.data_ptr(), out.data_ptr(), Cc,
                                        B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), _s()), "temporal block, bf16 mode")
    torch.cuda.synchronize()
    e1 = relerr(out.cpu() - x.reshape(B * T * HW, Cc), branch)
    assert 1e-4 < e1 < 2e-2, e1


@pytest.mark.parametrize("HW,ntok,per_frame", [(144, 5, 1), (16, 6, 0), (300, 0, 0), (144, 16, 0), (576, 11, 0), (33, 1, 0)])
def test_spatial_attention_core(gpu, HW, ntok, per_frame):
    N, lib = _lib()
    g = torch.Generator().manual_seed(7)
    B, T, heads = 2, 5, 2
    hid = heads * 32
    qkv = torch.randn(B, T, HW, 3, heads, 32, generator=g)
    q, k, v = (qkv[:, :, :, i].permute(0, 1, 3, 2, 4) for i in range(3))  # b t h n d
    ek = ev = None
    if ntok:
        ek, ev = torch.randn(B, ntok, heads, 32, generator=g), torch.randn(B, ntok, heads, 32, generator=g)
        if per_frame:
            ekf, evf = ek.permute(0, 1, 2, 3)[:, :, :, None], ev[:, :, :, None]  # b t h 1 d  (token t for frame t)
        else:
            ekf = ek.permute(0, 2, 1, 3)[:, None].expand(B, T, heads, ntok, 32)
            evf = ev.permute(0, 2, 1, 3)[:, None].expand(B, T, heads, ntok, 32)
        k, v = torch.cat([ekf, k], -2), torch.cat([evf, v], -2)
    ref = _attn_ref(q, k, v).permute(0, 1, 3, 2, 4).reshape(B * T * HW, hid)
    out = torch.empty(B * T * HW, hid, device=gpu)
    qg = qkv.reshape(B * T * HW, 3 * hid).to(gpu)
    ekg = ek.reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.reshape(B, ntok, hid).to(gpu) if ntok else None
    N.check(lib.vmm_spatial_attention(qg.data_ptr(), 3 * hid, ekg.data_ptr() if ntok else None, evg.data_ptr() if ntok else None, ntok, per_frame,
                                      out.data_ptr(), hid, B, T, HW, heads, 32, None, _s()), "spatial")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < 5e-6
    # the matrix-core (split-bf16) flash-attention version used for inference: partial last query / key tiles, token chunk, per-frame token
    out2 = torch.full((B * T * HW, hid), 7.0, device=gpu)
    rc = lib.vmm_spatial_attention_bf16x3(qg.data_ptr(), 3 * hid, ekg.data_ptr() if ntok else None, evg.data_ptr() if ntok else None, ntok, per_frame,
                                          out2.data_ptr(), hid, B, T, HW, heads, 32, _s())
    assert rc == 0, rc
    torch.cuda.synchronize()
    assert relerr(out2.cpu(), ref) < 3e-5


def _pack_frag(N, lib, gpu, w2d, fmt):
    """(out, in) weight -> vmm_pack_weights fragment-order operand (fmt 2 / 3)."""
    co, ci = w2d.shape
    wg = w2d.contiguous().to(gpu)
    packed = torch.zeros((co + 31) // 32 * 32 * ((ci + 31) // 32 * 32), device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 1, 1, ci, ci, co
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = ci, 1, 0, 0, 0, 0, 0, 0, 0, fmt
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, packed.numel(), 0, _s()), "pack")
    torch.cuda.synchronize()
    return packed


@pytest.mark.parametrize("B,T,HW,C1,C2,Cout,ln,rot,res", [
    (2, 3, 100, 128, 0, 768, True, True, False),   # temporal to_qkv: fused LayerNorm, q-scale, rotary; partial row tile
    (1, 11, 144, 256, 0, 128, False, False, True),  # to_out: bias + residual, Cout < column chunk
    (2, 2, 96 * 96, 64, 64, 64, False, False, False),  # res_conv on a skip concatenation, many row tiles
    (1, 2, 36, 64, 0, 768, True, False, False),     # few rows: column chunks spread over blockIdx.y
    (1, 3, 50, 16, 0, 96, True, False, True),       # K padded 16 -> 32, Cout not a multiple of the 64-column wave tile
    (1, 1, 40, 128, 128, 256, False, False, True),
    (2, 3, 50, 256, 0, 64, False, False, True),      # K = 256 with one 64-column slice: the four waves split the k16 steps
    (1, 2, 77, 128, 128, 32, True, False, False),
    (2, 3, 100, 128, 0, 768, True, 64, False),      # temporal to_qkv with attn_dim_head = 64: the first 32-column tile of every head rotates
    (2, 3, 70, 64, 0, 768, True, 16, False),        # ... = 16: two heads per tile
    (1, 4, 33, 64, 0, 768, False, 8, True),         # ... = 8
    (1, 4, 33, 64, 0, 768, False, 128, False)])     # ... = 128
@pytest.mark.parametrize("exact", [False, True])
def test_projection_a_stationary_bf16x3(gpu, B, T, HW, C1, C2, Cout, ln, rot, res, exact):
    """vmm_proj_bf16x3 (exact: its fp32-MFMA variant vmm_proj_f32, fmt-4 weights) against torch fp32: 1x1 projection with the row tile staged once (optional fused channel LayerNorm) and
    fragment-order weights; epilogue bias / q-scale / rotary / residual."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(31)
    M, K, hid = B * T * HW, C1 + C2, 256
    x1 = torch.randn(M, C1, generator=g) * 1.5 + 0.3
    x2 = torch.randn(M, C2, generator=g) if C2 else None
    xin = torch.cat([x1, x2], 1) if C2 else x1
    w = torch.randn(Cout, K, generator=g) / math.sqrt(K)
    bias = torch.randn(Cout, generator=g)
    gamma = 1 + 0.2 * torch.randn(K, generator=g)
    resid = torch.randn(M, Cout, generator=g)
    a = xin
    if ln:
        a = (xin - xin.mean(1, keepdim=True)) / (xin.var(1, unbiased=False, keepdim=True) + 1e-5).sqrt() * gamma
    ref = a @ w.t() + bias
    q_scale, rot_tab = 1.0, None
    dh = 32 if rot is True else int(rot)  # head width of the rotated columns (attn_dim_head); the rotary span is min(32, dh)
    if rot:
        q_scale = dh ** -0.5
        ref[:, :hid] *= q_scale
        span = min(32, dh) // 2
        ang = torch.zeros(T, dh // 2)
        ang[:, :span] = torch.rand(T, span, generator=g) * 6.28     # identity pairs beyond the span, as hostmath.rotary_table lays them out
        rot_tab = torch.stack([ang.cos(), ang.sin()], -1).contiguous()  # [T][dh/2][2]
        t_of_row = (torch.arange(M) // HW) % T
        cs = rot_tab[t_of_row]                                          # [M][dh/2][2]
        v = ref[:, :2 * hid].reshape(M, 2 * hid // dh, dh // 2, 2)
        e, o = v[..., 0].clone(), v[..., 1].clone()
        v[..., 0] = e * cs[:, None, :, 0] - o * cs[:, None, :, 1]
        v[..., 1] = o * cs[:, None, :, 0] + e * cs[:, None, :, 1]
        ref[:, :2 * hid] = v.reshape(M, 2 * hid)
    if res:
        ref = ref + resid
    wp = _pack_frag(N, lib, gpu, w, 4 if exact else 2)
    x1g, bg, gg, rg = x1.to(gpu), bias.to(gpu), gamma.to(gpu), resid.to(gpu)
    x2g = x2.to(gpu) if C2 else None
    rt = rot_tab.to(gpu) if rot else None
    out = torch.full((M, Cout), 7.0, device=gpu)
    d = N.ConvDesc()
    d.a1, d.C1, d.lda1, d.w, d.bias, d.out, d.ldo = x1g.data_ptr(), C1, C1, wp.data_ptr(), bg.data_ptr(), out.data_ptr(), Cout
    if C2:
        d.a2, d.C2, d.lda2 = x2g.data_ptr(), C2, C2
    if res:
        d.res, d.ldres = rg.data_ptr(), Cout
    d.nimg, d.Hin, d.Win, d.Hv, d.Wv, d.stride = B * T, HW, 1, HW, 1, 1
    d.KH, d.KW, d.sgn_h, d.sgn_w = 1, 1, 1, 1
    d.Hout, d.Wout, d.oscale, d.Cout, d.rot_dh, d.q_scale = HW, 1, 1, Cout, dh, q_scale
    if rot:
        d.rot_tab, d.rot_T, d.rot_HW, d.rot_ncols, d.q_ncols = rt.data_ptr(), T, HW, 2 * hid, hid
    N.check((lib.vmm_proj_f32 if exact else lib.vmm_proj_bf16x3)(C.byref(d), gg.data_ptr() if ln else None, 1e-5, _s()), "proj")
    torch.cuda.synchronize()
    assert relerr(out.cpu(), ref) < (3e-6 if exact else 5e-5)


@pytest.mark.parametrize("Cc", [64, 128])
@pytest.mark.parametrize("B,T,H,W,ntok", [(2, 3, 16, 16, 11), (1, 2, 96, 96, 0), (3, 1, 8, 12, 6)])
def test_fused_linear_attention_block(gpu, B, T, H, W, ntok, Cc):
    """vmm_linattn_block_bf16x3 (LayerNorm -> to_qkv -> linear attention with stacked tokens -> to_out -> +x in three launches, q/k/v on chip)
    against the oracle's block; several pixel splits per frame, more than one sample (token keys differ per sample)."""
    from oracle import unet3d_oracle as uo
    N, lib = _lib()
    g = torch.Generator().manual_seed(21)
    heads, hid = 8, 256
    x = torch.randn(B, Cc, T, H, W, generator=g)
    sd = {"a.fn.norm.gamma": 1 + 0.2 * torch.randn(1, Cc, 1, 1, 1, generator=g),
          "a.fn.fn.to_qkv.weight": torch.randn(3 * hid, Cc, 1, 1, generator=g) * 0.3,
          "a.fn.fn.to_out.weight": torch.randn(Cc, hid, 1, 1, generator=g) / 16,
          "a.fn.fn.to_out.bias": torch.randn(Cc, generator=g),
          "a.fn.fn.to_k.weight": torch.randn(hid, 32, generator=g) / 4,
          "a.fn.fn.to_v.weight": torch.randn(hid, 32, generator=g) / 4}
    tokens = torch.randn(B, ntok, 32, generator=g) if ntok else None
    cfg = uo.UnetCfg(cond_attention="self-stacked" if ntok else "none")
    ref = rows_of(uo.linear_attention_block(sd, "a", x, cfg, tokens))
    wq = _pack_frag(N, lib, gpu, sd["a.fn.fn.to_qkv.weight"].reshape(3 * hid, Cc), 2)
    wo = _pack_frag(N, lib, gpu, sd["a.fn.fn.to_out.weight"].reshape(Cc, hid), 3)
    xg, gam, bo = rows_of(x).to(gpu), sd["a.fn.norm.gamma"].reshape(-1).to(gpu), sd["a.fn.fn.to_out.bias"].to(gpu)
    ek = ev = None
    if ntok:
        ek = torch.nn.functional.linear(tokens, sd["a.fn.fn.to_k.weight"]).to(gpu)
        ev = torch.nn.functional.linear(tokens, sd["a.fn.fn.to_v.weight"]).to(gpu)
    ws = torch.empty(int(lib.vmm_linattn_block_workspace(B, T, H * W)), device=gpu)
    out = torch.empty_like(xg)
    N.check(lib.vmm_linattn_block_bf16x3(xg.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), bo.data_ptr(), ek.data_ptr() if ntok else None,
                                         ev.data_ptr() if ntok else None, ntok, ws.data_ptr(), out.data_ptr(), Cc, B, T, H * W, Cc, heads, 1e-5, _s()),
            "fused linear attention")
    torch.cuda.synchronize()
    assert relerr(out.cpu() - rows_of(x), ref - rows_of(x)) < 5e-5  # on the attention branch alone (the residual would mask errors)
    out3 = out.clone()
    out.fill_(7.0)  # the "bf16" throughput mode of the same block
    N.check(lib.vmm_linattn_block_bf16(xg.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), bo.data_ptr(), ek.data_ptr() if ntok else None,
                                       ev.data_ptr() if ntok else None, ntok, ws.data_ptr(), out.data_ptr(), Cc, B, T, H * W, Cc, heads, 1e-5, _s()),
            "fused linear attention, bf16 mode")
    torch.cuda.synchronize()
    e1 = relerr(out.cpu() - rows_of(x), ref - rows_of(x))
    assert e1 < 2e-2, e1
    # (the branch is dominated by to_out's bias here, so the rounding of the operands barely shows in e1: what proves that the single-pass kernels ran
    # is that the result differs from the three-pass one, by more than its error and less than bf16's)
    d13 = relerr(out.cpu() - rows_of(x), out3.cpu() - rows_of(x))
    assert 0 < d13 < 2e-2, d13


@pytest.mark.parametrize("HW,ntok,nsplit", [(144, 11, 1), (1000, 0, 4), (2304, 16, 7)])
def test_linear_attention_core(gpu, HW, ntok, nsplit):
    N, lib = _lib()
    g = torch.Generator().manual_seed(8)
    B, T, heads = 2, 3, 2
    hid = heads * 32
    qkv = torch.randn(B * T, HW, 3, heads, 32, generator=g) * 2
    q, k, v = (qkv[:, :, i].permute(0, 2, 3, 1) for i in range(3))  # bt h d n
    ek = ev = None
    if ntok:
        ek, ev = torch.randn(B, ntok, heads, 32, generator=g), torch.randn(B, ntok, heads, 32, generator=g)
        ekf = ek.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        evf = ev.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        k, v = torch.cat([ekf, k], -1), torch.cat([evf, v], -1)
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(-1), v / HW)
    ref = torch.einsum("bhde,bhdn->bhen", ctx, q.softmax(-2) * 32 ** -0.5).permute(0, 3, 1, 2).reshape(B * T * HW, hid)
    qg = qkv.reshape(B * T * HW, 3 * hid).to(gpu)
    part = torch.empty(B * T * heads * nsplit * (1024 + 64), device=gpu)
    ctxg = torch.empty(B * T * heads * 1024, device=gpu)
    out = torch.empty(B * T * HW, hid, device=gpu)
    ekg = ek.reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.reshape(B, ntok, hid).to(gpu) if ntok else None
    N.check(lib.vmm_linattn_context(qg.data_ptr(), 3 * hid, ekg.data_ptr() if ntok else None, evg.data_ptr() if ntok else None, ntok, B, T, HW, heads, 32,
                                    nsplit, part.data_ptr(), ctxg.data_ptr(), None, _s()), "ctx")
    N.check(lib.vmm_linattn_apply(qg.data_ptr(), 3 * hid, ctxg.data_ptr(), out.data_ptr(), hid, B, T, HW, heads, 32, _s()), "apply")
    torch.cuda.synchronize()
    assert relerr(ctxg.cpu().reshape(B * T, heads, 32, 32), ctx) < 5e-6
    assert relerr(out.cpu(), ref) < 5e-6
    # pass 1 on the split-bf16 matrix cores (inference): the same context within the split-bf16 tolerance
    part.fill_(float("nan"))
    ctx2 = torch.full_like(ctxg, float("nan"))
    N.check(lib.vmm_linattn_context_bf16x3(qg.data_ptr(), 3 * hid, ekg.data_ptr() if ntok else None, evg.data_ptr() if ntok else None, ntok, B, T, HW, heads,
                                           32, nsplit, part.data_ptr(), ctx2.data_ptr(), None, _s()), "ctx mfma")
    torch.cuda.synchronize()
    assert relerr(ctx2.cpu().reshape(B * T, heads, 32, 32), ctx) < 3e-5


def test_quantile_matches_torch(gpu):
    from videometamaterials_amd import hostmath
    from videometamaterials_amd.plan import Q_STRIDE
    N, lib = _lib()
    g = torch.Generator().manual_seed(9)
    for n, B in ((304128, 4), (1000, 3), (33792, 2)):
        x = torch.randn(B, n, generator=g).abs() * torch.tensor([0.2, 1.0, 3.0, 0.9][:B])[:, None]
        x[0, : n // 3] = x[0, 0]  # heavy ties
        want = torch.quantile(x, 0.9, dim=-1).clamp(min=1.0)
        k_lo, frac = hostmath.quantile_rank(n, 0.9)
        s = torch.empty(B, device=gpu)
        scratch = torch.empty(B * Q_STRIDE, dtype=torch.int32, device=gpu)
        xg = x.to(gpu)
        N.check(lib.vmm_quantile_rows(xg.data_ptr(), B, n, k_lo, frac, 1.0, s.data_ptr(), scratch.data_ptr(), _s()), "quantile")
        torch.cuda.synchronize()
        assert torch.equal(s.cpu(), want), (n, s.cpu(), want)


def test_dense_batched_and_embeddings(gpu):
    N, lib = _lib()
    g = torch.Generator().manual_seed(10)
    jobs_spec = [(5, 64, 256, 0, 2, True), (44, 256, 256, 0, 0, False), (4, 256, 1024, 1, 0, True), (3, 1, 32, 0, 1, True), (88, 256, 256, 0, 0, True),
                 (9, 1024, 40, 2, 1, True), (8, 1100, 12, 0, 0, False), (17, 36, 70, 1, 0, True)]
    arr = (N.DenseJob * len(jobs_spec))()
    keep, refs, outs = [], [], []
    max_units = 0
    for i, (rows, K, Nn, ai, ao, hasb) in enumerate(jobs_spec):
        x, w, b = torch.randn(rows, K, generator=g), torch.randn(Nn, K, generator=g) / math.sqrt(K), torch.randn(Nn, generator=g)
        act = {0: lambda t: t, 1: F.silu, 2: F.gelu}
        refs.append(act[ao](F.linear(act[ai](x), w, b if hasb else None)))
        xg, wg, bg, yg = x.to(gpu), w.to(gpu), b.to(gpu), torch.empty(rows, Nn, device=gpu)
        keep += [xg, wg, bg]
        outs.append(yg)
        a = arr[i]
        a.x, a.w, a.b, a.y = xg.data_ptr(), wg.data_ptr(), bg.data_ptr() if hasb else None, yg.data_ptr()
        a.rows, a.K, a.N, a.ldx, a.ldy, a.ldadd, a.act_in, a.act_out = rows, K, Nn, K, Nn, Nn, ai, ao
        max_units = max(max_units, Nn)
    tab = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_dense_batched(tab.data_ptr(), len(jobs_spec), max_units, _s()), "dense")
    torch.cuda.synchronize()
    for o, r in zip(outs, refs):
        assert relerr(o.cpu(), r) < 2e-6
    # sinusoidal embedding (vddp.py:139-151)
    t = torch.tensor([0, 1, 17, 255])
    dim = 64
    half = dim // 2
    emb = torch.exp(torch.arange(half) * -(math.log(10000) / (half - 1)))
    e = t[:, None] * emb[None, :]
    ref = torch.cat((e.sin(), e.cos()), -1)
    out = torch.empty(4, dim, device=gpu)
    tg = t.to(gpu)
    N.check(lib.vmm_sinusoidal_embed(tg.data_ptr(), 4, dim, -(math.log(10000) / (half - 1)), out.data_ptr(), _s()), "sin")
    torch.cuda.synchronize()
    assert torch.allclose(out.cpu(), ref, atol=2e-5)


def test_diffusion_elementwise(gpu):
    from videometamaterials_amd import GaussianDiffusion
    from oracle import diffusion_oracle as do

    class _Dummy(torch.nn.Module):
        pass

    g = torch.Generator().manual_seed(11)
    d = GaussianDiffusion(_Dummy(), image_size=8, num_frames=3, channels=2, timesteps=256, sampling_timesteps=256).to(gpu)
    sch = do.schedule_buffers(256)
    for name in do.SCHEDULE_NAMES:
        assert torch.equal(getattr(d, name).cpu(), sch[name])
    x0, noise = torch.rand(3, 2, 3, 8, 8, generator=g) * 2 - 1, torch.randn(3, 2, 3, 8, 8, generator=g)
    t = torch.tensor([0, 100, 255])
    got = d.q_sample(x0.to(gpu), t.to(gpu), noise.to(gpu)).cpu()
    assert torch.allclose(got, do.q_sample(sch, x0, t, noise), rtol=1e-6, atol=1e-7)
    eps = torch.randn(3, 2, 3, 8, 8, generator=g)
    got = d.predict_start_from_noise(x0.to(gpu), t.to(gpu), eps.to(gpu)).cpu()
    assert torch.allclose(got, do.predict_start_from_noise(sch, x0, t, eps), rtol=1e-5, atol=1e-6)
    mean, var, logvar = d.q_posterior(eps.to(gpu), x0.to(gpu), t.to(gpu))
    want_mean, want_logvar = do.q_posterior_mean_logvar(sch, eps, x0, t)
    assert torch.allclose(mean.cpu(), want_mean, rtol=1e-5, atol=1e-6) and torch.equal(logvar.cpu(), want_logvar)


@pytest.mark.parametrize("B,T,HW,ntok,use_bias,bias_on_cond,rot", [(2, 11, 37, 11, True, 1, True), (1, 16, 9, 16, True, 0, True), (3, 5, 130, 0, False, 0, False),
                                                             (1, 11, 700, 3, True, 0, True)])
@pytest.mark.parametrize("heads,dh", [(8, 32), (4, 32), (3, 16), (8, 64), (2, 24), (5, 8), (1, 128)])
def test_temporal_attention_backward(gpu, B, T, HW, ntok, use_bias, bias_on_cond, rot, heads, dh):
    """vmm_attention_bwd mode 0 (LDS-staged workgroup-per-pixel kernel, temporal_attn_bwd.hip) against torch autograd of the same
    attention: softmax over [conditioning tokens | frames] per (pixel, head), q-scale + rotary in front, relative-position bias."""
    N, lib = _lib()
    if (heads, dh) != (8, 32) and HW > 200:
        pytest.skip("the generic passes: the small shapes")
    g = torch.Generator().manual_seed(5)
    hid = heads * dh
    scale = dh ** -0.5
    raw = torch.randn(B, T, HW, 3 * hid, generator=g, dtype=torch.float64, requires_grad=True)
    ek = torch.randn(B, ntok, hid, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    ev = torch.randn(B, ntok, hid, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    bias = (torch.randn(heads, T, T, generator=g, dtype=torch.float64) * 0.5).requires_grad_(True) if use_bias else None
    ang = torch.zeros(T, dh // 2, dtype=torch.float64)
    ang[:, :min(32, dh) // 2] = torch.rand(T, min(32, dh) // 2, generator=g, dtype=torch.float64) * 6.28  # identity pairs beyond the rotary span
    cs, sn = ang.cos(), ang.sin()

    def rotate(x):  # x (B, T, HW, heads, dh), position = frame
        if not rot:
            return x
        e, o = x[..., 0::2], x[..., 1::2]
        c, s = cs[None, :, None, None, :], sn[None, :, None, None, :]
        return torch.stack([e * c - o * s, o * c + e * s], -1).flatten(-2)

    q = rotate(raw[..., :hid].reshape(B, T, HW, heads, dh) * scale)
    k = rotate(raw[..., hid:2 * hid].reshape(B, T, HW, heads, dh))
    v = raw[..., 2 * hid:].reshape(B, T, HW, heads, dh)
    sim = torch.einsum("biphd,bjphd->bphij", q, k)
    if use_bias:
        sim = sim + bias[None, None]
    if ntok:
        simt = torch.einsum("biphd,bjhd->bphij", q, ek.reshape(B, ntok, heads, dh))
        if use_bias and bias_on_cond:
            simt = simt + bias[None, None, :, :, :ntok]
        sim = torch.cat([simt, sim], -1)
    lse = torch.logsumexp(sim, -1)                       # (B, HW, heads, T)
    attn = (sim - lse[..., None]).exp()
    out = torch.einsum("bphij,bjphd->biphd", attn[..., ntok:], v)
    if ntok:
        out = out + torch.einsum("bphij,bjhd->biphd", attn[..., :ntok], ev.reshape(B, ntok, heads, dh))
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    (out * dout).sum().backward()

    f = lambda t: t.detach().float().contiguous().to(gpu)
    qkv_g = f(torch.cat([q.flatten(-2), k.flatten(-2), v.flatten(-2)], -1).reshape(-1, 3 * hid))
    out_g, dout_g = f(out.reshape(-1, hid)), f(dout.reshape(-1, hid))
    lse_g = f(lse.permute(0, 3, 1, 2).reshape(-1, heads))  # rows (b, t, pix) x heads
    tab = f(torch.stack([cs, sn], -1)) if rot else None
    ek_g, ev_g = (f(ek), f(ev)) if ntok else (None, None)
    bias_g = f(bias) if use_bias else None
    rows = B * T * HW
    dqkv = torch.full((rows, 3 * hid), float("nan"), device=gpu)
    dek, dev_ = torch.zeros(B, max(ntok, 1), hid, device=gpu), torch.zeros(B, max(ntok, 1), hid, device=gpu)
    dbias = torch.zeros(heads, T, T, device=gpu)
    dbuf = torch.zeros(int(lib.vmm_attention_bwd_scratch(0, B, T, HW, heads, ntok)), device=gpu)
    p = lambda t: t.data_ptr() if t is not None else None
    N.check(lib.vmm_attention_bwd(0, p(qkv_g), 3 * hid, p(ek_g), p(ev_g), ntok, 0, p(bias_g), bias_on_cond, p(out_g), p(dout_g), hid, p(lse_g), p(tab),
                                  scale, p(dqkv), p(dek), p(dev_), p(dbias), p(dbuf), B, T, HW, heads, dh, _s()), "attention bwd")
    torch.cuda.synchronize()
    assert relerr(dqkv.cpu(), raw.grad.reshape(rows, 3 * hid)) < 2e-5
    if ntok:
        assert relerr(dek.cpu(), ek.grad) < 2e-5
        assert relerr(dev_.cpu(), ev.grad) < 2e-5
    if use_bias:
        assert relerr(dbias.cpu(), bias.grad) < 2e-5


def test_dense_backward_batched(gpu):
    """vmm_dense_bwd_batched (autograd of the small Linear layers around the network: FiLM projections vddp.py:293-296, token keys / values :344-345,
    the embedding MLPs :633-660) against torch autograd, one launch over jobs of both kernels: no output activation (tiled kernel: 64 columns per
    workgroup, rows in LDS) and SiLU / GELU outputs (wave per column, g written back over dy), = and += accumulation, ragged N, shared x."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(31)
    act = {0: lambda v: v, 1: F.silu, 2: lambda v: F.gelu(v)}
    specs = [dict(rows=4, K=256, N=128, act_in=1, act_out=0, acc=0, bias=True), dict(rows=44, K=64, N=256, act_in=0, act_out=0, acc=1, bias=False),
             dict(rows=70, K=64, N=100, act_in=0, act_out=0, acc=0, bias=True), dict(rows=4, K=64, N=256, act_in=0, act_out=2, acc=0, bias=True),
             dict(rows=4, K=256, N=1024, act_in=1, act_out=0, acc=1, bias=True), dict(rows=9, K=16, N=64, act_in=0, act_out=1, acc=0, bias=True)]
    jobs = (N.DenseBwdJob * len(specs))()
    keep, checks = [], []
    for i, sp in enumerate(specs):
        x = torch.randn(sp["rows"], sp["K"], generator=g, dtype=torch.float64, requires_grad=True)
        w = (torch.randn(sp["N"], sp["K"], generator=g, dtype=torch.float64) / sp["K"] ** 0.5).requires_grad_(True)
        b = torch.randn(sp["N"], generator=g, dtype=torch.float64, requires_grad=True) if sp["bias"] else None
        y = act[sp["act_out"]](F.linear(act[sp["act_in"]](x), w, b))
        dy = torch.randn(y.shape, generator=g, dtype=torch.float64)
        (y * dy).sum().backward()
        f = lambda t: t.detach().float().contiguous().to(gpu)
        xg, wg, dyg = f(x), f(w), f(dy)
        bg = f(b) if b is not None else None
        pre_w, pre_b = torch.randn(sp["N"], sp["K"], generator=g).to(gpu), torch.randn(sp["N"], generator=g).to(gpu)
        dwg, dbg = (pre_w.clone(), pre_b.clone()) if sp["acc"] else (torch.full_like(pre_w, float("nan")), torch.full_like(pre_b, float("nan")))
        dxg = torch.zeros(sp["rows"], sp["K"], device=gpu)
        keep += [xg, wg, dyg, bg, dwg, dbg, dxg]
        j = jobs[i]
        j.x, j.w, j.b, j.dy, j.dx, j.dw, j.db = xg.data_ptr(), wg.data_ptr(), bg.data_ptr() if bg is not None else None, dyg.data_ptr(), dxg.data_ptr(), dwg.data_ptr(), dbg.data_ptr() if bg is not None else None
        j.rows, j.K, j.N, j.ldx, j.lddy, j.lddx = sp["rows"], sp["K"], sp["N"], sp["K"], sp["N"], sp["K"]
        j.act_in, j.act_out, j.accumulate = sp["act_in"], sp["act_out"], sp["acc"]
        checks.append((sp, dwg, dbg if b is not None else None, dxg, w.grad, b.grad if b is not None else None, x.grad, pre_w, pre_b))
    table = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(gpu)
    max_n = max(sp["N"] for sp in specs)
    max_x = max(sp["rows"] * ((sp["K"] + 63) // 64) for sp in specs)
    N.check(lib.vmm_dense_bwd_batched(table.data_ptr(), len(specs), max_n, max_x, _s()), "dense bwd")
    torch.cuda.synchronize()
    for sp, dwg, dbg, dxg, gw, gb, gx, pre_w, pre_b in checks:
        base_w, base_b = (pre_w.cpu(), pre_b.cpu()) if sp["acc"] else (0.0, 0.0)
        assert relerr(dwg.cpu() - base_w, gw) < 2e-5, sp
        if dbg is not None:
            assert relerr(dbg.cpu() - base_b, gb) < 2e-5, sp
        assert relerr(dxg.cpu(), gx) < 2e-5, sp


@pytest.mark.parametrize("B,T,HW,per_frame_tok", [(2, 11, 144, True), (1, 3, 64, False), (2, 4, 36, True), (1, 2, 400, False)])
def test_spatial_attention_backward(gpu, B, T, HW, per_frame_tok):
    """vmm_attention_bwd mode 1 (mid spatial attention, vddp.py:687-689 under autograd: softmax over [the frame's conditioning token | the frame's
    pixels] per (frame, head)) against torch autograd in float64.  HW < 256 takes the two LDS-staged kernels (keys / values staged for the query
    pass, queries for the key pass), HW = 400 the generic pair."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(9)
    heads, dh = 8, 32
    hid, scale = heads * dh, dh ** -0.5
    ntok = T if per_frame_tok else 0
    raw = torch.randn(B, T, HW, 3 * hid, generator=g, dtype=torch.float64, requires_grad=True)
    ek = torch.randn(B, ntok, hid, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    ev = torch.randn(B, ntok, hid, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    q = raw[..., :hid].reshape(B, T, HW, heads, dh) * scale
    k = raw[..., hid:2 * hid].reshape(B, T, HW, heads, dh)
    v = raw[..., 2 * hid:].reshape(B, T, HW, heads, dh)
    sim = torch.einsum("btihd,btjhd->bthij", q, k)
    if ntok:  # frame t sees token t only
        simt = torch.einsum("btihd,bthd->bthi", q, ek.reshape(B, T, heads, dh))[..., None]
        sim = torch.cat([simt, sim], -1)
    lse = torch.logsumexp(sim, -1)  # (B, T, heads, HW)
    attn = (sim - lse[..., None]).exp()
    out = torch.einsum("bthij,btjhd->btihd", attn[..., (1 if ntok else 0):], v)
    if ntok:
        out = out + attn[..., 0].permute(0, 1, 3, 2)[..., None] * ev.reshape(B, T, 1, heads, dh)
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    (out * dout).sum().backward()
    f = lambda t: t.detach().float().contiguous().to(gpu)
    qkv_g = f(torch.cat([q.flatten(-2), k.flatten(-2), v.flatten(-2)], -1).reshape(-1, 3 * hid))
    out_g, dout_g = f(out.reshape(-1, hid)), f(dout.reshape(-1, hid))
    lse_g = f(lse.permute(0, 1, 3, 2).reshape(-1, heads))  # rows (b, t, pix) x heads
    ek_g, ev_g = (f(ek), f(ev)) if ntok else (None, None)
    rows = B * T * HW
    dqkv = torch.full((rows, 3 * hid), float("nan"), device=gpu)
    dek, dev_ = torch.zeros(B, max(ntok, 1), hid, device=gpu), torch.zeros(B, max(ntok, 1), hid, device=gpu)
    dbuf = torch.zeros(int(lib.vmm_attention_bwd_scratch(1, B, T, HW, heads, ntok)), device=gpu)
    p = lambda t: t.data_ptr() if t is not None else None
    N.check(lib.vmm_attention_bwd(1, p(qkv_g), 3 * hid, p(ek_g), p(ev_g), ntok, 1 if ntok else 0, None, 0, p(out_g), p(dout_g), hid, p(lse_g), None,
                                  scale, p(dqkv), p(dek), p(dev_), None, p(dbuf), B, T, HW, heads, dh, _s()), "attention bwd (spatial)")
    torch.cuda.synchronize()
    assert relerr(dqkv.cpu(), raw.grad.reshape(rows, 3 * hid)) < 2e-5
    if ntok:
        assert relerr(dek.cpu(), ek.grad) < 2e-5
        assert relerr(dev_.cpu(), ev.grad) < 2e-5


@pytest.mark.parametrize("HW,ntok", [(144, 11), (100, 0), (2304, 5)])
def test_linear_attention_matrix_core_row_passes(gpu, HW, ntok):
    """heads = 8 takes the fp32 matrix-core row passes (linattn_rows.hip): vmm_linattn_apply forward and the row pass of
    vmm_linattn_bwd, both against torch (forward formula of vddp.py:313-378, gradients from autograd in float64)."""
    N, lib = _lib()
    g = torch.Generator().manual_seed(21)
    B, T, heads = 1, 2, 8
    hid, scale = heads * 32, 32 ** -0.5
    raw = (torch.randn(B * T, HW, 3, heads, 32, generator=g, dtype=torch.float64) * 1.5).requires_grad_(True)
    ek = torch.randn(B, ntok, heads, 32, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    ev = torch.randn(B, ntok, heads, 32, generator=g, dtype=torch.float64, requires_grad=True) if ntok else None
    q, k, v = (raw[:, :, i].permute(0, 2, 3, 1) for i in range(3))  # (bt, h, d, n)
    if ntok:
        ekf = ek.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        evf = ev.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        k, v = torch.cat([ekf, k], -1), torch.cat([evf, v], -1)
    ctx = torch.einsum("bhdn,bhen->bhde", k.softmax(-1), v / HW)
    out = torch.einsum("bhde,bhdn->bhen", ctx, q.softmax(-2) * scale).permute(0, 3, 1, 2).reshape(B * T * HW, hid)
    dout = torch.randn(out.shape, generator=g, dtype=torch.float64)
    (out * dout).sum().backward()

    f = lambda t: t.detach().float().contiguous().to(gpu)
    qg = f(raw.reshape(B * T * HW, 3 * hid))
    ekg, evg = (f(ek.reshape(B, ntok, hid)), f(ev.reshape(B, ntok, hid))) if ntok else (None, None)
    p = lambda t: t.data_ptr() if t is not None else None
    nsplit = 3
    part = torch.empty(B * T * heads * nsplit * (1024 + 64), device=gpu)
    ctxg = torch.empty(B * T * heads * 1024, device=gpu)
    kstat = torch.empty(B * T * heads * 64, device=gpu)
    og = torch.full((B * T * HW, hid), float("nan"), device=gpu)
    N.check(lib.vmm_linattn_context(p(qg), 3 * hid, p(ekg), p(evg), ntok, B, T, HW, heads, 32, nsplit, p(part), p(ctxg), p(kstat), _s()), "ctx")
    N.check(lib.vmm_linattn_apply(p(qg), 3 * hid, p(ctxg), p(og), hid, B, T, HW, heads, 32, _s()), "apply")
    torch.cuda.synchronize()
    assert relerr(og.cpu(), out.detach()) < 5e-6
    dctx = torch.empty(B * T * heads * 1024, device=gpu)
    dqkv = torch.full((B * T * HW, 3 * hid), float("nan"), device=gpu)
    dek, dev_ = torch.zeros(B, max(ntok, 1), hid, device=gpu), torch.zeros(B, max(ntok, 1), hid, device=gpu)
    N.check(lib.vmm_linattn_bwd(p(qg), 3 * hid, p(ekg), p(evg), ntok, p(ctxg), p(kstat), p(f(dout)), hid, p(dctx), p(dqkv), p(dek), p(dev_), B, T, HW,
                                heads, 32, _s()), "linattn bwd")
    torch.cuda.synchronize()
    assert relerr(dqkv.cpu(), raw.grad.reshape(B * T * HW, 3 * hid)) < 2e-5
    if ntok:
        assert relerr(dek.cpu(), ek.grad.reshape(B, ntok, hid)) < 2e-5
        assert relerr(dev_.cpu(), ev.grad.reshape(B, ntok, hid)) < 2e-5


def test_attention_blocks_on_fp16_split_operands_experiment(gpu):
    """libvmm_hip_exp.so only (skipped on the product library): vmm_temporal_block_f16x3 / vmm_linattn_block_f16x3 -- the fused attention blocks on IEEE-half
    hi | lo operands (three passes; weights fmt | 32) -- against the split-bf16 blocks on the same inputs: two fp32-class evaluations of one block, 2e-5 apart at
    most (the fp16 form is the closer one to fp64: tools/bench_attn_split.py)."""
    N, lib = _lib()
    if not hasattr(lib, "vmm_temporal_block_f16x3"):
        pytest.skip("experiments library not loaded")
    from videometamaterials_amd import hostmath
    for name, argtypes in N.EXPERIMENT_SIGNATURES.items():
        getattr(lib, name).argtypes = argtypes
    g = torch.Generator().manual_seed(5)
    B, T, HW, ntok, Cc, heads, hid = 2, 11, 64, 11, 64, 8, 256
    x = (torch.randn(B * T * HW, Cc, generator=g) * 1.5 + 0.3).to(gpu)
    wqkv, wout = torch.randn(3 * hid, Cc, generator=g) / 8, torch.randn(Cc, hid, generator=g) / 16
    gam, bias, rot = (1 + 0.2 * torch.randn(Cc, generator=g)).to(gpu), torch.randn(heads, T, T, generator=g).to(gpu), hostmath.rotary_table(T, 32).to(gpu)
    ek, ev = torch.randn(B, ntok, hid, generator=g).to(gpu), torch.randn(B, ntok, hid, generator=g).to(gpu)
    bo = torch.randn(Cc, generator=g).to(gpu)
    outs = {}
    for v, f in (("bf16x3", 0), ("f16x3", 32)):
        wq, wo = _pack_frag(N, lib, gpu, wqkv, 2 | f), _pack_frag(N, lib, gpu, wout, 3 | f)
        out = torch.empty_like(x)
        N.check(getattr(lib, "vmm_temporal_block_" + v)(x.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), ek.data_ptr(), ev.data_ptr(), ntok, bias.data_ptr(), 1,
                                                         rot.data_ptr(), out.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(32 ** -0.5), C.c_float(1e-5), _s()), v)
        ws = torch.empty(lib.vmm_linattn_block_workspace(B, T, HW), device=gpu)
        out2 = torch.empty_like(x)
        N.check(getattr(lib, "vmm_linattn_block_" + v)(x.data_ptr(), Cc, gam.data_ptr(), wq.data_ptr(), wo.data_ptr(), bo.data_ptr(), ek.data_ptr(), ev.data_ptr(), ntok,
                                                        ws.data_ptr(), out2.data_ptr(), Cc, B, T, HW, Cc, heads, C.c_float(1e-5), _s()), v)
        torch.cuda.synchronize()
        outs[v] = (out.cpu(), out2.cpu())
    assert relerr(outs["f16x3"][0] - x.cpu(), outs["bf16x3"][0] - x.cpu()) < 2e-5
    assert relerr(outs["f16x3"][1] - x.cpu(), outs["bf16x3"][1] - x.cpu()) < 2e-5
    assert not torch.equal(outs["f16x3"][0], outs["bf16x3"][0])  # (two operand forms, not one kernel under two names)


@pytest.mark.parametrize("env", [{"VMM_C3_PERSISTENT": "1"}, {"VMM_C3_PERSISTENT": "2"}, {"VMM_C3_NJ1": "1"}, {}])
def test_experiments_library(gpu, env):
    """libvmm_hip_exp.so (VMM_EXPERIMENTS=1 build, made by __graft_entry__.build()): the kernels that lost their A/B stay parity-green.  The
    persistent wave-specialised 3 x 3 kernel for the 64-column 2-D layers (1) / every shape (2), the 32-column-wave-tile instance at three
    workgroups per CU (VMM_C3_NJ1), each selected per process (read once): the 3 x 3 kernel tests and the denoiser goldens again; {}: the Winograd kernel, the balanced ("stream-K")
    launch of the few-tile 3 x 3 layers (kernel test + the goldens: plans built on this library use it)."""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exp = os.path.join(root, "videometamaterials_amd", "libvmm_hip_exp.so")
    if not os.path.exists(exp):
        pytest.skip("libvmm_hip_exp.so not built (VMM_EXPERIMENTS=1 python -m videometamaterials_amd.build)")
    e = dict(os.environ, VMM_LIB_PATH=exp, **env)
    e.pop("VMM_C3_LEGACY", None)
    sel = "conv3x3_halo or fused_gn or forward_matches_reference_golden or shared_source" if env else "winograd or balanced or fp16_split_operands or forward_matches_reference_golden"
    r = subprocess.run([sys.executable, "-m", "pytest", "-x", "-q", "-p", "no:cacheprovider", "tests/test_gpu_kernels.py", "tests/test_gpu_unet.py", "-k", sel],
                       cwd=root, env=e, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]


@pytest.mark.parametrize("rows,acc", [(64, False), (1920, True), (70400, True), (20032, False)])
def test_fused_to_qkv_backward_with_layernorm_epilogue(gpu, rows, acc):
    """vmm_qkv_bwd_ln_bf16x3: the to_qkv backward with the PreNorm LayerNorm's backward as its epilogue (vddp.py:245-264 differentiated): dx (= | +=), dgamma
    (+=), dW (+=) against torch autograd in fp64 of  qkv = LayerNorm(x) W^T  with the upstream gradient g; bit-reproducible."""
    N, lib = _lib()
    g_ = torch.Generator().manual_seed(rows + 1)
    Cc, Nq = 64, 768
    x = (torch.randn(rows, Cc, generator=g_) * 1.5 + 0.3).double().requires_grad_(True)
    gamma = (1 + 0.2 * torch.randn(Cc, generator=g_)).double().requires_grad_(True)
    w = (torch.randn(Nq, Cc, generator=g_) / 8).double().requires_grad_(True)
    g = torch.randn(rows, Nq, generator=g_)
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = 1 / (var + 1e-5).sqrt()
    y = (x - mean) * rstd * gamma
    (y @ w.t() * g.double()).sum().backward()
    res = torch.randn(rows, Cc, generator=g_)
    want_dx = x.grad + (res.double() if acc else 0)
    wp = _pack_frag(N, lib, gpu, w.detach().float().t().contiguous(), 2)
    xg, gg = x.detach().float().to(gpu), g.to(gpu)
    stats = torch.cat([mean.detach(), rstd.detach()], 1).float().contiguous().to(gpu)
    gam = gamma.detach().float().to(gpu)
    n_ws = int(lib.vmm_qkv_bwd_workspace(rows, Cc, Nq))
    outs = []
    for _ in range(2):
        ws = torch.full((n_ws,), float("nan"), device=gpu)
        dx = res.clone().to(gpu) if acc else torch.full((rows, Cc), 7.0, device=gpu)
        dw = torch.ones(Cc, Nq, device=gpu)
        dgam = torch.ones(Cc, device=gpu)
        rc = lib.vmm_qkv_bwd_ln_bf16x3(xg.data_ptr(), Cc, stats.data_ptr(), gam.data_ptr(), gg.data_ptr(), Nq, wp.data_ptr(), dx.data_ptr(), Cc, 1 if acc else 0,
                                       dgam.data_ptr(), dw.data_ptr(), ws.data_ptr(), rows, Cc, Nq, _s())
        assert rc == 0, rc
        torch.cuda.synchronize()
        assert relerr(dx.cpu().double(), want_dx) < 5e-5
        assert relerr(dw.cpu().double() - 1, w.grad.t()) < 5e-5
        assert relerr(dgam.cpu().double() - 1, gamma.grad) < 5e-5
        outs.append((dx, dw, dgam))
    assert all(torch.equal(a, b) for a, b in zip(outs[0], outs[1]))
    assert lib.vmm_qkv_bwd_ln_bf16x3(xg.data_ptr(), Cc, None, gam.data_ptr(), gg.data_ptr(), Nq, wp.data_ptr(), dx.data_ptr(), Cc, 0, dgam.data_ptr(),
                                     dw.data_ptr(), ws.data_ptr(), rows, Cc, Nq, _s()) == 1  # (the statistics are required)
