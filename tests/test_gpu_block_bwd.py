"""Backward of the fused attention blocks WITH RECOMPUTATION (temporal_block_bwd.hip, linattn_block_bwd.hip) against torch autograd of the same
block in fp32 on the CPU (autograd of vddp.py:313-378, 396-535 inside Residual(PreNorm(.))).  The kernels see only x, dOut and the weights: the
training forward (the fused block) stores no qkv rows, no attention output and no softmax statistics."""
import ctypes as C

import pytest
import torch

from test_gpu_kernels import _attn_ref, _lib, _pack_frag, _s, relerr

pytestmark = pytest.mark.gpu


def _pack_frag_t(N, lib, gpu, w2d, fmt=2):
    """(out, in) weight -> fmt-2 fragments of its TRANSPOSED use: K = out features, N = in features (the data-gradient operand, pack_linear_slice)."""
    co, ci = w2d.shape
    wg = w2d.contiguous().to(gpu)
    packed = torch.zeros((co + 31) // 32 * 32 * ((ci + 31) // 32 * 32), device=gpu)
    job = (N.PackJob * 1)()
    j = job[0]
    j.torch_w, j.packed = wg.data_ptr(), packed.data_ptr()
    j.TH, j.TW, j.C, j.Cp, j.N = 1, 1, co, co, ci
    j.sn, j.sc, j.sh, j.sw, j.h0, j.hs, j.w0, j.ws, j.accumulate, j.fmt = 1, ci, 0, 0, 0, 0, 0, 0, 0, fmt
    tab = torch.frombuffer(bytearray(bytes(job)), dtype=torch.uint8).to(gpu)
    N.check(lib.vmm_pack_weights(tab.data_ptr(), 1, packed.numel(), 0, _s()), "pack")
    torch.cuda.synchronize()
    return packed


# (entry-point suffix, element type of the dqkv rows the kernel stores, tolerance against fp32 autograd): the single-pass instances (train_precision "fp16" /
# "bf16") hand the gradient of the qkv rows to the to_qkv backward in their operand's own 16-bit type (csrc/vmm_common.h VMM_DQKV16)
VARIANTS = [("bf16x3", torch.float32, 1e-4), ("fp16", torch.float16, 6e-3), ("bf16", torch.bfloat16, 4e-2)]


def _dqkv_buffer(rows, hid, dtype, gpu):
    """rows x 768 of the variant's element type with a guard row behind it (a kernel that stored the wrong element width would write into it)."""
    buf = torch.full((rows + 1, 3 * hid), 7.0, device=gpu, dtype=dtype)
    return buf, buf[:rows]


@pytest.mark.parametrize("variant,qdtype,tol", VARIANTS)
@pytest.mark.parametrize("B,T,HW,ntok,bias_on_cond", [(2, 11, 36, 11, 1), (1, 16, 10, 0, 0), (3, 5, 130, 7, 0), (1, 11, 1152, 16, 0), (2, 1, 4, 3, 0),
                                                      (1, 11, 2, 11, 1)])
def test_fused_temporal_block_backward(gpu, B, T, HW, ntok, bias_on_cond, variant, qdtype, tol):
    """vmm_temporal_block_bwd_bf16x3: gradient of the raw to_qkv rows, LayerNorm statistics, dW_out, dbias, d(ek), d(ev) from x and dOut alone;
    several tiles per workgroup, padded frame slots, with / without tokens and the bias on them."""
    from videometamaterials_amd import hostmath
    N, lib = _lib()
    Cc, heads, hid = 64, 8, 256
    ws_n = lib.vmm_temporal_block_bwd_workspace(B, T, HW, Cc, heads, ntok)
    assert ws_n > 0
    g = torch.Generator().manual_seed(101 + T + HW)
    x = torch.randn(B, T, HW, Cc, generator=g) * 1.5 + 0.3
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    wqkv = torch.randn(3 * hid, Cc, generator=g) / 8
    wout = (torch.randn(Cc, hid, generator=g) / 16).requires_grad_()
    bias = torch.randn(heads, T, T, generator=g).requires_grad_()
    rot = hostmath.rotary_table(T, 32)
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = 1 / (var + 1e-5).sqrt()
    y = (x - mean) * rstd * gamma
    qkv_raw = (y @ wqkv.t()).requires_grad_()
    qkv = qkv_raw.reshape(B, T, HW, 3, heads, 32)
    cos, sin = rot[:, :, 0].repeat_interleave(2, -1)[None, :, None, None], rot[:, :, 1].repeat_interleave(2, -1)[None, :, None, None]

    def rotate(t):
        pr = t.reshape(*t.shape[:-1], 16, 2)
        return t * cos + torch.stack((-pr[..., 1], pr[..., 0]), -1).reshape(t.shape) * sin

    q, k, v = rotate(qkv[:, :, :, 0] * 32 ** -0.5), rotate(qkv[:, :, :, 1]), qkv[:, :, :, 2]
    q, k, v = (t.permute(0, 2, 3, 1, 4) for t in (q, k, v))  # b hw h t d
    bfull = bias[None, None]
    ek = ev = None
    if ntok:
        ek = torch.randn(B, ntok, heads, 32, generator=g).requires_grad_()
        ev = torch.randn(B, ntok, heads, 32, generator=g).requires_grad_()
        k = torch.cat([ek.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), k], dim=-2)
        v = torch.cat([ev.permute(0, 2, 1, 3)[:, None].expand(B, HW, heads, ntok, 32), v], dim=-2)
        bfull = torch.cat([bias if bias_on_cond else torch.zeros(heads, T, ntok), bias], dim=-1)[None, None]
    o = _attn_ref(q, k, v, bfull).permute(0, 3, 1, 2, 4).reshape(B * T * HW, hid)
    branch = o @ wout.t()
    dout = torch.randn(B * T * HW, Cc, generator=g)
    branch.backward(dout)

    half = 16 if variant == "fp16" else 0  # IEEE-half weight planes (vmm_pack_weights fmt | 16)
    wq, woT = _pack_frag(N, lib, gpu, wqkv, 2 | half), _pack_frag_t(N, lib, gpu, wout.detach(), 2 | half)
    xg, gg, bg, rg = x.reshape(B * T * HW, Cc).to(gpu), gamma.to(gpu), bias.detach().to(gpu), rot.to(gpu)
    ekg = ek.detach().reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.detach().reshape(B, ntok, hid).to(gpu) if ntok else None
    dg = dout.to(gpu)
    rows = B * T * HW
    dqkv_all, dqkv = _dqkv_buffer(rows, hid, qdtype, gpu)
    stats = torch.full((rows, 2), 7.0, device=gpu)
    dwo = torch.full((hid, Cc), 0.5, device=gpu)  # accumulated into (+=)
    dbias = torch.full((heads, T, T), 0.25, device=gpu)
    dek = torch.zeros(B, max(ntok, 1), hid, device=gpu)
    dev = torch.zeros(B, max(ntok, 1), hid, device=gpu)
    ws = torch.empty(ws_n, device=gpu)
    d = N.AttnBlockBwd()
    d.x, d.ldx, d.gamma = xg.data_ptr(), Cc, gg.data_ptr()
    d.wqkv_frag, d.wout_t_frag = wq.data_ptr(), woT.data_ptr()
    if ntok:
        d.ek, d.ev, d.ntok = ekg.data_ptr(), evg.data_ptr(), ntok
    d.bias, d.bias_on_cond, d.rot_tab = bg.data_ptr(), bias_on_cond, rg.data_ptr()
    d.dout, d.lddo = dg.data_ptr(), Cc
    d.dqkv, d.lddqkv, d.ln_stats = dqkv.data_ptr(), 3 * hid, stats.data_ptr()
    d.dwout_packed, d.dbias, d.dek, d.dev = dwo.data_ptr(), dbias.data_ptr(), dek.data_ptr(), dev.data_ptr()
    d.workspace = ws.data_ptr()
    d.B, d.T, d.HW, d.C, d.heads = B, T, HW, Cc, heads
    d.q_scale, d.eps = 32 ** -0.5, 1e-5
    N.check(getattr(lib, "vmm_temporal_block_bwd_" + variant)(C.byref(d), _s()), "temporal block backward")
    torch.cuda.synchronize()
    want = qkv_raw.grad.reshape(rows, 3 * hid)
    got = dqkv.float().cpu()
    assert bool((dqkv_all[rows] == 7.0).all()), "the kernel wrote past the rows of its element type"
    for i, nm in enumerate("qkv"):
        assert relerr(got[:, i * hid:(i + 1) * hid], want[:, i * hid:(i + 1) * hid]) < tol, nm
    assert relerr(stats.cpu()[:, 0], mean.reshape(-1)) < 1e-5 and relerr(stats.cpu()[:, 1], rstd.reshape(-1)) < 1e-5
    assert relerr(dwo.cpu() - 0.5, wout.grad.t()) < tol
    assert relerr(dbias.cpu() - 0.25, bias.grad) < tol
    if ntok:
        assert relerr(dek.cpu(), ek.grad.reshape(B, ntok, hid)) < tol
        assert relerr(dev.cpu(), ev.grad.reshape(B, ntok, hid)) < tol


@pytest.mark.parametrize("variant,qdtype,tol", VARIANTS)
@pytest.mark.parametrize("B,T,H,W,ntok", [(2, 3, 8, 8, 5), (1, 2, 16, 24, 0), (2, 11, 32, 32, 11), (1, 1, 4, 8, 16), (1, 3, 48, 48, 11)])
def test_fused_linear_attention_block_backward(gpu, B, T, H, W, ntok, variant, qdtype, tol):
    """vmm_linattn_block_bwd_bf16x3 after the fused forward (whose workspace it reads): gradient of the raw to_qkv rows, LayerNorm statistics,
    dW_out, the to_out bias gradient, d(ek), d(ev) -- against torch autograd of SpatialLinearAttention's arithmetic (vddp.py:313-378)."""
    N, lib = _lib()
    Cc, heads, hid = 64, 8, 256
    HW = H * W
    ws_n = lib.vmm_linattn_block_bwd_workspace(B, T, HW, Cc, heads, ntok)
    assert ws_n > 0
    g = torch.Generator().manual_seed(7 + HW + ntok)
    x = torch.randn(B * T, HW, Cc, generator=g) * 1.5 + 0.2
    gamma = 1 + 0.2 * torch.randn(Cc, generator=g)
    wqkv = torch.randn(3 * hid, Cc, generator=g) / 8
    wout = (torch.randn(Cc, hid, generator=g) / 16).requires_grad_()
    bout = torch.randn(Cc, generator=g).requires_grad_()
    mean, var = x.mean(-1, keepdim=True), x.var(-1, unbiased=False, keepdim=True)
    rstd = 1 / (var + 1e-5).sqrt()
    y = (x - mean) * rstd * gamma
    qkv_raw = (y @ wqkv.t()).requires_grad_()                       # (BT, HW, 768)
    q, k, v = (qkv_raw[..., i * hid:(i + 1) * hid].reshape(B * T, HW, heads, 32).permute(0, 2, 3, 1) for i in range(3))  # bt h d n
    ek = ev = None
    if ntok:
        ek = torch.randn(B, ntok, heads, 32, generator=g).requires_grad_()
        ev = torch.randn(B, ntok, heads, 32, generator=g).requires_grad_()
        ekf = ek.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        evf = ev.permute(0, 2, 3, 1)[:, None].expand(B, T, heads, 32, ntok).reshape(B * T, heads, 32, ntok)
        k, v = torch.cat([ekf, k], -1), torch.cat([evf, v], -1)
    qs = q.softmax(dim=-2) * 32 ** -0.5
    ks = k.softmax(dim=-1)
    ctx = torch.einsum("bhdn,bhen->bhde", ks, v / HW)
    out = torch.einsum("bhde,bhdn->bhen", ctx, qs)                  # bt h e n
    o = out.permute(0, 3, 1, 2).reshape(B * T * HW, hid)
    branch = o @ wout.t() + bout
    # (fp16 rows of the qkv gradient: the key softmax runs over HW + ntok entries, so dq / dk are ~1 / HW of dOut -- 1e-7 at 32 x 32 for a unit dOut, below the
    # normal range of IEEE half.  Training multiplies the loss by the GradScaler's 2^16 for exactly this reason (dp.py, vddp.py:1629-1633); the test scales dOut)
    dout = torch.randn(B * T * HW, Cc, generator=g) * (4096.0 if variant == "fp16" else 1.0)
    branch.backward(dout)

    rows = B * T * HW
    half = 16 if variant == "fp16" else 0
    wq, wo3, woT = _pack_frag(N, lib, gpu, wqkv, 2 | half), _pack_frag(N, lib, gpu, wout.detach(), 3 | half), _pack_frag_t(N, lib, gpu, wout.detach(), 2 | half)
    xg, gg, bg = x.reshape(rows, Cc).to(gpu), gamma.to(gpu), bout.detach().to(gpu)
    ekg = ek.detach().reshape(B, ntok, hid).to(gpu) if ntok else None
    evg = ev.detach().reshape(B, ntok, hid).to(gpu) if ntok else None
    fws = torch.empty(lib.vmm_linattn_block_workspace(B, T, HW), device=gpu)
    fout = torch.empty_like(xg)
    N.check(getattr(lib, "vmm_linattn_block_" + variant)(xg.data_ptr(), Cc, gg.data_ptr(), wq.data_ptr(), wo3.data_ptr(), bg.data_ptr(), ekg.data_ptr() if ntok else None,
                                                         evg.data_ptr() if ntok else None, ntok, fws.data_ptr(), fout.data_ptr(), Cc, B, T, HW, Cc, heads,
                                                         C.c_float(1e-5), _s()), "linear attention block")
    torch.cuda.synchronize()
    assert relerr(fout.cpu() - x.reshape(rows, Cc), branch.detach()) < tol / 2
    dg = dout.to(gpu)
    dqkv_all, dqkv = _dqkv_buffer(rows, hid, qdtype, gpu)
    stats = torch.full((rows, 2), 7.0, device=gpu)
    dwo = torch.full((hid, Cc), 0.5, device=gpu)
    dbo = torch.full((Cc,), 0.25, device=gpu)
    dek = torch.zeros(B, max(ntok, 1), hid, device=gpu)
    dev = torch.zeros(B, max(ntok, 1), hid, device=gpu)
    ws = torch.empty(ws_n, device=gpu)
    d = N.AttnBlockBwd()
    d.x, d.ldx, d.gamma = xg.data_ptr(), Cc, gg.data_ptr()
    d.wqkv_frag, d.wout_t_frag = wq.data_ptr(), woT.data_ptr()
    if ntok:
        d.ek, d.ev, d.ntok = ekg.data_ptr(), evg.data_ptr(), ntok
    d.fwd_workspace = fws.data_ptr()
    d.dout, d.lddo = dg.data_ptr(), Cc
    d.dqkv, d.lddqkv, d.ln_stats = dqkv.data_ptr(), 3 * hid, stats.data_ptr()
    d.dwout_packed, d.dbout, d.dek, d.dev = dwo.data_ptr(), dbo.data_ptr(), dek.data_ptr(), dev.data_ptr()
    d.workspace = ws.data_ptr()
    d.B, d.T, d.HW, d.C, d.heads = B, T, HW, Cc, heads
    d.q_scale, d.eps = 32 ** -0.5, 1e-5
    N.check(getattr(lib, "vmm_linattn_block_bwd_" + variant)(C.byref(d), _s()), "linear attention block backward")
    torch.cuda.synchronize()
    want = qkv_raw.grad.reshape(rows, 3 * hid)
    got = dqkv.float().cpu()
    assert bool((dqkv_all[rows] == 7.0).all()), "the kernel wrote past the rows of its element type"
    for i, nm in enumerate("qkv"):
        assert relerr(got[:, i * hid:(i + 1) * hid], want[:, i * hid:(i + 1) * hid]) < tol, nm
    assert relerr(stats.cpu()[:, 0], mean.reshape(-1)) < 1e-5 and relerr(stats.cpu()[:, 1], rstd.reshape(-1)) < 1e-5
    assert relerr(dwo.cpu() - 0.5, wout.grad.t()) < tol
    assert relerr(dbo.cpu() - 0.25, bout.grad) < 1e-5
    if ntok:
        assert relerr(dek.cpu(), ek.grad.reshape(B, ntok, hid)) < tol
        assert relerr(dev.cpu(), ev.grad.reshape(B, ntok, hid)) < tol


def test_training_plan_uses_the_fused_blocks_and_agrees_with_the_unfused_path(gpu, monkeypatch):
    """The split-bf16 training plan of the Lagrangian wiring at the real widths: the C = 64 attention sites run the fused forward blocks and the
    recomputing backward kernels (no launch of the unfused to_qkv / core / to_out chain there), the arena shrinks accordingly, and every
    parameter gradient equals the one of the UNFUSED training path (VMM_DISABLE=fused_attn_train: qkv rows through HBM) on the same inputs --
    two independent implementations of the same backward."""
    import helpers
    import videometamaterials_amd as vm
    kw, (B, T, H, W), _ = helpers.CONFIGS["lagr64"]
    sd = helpers.synth_state_dict(helpers.load_shapes("lagr64"))
    x, t, cond = (v.to(gpu) for v in helpers.synth_inputs("lagr64"))
    g = torch.Generator().manual_seed(3)
    dout = torch.randn(B, 3, T, H, W, generator=g).to(gpu)
    grads, arenas = {}, {}
    for mode in ("fused", "unfused"):
        monkeypatch.setenv("VMM_DISABLE", "" if mode == "fused" else "fused_attn_train")
        m = vm.Unet3D(**kw)
        m.load_state_dict(sd, strict=True)
        m = m.to(gpu)
        m.train_precision = "bf16x3"
        pl = m.get_plan(B, T, H, W, cond.shape[-1], gpu, training=True)
        fwd, bwd = [fn.__name__ for fn, _, _ in pl.steps], [fn.__name__ for fn, _, _ in pl.bwd_steps]
        if mode == "fused":
            # lagr64 at 32 x 32: C = 64 sites = init_temporal_attn, downs.0.{2,3}, ups.2.{2,3}, ups.3.{2,3}
            assert fwd.count("vmm_temporal_block_bf16x3") == 4 and bwd.count("vmm_temporal_block_bwd_bf16x3") == 4
            assert fwd.count("vmm_linattn_block_bf16x3") == 3 and bwd.count("vmm_linattn_block_bwd_bf16x3") == 3
        else:
            assert "vmm_temporal_block_bwd_bf16x3" not in bwd and "vmm_linattn_block_bwd_bf16x3" not in bwd
        pl.run(x, t, cond, torch.zeros(B, dtype=torch.uint8, device=gpu))
        pl.backward(dout)
        torch.cuda.synchronize()
        grads[mode] = {k: v.clone().cpu() for k, v in pl.grad_views(dict(m.named_parameters())).items()}
        arenas[mode] = pl.arena_floats
        del pl, m
    assert arenas["fused"] < 0.75 * arenas["unfused"]
    assert set(grads["fused"]) == set(grads["unfused"])
    typical = max(float(v.double().norm()) for v in grads["unfused"].values())
    for k, w in grads["unfused"].items():
        if float(w.double().norm()) < 1e-9 * typical:
            continue
        assert relerr(grads["fused"][k], w) < 2e-4, (k, relerr(grads["fused"][k], w))
