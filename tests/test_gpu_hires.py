"""BASELINE configs[3] shape (22 frames of 192 x 192, CNN-token conditioning, per_frame_cond=False) at the real widths: the frames exceed
the fused temporal block's 16-slot envelope and the 2-D-tiled 3x3 kernel runs on a 12 x 12 tile grid, so this exercises the fallbacks
and the big-frame paths.  The oracle needs minutes for this size; the check is the agreement of the two independent arithmetic paths
(exact-fp32 MFMA vs split-bf16), each of which is pinned against the reference at the smaller golden sizes."""
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_hires_22x192x192_paths_agree(gpu):
    import videometamaterials_amd as vm
    kw = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
              per_frame_cond=False)
    torch.manual_seed(0)
    m = vm.Unet3D(**kw).to(gpu).eval()
    B, T, H = 1, 22, 192
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, T, H, H, generator=g).to(gpu)
    t = torch.randint(0, 256, (B,), generator=g).to(gpu)
    cond = (torch.rand(B, 51, generator=g) * 2 - 1).to(gpu)
    outs = {}
    for prec in ("bf16x3", "fp32"):
        m.precision = prec
        with torch.no_grad():
            outs[prec] = m.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=5.0).double().cpu()
        assert torch.isfinite(outs[prec]).all()
    rel = float((outs["bf16x3"] - outs["fp32"]).norm() / outs["fp32"].norm())
    assert rel < 2e-4, rel
    m.precision = "bf16x3"
    plan = m.get_plan(2 * B, T, H, H, 51, gpu, mirrored=True)  # guidance runs the conditional and the null branch as one batch
    used = {fn.__name__ for fn, _, _ in plan.steps}
    assert "vmm_conv3x3_bf16x3" in used and "vmm_linattn_block_bf16x3" in used and "vmm_temporal_attention" in used


KW_HIRES = dict(dim=64, dim_mults=(1, 2, 4, 8), channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True,
                per_frame_cond=False)


def _rel(a, b):
    return float((a.double() - b.double()).norm() / b.double().norm())


def test_hires_22x192x192_matches_oracle(gpu, monkeypatch):
    """BASELINE configs[3] frames (22 x 192 x 192, Lagrangian widths, CNN-token conditioning) against the ORACLE itself, one sample (about
    a minute of CPU): the 12 x 12 tile grids, the 36864-pixel linear attention and the two-frame-tile temporal attention are not only
    self-consistent (the test above compares two instances of the same kernel templates) but right."""
    import videometamaterials_amd as vm
    from oracle import unet3d_oracle as uo
    torch.manual_seed(0)
    m = vm.Unet3D(**KW_HIRES).to(gpu).eval()
    B, T, H = 1, 22, 192
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, 3, T, H, H, generator=g)
    t = torch.randint(0, 256, (B,), generator=g)
    cond = torch.rand(B, 51, generator=g) * 2 - 1
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    # B = 1: the reference's SignalEmbedding squeezes the batch axis away too (vddp.py:571) and then fails, and the oracle restates that
    # faithfully; the embedding of ONE conditioning row is taken from a batch of two identical rows instead
    cnn = uo.signal_cnn
    monkeypatch.setattr(uo, "signal_cnn", lambda sd_, c: cnn(sd_, torch.cat([c, c]))[:1] if c.shape[0] == 1 else cnn(sd_, c))
    with torch.no_grad():
        want = uo.unet3d_forward(sd, uo.UnetCfg(**KW_HIRES), x, t, cond, torch.zeros(B, dtype=torch.bool))
        for prec, tol in (("bf16x3", 2e-4), ("fp32", 2e-5), ("bf16", 2e-2)):  # "bf16": the throughput mode this configuration is specified in (one pass)
            m.precision = prec
            got = m(x.to(gpu), t.to(gpu), cond=cond.to(gpu), null_cond_prob=0.0).cpu()
            assert _rel(got, want) < tol, (prec, _rel(got, want))
            m._plans.clear()


def test_hires_batch8_full_configuration(gpu, monkeypatch):
    """configs[3] at its full size: batch 8 per GPU, 22 x 192 x 192 (25.3 TFLOP, ~31 GB of plan memory), in ALL THREE arithmetic modes.
    Per mode: finite, bit-reproducible, and samples 0 and 5 of the batch equal to the same samples run alone at a stated bound (GroupNorm,
    attention and conditioning are per sample; tile / split decisions differ with the batch size: summation order only -- in the single-pass
    "bf16" mode a different summation order moves results by bf16 roundings of intermediate operands, hence its wider bound).  The batch-1
    evaluation of sample 5 is checked against the ORACLE in the same test, so the batch-8 output is tied to the oracle for that sample
    (within bound + tolerance) and not only to itself."""
    import videometamaterials_amd as vm
    from oracle import unet3d_oracle as uo
    torch.manual_seed(0)
    m = vm.Unet3D(**KW_HIRES).to(gpu).eval()
    B, T, H = 8, 22, 192
    g = torch.Generator().manual_seed(4)
    x = torch.randn(B, 3, T, H, H, generator=g)
    t = torch.randint(0, 256, (B,), generator=g)
    cond = torch.rand(B, 51, generator=g) * 2 - 1
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    cnn = uo.signal_cnn  # (B = 1 and SignalEmbedding's squeeze: see the oracle test above)
    monkeypatch.setattr(uo, "signal_cnn", lambda sd_, c: cnn(sd_, torch.cat([c, c]))[:1] if c.shape[0] == 1 else cnn(sd_, c))
    with torch.no_grad():
        want5 = uo.unet3d_forward(sd, uo.UnetCfg(**KW_HIRES), x[5:6], t[5:6], cond[5:6], torch.zeros(1, dtype=torch.bool))
    xg, tg, cg = x.to(gpu), t.to(gpu), cond.to(gpu)
    for prec, solo_bound, oracle_tol in (("bf16x3", 3e-5, 2e-4), ("fp32", 1e-5, 2e-5), ("bf16", 1e-2, 2e-2)):
        m.precision = prec
        with torch.no_grad():
            full = m(xg, tg, cond=cg, null_cond_prob=0.0).clone()
            assert torch.isfinite(full).all(), prec
            assert torch.equal(m(xg, tg, cond=cg, null_cond_prob=0.0), full), prec
            for i in (0, 5):
                solo = m(xg[i:i + 1], tg[i:i + 1], cond=cg[i:i + 1], null_cond_prob=0.0)
                assert _rel(solo, full[i:i + 1]) < solo_bound, (prec, i, _rel(solo, full[i:i + 1]))
                if i == 5:
                    assert _rel(solo.cpu(), want5) < oracle_tol, (prec, _rel(solo.cpu(), want5))
                    assert _rel(full[5:6].cpu(), want5) < oracle_tol + solo_bound, (prec, _rel(full[5:6].cpu(), want5))
        if prec == "bf16x3":
            plan = m.get_plan(B, T, H, H, 51, gpu)
            assert plan.arena_floats * 4 < 60e9
        m._plans.clear()
        torch.cuda.empty_cache()
