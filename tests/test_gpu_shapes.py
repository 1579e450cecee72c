"""Shape sweep of the whole denoiser on the MI355X against the oracle: frame sizes that are not multiples of the kernels' tiles, non-square
frames, one frame, more frames than the fused temporal kernels' slots, one sample, odd batch sizes, frames that shrink to 1 x 1 at the
deepest level -- the shapes the reference accepts (any H, W divisible by 8; vddp.py:730-821) beyond the ones the named configs pin with
goldens of the real reference.  Forward in both arithmetic modes, guidance, and the training gradients for a subset.  Weights are the
deterministic synthetic ones of tests/helpers.py, shaped from the model's own parameter tree."""
import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
TOL = 1e-3

LAGR = dict(channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True, per_frame_cond=True, cond_bias=True)
CNN = dict(channels=3, cond_attention="self-stacked", cond_attention_tokens=16, use_temporal_attention_cond=True, per_frame_cond=False)
PLAIN = dict(channels=1)

# (id, Unet3D kwargs, (B, T, H, W), cond length, also check gradients)
CASES = [
    ("min-8x8-one-frame", dict(dim=16, **PLAIN), (1, 1, 8, 8), 51, True),
    ("lagr16-24x40", dict(dim=16, **LAGR), (3, 11, 24, 40), 11, False),
    ("lagr16-40x24", dict(dim=16, **LAGR), (1, 11, 40, 24), 11, False),
    ("cnn16-T17-16x16", dict(dim=16, **CNN), (2, 17, 16, 16), 51, True),
    ("cnn16-T33-8x16", dict(dim=16, **CNN), (2, 33, 8, 16), 51, False),
    ("plain16-T2-56x8", dict(dim=16, **PLAIN), (5, 2, 56, 8), 40, False),
    ("lagr64-8x8", dict(dim=64, **LAGR), (2, 11, 8, 8), 11, True),
    ("lagr64-48x80", dict(dim=64, **LAGR), (1, 11, 48, 80), 11, False),
    ("cnn64-T5-24x72", dict(dim=64, **CNN), (3, 5, 24, 72), 51, False),
    ("cnn64-T16-16x16", dict(dim=64, **CNN), (2, 16, 16, 16), 51, False),
    ("lagr32-mults124-40x40", dict(dim=32, dim_mults=(1, 2, 4), **LAGR), (2, 11, 40, 40), 11, True),
    ("cross64-T11-24x8", dict(dim=64, channels=3, cond_attention="cross-attention", cond_attention_tokens=11, use_temporal_attention_cond=True,
                              per_frame_cond=False), (2, 11, 24, 8), 51, False),
    ("gru64-8x8", dict(dim=64, channels=3, cond_attention="self-stacked", cond_attention_tokens=40, use_temporal_attention_cond=True, per_frame_cond=False,
                       cond_att_GRU=True), (2, 3, 8, 8), 40, True),
    # square frames (GaussianDiffusion.image_size) of awkward sizes: also through the training step
    ("lagr16-40x40-B3", dict(dim=16, **LAGR), (3, 11, 40, 40), 11, True),
    ("cnn64-T5-24x24-B3", dict(dim=64, **CNN), (3, 5, 24, 24), 51, True),
    ("cross64-T11-24x24", dict(dim=64, channels=3, cond_attention="cross-attention", cond_attention_tokens=11, use_temporal_attention_cond=True,
                               per_frame_cond=False), (2, 11, 24, 24), 51, True),
]


def _build(kw, dev, precision):
    import videometamaterials_amd as vm
    m = vm.Unet3D(**kw)
    shapes = {k: tuple(v.shape) for k, v in m.state_dict().items()}
    sd = helpers.synth_state_dict(shapes)
    m.load_state_dict(sd, strict=True)
    m.precision = precision
    return m.to(dev), sd


def _inputs(kw, shape, cond_len):
    B, T, H, W = shape
    g = torch.Generator().manual_seed(B * 1000 + T * 100 + H + W)
    x = torch.randn(B, kw["channels"], T, H, W, generator=g)
    t = torch.randint(0, 256, (B,), generator=g)
    cond = torch.rand(B, cond_len, generator=g) * 2 - 1
    return x, t, cond


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", CASES, ids=[c[0] for c in CASES])
def test_forward_over_shapes_matches_oracle(gpu, case, precision):
    from oracle import unet3d_oracle as uo
    _, kw, shape, cond_len, _ = case
    model, sd = _build(kw, gpu, precision)
    model.eval()
    x, t, cond = _inputs(kw, shape, cond_len)
    B = shape[0]
    cfg = uo.UnetCfg(**kw)
    with torch.no_grad():
        for mask_val in (False, True):
            want = uo.unet3d_forward(sd, cfg, x, t, cond, torch.full((B,), mask_val))
            got = model(x.to(gpu), t.to(gpu), cond=cond.to(gpu), null_cond_prob=1.0 if mask_val else 0.0).cpu()
            assert got.shape == want.shape
            err = helpers.rel_err(got, want)
            assert err < TOL, f"{case[0]} {precision} null={mask_val}: rel {err:.3e}"
        want = uo.unet3d_guided(sd, cfg, x, t, cond, 3.0)
        got = model.forward_with_guidance_scale(x.to(gpu), t.to(gpu), cond=cond.to(gpu), guidance_scale=3.0).cpu()
        err = helpers.rel_err(got, want)
        assert err < TOL, f"{case[0]} {precision} guided: rel {err:.3e}"


@pytest.mark.parametrize("precision", ["fp32", "bf16x3"])
@pytest.mark.parametrize("case", [c for c in CASES if c[4]], ids=[c[0] for c in CASES if c[4]])
def test_gradients_over_shapes_match_oracle(gpu, case, precision):
    """l2 loss (smooth: with l1 the loss gradient is a sign and last-bit forward differences flip some of them) through the oracle's autograd."""
    import videometamaterials_amd as vm
    from oracle import diffusion_oracle as do
    from oracle import unet3d_oracle as uo
    _, kw, shape, cond_len, _ = case
    B, T, H, W = shape
    if H != W:
        pytest.skip("GaussianDiffusion takes square frames (image_size)")
    model, sd = _build(kw, gpu, "bf16x3")
    model.train_precision = precision
    diff = vm.GaussianDiffusion(model, image_size=H, num_frames=T, channels=kw["channels"], timesteps=256, loss_type="l2", sampling_timesteps=256).to(gpu)
    x, t, cond = _inputs(kw, shape, cond_len)
    g = torch.Generator().manual_seed(9)
    x0 = torch.rand(x.shape, generator=g) * 2 - 1
    noise = torch.randn(x.shape, generator=g)
    cfg = uo.UnetCfg(**kw)
    sdg = {k: v.clone().requires_grad_(not k.endswith("freqs")) for k, v in sd.items()}
    want_loss = do.p_losses(do.schedule_buffers(256), lambda a, b: uo.unet3d_forward(sdg, cfg, a, b, cond, torch.zeros(B, dtype=torch.bool)), x0, t, noise,
                            loss_type="l2")
    want_loss.backward()
    loss = diff.p_losses(x0.to(gpu), t.to(gpu), cond=cond.to(gpu), noise=noise.to(gpu), null_cond_prob=0.0)
    loss.backward()
    assert abs(float(loss.detach()) - float(want_loss.detach())) < 1e-4 * abs(float(want_loss.detach()))
    typical = max(float(v.grad.double().norm()) for v in sdg.values() if v.grad is not None)
    bad = []
    for k, p in model.named_parameters():
        w = sdg[model._ref_key(k)].grad
        if w is None or float(w.double().norm()) < 1e-9 * typical:
            if p.grad is not None and float(p.grad.double().norm()) > 1e-6 * typical:
                bad.append((k, "expected no gradient"))
            continue
        if p.grad is None:
            bad.append((k, "missing"))
            continue
        err = helpers.rel_err(p.grad.cpu(), w)
        if err > TOL:
            bad.append((k, f"{err:.3e}"))
    assert not bad, f"{case[0]} {precision}: {len(bad)} gradients off: {bad[:20]}"
