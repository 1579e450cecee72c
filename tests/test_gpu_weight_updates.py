"""Everything that changes parameters behind the back of a cached plan must reach the packed operand copies the kernels read:
optimiser / EMA launches that write through raw pointers, load_state_dict, in-place torch ops (Trainer: ema_model.sample() every
save_and_sample_every steps, vddp.py:1734/1826).  Plus the EMA arithmetic itself (vddp.py:116-129) and Adam state across batch shapes."""
import copy

import pytest
import torch

import helpers

pytestmark = pytest.mark.gpu
CFG = "lagr16"


def _build(dev, loss_type="l2"):
    import videometamaterials_amd as vm
    kw, (B, T, H, W), _ = helpers.CONFIGS[CFG]
    sd = helpers.synth_state_dict(helpers.load_shapes(CFG))
    model = vm.Unet3D(**kw)
    model.load_state_dict(sd)
    diff = vm.GaussianDiffusion(model.to(dev), image_size=H, num_frames=T, channels=3, timesteps=8, loss_type=loss_type, use_dynamic_thres=True,
                                sampling_timesteps=8).to(dev)
    return kw, sd, model, diff


def _oracle_sample(kw, sd, cond, w, x_T, noises):
    from oracle import diffusion_oracle as do
    from oracle import unet3d_oracle as uo
    cfg = uo.UnetCfg(**kw)
    sch = do.schedule_buffers(8)
    img = x_T.clone()
    with torch.no_grad():
        for j, i in enumerate(reversed(range(8))):
            t = torch.full((img.shape[0],), i, dtype=torch.long)
            img = do.p_sample_step(sch, lambda a, b: uo.unet3d_guided(sd, cfg, a, b, cond, w), img, t, noises[j])
    return (img + 1) * 0.5


@pytest.mark.parametrize("graph", [True, False])
def test_sample_after_weight_update_uses_the_new_weights(gpu, graph):
    """sample -> change the weights (three different ways) -> sample: each result must match the oracle run on the weights of that
    moment.  The hipGraph sampler keeps its captured step; only the packed weight buffers (static addresses) are rewritten."""
    kw, sd, model, diff = _build(gpu)
    diff.use_graph = graph
    _, (B, T, H, W), cl = helpers.CONFIGS[CFG]
    g = torch.Generator().manual_seed(8)
    cond = torch.rand(B, cl, generator=g) * 2 - 1
    x_T = torch.randn(B, 3, T, H, W, generator=g)
    noises = [torch.randn(B, 3, T, H, W, generator=g) for _ in range(8)]

    def run():
        if graph:  # the graphed stepper draws its own noise: compare through a seeded device RNG instead of injected tensors
            torch.manual_seed(123)
            return diff.p_sample_loop((B, 3, T, H, W), cond=cond.to(gpu), guidance_scale=5.0, x_T=x_T).cpu()
        return diff.p_sample_loop((B, 3, T, H, W), cond=cond.to(gpu), guidance_scale=5.0, x_T=x_T, noises=noises).cpu()

    first = run()  # (graph: the capture happens here; it consumes no randomness -- the step noise is generated in the kernel from a per-call key)
    if not graph:
        assert helpers.rel_err(first, _oracle_sample(kw, sd, cond, 5.0, x_T, noises)) < 1e-3
    assert torch.equal(run(), first)  # nothing changed: same result, and (graph) the same captured step
    states = []
    # (1) in-place torch update, as torch.optim does
    with torch.no_grad():
        for n_, p in model.named_parameters():
            if n_.endswith("proj.weight"):
                p.mul_(1.05)
    states.append(("in-place", {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    out1 = run()
    # (2) load_state_dict at the GaussianDiffusion level (Trainer.load)
    sd2 = {("denoise_fn." + k): helpers.synth_tensor(k, tuple(v.shape), 7) for k, v in sd.items()}
    full = {k: v for k, v in diff.state_dict().items() if not k.startswith("denoise_fn.")}
    full.update(sd2)
    diff.load_state_dict(full)
    states.append(("load_state_dict", {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    out2 = run()
    # (3) raw-pointer update: vmm_ema_step copying another model's weights into this one (what DataParallelTrainer's EMA model sees)
    other = copy.deepcopy(diff)
    with torch.no_grad():
        for p in other.parameters():
            p.add_(0.01 * torch.randn_like(p))
    from videometamaterials_amd.dp import DataParallelTrainer
    tr = DataParallelTrainer(other, update_ema_every=1)
    tr.ema_model = diff  # the model under test plays the EMA copy
    x = torch.rand(B, 3, T, H, W, generator=g).to(gpu)
    tr.train_step(x, cond.to(gpu))  # Adam on `other`, then EMA copy (step < step_start_ema) into `diff` through raw pointers
    states.append(("raw-pointer EMA copy", {k: v.detach().cpu().clone() for k, v in model.state_dict().items()}))
    assert torch.equal(model.state_dict()["init_conv.weight"], other.denoise_fn.state_dict()["init_conv.weight"])
    out3 = run()
    outs = [out1, out2, out3]
    for (what, st), out in zip(states, outs):
        assert not torch.equal(out, first), what
        if not graph:
            assert helpers.rel_err(out, _oracle_sample(kw, st, cond, 5.0, x_T, noises)) < 1e-3, what
    if graph:
        assert len(diff._graph_cache) == 1 and next(iter(diff._graph_cache.values())).graph is not None
        # the captured step and its own launch list run eagerly agree on every weight state (same seed -> same key of the in-kernel step
        # noise): checked on the last one
        diff._graph_cache.clear()
        diff.use_graph = "eager"
        torch.manual_seed(123)
        eager = diff.p_sample_loop((B, 3, T, H, W), cond=cond.to(gpu), guidance_scale=5.0, x_T=x_T).cpu()
        assert next(iter(diff._graph_cache.values())).graph is None
        assert torch.equal(eager, out3)


def test_ema_decay_branch(gpu):
    """vddp.py:121-129: past step_start_ema the EMA weights follow old * beta + (1 - beta) * new (parameters only); before, a copy."""
    from videometamaterials_amd.dp import DataParallelTrainer
    kw, sd, model, diff = _build(gpu)
    _, (B, T, H, W), cl = helpers.CONFIGS[CFG]
    g = torch.Generator().manual_seed(9)
    x = torch.rand(B, 3, T, H, W, generator=g).to(gpu)
    cond = (torch.rand(B, cl, generator=g) * 2 - 1).to(gpu)
    tr = DataParallelTrainer(diff, train_lr=1e-2, ema_decay=0.995, step_start_ema=1, update_ema_every=1)
    tr.train_step(x, cond)                                   # Trainer.step 0 < 1: reset -> EMA == weights
    w1 = {k: v.detach().clone() for k, v in model.named_parameters()}
    e1 = {k: v.detach().clone() for k, v in tr.ema_model.denoise_fn.named_parameters()}
    assert all(torch.equal(w1[k], e1[k]) for k in w1)
    tr.train_step(x, cond)                                   # Trainer.step 1: lerp
    w2 = dict(model.named_parameters())
    e2 = dict(tr.ema_model.denoise_fn.named_parameters())
    moved = 0
    for k in w1:
        want = e1[k] * 0.995 + (1 - 0.995) * w2[k].detach()  # update_average, vddp.py:126-129
        assert torch.allclose(e2[k].detach(), want, rtol=0, atol=2e-7 * float(want.abs().max()) + 1e-12), k
        moved += int(not torch.equal(w2[k].detach(), w1[k]))
    assert moved > 300
    tr.update_ema_every = 10                                  # Trainer.step 2: not a multiple of 10, no EMA update at all
    tr.train_step(x, cond)
    assert all(torch.equal(e2[k].detach(), dict(tr.ema_model.denoise_fn.named_parameters())[k].detach()) for k in e2)


def test_adam_state_survives_a_batch_shape_change(gpu):
    """The reference DataLoader has no drop_last: the last batch of an epoch is shorter.  Adam's moments belong to the parameters, not
    to the plan of one batch shape; checked against torch.optim.Adam fed with the engine's own gradients."""
    from videometamaterials_amd.dp import DataParallelTrainer
    kw, sd, model, diff = _build(gpu)
    _, (B, T, H, W), cl = helpers.CONFIGS[CFG]
    tr = DataParallelTrainer(diff, train_lr=1e-3)
    names = ["init_conv.weight", "downs.1.0.block1.proj.weight", "mid_temporal_attn.fn.fn.fn.to_qkv.weight", "final_conv.1.bias"]
    ref = {k: dict(model.named_parameters())[k].detach().clone().requires_grad_(True) for k in names}
    opt = torch.optim.Adam(list(ref.values()), lr=1e-3)
    g = torch.Generator().manual_seed(10)
    for step, b in enumerate((B, B, 1, B, 1)):
        x = torch.rand(b, 3, T, H, W, generator=g).to(gpu)
        cond = (torch.rand(b, cl, generator=g) * 2 - 1).to(gpu)
        tr.train_step(x, cond)
        pl = tr._plan
        for k in names:
            o, n = pl.param_slices[k]
            ref[k].grad = pl.pgrad[o:o + n].view(ref[k].shape).clone()
        opt.step()
        live = dict(model.named_parameters())
        for k in names:
            assert torch.allclose(live[k].detach(), ref[k].detach(), rtol=0, atol=3e-6), (step, k, float((live[k] - ref[k]).abs().max()))
    assert len({id(p) for p in model._plans.values() if p.training}) == 2  # two training plans shared one optimiser state
