"""First contact with RCCL on a one-GPU box: ONE rank drives the whole N-rank control flow of the training step through the real
collective library -- (a) the native engine of include/vmm_dp.h (its own communicator, side stream and events behind the C ABI), (b)
torch.distributed's "nccl" backend (= RCCL on ROCm) with a one-rank process group.  A one-rank all-reduce is the identity, so both must
reproduce the plain single-process trainer (to the run-to-run noise of the atomically added gradients) while every bucket really goes
through ncclAllReduce on the side stream
(main.py:31-34, vddp.py:1449, 1629; SURVEY 8(b) last row, 8(e))."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

import helpers

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))
CFG = "lagr16"
STEPS = 3


def _inputs(step):
    kw, (B, T, H, W), cl = helpers.CONFIGS[CFG]
    g = torch.Generator().manual_seed(70 + step)
    x = torch.rand(B, 3, T, H, W, generator=g)
    cond = torch.rand(B, cl, generator=g) * 2 - 1
    t = torch.randint(0, 256, (B,), generator=g)
    noise = torch.randn(B, 3, T, H, W, generator=g)
    mask = (torch.rand(B, generator=g) < 0.25).to(torch.uint8)
    return x, cond, t, noise, mask


def _trainer(dev, **kw_tr):
    import videometamaterials_amd as vm
    from videometamaterials_amd.dp import DataParallelTrainer
    kw, (B, T, H, W), _ = helpers.CONFIGS[CFG]
    model = vm.Unet3D(**kw)
    model.load_state_dict(helpers.synth_state_dict(helpers.load_shapes(CFG)))
    diff = vm.GaussianDiffusion(model.to(dev), image_size=H, num_frames=T, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=256).to(dev)
    return DataParallelTrainer(diff, train_lr=1e-3, update_ema_every=2, step_start_ema=2, bucket_floats=200_000, **kw_tr)


def _run(tr, dev, timing=False):
    losses = []
    for step in range(STEPS):
        x, cond, t, noise, mask = (a.to(dev) for a in _inputs(step))
        if timing and step == STEPS - 1:
            tr._reducer.timing = True
        losses.append(float(tr.train_step(x, cond, t=t, noise=noise, mask=mask)))
    torch.cuda.synchronize()
    return losses, {k: v.detach().clone() for k, v in tr.unet.state_dict().items()}


def _same_training(l0, w0, l1, w1):
    """Two runs of the same three optimiser steps.  Not bit for bit: a few gradient kernels add with fp32 atomics, and Adam's m / sqrt(v)
    turns a last-bit difference of a tiny gradient into a visible step -- the same bounds as tests/test_gpu_dp2.py (an unreduced or doubly
    reduced gradient buffer is off by ~lr everywhere)."""
    for a, b in zip(l0, l1):
        assert abs(a - b) <= 1e-4 * abs(a), (l0, l1)
    for k in w0:
        d = (w0[k] - w1[k]).abs()
        assert float(d.max()) <= 2.0 * STEPS * 1e-3, k
        assert float(d.mean()) <= 1e-4, (k, float(d.mean()))


def test_native_engine_one_rank_equals_the_plain_trainer(gpu):
    from videometamaterials_amd.dp import RcclEngine, plan_buckets
    dev = torch.device("cuda:0")
    plain = _trainer(dev, engine="torch")
    assert plain.engine is None
    l0, w0 = _run(plain, dev)
    assert plain._reducer.launched == []  # a single rank without an engine exchanges nothing

    tr = _trainer(dev, engine="native")
    eng = tr.engine
    assert isinstance(eng, RcclEngine) and eng.world == 1 and eng.rank == 0
    assert eng.rccl_version >= 20000, eng.rccl_version  # ncclGetVersion of the library it bound
    chk = tr.rccl_selfcheck()
    assert chk["ok"] and chk["backend"].startswith("vmm_dp"), chk
    l1, w1 = _run(tr, dev, timing=True)
    _same_training(l0, w0, l1, w1)
    # every bucket of the plan went through the engine, in the order the marks predict
    pl = tr._plan
    want = plan_buckets(pl.pgrad_floats, [x for _, x in sorted(dict(pl.bwd_marks).items())], 200_000)
    assert tr._reducer.launched == want and len(want) >= 3, (tr._reducer.launched, want)
    assert want[-1][0] == 0 and sum(hi - lo for lo, hi in want) == pl.pgrad_floats  # the buckets tile the gradient buffer
    tm = tr._reducer.last_step_timing()
    assert tm is not None and tm["allreduce_ms"] > 0 and tm["backward_ms"] > 0 and 0 <= tm["overlap_frac"] <= 1, tm

    # the plain collectives of sharded sampling
    a = torch.arange(1000, device=dev, dtype=torch.float32)
    b = a.clone()
    eng.all_reduce(b)
    eng.broadcast(b, 0)
    (g,) = eng.all_gather(b)
    i64 = torch.tensor([7, -3], device=dev)
    eng.all_reduce(i64, "max")
    torch.cuda.synchronize()
    assert torch.equal(a, b) and torch.equal(a, g) and i64.tolist() == [7, -3]
    # a bucket may not be issued twice without the compute stream having been made to wait for it
    eng.allreduce_bucket_async(0)
    with pytest.raises(Exception, match="issued twice"):
        eng.allreduce_bucket_async(0)
    eng.wait_all()
    out = tr.sample_sharded(torch.zeros(1, helpers.CONFIGS[CFG][2], device=dev), guidance_scale=1.0, batch=1)
    assert out is not None and out.shape[0] == 1 and torch.isfinite(out).all()
    eng.close()
    eng.close()  # idempotent


def _nccl_worker(port, outdir):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), VMM_DP_FORCE="1")
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1)
    try:
        dev = torch.device("cuda:0")
        tr = _trainer(dev, engine="torch")
        assert tr._reducer is None and tr.world == 1
        chk = tr.rccl_selfcheck()
        losses, weights = _run(tr, dev, timing=True)
        assert tr._reducer.active and not tr._reducer.host_staged
        torch.save(dict(losses=losses, weights={k: v.cpu() for k, v in weights.items()}, buckets=list(tr._reducer.launched), check=chk,
                        timing=tr._reducer.last_step_timing()), os.path.join(outdir, "nccl.pt"))
    finally:
        dist.destroy_process_group()


def test_torch_nccl_backend_one_rank_equals_the_plain_trainer(gpu, tmp_path):
    ctx = mp.get_context("spawn")
    p = ctx.Process(target=_nccl_worker, args=(29900 + os.getpid() % 90, str(tmp_path)))
    p.start()
    p.join(600)
    if p.is_alive():
        p.kill()
        pytest.fail("the one-rank RCCL process group hung")
    assert p.exitcode == 0
    res = torch.load(os.path.join(str(tmp_path), "nccl.pt"))
    dev = torch.device("cuda:0")
    l0, w0 = _run(_trainer(dev, engine="torch"), dev)
    _same_training(l0, {k: v.cpu() for k, v in w0.items()}, res["losses"], res["weights"])
    assert len(res["buckets"]) >= 3 and res["buckets"][-1][0] == 0
    assert res["timing"] is not None and res["timing"]["allreduce_ms"] > 0
