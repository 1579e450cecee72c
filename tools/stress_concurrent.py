"""Bit-reproducibility of the sampler while several processes share the GPU (kernels of different processes interleave on the CUs, so a missing
fence / an order-dependent reduction that never shows in a process running alone has a chance to):
    python tools/stress_concurrent.py [processes = 3] [samples per process = 12] [config = lagr16 | any of tests/helpers.CONFIGS | full]
Every process draws the same seeded sample() repeatedly; all digests of all processes must be equal.  VMM_DISABLE=... narrows a mismatch down."""
import hashlib
import os
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def worker(rank, n, cfg, q):
    import helpers
    import videometamaterials_amd as vm
    dev = torch.device("cuda:0")
    if cfg == "full":  # the benchmark model (Lagrangian widths, 11 x 96 x 96), random init
        import bench
        kw, (B, T, H, W), cl = bench.LAGRANGIAN, (2, bench.T, bench.HW, bench.HW), 11
        torch.manual_seed(0)
        model = vm.Unet3D(**kw)
    else:
        kw, (B, T, H, W), cl = helpers.CONFIGS[cfg]
        model = vm.Unet3D(**kw)
        model.load_state_dict(helpers.synth_state_dict(helpers.load_shapes(cfg)))
    if os.environ.get("STRESS_PREC"):
        model.precision = os.environ["STRESS_PREC"]
    for a in os.environ.get("STRESS_OFF", "").split(","):  # model switches, e.g. use_fused_temporal
        if a:
            setattr(model, a, False)
    diff = vm.GaussianDiffusion(model.to(dev).eval(), image_size=H, num_frames=T, channels=3, timesteps=6, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=6).to(dev)
    cond = (torch.rand(2, cl, generator=torch.Generator().manual_seed(77)) * 2 - 1).to(dev)
    mode = os.environ.get("STRESS_MODE", "graph")  # graph | eager (the step's launch list, not captured) | torch (p_sample + torch RNG) | unet (one denoiser forward)
    diff.use_graph = {"graph": True, "eager": "eager", "torch": False, "unet": True, "step": True, "trace": True, "regions": True, "dump": True, "embed": True, "unet2": True, "train": True}[mode]
    w = float(os.environ.get("STRESS_W", "3.0"))
    g = torch.Generator().manual_seed(3)
    x0 = torch.randn(2, 3, T, H, W, generator=g).to(dev)
    t0 = torch.randint(0, 6, (2,), generator=g).to(dev)
    ec, en, nz = (torch.randn(2, 3, T, H, W, generator=g).to(dev) for _ in range(3))

    def step_only():  # the arithmetic of one ancestral step behind the denoiser: x0 prediction, exact quantile, posterior update (fixed inputs)
        x0p, ax0 = diff._predict_x0(x0, t0, ec, en, w, True)
        return diff._posterior(x0p, x0, t0, nz, diff._quantile(ax0), 2)

    if mode == "trace":
        # launch by launch: a checksum of the whole arena after every step of the guided plan; reports the first step whose checksum ever differs
        # from the first forward's (the kernel whose result depends on timing)
        with torch.no_grad():
            model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w)
        pl = model.get_plan(4, T, H, W, cl, dev, mirrored=True)
        iv = pl.arena.view(torch.int32)
        s_ = torch.cuda.current_stream().cuda_stream
        import ctypes
        first, bad = None, {}
        for i in range(n):
            sums = []
            for fn, args, what in pl.steps:
                rc = fn(*args, ctypes.c_void_p(s_))
                assert rc == 0, what
                sums.append(iv.sum(dtype=torch.int64))
            sums = torch.stack(sums).cpu().tolist()
            if first is None:
                first = sums
            else:
                for k, (a, b_) in enumerate(zip(first, sums)):
                    if a != b_:
                        key = f"{k}: {pl.steps[k][0].__name__}: {pl.steps[k][2]}"
                        bad[key] = bad.get(key, 0) + 1
                        break
        q.put((rank, ([str(sorted(bad.items()))], 0.0)))
        return
    if mode == "dump":  # the guided plan's launch list
        with torch.no_grad():
            model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w)
        pl = model.get_plan(4, T, H, W, cl, dev, mirrored=True)
        if rank == 0:
            for si, (fn, args, what) in enumerate(pl.steps):
                vals = []
                for a_ in args:
                    v = getattr(a_, "value", a_)
                    vals.append(hex(v) if isinstance(v, int) and v > (1 << 32) else v if isinstance(v, (int, float)) or v is None else type(a_).__name__)
                print(si, fn.__name__, "|", what, "|", vals, file=sys.stderr)
            print("arena", hex(pl.arena.data_ptr()), "wbuf", hex(pl.wbuf.data_ptr()), file=sys.stderr)
        q.put((rank, ([""], 0.0)))
        return
    if mode == "train":  # loss + every parameter gradient of one training step (fixed t / noise); atomics make the last bits order-dependent by design, so
        # the report is the largest deviation from the first step relative to the gradient's own scale (a glitch of the 9.8 kind is 1e-2 .. 1)
        model.train()
        model.train_precision = os.environ.get("STRESS_PREC", "bf16x3")
        xs = torch.rand(2, 3, T, H, W, generator=g).to(dev)
        nz_ = torch.randn(2, 3, T, H, W, generator=g).to(dev)
        ts = torch.tensor([1, 4], device=dev)
        ref, events = None, []
        for i in range(n):
            model.zero_grad(set_to_none=True)
            loss = diff.p_losses(xs * 2 - 1, ts, cond=cond, noise=nz_, null_cond_prob=0.0)
            (loss * float(os.environ.get("STRESS_LOSS_SCALE", "1"))).backward()  # (STRESS_PREC=fp16: 65536, the GradScaler's initial scale)
            torch.cuda.synchronize()
            gr = {k: p_.grad.detach().clone() for k, p_ in model.named_parameters() if p_.grad is not None}
            if ref is None:
                ref = gr
            else:
                devs = {k: float((v - ref[k]).abs().max() / ref[k].abs().max().clamp_min(1e-20)) for k, v in gr.items()}
                big = sorted(((d, k) for k, d in devs.items() if d > 1e-4 and float(ref[k].abs().max()) > 1e-9), reverse=True)
                if big:
                    events.append(f"step {i}: {len(big)} gradients off by > 1e-4 of their scale, largest " + ", ".join(f"{k} {d:.2g}" for d, k in big[:5]))
        q.put((rank, ([f"{len(events)} events in {n} steps; " + " || ".join(events[:4])], 0.0)))
        return
    if mode == "unet2":  # two different inputs alternating: a kernel that sees the PREVIOUS forward's data anywhere (not only in an in-place update) shows
        x1 = torch.randn(2, 3, T, H, W, generator=g).to(dev)
        t1 = torch.randint(0, 6, (2,), generator=g).to(dev)
        c1 = (torch.rand(2, cl, generator=g) * 2 - 1).to(dev)
        ins = [(x0, t0, cond), (x1, t1, c1)]
        refs, odd, worst = [None, None], 0, 0.0
        for i in range(n):
            a_ = ins[i & 1]
            with torch.no_grad():
                y = model.forward_with_guidance_scale(a_[0], a_[1], cond=a_[2], guidance_scale=w).cpu()
            if refs[i & 1] is None:
                refs[i & 1] = y
            elif not torch.equal(y, refs[i & 1]):
                odd += 1
                worst = max(worst, float((y - refs[i & 1]).abs().max()))
        q.put((rank, ([f"odd {odd} of {n}, max abs diff {worst:.3g}"], 0.0)))
        return
    if mode == "embed":  # only the embedding chain (the launches before the first feature-map kernel), all early regions checksummed
        with torch.no_grad():
            model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w)
        pl = model.get_plan(4, T, H, W, cl, dev, mirrored=True)
        nst = int(os.environ.get("STRESS_EMBED_STEPS", "0")) or next(i for i, st in enumerate(pl.steps) if st[0].__name__ == "vmm_ncthw_to_rows")
        iv = pl.arena.view(torch.int32)
        import ctypes
        s_ = ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
        regs = [(o, m) for o, m in pl.alloc_log[:60]]
        first, bad = None, {}
        for i in range(n):
            for fn, args, what in pl.steps[:nst]:
                assert fn(*args, s_) == 0, what
            sums = torch.stack([iv[o:o + m].sum(dtype=torch.int64) for o, m in regs]).cpu().tolist()
            if first is None:
                first = sums
            else:
                diffs = [k for k, (a, b_) in enumerate(zip(first, sums)) if a != b_]
                if diffs:
                    key = "regions " + ",".join(f"#{k}(n {regs[k][1]})" for k in diffs[:6])
                    bad[key] = bad.get(key, 0) + 1
        q.put((rank, ([str(sorted(bad.items()))], 0.0)))
        return
    if mode == "regions":
        # VMM_KEEP_ALL=1: no arena slot is reused, so after a forward every allocation still holds what its producer wrote; the earliest allocation
        # whose checksum ever differs from the first forward's points at the first kernel whose result depended on timing (no per-launch syncs)
        assert os.environ.get("VMM_KEEP_ALL"), "run with VMM_KEEP_ALL=1"
        with torch.no_grad():
            model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w)
        pl = model.get_plan(4, T, H, W, cl, dev, mirrored=True)
        iv = pl.arena.view(torch.int32)
        out_ptr = pl.out.data_ptr()  # the static output slot differs whenever anything does: not a lead
        skip = {k for k, (o, m) in enumerate(pl.alloc_log) if pl.arena.data_ptr() + 4 * o <= out_ptr < pl.arena.data_ptr() + 4 * (o + m)}

        def users_of(k):  # launches that take region k's address as an argument (the first is its producer)
            ptr = pl.arena.data_ptr() + 4 * pl.alloc_log[k][0]
            found = []
            for si, (fn, args, what) in enumerate(pl.steps):
                flat = []
                for a_ in args:
                    flat.append(getattr(a_, "value", a_))
                    if hasattr(a_, "_fields_"):
                        flat += [getattr(a_, f[0]) for f in a_._fields_]
                    if hasattr(a_, "contents") and hasattr(a_.contents, "_fields_"):
                        flat += [getattr(a_.contents, f[0]) for f in a_.contents._fields_]
                if any(isinstance(v, int) and v == ptr for v in flat):
                    found.append(f"{si}:{fn.__name__}:{what}")
            return found

        first, keep, bad = None, None, {}
        for i in range(n):
            with torch.no_grad():
                model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w)
            sums = torch.stack([iv[o:o + m].sum(dtype=torch.int64) for o, m in pl.alloc_log]).cpu().tolist()
            if first is None:
                first, keep = sums, pl.arena.clone()  # (small test models: the whole arena of the first forward is kept for element-level diffs)
            else:
                diffs = [k for k, (a, b_) in enumerate(zip(first, sums)) if a != b_ and k not in skip]
                if diffs and len(bad) < 8:
                    k0 = diffs[0]
                    o, m = pl.alloc_log[k0]
                    dm = (pl.arena[o:o + m] != keep[o:o + m]).nonzero().flatten()
                    j0 = int(dm[0])
                    key = (f"{len(diffs)} regions differ; earliest allocations: " + ", ".join(f"#{k} (n {pl.alloc_log[k][1]})" for k in diffs[:4])
                           + f"; #{k0} used by " + " | ".join(users_of(k0)[:4]) + f"; {dm.numel()} elements of it differ, first at {j0}..{int(dm[-1])}: now "
                           + str([round(v, 6) for v in pl.arena[o + j0:o + j0 + 4].tolist()]) + " was " + str([round(v, 6) for v in keep[o + j0:o + j0 + 4].tolist()]))
                    bad[key] = bad.get(key, 0) + 1
        q.put((rank, ([str(sorted(bad.items()))], 0.0)))
        return
    out, ref, worst = [], None, 0.0
    for i in range(n):
        torch.manual_seed(500)
        with torch.no_grad():
            x = (model.forward_with_guidance_scale(x0, t0, cond=cond, guidance_scale=w) if mode == "unet" else step_only() if mode == "step"
                 else diff.sample(cond=cond, guidance_scale=w, batch_size=2))
        torch.cuda.synchronize()
        x = x.cpu()
        d = hashlib.sha1(x.numpy().tobytes()).hexdigest()[:12]
        if ref is None:
            ref = (d, x)
        elif d != ref[0]:
            worst = max(worst, float((x - ref[1]).abs().max()))
        out.append(d)
    q.put((rank, (out, worst)))


if __name__ == "__main__":
    P = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
    cfg = sys.argv[3] if len(sys.argv) > 3 else "lagr16"
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=worker, args=(r, n, cfg, q)) for r in range(P)]
    for p in procs:
        p.start()
    res = dict(q.get(timeout=1200) for _ in procs)
    for p in procs:
        p.join(timeout=60)
    if os.environ.get("STRESS_MODE") in ("trace", "regions", "dump", "embed", "unet2", "train"):
        for r, v in sorted(res.items()):
            print("process", r, "first differing step (count):", v[0][0])
        # train: a gradient off by more than 1e-4 of its scale in any step of any process is a failure (tests/test_gpu_concurrency.py)
        sys.exit(1 if os.environ.get("STRESS_MODE") == "train" and (len(res) < P or any(not v[0][0].startswith("0 events") for v in res.values())) else 0)
    digests = sorted({d for v in res.values() for d in v[0]})
    from collections import Counter
    cnt = Counter(d for v in res.values() for d in v[0])
    print({"mode": os.environ.get("STRESS_MODE", "graph"), "disable": os.environ.get("VMM_DISABLE", ""), "prec": os.environ.get("STRESS_PREC", ""), "off": os.environ.get("STRESS_OFF", ""), "processes": P, "samples_per_process": n, "config": cfg,
           "distinct_digests": len(digests), "majority": cnt.most_common(1)[0], "odd_samples": sum(cnt.values()) - cnt.most_common(1)[0][1],
           "max_abs_diff_inside_a_process": max(v[1] for v in res.values())})
    sys.exit(0 if len(digests) == 1 else 1)
