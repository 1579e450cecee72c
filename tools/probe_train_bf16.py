import os, sys, json, time
import numpy as np, torch
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import helpers
from test_gpu_train import _setup, _oracle_grads
gpu = torch.device("cuda:0")
ref = json.load(open(os.path.join(helpers.GOLDEN_DIR, "autocast_lagr16.json")))
for cfg in ("lagr16", "lagr64"):
    for prec in ("bf16x3", "bf16"):
        kw, sd, model, diff = _setup(cfg, gpu)
        model.train_precision = prec
        _, (B, T, H, W), _ = helpers.CONFIGS[cfg]
        _, t, cond = helpers.synth_inputs(cfg)
        g = torch.Generator().manual_seed(7)
        x0 = torch.rand((B, 3, T, H, W), generator=g) * 2 - 1
        noise = torch.randn((B, 3, T, H, W), generator=g)
        _, want = _oracle_grads(cfg, kw, sd, x0, t, cond, noise)
        loss = diff.p_losses(x0.to(gpu), t.to(gpu), cond=cond.to(gpu), noise=noise.to(gpu), null_cond_prob=0.0)
        loss.backward()
        got = {model._ref_key(k): p.grad for k, p in model.named_parameters()}
        vals = np.array([float((got[k].double().cpu() - w.double()).norm() / w.double().norm()) for k, w in want.items()
                         if w is not None and float(w.double().norm()) > 0 and got.get(k) is not None])
        print(cfg, prec, "median %.2e p90 %.2e max %.2e" % (np.median(vals), np.percentile(vals, 90), vals.max()), "| ref fp16 median %.2e p90 %.2e, bf16 median %.2e p90 %.2e max %.2e" % (ref["fp16"]["median"], ref["fp16"]["p90"], ref["bf16"]["median"], ref["bf16"]["p90"], ref["bf16"]["max"]), flush=True)
