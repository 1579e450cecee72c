"""precision = "bf16" with bf16 STORAGE of the feature maps (BASELINE.json configs[3]: "bf16 ... HBM-bound conv stress"): the two upper levels' maps
live in HBM as bf16, every kernel that touches them runs an instance templated on the element type of its activation pointers (the arithmetic is the
single-pass mode's: bf16-rounded operands on the matrix cores, fp32 accumulation, norms / softmax / residual sums in fp32, ONE rounding per stored
element).  Stated tolerance of the mode: 2e-2 relative on the denoiser output (tests/test_gpu_hires.py checks it against the oracle at 22 x 192 x 192)."""
import os

import numpy as np
import pytest
import torch

import helpers
from test_gpu_unet import make_model

pytestmark = pytest.mark.gpu


def _run(model, x, t, cond, w=None):
    with torch.no_grad():
        if w is None:
            return model(x, t, cond=cond, null_cond_prob=0.0).float().cpu()
        return model.forward_with_guidance_scale(x, t, cond=cond, guidance_scale=w).float().cpu()


@pytest.mark.parametrize("cfg_name", ["hires64t22", "lagr64", "circ64"])
def test_bf16_storage_matches_fp32_storage_and_reference(gpu, cfg_name, monkeypatch):
    """Same weights, same inputs: bf16-stored maps against (a) the same single-pass arithmetic on fp32-stored maps -- the difference is one bf16
    rounding per stored element of the two upper levels --, (b) the reference's golden output, inside the mode's stated 2e-2; the plan contains
    the bf16-storage kernels and is smaller."""
    kw, (B, T, H, W), _ = helpers.CONFIGS[cfg_name]
    x, t, cond = (v.to(gpu) for v in helpers.synth_inputs(cfg_name))
    gold = np.load(os.path.join(helpers.GOLDEN_DIR, f"unet_{cfg_name}.npz"))
    outs, plans = {}, {}
    for storage in (True, False):
        m = make_model(cfg_name, gpu, precision="bf16")
        m.bf16_storage = storage
        outs[storage] = (_run(m, x, t, cond), _run(m, x, t, cond, 5.0))
        pl = m.get_plan(B, T, H, W, cond.shape[-1], gpu)
        plans[storage] = ([fn.__name__ for fn, _, _ in pl.steps], pl.arena_floats)
    for i, key in enumerate(("eps_cond", "eps_w5")):
        want = torch.from_numpy(gold[key])
        e16, e32 = helpers.rel_err(outs[True][i], want), helpers.rel_err(outs[False][i], want)
        tol = 2e-2 if key == "eps_cond" else 5e-2  # (guidance at w = 5 extrapolates the difference of two outputs: the existing single-pass test's bounds)
        print(f"{cfg_name} {key}: bf16 storage {e16:.3e}, fp32 storage {e32:.3e}, between {helpers.rel_err(outs[True][i], outs[False][i]):.3e}")
        assert e32 < tol and e16 < tol, (key, e16, e32)
        assert helpers.rel_err(outs[True][i], outs[False][i]) < tol, (key, helpers.rel_err(outs[True][i], outs[False][i]))
        assert not torch.equal(outs[True][i], outs[False][i])
    names16, floats16 = plans[True]
    names32, floats32 = plans[False]
    print(cfg_name, "a16 kernels:", sorted({n for n in names16 if n.endswith("_a16")}), "conversions:", names16.count("vmm_convert_act"), "arena", floats16, floats32)
    assert any(n.endswith("_a16") for n in names16) and not any(n.endswith("_a16") for n in names32)
    assert floats16 < floats32


def test_bf16_storage_plan_has_no_conversions_at_the_benchmark_shape(gpu):
    """configs[3]'s wiring at its frame count: every kernel of the two upper levels has a bf16-storage instance -- the plan holds no fp32 <-> bf16
    conversion launch (they exist for configurations outside the instances' envelopes and as a development aid, VMM_A16_OPS)."""
    import videometamaterials_amd as vm
    from test_gpu_hires import KW_HIRES
    m = vm.Unet3D(**KW_HIRES).to(gpu).eval()
    m.precision = "bf16"
    pl = m.get_plan(2, 22, 64, 64, 51, gpu)
    names = [fn.__name__ for fn, _, _ in pl.steps]
    assert names.count("vmm_convert_act") == 0, [w for (fn, _, w) in pl.steps if fn.__name__ == "vmm_convert_act"]
