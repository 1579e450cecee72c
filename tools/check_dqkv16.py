"""Digest of the parameter gradients of ONE training forward + backward of the Lagrangian configuration (batch 2, fixed inputs) in a single-pass mode:

    python tools/check_dqkv16.py fp16                                             # product library: 16-bit rows of the qkv-row gradient
    VMM_DQKV16=0 VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so python tools/check_dqkv16.py fp16
                                                                                  # (python tools/build_ab.py temporal_block_bwd linattn_block_bwd qkv_bwd -DVMM_DQKV16=0)

The two digests must be EQUAL: the 16-bit hand-over moves the operand rounding of the to_qkv backward into the producer's store and changes no bit
(csrc/vmm_common.h, VMM_DQKV16).  Also prints the time of the step over a few repetitions."""
import hashlib
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402

prec = sys.argv[1] if len(sys.argv) > 1 else "fp16"
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev)
model.train_precision = prec
B = 2
g = torch.Generator().manual_seed(5)
x = torch.randn(B, 3, bench.T, bench.HW, bench.HW, generator=g).to(dev)
t = torch.tensor([17, 201][:B], device=dev)
cond = torch.rand(B, 11, generator=g).to(dev)
pl = model.get_plan(B, bench.T, bench.HW, bench.HW, 11, dev, training=True)
pl.x_in.copy_(x); pl.time_in.copy_(t); pl.cond_in.copy_(cond); pl.mask_in.zero_()
dout = (torch.randn(pl.out.shape, generator=g) * 1e-2).to(dev)
digests = []
for rep in range(2):
    pl.launch()  # (a fresh forward per repetition: the backward re-uses activation slots for gradients)
    pl.backward(dout)
    torch.cuda.synchronize()
    flat = pl.pgrad.cpu().numpy()
    digests.append({k: hashlib.sha256(flat[o:o + n].tobytes()).hexdigest()[:12] for k, (o, n) in pl.param_slices.items()})
    print(prec, "finite", bool(torch.isfinite(pl.pgrad).all()), "norm %.6e" % float(pl.pgrad.double().norm()), "arena GB %.3f" % (pl.arena.numel() * 4 / 2 ** 30),
          "lib", os.environ.get("VMM_LIB_PATH", "default"))
# Parameters whose gradient is the same bits in both repetitions (kernels that combine partial sums with fp32 atomics -- the 4 x 4 / 7 x 7 weight
# gradients, bias column sums, the tiny dense layers -- are not run-to-run reproducible)
stable = sorted(k for k in digests[0] if digests[0][k] == digests[1][k])
print(prec, "reproducible parameters", len(stable), "of", len(digests[0]))
if os.environ.get("VMM_DIGEST_NAMES"):
    import collections
    bad = [k for k in digests[0] if k not in stable]
    kinds = collections.Counter(".".join(k.split(".")[-2:]) if not k.split(".")[-2].isdigit() else k.split(".")[-1] for k in bad)
    print(prec, "not reproducible, by parameter kind:", dict(kinds))
    print(prec, "reproducible, by parameter kind:", dict(collections.Counter(".".join(k.split(".")[-2:]) for k in stable)))
out = os.environ.get("VMM_DIGEST_OUT")
if out:
    import json
    json.dump({k: digests[0][k] for k in stable}, open(out, "w"))
other = os.environ.get("VMM_DIGEST_CMP")
if other:
    import json
    ref = json.load(open(other))
    common = [k for k in stable if k in ref]
    diff = [k for k in common if ref[k] != digests[0][k]]
    qkv = [k for k in common if "to_qkv" in k]
    print(prec, "compared", len(common), "reproducible parameters with", other, "-> different:", len(diff), diff[:8], "| to_qkv weights compared:", len(qkv))
