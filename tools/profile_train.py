"""Training-step workload for rocprofv3 (per-kernel breakdown of one optimisation step of the Lagrangian configuration, batch 4):

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_train -o train -- \
        python $GRAFT_REPO_ROOT/tools/profile_train.py [fp32|bf16x3] [steps]
"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import videometamaterials_amd as vm  # noqa: E402

precision = sys.argv[1] if len(sys.argv) > 1 else "fp32"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev)
diff = vm.GaussianDiffusion(model, image_size=bench.HW, num_frames=bench.T, channels=3, timesteps=bench.TIMESTEPS, loss_type="l1",
                            use_dynamic_thres=True, sampling_timesteps=bench.TIMESTEPS).to(dev)
if os.environ.get("VMM_X3_WGRAD") == "0":
    model.use_x3_wgrad = False  # exact-fp32 weight gradients everywhere (default in bf16x3 mode: the 3 x 3 layers on the nine-tap split-bf16 kernel)
if os.environ.get("VMM_X3_WGRAD") == "generic":
    model.use_x3_wgrad_generic = True  # the generic split-bf16 kernel for the remaining shapes (measured slower than fp32)
print(bench.bench_training(vm, model, diff, dev, None, 1, 0, steps, precision))

if os.environ.get("VMM_TRAIN_DETAIL"):
    # per-launch table of one forward + backward of the training plan (HIP events around every launch; eager replay)
    pl = model.get_plan(bench.B_PER_GPU, bench.T, bench.HW, bench.HW, 11, dev, training=True)
    pl.launch()
    pl.pgrad.zero_(); pl.gscratch.zero_()
    fam = {}
    for name, steps, meta in (("fwd", pl.steps, pl.meta), ("bwd", pl.bwd_steps, pl.bwd_meta)):
        ms = pl.launch_timed(steps)
        for t, (fn, _, _) in zip(ms, steps):
            f = fam.setdefault(fn.__name__, [0.0, 0])
            f[0] += t
            f[1] += 1
        rows = sorted(zip(ms, steps, meta), key=lambda r: -r[0])
        print(f"== {name}: {sum(ms):.2f} ms over {len(ms)} launches", file=sys.stderr)
        for t, (fn, _, what), (_, fl, nb) in rows[:int(os.environ.get('VMM_TRAIN_DETAIL_ROWS', '60'))]:
            print(f"{t:8.3f} ms {fl / t / 1e9 if t > 0 else 0:8.1f} TFLOP/s {nb / t / 1e6 if t > 0 else 0:8.1f} GB/s  {fn.__name__:32s} {what}", file=sys.stderr)

    print("== families (ms, launches): " + ", ".join(f"{k} {v[0]:.2f}/{v[1]}" for k, v in sorted(fam.items(), key=lambda kv: -kv[1][0])), file=sys.stderr)
