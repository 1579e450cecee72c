import sys, os, json, torch
sys.path.insert(0,'/root/repo')
import bench, videometamaterials_amd as vm
dev=torch.device("cuda",0)
torch.manual_seed(0)
model = vm.Unet3D(**bench.LAGRANGIAN).to(dev)
diff = vm.GaussianDiffusion(model, image_size=bench.HW, num_frames=bench.T, channels=3, timesteps=bench.TIMESTEPS, loss_type="l1", use_dynamic_thres=True, sampling_timesteps=bench.TIMESTEPS).to(dev)
for prec in sys.argv[1:]:
    r = bench.bench_training(vm, model, diff, dev, None, 1, 0, steps=6, precision=prec, want_roofline=True)
    fam = r.get("roofline_training", {}).get("ms_by_kernel_family", {})
    print(prec, r["ms_per_step"], "loss", r["loss"], " ".join("%s=%.2f" % (k.replace("vmm_",""), v) for k, v in list(fam.items())[:18]), flush=True)
    for k in [k for k, v in model._plans.items() if v.training]:
        del model._plans[k]
    torch.cuda.empty_cache()
