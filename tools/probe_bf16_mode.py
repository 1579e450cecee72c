import sys, torch
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import videometamaterials_amd as vm
from test_gpu_hires import KW_HIRES
import bench
gpu = torch.device('cuda:0')
for name, kw, (B, T, H) in (("hires", KW_HIRES, (1, 22, 192)), ("lagr", bench.LAGRANGIAN, (2, 11, 96))):
    torch.manual_seed(0)
    m = vm.Unet3D(**kw).to(gpu).eval()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(B, 3, T, H, H, generator=g).to(gpu)
    t = torch.randint(0, 256, (B,), generator=g).to(gpu)
    cl = 51 if name == "hires" else 11
    cond = (torch.rand(B, cl, generator=g) * 2 - 1).to(gpu)
    outs = {}
    for prec in ("fp32", "bf16x3", "bf16"):
        m.precision = prec
        with torch.no_grad():
            outs[prec] = m(x, t, cond=cond, null_cond_prob=0.0).double().cpu()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(3): m(x, t, cond=cond, null_cond_prob=0.0)
            e1.record(); torch.cuda.synchronize()
        ref = outs["fp32"]
        print(name, prec, "rel vs fp32 %.3e" % float((outs[prec] - ref).norm() / ref.norm()), "max %.3e" % float((outs[prec] - ref).abs().max() / ref.abs().max()), "%.2f ms" % (e0.elapsed_time(e1) / 3), flush=True)
        plan = m.get_plan(B, T, H, H, cl, gpu)
        used = sorted({fn.__name__ for fn, _, _ in plan.steps if 'bf16' in fn.__name__})
        if prec == "bf16": print("  kernels:", [u for u in used if not u.endswith('x3')])
