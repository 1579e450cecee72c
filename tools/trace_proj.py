"""Measurement aid: phase time line of one launch of the A-stationary projection kernel (proj_x3_kernel).

    python tools/build_ab.py proj_bf16x3 -DVMM_PJ_TRACE_BUILD=1   # (the stamps are not in the default build)
    VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so VMM_PJ_TRACE=<k> VMM_PJ_TRACE_FILE=gpurun_out/pj_trace.txt python tools/trace_proj.py

runs three eager denoiser forwards at the bench shape; the k-th projection launch of the process writes 18 stamps per workgroup (wave 0:
entry, row requests issued, rows staged, barrier, then per column chunk: end of the k16 steps, end of the epilogue; exit)."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))


def analyse(path):
    with open(path) as f:
        head = f.readline().strip()
        rows = np.array([[int(x) for x in line.split()] for line in f if line.strip()], dtype=np.int64)
    print(head)
    rows = rows[rows[:, 0] > 0]
    nch = int((rows[:, 4:16] > 0).sum(1).max() // 2)
    total = rows[:, 16] - rows[:, 0]
    print(f"{len(rows)} workgroups, lifetime mean {total.mean():.0f} p10 {np.percentile(total, 10):.0f} p90 {np.percentile(total, 90):.0f} cycles")
    spans = [("entry -> row requests issued", 0, 1), ("rows arrive, LayerNorm / split, LDS store", 1, 2), ("barrier", 2, 3)]
    prev = 3
    for c in range(nch):
        spans.append((f"chunk {c}: k16 steps", prev, 4 + 2 * c))
        spans.append((f"chunk {c}: epilogue", 4 + 2 * c, 5 + 2 * c))
        prev = 5 + 2 * c
    for name, a, b in spans:
        d = rows[:, b] - rows[:, a]
        print(f"  {name:45s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p50 {np.percentile(d, 50):8.0f}  p90 {np.percentile(d, 90):8.0f}   {100 * d.mean() / total.mean():5.1f} %")


if __name__ == "__main__":
    if len(sys.argv) > 1:
        analyse(sys.argv[1])
        sys.exit(0)
    import torch
    import bench
    import videometamaterials_amd as vm
    dev = torch.device("cuda", 0)
    torch.manual_seed(0)
    model = vm.Unet3D(**bench.LAGRANGIAN).to(dev).eval()
    B = 2 * bench.B_PER_GPU
    x = torch.randn(B, 3, bench.T, bench.HW, bench.HW, device=dev)
    t = torch.randint(0, 256, (B,), device=dev)
    cond = torch.rand(B, 11, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(3):
            model(x, t, cond=cond, null_cond_prob=0.0)
        torch.cuda.synchronize()
    analyse(os.environ.get("VMM_PJ_TRACE_FILE", "pj_trace.txt"))
