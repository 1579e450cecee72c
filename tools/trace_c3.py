"""Measurement aid: phase time line of one launch of the one-tile-per-workgroup 3x3 kernel (conv3x3_x3_kernel).

    python tools/build_ab.py conv3x3_bf16x3 -DVMM_C3_TRACE_BUILD=1    # (the stamps are not in the default build: they cost registers)
    VMM_LIB_PATH=$PWD/videometamaterials_amd/libvmm_hip_ab.so VMM_C3_TRACE=<k> VMM_C3_TRACE_FILE=gpurun_out/c3_trace.txt python tools/trace_c3.py

runs one eager denoiser forward at the bench shape; the k-th 3x3 launch (network order: 0 = downs.0.0.block1, 1 = downs.0.0.block2, ...)
writes 16 stamps per workgroup (wave 0: entry, loads issued, first patch stored, first barrier, [end of chunk c's steps, next patch
stored + barrier]..., output stores issued, end, HW_ID, XCC_ID); this script then prints where a workgroup's cycles go and how the
workgroups that shared a CU overlapped."""
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
    t0 = rows[:, 0].min()
    nch = int(((rows[:, 4:12] > 0).sum(1).max() + 1) // 2)
    names = ["entry->loads issued", "first patch arrives + stored", "first barrier"]
    cols = [(0, 1), (1, 2), (2, 3)]
    prev = 3
    for c in range(nch):
        names.append(f"chunk {c} steps")
        cols.append((prev, 4 + 2 * c))
        prev = 4 + 2 * c
        if c + 1 < nch:
            names.append(f"chunk {c + 1} patch store + 2 barriers")
            cols.append((prev, 5 + 2 * c))
            prev = 5 + 2 * c
    names += ["epilogue: bias/res + output stores issued", "GroupNorm sums + slots"]
    cols += [(prev, 12), (12, 13)]
    total = rows[:, 13] - rows[:, 0]
    print(f"{len(rows)} workgroups; launch span {rows[:, 13].max() - t0} cycles; workgroup lifetime mean {total.mean():.0f} p10 {np.percentile(total, 10):.0f} "
          f"p90 {np.percentile(total, 90):.0f}")
    for n, (a, b) in zip(names, cols):
        d = rows[:, b] - rows[:, a]
        print(f"  {n:45s} mean {d.mean():8.0f}  p10 {np.percentile(d, 10):8.0f}  p50 {np.percentile(d, 50):8.0f}  p90 {np.percentile(d, 90):8.0f}   {100 * d.mean() / total.mean():5.1f} %")
    # occupancy over time: how many workgroups are alive / inside their step loops, per CU slot
    hw, xcc = rows[:, 14], rows[:, 15] & 0xf
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 0x1) << 4) | (((hw >> 13) & 0x7) << 5) | (xcc << 8)
    ncu = len(np.unique(cu))
    span = rows[:, 13].max() - t0
    grid = np.linspace(0, span, 200)
    alive = np.array([((rows[:, 0] - t0 <= g) & (rows[:, 13] - t0 > g)).sum() for g in grid])
    inloop = np.zeros_like(grid)
    for c in range(nch):
        a = 3 if c == 0 else 5 + 2 * (c - 1)
        b = 4 + 2 * c
        inloop += np.array([((rows[:, a] - t0 <= g) & (rows[:, b] - t0 > g)).sum() for g in grid])
    print(f"distinct CUs seen {ncu}; workgroups alive per CU over the launch (20 bins): " + " ".join(f"{x / ncu:.2f}" for x in alive.reshape(20, 10).mean(1)))
    print("workgroups inside a step loop per CU:                                    " + " ".join(f"{x / ncu:.2f}" for x in inloop.reshape(20, 10).mean(1)))
    order = np.argsort(rows[:, 0])
    print("first workgroups by entry time (entry, lifetime):", [(int(rows[i, 0] - t0), int(total[i])) for i in order[:6]])
    print("last workgroups by entry time  (entry, lifetime):", [(int(rows[i, 0] - t0), int(total[i])) for i in order[-6:]])


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
    if os.environ.get("VMM_TRACE_PRECISION"):  # e.g. bf16 (the single-pass instances; VMM_TRACE_FP32_STORE=1: over fp32-stored maps, the training legs' form)
        model.precision = os.environ["VMM_TRACE_PRECISION"]
        model.bf16_storage = not os.environ.get("VMM_TRACE_FP32_STORE")
    B = 2 * bench.B_PER_GPU
    x = torch.randn(B, 3, bench.T, bench.HW, bench.HW, device=dev)
    t = torch.randint(0, 256, (B,), device=dev)
    cond = torch.rand(B, 11, device=dev) * 2 - 1
    with torch.no_grad():
        for _ in range(3):  # (38 launches per forward: VMM_C3_TRACE=77 is downs.0.0.block2 of the third, warm, forward)
            model(x, t, cond=cond, null_cond_prob=0.0)
        torch.cuda.synchronize()
    analyse(os.environ.get("VMM_C3_TRACE_FILE", "c3_trace.txt"))
