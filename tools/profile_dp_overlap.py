"""Two data-parallel ranks on ONE GPU (gloo carries the bucketed all-reduce; RCCL refuses duplicate devices) for a rocprofv3 trace that
shows the gradient reduction of the deep layers running while the backward of the shallow layers is still executing:

    cd /tmp && export TMPDIR=/tmp
    rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dp -o dp_%pid% -- \
        python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29533 \
        $GRAFT_REPO_ROOT/tools/profile_dp_overlap.py
    python tools/profile_dp_overlap.py summarize gpurun_out/prof_dp > profiles/r02_dp_overlap.txt

On a multi-GPU node the same script runs with the nccl backend (VMM_DIST_BACKEND=nccl): the reductions then appear as RCCL kernels.
"""
import csv
import glob
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run():
    import torch
    import torch.distributed as dist
    import bench
    import videometamaterials_amd as vm
    from videometamaterials_amd.dp import DataParallelTrainer
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    backend = os.environ.get("VMM_DIST_BACKEND", "gloo")
    ndev = torch.cuda.device_count()
    torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", "0")) % ndev)
    dev = torch.device("cuda", torch.cuda.current_device())
    dist.init_process_group(backend, rank=rank, world_size=world)
    torch.manual_seed(0)
    model = vm.Unet3D(**bench.LAGRANGIAN).to(dev)
    diff = vm.GaussianDiffusion(model, image_size=bench.HW, num_frames=bench.T, channels=3, timesteps=256, loss_type="l1", use_dynamic_thres=True,
                                sampling_timesteps=256).to(dev)
    tr = DataParallelTrainer(diff, train_lr=1e-4)
    g = torch.Generator().manual_seed(100 + rank)
    B = int(os.environ.get("VMM_DP_BATCH", "2"))
    x = torch.rand(B, 3, bench.T, bench.HW, bench.HW, generator=g).to(dev)
    cond = (torch.rand(B, 11, generator=g) * 2 - 1).to(dev)
    for _ in range(3):
        loss = tr.train_step(x, cond)
    torch.cuda.synchronize()
    dist.barrier()
    if rank == 0:
        print(f"world {world} backend {backend} host_staged {tr._reducer.host_staged} buckets {tr._reducer.launched} loss {float(loss):.5f}", flush=True)
    dist.destroy_process_group()


def summarize(root):
    """Per process: the last optimisation step's backward window and what the copy engines / communication kernels did inside it."""
    for kt in sorted(glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)):
        mc = kt.replace("kernel_trace", "memory_copy_trace")
        ks = [r for r in csv.DictReader(open(kt))]
        if not any("wgrad" in r["Kernel_Name"] for r in ks):
            continue
        for r in ks:
            r["s"], r["e"] = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        ks.sort(key=lambda r: r["s"])
        lg = [i for i, r in enumerate(ks) if "loss_grad" in r["Kernel_Name"]]
        adam = [i for i, r in enumerate(ks) if "adam" in r["Kernel_Name"]]
        if not lg or not adam or adam[-1] < lg[-1]:
            continue
        bwd = ks[lg[-1]:adam[-1] + 1]            # backward of the last optimisation step: loss_grad .. adam
        t0, t1 = bwd[0]["s"], bwd[-1]["e"]
        # communication: RCCL kernels, or (gloo rehearsal) the blit kernels that move a bucket between HBM and gloo's pinned host buffers
        comm = [r for r in bwd if any(s in r["Kernel_Name"].lower() for s in ("nccl", "rccl", "allreduce", "all_reduce"))
                or ("copyBuffer" in r["Kernel_Name"] and r["e"] - r["s"] > 20_000)]
        copies = []
        if os.path.exists(mc):
            for r in csv.DictReader(open(mc)):
                s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
                if e > t0 and s < t1 and (e - s) > 20_000:   # the bucket-sized transfers (gloo stages the reduction through the host)
                    copies.append((s, e, r.get("Direction", "")))
        print(f"== {os.path.relpath(kt, root)}")
        print(f"backward of the last step: {len(bwd)} kernels, {(t1 - t0) / 1e6:.2f} ms (loss_grad .. adam)")
        compute = [r for r in bwd if r not in comm]

        def busy_inside(s, e):
            tot, names = 0, []
            for r in compute:
                a, b = max(s, r["s"]), min(e, r["e"])
                if b > a:
                    tot += b - a
                    names.append(r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0].split("<")[0][:40])
            return tot, names
        events = [(r["s"], r["e"], "comm kernel " + r["Kernel_Name"][:40]) for r in comm] + [(s, e, "copy " + d) for s, e, d in copies]
        events.sort()
        hidden = total = 0
        for s, e, what in events:
            b, names = busy_inside(s, e)
            total += e - s
            hidden += min(b, e - s)
            uniq = []
            for n in names:
                if n not in uniq:
                    uniq.append(n)
            print(f"  +{(s - t0) / 1e6:8.3f} ms  {(e - s) / 1e6:7.3f} ms  {what:34s} concurrent backward kernels {min(b, e - s) / max(e - s, 1) * 100:5.1f}% of it: {', '.join(uniq[:4])}")
        if total:
            print(f"reduction activity inside the backward window: {total / 1e6:.2f} ms, of which {hidden / 1e6:.2f} ms ({hidden / total * 100:.0f}%) ran beside backward kernels")
        tail = max([e for _, e, _ in events], default=t0)
        print(f"last reduction event ends {(tail - t0) / 1e6:.2f} ms into the {(t1 - t0) / 1e6:.2f} ms window (adam starts at {(bwd[-1]['s'] - t0) / 1e6:.2f} ms)")


if __name__ == "__main__":
    if len(sys.argv) > 2 and sys.argv[1] == "summarize":
        summarize(sys.argv[2])
    else:
        run()
