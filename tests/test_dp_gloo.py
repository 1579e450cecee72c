"""N > 1 host logic on CPU with gloo (world_size 2): the bucketed tail-slice all-reduce and the sampling shard / gather rules."""
import os
import sys

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

HERE = os.path.dirname(os.path.abspath(__file__))


def _worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videometamaterials_amd import hostmath
        from videometamaterials_amd.dp import BucketedAllReduce
        n = 10_000
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        red = BucketedAllReduce(flat, n, bucket_floats=1500)
        red.start()
        for x in (9000, 8800, 7000, 6999, 3000, 100, 0):  # backward marks: the tail flat[x:] is final
            red.mark(x)
        red.finish()
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(flat, want)
        covered = sorted(red.launched)
        contiguous = covered[0][0] == 0 and covered[-1][1] == n and all(a[1] == b[0] for a, b in zip(covered, covered[1:]))
        # sampling shards: contiguous floor(N/P) blocks, remainder on the last rank, pad -> all_gather -> strip
        N_rows = 7
        rows = hostmath.shard_rows(N_rows, rank, world, 2)
        mine = torch.cat([torch.arange(a, b, dtype=torch.float32) for a, b in rows]) if rows else torch.zeros(0)
        lengths = [sum(b - a for a, b in hostmath.shard_rows(N_rows, r, world, 2)) for r in range(world)]
        max_len = max(lengths)
        padded = torch.zeros(max_len)
        padded[: mine.numel()] = mine
        gathered = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(gathered, padded)
        full = hostmath.strip_padding(torch.cat(gathered), lengths, max_len)
        q.put((rank, ok, contiguous, len(red.launched), full.tolist()))
    finally:
        dist.destroy_process_group()


def test_bucketed_allreduce_and_sharded_gather_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29600 + os.getpid() % 300
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok, contiguous, nlaunch, full in res:
        assert ok, f"rank {rank}: reduced buffer wrong"
        assert contiguous, f"rank {rank}: buckets do not tile the buffer"
        assert 2 <= nlaunch <= 7
        assert full == [float(i) for i in range(7)]


class _GlooEngine:
    """Stand-in with the surface of dp.RcclEngine (register / allreduce_bucket_async / wait_all / timing hooks) whose collectives go over gloo: the
    native engine's N-rank CONTROL FLOW in BucketedAllReduce -- static bucket list from the plan's marks, buckets issued by index in mark order, one
    wait before the optimiser -- on CPU, where RCCL itself cannot run."""

    def __init__(self, world):
        self.world, self.log, self.pending = world, [], []

    def register(self, flat, slices):
        self.flat, self.slices = flat, list(slices)
        self.log.append(("register", len(slices)))

    def allreduce_bucket_async(self, i):
        lo, hi = self.slices[i]
        self.pending.append(dist.all_reduce(self.flat[lo:hi], async_op=True))
        self.log.append(("bucket", i))

    def wait_all(self):
        for w in self.pending:
            w.wait()
        self.pending = []
        self.log.append(("wait",))

    def set_timing(self, on):
        pass


def _engine_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from videometamaterials_amd.dp import BucketedAllReduce, plan_buckets
        n, marks = 10_000, [9000, 8800, 7000, 6999, 3000, 100, 0]
        flat = torch.arange(n, dtype=torch.float32) * (rank + 1)
        eng = _GlooEngine(world)
        red = BucketedAllReduce(flat, n, bucket_floats=1500, engine=eng, marks=marks)
        want_slices = plan_buckets(n, marks, 1500)
        for _ in range(2):  # two steps: the registered list is reused
            flat.copy_(torch.arange(n, dtype=torch.float32) * (rank + 1))
            red.start()
            for x in marks:
                red.mark(x)
            red.finish()
        want = torch.arange(n, dtype=torch.float32) * sum(r + 1 for r in range(world))
        ok = torch.equal(flat, want)
        issued = [e[1] for e in eng.log if e[0] == "bucket"]
        q.put((rank, ok, red.launched == want_slices, issued, [e[0] for e in eng.log].count("register"), [e[0] for e in eng.log].count("wait")))
        # a mark sequence the registered list does not cover is an error, not a silent skip
        bad = BucketedAllReduce(flat, n, bucket_floats=1500, engine=_GlooEngine(world), marks=[5000, 0])
        bad.start()
        try:
            bad.mark(7000)
            q.put((rank, "no error"))
        except RuntimeError:
            q.put((rank, "raised"))
    finally:
        dist.destroy_process_group()


def test_native_engine_control_flow_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 250
    procs = [ctx.Process(target=_engine_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(4)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        if len(r) == 2:
            assert r[1] == "raised", r
            continue
        rank, ok, same_slices, issued, n_reg, n_wait = r
        assert ok and same_slices, f"rank {rank}"
        nb = len(issued) // 2
        assert issued == list(range(nb)) * 2 and nb >= 3  # by index, in mark order, both steps
        assert n_reg == 1 and n_wait == 2


def _bcast_worker(rank, world, port, q):
    sys.path.insert(0, os.path.dirname(HERE))
    sys.path.insert(0, HERE)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import helpers
        import videometamaterials_amd as vm
        from videometamaterials_amd.dp import DataParallelTrainer
        kw, (B, T, H, W), _ = helpers.CONFIGS["lagr16"]
        torch.manual_seed(10 + rank)  # every rank starts from DIFFERENT weights
        model = vm.Unet3D(**kw)
        with torch.no_grad():
            for p in model.parameters():
                p.add_(torch.randn_like(p) * 0.01 * (rank + 1))
        diff = vm.GaussianDiffusion(model, image_size=H, num_frames=T, channels=3, timesteps=16, sampling_timesteps=16, loss_type="l1")
        before = float(sum(p.double().sum() for p in model.parameters()))
        tr = DataParallelTrainer(diff)  # constructor broadcast: ONE flat collective per dtype
        after = float(sum(p.double().sum() for p in tr.unet.parameters()))
        ema = float(sum(p.double().sum() for p in tr.ema_model.denoise_fn.parameters()))
        check = tr.rccl_selfcheck()
        q.put((rank, before, after, ema, check))
    finally:
        dist.destroy_process_group()


def test_constructor_broadcast_makes_replicas_identical_world2():
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = 29300 + os.getpid() % 300
    procs = [ctx.Process(target=_bcast_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in procs)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (_, b0, a0, e0, c0), (_, b1, a1, e1, c1) = res
    assert b0 != b1                      # the replicas really differed
    assert a0 == b0 and a1 == b0         # rank 0's weights everywhere, rank 0 unchanged
    assert e0 == b0 and e1 == b0         # the EMA copies follow
    assert c0["ok"] and c1["ok"] and c0["sum_of_rank_ids"] == 1.0
