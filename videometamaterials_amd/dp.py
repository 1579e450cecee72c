"""Data-parallel training and sharded sampling over RCCL (replaces the reference's Accelerate + Gloo wrapper,
main.py:31-34, vddp.py:1449, 1594-1672, 1506-1532, 1745-1749).

One process per GPU; `torch.distributed` backend "nccl" IS RCCL on ROCm (gloo in the CPU tests).
* Training: every rank runs the same training plan on its own minibatch (per-GPU batch = model.yaml `batch_size`), the
  hand-written backward fills ONE flat fp32 gradient buffer laid out in first-use order, and `BucketedAllReduce` sums
  contiguous tail slices of that buffer on a side stream as soon as the backward marks them final -- the all-reduce of
  the deep layers overlaps the backward of the shallow ones.  The 31 parameters that never receive gradients in the
  Lagrangian config are simply absent from the buffer (static set => no DDP unused-parameter bitmap exchange).
  Then one multi-tensor Adam launch (gradient pre-scaled by 1/world) and, every 10 steps, one EMA launch.
* Sampling: rows of the conditioning matrix are split exactly like Trainer.cond_to_gpu; results are padded to the longest
  shard, all-gathered once and un-padded like Trainer.remove_padding.
"""
from __future__ import annotations

import copy
from typing import List, Optional, Tuple

import torch
import torch.distributed as dist

import ctypes as C
import os

from . import _native as N
from . import hostmath
from .plan import _stream


def _host_rccl_path() -> Optional[str]:
    """The RCCL PyTorch-ROCm ships and maps for its own "nccl" backend: binding the engine to the same file keeps ONE copy of the
    library in the process (None: the engine's default, librccl.so.1 from the loader path)."""
    p = os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so")
    return p if os.path.exists(p) else None


class RcclEngine:
    """The native data-parallel engine (include/vmm_dp.h, csrc/dp_engine.hip): RCCL communicator + side stream + events behind the C ABI.
    torch.distributed is used for ONE thing only -- handing rank 0's 128-byte rendezvous id to the other ranks (any backend; a host
    without PyTorch would use a file or its own store)."""

    def __init__(self, rank: int, world: int, device: torch.device, unique_id: bytes, rccl_path: Optional[str] = None):
        assert len(unique_id) == 128
        self.lib = N.lib()
        self.rank, self.world, self.device = rank, world, torch.device(device)
        path = rccl_path if rccl_path is not None else _host_rccl_path()
        self._h = C.c_void_p()
        self._id = C.create_string_buffer(unique_id, 128)
        idx = self.device.index if self.device.index is not None else torch.cuda.current_device()
        code = self.lib.vmm_dp_init(C.byref(self._h), path.encode() if path else None, rank, world, self._id, idx)
        if code != 0 or not self._h.value:
            raise N.NativeError(f"vmm_dp_init failed with code {code} (rank {rank} of {world}, RCCL at {path or 'librccl.so.1'})")
        self._keep = None  # ctypes arrays of the registered bucket list

    @staticmethod
    def new_unique_id(rccl_path: Optional[str] = None) -> bytes:
        buf = C.create_string_buffer(128)
        path = rccl_path if rccl_path is not None else _host_rccl_path()
        N.check(N.lib().vmm_dp_get_unique_id(path.encode() if path else None, buf), "vmm_dp_get_unique_id")
        return buf.raw

    @classmethod
    def create(cls, device: torch.device, group=None) -> "RcclEngine":
        """One engine per process.  With torch.distributed initialised: its rank / world, the id broadcast from rank 0 as a Python
        object; otherwise a single-rank communicator (a one-GPU box still goes through RCCL: first-contact rehearsal)."""
        if dist.is_available() and dist.is_initialized():
            rank, world = dist.get_rank(group), dist.get_world_size(group)
            box = [cls.new_unique_id() if rank == 0 else None]
            dist.broadcast_object_list(box, src=0, group=group)
            return cls(rank, world, device, box[0])
        return cls(0, 1, device, cls.new_unique_id())

    def _check(self, code: int, what: str) -> None:
        if code != 0:
            msg = self.lib.vmm_dp_last_error(self._h)
            raise N.NativeError(f"{what} failed with code {code}: {msg.decode() if msg else ''}")

    @property
    def rccl_version(self) -> int:
        return int(self.lib.vmm_dp_rccl_version(self._h))

    def register(self, flat: torch.Tensor, slices: List[Tuple[int, int]]) -> None:
        n = len(slices)
        ptrs = (C.c_void_p * max(n, 1))(*[flat.data_ptr() + 4 * lo for lo, _ in slices])
        counts = (C.c_int64 * max(n, 1))(*[hi - lo for lo, hi in slices])
        self._keep = (ptrs, counts, flat)
        self._check(self.lib.vmm_dp_register_buckets(self._h, ptrs, counts, n), "vmm_dp_register_buckets")

    def allreduce_bucket_async(self, i: int) -> None:
        self._check(self.lib.vmm_dp_allreduce_bucket_async(self._h, i, _stream()), "vmm_dp_allreduce_bucket_async")

    def wait_all(self) -> None:
        self._check(self.lib.vmm_dp_wait_all(self._h, _stream()), "vmm_dp_wait_all")

    def set_timing(self, on: bool) -> None:
        self._check(self.lib.vmm_dp_set_timing(self._h, 1 if on else 0), "vmm_dp_set_timing")

    def window_mark(self, which: int) -> None:
        self._check(self.lib.vmm_dp_window_mark(self._h, which, _stream()), "vmm_dp_window_mark")

    def timing(self) -> Tuple[float, float, float]:
        out = (C.c_float * 3)()
        self._check(self.lib.vmm_dp_timing(self._h, out), "vmm_dp_timing")
        return float(out[0]), float(out[1]), float(out[2])

    def bucket_timing(self, n: int) -> List[Tuple[float, float]]:
        """(start, done) of every registered bucket's reduction in ms after the backward window opened (after a synchronize)."""
        out = (C.c_float * max(2 * n, 2))()
        self._check(self.lib.vmm_dp_bucket_timing(self._h, out, n), "vmm_dp_bucket_timing")
        return [(round(float(out[2 * i]), 3), round(float(out[2 * i + 1]), 3)) for i in range(n)]

    _DTYPES = {torch.float32: 0, torch.float64: 1, torch.int32: 2, torch.int64: 3}

    def all_reduce(self, t: torch.Tensor, op: str = "sum") -> None:
        assert t.is_cuda and t.is_contiguous()
        self._check(self.lib.vmm_dp_allreduce(self._h, t.data_ptr(), t.numel(), self._DTYPES[t.dtype], {"sum": 0, "max": 1, "min": 2}[op], _stream()),
                    "vmm_dp_allreduce")

    def broadcast(self, t: torch.Tensor, root: int = 0) -> None:
        assert t.is_cuda and t.is_contiguous()
        if t.numel():
            self._check(self.lib.vmm_dp_broadcast(self._h, t.data_ptr(), t.numel() * t.element_size(), root, _stream()), "vmm_dp_broadcast")

    def all_gather(self, t: torch.Tensor) -> List[torch.Tensor]:
        assert t.is_cuda and t.is_contiguous()
        out = torch.empty((self.world,) + tuple(t.shape), dtype=t.dtype, device=t.device)
        self._check(self.lib.vmm_dp_all_gather(self._h, t.data_ptr(), out.data_ptr(), t.numel() * t.element_size(), _stream()), "vmm_dp_all_gather")
        return list(out.unbind(0))

    def close(self) -> None:
        if self._h is not None and self._h.value:
            torch.cuda.synchronize(self.device)
            code = self.lib.vmm_dp_finalize(self._h)
            self._h = None
            N.check(code, "vmm_dp_finalize")


def plan_buckets(n_valid: int, marks: List[int], bucket_floats: int) -> List[Tuple[int, int]]:
    """The (lo, hi) slices BucketedAllReduce will reduce for a plan's mark sequence, in issue order (the sequence is static per plan, so the
    native engine registers them once)."""
    out, hi = [], n_valid
    for x in marks:
        x = max(0, min(x, hi))
        if hi - x >= bucket_floats or x == 0:
            if hi > x:
                out.append((x, hi))
            hi = x
    if hi > 0:
        out.append((0, hi))
    return out


def _device_collectives_ok(t: torch.Tensor, group=None) -> bool:
    """RCCL takes device buffers.  The gloo rehearsal backend (several ranks sharing one GPU on a single-GPU box; CPU tests) may lack
    device-tensor support in this build: then collectives on device tensors are staged through host memory."""
    if not t.is_cuda or dist.get_backend(group) != "gloo":
        return True
    try:
        probe = torch.zeros(4, device=t.device)
        dist.all_reduce(probe, group=group)
        torch.cuda.synchronize()
        return True
    except RuntimeError:
        return False


def broadcast_tensor(t: torch.Tensor, src: int = 0, group=None) -> None:
    """In-place broadcast that also works for device tensors under the gloo rehearsal backend."""
    if _device_collectives_ok(t, group):
        dist.broadcast(t, src=src, group=group)
    else:
        host = t.detach().cpu()
        dist.broadcast(host, src=src, group=group)
        t.copy_(host)


def all_gather_tensor(t: torch.Tensor, world: int, group=None) -> List[torch.Tensor]:
    if _device_collectives_ok(t, group):
        out = [torch.empty_like(t) for _ in range(world)]
        dist.all_gather(out, t, group=group)
        return out
    host = t.detach().cpu()
    out = [torch.empty_like(host) for _ in range(world)]
    dist.all_gather(out, host, group=group)
    return [o.to(t.device) for o in out]


class BucketedAllReduce:
    """Sum-all-reduce of a flat gradient buffer in contiguous buckets, driven by "tail is final" marks.

    mark(X) promises that flat[X:] will not change any more.  Finished tail slices are merged until they reach
    `bucket_floats`, then reduced asynchronously (on `comm_stream` for CUDA tensors).  finish() flushes the rest and
    makes the caller's stream wait for every bucket."""

    def __init__(self, flat: torch.Tensor, n_valid: int, bucket_floats: int = 4_000_000, group=None, engine: Optional[RcclEngine] = None,
                 marks: Optional[List[int]] = None):
        self.flat, self.n, self.bucket = flat, n_valid, bucket_floats
        self.group = group
        self.engine = engine
        self.world = engine.world if engine is not None else (dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1)
        self.cuda = flat.is_cuda
        # a single rank normally skips the exchange; the native engine, or VMM_DP_FORCE=1 with an initialised one-rank process group, runs it
        # anyway (a one-GPU box then meets RCCL through exactly the code path of N ranks)
        self.active = self.world > 1 or engine is not None or (os.environ.get("VMM_DP_FORCE") == "1" and dist.is_available() and dist.is_initialized())
        self.comm_stream = torch.cuda.Stream() if (self.cuda and self.active and engine is None) else None
        self._index = {}
        if engine is not None:
            if marks is None:
                raise ValueError("the native engine registers a static bucket list: pass the plan's marks")
            slices = plan_buckets(n_valid, marks, bucket_floats)
            engine.register(flat, slices)
            self._index = {sl: i for i, sl in enumerate(slices)}
        self.hi = n_valid
        self.works = []
        self.launched: List[Tuple[int, int]] = []
        # RCCL reduces device buffers in place.  The gloo rehearsal backend (several ranks sharing one GPU on a single-GPU box; CPU
        # tests) may lack device-tensor support in this build: then slices are staged through pinned host memory on the side stream.
        self.host_staged = bool(self.cuda and self.active and engine is None and not _device_collectives_ok(flat, group))
        # timing of the last step (events with timing on the compute / side stream): bench.py reports allreduce_ms / overlap_frac from them
        self.timing = False
        self._ev_buckets: List[Tuple[torch.cuda.Event, torch.cuda.Event]] = []
        self._ev_window: List[torch.cuda.Event] = []

    def start(self) -> None:
        self.hi = self.n
        self.works, self.launched = [], []
        self._ev_buckets, self._ev_window = [], []
        if self.engine is not None:
            self.engine.set_timing(self.timing)
            if self.timing:
                self.engine.window_mark(0)
            return
        if self.timing and self.comm_stream is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()  # the backward window opens here (compute stream)
            self._ev_window.append(e)

    def backward_done(self) -> None:
        """Call when the last backward launch has been enqueued: closes the window overlap_frac is measured against."""
        if self.engine is not None:
            if self.timing:
                self.engine.window_mark(1)
            return
        if self.timing and self.comm_stream is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self._ev_window.append(e)

    def last_step_timing(self) -> Optional[dict]:
        """(after a synchronize) side-stream busy time of the last step's buckets and the share of it inside the backward window."""
        if self.engine is not None:
            if not self.timing or not self.launched:
                return None
            busy, inside, win = self.engine.timing()
            spans = self.engine.bucket_timing(len(self._index))
            by_slice = {v: k for k, v in self._index.items()}
            return {"allreduce_ms": round(busy, 3), "overlap_frac": round(inside / busy, 4) if busy > 0 else None, "backward_ms": round(win, 3),
                    "bucket_spans_ms": [spans[self._index[sl]] for sl in self.launched if sl in self._index],
                    "buckets_MB": [round((hi - lo) * 4 / 1e6, 1) for lo, hi in self.launched], "registered_buckets": len(by_slice)}
        if len(self._ev_window) < 2 or not self._ev_buckets:
            return None
        w0, w1 = self._ev_window
        win = w0.elapsed_time(w1)
        busy = inside = 0.0
        spans = []
        for a, b in self._ev_buckets:
            s0, s1 = w0.elapsed_time(a), w0.elapsed_time(b)
            busy += s1 - s0
            inside += max(0.0, min(s1, win) - max(s0, 0.0))
            spans.append((round(s0, 3), round(s1, 3)))
        return {"allreduce_ms": round(busy, 3), "overlap_frac": round(inside / busy, 4) if busy > 0 else None, "backward_ms": round(win, 3),
                "bucket_spans_ms": spans, "buckets_MB": [round((hi - lo) * 4 / 1e6, 1) for lo, hi in self.launched]}

    def _reduce(self, lo: int, hi: int) -> None:
        if hi <= lo or not self.active:
            return
        sl = self.flat[lo:hi]
        self.launched.append((lo, hi))
        if self.engine is not None:
            if (lo, hi) not in self._index:
                raise RuntimeError(f"gradient slice [{lo}, {hi}) is not in the bucket list registered for this plan")
            self.engine.allreduce_bucket_async(self._index[(lo, hi)])
        elif self.cuda:
            ev = torch.cuda.Event()
            ev.record()  # everything enqueued so far on the compute stream produced flat[lo:hi]
            self.comm_stream.wait_event(ev)
            with torch.cuda.stream(self.comm_stream):
                if self.timing:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                if self.host_staged:
                    host = torch.empty(hi - lo, dtype=sl.dtype, pin_memory=True)
                    host.copy_(sl, non_blocking=True)
                    self.comm_stream.synchronize()
                    dist.all_reduce(host, op=dist.ReduceOp.SUM, group=self.group)
                    sl.copy_(host, non_blocking=True)
                    self._pinned = getattr(self, "_pinned", []) + [host]  # alive until finish()
                else:
                    w = dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
                    if self.timing:
                        w.wait()  # orders the side stream behind the collective (no host block): e1 then stamps its completion
                    else:
                        self.works.append(w)
                if self.timing:
                    e1.record()
                    self._ev_buckets.append((e0, e1))
        else:
            self.works.append(dist.all_reduce(sl, op=dist.ReduceOp.SUM, group=self.group, async_op=True))

    def mark(self, x: int) -> None:
        x = max(0, min(x, self.hi))
        if self.hi - x >= self.bucket or x == 0:
            self._reduce(x, self.hi)
            self.hi = x

    def finish(self) -> None:
        self._reduce(0, self.hi)
        self.hi = 0
        if self.engine is not None:
            self.engine.wait_all()
        elif self.comm_stream is not None:
            with torch.cuda.stream(self.comm_stream):
                for w in self.works:
                    w.wait()  # (device collectives: orders the side stream behind the collective, does not block the host)
            torch.cuda.current_stream().wait_stream(self.comm_stream)
            if self.host_staged:
                self.comm_stream.synchronize()
                self._pinned = []
        else:
            for w in self.works:
                w.wait()


class DataParallelTrainer:
    """Minimal counterpart of the reference Trainer's inner loop (vddp.py:1612-1640) for the HIP path."""

    def __init__(self, diffusion, *, train_lr: float = 1e-4, ema_decay: float = 0.995, step_start_ema: int = 2000, update_ema_every: int = 10,
                 null_cond_prob: float = 0.1, betas=(0.9, 0.999), eps: float = 1e-8, bucket_floats: int = 4_000_000, group=None,
                 engine: Optional[str] = None):
        """engine: "torch" = torch.distributed collectives (backend "nccl" is RCCL on ROCm; the default), "native" = the C-ABI engine of
        include/vmm_dp.h (its own RCCL communicator and side stream; also at world 1).  Default: $VMM_DP_ENGINE or "torch"."""
        self.model = diffusion
        self.unet = diffusion.denoise_fn
        self.ema_model = copy.deepcopy(diffusion)  # vddp.py:1453
        self.lr, self.betas, self.eps = train_lr, betas, eps
        self.ema_decay, self.step_start_ema, self.update_ema_every = ema_decay, step_start_ema, update_ema_every
        self.null_cond_prob = null_cond_prob
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        kind = engine if engine is not None else os.environ.get("VMM_DP_ENGINE", "torch")
        if kind not in ("torch", "native"):
            raise ValueError(f"engine must be 'torch' or 'native', got {kind!r}")
        self.engine: Optional[RcclEngine] = None
        if kind == "native":
            dev = next(diffusion.parameters()).device
            if dev.type != "cuda":
                raise N.NativeError("the native data-parallel engine runs over RCCL: the model must live on a GPU")
            self.engine = RcclEngine.create(dev, group)
        self.step = 0
        # loss scaling of the fp16-operand leg (Unet3D.train_precision = "fp16"): torch.cuda.amp.GradScaler's defaults, which Accelerate(mixed_precision='fp16')
        # -- the reference's configuration, main.py:34 -- uses as they are; the state lives on the device (vmm_scaler_*, include/vmm_kernels.h)
        self.loss_scale_init, self.loss_scale_growth, self.loss_scale_backoff, self.loss_scale_interval = 65536.0, 2.0, 0.5, 2000
        self._scaler = None
        self.bucket_floats = bucket_floats
        self._plan = None
        self._reducer = None
        self._adam_table = self._ema_table = None
        if self.world > 1:
            self.broadcast_parameters()

    def broadcast_parameters(self) -> None:
        """DDP's constructor broadcast (rank 0 -> all), SURVEY C1: ONE flat broadcast per dtype (the ~400 tensors of the model are
        158 MB in all; tensor by tensor that is ~400 latency-bound collectives), copied back into the live tensors in place."""
        by_dtype = {}
        for t in list(self.model.parameters()) + list(self.model.buffers()):
            by_dtype.setdefault(t.dtype, []).append(t.data)
        for dt in sorted(by_dtype, key=str):  # same order on every rank
            ts = by_dtype[dt]
            flat = torch.cat([t.reshape(-1) for t in ts])
            if self.engine is not None and flat.is_cuda:
                self.engine.broadcast(flat, 0)
            else:
                broadcast_tensor(flat, 0, self.group)
            o = 0
            for t in ts:
                t.copy_(flat[o:o + t.numel()].view_as(t))
                o += t.numel()
        self.unet.bump_generation()
        self.ema_model.load_state_dict(self.model.state_dict())

    def rccl_selfcheck(self) -> dict:
        """First contact with the collective library: the all-reduce of the rank ids must be N (N - 1) / 2 on every rank."""
        dev = next(self.model.parameters()).device
        t = torch.full((1,), float(self.rank), device=dev)
        if self.engine is not None:
            self.engine.all_reduce(t)
            got, want = float(t.item()), self.world * (self.world - 1) / 2
            return {"sum_of_rank_ids": got, "expected": want, "ok": got == want, "backend": f"vmm_dp (RCCL {self.engine.rccl_version})"}
        through = self.world > 1 or (os.environ.get("VMM_DP_FORCE") == "1" and dist.is_initialized())
        if through:
            if _device_collectives_ok(t, self.group):
                dist.all_reduce(t, group=self.group)
            else:
                h = t.cpu()
                dist.all_reduce(h, group=self.group)
                t = h.to(dev)
        got, want = float(t.item()), self.world * (self.world - 1) / 2
        return {"sum_of_rank_ids": got, "expected": want, "ok": got == want, "backend": dist.get_backend(self.group) if through else None}

    # ------------------------------------------------------------------ checkpoints in the reference's layout (vddp.py:1548-1585)
    def _optimizer_param_names(self) -> list:
        """Names of GaussianDiffusion.parameters() in the REFERENCE's order (= torch.optim.Adam's state index): this tree registers its
        parameters in the reference's order (unet3d.py), and the reference additionally carries the shared rotary table
        `init_temporal_attn.fn.fn.fn.rotary_emb.freqs` as a frozen nn.Parameter (a buffer here) in front of that attention's projections --
        it is inside Adam(model.parameters()) (vddp.py:1455) without ever receiving state.  None marks its slot."""
        names = [n for n, _ in self.unet.named_parameters()]
        first = next((i for i, n in enumerate(names) if n.startswith("init_temporal_attn.fn.fn.fn.")), None)
        if first is not None:
            names.insert(first, None)
        return names

    def state_dict(self) -> dict:
        """{model, optimizer, steps, ema} as Trainer.save writes it: `optimizer` in torch.optim.Adam's format over
        GaussianDiffusion.parameters() (index = position in the reference's list; parameters that never received a gradient have no state)."""
        names = self._optimizer_param_names()
        params = dict(self.unet.named_parameters())
        moments = getattr(self, "_moments", {})
        state = {}
        for i, n in enumerate(names):
            if n is not None and n in moments:
                m, v = moments[n]
                shape = params[n].shape
                state[i] = {"step": torch.tensor(float(self.step)), "exp_avg": m.detach().clone().view(shape).cpu(),
                            "exp_avg_sq": v.detach().clone().view(shape).cpu()}
        group = {"lr": self.lr, "betas": tuple(self.betas), "eps": self.eps, "weight_decay": 0, "amsgrad": False, "maximize": False, "foreach": None,
                 "capturable": False, "differentiable": False, "fused": None, "params": list(range(len(names)))}
        return {"model": self.model.state_dict(), "optimizer": {"state": state, "param_groups": [group]}, "steps": self.step,
                "ema": self.ema_model.state_dict()}

    def load_state_dict(self, obj: dict, strict: bool = True) -> None:
        self.model.load_state_dict(obj["model"], strict=strict)
        if "ema" in obj:
            self.ema_model.load_state_dict(obj["ema"], strict=strict)
        self.step = int(obj.get("steps", 0))
        opt = obj.get("optimizer")
        if opt:
            names = self._optimizer_param_names()
            params = dict(self.unet.named_parameters())
            n_ckpt = sum(len(g["params"]) for g in opt["param_groups"])
            if n_ckpt == len(names) - 1 and None in names:
                # a reference checkpoint written under a rotary_embedding_torch release that registers `freqs` as a buffer (the pinned 0.2.3 makes it
                # a frozen nn.Parameter): the same order without the frozen slot.  The count alone cannot tell that layout from a checkpoint written in
                # ANOTHER parameter order (this package before round 4), and the per-index shape check below cannot tell equal-shaped parameters apart
                # (to_q / to_k / to_v, same-shape convolutions, gammas): the checkpoint's own `model` keys -- a state_dict lists parameters in
                # parameters() order -- must name the parameters in this tree's order.
                names = [n for n in names if n is not None]
                self._check_model_key_order(obj["model"], names)
            elif n_ckpt != len(names):
                raise ValueError(f"optimizer state covers {n_ckpt} parameters, this model has {len(names)} (reference order, frozen rotary table included) "
                                 f"or {len(names) - 1} (without it)")
            dev = next(self.unet.parameters()).device
            if not hasattr(self, "_moments"):
                self._moments = {}
            for i, st in opt["state"].items():
                n = names[int(i)]
                if n is None:
                    continue  # (the frozen rotary table: torch never creates state for it)
                for key in ("exp_avg", "exp_avg_sq"):  # a moment bound to a parameter of another size would be read out of bounds by vmm_adam_step
                    # (shapes, not element counts: to_q / to_k / to_v and many convolutions have equal sizes, and a checkpoint written in another
                    # parameter order -- e.g. by this package before round 4 -- must not bind silently; flat moments of the right length are accepted)
                    if tuple(st[key].shape) != tuple(params[n].shape) and not (st[key].dim() == 1 and st[key].numel() == params[n].numel()):
                        raise ValueError(f"optimizer state {i} ({key}: shape {tuple(st[key].shape)}) does not fit parameter {n} {tuple(params[n].shape)}")
                if n in self._moments:  # keep the storage the device job tables point at
                    self._moments[n][0].copy_(st["exp_avg"].reshape(-1))
                    self._moments[n][1].copy_(st["exp_avg_sq"].reshape(-1))
                else:
                    self._moments[n] = (st["exp_avg"].detach().to(dev, torch.float32).reshape(-1).clone(),
                                        st["exp_avg_sq"].detach().to(dev, torch.float32).reshape(-1).clone())
                if "steps" not in obj and "step" in st:
                    self.step = max(self.step, int(float(st["step"])))
            g = opt["param_groups"][0]
            self.lr, self.betas, self.eps = g["lr"], tuple(g["betas"]), g["eps"]
        self._ptr_sig = None  # parameters may have been re-homed: rebuild the job tables at the next step

    def _check_model_key_order(self, model_sd: dict, names: list) -> None:
        """The parameter entries of a checkpoint's `model` state_dict, in the order it lists them, against `names` (this tree's parameters() order):
        an optimizer state indexed by position is only meaningful when the two agree."""
        own = set(names)
        seen = []
        for k in model_sd:
            k = k[len("module."):] if k.startswith("module.") else k
            if not k.startswith("denoise_fn."):
                continue
            k = k[len("denoise_fn."):]
            k = k[len("module."):] if k.startswith("module.") else k
            k = self.unet._own_key(k)
            if k in own:
                seen.append(k)
        if seen != names:
            first = next((i for i, (a, b) in enumerate(zip(seen, names)) if a != b), min(len(seen), len(names)))
            raise ValueError("the checkpoint's optimizer state is one slot short of this model's parameter list and its `model` entries do not follow this "
                             f"model's parameter order (first difference at position {first}: {seen[first] if first < len(seen) else None!r} against "
                             f"{names[first] if first < len(names) else None!r}): refusing to bind Adam moments by position")

    def save(self, path: str) -> None:
        torch.save(self.state_dict(), path)

    def load(self, path: str, strict: bool = True) -> dict:
        obj = torch.load(path, map_location="cpu")
        self.load_state_dict(obj, strict=strict)
        return obj

    # ------------------------------------------------------------------ setup for one input shape
    def _pointer_signature(self) -> tuple:
        return tuple(p.data_ptr() for p in self.unet.parameters()) + tuple(p.data_ptr() for p in self.ema_model.denoise_fn.parameters())

    def _moments_for(self, pl, dev):
        """Adam's first / second moments live per PARAMETER NAME and survive a change of training plan (another batch shape, e.g.
        the short last batch of an epoch -- the reference's DataLoader has no drop_last): only the job tables are rebuilt."""
        if not hasattr(self, "_moments"):
            self._moments = {}
        missing = [(name, n) for name, (_, n) in pl.param_slices.items() if name not in self._moments]
        if missing:
            flat = torch.zeros(2 * sum((n + 3) // 4 * 4 for _, n in missing), dtype=torch.float32, device=dev)
            o = 0
            for name, n in missing:
                npad = (n + 3) // 4 * 4
                self._moments[name] = (flat[o:o + n], flat[o + npad:o + npad + n])
                o += 2 * npad
        return self._moments

    def _build_tables(self, pl, dev) -> None:
        params = dict(self.unet.named_parameters())
        ema_params = dict(self.ema_model.denoise_fn.named_parameters())
        moments = self._moments_for(pl, dev)
        jobs = (N.OptimJob * len(pl.param_slices))()
        ejobs = (N.OptimJob * len(params))()
        self._max_n = 0
        for i, (name, (off, n)) in enumerate(pl.param_slices.items()):
            j = jobs[i]
            m, v = moments[name]
            j.p, j.g = params[name].data_ptr(), pl.pgrad.data_ptr() + 4 * off
            j.m, j.v, j.n = m.data_ptr(), v.data_ptr(), n
            self._max_n = max(self._max_n, n)
        for i, (name, p) in enumerate(params.items()):  # EMA covers every parameter (vddp.py:121-124)
            j = ejobs[i]
            j.p, j.m, j.n = p.data_ptr(), ema_params[name].data_ptr(), p.numel()
            self._max_n = max(self._max_n, p.numel())
        self._adam_table = (torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev), len(jobs))
        self._ema_table = (torch.frombuffer(bytearray(bytes(ejobs)), dtype=torch.uint8).to(dev), len(ejobs))
        self._ptr_sig = self._pointer_signature()

    def _prepare(self, x, cond, focus: bool = False):
        B, _, T, H, W = x.shape
        pl = self.unet.get_plan(B, T, H, W, cond.shape[-1], x.device, training=True, focus=focus)
        dev = x.device
        if pl is not self._plan:
            self._plan = pl
            self._build_tables(pl, dev)
            if self._reducer is not None and self.engine is not None:
                torch.cuda.current_stream().synchronize()  # the engine's bucket list is replaced: nothing of the old plan may be in flight
            self._reducer = BucketedAllReduce(pl.pgrad, pl.pgrad_floats, self.bucket_floats, self.group, engine=self.engine,
                                              marks=[x for _, x in sorted(dict(pl.bwd_marks).items())])
            self._acc = torch.empty(1, dtype=torch.float64, device=dev)
            self._loss = torch.empty((), dtype=torch.float32, device=dev)
        elif self._ptr_sig != self._pointer_signature():
            # parameters were re-homed (.to(), `.data =` as in the reference EMA, load_state_dict(assign=True)): the device job
            # tables hold raw pointers the Adam / EMA kernels write through
            self._build_tables(pl, dev)
        return pl

    # ------------------------------------------------------------------ one optimisation step
    def train_step(self, x: torch.Tensor, cond: torch.Tensor, *, t: Optional[torch.Tensor] = None, noise: Optional[torch.Tensor] = None,
                   mask: Optional[torch.Tensor] = None, prob_focus_present: float = 0.0, focus_present_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
        """x in [0,1] (B,C,T,H,W), cond (B,F).  Returns the (device) loss of this rank.  t / noise / mask may be injected.
        prob_focus_present / focus_present_mask: forwarded like Trainer.train does (vddp.py:1626-1627); the mask is drawn before the
        classifier-free-guidance mask, as in the reference's forward (vddp.py:740, 749)."""
        d, lib = self.model, N.lib()
        B = x.shape[0]
        dev = x.device
        focus = self.unet._focus(B, focus_present_mask, prob_focus_present, dev)
        pl = self._prepare(x, cond, focus=focus is not None)
        self.step += 1
        t = torch.randint(0, d.num_timesteps, (B,), device=dev).long() if t is None else t  # vddp.py:1065
        noise = torch.randn_like(x) if noise is None else noise  # vddp.py:1047
        mask = self.unet._mask(B, self.null_cond_prob, dev) if mask is None else mask  # vddp.py:749
        per = x.numel() // B
        # x*2-1 and q_sample fused, straight into the plan's input slot (vddp.py:1066,1036-1042)
        N.check(lib.vmm_q_sample(x.contiguous().data_ptr(), noise.contiguous().data_ptr(), t.data_ptr(), d.sqrt_alphas_cumprod.data_ptr(),
                                 d.sqrt_one_minus_alphas_cumprod.data_ptr(), 1, pl.x_in.data_ptr(), B, per, _stream()), "vmm_q_sample")
        pl.time_in.copy_(t)
        pl.cond_in.copy_(cond)
        pl.mask_in.copy_(mask)
        if focus is not None:
            pl.focus_in.copy_(focus)
        # (_prepare -> get_plan re-packed the operand layouts from the live parameters, one launch: the last optimiser step bumped the
        # model's generation)
        pl.launch()
        sq = 1 if d.loss_type == "l2" else 0
        N.check(lib.vmm_loss_reduce(noise.data_ptr(), pl.out.data_ptr(), pl.out.numel(), sq, self._acc.data_ptr(), self._loss.data_ptr(), _stream()), "loss")
        scaled = self.unet.train_precision == "fp16"  # fp16 operands: gradients of the l1 loss (~1e-6 per element) need the loss scale to stay representable
        st = self._scaler_state(dev) if scaled else None
        N.check(lib.vmm_loss_grad(noise.data_ptr(), pl.out.data_ptr(), pl.out.numel(), sq, st.data_ptr() if scaled else None, pl.dout.data_ptr(), _stream()),
                "loss grad")
        self._reducer.start()
        pl.backward(None, on_mark=self._reducer.mark if self._reducer.active else None)
        self._reducer.backward_done()
        self._reducer.finish()
        b1, b2 = self.betas
        tab, n = self._adam_table
        if scaled:
            # GradScaler.step / update without a host round trip (vddp.py:1629-1633 under Accelerate's fp16): an inf / nan anywhere in the (all-reduced, hence
            # rank-consistent) gradient buffer turns the optimiser launch into a no-op and halves the scale; 2000 clean steps double it
            N.check(lib.vmm_grad_nonfinite(pl.pgrad.data_ptr(), pl.pgrad.numel(), st.data_ptr(), _stream()), "vmm_grad_nonfinite")
            N.check(lib.vmm_adam_step_scaled(tab.data_ptr(), n, self._max_n, self.lr, b1, b2, self.eps, 1.0 / self.world, st.data_ptr(), _stream()), "vmm_adam_step_scaled")
            N.check(lib.vmm_scaler_update(st.data_ptr(), self.loss_scale_growth, self.loss_scale_backoff, int(self.loss_scale_interval), _stream()), "vmm_scaler_update")
        else:
            N.check(lib.vmm_adam_step(tab.data_ptr(), n, self._max_n, self.lr, b1, b2, self.eps, self.step, 1.0 / self.world, _stream()), "vmm_adam_step")
        self.unet.bump_generation()  # written through raw pointers: autograd's version counters did not move
        ref_step = self.step - 1  # Trainer.step while this optimiser step runs: the reference counts from 0 (vddp.py:1612-1640)
        if ref_step % self.update_ema_every == 0:  # vddp.py:1637-1639, 1500-1504
            tab, n = self._ema_table
            N.check(lib.vmm_ema_step(tab.data_ptr(), n, self._max_n, self.ema_decay, 1 if ref_step < self.step_start_ema else 0, _stream()), "vmm_ema_step")
            self.ema_model.denoise_fn.bump_generation()
        return self._loss

    def _scaler_state(self, dev) -> torch.Tensor:
        if self._scaler is None or self._scaler.device != torch.device(dev):
            self._scaler = torch.empty(8, dtype=torch.float32, device=dev)
            N.check(N.lib().vmm_scaler_init(self._scaler.data_ptr(), float(self.loss_scale_init), _stream()), "vmm_scaler_init")
        return self._scaler

    def loss_scale_state(self) -> Optional[dict]:
        """The device-side GradScaler state of the fp16 leg (one device-to-host copy; None before its first step): scale, growth tracker, steps skipped for a
        non-finite gradient, optimiser steps taken."""
        if self._scaler is None:
            return None
        v = self._scaler[:5].cpu().tolist()
        return {"scale": v[0], "growth_tracker": int(v[1]), "skipped_steps": int(v[3]), "optimizer_steps": int(v[4])}

    # ------------------------------------------------------------------ sharded sampling (vddp.py:1506-1532, 1816-1845)
    @torch.no_grad()
    def sample_sharded(self, cond_all: torch.Tensor, guidance_scale: float = 5.0, batch: int = 2, use_ema: bool = True,
                       seed: Optional[int] = None) -> Optional[torch.Tensor]:
        """Every rank samples its contiguous block of rows; rank 0 returns the (N, C, T, H, W) result, others None.
        seed: re-seed the generator with seed + (first row of the batch) before every batch -- the noise of a row then does not depend on
        which rank samples it (with batch = 1 the result is independent of the world size; the tests use that)."""
        model = self.ema_model if use_ema else self.model
        dev = next(model.parameters()).device
        if self.world > 1:  # the reference broadcasts the conditioning matrix as a pickled object; here: one raw tensor
            cond_all = cond_all.to(dev).contiguous()
            if self.engine is not None:
                self.engine.broadcast(cond_all, 0)
            else:
                broadcast_tensor(cond_all, 0, self.group)
        n_rows = cond_all.shape[0]
        outs = []
        for a, b in hostmath.shard_rows(n_rows, self.rank, self.world, batch):
            if seed is not None:
                torch.manual_seed(seed + a)
            outs.append(model.sample(cond=cond_all[a:b].to(dev), guidance_scale=guidance_scale))
        shp = (model.channels, model.num_frames, model.image_size, model.image_size)
        mine = torch.cat(outs, dim=0) if outs else torch.zeros((0,) + shp, device=dev)
        if self.world == 1:
            return mine
        lengths = [sum(b - a for a, b in hostmath.shard_rows(n_rows, rk, self.world, batch)) for rk in range(self.world)]
        max_len = max(lengths)
        padded = torch.zeros((max_len,) + shp, device=dev)
        padded[: mine.shape[0]] = mine
        gathered = self.engine.all_gather(padded) if self.engine is not None else all_gather_tensor(padded, self.world, self.group)
        if self.rank != 0:
            return None
        return hostmath.strip_padding(torch.cat(gathered, dim=0), lengths, max_len)
