"""Static execution plans for the Unet3D hot path.

A plan = (arena in HBM, packed weights, ordered list of C-ABI kernel launches) for one
input shape.  It is built once (``build_plan``), its device addresses never change, and
``Plan.run`` just replays the launches on the caller's current HIP stream -- which also
makes the whole denoiser capturable into a hipGraph (diffusion.py does that for the
256-step sampling loop).

Inference plans reuse dead buffers (first-fit arena).  Training plans keep every
intermediate, and carry a second launch list -- the hand-derived backward pass -- that
writes parameter gradients (torch layout) into ONE flat buffer ordered by first use in
the forward pass.  The backward list is annotated with marks "every gradient at offset
>= X is final now", so the data-parallel engine (dp.py) can all-reduce contiguous tail
slices of the buffer over RCCL while the rest of the backward is still running.

Working layout: every feature map is "rows x channels" fp32 with rows = (b, t, h, w)
(frame-major channels-last).  See DESIGN.md for the kernel inventory and data layout.
"""
from __future__ import annotations

import ctypes as C
import math
import os
import sys
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _native as N
from . import hostmath

ALIGN = 64  # floats (256 B)
LA_PART = 32 * 32 + 64  # linear-attention partial record (attention.hip)
# elements per (sample, group) up to which vmm_groupnorm_coef reduces the slice itself; measured on MI355X: one workgroup streams a
# 100-200 K-element slice in ~45 us, the two-launch path (statistics over many workgroups + coefficients) takes ~30 us, so only tiny
# layers go direct
GN_DIRECT_MAX = 1 << 14
# workgroups a launch of the exact-fp32 weight-gradient kernel is split into (row slices x 64 x 64 tiles): every workgroup ends with 4 096 fp32 atomics
WGRAD_F32_WGS = int(os.environ.get("VMM_WGRAD_F32_WGS", "2048"))
# (A/B aid: VMM_DQKV16=0 goes with a library whose single-pass objects were built with -DVMM_DQKV16=0, i.e. fp32 rows of the qkv-row gradient; tools/check_dqkv16.py)
DQKV16 = os.environ.get("VMM_DQKV16", "1") != "0"
SK_SLOTS = 512    # partial-tile slots of the balanced 3 x 3 launch (two workgroups per CU; entries 2048.. of the tickets are their flags)
N_TICKETS = 4096  # ints for the ordered split reduction of vmm_conv3x3_bf16x3 (one per output tile)
Q_STRIDE = 4096 + 16  # quantile scratch words per sample (diffusion.hip)
# op classes whose kernels have an instance over bf16-STORED feature maps ("bf16" mode; _Builder.nat16): 3 x 3 convolutions, the stride-2 resampling
# layers, the ResnetBlock output pass, the A-stationary projections (+ the res_conv tail), the temporal-attention core, the fused temporal / linear
# attention blocks, the final block's tail, the stem
A16_KERNELS = {"conv3x3", "s2", "affine", "proj", "narrow", "tattn", "tb", "la", "final", "stem"}
PACK_FIELDS = ("TH", "TW", "C", "Cp", "N", "sn", "sc", "sh", "sw", "h0", "hs", "w0", "ws", "accumulate", "fmt")


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class Act:
    """A feature map living in the arena."""
    off: int  # float offset in the arena
    C: int
    H: int
    W: int
    n: int  # elements
    ptr: int = 0
    bf: bool = False  # stored as bf16 (the "bf16" throughput mode keeps the two upper levels' feature maps in bf16, section 3 of DESIGN.md)
    na: int = -1      # floats of arena it occupies (n, or n / 2 when bf)

    def __post_init__(self):
        if self.na < 0:
            self.na = self.n // 2 if self.bf else self.n

    @property
    def ld(self):
        return self.C


@dataclass
class ActView:
    """A column slice of a feature map (same rows, `C` of its `ld` columns)."""
    ptr: int
    C: int
    ld: int
    H: int
    W: int


class _Arena:
    """First-fit allocator over float offsets; used identically in the sizing pass and the real pass."""

    def __init__(self, keep_all: bool):
        self.free_list: List[Tuple[int, int]] = []  # (off, n) sorted by off
        self.top = 0
        self.peak = 0
        self.keep_all = keep_all
        self.log: List[Tuple[int, int]] = []  # every allocation in order (debug: tools/stress_concurrent.py checksums them with VMM_KEEP_ALL=1)

    def alloc(self, n: int) -> int:
        n = (n + ALIGN - 1) // ALIGN * ALIGN
        for i, (off, sz) in enumerate(self.free_list):
            if sz >= n:
                if sz == n:
                    self.free_list.pop(i)
                else:
                    self.free_list[i] = (off + n, sz - n)
                self.log.append((off, n))
                return off
        off = self.top
        self.top += n
        self.peak = max(self.peak, self.top)
        self.log.append((off, n))
        return off

    def free(self, off: int, n: int, force: bool = False) -> None:
        if self.keep_all and not force:
            return
        n = (n + ALIGN - 1) // ALIGN * ALIGN
        fl = self.free_list
        fl.append((off, n))
        fl.sort()
        merged: List[Tuple[int, int]] = []
        for o, s in fl:
            if merged and merged[-1][0] + merged[-1][1] == o:
                merged[-1] = (merged[-1][0], merged[-1][1] + s)
            else:
                merged.append((o, s))
        if merged and merged[-1][0] + merged[-1][1] == self.top:
            self.top = merged[-1][0]
            merged.pop()
        self.free_list = merged


def _enabled(feature: str) -> bool:
    """Measurement aid: VMM_DISABLE=stem,s2,res_tail,final_tail in the environment builds plans without those fusions (A/B runs on one box)."""
    return feature not in os.environ.get("VMM_DISABLE", "").split(",")


def _pack_table(jobs: List[dict], base_of: Callable[[dict], int]):
    arr = (N.PackJob * len(jobs))()
    max_elems = 0
    for i, j in enumerate(jobs):
        a = arr[i]
        a.torch_w = base_of(j)
        a.packed = j["packed"]
        for f in PACK_FIELDS:
            setattr(a, f, j.get(f, 0))
        max_elems = max(max_elems, j["TH"] * j["TW"] * j["Cp"] * j["N"])
    return arr, max_elems


class Plan:
    def __init__(self):
        self.steps: List[Tuple[Callable, tuple, str]] = []
        self.meta: List[Tuple[str, float, float]] = []  # per step: (kernel family, algorithmic flops, algorithmic bytes)
        self.bwd_steps: List[Tuple[Callable, tuple, str]] = []
        self.bwd_meta: List[Tuple[str, float, float]] = []
        self.bwd_marks: List[Tuple[int, int]] = []  # (index into bwd_steps, X): gradients at float offsets >= X are final after that step
        self.dx_steps: List[Tuple[Callable, tuple, str]] = []  # gradient with respect to the network input (run on request, after bwd_steps)
        self.dx: Optional[torch.Tensor] = None
        self.arena: Optional[torch.Tensor] = None
        self.wbuf: Optional[torch.Tensor] = None
        self.pgrad: Optional[torch.Tensor] = None  # flat parameter gradients, torch layouts, first-use order
        self.gscratch: Optional[torch.Tensor] = None  # zeroed every backward: packed weight grads + atomically accumulated grads
        self.pack_jobs: List[dict] = []  # forward / data-gradient operand layouts (torch -> packed)
        self.pack_table = None  # (device tensor, njobs, max_elems, pointer signature)
        self.param_slices: Dict[str, Tuple[int, int]] = {}  # name -> (offset, numel) in pgrad
        self.keepalive: list = []
        self.x_in = self.time_in = self.cond_in = self.mask_in = self.out = self.dout = None
        self.focus_in = None  # (B,) uint8 focus_present_mask slot of plans built with focus=True
        self.weights_version = None
        self.named: Dict[str, Act] = {}  # debug taps (name -> feature map)
        self.shape = None
        self.training = False
        self.mirrored = False  # built for x[B/2:] == x[:B/2] (guidance): only the first half of x_in is read
        self.arena_floats = 0

    # ------------------------------------------------------------------ weights
    def refresh_weights(self, params: Dict[str, torch.Tensor]) -> None:
        """Re-pack every operand layout from the current parameter values: ONE batched HIP launch."""
        sig = tuple(params[j["name"]].data_ptr() for j in self.pack_jobs)
        if self.pack_table is None or self.pack_table[3] != sig:
            for j in self.pack_jobs:
                if not params[j["name"]].is_contiguous():
                    raise RuntimeError(f"parameter {j['name']} must be contiguous")
            arr, mx = _pack_table(self.pack_jobs, lambda j: params[j["name"]].data_ptr() + 4 * j.get("src_off", 0))
            dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self.wbuf.device)
            self.pack_table = (dev, len(self.pack_jobs), mx, sig)
        dev, n, mx, _ = self.pack_table
        N.check(N.lib().vmm_pack_weights(dev.data_ptr(), n, mx, 0, _stream()), "vmm_pack_weights")

    # ------------------------------------------------------------------ forward
    def launch(self) -> None:
        s = _stream()
        for fn, args, what in self.steps:
            rc = fn(*args, s)
            if rc != 0:
                N.check(rc, what)

    def launch_timed(self, steps=None) -> List[float]:
        """Replay with a HIP event pair around every launch (on the current stream); returns ms per step."""
        s = _stream()
        evs = []
        for fn, args, what in (self.steps if steps is None else steps):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, s)
            e1.record()
            if rc != 0:
                N.check(rc, what)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    def run(self, x, time, cond, mask, focus=None) -> torch.Tensor:
        """Copy the inputs into the static slots, replay, return the static (B,C,T,H,W) output view."""
        self.x_in.copy_(x)
        self.time_in.copy_(time)
        self.cond_in.copy_(cond)
        self.mask_in.copy_(mask)
        if self.focus_in is not None:
            self.focus_in.copy_(focus)
        elif focus is not None:
            raise RuntimeError("this plan was built without a focus_present_mask slot")
        self.launch()
        return self.out

    # ------------------------------------------------------------------ backward (training plans)
    def backward(self, dout: Optional[torch.Tensor] = None, on_mark: Optional[Callable[[int], None]] = None, want_dx: bool = False) -> torch.Tensor:
        """Run the backward launch list for the forward just executed.  `dout` = d loss / d output (B,C,T,H,W)
        (or already written to self.dout).  Returns the flat gradient buffer; `on_mark(X)` is called as soon as every
        gradient at float offsets >= X is final (dp.py starts the all-reduce of that slice on its own stream)."""
        assert self.training
        if dout is not None:
            self.dout.copy_(dout)
        self.pgrad.zero_()
        self.gscratch.zero_()
        s = _stream()
        marks = dict(self.bwd_marks)
        debug = bool(os.environ.get("VMM_DEBUG_SYNC"))
        for i, (fn, args, what) in enumerate(self.bwd_steps):
            if debug:
                print(f"[vmm bwd {i}/{len(self.bwd_steps)}] {fn.__name__}: {what}", file=sys.stderr, flush=True)
            rc = fn(*args, s)
            if rc != 0:
                N.check(rc, what)
            if debug:
                torch.cuda.synchronize()
            if on_mark is not None and i in marks:
                on_mark(marks[i])
        if want_dx:  # d loss / d x into self.dx (B, C, T, H, W)
            for fn, args, what in self.dx_steps:
                N.check(fn(*args, s), what)
        return self.pgrad

    def backward_timed(self) -> List[float]:
        """The backward launch list with a HIP event pair around every launch (bench.py's training roofline); ms per step."""
        assert self.training
        self.pgrad.zero_()
        self.gscratch.zero_()
        return self.launch_timed(self.bwd_steps)

    def grad_views(self, params: Dict[str, torch.Tensor]) -> Dict[str, torch.Tensor]:
        return {k: self.pgrad[o:o + n].view(params[k].shape) for k, (o, n) in self.param_slices.items() if k in params}

    def tap(self, name: str) -> torch.Tensor:
        """Debug: a (rows, C) view of a named intermediate (only meaningful for training plans)."""
        a = self.named[name]
        return self.arena[a.off:a.off + a.n].view(-1, a.C)


def cfg_combine(eps_c: torch.Tensor, eps_n: torch.Tensor, w: float) -> torch.Tensor:
    """null + (cond - null) * w  (vddp.py:728), same association as the reference."""
    out = torch.empty_like(eps_c)
    rc = N.lib().vmm_cfg_combine(eps_c.data_ptr(), eps_n.data_ptr(), C.c_float(w), out.data_ptr(), eps_c.numel(), _stream())
    N.check(rc, "vmm_cfg_combine")
    return out


# ====================================================================================== builder
class _Builder:
    def __init__(self, model, B, T, H, W, cond_len, device, bases: Tuple[int, int, int, int], training: bool, mirrored: bool = False,
                 focus: bool = False):
        self.m = model
        # focus: the plan takes a per-sample focus_present_mask (vddp.py:431): every temporal attention except init_temporal_attn runs its unfused
        # form with the row patches of vmm_focus_rows around the core (plans without it -- every shipped config -- are unchanged)
        self.focus = bool(focus)
        # mirrored: the caller guarantees x[B/2:] == x[:B/2] (the two branches of classifier-free guidance as one batch, vddp.py:715-728):
        # what the network computes from x alone -- the stem and init_temporal_attn, before time / conditioning enter -- is computed for
        # one half and duplicated
        self.mirrored = bool(mirrored and not training and B % 2 == 0 and _enabled("mirror"))
        self.B, self.T, self.H, self.W, self.cond_len = B, T, H, W, cond_len
        self.device = device
        self.base, self.wbase, self.pgbase, self.gsbase = bases
        self.training = training
        self.arena = _Arena(keep_all=training or bool(os.environ.get("VMM_KEEP_ALL")))  # (VMM_KEEP_ALL: debug, no slot of an inference plan is reused)
        self.wtop = self.pgtop = self.gstop = 0
        self.plan = Plan()
        self.plan.training = training
        self.lib = N.lib()
        self.shapes = {k: tuple(v.shape) for k, v in model._params_flat().items()}
        self.trainable = {k for k, p in model.named_parameters() if p.requires_grad}
        self.G = model.resnet_groups
        pm = getattr(model, "padding_mode", "zeros")  # 'circular': both image axes periodic; 'circular_1d': the horizontal one (vddp.py:163-243)
        self.wrap_h, self.wrap_w = int(pm == "circular"), int(pm in ("circular", "circular_1d"))
        self.heads = model.attn_heads
        # head width of the TEMPORAL attentions (attn_dim_head, vddp.py:582, 615); the linear and the mid spatial attention keep their constructor
        # default of 32 (Unet3D does not forward it to them, vddp.py:679, 687).  The rotary table spans min(32, dh_t) features (vddp.py:612).
        self.dh_t = int(getattr(model, "attn_dim_head", 32))
        if self.dh_t < 4 or self.dh_t > 128 or self.dh_t % 4:
            raise NotImplementedError(f"attn_dim_head = {self.dh_t}: the temporal attention kernels take head widths that are multiples of 4 in 4 .. 128")
        # split-bf16 matrix-core path for the forward contractions of inference plans (training keeps exact fp32 everywhere)
        # bf16x3: split-bf16 matrix-core GEMMs (forward, and in training also the data gradients; weight gradients stay exact fp32)
        prec = getattr(model, "train_precision" if training else "precision", "fp32")
        if prec not in ("fp32", "bf16x3", "bf16", "fp16"):
            raise ValueError(f"unknown arithmetic mode {prec!r}: 'fp32' (exact), 'bf16x3' (split-bf16, fp32-class), 'bf16' (single pass: the sampling throughput mode and the "
                             "reduced-precision training leg), 'fp16' (single pass on IEEE-half operands: the reference's own autocast dtype, main.py:34)")
        # (training in "bf16": the reduced-precision leg -- one matrix pass on bf16-rounded operands wherever a kernel has such an instance, fp32 master
        # weights, fp32 activations in HBM, fp32 accumulation / norms / softmax / optimizer; the counterpart of the reference's fp16 autocast, main.py:34)
        self.x3 = prec in ("bf16x3", "bf16", "fp16")
        # "bf16": the 3 x 3 convolutions and the fused attention blocks run ONE matrix pass on the operands' bf16 roundings (same packed weights, same
        # launch list); the bandwidth-bound kernels keep their three passes, which cost them no time.  fp32 activations in HBM either way.
        self.one = prec in ("bf16", "fp16")
        # "fp16": the single-pass kernels on IEEE-half operands (v_mfma_f32_32x32x16_f16; the `_fp16` entry points, weight operands packed as fp16 planes:
        # vmm_pack_weights fmt | 16) -- what torch autocast makes of the reference's convolutions / Linears / einsums under Accelerate(mixed_precision='fp16')
        # (main.py:34).  The bandwidth-bound contractions (1 x 1 projections, the generic implicit GEMM, the stem) keep their three split-bf16 passes: more
        # accurate than either 16-bit format and no slower.  fp32 feature maps in HBM.
        self.half = prec == "fp16"
        self.a16 = bool(self.one and not self.half and not training and getattr(model, "bf16_storage", True) and _enabled("a16"))
        # fp16 TRAINING plans: the generic implicit GEMM (the layers no specialised kernel takes: to_qkv and its data gradient at C >= 128, res_conv, the
        # transposed convolutions' phases -- and every convolution of a model narrower than 64 channels) runs its single-pass instance too (round 6).  The bf16
        # leg keeps the three-pass GEMM there: with 8 operand bits in those layers as well its gradients at dim 16 leave the reference's fp16-autocast figures
        # (median 1.55e-2 against 1.31e-2; tests/test_gpu_train.py), which that leg states it stays inside.  VMM_IGEMM_ONE=bf16 switches it on for measurements.
        self.igemm_one = bool(training and (self.half or (self.one and os.environ.get("VMM_IGEMM_ONE") == "bf16")) and _enabled("igemm_one"))
        self.a16_ops = set(os.environ.get("VMM_A16_OPS", "all").split(","))
        # exact-fp32 mode: the 3x3 and projection kernels run their v_mfma_f32_32x32x2_f32 variants on fp32 fragment-order weights (fmt 4)
        self.f32frag = not self.x3 and getattr(model, "use_f32_frag_kernels", True)
        self.tape: List[Tuple[Callable[[], None], int, int]] = []  # (backward emitter, pgtop at block start, first unpack job)
        self.unpack_jobs: List[dict] = []
        self.gacts: Dict[int, Act] = {}  # activation offset -> gradient buffer
        self.gwritten: set = set()
        self.in_bwd = False
        self.job_uploads: List[Tuple[int, torch.Tensor]] = []
        self.pending_reduce: list = []  # weight-gradient launches whose partial blocks are not totalled yet (flush_reductions)
        self.raw_slots: Dict[str, int] = {}
        self.tickets_ptr: Optional[int] = None
        self.sk_ptr: Optional[int] = None

    # ---------------------------------------------------------------- memory
    def alloc(self, n: int) -> int:
        return self.arena.alloc(n)

    def free(self, off: int, n: int) -> None:
        self.arena.free(off, n)

    def ptr(self, off: int) -> int:
        return self.base + off * 4

    def act(self, C_: int, H: int, W: int, bf: bool = False) -> Act:
        n = self.B * self.T * H * W * C_
        off = self.alloc(n // 2 if bf else n)
        return Act(off, C_, H, W, n, self.ptr(off), bool(bf))

    def free_act(self, a: Act) -> None:
        self.free(a.off, a.na)

    # ---- bf16 STORAGE of the feature maps (precision "bf16", inference): the two upper levels' maps -- the bulk of the HBM traffic -- are kept as
    # bf16, every kernel that touches them has an instance templated on the element type of its activation pointers.  An op whose kernel has no
    # such instance (VMM_A16_OPS restricts the set: a development / bisection aid) gets fp32 copies of its inputs and makes an fp32 output.
    def lvl16(self, H: int) -> bool:
        return bool(self.a16 and 2 * H >= self.H)

    def nat16(self, op: str, H: int) -> bool:
        return self.lvl16(H) and op in A16_KERNELS and ("all" in self.a16_ops or op in self.a16_ops)

    def cast(self, a: Act, bf: bool, temps: list) -> Act:
        """`a` in the wanted storage type; a converted copy (appended to `temps`, freed by the caller with free_temps) when it is stored otherwise."""
        if a is None or bool(a.bf) == bool(bf):
            return a
        # sized from the SOURCE: resnet_block's mirror_in path casts while self.B is temporarily half the batch, and vmm_convert_act converts a.n
        # elements whatever self.B says (round-4 advisor: the copy was half the size it needed, the conversion wrote past its arena slot)
        off = self.alloc(a.n // 2 if bf else a.n)
        c = Act(off, a.C, a.H, a.W, a.n, self.ptr(off), bool(bf))
        assert c.na >= (a.n // 2 if bf else a.n)
        self.step(self.lib.vmm_convert_act, (a.ptr, 1 if a.bf else 0, c.ptr, 1 if bf else 0, a.n), "storage conversion (fp32 <-> bf16)", nbytes=6.0 * a.n)
        temps.append(c)
        return c

    def free_temps(self, temps: list) -> None:
        for c in temps:
            self.free_act(c)
        temps.clear()

    def tmp_free(self, a) -> None:
        """Release a backward-phase temporary (allowed even in keep-all plans: it was allocated after every forward buffer)."""
        if isinstance(a, Act):
            self.arena.free(a.off, a.na, force=True)
        else:
            self.arena.free(a[0], a[1], force=True)

    def scratch(self, n: int) -> int:
        """Pointer to n floats that are zero at the start of every backward pass."""
        off = self.gstop
        self.gstop += (n + ALIGN - 1) // ALIGN * ALIGN
        return self.gsbase + off * 4

    def _touch(self, name: str) -> None:
        """Fix the position of a parameter's gradient in the flat buffer: order of first use in the forward."""
        if name not in self.plan.param_slices and name in self.trainable:
            n = int(math.prod(self.shapes[name]))
            self.plan.param_slices[name] = (self.pgtop, n)
            self.pgtop += (n + 3) // 4 * 4

    def pg(self, name: str) -> int:
        """Device pointer of the (torch-layout) gradient of parameter `name`; 0 if it is frozen."""
        self._touch(name)
        if name not in self.plan.param_slices:
            return 0
        return self.pgbase + self.plan.param_slices[name][0] * 4

    def wslot(self, n: int) -> int:
        off = self.wtop
        self.wtop += (n + ALIGN - 1) // ALIGN * ALIGN
        return self.wbase + off * 4

    def pack(self, name: str, n_elems: int, want_grad: bool = True, gemm: Optional[bool] = None, **desc) -> Tuple[int, int]:
        """Register an operand layout of parameter `name` (see vmm_pack_job); returns (packed ptr, packed-gradient ptr).
        gemm: the layout is a GEMM operand (forward weights, or the transposed operand of a data gradient) and takes the
        split-bf16 formats in bf16x3 mode; default = want_grad (raw parameter copies stay fp32)."""
        self._touch(name)
        frag = desc.pop("frag", False)
        half = bool(desc.pop("half", False)) and self.half  # the consumer is an `_fp16` entry point: IEEE-half planes (fmt | 16)
        n_fp32, fp32_desc = n_elems, desc
        if "fmt" in desc:  # an explicit operand format (5 / 6: the resampling layers' fragment planes); n_elems is the packed size
            assert not want_grad
        elif (want_grad if gemm is None else gemm) and self.f32frag and frag:  # fp32 fragment order (vmm_conv3x3_f32 / vmm_proj_f32)
            assert frag != 3
            kpad = (desc["TH"] * desc["TW"] * desc["Cp"] + 31) // 32 * 32
            n_elems = (desc["N"] + 31) // 32 * 32 * kpad
            desc = dict(desc, fmt=4)
        elif (want_grad if gemm is None else gemm) and self.x3:  # GEMM operand -> pre-split bf16 hi|lo
            kpad = (desc["TH"] * desc["TW"] * desc["Cp"] + 31) // 32 * 32
            if frag:  # MFMA fragment order, read straight into registers (conv3x3_bf16x3.hip; 3 = permuted k, linattn_block.hip)
                n_elems = (desc["N"] + 31) // 32 * 32 * kpad
                desc = dict(desc, fmt=3 if frag == 3 else 2)
            else:     # [N][Kpad] planes, staged through LDS (igemm_bf16x3.hip)
                n_elems = desc["N"] * kpad
                desc = dict(desc, fmt=1)
        if self.igemm_one and self.half and desc.get("fmt") == 1:
            half = True  # the [N][Kpad] planes have one consumer family, the generic implicit GEMM: vmm_conv_igemm_fp16 in these plans
        if half:
            assert desc.get("fmt") in (1, 2, 3, 5, 6), "fp16 planes exist for the matrix-operand formats"
            desc = dict(desc, fmt=desc["fmt"] | 16)
        ptr = self.wslot(n_elems)
        job = dict(name=name, packed=ptr, **desc)
        self.plan.pack_jobs.append(job)
        gptr = 0
        if self.training and want_grad and name in self.trainable:
            # weight gradients are always produced (vmm_conv_wgrad_f32) and scattered back in the fp32 k-major layout
            gptr = self.scratch(n_fp32)
            self.unpack_jobs.append(dict(name=name, packed=gptr, accumulate=1, **fp32_desc))
        return ptr, gptr

    def sp(self, stem: str, tail: str = ""):
        """Entry point of a kernel family for the plan's arithmetic: stem + 'bf16x3' | 'bf16' | 'fp16' + tail (single-pass instances where the mode has them)."""
        return getattr(self.lib, stem + ("fp16" if self.half else "bf16" if self.one else "bf16x3") + tail)

    def wraw(self, name: str) -> int:
        """Static copy of a parameter in its torch layout (one per parameter)."""
        if name not in self.raw_slots:
            n = int(math.prod(self.shapes[name]))
            self.raw_slots[name] = self.pack(name, n, want_grad=False, TH=1, TW=1, C=1, Cp=1, N=n, sn=1)[0]
        return self.raw_slots[name]

    def pack_conv(self, name: str, pad_cin_to: int = 0, frag: bool = False) -> Tuple[int, int]:
        """(Cout, Cin, 1, KH, KW) -> [(kh, kw, ci)][Cout]  (fragment order = the 3 x 3 halo kernel's operand: fp16 planes in an fp16 plan)"""
        co, ci, _, kh, kw = self.shapes[name]
        cip = max(ci, pad_cin_to)
        return self.pack(name, kh * kw * cip * co, TH=kh, TW=kw, C=ci, Cp=cip, N=co, sn=ci * kh * kw, sc=kh * kw, sh=kw, sw=1, hs=1, ws=1, frag=frag, half=bool(frag))

    def halo_ok(self, c1: int, c2: int, cout: int, H: int, W: int) -> bool:
        """Envelope of vmm_conv3x3_bf16x3 (LDS halo patch + register-fed fragment-order weights)."""
        if not ((self.x3 or self.f32frag) and getattr(self.m, "use_halo_conv", True)):
            return False
        if c1 % 32 or c2 % 32 or not (cout == 64 or cout % 128 == 0):
            return False
        bm = 128 if cout >= 128 else 256
        if W >= 32 and W % 16 == 0 and H % (bm // 16) == 0:
            return True
        if self.wrap_h or self.wrap_w:  # periodic padding: the 2-D-tiled instances only (flat row tiles assume a contiguous neighbourhood)
            return False
        return bm + 2 * (W + 1) <= (6 if cout >= 128 else 11) * 32

    def pack_conv_dgrad(self, name: str, ci0: int, nci: int, flip: bool = False, frag=False, gemm: bool = False) -> int:
        """(Cout, Cin, 1, KH, KW) -> [(kh, kw, co)][ci0 : ci0+nci]  (data gradient of a stride-1 conv).  flip = False: taps in forward
        order, the descriptor mirrors them (off = +pad, sgn = -1); flip = True: taps reversed here, so that the data gradient is an
        ordinary 'same' convolution of dY (what vmm_conv3x3_bf16x3 runs).  gemm: split-bf16 operand in bf16x3 mode."""
        co, ci, _, kh, kw = self.shapes[name]
        geo = dict(h0=kh - 1, hs=-1, w0=kw - 1, ws=-1) if flip else dict(hs=1, ws=1)
        return self.pack(name, kh * kw * co * nci, want_grad=False, gemm=gemm, TH=kh, TW=kw, C=co, Cp=co, N=nci, sn=kh * kw, sc=ci * kh * kw, sh=kw, sw=1,
                         src_off=ci0 * kh * kw, frag=frag, half=bool(frag), **geo)[0]

    def pack_linear(self, name: str, frag=False, half: bool = False) -> Tuple[int, int]:
        """(out, in[,1,1[,1]]) -> [in][out]   (half: the operand of an `_fp16` entry point -- the fused attention blocks -- in an fp16 plan)"""
        shp = self.shapes[name]
        co, ci = shp[0], shp[1]
        return self.pack(name, co * ci, TH=1, TW=1, C=ci, Cp=ci, N=co, sn=ci, sc=1, frag=frag, half=half)

    def pack_linear_slice(self, name: str, ci0: int, nci: int, frag=False, gemm: bool = False, half: bool = False) -> int:
        """torch (out, in) restricted to input columns [ci0, ci0+nci) as [out][nci]: the k-major operand of the data gradient."""
        co, ci = self.shapes[name][0], self.shapes[name][1]
        return self.pack(name, co * nci, want_grad=False, gemm=gemm, TH=1, TW=1, C=co, Cp=co, N=nci, sn=1, sc=ci, src_off=ci0, frag=frag, half=half)[0]

    def dgrad_3x3(self, name: str, ci0: int, nci: int, what: str, **kw) -> None:
        """dX[:, ci0:ci0+nci] (+)= data gradient of the 3x3 'same' conv with weight `name`: with the taps reversed it is itself a
        3x3 'same' convolution of dY, so bf16x3 mode runs it on vmm_conv3x3_bf16x3 where the envelope fits."""
        co = self.shapes[name][0]
        a1 = kw["a1"]
        if self.halo_ok(co, 0, nci, a1.H, a1.W):
            w = self.pack_conv_dgrad(name, ci0, nci, flip=True, frag=True, gemm=True)
            self.conv(w=w, Cout=nci, KH=3, KW=3, off=(-1, -1), what=what, halo=True, x3w=True, **kw)
        else:
            w = self.pack_conv_dgrad(name, ci0, nci, gemm=self.x3)
            self.conv(w=w, Cout=nci, KH=3, KW=3, off=(1, 1), sgn=(-1, -1), what=what, x3w=self.x3, **kw)

    def dgrad_1x1(self, name: str, ci0: int, nci: int, what: str, **kw) -> None:
        """dX[:, ci0:ci0+nci] (+)= dY . W[:, ci0:ci0+nci] for a 1x1 conv / Linear weight `name` (out, in): split-bf16 projection kernel in
        bf16x3 mode (K = out features), exact fp32 implicit GEMM otherwise."""
        co = self.shapes[name][0]
        pj = self.proj_ok(co, nci)
        w = self.pack_linear_slice(name, ci0, nci, frag=2 if pj else False, gemm=self.x3 or pj)
        self.conv(w=w, Cout=nci, what=what, proj=pj, x3w=self.x3, **kw)

    def step(self, fn, args: tuple, what: str, flops: float = 0.0, nbytes: float = 0.0) -> None:
        if self.in_bwd:
            self.plan.bwd_steps.append((fn, args, what))
            self.plan.bwd_meta.append((fn.__name__, float(flops), float(nbytes)))
        else:
            self.plan.steps.append((fn, args, what))
            self.plan.meta.append((fn.__name__, float(flops), float(nbytes)))

    def on_backward(self, emit: Callable[[], None], pg_start: int, uj_start: int) -> None:
        if self.training:
            self.tape.append((emit, pg_start, uj_start))

    # ---------------------------------------------------------------- gradient bookkeeping
    def grad_of(self, a: Act) -> Tuple[Act, int]:
        """(gradient buffer of activation a, accumulate flag for the next writer)."""
        g = self.gacts.get(a.off)
        if g is None:
            off = self.alloc(a.n)
            g = Act(off, a.C, a.H, a.W, a.n, self.ptr(off))
            self.gacts[a.off] = g
        acc = 1 if a.off in self.gwritten else 0
        self.gwritten.add(a.off)
        return g, acc

    def add_into(self, dst: Act, src_ptr: int, src_act: Optional[Act] = None) -> None:
        """grad(dst) += src (or = src for the first writer).  src_act: src is the COMPLETE gradient buffer of a residual block's output that
        nobody reads after this block's own backward launches -- when dst has no gradient yet it simply takes that buffer over (the block's
        last kernel accumulates into it in place, after every read of it): one 12-byte-per-element pass less per residual block."""
        if src_act is not None and dst.off not in self.gacts and src_act.n == dst.n and _enabled("grad_alias"):
            self.gacts[dst.off] = src_act
            self.gwritten.add(dst.off)
            return
        g, acc = self.grad_of(dst)
        self.step(self.lib.vmm_lincomb, (src_ptr, g.ptr if acc else None, None, 1.0, 1.0, 0.0, 0.0, g.ptr, g.n), "grad accumulate", nbytes=12.0 * g.n)

    # ---------------------------------------------------------------- emitters
    def conv_desc(self, *, a1: Act, a2: Optional[Act] = None, w: int, bias: int = 0, Cout: int, KH: int = 1, KW: int = 1, stride: int = 1,
                  off: Tuple[int, int] = (0, 0), sgn: Tuple[int, int] = (1, 1), out_ptr: int, ldo: int, Hv: int, Wv: int, Hout: int = 0,
                  Wout: int = 0, oscale: int = 1, oo: Tuple[int, int] = (0, 0), res_ptr: int = 0, ldres: int = 0, rot_tab: int = 0,
                  rot_ncols: int = 0, q_scale: float = 1.0, q_ncols: int = 0, a_coef: int = 0, out_bf: bool = False) -> "N.ConvDesc":
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1 = a1.ptr, a1.C, a1.ld
        if a2 is not None:
            d.a2, d.C2, d.lda2 = a2.ptr, a2.C, a2.ld
            assert bool(getattr(a2, "bf", False)) == bool(getattr(a1, "bf", False)), "the two sources of a concatenation share a storage type"
        # storage of the feature maps: bit 0 = a1 / a2 are bf16, bit 1 = out (and res, which is read at out's rows) are bf16
        d.act_bf16 = (1 if getattr(a1, "bf", False) else 0) | (2 if out_bf else 0)
        self._desc_a16 = bool(d.act_bf16)
        d.w, d.bias = w, bias or None
        d.res, d.ldres = res_ptr or None, ldres
        d.out, d.ldo = out_ptr, ldo
        d.nimg, d.Hin, d.Win = self.B * self.T, a1.H, a1.W
        d.Hv, d.Wv, d.stride = Hv, Wv, stride
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = KH, KW, off[0], off[1], sgn[0], sgn[1]
        d.Hout, d.Wout, d.oscale, d.ooh, d.oow = Hout or Hv, Wout or Wv, oscale, oo[0], oo[1]
        d.Cout = Cout
        d.rot_tab = rot_tab or None
        d.rot_T, d.rot_HW, d.rot_ncols, d.rot_dh = self.T, a1.H * a1.W, rot_ncols, self.dh_t
        d.q_scale, d.q_ncols = q_scale, q_ncols
        d.a_mode, d.a_coef, d.a_imgs_per_sample = (1 if a_coef else 0), a_coef or None, self.T
        if self.tickets_ptr is None:  # zero-initialised (wbuf is) and left zero by the kernel; shared by all convs of the (single-stream) plan
            self.tickets_ptr = self.wslot(N_TICKETS)
        d.split_tickets, d.n_tickets = self.tickets_ptr, N_TICKETS
        if KH == 3 and KW == 3 and stride == 1 and Cout >= 128 and not self._desc_a16 and _enabled("conv_sk") and N.experiments_built():
            # workspace of the balanced launch of the few-tile 3 x 3 layers (conv3x3_sk_kernel): one partial 128 x 128 tile per workgroup of a grid of two
            # per CU; shared by every convolution of the (single-stream) plan like the tickets
            if self.sk_ptr is None:
                self.sk_ptr = self.wslot(SK_SLOTS * 128 * 128)
            d.sk_work, d.sk_slots = self.sk_ptr, SK_SLOTS
        if self._desc_a16 or not _enabled("conv_split"):  # bf16-stored maps: the unsplit instances (no ordered atomic accumulation onto a bf16 output)
            d.split_tickets, d.n_tickets = None, 0
        d.gn_part, d.gn_groups = None, self.G
        if KH > 1 or KW > 1:  # padding_mode 'circular' / 'circular_1d' (vddp.py:163-243): every spatial kernel wraps instead of zero-padding
            d.wrap_h, d.wrap_w = self.wrap_h, self.wrap_w
        self.plan.keepalive.append(d)
        return d

    def tickets(self) -> int:
        """The plan's zero-initialised split tickets (shared by every launch that may split a channel reduction; one stream)."""
        if self.tickets_ptr is None:
            self.tickets_ptr = self.wslot(N_TICKETS)
        return self.tickets_ptr

    def proj_ok(self, k: int, cout: int) -> bool:
        """Envelope of vmm_proj_bf16x3 (1x1 / Linear with an A-stationary LDS row tile and fragment-order weights)."""
        if not ((self.x3 or self.f32frag) and getattr(self.m, "use_proj_kernel", True)):
            return False
        kp = (k + 31) // 32 * 32
        if self.narrow_ok(k, cout):  # wide contraction, 64 output columns: the barrier-free streaming kernel (narrow_proj.hip), same fmt-2 weights
            return True
        if kp == 256 and cout <= 64:  # one 64-row tile per CU and a single column slice: no faster than the streaming implicit GEMM (measured twice)
            return False
        return kp in (32, 64, 128, 256) and k % 4 == 0 and cout % 32 == 0

    def qkv_backward(self, dq: "N.ConvDesc", wname: str, x: Act, gqkv: Act, gwq: int, what: str, ln_gamma_name: Optional[str] = None) -> Optional[Act]:
        """Backward of to_qkv: the weight gradient into gwq and the data gradient gy (returned; the caller's LayerNorm backward consumes and frees it).
        At the C = 64 levels both come from ONE pass over the 3 KB rows of gqkv (qkv_bwd.hip); elsewhere a weight-gradient and a data-gradient launch.
        ln_gamma_name (the PreNorm's gamma; only with the forward's LayerNorm statistics in dq._ln): where the one-pass kernel takes the layer, the
        LayerNorm backward runs as its epilogue (vmm_qkv_bwd_ln_*) -- the gradient of x is complete, gy never exists and None is returned."""
        rows = self.B * self.T * x.H * x.W
        n_out = gqkv.C
        ws_n = int(self.lib.vmm_qkv_bwd_workspace(rows, x.C, n_out)) if (self.x3 and gwq and dq is not None and getattr(self.m, "use_x3_wgrad", True)
                                                                          and _enabled("qkv_bwd")) else 0
        ln = getattr(dq, "_ln", None) if dq is not None else None
        if ws_n and ln and ln_gamma_name and not (x.ld & 3) and _enabled("qkv_bwd_ln"):
            wd = self.pack_linear_slice(wname, 0, x.C, frag=2, gemm=True, half=True)
            ws = self.alloc(ws_n)
            gx, acc = self.grad_of(x)
            gg = self.pg(ln_gamma_name) or self.scratch(x.C)
            self.step(self.sp("vmm_qkv_bwd_ln_"),
                      (dq.a1, dq.lda1, ln[0], ln[1], gqkv.ptr, n_out, wd, gx.ptr, x.C, acc, gg, gwq, self.ptr(ws), rows, x.C, n_out),
                      what + " backward (data + weight gradient, LayerNorm backward)", flops=4.0 * rows * x.C * n_out, nbytes=rows * ((2.0 if gqkv.bf else 4.0) * n_out + 12.0 * x.C))
            self.tmp_free((ws, ws_n))
            return None
        g16 = None
        if gqkv.bf and not ws_n:
            # the one-pass kernel does not take the shape (rows no multiple of 64: small geometries) or is switched off: the separate launches below read fp32
            # rows -- the 16-bit rows widened exactly (they round to the same 16 bits again)
            g16, gqkv = gqkv, self.act(n_out, x.H, x.W)
            self.step(self.sp("vmm_dqkv_widen_"), (g16.ptr, gqkv.ptr, rows * n_out), what + " (16-bit rows of the qkv gradient widened)", nbytes=6.0 * rows * n_out)
        gy = self.act(x.C, x.H, x.W)
        if ws_n:
            wd = self.pack_linear_slice(wname, 0, x.C, frag=2, gemm=True, half=True)
            ws = self.alloc(ws_n)
            self.step(self.sp("vmm_qkv_bwd_"), (dq.a1, dq.lda1, ln[0] if ln else None, ln[1] if ln else None, gqkv.ptr, n_out, wd, gy.ptr, x.C, gwq,
                                                    self.ptr(ws), rows, x.C, n_out), what + " backward (data + weight gradient)",
                      flops=4.0 * rows * x.C * n_out, nbytes=rows * ((2.0 if gqkv.bf else 4.0) * n_out + 8.0 * x.C))
            self.tmp_free((ws, ws_n))
            return gy
        self.wgrad(dq, gqkv.ptr, n_out, gwq, what)
        self.dgrad_1x1(wname, 0, x.C, what + " dgrad", a1=gqkv, out_ptr=gy.ptr, ldo=x.C, Hv=x.H, Wv=x.W)
        if g16 is not None:
            self.tmp_free(gqkv)  # (the widened copy; the caller frees the 16-bit rows it allocated)
        return gy

    def ln_fused_training_ok(self, k: int, cout: int) -> bool:
        """Training forward of PreNorm(to_qkv / to_q) with the LayerNorm fused into the projection's row staging (statistics kept, vmm_proj_bf16x3_ln_stats):
        possible when the 1 x 1 split-bf16 weight-gradient kernel (which re-normalises x from those statistics) takes the layer."""
        return bool(self.training and self.x3 and getattr(self.m, "use_x3_wgrad", True) and _enabled("wgrad1x1") and _enabled("ln_fused_training")
                    and k % 64 == 0 and cout % 64 == 0 and self.proj_ok(k, cout) and not self.narrow_ok(k, cout))

    def narrow_ok(self, k: int, cout: int) -> bool:
        """Envelope of vmm_proj_narrow_bf16x3 (K >= 256 -> 64 columns, split-bf16 only): to_out and the to_qkv data gradient at the C = 64 levels."""
        return bool(self.x3 and cout == 64 and k >= 256 and k % 32 == 0 and getattr(self.m, "use_proj_kernel", True) and _enabled("narrow"))

    def conv(self, what: str = "conv", halo: bool = False, proj: bool = False, ln_gamma: int = 0, x3w: bool = False, ln_stats: int = 0, **kw) -> "N.ConvDesc":
        d = self.conv_desc(**kw)
        M = d.nimg * d.Hv * d.Wv
        K = d.KH * d.KW * (d.C1 + d.C2)
        # algorithmic work: every input element, weight and output element touched once
        nbytes = 4.0 * (d.nimg * d.Hin * d.Win * (d.C1 + d.C2) + K * d.Cout + M * d.Cout + (M * d.Cout if kw.get("res_ptr") else 0))
        if proj:  # weights were packed in fragment order for it (proj_ok); ln_gamma: PreNorm LayerNorm fused into the row staging
            if self.narrow_ok(K, d.Cout) and not ln_gamma and not d.rot_ncols and not d.q_ncols:
                self.step(self.lib.vmm_proj_narrow_bf16x3, (C.byref(d),), what, flops=2.0 * M * K * d.Cout, nbytes=nbytes)
                return d
            if ln_stats:  # training forward: the statistics stay for the weight gradient
                d._ln = (ln_stats, ln_gamma)
                self.step(self.lib.vmm_proj_bf16x3_ln_stats, (C.byref(d), ln_gamma, C.c_float(1e-5), ln_stats), what, flops=2.0 * M * K * d.Cout, nbytes=nbytes)
                return d
            # (fp16 plans: the projections keep their three split-bf16 passes -- bandwidth-bound kernels, and their packed weights serve the x3-only
            # instances of the same family (LayerNorm statistics, narrow streaming) as well)
            fn = (self.lib.vmm_proj_bf16 if (self.one and not self.half) else self.lib.vmm_proj_bf16x3) if self.x3 else self.lib.vmm_proj_f32
            self.step(fn, (C.byref(d), ln_gamma or None, C.c_float(1e-5)), what, flops=2.0 * M * K * d.Cout, nbytes=nbytes)
            return d
        assert not ln_gamma
        fn = self.lib.vmm_conv3x3_f32 if halo and not self.x3 else self.lib.vmm_conv_igemm_f32
        if self.x3 and (x3w or not self.in_bwd):  # x3w: a backward GEMM whose weight operand was packed split-bf16 (pack(..., gemm=True))
            fn = self.sp("vmm_conv_igemm_") if self.igemm_one else self.lib.vmm_conv_igemm_bf16x3
            if halo:  # weights were packed in fragment order for it (halo_ok)
                fn = self.sp("vmm_conv3x3_")
        self.step(fn, (C.byref(d),), what, flops=2.0 * M * K * d.Cout, nbytes=nbytes)
        return d

    def split_k_ok(self, k: int, cout: int) -> bool:
        """K = 512 projections (inference): two K = 256 launches of the A-stationary projection kernel, the second adding onto the first
        (its epilogue's scale / rotary are linear), instead of one generic implicit GEMM at ~95 TFLOP/s."""
        return bool(self.x3 and not self.training and k == 512 and cout >= 128 and self.proj_ok(256, cout) and _enabled("split_k"))

    def proj_split_k(self, y: Act, wname: str, Cout: int, out_ptr: int, what: str, **epi) -> None:
        half = y.C // 2
        for i in range(2):
            w = self.pack(wname, Cout * half, want_grad=False, gemm=True, TH=1, TW=1, C=half, Cp=half, N=Cout, sn=y.C, sc=1, src_off=i * half, frag=2)[0]
            a = ActView(y.ptr + 4 * i * half, half, y.C, y.H, y.W)
            self.conv(a1=a, w=w, Cout=Cout, out_ptr=out_ptr, ldo=Cout, Hv=y.H, Wv=y.W, what=f"{what} (K half {i})", proj=True,
                      res_ptr=out_ptr if i else 0, ldres=Cout, **epi)

    def wgrad(self, d: "N.ConvDesc", dy_ptr: int, lddy: int, gw_ptr: int, what: str, gb_ptr: int = 0) -> None:
        """Weight gradient of the layer with forward descriptor d; gb_ptr: its bias gradient (column sums of dY) from the same launch."""
        if not gw_ptr:
            if gb_ptr and d.ooh == 0 and d.oow == 0:  # (one call per layer: phase (0, 0) of a transposed convolution covers all its rows)
                self.colsum(dy_ptr, lddy, d.nimg * d.Hout * d.Wout, d.Cout, gb_ptr, what)
            return
        M = d.nimg * d.Hv * d.Wv
        K = d.KH * d.KW * (d.C1 + d.C2)
        wbytes = 4.0 * (d.nimg * d.Hin * d.Win * (d.C1 + d.C2) + M * d.Cout + K * d.Cout)
        if self.x3 and getattr(self.m, "use_x3_wgrad", True):
            # the 3 x 3 layers: nine taps per workgroup on the split-bf16 matrix cores, partial blocks in a workspace + fixed-order reduction
            ws_n = int(self.lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d), lddy))
            if ws_n:
                ws = self.alloc(ws_n)
                dd = self.deferred(d)
                self.step(self.sp("vmm_conv3x3_wgrad_"), (C.byref(dd), dy_ptr, lddy, gw_ptr, gb_ptr or None, self.ptr(ws)), what + " wgrad",
                          flops=2.0 * M * K * d.Cout, nbytes=wbytes)
                self.reduce_later(1, dd, lddy, gw_ptr, gb_ptr, ws, ws_n)
                return
            # the 1 x 1 layers (to_qkv, to_out, res_conv): 128 x 128 channel blocks on the split-bf16 matrix cores, same reduction scheme
            ws_n = int(self.lib.vmm_conv1x1_wgrad_bf16x3_workspace(C.byref(d), lddy)) if _enabled("wgrad1x1") else 0
            ln = getattr(d, "_ln", None)  # the forward fused the PreNorm LayerNorm: d.a1 is the un-normalised x
            if ln and not ws_n:
                raise RuntimeError("LayerNorm-fused training forward without the 1 x 1 split-bf16 weight-gradient kernel")
            if ws_n:
                ws = self.alloc(ws_n)
                dd = self.deferred(d)
                if ln:
                    assert not gb_ptr
                    self.step(self.sp("vmm_conv1x1_wgrad_", "_ln"), (C.byref(dd), dy_ptr, lddy, gw_ptr, self.ptr(ws), ln[0], ln[1]), what + " wgrad (LayerNorm operand)",
                              flops=2.0 * M * K * d.Cout, nbytes=wbytes)
                else:
                    self.step(self.sp("vmm_conv1x1_wgrad_"), (C.byref(dd), dy_ptr, lddy, gw_ptr, gb_ptr or None, self.ptr(ws)), what + " wgrad",
                              flops=2.0 * M * K * d.Cout, nbytes=wbytes)
                self.reduce_later(2, dd, lddy, gw_ptr, gb_ptr, ws, ws_n)
                return
        if (self.x3 and getattr(self.m, "use_x3_wgrad", True) and not getattr(self.m, "use_x3_wgrad_generic", False) and _enabled("wgrad_tap")
                and not getattr(d, "_ln", None)):
            # every other geometry (the 4 x 4 stride-2 layers, the transposed ones' phases, the stem): the 1 x 1 kernel with a tap-decoding loader (round 6;
            # the exact-fp32 kernel below before)
            ws_n = int(self.lib.vmm_conv_wgrad_tap_workspace(C.byref(d), lddy))
            if ws_n:
                ws = self.alloc(ws_n)
                # (the four phases of a transposed convolution add into ONE bias gradient: their second stages run one after the other, each behind its own launch --
                # jobs of one batched launch must not share a destination)
                dd = d if (gb_ptr and d.oscale > 1) else self.deferred(d)
                self.step(self.sp("vmm_conv_wgrad_tap_"), (C.byref(dd), dy_ptr, lddy, gw_ptr, gb_ptr or None, self.ptr(ws)), what + " wgrad", flops=2.0 * M * K * d.Cout,
                          nbytes=wbytes)
                self.reduce_later(3, dd, lddy, gw_ptr, gb_ptr, ws, ws_n)
                return
        if self.x3 and getattr(self.m, "use_x3_wgrad_generic", False):  # opt-in: 128 x 128 tiles, split-bf16 operands (measured slower)
            fn = self.lib.vmm_conv_wgrad_bf16x3
            tiles = -(-K // 128) * -(-d.Cout // 128)
            nsplit = max(1, min(-(-1024 // tiles), max(1, M // 512)))
        else:                                                  # 64 x 64 workgroup tiles, exact fp32
            fn = self.lib.vmm_conv_wgrad_f32
            tiles = -(-K // 64) * -(-d.Cout // 64)
            nsplit = max(1, min(-(-WGRAD_F32_WGS // tiles), max(1, M // 256)))
        sc = self.alloc(nsplit * d.Cout) if gb_ptr else None  # one partial row of the bias gradient per row slice
        self.step(fn, (C.byref(d), dy_ptr, lddy, gw_ptr, nsplit, gb_ptr or None, self.ptr(sc) if gb_ptr else None), what + " wgrad", flops=2.0 * M * K * d.Cout,
                  nbytes=4.0 * (d.nimg * d.Hin * d.Win * (d.C1 + d.C2) + M * d.Cout + K * d.Cout))
        if gb_ptr:
            self.tmp_free((sc, nsplit * d.Cout))

    # ---- second stages of the weight-gradient launches, batched: every 3 x 3 / 1 x 1 weight gradient leaves one partial block per row slice; instead of a
    # 5-13 us totalling launch behind each of them (70 per step) the blocks stay in their workspaces and ONE launch totals everything that is pending when the
    # packed gradients are next scattered (flush_reductions; same summation order: the gradients are the same bits).  VMM_DISABLE=defer_reduce: per layer again.
    def deferred(self, d: "N.ConvDesc") -> "N.ConvDesc":
        if not (self.in_bwd and _enabled("defer_reduce")):
            return d
        dd = N.ConvDesc.from_buffer_copy(d)  # (the forward launches keep the original)
        dd.defer_reduce = 1
        self.plan.keepalive.append(dd)
        return dd

    def reduce_later(self, kind: int, dd: "N.ConvDesc", lddy: int, gw_ptr: int, gb_ptr: int, ws: int, ws_n: int) -> None:
        if dd.defer_reduce:
            self.pending_reduce.append((kind, dd, lddy, gw_ptr, gb_ptr, ws, ws_n))
        else:
            self.tmp_free((ws, ws_n))

    def flush_reductions(self) -> None:
        if not self.pending_reduce:
            return
        jobs = (N.ReduceJob * len(self.pending_reduce))()
        dests = [p[3] for p in self.pending_reduce] + [p[4] for p in self.pending_reduce if p[4]]
        assert len(dests) == len(set(dests)), "two jobs of one batched reduction share a destination (their += would race)"
        wg = 0
        for i, (kind, dd, lddy, gw_ptr, gb_ptr, ws, ws_n) in enumerate(self.pending_reduce):
            fn = {1: self.lib.vmm_conv3x3_wgrad_reduce_job, 2: self.lib.vmm_conv1x1_wgrad_reduce_job, 3: self.lib.vmm_conv_wgrad_tap_reduce_job}[kind]
            rc = fn(C.byref(dd), lddy, gw_ptr, gb_ptr or None, self.ptr(ws), C.byref(jobs[i]))
            if rc != 0:
                raise RuntimeError(f"no reduction job for a deferred weight-gradient launch (kind {kind}, rc {rc})")
            jobs[i].wg0 = wg
            wg += jobs[i].wgs
        n = len(self.pending_reduce)
        self.step(self.lib.vmm_reduce_batch, (self._upload_table(jobs), n, wg), f"totals of the partial blocks of {n} weight-gradient launches",
                  nbytes=4.0 * sum(p[6] for p in self.pending_reduce))
        for _, _, _, _, _, ws, ws_n in self.pending_reduce:
            self.tmp_free((ws, ws_n))
        self.pending_reduce = []

    def colsum(self, x_ptr: int, ldx: int, rows: int, C_: int, out_ptr: int, what: str) -> None:
        if out_ptr:
            self.step(self.lib.vmm_colsum_accumulate, (x_ptr, ldx, rows, C_, out_ptr), what + " bias grad", nbytes=4.0 * rows * C_)

    def gn_coef(self, h: Act, prefix: str, film_ptr: int, ldfilm: int, conv_desc: Optional["N.ConvDesc"] = None, mirror: bool = False):
        """GroupNorm statistics of h and the fused (scale, shift) coefficients; returns (coef_off, n, ptr, stats_ptr).
        conv_desc: descriptor of the vmm_conv3x3_bf16x3 launch that produced h -- where that kernel can, it leaves per-workgroup
        partial sums of its output behind (conv3x3 epilogue) and the statistics pass over h is skipped.
        mirror (self.B is HALF the batch, h was computed for that half and holds for both): coefficients for both halves -- the same
        statistics under each half's own FiLM rows -- from two coefficient launches over the one set of partial sums."""
        B, G, C_ = self.B, self.G, h.C
        rows_ps = self.T * h.H * h.W
        n_part = 0
        if conv_desc is not None:
            conv_desc.gn_part, conv_desc.gn_groups = 1, G  # placeholder pointer for the host-side query
            n_part = int(self.lib.vmm_conv3x3_fuses_gn(C.byref(conv_desc)))
            conv_desc.gn_part = None
        coef_off = self.alloc((2 if mirror else 1) * B * C_ * 2)
        stats_ptr = self.ptr(self.alloc(B * G * 2)) if self.training else 0
        coef_args = (rows_ps * (C_ // G), C.c_float(1e-5), self.wraw(prefix + ".norm.weight"), self.wraw(prefix + ".norm.bias"), film_ptr or None, ldfilm,
                     B, C_, G, self.ptr(coef_off), stats_ptr or None)
        if n_part:
            part_off = self.alloc(B * G * n_part * 2)
            conv_desc.gn_part = self.ptr(part_off)
            self.step(self.lib.vmm_groupnorm_coef, (None, *coef_args, self.ptr(part_off), n_part, None, 0), prefix + ".norm coef")
            if mirror:
                args2 = (rows_ps * (C_ // G), C.c_float(1e-5), self.wraw(prefix + ".norm.weight"), self.wraw(prefix + ".norm.bias"),
                         (film_ptr + 4 * B * ldfilm) if film_ptr else None, ldfilm, B, C_, G, self.ptr(coef_off + B * C_ * 2), None)
                self.step(self.lib.vmm_groupnorm_coef, (None, *args2, self.ptr(part_off), n_part, None, 0), prefix + ".norm coef (second half)")
            self.free(part_off, B * G * n_part * 2)
        elif h.bf:  # (outside the fusing envelope: the statistics pass reads an fp32 copy)
            tmp: list = []
            hf = self.cast(h, False, tmp)
            nslots = int(self.lib.vmm_groupnorm_stats_slots(B, rows_ps, C_))
            part_off = self.alloc(B * G * nslots * 2)
            self.step(self.lib.vmm_groupnorm_stats_partials, (hf.ptr, hf.ld, B, rows_ps, C_, G, self.ptr(part_off)), prefix + ".norm stats", nbytes=4.0 * h.n)
            self.step(self.lib.vmm_groupnorm_coef, (None, *coef_args, self.ptr(part_off), nslots, None, 0), prefix + ".norm coef")
            self.free(part_off, B * G * nslots * 2)
            self.free_temps(tmp)
        elif rows_ps * (C_ // G) <= GN_DIRECT_MAX and (C_ // G) % 4 == 0:
            # small layer: one workgroup per (sample, group) reduces its slice itself (fixed order, no statistics launch, no atomics)
            self.step(self.lib.vmm_groupnorm_coef, (None, *coef_args, None, 0, h.ptr, h.ld), prefix + ".norm stats+coef", nbytes=4.0 * h.n)
        else:
            nslots = int(self.lib.vmm_groupnorm_stats_slots(B, rows_ps, C_))  # one (sum, sum of squares) slot per workgroup of the statistics pass
            part_off = self.alloc(B * G * nslots * 2)
            self.step(self.lib.vmm_groupnorm_stats_partials, (h.ptr, h.ld, B, rows_ps, C_, G, self.ptr(part_off)), prefix + ".norm stats", nbytes=4.0 * h.n)
            self.step(self.lib.vmm_groupnorm_coef, (None, *coef_args, self.ptr(part_off), nslots, None, 0), prefix + ".norm coef")
            self.free(part_off, B * G * nslots * 2)
        if mirror and not n_part:
            raise RuntimeError("mirrored ResnetBlock without fused GroupNorm partial sums (vmm_conv3x3_accepts and vmm_conv3x3_fuses_gn disagree)")
        return coef_off, (2 if mirror else 1) * B * C_ * 2, self.ptr(coef_off), stats_ptr

    def gn_bwd(self, prefix: str, dz_ptr: int, h: Act, coef_ptr: int, stats_ptr: int, film_ptr: int, ldfilm: int, dh_ptr: int, dfilm_ptr: int) -> None:
        B, G, C_ = self.B, self.G, h.C
        n_sc = int(self.lib.vmm_groupnorm_bwd_scratch(B, self.T * h.H * h.W, C_, G))
        sc = self.alloc(n_sc)
        gnw = self.pg(prefix + ".norm.weight") or self.scratch(C_)
        gnb = self.pg(prefix + ".norm.bias") or self.scratch(C_)
        self.step(self.lib.vmm_groupnorm_bwd,
                  (dz_ptr, C_, h.ptr, h.ld, coef_ptr, stats_ptr, self.wraw(prefix + ".norm.weight"), self.wraw(prefix + ".norm.bias"), film_ptr or None, ldfilm,
                   B, self.T * h.H * h.W, C_, G, self.ptr(sc), dh_ptr, C_, 0, gnw, gnb, dfilm_ptr or None), prefix + ".norm bwd", nbytes=20.0 * h.n)
        self.tmp_free((sc, n_sc))

    def resnet_block(self, name: str, x1: Act, x2: Optional[Act], film: Optional[Tuple[int, int]], tail: Optional[Callable] = None,
                     mirror_in: bool = False) -> Optional[Act]:
        """ResnetBlock (vddp.py:287-311): conv-GN-FiLM-SiLU, conv-GN-SiLU, + res_conv(x).  film = (ptr, grad ptr).
        tail (inference only): emits the consumer of the block's output fused with the output pass -- called with (h2, coef ptr, residual ptr,
        residual ld) instead of vmm_affine_silu; the block then returns None (its output is never materialised)."""
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        Cout = self.shapes[name + ".block1.proj.weight"][0]
        H, W = x1.H, x1.W
        rows = self.B * self.T * H * W
        film_ptr, dfilm_ptr = film if film else (0, 0)
        halo1 = self.halo_ok(x1.C, x2.C if x2 is not None else 0, Cout, H, W)
        w1, gw1 = self.pack_conv(name + ".block1.proj.weight", frag=halo1)
        halo2 = self.halo_ok(Cout, 0, Cout, H, W)
        # mirror_in (mirrored plans, the first block): x1's two batch halves are identical, so block1's convolution and its GroupNorm statistics
        # are too -- time and conditioning enter with the FiLM rows of the coefficients.  The convolution runs on one half; block2's loader
        # reads that half for both (a_img_mod) under each sample's own coefficients.  Needs the 2-D-tiled unsplit 3 x 3 kernel on both.
        half = self.B // 2
        mirror_in = bool(mirror_in and halo1 and halo2 and x2 is None and half * self.T * H * W >= 128 * 256 and _enabled("mirror_conv"))
        if mirror_in:
            # the kernel's own envelope for shared source frames (unsplit 2-D tiles), asked of the library instead of restated here: a
            # throw-away descriptor of block2's launch (any non-NULL pointers; nothing is launched)
            probe = N.ConvDesc()
            probe.a1, probe.C1, probe.lda1, probe.w, probe.out, probe.ldo, probe.Cout = self.base, Cout, Cout, self.wbase, self.base, Cout, Cout
            probe.nimg, probe.Hin, probe.Win, probe.Hv, probe.Wv, probe.stride = self.B * self.T, H, W, H, W, 1
            probe.KH, probe.KW, probe.off_h, probe.off_w, probe.sgn_h, probe.sgn_w = 3, 3, -1, -1, 1, 1
            probe.Hout, probe.Wout, probe.oscale = H, W, 1
            probe.a_mode, probe.a_coef, probe.a_imgs_per_sample, probe.a_img_mod = 1, self.base, self.T, half * self.T
            probe.wrap_h, probe.wrap_w = self.wrap_h, self.wrap_w
            mirror_in = bool(self.lib.vmm_conv3x3_accepts(C.byref(probe)))
        if mirror_in:
            self.B = half
        # storage types ("bf16" mode, section 3 of DESIGN.md): each op runs on bf16-stored maps where its kernel has such an instance, else on fp32 copies
        tmps: list = []
        n1 = self.nat16("conv3x3", H) and halo1
        n2 = self.nat16("conv3x3", H) and halo2
        xa1, xa2 = self.cast(x1, n1, tmps), self.cast(x2, n1, tmps)
        h1 = self.act(Cout, H, W, bf=n1)
        d1 = self.conv(a1=xa1, a2=xa2, w=w1, bias=self.wraw(name + ".block1.proj.bias"), Cout=Cout, KH=3, KW=3, off=(-1, -1), out_ptr=h1.ptr, ldo=Cout, Hv=H,
                       Wv=W, what=name + ".block1.proj", halo=halo1, out_bf=n1)
        c1_off, c1_n, c1_ptr, st1 = self.gn_coef(h1, name + ".block1", film_ptr, 2 * Cout, conv_desc=d1 if halo1 else None, mirror=mirror_in)
        if mirror_in:
            self.B = 2 * half
        w2, gw2 = self.pack_conv(name + ".block2.proj.weight", frag=halo2)
        h1b = self.cast(h1, n2, tmps)
        h2 = self.act(Cout, H, W, bf=n2)
        d2 = self.conv(a1=h1b, w=w2, bias=self.wraw(name + ".block2.proj.bias"), Cout=Cout, KH=3, KW=3, off=(-1, -1), out_ptr=h2.ptr, ldo=Cout, Hv=H, Wv=W,
                       a_coef=c1_ptr, what=name + ".block2.proj", halo=halo2, out_bf=n2)
        if mirror_in:
            d2.a_img_mod = half * self.T
        # (h1 and its coefficients are conv2's INPUT: they are released only after the GroupNorm partial sums of conv2's output have
        # their buffer -- conv2 writes those while it is still reading h1, so the two must never share memory)
        c2_off, c2_n, c2_ptr, st2 = self.gn_coef(h2, name + ".block2", 0, 0, conv_desc=d2 if halo2 else None)
        if h1b is not h1:
            tmps.remove(h1b)
            self.free_act(h1b)
        self.free_act(h1)
        self.free(c1_off, c1_n)
        has_res = (name + ".res_conv.weight") in self.shapes
        out = self.act(Cout, H, W) if self.training else h2  # training keeps the pre-norm h2 for the backward pass
        dr, gwr = None, 0
        kin = x1.C + (x2.C if x2 is not None else 0)
        if (has_res and tail is None and not self.training and self.x3 and _enabled("res_tail") and self.proj_ok(kin, Cout)):
            # out = silu(GN(h2)) + res_conv(x) in ONE launch: the projection kernel takes h2 as its residual and normalises it on the way in
            # (in place: out = h2); no res_conv output buffer, no separate output pass
            narrow = self.narrow_ok(kin, Cout) and _enabled("narrow_tail")
            nt = self.nat16("narrow" if narrow else "proj", H)
            xt1, xt2, h2t = self.cast(x1, nt, tmps), self.cast(x2, nt, tmps), self.cast(h2, nt, tmps)
            wr, _ = self.pack_linear(name + ".res_conv.weight", frag=2)
            dr = self.conv_desc(a1=xt1, a2=xt2, w=wr, bias=self.wraw(name + ".res_conv.bias"), Cout=Cout, out_ptr=h2t.ptr, ldo=Cout, Hv=H, Wv=W,
                                res_ptr=h2t.ptr, ldres=Cout, out_bf=nt)
            tail_fn = (self.lib.vmm_proj_narrow_bf16x3_res_silu if narrow
                       else self.lib.vmm_proj_bf16_res_silu if (self.one and not self.half) else self.lib.vmm_proj_bf16x3_res_silu)
            self.step(tail_fn, (C.byref(dr), c2_ptr, self.T * H * W), name + ".res_conv + out",
                      flops=2.0 * rows * kin * Cout, nbytes=4.0 * rows * (kin + 2 * Cout))
            if h2t is not h2:  # (the converted copy IS the block's output)
                tmps.remove(h2t)
                self.free_act(h2)
                h2 = h2t
            self.free_temps(tmps)
            self.free(c2_off, c2_n)
            self.plan.named[name] = h2
            return h2
        if has_res:
            pjr = self.proj_ok(kin, Cout)
            npj = self.nat16("proj", H) and pjr and not self.narrow_ok(kin, Cout)
            xr1, xr2 = self.cast(x1, npj, tmps), self.cast(x2, npj, tmps)
            wr, gwr = self.pack_linear(name + ".res_conv.weight", frag=2 if pjr else False)
            r = self.act(Cout, H, W, bf=npj)
            dr = self.conv(a1=xr1, a2=xr2, w=wr, bias=self.wraw(name + ".res_conv.bias"), Cout=Cout, out_ptr=r.ptr, ldo=Cout, Hv=H, Wv=W, what=name + ".res_conv",
                           proj=pjr, out_bf=npj)
            res_act = r
        else:
            assert x2 is None and x1.C == Cout
            r, res_act = None, x1
        if tail is not None:
            assert not self.training
            nf = self.nat16("final", H)
            h2f, rf = self.cast(h2, nf, tmps), self.cast(res_act, nf, tmps)
            tail(h2f, c2_ptr, rf.ptr, rf.ld, nf)
            if r is not None:
                self.free_act(r)
            self.free_temps(tmps)
            self.free(c2_off, c2_n)
            self.free_act(h2)
            return None
        na = self.nat16("affine", H)
        h2a, ra = self.cast(h2, na, tmps), self.cast(res_act, na, tmps)
        if self.training:
            outa = out
        elif h2a is h2:
            outa = h2
        else:  # the converted copy of h2 becomes the block's output (written in place)
            tmps.remove(h2a)
            self.free_act(h2)
            out = outa = h2 = h2a
        self.step(self.lib.vmm_affine_silu_a16 if na else self.lib.vmm_affine_silu,
                  (h2a.ptr, Cout, c2_ptr, ra.ptr, ra.ld, outa.ptr, Cout, rows, self.T * H * W, Cout), name + " out", nbytes=(6.0 if na else 12.0) * h2.n)
        if r is not None:
            self.free_act(r)
        self.free_temps(tmps)
        self.free(c2_off, c2_n)
        self.plan.named[name] = out

        def bwd():
            gout, _ = self.grad_of(out)  # complete by now
            srcs = [(x1, 0)] + ([(x2, x1.C)] if x2 is not None else [])
            if has_res:  # residual branch through the 1x1 conv
                self.wgrad(dr, gout.ptr, Cout, gwr, name + ".res_conv", gb_ptr=self.pg(name + ".res_conv.bias"))
                for xs, c0 in srcs:
                    gx, acc = self.grad_of(xs)
                    self.dgrad_1x1(name + ".res_conv.weight", c0, xs.C, name + ".res_conv dgrad", a1=gout, out_ptr=gx.ptr, ldo=xs.C, Hv=H, Wv=W,
                                   res_ptr=gx.ptr if acc else 0, ldres=xs.C)
            else:
                self.add_into(x1, gout.ptr, gout)
            # main branch: GN2+SiLU, conv2, GN1+FiLM+SiLU, conv1
            dh2 = self.act(Cout, H, W)
            self.gn_bwd(name + ".block2", gout.ptr, h2, c2_ptr, st2, 0, 0, dh2.ptr, 0)
            if (self.x3 and getattr(self.m, "use_x3_wgrad", True) and _enabled("wgrad_fused_operand")
                    and int(self.lib.vmm_conv3x3_wgrad_bf16x3_workspace(C.byref(d2), Cout))):
                # the nine-tap split-bf16 kernel stages every x element once per 64 output channels: block2's operand silu(GN1(h1)) is
                # formed in its loader (a_mode 1 of the forward descriptor), never materialised
                self.wgrad(d2, dh2.ptr, Cout, gw2, name + ".block2.proj", gb_ptr=self.pg(name + ".block2.proj.bias"))
            else:
                # materialised once for the generic weight-gradient kernels: fused into their loaders it would be recomputed by every tap /
                # channel tile of the weight (measured 61-68 vs 84-88 TFLOP/s for the same shapes)
                a1m = self.act(Cout, H, W)
                self.step(self.lib.vmm_affine_silu, (h1.ptr, Cout, c1_ptr, None, 0, a1m.ptr, Cout, rows, self.T * H * W, Cout), name + " block2 operand",
                          nbytes=8.0 * h1.n)
                d2m = N.ConvDesc.from_buffer_copy(d2)
                d2m.a1, d2m.a_mode, d2m.a_coef = a1m.ptr, 0, None
                self.plan.keepalive.append(d2m)
                self.wgrad(d2m, dh2.ptr, Cout, gw2, name + ".block2.proj", gb_ptr=self.pg(name + ".block2.proj.bias"))
                self.tmp_free(a1m)
            da1 = self.act(Cout, H, W)
            self.dgrad_3x3(name + ".block2.proj.weight", 0, Cout, name + ".block2.proj dgrad", a1=dh2, out_ptr=da1.ptr, ldo=Cout, Hv=H, Wv=W)
            self.tmp_free(dh2)
            self.gn_bwd(name + ".block1", da1.ptr, h1, c1_ptr, st1, film_ptr, 2 * Cout, da1.ptr, dfilm_ptr)  # in place: dh1 overwrites da1
            self.wgrad(d1, da1.ptr, Cout, gw1, name + ".block1.proj", gb_ptr=self.pg(name + ".block1.proj.bias"))
            for xs, c0 in srcs:
                gx, acc = self.grad_of(xs)
                self.dgrad_3x3(name + ".block1.proj.weight", c0, xs.C, name + ".block1.proj dgrad", a1=da1, out_ptr=gx.ptr, ldo=xs.C, Hv=H, Wv=W,
                               res_ptr=gx.ptr if acc else 0, ldres=xs.C)
            self.tmp_free(da1)
        self.on_backward(bwd, pg_start, uj_start)
        return out

    def layernorm(self, x: Act, gamma_name: str) -> Act:
        y = self.act(x.C, x.H, x.W)
        rows = self.B * self.T * x.H * x.W
        self.step(self.lib.vmm_channel_layernorm, (x.ptr, x.ld, self.wraw(gamma_name), y.ptr, y.ld, rows, x.C, C.c_float(1e-5)), gamma_name, nbytes=8.0 * x.n)
        return y

    def layernorm_bwd(self, x: Act, gamma_name: str, dy_ptr: int) -> None:
        gx, acc = self.grad_of(x)
        rows = self.B * self.T * x.H * x.W
        gg = self.pg(gamma_name) or self.scratch(x.C)
        sc = self.alloc(2048 * x.C)  # VMM_LN_BWD_MAX_BLOCKS partial rows of dgamma
        self.step(self.lib.vmm_channel_layernorm_bwd, (x.ptr, x.ld, self.wraw(gamma_name), dy_ptr, x.C, gx.ptr, x.C, acc, gg, rows, x.C, C.c_float(1e-5),
                                                       self.ptr(sc)), gamma_name + " bwd", nbytes=16.0 * x.n)
        self.tmp_free((sc, 2048 * x.C))

    def token_kv_bwd(self, site: str) -> None:
        """d(ek), d(ev) of one attention site -> dense backward jobs (collected, launched with the embedding backward)."""
        pfx, eo, vo, geo, gvo, rotated, hid = self.ekv_info[site]
        if rotated:
            self.step(self.lib.vmm_rotary_rows, (geo, self.rot_t_ptr, self.B, self.ntok, self.heads, hid // self.heads), site + " un-rotate d(ek)")
        D, rows = self.m.cond_dim, self.B * self.ntok
        for wname, gy in ((pfx + ".to_k.weight", geo), (pfx + ".to_v.weight", gvo)):
            self.bwd_lvl3.append(dict(x=self.tokens_ptr, w=self.wraw(wname), dy=gy, dx=self.dtokens_ptr, dw=self.pg(wname), rows=rows, K=D, N=hid))

    def linear_attn_block(self, name: str, x: Act, site: Optional[str]) -> Act:
        """Residual(PreNorm(SpatialLinearAttention)) (vddp.py:313-378, 679)."""
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        B, T, heads = self.B, self.T, self.heads
        hid = 32 * heads
        HW = x.H * x.W
        rows = B * T * HW
        if site and self.m.cond_attention == "cross-attention":
            return self.cross_attn_block(name, x, site, name + ".fn.fn", linear=True)
        ntok_s = self.ntok if site else 0
        fused_fwd = bool(self.x3 and (x.C == 64 or (x.C == 128 and _enabled("la_c128"))) and heads == 8 and HW % 32 == 0
                         and getattr(self.m, "use_fused_linattn", True))
        # training: the fused block keeps its small workspace; the backward re-forms q, k, v on chip (linattn_block_bwd.hip)
        bwd_ws_n = int(self.lib.vmm_linattn_block_bwd_workspace(B, T, HW, x.C, heads, ntok_s)) if (
            fused_fwd and self.training and getattr(self.m, "use_x3_wgrad", True) and _enabled("fused_attn_train")) else 0
        if fused_fwd and (not self.training or bwd_ws_n):
            # the two upper levels (C = 64, 128): q, k, v stay on chip (x read twice, out written once; linattn_block.hip)
            p = name + ".fn.fn"
            wq, gwq = self.pack_linear(p + ".to_qkv.weight", frag=2, half=True)
            wo, gwo = self.pack_linear(p + ".to_out.weight", frag=3, half=True)
            ws_n = int(self.lib.vmm_linattn_block_workspace(B, T, HW))
            ws = self.alloc(ws_n)
            ek, ev = (self.ekv_info[site][1], self.ekv_info[site][2]) if site else (0, 0)
            tmps: list = []
            nla = self.nat16("la", x.H)
            xc = self.cast(x, nla, tmps)
            out = self.act(x.C, x.H, x.W, bf=nla)
            flops = 2.0 * rows * x.C * 3 * hid + 2.0 * rows * hid * x.C + 4.0 * rows * hid * 32
            self.step(self.lib.vmm_linattn_block_bf16_a16 if nla else self.sp("vmm_linattn_block_"),
                      (xc.ptr, xc.ld, self.wraw(name + ".fn.norm.gamma"), wq, wo, self.wraw(p + ".to_out.bias"), ek or None, ev or None,
                       ntok_s, self.ptr(ws), out.ptr, out.ld, B, T, HW, x.C, heads, C.c_float(1e-5)),
                      name + " fused block", flops=flops, nbytes=(6.0 if nla else 12.0) * x.n)
            self.free_temps(tmps)
            self.free(ws, ws_n)  # (a no-op in training plans: the backward reads the key-softmax partials and context fragments it holds)
            self.plan.named[name] = out
            if self.training:
                wo_t = self.pack_linear_slice(p + ".to_out.weight", 0, hid, frag=2, gemm=True, half=True)
                gamma_ptr = self.wraw(name + ".fn.norm.gamma")
                dq = self.conv_desc(a1=x, w=wq, Cout=3 * hid, out_ptr=out.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W)  # (descriptor of to_qkv for its backward; never launched)

                def bwd_fused():
                    gout, _ = self.grad_of(out)
                    self.add_into(x, gout.ptr, gout)  # residual
                    gqkv = self.act(3 * hid, x.H, x.W, bf=self.one and DQKV16)  # (single-pass builds: the to_qkv backward's 16-bit operand type, csrc/vmm_common.h VMM_DQKV16)
                    stats, bws = self.alloc(2 * rows), self.alloc(bwd_ws_n)
                    geo, gvo = (self.ekv_info[site][3], self.ekv_info[site][4]) if site else (0, 0)
                    d = N.AttnBlockBwd()
                    d.x, d.ldx, d.gamma, d.wqkv_frag, d.wout_t_frag = x.ptr, x.ld, gamma_ptr, wq, wo_t
                    d.ek, d.ev, d.ntok = ek or None, ev or None, ntok_s
                    d.fwd_workspace = self.ptr(ws)
                    d.dout, d.lddo, d.dqkv, d.lddqkv, d.ln_stats = gout.ptr, x.C, gqkv.ptr, 3 * hid, self.ptr(stats)
                    d.dwout_packed, d.dbout = gwo or self.scratch(hid * x.C), self.pg(p + ".to_out.bias") or None
                    d.dek, d.dev = geo or None, gvo or None
                    d.workspace = self.ptr(bws)
                    d.B, d.T, d.HW, d.C, d.heads, d.q_scale, d.eps = B, T, HW, x.C, heads, 32 ** -0.5, 1e-5
                    self.plan.keepalive.append(d)
                    self.step(self.sp("vmm_linattn_block_bwd_"), (C.byref(d),), name + " fused block bwd (recomputation)", flops=2.0 * flops,
                              nbytes=rows * (16.0 * x.C + (2.0 if gqkv.bf else 4.0) * 3 * hid))
                    self.tmp_free((bws, bwd_ws_n))
                    dq._ln = (self.ptr(stats), gamma_ptr)
                    gy = self.qkv_backward(dq, p + ".to_qkv.weight", x, gqkv, gwq, name + " to_qkv", ln_gamma_name=name + ".fn.norm.gamma")
                    self.tmp_free(gqkv)
                    self.tmp_free((stats, 2 * rows))
                    if gy is not None:  # (else the LayerNorm backward ran as the epilogue of the to_qkv backward)
                        self.layernorm_bwd(x, name + ".fn.norm.gamma", gy.ptr)
                        self.tmp_free(gy)
                    if site:
                        self.token_kv_bwd(site)
                self.on_backward(bwd_fused, pg_start, uj_start)
            return out
        x_orig, tmps_x = x, []
        x = self.cast(x, False, tmps_x)  # (the unfused chain runs on fp32 maps; a bf16-stored input -- only when the fused block does not apply -- is converted)
        pj = self.proj_ok(x.C, 3 * hid)  # A-stationary projection kernel ...
        ln_tr = self.ln_fused_training_ok(x.C, 3 * hid)
        fuse_ln = pj and (not self.training or ln_tr)  # ... with the PreNorm LayerNorm run while the rows are staged (training: statistics kept for the wgrad)
        y = x if fuse_ln else self.layernorm(x, name + ".fn.norm.gamma")
        qkv = self.act(3 * hid, x.H, x.W)
        if not pj and self.split_k_ok(x.C, 3 * hid):
            self.proj_split_k(y, name + ".fn.fn.to_qkv.weight", 3 * hid, qkv.ptr, name + " to_qkv")
            dq = gwq = None
        else:
            wq, gwq = self.pack_linear(name + ".fn.fn.to_qkv.weight", frag=2 if pj else False)
            dq = self.conv(a1=y, w=wq, Cout=3 * hid, out_ptr=qkv.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W, what=name + " to_qkv", proj=pj,
                           ln_gamma=self.wraw(name + ".fn.norm.gamma") if fuse_ln else 0, ln_stats=self.ptr(self.alloc(2 * rows)) if (fuse_ln and ln_tr) else 0)
        if not fuse_ln:
            self.free_act(y)
        # (pass 1 on the split-bf16 matrix cores in both inference and bf16x3 training: the merge pass, which also leaves the softmax statistics the
        # backward reads, is the same kernel either way)
        la_mfma = self.x3 and _enabled("la_mfma")
        # (position slices per (frame, head); swept for the matrix-core pass -- one wave per slice -- too: 2048 / 4096 / 8192 / 16384 slices
        # gave 0.122 / 0.145 / 0.157 / 0.173 ms at the C = 128 level, the partial records and their merge grow with the slices)
        nsplit = max(1, min((HW + 63) // 64, -(-2048 // (B * T * heads))))
        part_n, ctx_n = B * T * heads * nsplit * LA_PART, B * T * heads * 1024
        part, ctx = self.alloc(part_n), self.alloc(ctx_n)
        kstat_ptr = self.ptr(self.alloc(B * T * heads * 64)) if self.training else 0
        ek, ev = (self.ekv_info[site][1], self.ekv_info[site][2]) if site else (0, 0)
        ntok = self.ntok if site else 0
        ctx_fn = self.lib.vmm_linattn_context_bf16x3 if la_mfma else self.lib.vmm_linattn_context
        self.step(ctx_fn, (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, B, T, HW, heads, 32, nsplit, self.ptr(part), self.ptr(ctx),
                                                 kstat_ptr or None), name + " context", nbytes=4.0 * rows * 2 * hid)
        o = self.act(hid, x.H, x.W)
        self.step(self.lib.vmm_linattn_apply, (qkv.ptr, 3 * hid, self.ptr(ctx), o.ptr, hid, B, T, HW, heads, 32), name + " apply", nbytes=4.0 * rows * 2 * hid)
        self.free_act(qkv)
        self.free(part, part_n)
        self.free(ctx, ctx_n)
        pjo = self.proj_ok(hid, x.C)
        wo, gwo = self.pack_linear(name + ".fn.fn.to_out.weight", frag=2 if pjo else False)
        out = self.act(x.C, x.H, x.W)
        do = self.conv(a1=o, w=wo, bias=self.wraw(name + ".fn.fn.to_out.bias"), Cout=x.C, out_ptr=out.ptr, ldo=x.C, Hv=x.H, Wv=x.W, res_ptr=x.ptr, ldres=x.ld,
                       what=name + " to_out", proj=pjo)
        self.free_act(o)
        self.free_temps(tmps_x)
        self.plan.named[name] = out

        def bwd():
            gout, _ = self.grad_of(out)
            self.add_into(x, gout.ptr, gout)  # residual
            self.wgrad(do, gout.ptr, x.C, gwo, name + " to_out", gb_ptr=self.pg(name + ".fn.fn.to_out.bias"))
            go = self.act(hid, x.H, x.W)
            self.dgrad_1x1(name + ".fn.fn.to_out.weight", 0, hid, name + " to_out dgrad", a1=gout, out_ptr=go.ptr, ldo=hid, Hv=x.H, Wv=x.W)
            gqkv = self.act(3 * hid, x.H, x.W)
            dctx = self.alloc(ctx_n)
            geo, gvo = (self.ekv_info[site][3], self.ekv_info[site][4]) if site else (0, 0)
            self.step(self.lib.vmm_linattn_bwd, (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, self.ptr(ctx), kstat_ptr, go.ptr, hid, self.ptr(dctx), gqkv.ptr,
                                                 geo or None, gvo or None, B, T, HW, heads, 32), name + " core bwd", nbytes=4.0 * rows * 7 * hid)
            self.tmp_free(go)
            self.tmp_free((dctx, ctx_n))
            gy = self.qkv_backward(dq, name + ".fn.fn.to_qkv.weight", x, gqkv, gwq, name + " to_qkv")
            self.tmp_free(gqkv)
            self.layernorm_bwd(x, name + ".fn.norm.gamma", gy.ptr)
            self.tmp_free(gy)
            if site:
                self.token_kv_bwd(site)
        self.on_backward(bwd, pg_start, uj_start)
        return out

    def softmax_attn_block(self, name: str, x: Act, site: Optional[str], *, temporal: bool) -> Act:
        """Residual(PreNorm(EinopsToAndFrom(Attention))) (vddp.py:396-535; 615/630/680 temporal, 687-689 mid spatial)."""
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        B, T, heads = self.B, self.T, self.heads
        dh = self.dh_t if temporal else 32  # (attn_dim_head reaches the temporal attentions only, vddp.py:615 against 687)
        hid = dh * heads
        HW = x.H * x.W
        rows = B * T * HW
        p = name + ".fn.fn.fn"
        focus = self.focus and temporal and name != "init_temporal_attn"  # (vddp.py:743: init_temporal_attn is called without the mask)
        if focus and site:
            raise ValueError("focus_present_mask with conditioning tokens at the temporal attentions: the reference's (frames x frames) mask does not "
                             "broadcast against the (frames x (tokens + frames)) scores (vddp.py:514-524); use_temporal_attention_cond=False or "
                             "cond_attention='none' accept one")
        if site and self.m.cond_attention == "cross-attention":
            return self.cross_attn_block(name, x, site, p, linear=False, temporal=temporal)
        ntok_s = self.ntok if site else 0
        fused_fwd = bool(temporal and dh == 32 and not focus and self.x3 and getattr(self.m, "use_fused_temporal", True) and (x.C != 128 or _enabled("tb_c128"))
                         and self.lib.vmm_temporal_block_supported(T, ntok_s, HW, x.C, heads) > 0)
        # training: the fused block stores nothing but its output; the backward re-forms q, k, v and the probabilities on chip (temporal_block_bwd.hip)
        bwd_ws_n = int(self.lib.vmm_temporal_block_bwd_workspace(B, T, HW, x.C, heads, ntok_s)) if (
            fused_fwd and self.training and getattr(self.m, "use_x3_wgrad", True) and _enabled("fused_attn_train")) else 0
        if fused_fwd and (not self.training or bwd_ws_n):
            # the two upper levels (C = 64, 128): the whole block in ONE kernel (x read once, out written once; temporal_block.hip)
            wq, gwq = self.pack_linear(p + ".to_qkv.weight", frag=2, half=True)
            wo, gwo = self.pack_linear(p + ".to_out.weight", frag=3, half=True)
            ek, ev = (self.ekv_info[site][1], self.ekv_info[site][2]) if site else (0, 0)
            tmps: list = []
            ntb = self.nat16("tb", x.H) and x.C == 64  # (the C = 128 kernel has no bf16-storage instance: fp32 copies there)
            xc = self.cast(x, ntb, tmps)
            out = self.act(x.C, x.H, x.W, bf=ntb)
            flops = 2.0 * rows * x.C * 3 * hid + 2.0 * rows * hid * x.C + 4.0 * rows * heads * 32 * (T + ntok_s)
            pfc_ = 1 if self.m.per_frame_cond else 0
            self.step(self.lib.vmm_temporal_block_bf16_a16 if ntb else self.sp("vmm_temporal_block_"),
                      (xc.ptr, xc.ld, self.wraw(name + ".fn.norm.gamma"), wq, wo, ek or None, ev or None, ntok_s, self.bias_ptr,
                       pfc_, self.rot_ptr, out.ptr, out.ld, B, T, HW, x.C, heads, C.c_float(32 ** -0.5), C.c_float(1e-5)),
                      name + " fused block", flops=flops, nbytes=(4.0 if ntb else 8.0) * x.n)
            self.free_temps(tmps)
            self.plan.named[name] = out
            if self.training:
                wo_t = self.pack_linear_slice(p + ".to_out.weight", 0, hid, frag=2, gemm=True, half=True)
                gamma_ptr = self.wraw(name + ".fn.norm.gamma")
                dq = self.conv_desc(a1=x, w=wq, Cout=3 * hid, out_ptr=out.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W)  # (descriptor of to_qkv for its backward; never launched)

                def bwd_fused():
                    gout, _ = self.grad_of(out)
                    self.add_into(x, gout.ptr, gout)  # residual
                    gqkv = self.act(3 * hid, x.H, x.W, bf=self.one and DQKV16)  # (single-pass builds: the to_qkv backward's 16-bit operand type, csrc/vmm_common.h VMM_DQKV16)
                    stats, ws = self.alloc(2 * rows), self.alloc(bwd_ws_n)
                    geo, gvo = (self.ekv_info[site][3], self.ekv_info[site][4]) if site else (0, 0)
                    d = N.AttnBlockBwd()
                    d.x, d.ldx, d.gamma, d.wqkv_frag, d.wout_t_frag = x.ptr, x.ld, gamma_ptr, wq, wo_t
                    d.ek, d.ev, d.ntok = ek or None, ev or None, ntok_s
                    d.bias, d.bias_on_cond, d.rot_tab = self.bias_ptr, pfc_, self.rot_ptr
                    d.dout, d.lddo, d.dqkv, d.lddqkv, d.ln_stats = gout.ptr, x.C, gqkv.ptr, 3 * hid, self.ptr(stats)
                    d.dwout_packed, d.dbias, d.dek, d.dev = gwo or self.scratch(hid * x.C), self.dbias_ptr, geo or None, gvo or None
                    d.workspace = self.ptr(ws)
                    d.B, d.T, d.HW, d.C, d.heads, d.q_scale, d.eps = B, T, HW, x.C, heads, 32 ** -0.5, 1e-5
                    self.plan.keepalive.append(d)
                    self.step(self.sp("vmm_temporal_block_bwd_"), (C.byref(d),), name + " fused block bwd (recomputation)", flops=2.2 * flops,
                              nbytes=rows * (8.0 * x.C + (2.0 if gqkv.bf else 4.0) * 3 * hid))
                    self.tmp_free((ws, bwd_ws_n))
                    dq._ln = (self.ptr(stats), gamma_ptr)
                    gy = self.qkv_backward(dq, p + ".to_qkv.weight", x, gqkv, gwq, name + " to_qkv", ln_gamma_name=name + ".fn.norm.gamma")
                    self.tmp_free(gqkv)
                    self.tmp_free((stats, 2 * rows))
                    if gy is not None:  # (else the LayerNorm backward ran as the epilogue of the to_qkv backward)
                        self.layernorm_bwd(x, name + ".fn.norm.gamma", gy.ptr)
                        self.tmp_free(gy)
                    if site:
                        self.token_kv_bwd(site)
                self.on_backward(bwd_fused, pg_start, uj_start)
            return out
        # A-stationary projection kernel ... (its q-scale / rotary epilogue works on whole 32-column tiles: the first tile of a head rotates when the head
        # is a multiple of 32 wide, the head's pairs repeat inside a tile when it divides 32; other widths take the generic implicit GEMM's per-column epilogue)
        pj = self.proj_ok(x.C, 3 * hid) and hid % 32 == 0 and (not temporal or dh % 32 == 0 or 32 % dh == 0)
        ln_tr = self.ln_fused_training_ok(x.C, 3 * hid)
        fuse_ln = pj and (not self.training or ln_tr)  # ... with the PreNorm LayerNorm run while the rows are staged (training: statistics kept for the wgrad)
        ntok = self.ntok if site else 0
        # bf16-stored maps through the unfused chain (the C = 128 level of the 22-frame configuration): to_qkv and to_out on the A-stationary projection
        # kernel, the core on vmm_temporal_attention, all three with bf16 instances; anything else gets an fp32 copy of x and makes an fp32 output
        n16 = bool(dh == 32 and self.nat16("proj", x.H) and self.nat16("tattn", x.H) and temporal and not focus and pj and fuse_ln and self.proj_ok(hid, x.C)
                   and not self.narrow_ok(hid, x.C) and not (heads == 8 and x.C % 128 == 0 and T <= 16 and HW % 2 == 0 and ntok <= 16))
        tmps_x: list = []
        x = self.cast(x, n16, tmps_x)
        y = x if fuse_ln else self.layernorm(x, name + ".fn.norm.gamma")
        qkv = self.act(3 * hid, x.H, x.W, bf=n16)
        q_scale = dh ** -0.5
        if not pj and self.split_k_ok(x.C, 3 * hid) and hid % 32 == 0 and (not temporal or dh % 32 == 0 or 32 % dh == 0):
            self.proj_split_k(y, p + ".to_qkv.weight", 3 * hid, qkv.ptr, name + " to_qkv", rot_tab=self.rot_ptr if temporal else 0,
                              rot_ncols=2 * hid if temporal else 0, q_scale=q_scale, q_ncols=hid)
            dq = gwq = None
        else:
            wq, gwq = self.pack_linear(p + ".to_qkv.weight", frag=2 if pj else False)
            dq = self.conv(a1=y, w=wq, Cout=3 * hid, out_ptr=qkv.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W, rot_tab=self.rot_ptr if temporal else 0,
                           rot_ncols=2 * hid if temporal else 0, q_scale=q_scale, q_ncols=hid, what=name + " to_qkv", proj=pj,
                           ln_gamma=self.wraw(name + ".fn.norm.gamma") if fuse_ln else 0, ln_stats=self.ptr(self.alloc(2 * rows)) if (fuse_ln and ln_tr) else 0,
                           out_bf=n16)
        if not fuse_ln:
            self.free_act(y)
        ek, ev = (self.ekv_info[site][1], self.ekv_info[site][2]) if site else (0, 0)
        pfc = 1 if self.m.per_frame_cond else 0
        if (temporal and dh == 32 and not focus and self.x3 and not self.training and heads == 8 and x.C % 128 == 0 and T <= 16 and HW % 2 == 0 and ntok <= 16
                and getattr(self.m, "use_fused_temporal", True)):
            # scores, value mix and to_out on the matrix cores in one kernel: qkv read once, the attention output never stored
            wo, _ = self.pack_linear(p + ".to_out.weight", frag=3)
            out = self.act(x.C, x.H, x.W)
            self.step(self.lib.vmm_temporal_core_bf16x3,
                      (qkv.ptr, 3 * hid, x.ptr, x.ld, wo, ek or None, ev or None, ntok, self.bias_ptr, pfc, out.ptr, out.ld, B, T, HW, x.C, heads),
                      name + " core+to_out", flops=2.0 * rows * hid * x.C + 4.0 * rows * hid * (T + ntok), nbytes=4.0 * rows * (3 * hid + 2 * x.C))
            self.free_act(qkv)
            self.free_temps(tmps_x)
            self.plan.named[name] = out
            return out
        o = self.act(hid, x.H, x.W, bf=n16)
        lse_ptr = self.ptr(self.alloc(rows * heads)) if self.training else 0
        if temporal:
            self.step(self.lib.vmm_temporal_attention_a16 if n16 else self.lib.vmm_temporal_attention,
                      (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, self.bias_ptr, pfc, o.ptr, hid, B, T, HW, heads, dh,
                       lse_ptr or None), name + " core", nbytes=(2.0 if n16 else 4.0) * rows * 4 * hid)
            if focus:  # masked samples: softmax over their own frame alone = 1, the output is the value row
                self.step(self.lib.vmm_focus_rows, (0, qkv.ptr + 8 * hid, 3 * hid, o.ptr, hid, self.focus_ptr, B, T * HW, hid), name + " focus (o = v)",
                          nbytes=8.0 * rows * hid)
        elif self.x3 and not self.training and ntok <= 32 and (not (ek and pfc) or ntok >= T) and _enabled("sa_mfma"):
            # flash-attention forward on the split-bf16 matrix cores (temporal_core.hip)
            self.step(self.lib.vmm_spatial_attention_bf16x3, (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, pfc, o.ptr, hid, B, T, HW, heads, 32),
                      name + " core", flops=4.0 * rows * heads * 32 * (HW + ntok), nbytes=4.0 * rows * 4 * hid)
        else:
            self.step(self.lib.vmm_spatial_attention, (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, pfc, o.ptr, hid, B, T, HW, heads, 32, lse_ptr or None),
                      name + " core", nbytes=4.0 * rows * 4 * hid)
        self.free_act(qkv)
        pjo = self.proj_ok(hid, x.C)
        wo, gwo = self.pack_linear(p + ".to_out.weight", frag=2 if pjo else False)
        out = self.act(x.C, x.H, x.W, bf=n16)
        do = self.conv(a1=o, w=wo, Cout=x.C, out_ptr=out.ptr, ldo=x.C, Hv=x.H, Wv=x.W, res_ptr=x.ptr, ldres=x.ld, what=name + " to_out", proj=pjo, out_bf=n16)
        self.free_act(o)
        self.free_temps(tmps_x)
        self.plan.named[name] = out

        def bwd():
            gout, _ = self.grad_of(out)
            self.add_into(x, gout.ptr, gout)
            self.wgrad(do, gout.ptr, x.C, gwo, name + " to_out")
            go = self.act(hid, x.H, x.W)
            self.dgrad_1x1(p + ".to_out.weight", 0, hid, name + " to_out dgrad", a1=gout, out_ptr=go.ptr, ldo=hid, Hv=x.H, Wv=x.W)
            gqkv = self.act(3 * hid, x.H, x.W)
            ndbuf = int(self.lib.vmm_attention_bwd_scratch(0 if temporal else 1, B, T, HW, heads, ntok))
            dbuf = self.alloc(ndbuf)
            geo, gvo = (self.ekv_info[site][3], self.ekv_info[site][4]) if site else (0, 0)
            go_core = go
            if focus:  # the core's backward sees dO = 0 for the masked samples (nothing flows through their p); their dv = dO is added afterwards
                go_core = self.act(hid, x.H, x.W)
                self.step(self.lib.vmm_focus_rows, (1, go.ptr, hid, go_core.ptr, hid, self.focus_ptr, B, T * HW, hid), name + " focus (dO of the core)",
                          nbytes=8.0 * rows * hid)
            self.step(self.lib.vmm_attention_bwd,
                      (0 if temporal else 1, qkv.ptr, 3 * hid, ek or None, ev or None, ntok, 0 if temporal else pfc, self.bias_ptr if temporal else None,
                       pfc if temporal else 0, o.ptr, go_core.ptr, hid, lse_ptr, self.rot_ptr if temporal else None, C.c_float(q_scale), gqkv.ptr, geo or None,
                       gvo or None, self.dbias_ptr if temporal else None, self.ptr(dbuf), B, T, HW, heads, dh), name + " core bwd", nbytes=4.0 * rows * 9 * hid)
            if focus:
                self.step(self.lib.vmm_focus_rows, (2, go.ptr, hid, gqkv.ptr + 8 * hid, 3 * hid, self.focus_ptr, B, T * HW, hid), name + " focus (dv += dO)",
                          nbytes=12.0 * rows * hid)
                self.tmp_free(go_core)
            self.tmp_free(go)
            self.tmp_free((dbuf, ndbuf))
            gy = self.qkv_backward(dq, p + ".to_qkv.weight", x, gqkv, gwq, name + " to_qkv")
            self.tmp_free(gqkv)
            self.layernorm_bwd(x, name + ".fn.norm.gamma", gy.ptr)
            self.tmp_free(gy)
            if site:
                self.token_kv_bwd(site)
        self.on_backward(bwd, pg_start, uj_start)
        return out

    def cross_attn_block(self, name: str, x: Act, site: str, p: str, *, linear: bool, temporal: bool = False) -> Act:
        """Residual(PreNorm(attention)) with cond_attention = 'cross-attention' (vddp.py:354-363 linear, 476-485 softmax): q = to_q(LayerNorm(x)),
        keys / values = the conditioning tokens alone (to_k / to_v rows from the batched embedding launch), then to_out + residual.  Training: the
        backward of the two cores (vmm_cross_attention_bwd, vmm_linattn_cross_bwd) between the usual projection / LayerNorm / token backwards."""
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        B, T, heads = self.B, self.T, self.heads
        dh = self.dh_t if (temporal and not linear) else 32
        hid = dh * heads
        HW = x.H * x.W
        rows = B * T * HW
        ek, ev = self.ekv_info[site][1], self.ekv_info[site][2]
        ntok = self.ntok
        tmps_x: list = []
        x = self.cast(x, False, tmps_x)  # (no bf16-storage instances of the cross-attention kernels)
        if temporal and ntok != T:
            raise ValueError(f"cross-attention at the temporal sites adds the ({T} x {T}) positional bias to the ({T} x {ntok}) scores: "
                             "cond_attention_tokens must equal the number of frames (vddp.py:513)")
        if ntok > 64:  # (a row's scores live in registers: csrc/cross_attention.hip TOK_MAX; the 51-point stress-strain signal as GRU tokens fits)
            raise NotImplementedError("cross-attention with more than 64 conditioning tokens")
        pj = self.proj_ok(x.C, hid) and hid % 32 == 0 and (not temporal or dh % 32 == 0 or 32 % dh == 0)
        ln_tr = self.ln_fused_training_ok(x.C, hid)
        fuse_ln = pj and (not self.training or ln_tr)  # (training: the LayerNorm statistics stay for the weight gradient, as at the to_qkv sites)
        y = x if fuse_ln else self.layernorm(x, name + ".fn.norm.gamma")
        wq, gwq = self.pack_linear(p + ".to_q.weight", frag=2 if pj else False)
        q = self.act(hid, x.H, x.W)
        q_scale = dh ** -0.5
        epi = {} if linear else dict(q_scale=q_scale, q_ncols=hid, rot_tab=self.rot_ptr if temporal else 0, rot_ncols=hid if temporal else 0)
        dq = self.conv(a1=y, w=wq, Cout=hid, out_ptr=q.ptr, ldo=hid, Hv=x.H, Wv=x.W, what=name + " to_q", proj=pj,
                       ln_gamma=self.wraw(name + ".fn.norm.gamma") if fuse_ln else 0, ln_stats=self.ptr(self.alloc(2 * rows)) if (fuse_ln and ln_tr) else 0,
                       **epi)
        if not fuse_ln:
            self.free_act(y)
        o = self.act(hid, x.H, x.W)
        ctx_n = B * T * heads * 1024
        ctx = kstat_ptr = 0
        if linear:
            ctx = self.alloc(ctx_n)
            kstat_ptr = self.ptr(self.alloc(B * T * heads * 64)) if self.training else 0
            self.step(self.lib.vmm_linattn_cross_context, (ek, ev, ntok, B, T, HW, heads, 32, self.ptr(ctx), kstat_ptr or None), name + " context (tokens)")
            self.step(self.lib.vmm_linattn_apply, (q.ptr, hid, self.ptr(ctx), o.ptr, hid, B, T, HW, heads, 32), name + " apply", nbytes=4.0 * rows * 2 * hid)
            self.free(ctx, ctx_n)
        else:
            self.step(self.lib.vmm_cross_attention, (q.ptr, hid, ek, ev, ntok, self.bias_ptr if temporal else None, o.ptr, hid, B, T, HW, heads, dh),
                      name + " core (tokens)", nbytes=4.0 * rows * 2 * hid)
        self.free_act(q)
        pjo = self.proj_ok(hid, x.C)
        wo, gwo = self.pack_linear(p + ".to_out.weight", frag=2 if pjo else False)
        out = self.act(x.C, x.H, x.W)
        do = self.conv(a1=o, w=wo, bias=self.wraw(p + ".to_out.bias") if linear else 0, Cout=x.C, out_ptr=out.ptr, ldo=x.C, Hv=x.H, Wv=x.W, res_ptr=x.ptr,
                       ldres=x.ld, what=name + " to_out", proj=pjo)
        self.free_act(o)
        self.free_temps(tmps_x)
        self.plan.named[name] = out

        def bwd():
            gout, _ = self.grad_of(out)
            self.add_into(x, gout.ptr, gout)  # residual
            self.wgrad(do, gout.ptr, x.C, gwo, name + " to_out", gb_ptr=self.pg(p + ".to_out.bias") if linear else 0)
            go = self.act(hid, x.H, x.W)
            self.dgrad_1x1(p + ".to_out.weight", 0, hid, name + " to_out dgrad", a1=gout, out_ptr=go.ptr, ldo=hid, Hv=x.H, Wv=x.W)
            gq = self.act(hid, x.H, x.W)
            geo, gvo = self.ekv_info[site][3], self.ekv_info[site][4]
            if linear:
                dctx = self.alloc(ctx_n)
                self.step(self.lib.vmm_linattn_cross_bwd, (q.ptr, hid, ek, ev, ntok, self.ptr(ctx), kstat_ptr, go.ptr, hid, self.ptr(dctx), gq.ptr, hid, geo, gvo,
                                                           B, T, HW, heads, 32), name + " core bwd (tokens)", nbytes=4.0 * rows * 4 * hid)
                self.tmp_free((dctx, ctx_n))
            else:
                nsc = int(self.lib.vmm_cross_attention_bwd_scratch(B, T, HW, heads, dh, ntok))  # (0 inside the one-pass kernel's envelope: 8 heads of 32, <= 16 tokens)
                sc = self.alloc(nsc) if nsc else 0
                self.step(self.lib.vmm_cross_attention_bwd,
                          (q.ptr, hid, ek, ev, ntok, self.bias_ptr if temporal else None, go.ptr, hid, self.rot_ptr if temporal else None, C.c_float(q_scale),
                           gq.ptr, hid, geo, gvo, self.dbias_ptr if temporal else None, self.ptr(sc) if nsc else None, B, T, HW, heads, dh),
                          name + " core bwd (tokens)", nbytes=4.0 * rows * 5 * hid)
                if nsc:
                    self.tmp_free((sc, nsc))
            self.tmp_free(go)
            gy = self.qkv_backward(dq, p + ".to_q.weight", x, gq, gwq, name + " to_q")
            self.tmp_free(gq)
            self.layernorm_bwd(x, name + ".fn.norm.gamma", gy.ptr)
            self.tmp_free(gy)
            self.token_kv_bwd(site)
        self.on_backward(bwd, pg_start, uj_start)
        return out

    # ---------------------------------------------------------------- job tables
    def _upload_table(self, arr) -> int:
        # Job tables are written once at build time, so they must never share memory with buffers that kernels write at run
        # time (the arena recycles backward temporaries): they live in the weight buffer, next to the packed operands.
        nbytes = C.sizeof(arr)
        ptr = self.wslot((nbytes + 3) // 4)
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self.job_uploads.append(((ptr - self.wbase) // 4, host))
        return ptr

    def dense_level(self, jobs: List[dict], what: str) -> None:
        if not jobs:
            return
        arr = (N.DenseJob * len(jobs))()
        max_units = 0
        for i, j in enumerate(jobs):
            a = arr[i]
            a.x, a.w, a.b, a.add, a.y = j["x"], j["w"], j.get("b") or None, j.get("add") or None, j["y"]
            a.rows, a.K, a.N = j["rows"], j["K"], j["N"]
            a.ldx, a.ldy, a.ldadd = j.get("ldx", j["K"]), j.get("ldy", j["N"]), j.get("ldadd", j["N"])
            a.act_in, a.act_out = j.get("act_in", 0), j.get("act_out", 0)
            max_units = max(max_units, j["N"])
        self.step(self.lib.vmm_dense_batched, (self._upload_table(arr), len(jobs), max_units), what)

    def dense_bwd_level(self, jobs: List[dict], what: str) -> None:
        jobs = [j for j in jobs if j.get("dw") or j.get("dx") or j.get("db")]
        if not jobs:
            return
        arr = (N.DenseBwdJob * len(jobs))()
        max_n = max_x = 0
        for i, j in enumerate(jobs):
            a = arr[i]
            a.x, a.w, a.b, a.dy = j["x"], j["w"], j.get("b") or None, j["dy"]
            a.dx, a.dw, a.db = j.get("dx") or None, j.get("dw") or None, j.get("db") or None
            a.rows, a.K, a.N = j["rows"], j["K"], j["N"]
            a.ldx, a.lddy, a.lddx = j["K"], j["N"], j["K"]
            a.act_in, a.act_out, a.accumulate = j.get("act_in", 0), j.get("act_out", 0), 1
            max_n = max(max_n, j["N"])
            if j.get("dx"):
                max_x = max(max_x, j["rows"] * ((j["K"] + 63) // 64))
        self.step(self.lib.vmm_dense_bwd_batched, (self._upload_table(arr), len(jobs), max_n, max_x), what)

    def emit_unpack(self, jobs: List[dict], what: str) -> None:
        """Scatter packed weight gradients into the flat torch-layout gradient buffer (one batched launch)."""
        if not jobs:
            return
        arr, mx = _pack_table(jobs, lambda j: self.pgbase + 4 * (self.plan.param_slices[j["name"]][0] + j.get("src_off", 0)))
        self.step(self.lib.vmm_pack_weights, (self._upload_table(arr), len(jobs), mx, 1), what)

    # ---------------------------------------------------------------- the network
    def build(self) -> Plan:
        m, B, T, H, W = self.m, self.B, self.T, self.H, self.W
        lib, heads = self.lib, self.heads
        td, cd, D = m.time_dim, m.cond_dim, m.cond_dim
        Cx = m.channels
        hid = 32 * heads
        tr = self.training
        wrap = bool(self.wrap_h or self.wrap_w)
        rows0 = B * T * H * W
        # static inputs / outputs
        x_in_off = self.alloc(B * Cx * T * H * W)
        time_off = self.alloc(B * 2)
        cond_off = self.alloc(B * self.cond_len)
        mask_off = self.alloc((B + 3) // 4)
        self.focus_off = self.alloc((B + 3) // 4) if self.focus else 0
        self.focus_ptr = self.ptr(self.focus_off) if self.focus else 0
        out_off = self.alloc(B * m.out_dim * T * H * W)
        dout_off = self.alloc(B * m.out_dim * T * H * W) if tr else 0
        self.dx_off = self.alloc(B * Cx * T * H * W) if tr else 0
        self.io = (x_in_off, time_off, cond_off, mask_off, out_off, dout_off)
        # constant tables
        rot = hostmath.rotary_table(T, self.dh_t)
        rot_t = rot.clone()
        rot_t[..., 1] = -rot_t[..., 1]  # transpose of the rotation (backward of rotated token keys)
        rot_off, rot_t_off = self.alloc(rot.numel()), self.alloc(rot.numel())
        bk = hostmath.relpos_buckets(T, 32, 32)
        bk_off = self.alloc(bk.numel())
        self.consts = [(rot_off, rot.reshape(-1)), (rot_t_off, rot_t.reshape(-1)), (bk_off, bk.reshape(-1).view(torch.float32))]
        self.rot_ptr, self.rot_t_ptr = self.ptr(rot_off), self.ptr(rot_t_off)
        bias_off = self.alloc(heads * T * T)
        self.bias_ptr = self.ptr(bias_off)
        self.dbias_ptr = self.scratch(heads * T * T) if tr else 0
        emb_name = "time_rel_pos_bias.relative_attention_bias.weight"
        self.step(lib.vmm_relpos_bias, (self.wraw(emb_name), self.ptr(bk_off), T, heads, self.bias_ptr), "time_rel_pos_bias")

        # ---- conditioning / time embedding (vddp.py:745-795)
        se = self.alloc(B * m.dim)
        self.step(lib.vmm_sinusoidal_embed, (self.ptr(time_off), B, m.dim, C.c_float(-(math.log(10000) / (m.dim // 2 - 1))), self.ptr(se)), "time sinusoid")
        concat = getattr(m, "cond_to_time", "add") == "concat"  # vddp.py:786-789: t = cat(t, hidden) instead of t + hidden; the ResnetBlock mlps take 2 td inputs
        tw = 2 * td if concat else td
        h1, t2, hidden, temb = self.alloc(B * td), self.alloc(B * td), self.alloc(B * td), self.alloc(B * tw)
        gh1, gt2, ghidden = [self.scratch(B * td) for _ in range(3)] if tr else (0, 0, 0)
        gtemb = self.scratch(B * tw) if tr else 0
        ntok = self.ntok = m.cond_attention_tokens
        tokens = self.alloc(B * ntok * D) if m.cond_attention != "none" else None
        self.tokens_ptr = self.ptr(tokens) if tokens is not None else 0
        self.dtokens_ptr = self.scratch(B * ntok * D) if (tr and tokens is not None) else 0
        lvl1 = [dict(x=self.ptr(se), w=self.wraw("time_mlp.1.weight"), b=self.wraw("time_mlp.1.bias"), y=self.ptr(h1), rows=B, K=m.dim, N=td, act_out=2)]
        lvl2 = [dict(x=self.ptr(h1), w=self.wraw("time_mlp.3.weight"), b=self.wraw("time_mlp.3.bias"), y=self.ptr(t2), rows=B, K=td, N=td)]
        bwd_lvl1 = [dict(x=self.ptr(se), w=self.wraw("time_mlp.1.weight"), b=self.wraw("time_mlp.1.bias"), dy=gh1, dw=self.pg("time_mlp.1.weight"),
                         db=self.pg("time_mlp.1.bias"), rows=B, K=m.dim, N=td, act_out=2)] if tr else []
        bwd_lvl2 = [dict(x=self.ptr(h1), w=self.wraw("time_mlp.3.weight"), dy=gt2, dx=gh1, dw=self.pg("time_mlp.3.weight"), db=self.pg("time_mlp.3.bias"),
                         rows=B, K=td, N=td)] if tr else []
        cond_bwd: List[Callable[[], None]] = []
        if m.per_frame_cond:
            if self.cond_len != ntok:
                raise ValueError(f"per_frame_cond expects cond of shape (b, {ntok})")
            pooled, pl, c1 = self.alloc(B * D), self.alloc(B * D), self.alloc(B * D)
            gpooled, gpl, gc1 = [self.scratch(B * D) for _ in range(3)] if tr else (0, 0, 0)
            self.step(lib.vmm_cond_tokens, (self.ptr(cond_off), self.wraw("sign_emb.weight"), self.wraw("sign_emb.bias"), self.wraw("null_text_token"),
                                            self.ptr(mask_off), B, ntok, D, self.ptr(tokens), self.ptr(pooled)), "sign_emb tokens")
            self.step(lib.vmm_rows_layernorm_affine, (self.ptr(pooled), self.wraw("cond_token_to_hidden.0.weight"), self.wraw("cond_token_to_hidden.0.bias"),
                                                      self.ptr(pl), B, D, C.c_float(1e-5)), "cond_token_to_hidden.0")
            lvl1.append(dict(x=self.ptr(pl), w=self.wraw("cond_token_to_hidden.1.weight"), b=self.wraw("cond_token_to_hidden.1.bias"), y=self.ptr(c1), rows=B,
                             K=D, N=D, act_out=1))
            lvl2.append(dict(x=self.ptr(c1), w=self.wraw("cond_token_to_hidden.3.weight"), b=self.wraw("cond_token_to_hidden.3.bias"), y=self.ptr(hidden),
                             rows=B, K=D, N=td))
            if tr:
                bwd_lvl2.append(dict(x=self.ptr(c1), w=self.wraw("cond_token_to_hidden.3.weight"), dy=ghidden, dx=gc1, dw=self.pg("cond_token_to_hidden.3.weight"),
                                     db=self.pg("cond_token_to_hidden.3.bias"), rows=B, K=D, N=td))
                bwd_lvl1.append(dict(x=self.ptr(pl), w=self.wraw("cond_token_to_hidden.1.weight"), b=self.wraw("cond_token_to_hidden.1.bias"), dy=gc1, dx=gpl,
                                     dw=self.pg("cond_token_to_hidden.1.weight"), db=self.pg("cond_token_to_hidden.1.bias"), rows=B, K=D, N=D, act_out=1))
                for nm_ in ("cond_token_to_hidden.0.weight", "cond_token_to_hidden.0.bias", "sign_emb.weight", "sign_emb.bias", "null_text_token"):
                    self._touch(nm_)

                def cond_b():
                    self.step(lib.vmm_rows_layernorm_affine_bwd, (self.ptr(pooled), self.wraw("cond_token_to_hidden.0.weight"), gpl, gpooled,
                                                                  self.pg("cond_token_to_hidden.0.weight"), self.pg("cond_token_to_hidden.0.bias"), B, D,
                                                                  C.c_float(1e-5)), "cond_token_to_hidden.0 bwd")
                    self.step(lib.vmm_cond_tokens_bwd, (self.ptr(cond_off), self.ptr(mask_off), self.dtokens_ptr, gpooled, B, ntok, D, self.pg("sign_emb.weight"),
                                                        self.pg("sign_emb.bias"), self.pg("null_text_token")), "sign_emb bwd")
                cond_bwd.append(cond_b)
        else:
            chain = [1, 16, 32, 64, 128, cd]
            L = self.cond_len
            cur, cur_C = self.ptr(cond_off), 1
            stages = []
            for i, co in enumerate(chain[1:]):
                Lout = (L + 2 - 4) // 2 + 1
                nxt = self.ptr(hidden) if i == 4 else self.ptr(self.alloc(B * co * Lout))
                gnx = (ghidden if i == 4 else self.scratch(B * co * Lout)) if tr else 0
                wn, bn = f"sign_emb_CNN.emb_model.{2 * i}.weight", f"sign_emb_CNN.emb_model.{2 * i}.bias"
                self.step(lib.vmm_conv1d_k4s2_silu, (cur, self.wraw(wn), self.wraw(bn), nxt, B, cur_C, co, L), f"sign_emb_CNN.{2 * i}")
                self._touch(wn)
                self._touch(bn)
                stages.append((i, cur, cur_C, co, L, gnx))
                cur, cur_C, L = nxt, co, Lout
            if L != 1:
                raise ValueError("sign_emb_CNN must reduce the conditioning signal to length 1 (cond length 32..63)")
            gru = []  # cond_att_GRU (vddp.py:546-549, 646-649, 769-770): the tokens are the states of a 3-layer GRU over the signal, one per sample of it
            if tokens is not None and getattr(m, "cond_att_GRU", False):
                Lg = self.cond_len
                if Lg != ntok:
                    raise ValueError(f"cond_att_GRU makes one token per sample of the conditioning signal: cond_attention_tokens ({ntok}) must equal its "
                                     f"length ({Lg}) (the reference's torch.where against null_text_token, vddp.py:778)")
                if D > 1024:
                    raise NotImplementedError("cond_att_GRU with a hidden width above 1024")
                xin, in_dim = self.ptr(cond_off), 1
                for l in range(3):
                    pf = "sign_emb_GRU.emb_model."
                    wih, whh, bih, bhh = (pf + f"{n}_l{l}" for n in ("weight_ih", "weight_hh", "bias_ih", "bias_hh"))
                    gi, yl = self.alloc(B * Lg * 3 * D), self.alloc(B * Lg * D)
                    hp, gates = (self.ptr(self.alloc(B * Lg * D)), self.ptr(self.alloc(B * Lg * 4 * D))) if tr else (0, 0)
                    self.dense_level([dict(x=xin, w=self.wraw(wih), b=self.wraw(bih), y=self.ptr(gi), rows=B * Lg, K=in_dim, N=3 * D)],
                                     f"sign_emb_GRU layer {l}: W_ih x + b_ih for every step")
                    whh_t = self.pack(whh, 3 * D * D, want_grad=False, TH=1, TW=1, C=D, Cp=D, N=3 * D, sn=D, sc=1)[0]  # [H][3H]
                    self.step(lib.vmm_gru_recurrent, (self.ptr(gi), whh_t, self.wraw(bhh), self.ptr(yl), hp or None, gates or None, B, Lg, D),
                              f"sign_emb_GRU layer {l}: recurrence")
                    for nm_ in (wih, whh, bih, bhh):
                        self._touch(nm_)
                    gru.append((xin, in_dim, wih, whh, bih, bhh, hp, gates))
                    xin, in_dim = self.ptr(yl), D
                self.step(lib.vmm_tokens_select, (xin, self.wraw("null_text_token"), self.ptr(mask_off), B, ntok, D, self.ptr(tokens)), "tokens (GRU states)")
            elif tokens is not None:
                self.step(lib.vmm_tokens_from_hidden, (self.ptr(hidden), self.wraw("null_text_token"), self.ptr(mask_off), B, ntok, D, self.ptr(tokens)), "tokens")
            if tr:
                def cond_b():
                    if gru:
                        Lg = self.cond_len
                        dy = self.ptr(self.alloc(B * Lg * D))
                        self.step(lib.vmm_tokens_select_bwd, (self.dtokens_ptr, self.ptr(mask_off), B, ntok, D, dy, self.pg("null_text_token")), "tokens bwd")
                        for l in (2, 1, 0):
                            xin, in_dim, wih, whh, bih, bhh, hp, gates = gru[l]
                            dgi, dgh = self.ptr(self.alloc(B * Lg * 3 * D)), self.ptr(self.alloc(B * Lg * 3 * D))
                            self.step(lib.vmm_gru_recurrent_bwd, (dy, gates, hp, self.wraw(whh), dgi, dgh, B, Lg, D), f"sign_emb_GRU layer {l}: recurrence bwd")
                            dx = self.scratch(B * Lg * in_dim) if l > 0 else 0  # (zeroed every backward: the dense backward accumulates into it)
                            self.dense_bwd_level([dict(x=xin, w=self.wraw(wih), dy=dgi, dx=dx, dw=self.pg(wih), db=self.pg(bih), rows=B * Lg, K=in_dim, N=3 * D),
                                                  dict(x=hp, w=self.wraw(whh), dy=dgh, dw=self.pg(whh), db=self.pg(bhh), rows=B * Lg, K=D, N=3 * D)],
                                                 f"sign_emb_GRU layer {l}: weight / bias / input gradients")
                            dy = dx
                    elif tokens is not None:
                        self.step(lib.vmm_tokens_from_hidden_bwd, (self.dtokens_ptr, self.ptr(mask_off), B, ntok, D, ghidden, self.pg("null_text_token")),
                                  "tokens bwd")
                    rev = list(reversed(stages))
                    for (i, xin, cin, co, Lin, gy), prev in zip(rev, rev[1:] + [None]):
                        gx = prev[5] if prev is not None else None
                        wn, bn = f"sign_emb_CNN.emb_model.{2 * i}.weight", f"sign_emb_CNN.emb_model.{2 * i}.bias"
                        self.step(lib.vmm_conv1d_k4s2_silu_bwd, (xin, self.wraw(wn), self.wraw(bn), gy, gx, self.pg(wn) or self.scratch(co * cin * 4),
                                                                 self.pg(bn) or self.scratch(co), B, cin, co, Lin), f"sign_emb_CNN.{2 * i} bwd")
                cond_bwd.append(cond_b)
        self.dense_level(lvl1, "embed level 1")
        self.dense_level(lvl2, "embed level 2")
        self.step(lib.vmm_select_concat if concat else lib.vmm_select_add,
                  (self.ptr(hidden), self.wraw("null_text_hidden"), self.ptr(mask_off), self.ptr(t2), self.ptr(temb), B, td), "t | hidden" if concat else "t + hidden")
        self._touch("null_text_hidden")

        # level 3: every ResnetBlock.mlp and every to_k/to_v on the tokens, one launch
        lvl3: List[dict] = []
        self.bwd_lvl3: List[dict] = []
        film: Dict[str, Tuple[int, int]] = {}
        res_names = [f"downs.{i}.{j}" for i in range(len(m.in_out)) for j in (0, 1)] + ["mid_block1", "mid_block2"] + \
                    [f"ups.{i}.{j}" for i in range(len(m.in_out)) for j in (0, 1)]
        for rn in res_names:
            n_out = self.shapes[rn + ".mlp.1.weight"][0]
            fo = self.alloc(B * n_out)
            gfo = self.scratch(B * n_out) if tr else 0
            film[rn] = (self.ptr(fo), gfo)
            lvl3.append(dict(x=self.ptr(temb), w=self.wraw(rn + ".mlp.1.weight"), b=self.wraw(rn + ".mlp.1.bias"), y=self.ptr(fo), rows=B, K=tw, N=n_out, act_in=1))
            if tr:
                self.bwd_lvl3.append(dict(x=self.ptr(temb), w=self.wraw(rn + ".mlp.1.weight"), dy=gfo, dx=gtemb, dw=self.pg(rn + ".mlp.1.weight"),
                                          db=self.pg(rn + ".mlp.1.bias"), rows=B, K=tw, N=n_out, act_in=1))
        self.ekv_info: Dict[str, tuple] = {}
        rot_sites: List[int] = []

        # rotated token keys of all temporal sites live in ONE block: a single rotary launch covers them (they were nine 3-microsecond launches)
        n_rot = (2 * len(m.in_out) + 1) if (tokens is not None and m.use_temporal_attention_cond and m.per_frame_cond) else 0
        hid_t = self.dh_t * heads  # (the rotated keys all belong to temporal sites)
        rot_block = self.alloc(n_rot * B * ntok * hid_t) if n_rot else 0

        def add_ekv(site: str, pfx: str, rotate: bool, hid: int = hid):
            if rotate:
                eo, vo = rot_block + len(rot_sites) * B * ntok * hid, self.alloc(B * ntok * hid)
            else:
                eo, vo = self.alloc(B * ntok * hid), self.alloc(B * ntok * hid)
            lvl3.append(dict(x=self.ptr(tokens), w=self.wraw(pfx + ".to_k.weight"), y=self.ptr(eo), rows=B * ntok, K=D, N=hid))
            lvl3.append(dict(x=self.ptr(tokens), w=self.wraw(pfx + ".to_v.weight"), y=self.ptr(vo), rows=B * ntok, K=D, N=hid))
            geo, gvo = (self.scratch(B * ntok * hid), self.scratch(B * ntok * hid)) if tr else (0, 0)
            self._touch(pfx + ".to_k.weight")
            self._touch(pfx + ".to_v.weight")
            self.ekv_info[site] = (pfx, self.ptr(eo), self.ptr(vo), geo, gvo, rotate, hid)
            if rotate:
                rot_sites.append(self.ptr(eo))

        if tokens is not None:
            for i in range(len(m.in_out)):
                for side in ("downs", "ups"):
                    if m.use_sparse_linear_attn:
                        add_ekv(f"{side}.{i}.2", f"{side}.{i}.2.fn.fn", False)
                    if m.use_temporal_attention_cond:
                        add_ekv(f"{side}.{i}.3", f"{side}.{i}.3.fn.fn.fn", m.per_frame_cond, hid_t)
            add_ekv("mid_spatial_attn", "mid_spatial_attn.fn.fn.fn", False)
            if m.use_temporal_attention_cond:
                add_ekv("mid_temporal_attn", "mid_temporal_attn.fn.fn.fn", m.per_frame_cond, hid_t)
        self.dense_level(lvl3, "embed level 3 (film + token k/v)")
        if rot_sites:
            if ntok > T:
                raise ValueError("rotating token keys needs tokens <= frames")
            assert len(rot_sites) == n_rot
            self.step(lib.vmm_rotary_rows, (self.ptr(rot_block), self.rot_ptr, B * n_rot, ntok, heads, self.dh_t), "rotate token keys (all sites)")

        def embed_bwd():
            self.dense_bwd_level(self.bwd_lvl3, "embed level 3 bwd")
            self.step(lib.vmm_select_concat_bwd if concat else lib.vmm_select_add_bwd,
                      (gtemb, self.ptr(mask_off), ghidden, self.pg("null_text_hidden") or None, gt2, B, td), "t | hidden bwd" if concat else "t + hidden bwd")
            self.dense_bwd_level(bwd_lvl2, "embed level 2 bwd")
            self.dense_bwd_level(bwd_lvl1, "embed level 1 bwd")
            for f_ in cond_bwd:
                f_()
            self.step(lib.vmm_relpos_bias_bwd, (self.dbias_ptr, self.ptr(bk_off), T, heads, self.pg(emb_name) or self.scratch(32 * heads)), "time_rel_pos_bias bwd")
        self.on_backward(embed_bwd, 0, 0)

        # ---- stem
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        if self.mirrored:  # (the builder's batch is what every emitter sizes its launches and buffers by)
            self.B = B = B // 2
            rows0 = B * T * H * W
        Cxp = (Cx + 3) // 4 * 4  # input rows padded to a multiple of four channels (zeros; the reference takes any channel count, vddp.py:576, 624)
        xin = Act(self.alloc(rows0 * Cxp), Cxp, H, W, rows0 * Cxp)
        xin.ptr = self.ptr(xin.off)
        self.step(lib.vmm_ncthw_to_rows, (self.ptr(x_in_off), B, Cx, T, H * W, xin.ptr, Cxp), "ncthw -> rows")
        k = m.init_kernel_size
        # (the stem kernel's K order holds four channels per tap: wider inputs take the generic implicit GEMM)
        stem_ok = bool(self.x3 and Cx <= 4 and m.init_dim == 64 and k % 2 == 1 and k <= 8 and rows0 * 64 < 2 ** 31 and not wrap and _enabled("stem") and (not tr or _enabled("stem_train")))
        nstem = stem_ok and self.nat16("stem", H)
        x = self.act(m.init_dim, H, W, bf=nstem)
        if stem_ok:
            # the stem on its own kernel: the tile's neighbourhood staged once in LDS, four neighbouring taps per k16 step (stem_conv.hip)
            wi = self.pack("init_conv.weight", 2048 * k, want_grad=False, TH=k, TW=k, C=Cx, Cp=Cx, N=64, sn=Cx * k * k, sc=k * k, sh=k, sw=1, fmt=7)[0]
            self.step(lib.vmm_stem_conv_bf16x3_a16 if nstem else lib.vmm_stem_conv_bf16x3,
                      (xin.ptr, wi, self.wraw("init_conv.bias"), x.ptr, m.init_dim, B * T, H, W, m.init_dim, k), "init_conv",
                      flops=2.0 * rows0 * k * k * Cx * 64, nbytes=4.0 * rows0 * (4 + 64))
            dinit = gwi = None
            if tr:  # the weight gradient works from the layer's descriptor and the plain packed layout (the operand copy itself is not used)
                _, gwi = self.pack_conv("init_conv.weight", pad_cin_to=Cxp)
                dinit = self.conv_desc(a1=xin, w=0, bias=self.wraw("init_conv.bias"), Cout=m.init_dim, KH=k, KW=k, off=(-(k // 2), -(k // 2)), out_ptr=x.ptr,
                                       ldo=m.init_dim, Hv=H, Wv=W)
        else:
            wi, gwi = self.pack_conv("init_conv.weight", pad_cin_to=Cxp)
            dinit = self.conv(a1=xin, w=wi, bias=self.wraw("init_conv.bias"), Cout=m.init_dim, KH=k, KW=k, off=(-(k // 2), -(k // 2)), out_ptr=x.ptr,
                              ldo=m.init_dim, Hv=H, Wv=W, what="init_conv")
        self.free_act(xin)
        x0 = x

        def init_bwd():
            g0, _ = self.grad_of(x0)
            self.wgrad(dinit, g0.ptr, m.init_dim, gwi, "init_conv", gb_ptr=self.pg("init_conv.bias"))
            # d loss / d x: a launch of its own list -- Plan.backward runs it only when the input gradient is asked for
            self.plan.dx_steps.append((lib.vmm_stem_conv_dgrad, (g0.ptr, m.init_dim, self.wraw("init_conv.weight"), self.ptr(self.dx_off), B, Cx, T, H, W,
                                                                 m.init_dim, k, int(self.wrap_h), int(self.wrap_w)), "init_conv dgrad (input gradient)"))
        self.on_backward(init_bwd, pg_start, uj_start)
        x_new = self.softmax_attn_block("init_temporal_attn", x, None, temporal=True)
        self.free_act(x)
        x = x_new
        if self.mirrored:
            self.B = B = 2 * B
            rows0 = B * T * H * W
            both = self.act(x.C, H, W, bf=x.bf)
            self.step(lib.vmm_copy2, (x.ptr, both.ptr, both.ptr + 4 * x.na, x.na), "shared prefix -> both guidance branches", nbytes=12.0 * x.na)
            self.free_act(x)
            x = both
        r = x  # kept until the final block (vddp.py:744; no clone needed, every op is out of place)

        def stage(side: str, i: int, x1: Act, x2: Optional[Act]) -> Act:
            y1 = self.resnet_block(f"{side}.{i}.0", x1, x2, film[f"{side}.{i}.0"], mirror_in=self.mirrored and side == "downs" and i == 0)
            if x1 is not r:
                self.free_act(x1)
            if x2 is not None:
                self.free_act(x2)
            y2 = self.resnet_block(f"{side}.{i}.1", y1, None, film[f"{side}.{i}.1"])
            self.free_act(y1)
            if m.use_sparse_linear_attn:
                s2 = f"{side}.{i}.2"
                y3 = self.linear_attn_block(s2, y2, s2 if s2 in self.ekv_info else None)
                self.free_act(y2)
            else:
                y3 = y2
            s3 = f"{side}.{i}.3"
            y4 = self.softmax_attn_block(s3, y3, s3 if s3 in self.ekv_info else None, temporal=True)
            self.free_act(y3)
            return y4

        skips: List[Act] = []
        n_lvl = len(m.in_out)
        for i in range(n_lvl):
            x = stage("downs", i, x, None)
            skips.append(x)
            if i < n_lvl - 1:
                pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
                xs = x
                nm = f"downs.{i}.4"
                co_, ci_ = self.shapes[nm + ".weight"][0], self.shapes[nm + ".weight"][1]
                s2_ok = bool(self.x3 and not wrap and _enabled("s2") and lib.vmm_conv_s2_supported(B * T, xs.H, xs.W, ci_, co_, 0))
                tmps: list = []
                in16 = s2_ok and self.nat16("s2", xs.H)
                out16 = in16 and self.nat16("s2", xs.H // 2)
                xsc = self.cast(xs, in16, tmps)
                d = self.act(x.C, x.H // 2, x.W // 2, bf=out16)
                if s2_ok:
                    # Downsample as a 3 x 3 convolution over 2 x 2 input cells (halo patch in LDS, four of the nine taps per sub-pixel)
                    wd = self.pack(nm + ".weight", 36 * ci_ * co_, want_grad=False, TH=4, TW=4, C=ci_, Cp=ci_, N=co_, sn=ci_ * 16, sc=16, sh=4, sw=1, fmt=5, half=True)[0]
                    if in16:  # bf16-stored input (and output, unless this layer leaves the bf16 levels)
                        self.step(lib.vmm_conv_s2_acc_bf16_a16, (xsc.ptr, xsc.ld, wd, self.wraw(nm + ".bias"), 0, 0, d.ptr, co_, B * T, xs.H, xs.W, ci_, co_, 0,
                                                                1 if out16 else 2), nm, flops=2.0 * B * T * d.H * d.W * 16 * ci_ * co_,
                                  nbytes=2.0 * xs.n + (2.0 if out16 else 4.0) * d.n + 4.0 * 16 * ci_ * co_)
                    else:
                        self.step(self.sp("vmm_conv_s2_acc_"),
                                  (xsc.ptr, xsc.ld, wd, self.wraw(nm + ".bias"), 0, 0, d.ptr, co_, B * T, xs.H, xs.W, ci_, co_, 0,
                                                               self.tickets() if _enabled("s2_split") else 0, N_TICKETS), nm,
                                  flops=2.0 * B * T * d.H * d.W * 16 * ci_ * co_, nbytes=4.0 * (xs.n + d.n + 16 * ci_ * co_))
                    dd = gwd = None
                    if tr:  # the weight gradient wants the layer's descriptor and a k-major gradient slot (no forward launch from them)
                        _, gwd = self.pack_conv(nm + ".weight")
                        dd = self.conv_desc(a1=xs, w=wd, Cout=xs.C, KH=4, KW=4, stride=2, off=(-1, -1), out_ptr=d.ptr, ldo=xs.C, Hv=xs.H // 2, Wv=xs.W // 2)
                else:
                    wd, gwd = self.pack_conv(nm + ".weight")
                    dd = self.conv(a1=xsc, w=wd, bias=self.wraw(nm + ".bias"), Cout=xs.C, KH=4, KW=4, stride=2, off=(-1, -1), out_ptr=d.ptr, ldo=xs.C,
                                   Hv=xs.H // 2, Wv=xs.W // 2, what=nm)
                self.free_temps(tmps)

                def down_bwd(nm=nm, xs=xs, d=d, dd=dd, gwd=gwd):
                    gd, _ = self.grad_of(d)
                    self.wgrad(dd, gd.ptr, xs.C, gwd, nm, gb_ptr=self.pg(nm + ".bias"))
                    gx, acc = self.grad_of(xs)
                    co_, ci_ = self.shapes[nm + ".weight"][0], self.shapes[nm + ".weight"][1]
                    if self.x3 and not wrap and _enabled("s2_dgrad") and lib.vmm_conv_s2_supported(B * T, d.H, d.W, co_, ci_, 1):
                        # dIn = ConvTranspose(dOut, W) with the convolution's own (Cout, Cin, 1, 4, 4) tensor read as a (C = Cout, N = Cin) transposed-convolution
                        # weight: ONE tap-subset launch (the Upsample kernel) instead of four phase GEMMs
                        wt = self.pack(nm + ".weight", 36 * co_ * ci_, want_grad=False, TH=4, TW=4, C=co_, Cp=co_, N=ci_, sn=16, sc=ci_ * 16, sh=4, sw=1, fmt=6)[0]
                        self.step(lib.vmm_conv_s2_acc_bf16x3, (gd.ptr, gd.ld, wt, 0, gx.ptr if acc else 0, ci_, gx.ptr, ci_, B * T, d.H, d.W, co_, ci_, 1, 0, 0),
                                  nm + " dgrad", flops=2.0 * B * T * d.H * d.W * 16 * ci_ * co_, nbytes=4.0 * (gd.n + (2 if acc else 1) * gx.n + 16 * ci_ * co_))
                        return
                    for ph in range(2):
                        for pw in range(2):
                            # dIn[2c+ph] = sum_kh' dOut[c + ph - kh'] W[(1-ph)+2kh']^T : [(kh', kw', co)][ci]
                            wp = self.pack(nm + ".weight", 4 * co_ * ci_, want_grad=False, gemm=self.x3, TH=2, TW=2, C=co_, Cp=co_, N=ci_, sn=16, sc=ci_ * 16, sh=4,
                                           sw=1, h0=1 - ph, hs=2, w0=1 - pw, ws=2)[0]
                            self.conv(a1=gd, w=wp, Cout=ci_, KH=2, KW=2, off=(ph, pw), sgn=(-1, -1), out_ptr=gx.ptr, ldo=ci_, Hv=d.H, Wv=d.W, Hout=xs.H,
                                      Wout=xs.W, oscale=2, oo=(ph, pw), res_ptr=gx.ptr if acc else 0, ldres=ci_, what=nm + f" dgrad phase {ph}{pw}",
                                      x3w=self.x3)
                self.on_backward(down_bwd, pg_start, uj_start)
                x = d
        # the deepest skip is also the mid input: keep it alive, do not free through `stage`
        mid_in = x
        y = self.resnet_block("mid_block1", mid_in, None, film["mid_block1"])
        y2 = self.softmax_attn_block("mid_spatial_attn", y, "mid_spatial_attn" if "mid_spatial_attn" in self.ekv_info else None, temporal=False)
        self.free_act(y)
        y3 = self.softmax_attn_block("mid_temporal_attn", y2, "mid_temporal_attn" if "mid_temporal_attn" in self.ekv_info else None, temporal=True)
        self.free_act(y2)
        x = self.resnet_block("mid_block2", y3, None, film["mid_block2"])
        self.free_act(y3)
        for i in range(n_lvl):
            x = stage("ups", i, x, skips.pop())
            if i < n_lvl - 1:
                pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
                xs = x
                nm = f"ups.{i}.4"
                ci_, co_ = self.shapes[nm + ".weight"][0], self.shapes[nm + ".weight"][1]
                phases = []
                one_launch = self.x3  # bf16x3: the four phases as ONE launch (they are small at the coarse levels)
                s2 = self.x3 and not wrap and _enabled("s2") and lib.vmm_conv_s2_supported(B * T, xs.H, xs.W, ci_, co_, 1)
                tmps_u: list = []
                out16 = bool(s2) and self.nat16("s2", xs.H * 2)
                in16 = out16 and self.nat16("s2", xs.H)
                xsc = self.cast(xs, in16, tmps_u)
                u = self.act(co_, xs.H * 2, xs.W * 2, bf=out16)
                if s2:
                    # Upsample as ONE 3 x 3 convolution over the input tile with the four output phases as 4 x Cout columns
                    wu = self.pack(nm + ".weight", 36 * ci_ * co_, want_grad=False, TH=4, TW=4, C=ci_, Cp=ci_, N=co_, sn=16, sc=co_ * 16, sh=4, sw=1, fmt=6, half=True)[0]
                    if out16:  # bf16-stored output (the input too, unless this layer enters the bf16 levels)
                        self.step(lib.vmm_conv_s2_acc_bf16_a16, (xsc.ptr, xsc.ld, wu, self.wraw(nm + ".bias"), 0, 0, u.ptr, co_, B * T, xs.H, xs.W, ci_, co_, 1,
                                                                1 if in16 else 3), nm, flops=2.0 * B * T * xs.H * xs.W * 16 * ci_ * co_,
                                  nbytes=(2.0 if in16 else 4.0) * xs.n + 2.0 * u.n + 4.0 * 16 * ci_ * co_)
                    else:
                        self.step(self.sp("vmm_conv_s2_acc_"),
                                  (xsc.ptr, xsc.ld, wu, self.wraw(nm + ".bias"), 0, 0, u.ptr, co_, B * T, xs.H, xs.W, ci_, co_, 1, 0, 0), nm,
                                  flops=2.0 * B * T * xs.H * xs.W * 16 * ci_ * co_, nbytes=4.0 * (xs.n + u.n + 16 * ci_ * co_))
                for ph in range(2 if (not s2 or tr) else 0):  # (training: the phase descriptors feed the weight gradients even when the forward is one s2 launch)
                    for pw in range(2):
                        # ConvTranspose (Cin, Cout, 1, 4, 4), output phase (ph, pw): taps kh = (1-ph) + 2*kh', dh = ph - kh'
                        wp, gwp = self.pack(nm + ".weight", 4 * ci_ * co_, TH=2, TW=2, C=ci_, Cp=ci_, N=co_, sn=16, sc=co_ * 16, sh=4, sw=1, h0=1 - ph, hs=2,
                                            w0=1 - pw, ws=2)
                        kw_ = dict(a1=xsc, w=wp, bias=self.wraw(nm + ".bias"), Cout=co_, KH=2, KW=2, off=(ph, pw), sgn=(-1, -1), out_ptr=u.ptr, ldo=co_,
                                   Hv=xs.H, Wv=xs.W, Hout=xs.H * 2, Wout=xs.W * 2, oscale=2, oo=(ph, pw))
                        du = self.conv_desc(**kw_) if (one_launch or s2) else self.conv(what=nm + f" phase {ph}{pw}", **kw_)
                        phases.append((du, gwp))
                if one_launch and not s2:
                    arr = (N.ConvDesc * 4)(*[du for du, _ in phases])
                    self.plan.keepalive.append(arr)
                    rows_in = B * T * xs.H * xs.W
                    self.step(self.sp("vmm_conv_igemm_", "_batched") if self.igemm_one else lib.vmm_conv_igemm_bf16x3_batched, (arr, 4), nm + " (4 phases)", flops=4 * 2.0 * rows_in * 4 * ci_ * co_,
                              nbytes=4.0 * (rows_in * ci_ + 16 * ci_ * co_ + 4 * rows_in * co_))
                self.free_temps(tmps_u)

                def up_bwd(nm=nm, xs=xs, u=u, phases=phases, ci_=ci_, co_=co_):
                    gu, _ = self.grad_of(u)
                    for du, gwp in phases:  # every output row belongs to exactly one phase: the four launches together give the bias gradient
                        self.wgrad(du, gu.ptr, co_, gwp, nm, gb_ptr=self.pg(nm + ".bias"))
                    gx, acc = self.grad_of(xs)
                    # dX[a][ci] = sum_{kh,kw,co} dU[2a-1+kh][co] W[ci][co][kh][kw]: a stride-2 conv over dU with [(kh,kw,co)][ci]
                    if self.x3 and not wrap and _enabled("s2_dgrad") and lib.vmm_conv_s2_supported(B * T, gu.H, gu.W, co_, ci_, 0):
                        # ... i.e. the Downsample kernel over dU with the (Cin, Cout, 1, 4, 4) tensor read as a convolution weight (N = Cin, C = Cout)
                        wt = self.pack(nm + ".weight", 36 * co_ * ci_, want_grad=False, TH=4, TW=4, C=co_, Cp=co_, N=ci_, sn=co_ * 16, sc=16, sh=4, sw=1, fmt=5)[0]
                        self.step(lib.vmm_conv_s2_acc_bf16x3, (gu.ptr, gu.ld, wt, 0, gx.ptr if acc else 0, ci_, gx.ptr, ci_, B * T, gu.H, gu.W, co_, ci_, 0, self.tickets(), N_TICKETS),
                                  nm + " dgrad", flops=2.0 * B * T * xs.H * xs.W * 16 * ci_ * co_, nbytes=4.0 * (gu.n + (2 if acc else 1) * gx.n + 16 * ci_ * co_))
                        return
                    wp = self.pack(nm + ".weight", 16 * co_ * ci_, want_grad=False, gemm=self.x3, TH=4, TW=4, C=co_, Cp=co_, N=ci_, sn=co_ * 16, sc=16, sh=4, sw=1,
                                   hs=1, ws=1)[0]
                    self.conv(a1=gu, w=wp, Cout=ci_, KH=4, KW=4, stride=2, off=(-1, -1), out_ptr=gx.ptr, ldo=ci_, Hv=xs.H, Wv=xs.W, res_ptr=gx.ptr if acc else 0,
                              ldres=ci_, what=nm + " dgrad", x3w=self.x3)
                self.on_backward(up_bwd, pg_start, uj_start)
                self.free_act(xs)
                x = u
        fc0 = self.shapes["final_conv.0.block1.proj.weight"][0]
        if not tr and fc0 == 64 and m.out_dim <= 4 and _enabled("final_tail"):
            # the last block's output pass and the final 1x1 convolution in one kernel: the block's output is never stored
            def fused_tail(h2, c2_ptr, res_ptr, ldres, bf16_maps=False):
                self.step(lib.vmm_affine_silu_pointwise_to_ncthw_a16 if bf16_maps else lib.vmm_affine_silu_pointwise_to_ncthw,
                          (h2.ptr, h2.ld, c2_ptr, res_ptr, ldres, 64, self.wraw("final_conv.1.weight"),
                           self.wraw("final_conv.1.bias"), B, m.out_dim, T, H * W, self.ptr(out_off)),
                          "final_conv.0 out + final_conv.1", nbytes=4.0 * (2 * h2.n + B * m.out_dim * T * H * W))
            self.resnet_block("final_conv.0", x, r, None, tail=fused_tail)
            self.free_act(x)
            self.free_act(r)
            self._touch("final_conv.1.weight")
            self._touch("final_conv.1.bias")
            return self.plan
        f = self.resnet_block("final_conv.0", x, r, None)
        self.free_act(x)
        self.free_act(r)
        pg_start, uj_start = self.pgtop, len(self.unpack_jobs)
        self.step(lib.vmm_pointwise_to_ncthw, (f.ptr, f.ld, f.C, self.wraw("final_conv.1.weight"), self.wraw("final_conv.1.bias"), B, m.out_dim, T, H * W,
                                               self.ptr(out_off)), "final_conv.1")
        self._touch("final_conv.1.weight")
        self._touch("final_conv.1.bias")

        def final_bwd():
            gf, _ = self.grad_of(f)
            self.step(lib.vmm_pointwise_to_ncthw_bwd, (f.ptr, f.ld, f.C, self.wraw("final_conv.1.weight"), self.ptr(dout_off), B, m.out_dim, T, H * W, gf.ptr, f.C,
                                                       self.pg("final_conv.1.weight") or self.scratch(m.out_dim * f.C),
                                                       self.pg("final_conv.1.bias") or self.scratch(m.out_dim)), "final_conv.1 bwd")
        self.on_backward(final_bwd, pg_start, uj_start)
        self.free_act(f)

        if tr:
            # Emit the backward list in reverse block order.  After block k's emitter (plus the scatter of the packed weight
            # gradients registered by blocks >= k) every gradient at offsets >= pg_start(k) is final: parameters are laid out in
            # first-use order, and the only parameters touched again later in the backward (conditioning / time embedding,
            # token k/v, FiLM layers) were first used in the embedding stage, i.e. they sit in front of every block.
            self.in_bwd = True
            uj_hi = len(self.unpack_jobs)
            # The packed weight gradients of a block are scattered into the flat torch-layout buffer -- and the buffer's tail marked final for the
            # data-parallel reducer -- once at least `span` floats of it are pending: a scatter launch and a mark per block were 45 launches of
            # ~15 us, most of them for the full-resolution blocks whose parameters are a few tens of thousands of floats (the reducer merges marks
            # into >= 16 MB buckets anyway).
            # (round 4: 250 k -> 2 M floats.  The reducer's buckets are >= 4 M floats, so marks finer than that bought nothing, and every scatter launch
            # is ~19 us: 31 -> ~19 launches per step)
            span = min(2_000_000, max(1, self.pgtop // 12)) if _enabled("scatter_merge") else 1
            marked_pg, pend_lo = self.pgtop, uj_hi
            rev = list(reversed(self.tape))
            for idx, (emit, pg_start, uj_start) in enumerate(rev):
                emit()
                pend_lo = min(pend_lo, uj_start)
                if marked_pg - pg_start >= span or idx == len(rev) - 1:
                    self.flush_reductions()  # (the scatter reads the packed gradients: every pending partial block is totalled first)
                    self.emit_unpack(self.unpack_jobs[pend_lo:uj_hi], "scatter weight gradients")
                    uj_hi = min(uj_hi, pend_lo)
                    marked_pg = pg_start
                    if self.plan.bwd_steps:
                        self.plan.bwd_marks.append((len(self.plan.bwd_steps) - 1, pg_start))
            self.in_bwd = False
        return self.plan


def build_plan(model, B: int, T: int, H: int, W: int, cond_len: int, device, *, training: bool = False, mirrored: bool = False, focus: bool = False) -> Plan:
    # plan buffers outlive the caller's autograd mode: never create them as inference tensors
    with torch.inference_mode(False), torch.no_grad():
        return _build_plan(model, B, T, H, W, cond_len, device, training, mirrored, focus)


def _build_plan(model, B: int, T: int, H: int, W: int, cond_len: int, device, training: bool, mirrored: bool = False, focus: bool = False) -> Plan:
    # pass 1: sizes only (addresses relative to 0); pass 2: identical allocation order over real buffers
    # (fake but non-zero bases, so "pointer or None" decisions are identical in both passes)
    sizing = _Builder(model, B, T, H, W, cond_len, device, (1 << 40, 1 << 41, 1 << 42, 1 << 43), training, mirrored, focus)
    sizing.build()
    arena = torch.empty(sizing.arena.peak + ALIGN, dtype=torch.float32, device=device)
    if os.environ.get("VMM_POISON_ARENA"):  # debug runs: NaN in every activation slot, so that a kernel that reads what no launch of the plan wrote shows
        arena.fill_(float("nan"))
    wbuf = torch.zeros(sizing.wtop + ALIGN, dtype=torch.float32, device=device)
    pgrad = torch.zeros(sizing.pgtop + ALIGN, dtype=torch.float32, device=device) if training else None
    gscr = torch.zeros(sizing.gstop + ALIGN, dtype=torch.float32, device=device) if training else None
    bases = (arena.data_ptr(), wbuf.data_ptr(), pgrad.data_ptr() if training else 0, gscr.data_ptr() if training else 0)
    b = _Builder(model, B, T, H, W, cond_len, device, bases, training, mirrored, focus)
    plan = b.build()
    assert b.arena.peak == sizing.arena.peak and b.wtop == sizing.wtop and b.pgtop == sizing.pgtop and b.gstop == sizing.gstop
    plan.arena, plan.wbuf, plan.pgrad, plan.gscratch = arena, wbuf, pgrad, gscr
    plan.alloc_log = list(b.arena.log)
    plan.shape = (B, T, H, W, cond_len)
    plan.mirrored = b.mirrored
    for off, host in b.job_uploads:
        wbuf[off:off + (host.numel() + 3) // 4].view(torch.uint8)[: host.numel()].copy_(host)
    for off, t in b.consts:
        arena[off:off + t.numel()].copy_(t.to(device))
    x_off, t_off, c_off, m_off, o_off, do_off = b.io
    Cx = model.channels
    plan.x_in = arena[x_off:x_off + B * Cx * T * H * W].view(B, Cx, T, H, W)
    plan.time_in = arena[t_off:t_off + B * 2].view(torch.int64)
    plan.cond_in = arena[c_off:c_off + B * cond_len].view(B, cond_len)
    plan.mask_in = arena[m_off:m_off + (B + 3) // 4].view(torch.uint8)[:B]
    if b.focus:
        plan.focus_in = arena[b.focus_off:b.focus_off + (B + 3) // 4].view(torch.uint8)[:B]
    plan.out = arena[o_off:o_off + B * model.out_dim * T * H * W].view(B, model.out_dim, T, H, W)
    plan.arena_floats = sizing.arena.peak
    if training:
        plan.dx = arena[b.dx_off:b.dx_off + B * Cx * T * H * W].view(B, Cx, T, H, W)
        plan.dout = arena[do_off:do_off + B * model.out_dim * T * H * W].view(B, model.out_dim, T, H, W)
        plan.pgrad_floats = sizing.pgtop
    return plan
