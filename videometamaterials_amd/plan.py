"""Static execution plans for the Unet3D hot path.

A plan = (arena in HBM, packed weights, ordered list of C-ABI kernel launches) for one
input shape.  It is built once (``build_plan``), its device addresses never change, and
``Plan.run`` just replays the launches on the caller's current HIP stream -- which also
makes the whole denoiser capturable into a hipGraph (diffusion.py does that for the
256-step sampling loop).

Working layout: every feature map is "rows x channels" fp32 with rows = (b, t, h, w)
(frame-major channels-last).  See DESIGN.md for the kernel inventory and data layout.
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Tuple

import torch

from . import _native as N
from . import hostmath

ALIGN = 64  # floats (256 B)
LA_PART = 32 * 32 + 64  # linear-attention partial record (attention.hip)
Q_STRIDE = 4096 + 16  # quantile scratch words per sample (diffusion.hip)


def _stream() -> C.c_void_p:
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


@dataclass
class Act:
    """A feature map living in the arena."""
    off: int  # float offset in the arena
    C: int
    H: int
    W: int
    n: int  # floats
    ptr: int = 0

    @property
    def ld(self):
        return self.C


class _Arena:
    """First-fit allocator over float offsets; used identically in the sizing pass and the real pass."""

    def __init__(self, keep_all: bool):
        self.free_list: List[Tuple[int, int]] = []  # (off, n) sorted by off
        self.top = 0
        self.peak = 0
        self.keep_all = keep_all

    def alloc(self, n: int) -> int:
        n = (n + ALIGN - 1) // ALIGN * ALIGN
        for i, (off, sz) in enumerate(self.free_list):
            if sz >= n:
                if sz == n:
                    self.free_list.pop(i)
                else:
                    self.free_list[i] = (off + n, sz - n)
                return off
        off = self.top
        self.top += n
        self.peak = max(self.peak, self.top)
        return off

    def free(self, off: int, n: int) -> None:
        if self.keep_all:
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


class Plan:
    def __init__(self):
        self.steps: List[Tuple[Callable, tuple, str]] = []
        self.meta: List[Tuple[str, float, float]] = []  # per step: (kernel family, algorithmic flops, algorithmic bytes)
        self.arena: Optional[torch.Tensor] = None
        self.wbuf: Optional[torch.Tensor] = None
        self.packers: List[Tuple[int, int, Callable]] = []  # (off, n, fn(params) -> tensor)
        self.keepalive: list = []
        self.x_in = self.time_in = self.cond_in = self.mask_in = self.out = None
        self.weights_version = None
        self.named: Dict[str, Act] = {}  # debug taps (name -> feature map)
        self.shape = None

    def refresh_weights(self, params: Dict[str, torch.Tensor]) -> None:
        with torch.no_grad():
            for off, n, fn in self.packers:
                src = fn(params).reshape(-1)
                self.wbuf[off:off + src.numel()].copy_(src)

    def launch(self) -> None:
        s = _stream()
        for fn, args, what in self.steps:
            rc = fn(*args, s)
            if rc != 0:
                N.check(rc, what)

    def launch_timed(self) -> List[float]:
        """Replay with a HIP event pair around every launch (on the current stream); returns ms per step."""
        s = _stream()
        evs = []
        for fn, args, what in self.steps:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            rc = fn(*args, s)
            e1.record()
            if rc != 0:
                N.check(rc, what)
            evs.append((e0, e1))
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in evs]

    def run(self, x, time, cond, mask) -> torch.Tensor:
        """Copy the inputs into the static slots, replay, return the static (B,C,T,H,W) output view."""
        self.x_in.copy_(x)
        self.time_in.copy_(time)
        self.cond_in.copy_(cond)
        self.mask_in.copy_(mask)
        self.launch()
        return self.out

    def tap(self, name: str) -> torch.Tensor:
        """Debug: a (rows, C) view of a named intermediate (only meaningful for keep_all plans)."""
        a = self.named[name]
        return self.arena[a.off:a.off + a.n].view(-1, a.C)


def cfg_combine(eps_c: torch.Tensor, eps_n: torch.Tensor, w: float) -> torch.Tensor:
    """null + (cond - null) * w  (vddp.py:728), on device through the diffusion kernel set."""
    out = torch.empty_like(eps_c)
    lib = N.lib()
    # predict_x0 with c_recip = 0, c_recipm1 = -1 would also do it; a dedicated axpby keeps the rounding of the reference:
    # null + (cond - null) * w  ==  vmm_predict_x0's eps formula, exposed via vmm_axpby on (cond - null) is not bit-identical,
    # so the guidance formula lives in vmm_cfg_combine.
    rc = lib.vmm_cfg_combine(eps_c.data_ptr(), eps_n.data_ptr(), C.c_float(w), out.data_ptr(), eps_c.numel(), _stream())
    N.check(rc, "vmm_cfg_combine")
    return out


# ====================================================================================== builder
class _Builder:
    def __init__(self, model, B, T, H, W, cond_len, device, base_ptr: int, wbase_ptr: int, keep_all: bool):
        self.m = model
        self.B, self.T, self.H, self.W, self.cond_len = B, T, H, W, cond_len
        self.device = device
        self.base, self.wbase = base_ptr, wbase_ptr
        self.arena = _Arena(keep_all)
        self.wtop = 0
        self.plan = Plan()
        self.lib = N.lib()
        self.shapes = {k: tuple(v.shape) for k, v in model._params_flat().items()}
        self.G = model.resnet_groups
        self.heads = model.attn_heads

    # ---------------------------------------------------------------- memory
    def alloc(self, n: int) -> int:
        return self.arena.alloc(n)

    def free(self, off: int, n: int) -> None:
        self.arena.free(off, n)

    def ptr(self, off: int) -> int:
        return self.base + off * 4

    def act(self, C_: int, H: int, W: int) -> Act:
        n = self.B * self.T * H * W * C_
        off = self.alloc(n)
        return Act(off, C_, H, W, n, self.ptr(off))

    def free_act(self, a: Act) -> None:
        self.free(a.off, a.n)

    def wslot(self, n: int, packer: Callable) -> int:
        """Reserve n floats in the packed-weight buffer; `packer(params) -> tensor` fills it on refresh."""
        off = self.wtop
        self.wtop += (n + ALIGN - 1) // ALIGN * ALIGN
        self.plan.packers.append((off, n, packer))
        return self.wbase + off * 4

    def wraw(self, name: str) -> int:
        """Static copy of a parameter in its torch layout."""
        n = int(math.prod(self.shapes[name]))
        return self.wslot(n, lambda p, name=name: p[name])

    def step(self, fn, args: tuple, what: str, flops: float = 0.0, nbytes: float = 0.0) -> None:
        self.plan.steps.append((fn, args, what))
        self.plan.meta.append((fn.__name__, float(flops), float(nbytes)))

    # ---------------------------------------------------------------- emitters
    def conv(self, *, a1: Act, a2: Optional[Act] = None, w: int, bias: int = 0, Cout: int, KH: int = 1, KW: int = 1, stride: int = 1,
             off: Tuple[int, int] = (0, 0), sgn: Tuple[int, int] = (1, 1), out_ptr: int, ldo: int, Hv: int, Wv: int, Hout: int = 0,
             Wout: int = 0, oscale: int = 1, oo: Tuple[int, int] = (0, 0), res_ptr: int = 0, ldres: int = 0, rot_tab: int = 0,
             rot_ncols: int = 0, q_scale: float = 1.0, q_ncols: int = 0, a_coef: int = 0, what: str = "conv", a1_C: Optional[int] = None):
        d = N.ConvDesc()
        d.a1, d.C1, d.lda1 = a1.ptr, (a1_C if a1_C is not None else a1.C), a1.ld
        if a2 is not None:
            d.a2, d.C2, d.lda2 = a2.ptr, a2.C, a2.ld
        d.w, d.bias = w, bias or None
        d.res, d.ldres = res_ptr or None, ldres
        d.out, d.ldo = out_ptr, ldo
        d.nimg, d.Hin, d.Win = self.B * self.T, a1.H, a1.W
        d.Hv, d.Wv, d.stride = Hv, Wv, stride
        d.KH, d.KW, d.off_h, d.off_w, d.sgn_h, d.sgn_w = KH, KW, off[0], off[1], sgn[0], sgn[1]
        d.Hout, d.Wout, d.oscale, d.ooh, d.oow = Hout or Hv, Wout or Wv, oscale, oo[0], oo[1]
        d.Cout = Cout
        d.rot_tab = rot_tab or None
        d.rot_T, d.rot_HW, d.rot_ncols, d.rot_dh = self.T, a1.H * a1.W, rot_ncols, 32
        d.q_scale, d.q_ncols = q_scale, q_ncols
        d.a_mode, d.a_coef, d.a_imgs_per_sample = (1 if a_coef else 0), a_coef or None, self.T
        self.plan.keepalive.append(d)
        M = d.nimg * Hv * Wv
        K = KH * KW * (d.C1 + d.C2)
        # algorithmic work: every input element, weight and output element touched once
        nbytes = 4.0 * (d.nimg * a1.H * a1.W * (d.C1 + d.C2) + K * Cout + M * Cout + (M * Cout if res_ptr else 0))
        self.step(self.lib.vmm_conv_igemm_f32, (C.byref(d),), what, flops=2.0 * M * K * Cout, nbytes=nbytes)

    def pack_conv(self, name: str, pad_cin_to: int = 0) -> int:
        """(Cout, Cin, 1, KH, KW) -> [(kh, kw, ci)][Cout]"""
        co, ci, _, kh, kw = self.shapes[name]
        cip = max(ci, pad_cin_to)

        def packer(p, name=name, ci=ci, cip=cip):
            w = p[name][:, :, 0].permute(2, 3, 1, 0)  # kh kw ci co
            if cip != ci:
                w = torch.nn.functional.pad(w, (0, 0, 0, cip - ci))
            return w.contiguous()

        return self.wslot(kh * kw * cip * co, packer)

    def pack_linear(self, name: str) -> int:
        """(out, in[,1,1[,1]]) -> [in][out]"""
        shp = self.shapes[name]
        co, ci = shp[0], shp[1]
        return self.wslot(co * ci, lambda p, name=name, co=co, ci=ci: p[name].reshape(co, ci).t().contiguous())

    def pack_convT_phase(self, name: str, ph: int, pw: int) -> int:
        """ConvTranspose (Cin, Cout, 1, 4, 4), output phase (ph, pw): taps kh = (1-ph) + 2*kh', dh = ph - kh'."""
        ci, co = self.shapes[name][0], self.shapes[name][1]

        def packer(p, name=name, ph=ph, pw=pw):
            w = p[name][:, :, 0]  # ci co kh kw
            w = w[:, :, [1 - ph, 3 - ph], :][:, :, :, [1 - pw, 3 - pw]]
            return w.permute(2, 3, 0, 1).contiguous()  # kh' kw' ci co

        return self.wslot(4 * ci * co, packer)

    def gn_coef(self, h: Act, prefix: str, film_ptr: int, ldfilm: int) -> Tuple[int, int, int]:
        """GroupNorm statistics of h and the fused (scale, shift) coefficients; returns (coef_off, n, ptr)."""
        B, G, C_ = self.B, self.G, h.C
        rows_ps = self.T * h.H * h.W
        sums_off = self.alloc(B * G * 4)
        coef_off = self.alloc(B * C_ * 2)
        self.step(self.lib.vmm_groupnorm_stats, (h.ptr, h.ld, B, rows_ps, C_, G, self.ptr(sums_off)), prefix + ".norm stats")
        self.step(self.lib.vmm_groupnorm_coef,
                  (self.ptr(sums_off), rows_ps * (C_ // G), C.c_float(1e-5), self.wraw(prefix + ".norm.weight"), self.wraw(prefix + ".norm.bias"),
                   film_ptr or None, ldfilm, B, C_, G, self.ptr(coef_off), None), prefix + ".norm coef")
        self.free(sums_off, B * G * 4)
        return coef_off, B * C_ * 2, self.ptr(coef_off)

    def resnet_block(self, name: str, x1: Act, x2: Optional[Act], film_ptr: int) -> Act:
        """ResnetBlock (vddp.py:287-311): conv-GN-FiLM-SiLU, conv-GN-SiLU, + res_conv(x)."""
        Cout = self.shapes[name + ".block1.proj.weight"][0]
        H, W = x1.H, x1.W
        rows = self.B * self.T * H * W
        h1 = self.act(Cout, H, W)
        self.conv(a1=x1, a2=x2, w=self.pack_conv(name + ".block1.proj.weight"), bias=self.wraw(name + ".block1.proj.bias"), Cout=Cout, KH=3, KW=3,
                  off=(-1, -1), out_ptr=h1.ptr, ldo=Cout, Hv=H, Wv=W, what=name + ".block1.proj")
        c1_off, c1_n, c1_ptr = self.gn_coef(h1, name + ".block1", film_ptr, 2 * Cout)
        h2 = self.act(Cout, H, W)
        self.conv(a1=h1, w=self.pack_conv(name + ".block2.proj.weight"), bias=self.wraw(name + ".block2.proj.bias"), Cout=Cout, KH=3, KW=3,
                  off=(-1, -1), out_ptr=h2.ptr, ldo=Cout, Hv=H, Wv=W, a_coef=c1_ptr, what=name + ".block2.proj")
        self.free_act(h1)
        self.free(c1_off, c1_n)
        c2_off, c2_n, c2_ptr = self.gn_coef(h2, name + ".block2", 0, 0)
        if (name + ".res_conv.weight") in self.shapes:
            r = self.act(Cout, H, W)
            self.conv(a1=x1, a2=x2, w=self.pack_linear(name + ".res_conv.weight"), bias=self.wraw(name + ".res_conv.bias"), Cout=Cout,
                      out_ptr=r.ptr, ldo=Cout, Hv=H, Wv=W, what=name + ".res_conv")
            res_ptr, ldres = r.ptr, Cout
        else:
            assert x2 is None and x1.C == Cout
            r, res_ptr, ldres = None, x1.ptr, x1.ld
        self.step(self.lib.vmm_affine_silu, (h2.ptr, Cout, c2_ptr, res_ptr, ldres, h2.ptr, Cout, rows, self.T * H * W, Cout), name + " out")
        if r is not None:
            self.free_act(r)
        self.free(c2_off, c2_n)
        self.plan.named[name] = h2
        return h2

    def layernorm(self, x: Act, gamma_name: str) -> Act:
        y = self.act(x.C, x.H, x.W)
        rows = self.B * self.T * x.H * x.W
        self.step(self.lib.vmm_channel_layernorm, (x.ptr, x.ld, self.wraw(gamma_name), y.ptr, y.ld, rows, x.C, C.c_float(1e-5)), gamma_name)
        return y

    def linear_attn_block(self, name: str, x: Act, ekv: Optional[Tuple[int, int, int]]) -> Act:
        """Residual(PreNorm(SpatialLinearAttention)) (vddp.py:313-378, 679)."""
        B, T, heads = self.B, self.T, self.heads
        hid = 32 * heads
        HW = x.H * x.W
        y = self.layernorm(x, name + ".fn.norm.gamma")
        qkv = self.act(3 * hid, x.H, x.W)
        self.conv(a1=y, w=self.pack_linear(name + ".fn.fn.to_qkv.weight"), Cout=3 * hid, out_ptr=qkv.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W, what=name + " to_qkv")
        self.free_act(y)
        nsplit = max(1, min((HW + 63) // 64, -(-2048 // (B * T * heads))))
        part_n, ctx_n = B * T * heads * nsplit * LA_PART, B * T * heads * 1024
        part, ctx = self.alloc(part_n), self.alloc(ctx_n)
        ek, ev, ntok = ekv if ekv else (0, 0, 0)
        self.step(self.lib.vmm_linattn_context, (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, B, T, HW, heads, 32, nsplit, self.ptr(part), self.ptr(ctx)),
                  name + " context")
        o = self.act(hid, x.H, x.W)
        self.step(self.lib.vmm_linattn_apply, (qkv.ptr, 3 * hid, self.ptr(ctx), o.ptr, hid, B, T, HW, heads, 32), name + " apply")
        self.free_act(qkv)
        self.free(part, part_n)
        self.free(ctx, ctx_n)
        out = self.act(x.C, x.H, x.W)
        self.conv(a1=o, w=self.pack_linear(name + ".fn.fn.to_out.weight"), bias=self.wraw(name + ".fn.fn.to_out.bias"), Cout=x.C, out_ptr=out.ptr,
                  ldo=x.C, Hv=x.H, Wv=x.W, res_ptr=x.ptr, ldres=x.ld, what=name + " to_out")
        self.free_act(o)
        self.plan.named[name] = out
        return out

    def softmax_attn_block(self, name: str, x: Act, ekv: Optional[Tuple[int, int, int]], *, temporal: bool) -> Act:
        """Residual(PreNorm(EinopsToAndFrom(Attention))) (vddp.py:396-535; 615/630/680 temporal, 687-689 mid spatial)."""
        B, T, heads = self.B, self.T, self.heads
        hid = 32 * heads
        HW = x.H * x.W
        p = name + ".fn.fn.fn"
        y = self.layernorm(x, name + ".fn.norm.gamma")
        qkv = self.act(3 * hid, x.H, x.W)
        self.conv(a1=y, w=self.pack_linear(p + ".to_qkv.weight"), Cout=3 * hid, out_ptr=qkv.ptr, ldo=3 * hid, Hv=x.H, Wv=x.W,
                  rot_tab=self.rot_ptr if temporal else 0, rot_ncols=2 * hid if temporal else 0, q_scale=32 ** -0.5, q_ncols=hid, what=name + " to_qkv")
        self.free_act(y)
        o = self.act(hid, x.H, x.W)
        ek, ev, ntok = ekv if ekv else (0, 0, 0)
        if temporal:
            self.step(self.lib.vmm_temporal_attention,
                      (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, self.bias_ptr, 1 if self.m.per_frame_cond else 0, o.ptr, hid, B, T, HW, heads, 32),
                      name + " core")
        else:
            self.step(self.lib.vmm_spatial_attention,
                      (qkv.ptr, 3 * hid, ek or None, ev or None, ntok, 1 if self.m.per_frame_cond else 0, o.ptr, hid, B, T, HW, heads, 32), name + " core")
        self.free_act(qkv)
        out = self.act(x.C, x.H, x.W)
        self.conv(a1=o, w=self.pack_linear(p + ".to_out.weight"), Cout=x.C, out_ptr=out.ptr, ldo=x.C, Hv=x.H, Wv=x.W, res_ptr=x.ptr, ldres=x.ld,
                  what=name + " to_out")
        self.free_act(o)
        self.plan.named[name] = out
        return out

    # ---------------------------------------------------------------- dense job tables
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
            max_units = max(max_units, j["N"] * ((j["rows"] + 7) // 8))
        nbytes = C.sizeof(arr)
        nfl = (nbytes + 3) // 4
        off = self.alloc(nfl)  # never freed: the table must stay resident
        host = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).clone()
        self.plan.keepalive.append((off, host))
        self.job_uploads.append((off, host))
        self.step(self.lib.vmm_dense_batched, (self.ptr(off), len(jobs), max_units), what)

    # ---------------------------------------------------------------- the network
    def build(self) -> Plan:
        m, B, T, H, W = self.m, self.B, self.T, self.H, self.W
        lib, heads = self.lib, self.heads
        td, cd, D = m.time_dim, m.cond_dim, m.cond_dim
        Cx = m.channels
        self.job_uploads: List[Tuple[int, torch.Tensor]] = []
        rows0 = B * T * H * W
        # static inputs / outputs
        x_in_off = self.alloc(B * Cx * T * H * W)
        time_off = self.alloc(B * 2)
        cond_off = self.alloc(B * self.cond_len)
        mask_off = self.alloc((B + 3) // 4)
        out_off = self.alloc(B * m.out_dim * T * H * W)
        self.io = (x_in_off, time_off, cond_off, mask_off, out_off)
        # constant tables
        rot = hostmath.rotary_table(T, 32)
        rot_off = self.alloc(rot.numel())
        bk = hostmath.relpos_buckets(T, 32, 32)
        bk_off = self.alloc(bk.numel())
        self.consts = [(rot_off, rot.reshape(-1)), (bk_off, bk.reshape(-1).view(torch.float32))]
        self.rot_ptr = self.ptr(rot_off)
        bias_off = self.alloc(heads * T * T)
        self.bias_ptr = self.ptr(bias_off)
        self.step(lib.vmm_relpos_bias, (self.wraw("time_rel_pos_bias.relative_attention_bias.weight"), self.ptr(bk_off), T, heads, self.bias_ptr),
                  "time_rel_pos_bias")

        # ---- conditioning / time embedding (vddp.py:745-795)
        se = self.alloc(B * m.dim)
        self.step(lib.vmm_sinusoidal_embed, (self.ptr(time_off), B, m.dim, C.c_float(-(math.log(10000) / (m.dim // 2 - 1))), self.ptr(se)), "time sinusoid")
        h1, t2, hidden, temb = self.alloc(B * td), self.alloc(B * td), self.alloc(B * td), self.alloc(B * td)
        ntok = m.cond_attention_tokens
        tokens = self.alloc(B * ntok * D) if m.cond_attention != "none" else None
        lvl1 = [dict(x=self.ptr(se), w=self.wraw("time_mlp.1.weight"), b=self.wraw("time_mlp.1.bias"), y=self.ptr(h1), rows=B, K=m.dim, N=td, act_out=2)]
        lvl2 = [dict(x=self.ptr(h1), w=self.wraw("time_mlp.3.weight"), b=self.wraw("time_mlp.3.bias"), y=self.ptr(t2), rows=B, K=td, N=td)]
        if m.per_frame_cond:
            if self.cond_len != ntok:
                raise ValueError(f"per_frame_cond expects cond of shape (b, {ntok})")
            pooled, pl, c1 = self.alloc(B * D), self.alloc(B * D), self.alloc(B * D)
            self.step(lib.vmm_cond_tokens, (self.ptr(cond_off), self.wraw("sign_emb.weight"), self.wraw("sign_emb.bias"), self.wraw("null_text_token"),
                                            self.ptr(mask_off), B, ntok, D, self.ptr(tokens), self.ptr(pooled)), "sign_emb tokens")
            self.step(lib.vmm_rows_layernorm_affine, (self.ptr(pooled), self.wraw("cond_token_to_hidden.0.weight"), self.wraw("cond_token_to_hidden.0.bias"),
                                                      self.ptr(pl), B, D, C.c_float(1e-5)), "cond_token_to_hidden.0")
            lvl1.append(dict(x=self.ptr(pl), w=self.wraw("cond_token_to_hidden.1.weight"), b=self.wraw("cond_token_to_hidden.1.bias"), y=self.ptr(c1), rows=B,
                             K=D, N=D, act_out=1))
            lvl2.append(dict(x=self.ptr(c1), w=self.wraw("cond_token_to_hidden.3.weight"), b=self.wraw("cond_token_to_hidden.3.bias"), y=self.ptr(hidden),
                             rows=B, K=D, N=td))
        else:
            chain = [1, 16, 32, 64, 128, cd]
            L = self.cond_len
            cur, cur_C = self.ptr(cond_off), 1
            for i, co in enumerate(chain[1:]):
                Lout = (L + 2 - 4) // 2 + 1
                nxt = self.ptr(hidden) if i == 4 else self.ptr(self.alloc(B * co * Lout))
                self.step(lib.vmm_conv1d_k4s2_silu, (cur, self.wraw(f"sign_emb_CNN.emb_model.{2 * i}.weight"), self.wraw(f"sign_emb_CNN.emb_model.{2 * i}.bias"), nxt,
                                                     B, cur_C, co, L), f"sign_emb_CNN.{2 * i}")
                cur, cur_C, L = nxt, co, Lout
            if L != 1:
                raise ValueError("sign_emb_CNN must reduce the conditioning signal to length 1 (cond length 32..63)")
            if tokens is not None:
                self.step(lib.vmm_tokens_from_hidden, (self.ptr(hidden), self.wraw("null_text_token"), self.ptr(mask_off), B, ntok, D, self.ptr(tokens)), "tokens")
        self.dense_level(lvl1, "embed level 1")
        self.dense_level(lvl2, "embed level 2")
        self.step(lib.vmm_select_add, (self.ptr(hidden), self.wraw("null_text_hidden"), self.ptr(mask_off), self.ptr(t2), self.ptr(temb), B, td), "t + hidden")

        # level 3: every ResnetBlock.mlp and every to_k/to_v on the tokens, one launch
        lvl3: List[dict] = []
        film: Dict[str, int] = {}
        res_names = [f"downs.{i}.{j}" for i in range(len(m.in_out)) for j in (0, 1)] + ["mid_block1", "mid_block2"] + \
                    [f"ups.{i}.{j}" for i in range(len(m.in_out)) for j in (0, 1)]
        for rn in res_names:
            n_out = self.shapes[rn + ".mlp.1.weight"][0]
            fo = self.alloc(B * n_out)
            film[rn] = self.ptr(fo)
            lvl3.append(dict(x=self.ptr(temb), w=self.wraw(rn + ".mlp.1.weight"), b=self.wraw(rn + ".mlp.1.bias"), y=self.ptr(fo), rows=B, K=td, N=n_out, act_in=1))
        ekv: Dict[str, Tuple[int, int, int]] = {}
        hid = 32 * heads
        rot_sites: List[int] = []

        def add_ekv(site: str, pfx: str, rotate: bool):
            eo, vo = self.alloc(B * ntok * hid), self.alloc(B * ntok * hid)
            lvl3.append(dict(x=self.ptr(tokens), w=self.wraw(pfx + ".to_k.weight"), y=self.ptr(eo), rows=B * ntok, K=D, N=hid))
            lvl3.append(dict(x=self.ptr(tokens), w=self.wraw(pfx + ".to_v.weight"), y=self.ptr(vo), rows=B * ntok, K=D, N=hid))
            ekv[site] = (self.ptr(eo), self.ptr(vo), ntok)
            if rotate:
                rot_sites.append(self.ptr(eo))

        if tokens is not None:
            lvls = range(len(m.in_out))
            for i in lvls:
                for side in ("downs", "ups"):
                    if m.use_sparse_linear_attn:
                        add_ekv(f"{side}.{i}.2", f"{side}.{i}.2.fn.fn", False)
                    if m.use_temporal_attention_cond:
                        add_ekv(f"{side}.{i}.3", f"{side}.{i}.3.fn.fn.fn", m.per_frame_cond)
            add_ekv("mid_spatial_attn", "mid_spatial_attn.fn.fn.fn", False)
            if m.use_temporal_attention_cond:
                add_ekv("mid_temporal_attn", "mid_temporal_attn.fn.fn.fn", m.per_frame_cond)
        self.dense_level(lvl3, "embed level 3 (film + token k/v)")
        if rot_sites:
            if ntok > T:
                raise ValueError("rotating token keys needs tokens <= frames")
            for p_ in rot_sites:
                self.step(lib.vmm_rotary_rows, (p_, self.rot_ptr, B, ntok, heads, 32), "rotate token keys")

        # ---- stem
        xin = Act(self.alloc(rows0 * 4), 4, H, W, rows0 * 4)
        xin.ptr = self.ptr(xin.off)
        self.step(lib.vmm_ncthw_to_rows, (self.ptr(x_in_off), B, Cx, T, H * W, xin.ptr, 4), "ncthw -> rows")
        k = m.init_kernel_size
        x = self.act(m.init_dim, H, W)
        if Cx > 4:
            raise NotImplementedError("more than 4 input channels")
        self.conv(a1=xin, w=self.pack_conv("init_conv.weight", pad_cin_to=4), bias=self.wraw("init_conv.bias"), Cout=m.init_dim, KH=k, KW=k, off=(-(k // 2), -(k // 2)),
                  out_ptr=x.ptr, ldo=m.init_dim, Hv=H, Wv=W, what="init_conv")
        self.free_act(xin)
        x_new = self.softmax_attn_block("init_temporal_attn", x, None, temporal=True)
        self.free_act(x)
        x = x_new
        r = x  # kept until the final block (vddp.py:744; no clone needed, every op is out of place)

        def stage(side: str, i: int, x1: Act, x2: Optional[Act]) -> Act:
            y1 = self.resnet_block(f"{side}.{i}.0", x1, x2, film[f"{side}.{i}.0"])
            if x1 is not r:
                self.free_act(x1)
            if x2 is not None:
                self.free_act(x2)
            y2 = self.resnet_block(f"{side}.{i}.1", y1, None, film[f"{side}.{i}.1"])
            self.free_act(y1)
            if m.use_sparse_linear_attn:
                y3 = self.linear_attn_block(f"{side}.{i}.2", y2, ekv.get(f"{side}.{i}.2"))
                self.free_act(y2)
            else:
                y3 = y2
            y4 = self.softmax_attn_block(f"{side}.{i}.3", y3, ekv.get(f"{side}.{i}.3"), temporal=True)
            self.free_act(y3)
            return y4

        skips: List[Act] = []
        n_lvl = len(m.in_out)
        for i in range(n_lvl):
            x = stage("downs", i, x, None)
            skips.append(x)
            if i < n_lvl - 1:
                d = self.act(x.C, x.H // 2, x.W // 2)
                self.conv(a1=x, w=self.pack_conv(f"downs.{i}.4.weight"), bias=self.wraw(f"downs.{i}.4.bias"), Cout=x.C, KH=4, KW=4, stride=2, off=(-1, -1),
                          out_ptr=d.ptr, ldo=x.C, Hv=x.H // 2, Wv=x.W // 2, what=f"downs.{i}.4")
                x = d
        # the deepest skip is also the mid input: keep it alive, do not free through `stage`
        mid_in = x
        y = self.resnet_block("mid_block1", mid_in, None, film["mid_block1"])
        y2 = self.softmax_attn_block("mid_spatial_attn", y, ekv.get("mid_spatial_attn"), temporal=False)
        self.free_act(y)
        y3 = self.softmax_attn_block("mid_temporal_attn", y2, ekv.get("mid_temporal_attn"), temporal=True)
        self.free_act(y2)
        x = self.resnet_block("mid_block2", y3, None, film["mid_block2"])
        self.free_act(y3)
        for i in range(n_lvl):
            x = stage("ups", i, x, skips.pop())
            if i < n_lvl - 1:
                u = self.act(x.C, x.H * 2, x.W * 2)
                for ph in range(2):
                    for pw in range(2):
                        self.conv(a1=x, w=self.pack_convT_phase(f"ups.{i}.4.weight", ph, pw), bias=self.wraw(f"ups.{i}.4.bias"), Cout=x.C, KH=2, KW=2,
                                  off=(ph, pw), sgn=(-1, -1), out_ptr=u.ptr, ldo=x.C, Hv=x.H, Wv=x.W, Hout=x.H * 2, Wout=x.W * 2, oscale=2, oo=(ph, pw),
                                  what=f"ups.{i}.4 phase {ph}{pw}")
                self.free_act(x)
                x = u
        f = self.resnet_block("final_conv.0", x, r, 0)
        self.free_act(x)
        self.free_act(r)
        self.step(lib.vmm_pointwise_to_ncthw, (f.ptr, f.ld, f.C, self.wraw("final_conv.1.weight"), self.wraw("final_conv.1.bias"), B, m.out_dim, T, H * W,
                                               self.ptr(out_off)), "final_conv.1")
        self.free_act(f)
        return self.plan


def build_plan(model, B: int, T: int, H: int, W: int, cond_len: int, device, *, training: bool = False) -> Plan:
    # plan buffers outlive the caller's autograd mode: never create them as inference tensors
    with torch.inference_mode(False), torch.no_grad():
        return _build_plan(model, B, T, H, W, cond_len, device, training)


def _build_plan(model, B: int, T: int, H: int, W: int, cond_len: int, device, training: bool) -> Plan:
    # pass 1: sizes only (addresses relative to 0); pass 2: identical allocation order over real buffers
    sizing = _Builder(model, B, T, H, W, cond_len, device, 0, 0, keep_all=training)
    sizing.build()
    arena = torch.empty(sizing.arena.peak + ALIGN, dtype=torch.float32, device=device)
    wbuf = torch.zeros(sizing.wtop + ALIGN, dtype=torch.float32, device=device)
    b = _Builder(model, B, T, H, W, cond_len, device, arena.data_ptr(), wbuf.data_ptr(), keep_all=training)
    plan = b.build()
    assert b.arena.peak == sizing.arena.peak and b.wtop == sizing.wtop
    plan.arena, plan.wbuf = arena, wbuf
    plan.shape = (B, T, H, W, cond_len)
    for off, host in b.job_uploads:
        arena[off:off + (host.numel() + 3) // 4].view(torch.uint8)[: host.numel()].copy_(host)
    for off, t in b.consts:
        arena[off:off + t.numel()].copy_(t.to(device))
    x_off, t_off, c_off, m_off, o_off = b.io
    Cx = model.channels
    plan.x_in = arena[x_off:x_off + B * Cx * T * H * W].view(B, Cx, T, H, W)
    plan.time_in = arena[t_off:t_off + B * 2].view(torch.int64)
    plan.cond_in = arena[c_off:c_off + B * cond_len].view(B, cond_len)
    plan.mask_in = arena[m_off:m_off + (B + 3) // 4].view(torch.uint8)[:B]
    plan.out = arena[o_off:o_off + B * model.out_dim * T * H * W].view(B, model.out_dim, T, H, W)
    plan.arena_floats = sizing.arena.peak
    return plan
