"""Unet3D -- MI355X-native drop-in for the reference denoiser.

Same constructor / forward / forward_with_guidance_scale signatures and the same
``state_dict`` key names as ``Unet3D`` in the reference
(denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py:574-821, "vddp.py"),
so ``main.py:62-80`` can construct it unchanged and a reference checkpoint loads.

Nothing here executes arithmetic in PyTorch: ``forward`` replays a *plan* -- a
static list of HIP kernel launches (libvmm_hip.so, include/vmm_kernels.h) over a
static arena in HBM, built once per input shape by ``plan.build_plan``.  There is
no fallback path; without the built library construction works (parameters are
plain tensors) but ``forward`` raises.
"""
from __future__ import annotations

import math
import os
from typing import Dict, List, Optional, Tuple

import torch
from torch import nn

from . import plan as _plan


class _Node(nn.Module):
    """Pure parameter container; gives parameters the reference's dotted names."""


def _attach(root: nn.Module, dotted: str, value: torch.Tensor, *, buffer: bool = False) -> None:
    parts = dotted.split(".")
    mod = root
    for p in parts[:-1]:
        if not hasattr(mod, p):
            mod.add_module(p, _Node())
        mod = getattr(mod, p)
    if buffer:
        mod.register_buffer(parts[-1], value, persistent=True)
    else:
        mod.register_parameter(parts[-1], nn.Parameter(value))


def _uniform(shape, fan_in: int) -> torch.Tensor:
    bound = 1.0 / math.sqrt(max(fan_in, 1))  # kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in), 1/sqrt(fan_in))
    return (torch.rand(shape) * 2 - 1) * bound


PRECISIONS = ("fp32", "bf16x3", "bf16", "fp16")


class Unet3D(nn.Module):
    PRECISIONS = PRECISIONS  # arithmetic modes of `precision` (sampling) and `train_precision`
    def __init__(
        self,
        dim,
        out_dim=None,
        dim_mults=(1, 2, 4, 8),
        channels=3,
        attn_heads=8,
        attn_dim_head=32,
        init_dim=None,
        init_kernel_size=7,
        use_sparse_linear_attn=True,
        resnet_groups=8,
        cond_bias=False,
        cond_attention="none",
        cond_attention_tokens=6,
        cond_att_GRU=False,
        use_temporal_attention_cond=False,
        cond_to_time="add",
        per_frame_cond=False,
        padding_mode="zeros",
    ):
        super().__init__()
        assert init_kernel_size % 2 == 1  # vddp.py:621
        if padding_mode not in ("zeros", "circular", "circular_1d"):
            raise ValueError(f"padding_mode {padding_mode!r}: 'zeros', 'circular' or 'circular_1d' (vddp.py:153-243)")
        if cond_to_time not in ("add", "concat"):
            raise ValueError("cond_to_time must be 'add' or 'concat' (vddp.py:786-789)")
        if not (isinstance(attn_dim_head, int) and 4 <= attn_dim_head <= 128 and attn_dim_head % 4 == 0):
            # (the reference takes any even width, vddp.py:612-615; model.yaml:16 ships 32.  The fused attention blocks are dim_head = 32 kernels; every other
            # width runs the projection + attention-core chain, whose kernels move a head slice in 16-byte pieces)
            raise NotImplementedError(f"attn_dim_head = {attn_dim_head!r}: the temporal attention kernels take multiples of 4 in 4 .. 128")
        self.channels = channels
        self.dim = dim
        self.time_dim = dim * 4
        self.cond_bias = cond_bias
        self.cond_attention = cond_attention if not per_frame_cond else "self-stacked"  # vddp.py:602
        self.cond_attention_tokens = cond_attention_tokens if not per_frame_cond else 11  # vddp.py:603
        if self.cond_attention not in ("none", "self-stacked", "cross-attention"):
            raise ValueError("cond_attention must be none, self-stacked or cross-attention")
        self.cond_att_GRU = cond_att_GRU
        self.cond_dim = self.time_dim
        self.use_temporal_attention_cond = use_temporal_attention_cond
        self.cond_to_time = cond_to_time
        self.per_frame_cond = per_frame_cond
        self.padding_mode = padding_mode
        self.attn_heads = attn_heads
        self.attn_dim_head = attn_dim_head
        self.use_sparse_linear_attn = use_sparse_linear_attn
        self.resnet_groups = resnet_groups
        self.init_kernel_size = init_kernel_size
        self.init_dim = init_dim if init_dim is not None else dim
        self.out_dim = out_dim if out_dim is not None else channels
        self.dim_mults = tuple(dim_mults)
        dims = [self.init_dim] + [dim * m for m in dim_mults]
        self.in_out: List[Tuple[int, int]] = list(zip(dims[:-1], dims[1:]))

        self._build_parameters()
        self._plans: Dict[tuple, "_plan.Plan"] = {}
        self._generation = 0
        self.register_load_state_dict_pre_hook(Unet3D._ckpt_pre_hook)
        self._register_state_dict_hook(Unet3D._ref_names_hook)
        self.static_weights = False  # set True to skip the per-call parameter-version scan (sampling loops)
        # matrix-core arithmetic of the contractions: "bf16x3" = split-bf16 operands, fp32 accumulate (~1e-5 relative, 5x the MFMA
        # rate); "fp32" = exact fp32 MFMA (1e-6).  `precision` governs inference / sampling, `train_precision` the training plans
        # (forward, data gradients and -- with use_x3_wgrad, the default -- the 3 x 3 / 1 x 1 weight gradients on the split-bf16 kernels, ~1e-5
        # relative per contraction; since round 6 the remaining geometries -- 4 x 4 stride-2, transposed phases, the stem -- too, on the tap-decoding form of the 1 x 1 kernel).  With the reference's l1 loss
        # the gradient is sign(pred - noise) / N, and a 1e-5 forward error flips enough signs to move parameter gradients by ~2e-3
        # (an order of magnitude inside the reference's own fp16-autocast deviation, tests/test_gpu_train.py).
        # "bf16" = the throughput mode of BASELINE.json configs[3]: one matrix pass on bf16-rounded operands in the 3 x 3 convolutions
        # and the fused attention blocks (~1e-2 relative on the denoiser output; tests/test_gpu_hires.py states and checks the tolerance).  As a
        # `train_precision` it is the reduced-precision training leg (the counterpart of the reference's fp16 autocast, main.py:34): single-pass forward,
        # data gradients, 3 x 3 / 1 x 1 / to_qkv weight gradients and recomputing attention backward over fp32 master weights and fp32-stored maps; its
        # gradients stay inside the reference's bf16-autocast deviation at both widths and inside its fp16-autocast deviation at dim 16 (1.9x its fp16 median at dim 64:
        # 8 against 11 operand mantissa bits; tests/test_gpu_train.py).
        # "fp16" (training; also accepted for sampling): the same single-pass kernels on IEEE-half operands -- the reference's OWN training arithmetic (main.py:34
        # mixed_precision='fp16': autocast runs every convolution / Linear / einsum on fp16 operands with fp32 accumulation), with the loss scaling Accelerate wraps
        # around it (GradScaler on the device, dp.py).  Gradient deviation from fp32 autograd inside the reference's own fp16-autocast figures at both widths.
        self.precision = "bf16x3"
        # (the default a drop-in user trains with is the arithmetic bench.py measures: split-bf16, fp32-class -- each contraction within ~1e-5 of fp32, l1 gradients an
        # order of magnitude inside the reference's own fp16-autocast deviation; "fp32" = every gradient within 1e-3 of the reference's fp32 autograd at twice the step
        # time; "fp16" / "bf16" = the single-pass legs)
        self.train_precision = "bf16x3"
        # precision "bf16" only: the feature maps of the two upper levels live in HBM as bf16 (one rounding per stored element; half the bytes of the
        # bandwidth-bound passes, half the plan memory).  False: the same single-pass arithmetic over fp32-stored maps.
        self.bf16_storage = True

    # ------------------------------------------------------------------ parameters (names = reference module tree)
    def _conv(self, name, cout, cin, k, bias=True):
        _attach(self, name + ".weight", _uniform((cout, cin, 1, k, k), cin * k * k))
        if bias:
            _attach(self, name + ".bias", _uniform((cout,), cin * k * k))

    def _linear(self, name, cout, cin, bias=True):
        _attach(self, name + ".weight", _uniform((cout, cin), cin))
        if bias:
            _attach(self, name + ".bias", _uniform((cout,), cin))

    def _resnet(self, name, cin, cout, temb: Optional[int]):
        if temb is not None:
            self._linear(name + ".mlp.1", cout * 2, temb)
        for blk, ci in (("block1", cin), ("block2", cout)):
            self._conv(f"{name}.{blk}.proj", cout, ci, 3)
            _attach(self, f"{name}.{blk}.norm.weight", torch.ones(cout))
            _attach(self, f"{name}.{blk}.norm.bias", torch.zeros(cout))
        if cin != cout:
            self._conv(name + ".res_conv", cout, cin, 1)

    def _softmax_attn(self, name, dim, rotary: bool, dim_head: int):
        hid = dim_head * self.attn_heads
        # (registration order = the reference's: PreNorm registers `fn` before `norm` (vddp.py:256-263) -- torch.optim.Adam's state is indexed by
        # position in parameters(), so the order is part of the checkpoint format)
        p = name + ".fn.fn"
        if rotary:
            rot = min(32, dim_head)
            freqs = 1.0 / (10000 ** (torch.arange(0, rot, 2).float() / rot))
            _attach(self, p + ".rotary_emb.freqs", freqs, buffer=True)
        self._linear(p + ".to_qkv", hid * 3, dim, bias=False)
        self._linear(p + ".to_q", hid, dim, bias=False)
        self._linear(p + ".to_k", hid, self.cond_dim, bias=False)
        self._linear(p + ".to_v", hid, self.cond_dim, bias=False)
        self._linear(p + ".to_out", dim, hid, bias=False)
        _attach(self, name + ".norm.gamma", torch.ones(1, dim, 1, 1, 1))

    def _linear_attn(self, name, dim):
        hid = 32 * self.attn_heads  # dim_head default (vddp.py:314, not forwarded at 679/700)
        p = name + ".fn"
        _attach(self, p + ".to_qkv.weight", _uniform((hid * 3, dim, 1, 1), dim))
        _attach(self, p + ".to_q.weight", _uniform((hid, dim, 1, 1), dim))
        self._linear(p + ".to_k", hid, self.cond_dim, bias=False)
        self._linear(p + ".to_v", hid, self.cond_dim, bias=False)
        _attach(self, p + ".to_out.weight", _uniform((dim, hid, 1, 1), hid))
        _attach(self, p + ".to_out.bias", _uniform((dim,), hid))
        _attach(self, name + ".norm.gamma", torch.ones(1, dim, 1, 1, 1))

    def _build_parameters(self):
        heads, td, cd = self.attn_heads, self.time_dim, self.cond_dim
        te = td + cd if self.cond_to_time == "concat" else cd  # ResnetBlock.mlp input width (vddp.py:670)
        _attach(self, "time_rel_pos_bias.relative_attention_bias.weight", torch.randn(32, heads))
        k = self.init_kernel_size
        self._conv("init_conv", self.init_dim, self.channels, k)
        self._softmax_attn("init_temporal_attn.fn", self.init_dim, True, self.attn_dim_head)
        self._linear("time_mlp.1", td, self.dim)
        self._linear("time_mlp.3", td, td)
        chain = [1, 16, 32, 64, 128, cd]
        for i, (ci, co) in enumerate(zip(chain[:-1], chain[1:])):
            _attach(self, f"sign_emb_CNN.emb_model.{2 * i}.weight", _uniform((co, ci, 4), ci * 4))
            _attach(self, f"sign_emb_CNN.emb_model.{2 * i}.bias", _uniform((co,), ci * 4))
        if self.cond_att_GRU:  # SignalEmbedding('GRU') = nn.GRU(1, cond_dim, num_layers=3) (vddp.py:546-549, 646-649): uniform(+-1/sqrt(H)) like torch
            for l in range(3):
                for nm_, shp in (("weight_ih", (3 * cd, 1 if l == 0 else cd)), ("weight_hh", (3 * cd, cd)), ("bias_ih", (3 * cd,)), ("bias_hh", (3 * cd,))):
                    _attach(self, f"sign_emb_GRU.emb_model.{nm_}_l{l}", _uniform(shp, cd))
        if self.per_frame_cond:
            self._linear("sign_emb", cd, 1)
            _attach(self, "cond_token_to_hidden.0.weight", torch.ones(cd))
            _attach(self, "cond_token_to_hidden.0.bias", torch.zeros(cd))
            self._linear("cond_token_to_hidden.1", cd, cd)
            self._linear("cond_token_to_hidden.3", td, cd)
        n_lvl = len(self.in_out)
        for i, (ci, co) in enumerate(self.in_out):
            self._resnet(f"downs.{i}.0", ci, co, te)
            self._resnet(f"downs.{i}.1", co, co, te)
            if self.use_sparse_linear_attn:
                self._linear_attn(f"downs.{i}.2.fn", co)
            self._softmax_attn(f"downs.{i}.3.fn", co, True, self.attn_dim_head)
            if i < n_lvl - 1:
                self._conv(f"downs.{i}.4", co, co, 4)
        self.add_module("ups", _Node())  # (the reference creates both ModuleLists before the middle blocks, vddp.py:653-654: parameters() order)
        mid = self.in_out[-1][1]
        self._resnet("mid_block1", mid, mid, te)
        self._softmax_attn("mid_spatial_attn.fn", mid, False, 32)
        self._softmax_attn("mid_temporal_attn.fn", mid, True, self.attn_dim_head)
        self._resnet("mid_block2", mid, mid, te)
        for i, (ci, co) in enumerate(reversed(self.in_out)):
            self._resnet(f"ups.{i}.0", co * 2, ci, te)
            self._resnet(f"ups.{i}.1", ci, ci, te)
            if self.use_sparse_linear_attn:
                self._linear_attn(f"ups.{i}.2.fn", ci)
            self._softmax_attn(f"ups.{i}.3.fn", ci, True, self.attn_dim_head)
            if i < n_lvl - 1:
                self._conv(f"ups.{i}.4", ci, ci, 4)  # ConvTranspose3d weight is (in, out, 1, 4, 4); in == out here
        self._resnet("final_conv.0", self.dim * 2, self.dim, None)
        self._conv("final_conv.1", self.out_dim, self.dim, 1)
        _attach(self, "null_text_token", torch.randn(1, self.cond_attention_tokens, cd))
        _attach(self, "null_text_hidden", torch.randn(1, td))

    # The periodic-padding variants wrap their convolutions in helper modules (vddp.py:153-243: Circular_1d_Conv3d.conv, CircularUpsample /
    # Circular_1d_Upsample.conv_transpose), which moves the parameters one level down in the reference's state_dict.  The parameter tree
    # here keeps ONE set of names (the 'zeros' ones, which the launch plans use); state_dict() / load_state_dict() translate.
    def _ref_key(self, k: str) -> str:
        import re
        if self.padding_mode in ("circular", "circular_1d"):
            k = re.sub(r"^(ups\.\d+\.4)\.(weight|bias)$", r"\1.conv_transpose.\2", k)
        if self.padding_mode == "circular_1d":
            k = re.sub(r"^(.*\.block[12]\.proj|init_conv|downs\.\d+\.4)\.(weight|bias)$", r"\1.conv.\2", k)
        return k

    def _own_key(self, k: str) -> str:
        import re
        if self.padding_mode in ("circular", "circular_1d"):
            k = re.sub(r"^(ups\.\d+\.4)\.conv_transpose\.(weight|bias)$", r"\1.\2", k)
        if self.padding_mode == "circular_1d":
            k = re.sub(r"^(.*\.block[12]\.proj|init_conv|downs\.\d+\.4)\.conv\.(weight|bias)$", r"\1.\2", k)
        return k

    @staticmethod
    def _ref_names_hook(module, state_dict, prefix, local_metadata):
        if module.padding_mode == "zeros":
            return
        items = list(state_dict.items())
        state_dict.clear()
        for k, v in items:
            if k.startswith(prefix):
                k = prefix + module._ref_key(k[len(prefix):])
            state_dict[k] = v

    @staticmethod
    def _ckpt_pre_hook(module, state_dict, prefix, local_metadata, strict, missing_keys, unexpected_keys, error_msgs):
        """Checkpoint tolerances, applied wherever this module sits in the tree being loaded -- the reference loads at the
        GaussianDiffusion level (Trainer.load -> model.load_state_dict / ema_model.load_state_dict, vddp.py:1577,1587), where
        nn.Module recursion reaches children through _load_from_state_dict only:
        * a DDP / Accelerate `module.` level in front of the denoiser's keys is stripped;
        * `...rotary_emb.*` entries are optional both ways: whether `freqs` (and newer versions' cached tables) appear in a
          checkpoint depends on the rotary_embedding_torch version -- missing ones keep the constructor's table,
          extra ones are dropped."""
        wrapped = prefix + "module."
        for k in [k for k in state_dict if k.startswith(wrapped)]:
            state_dict[prefix + k[len(wrapped):]] = state_dict.pop(k)
        if module.padding_mode != "zeros":  # the reference's names of the wrapped convolutions -> this tree's
            for k in [k for k in state_dict if k.startswith(prefix)]:
                own_k = prefix + module._own_key(k[len(prefix):])
                if own_k != k:
                    state_dict[own_k] = state_dict.pop(k)
        own = {prefix + k: v for k, v in module.state_dict().items() if ".rotary_emb." in k}
        for k in [k for k in state_dict if k.startswith(prefix) and ".rotary_emb." in k and k not in own]:
            del state_dict[k]
        for k, v in own.items():
            state_dict.setdefault(k, v)
        module.bump_generation()

    def bump_generation(self) -> None:
        """Parameters were (or are about to be) rewritten through a path autograd's version counters cannot see -- raw-pointer
        HIP launches (vmm_adam_step / vmm_ema_step), load_state_dict, `.data` re-homing: every cached plan re-packs its operand
        layouts on next use."""
        self._generation += 1

    # ------------------------------------------------------------------ execution
    def __getstate__(self):
        st = self.__dict__.copy()
        st["_plans"] = {}  # plans hold ctypes descriptors and device arenas: rebuilt on demand
        return st

    def _params_flat(self) -> Dict[str, torch.Tensor]:
        d = dict(self.named_parameters())
        d.update(dict(self.named_buffers()))
        return d

    def _version_key(self):
        return (self._generation,) + tuple((p.data_ptr(), p._version) for p in self.parameters())

    def refresh_plans(self) -> None:
        """Bring the packed weights of every cached inference plan up to date with the live parameters (one pack launch per stale
        plan).  Sampling loops call this once and then run with `static_weights` (no per-step version scan)."""
        ver = None
        for pl in self._plans.values():
            if pl.training:
                continue  # training plans re-pack at every step anyway (dp.py / get_plan)
            ver = self._version_key() if ver is None else ver
            if pl.weights_version != ver:
                pl.refresh_weights(self._params_flat())
                pl.weights_version = ver

    def get_plan(self, B: int, T: int, H: int, W: int, cond_len: int, device, *, training: bool = False, mirrored: bool = False,
                 focus: bool = False) -> "_plan.Plan":
        """mirrored: the caller feeds x[B/2:] == x[:B/2] (guidance: both branches in one batch) -- the conditioning-free prefix is shared.
        focus: a plan with a focus_present_mask slot (vddp.py:431; the temporal attentions in their unfused form)."""
        # (the measurement switches that change a plan's structure are part of its identity: flipping VMM_DISABLE between calls must not
        # hand back a plan built under the other setting)
        key = (B, T, H, W, cond_len, str(device), training, self.train_precision if training else self.precision, bool(mirrored),
               os.environ.get("VMM_DISABLE", ""), bool(getattr(self, "use_x3_wgrad", True)), bool(getattr(self, "use_x3_wgrad_generic", False)), bool(focus),
               bool(getattr(self, "bf16_storage", True)), os.environ.get("VMM_A16_OPS", "all"))
        pl = self._plans.get(key)
        if pl is None:
            pl = _plan.build_plan(self, B, T, H, W, cond_len, device, training=training, mirrored=mirrored, focus=focus)
            self._plans[key] = pl
            pl.weights_version = None
        if not (self.static_weights and pl.weights_version is not None):
            ver = self._version_key()
            if pl.weights_version != ver:
                pl.refresh_weights(self._params_flat())
                pl.weights_version = ver
        return pl

    def _mask(self, batch: int, prob: float, device) -> torch.Tensor:
        """prob_mask_like (vddp.py:55-61) as uint8."""
        if prob == 1:
            return torch.ones(batch, dtype=torch.uint8, device=device)
        if prob == 0:
            return torch.zeros(batch, dtype=torch.uint8, device=device)
        return (torch.zeros(batch, device=device).float().uniform_(0, 1) < prob).to(torch.uint8)

    def _check_inputs(self, x, cond, focus_present_mask, prob_focus_present):
        if x.dim() != 5 or x.shape[1] != self.channels:
            raise ValueError(f"expected x of shape (b, {self.channels}, f, h, w), got {tuple(x.shape)}")
        if self.init_dim != self.dim:
            # the reference constructs such a model and fails in its first forward: final_conv = block_klass(dim * 2, dim) reads cat(x, r), which has
            # 2 * init_dim channels (vddp.py:706, 820) -- a RuntimeError from the convolution's shape check there, the same class here
            raise RuntimeError(f"init_dim = {self.init_dim} but dim = {self.dim}: final_conv expects {2 * self.dim} input channels and receives "
                               f"{2 * self.init_dim} (vddp.py:706, 820: the reference's forward fails on the same configuration)")
        if not x.is_cuda:
            raise RuntimeError("videometamaterials_amd.Unet3D runs on an MI355X only; move the model and inputs to 'cuda'")
        if focus_present_mask is not None and tuple(focus_present_mask.shape) != (x.shape[0],):
            raise ValueError(f"focus_present_mask must have shape ({x.shape[0]},), got {tuple(focus_present_mask.shape)}")
        if cond is None:
            raise ValueError("cond is required (the reference dereferences it unconditionally, vddp.py:753,761)")
        if self.per_frame_cond and x.shape[2] != 11:
            raise ValueError("per_frame_cond is hard-wired to 11 frames (vddp.py:603)")

    def _focus(self, batch: int, focus_present_mask, prob_focus_present: float, device):
        """vddp.py:740: the caller's mask, else prob_mask_like((batch,), prob_focus_present) -- drawn BEFORE the classifier-free-guidance mask, like
        the reference.  None when no sample focuses on the present (the case of every shipped config: the regular plans run)."""
        if focus_present_mask is None:
            if prob_focus_present == 0:
                return None
            f = self._mask(batch, prob_focus_present, device)
        else:
            f = focus_present_mask.to(device=device).reshape(batch).ne(0).to(torch.uint8)
        return f if bool(f.any()) else None

    def forward(self, x, time, cond=None, null_cond_prob=0.0, focus_present_mask=None, prob_focus_present=0.0):
        self._check_inputs(x, cond, focus_present_mask, prob_focus_present)
        B, _, T, H, W = x.shape
        focus = self._focus(B, focus_present_mask, prob_focus_present, x.device)
        mask = self._mask(B, null_cond_prob, x.device)
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            from .autograd import unet_forward_with_grad
            return unet_forward_with_grad(self, x, time, cond, mask, focus)
        pl = self.get_plan(B, T, H, W, cond.shape[-1], x.device, focus=focus is not None)
        return pl.run(x, time, cond, mask, focus).clone()

    def forward_with_guidance_scale(self, *args, **kwargs):
        """vddp.py:715-728.  Both branches run as ONE batch of 2B (legal: every op is per-sample)."""
        guidance_scale = kwargs.pop("guidance_scale", 5.0)
        if guidance_scale == 1:
            return self.forward(*args, null_cond_prob=0.0, **kwargs)
        names = ("x", "time", "cond")
        bound = dict(zip(names, args))
        bound.update(kwargs)
        x, time, cond = bound["x"], bound["time"], bound.get("cond")
        fpm, pfp = bound.get("focus_present_mask"), bound.get("prob_focus_present", 0.0)
        self._check_inputs(x, cond, fpm, pfp)
        B = x.shape[0]
        # (the reference calls forward twice: with a probability strictly between 0 and 1 each call draws its own mask)
        f_c, f_n = self._focus(B, fpm, pfp, x.device), self._focus(B, fpm, pfp, x.device)
        focus = None
        if f_c is not None or f_n is not None:
            z = torch.zeros(B, dtype=torch.uint8, device=x.device)
            focus = torch.cat([z if f_c is None else f_c, z if f_n is None else f_n])
        eps_c, eps_n = self.guided_pair(x, time, cond, focus)
        return _plan.cfg_combine(eps_c, eps_n, float(guidance_scale))

    @torch.no_grad()
    def guided_pair(self, x, time, cond, focus=None):
        """(eps_cond, eps_null) views into the plan's static output (valid until the next call).  focus: (2B,) uint8 or None."""
        B, _, T, H, W = x.shape
        pl = self.get_plan(2 * B, T, H, W, cond.shape[-1], x.device, mirrored=True, focus=focus is not None)
        mask = torch.cat([torch.zeros(B, dtype=torch.uint8, device=x.device), torch.ones(B, dtype=torch.uint8, device=x.device)])
        out = pl.run(torch.cat([x, x]), torch.cat([time, time]), torch.cat([cond, cond]), mask, focus)
        return out[:B], out[B:]
