"""ctypes binding of libvmm_hip.so (C ABI declared in include/vmm_kernels.h).

The product path has NO fallback: if the HIP library is missing or a launch fails the
call raises.  Build it with ``python -c "import __graft_entry__ as g; g.build()"`` or
``python -m videometamaterials_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libvmm_hip.so")

c_f32p = C.c_void_p  # device pointers travel as integers
c_i32 = C.c_int32
c_i64 = C.c_int64
c_f32 = C.c_float
c_ptr = C.c_void_p


class ConvDesc(C.Structure):
    """vmm_conv_desc (include/vmm_kernels.h)."""

    _fields_ = [
        ("a1", c_ptr), ("a2", c_ptr),
        ("C1", c_i32), ("C2", c_i32), ("lda1", c_i32), ("lda2", c_i32),
        ("w", c_ptr), ("bias", c_ptr),
        ("res", c_ptr), ("ldres", c_i32),
        ("out", c_ptr), ("ldo", c_i32),
        ("nimg", c_i32), ("Hin", c_i32), ("Win", c_i32),
        ("Hv", c_i32), ("Wv", c_i32), ("stride", c_i32),
        ("KH", c_i32), ("KW", c_i32), ("off_h", c_i32), ("off_w", c_i32), ("sgn_h", c_i32), ("sgn_w", c_i32),
        ("Hout", c_i32), ("Wout", c_i32), ("oscale", c_i32), ("ooh", c_i32), ("oow", c_i32),
        ("Cout", c_i32),
        ("rot_tab", c_ptr),
        ("rot_T", c_i32), ("rot_HW", c_i32), ("rot_ncols", c_i32), ("rot_dh", c_i32),
        ("q_scale", c_f32), ("q_ncols", c_i32),
        ("a_mode", c_i32), ("a_coef", c_ptr), ("a_imgs_per_sample", c_i32),
        ("split_tickets", c_ptr), ("n_tickets", c_i32),
        ("gn_part", c_ptr), ("gn_groups", c_i32),
        ("wrap_h", c_i32), ("wrap_w", c_i32), ("a_img_mod", c_i32),
        ("act_bf16", c_i32),
        ("sk_work", c_ptr), ("sk_slots", c_i32),
        ("defer_reduce", c_i32),
    ]


class ReduceJob(C.Structure):
    """vmm_reduce_job (include/vmm_kernels.h)."""

    _fields_ = [
        ("part", c_ptr), ("out", c_ptr), ("bias_part", c_ptr), ("dbias", c_ptr),
        ("kind", c_i32), ("nz", c_i32), ("tiles_x", c_i32), ("tiles_y", c_i32), ("Cin", c_i32), ("Cout", c_i32), ("ld", c_i32), ("n_main", c_i32),
        ("gx", c_i32), ("wgs", c_i32), ("wg0", c_i32), ("pad_", c_i32),
    ]


class DenseJob(C.Structure):
    """vmm_dense_job (include/vmm_kernels.h)."""

    _fields_ = [
        ("x", c_ptr), ("w", c_ptr), ("b", c_ptr), ("add", c_ptr), ("y", c_ptr),
        ("rows", c_i32), ("K", c_i32), ("N", c_i32), ("ldx", c_i32), ("ldy", c_i32), ("ldadd", c_i32),
        ("act_in", c_i32), ("act_out", c_i32),
    ]


class PackJob(C.Structure):
    """vmm_pack_job (include/vmm_kernels.h)."""

    _fields_ = [("torch_w", c_ptr), ("packed", c_ptr)] + [(n, c_i32) for n in (
        "TH", "TW", "C", "Cp", "N", "sn", "sc", "sh", "sw", "h0", "hs", "w0", "ws", "accumulate", "fmt")]


class DenseBwdJob(C.Structure):
    """vmm_dense_bwd_job (include/vmm_kernels.h)."""

    _fields_ = [("x", c_ptr), ("w", c_ptr), ("b", c_ptr), ("dy", c_ptr), ("dx", c_ptr), ("dw", c_ptr), ("db", c_ptr)] + [(n, c_i32) for n in (
        "rows", "K", "N", "ldx", "lddy", "lddx", "act_in", "act_out", "accumulate")]


class AttnBlockBwd(C.Structure):
    """vmm_attn_block_bwd (include/vmm_kernels.h)."""

    _fields_ = [
        ("x", c_ptr), ("ldx", c_i32),
        ("gamma", c_ptr),
        ("wqkv_frag", c_ptr),
        ("wout_t_frag", c_ptr),
        ("ek", c_ptr), ("ev", c_ptr), ("ntok", c_i32),
        ("bias", c_ptr), ("bias_on_cond", c_i32),
        ("rot_tab", c_ptr),
        ("fwd_workspace", c_ptr),
        ("dout", c_ptr), ("lddo", c_i32),
        ("dqkv", c_ptr), ("lddqkv", c_i32),
        ("ln_stats", c_ptr),
        ("dwout_packed", c_ptr),
        ("dbout", c_ptr),
        ("dbias", c_ptr),
        ("dek", c_ptr), ("dev", c_ptr),
        ("workspace", c_ptr),
        ("B", c_i32), ("T", c_i32), ("HW", c_i32), ("C", c_i32), ("heads", c_i32),
        ("q_scale", c_f32), ("eps", c_f32),
    ]


class OptimJob(C.Structure):
    """vmm_optim_job (include/vmm_kernels.h)."""

    _fields_ = [("p", c_ptr), ("g", c_ptr), ("m", c_ptr), ("v", c_ptr), ("n", c_i64)]


# name -> argtypes (restype is always int); must list EVERY symbol include/vmm_kernels.h declares
SIGNATURES = {
    "vmm_conv_igemm_f32": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv_igemm_bf16x3": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv_igemm_bf16x3_batched": [C.POINTER(ConvDesc), c_i32, c_ptr],
    "vmm_conv3x3_bf16x3": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv3x3_bf16": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv3x3_f32": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv3x3_fuses_gn": [C.POINTER(ConvDesc)],
    "vmm_conv3x3_accepts": [C.POINTER(ConvDesc)],
    "vmm_conv_wgrad_f32": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_conv3x3_wgrad_f32": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_sum_partials": [c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_conv3x3_wgrad_reduce_job": [C.POINTER(ConvDesc), c_i32, c_ptr, c_ptr, c_ptr, C.POINTER(ReduceJob)],
    "vmm_conv1x1_wgrad_reduce_job": [C.POINTER(ConvDesc), c_i32, c_ptr, c_ptr, c_ptr, C.POINTER(ReduceJob)],
    "vmm_conv_wgrad_tap_reduce_job": [C.POINTER(ConvDesc), c_i32, c_ptr, c_ptr, c_ptr, C.POINTER(ReduceJob)],
    "vmm_conv_wgrad_tap_workspace": [C.POINTER(ConvDesc), c_i32],
    "vmm_conv_wgrad_tap_bf16x3": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv_wgrad_tap_bf16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv_wgrad_tap_fp16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_reduce_batch": [c_ptr, c_i32, c_i32, c_ptr],
    "vmm_conv3x3_wgrad_bf16x3": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv3x3_wgrad_bf16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv3x3_wgrad_bf16x3_workspace": [C.POINTER(ConvDesc), c_i32],
    "vmm_conv1x1_wgrad_bf16x3": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv1x1_wgrad_bf16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv1x1_wgrad_bf16x3_workspace": [C.POINTER(ConvDesc), c_i32],
    "vmm_conv1x1_wgrad_bf16x3_ln": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv1x1_wgrad_bf16_ln": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_qkv_bwd_workspace": [c_i64, c_i32, c_i32],
    "vmm_qkv_bwd_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_qkv_bwd_bf16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_qkv_bwd_ln_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_qkv_bwd_ln_bf16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_proj_bf16x3_ln_stats": [C.POINTER(ConvDesc), c_ptr, c_f32, c_ptr, c_ptr],
    "vmm_conv_wgrad_bf16x3": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_colsum_accumulate": [c_ptr, c_i32, c_i64, c_i32, c_ptr, c_ptr],
    "vmm_pack_weights": [c_ptr, c_i32, c_i32, c_i32, c_ptr],
    "vmm_groupnorm_bwd_scratch": [c_i32, c_i32, c_i32, c_i32],
    "vmm_groupnorm_bwd": [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_i32, c_i32,
                          c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_channel_layernorm_bwd": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_ptr, c_i64, c_i32, c_f32, c_ptr, c_ptr],
    "vmm_attention_bwd": [c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr,
                          c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_temporal_attention_bwd": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr,
                                   c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_attention_bwd_scratch": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_temporal_block_bwd_workspace": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_temporal_block_bwd_bf16x3": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_temporal_block_bwd_bf16": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_linattn_block_bwd_workspace": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_linattn_block_bwd_bf16x3": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_linattn_block_bwd_bf16": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_linattn_bwd": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_linattn_apply_mfma": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_f32, c_ptr],
    "vmm_linattn_bwd_rows_mfma": [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_f32, c_ptr],
    "vmm_dense_bwd_batched": [c_ptr, c_i32, c_i32, c_i32, c_ptr],
    "vmm_cond_tokens_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_rows_layernorm_affine_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_ptr],
    "vmm_select_add_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    "vmm_select_concat_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    "vmm_relpos_bias_bwd": [c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_tokens_from_hidden_bwd": [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_conv1d_k4s2_silu_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_pointwise_to_ncthw_bwd": [c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_stem_conv_dgrad": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_loss_grad": [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_adam_step": [c_ptr, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_i32, c_f32, c_ptr],
    "vmm_ema_step": [c_ptr, c_i32, c_i64, c_f32, c_i32, c_ptr],
    "vmm_groupnorm_stats": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_groupnorm_stats_slots": [c_i32, c_i32, c_i32],
    "vmm_groupnorm_stats_partials": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_groupnorm_coef": [c_ptr, c_i64, c_f32, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr],
    "vmm_affine_silu_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i64, c_i32, c_i32, c_ptr],
    "vmm_affine_silu_pointwise_to_ncthw_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_stem_conv_bf16x3_a16": [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_temporal_attention_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_temporal_block_bf16_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_i32, c_f32, c_f32, c_ptr],
    "vmm_linattn_block_bf16_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_f32, c_ptr],
    "vmm_convert_act": [c_ptr, c_i32, c_ptr, c_i32, c_i64, c_ptr],
    "vmm_affine_silu": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i64, c_i32, c_i32, c_ptr],
    "vmm_affine_silu_pointwise_to_ncthw": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_channel_layernorm": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i64, c_i32, c_f32, c_ptr],
    "vmm_temporal_attention": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_temporal_attention_staged": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_stem_conv_bf16x3": [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_conv_s2_supported": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_conv_s2_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_conv_s2_acc_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32, c_ptr],
    "vmm_conv_s2_acc_bf16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32, c_ptr],
    "vmm_conv_s2_acc_bf16_a16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_temporal_block_supported": [c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_temporal_block_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_i32, c_f32, c_f32, c_ptr],
    "vmm_temporal_block_bf16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_i32, c_f32, c_f32, c_ptr],
    "vmm_spatial_attention": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_spatial_attention_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_proj_bf16x3": [C.POINTER(ConvDesc), c_ptr, c_f32, c_ptr],
    "vmm_proj_bf16": [C.POINTER(ConvDesc), c_ptr, c_f32, c_ptr],
    "vmm_proj_bf16_res_silu": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr],
    "vmm_proj_bf16x3_res_silu": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr],
    "vmm_proj_f32": [C.POINTER(ConvDesc), c_ptr, c_f32, c_ptr],
    "vmm_proj_narrow_bf16x3": [C.POINTER(ConvDesc), c_ptr],
    "vmm_proj_narrow_bf16x3_res_silu": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr],
    "vmm_copy2": [c_ptr, c_ptr, c_ptr, c_i64, c_ptr],
    "vmm_extract_geometry": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_f32, c_ptr, c_ptr],
    "vmm_fields_to_samples": [c_ptr, c_i32, c_i32, c_i64, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_temporal_core_bf16x3": [c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_ptr],
    "vmm_linattn_block_workspace": [c_i32, c_i32, c_i32],
    "vmm_linattn_block_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_f32, c_ptr],
    "vmm_linattn_block_bf16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_f32, c_ptr],
    "vmm_linattn_context": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_linattn_context_bf16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_linattn_partial_bf16x3": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_linattn_apply": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_cross_attention": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_linattn_cross_context": [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_cross_attention_bwd_scratch": [c_i32, c_i32, c_i32, c_i32, c_i32, c_i32],
    "vmm_cross_attention_bwd": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_f32, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32,
                                c_i32, c_i32, c_i32, c_ptr],
    "vmm_linattn_cross_bwd": [c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32,
                              c_i32, c_ptr],
    # the fp16-operand twins of the single-pass entry points (train_precision = "fp16") and the device-side GradScaler
    "vmm_conv3x3_fp16": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv_s2_acc_fp16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32, c_ptr],
    "vmm_linattn_block_fp16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_f32, c_ptr],
    "vmm_temporal_block_fp16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_i32, c_f32, c_f32, c_ptr],
    "vmm_conv3x3_wgrad_fp16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv1x1_wgrad_fp16": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_conv1x1_wgrad_fp16_ln": [C.POINTER(ConvDesc), c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr],
    "vmm_qkv_bwd_fp16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_qkv_bwd_ln_fp16": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_temporal_block_bwd_fp16": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_dqkv_widen_fp16": [c_ptr, c_ptr, c_i64, c_ptr],
    "vmm_conv_igemm_bf16": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv_igemm_fp16": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv_igemm_bf16_batched": [C.POINTER(ConvDesc), c_i32, c_ptr],
    "vmm_conv_igemm_fp16_batched": [C.POINTER(ConvDesc), c_i32, c_ptr],
    "vmm_dqkv_widen_bf16": [c_ptr, c_ptr, c_i64, c_ptr],
    "vmm_linattn_block_bwd_fp16": [C.POINTER(AttnBlockBwd), c_ptr],
    "vmm_scaler_init": [c_ptr, c_f32, c_ptr],
    "vmm_grad_nonfinite": [c_ptr, c_i64, c_ptr, c_ptr],
    "vmm_adam_step_scaled": [c_ptr, c_i32, c_i64, c_f32, c_f32, c_f32, c_f32, c_f32, c_ptr, c_ptr],
    "vmm_scaler_update": [c_ptr, c_f32, c_f32, c_i32, c_ptr],
    "vmm_dense_batched": [c_ptr, c_i32, c_i32, c_ptr],
    "vmm_sinusoidal_embed": [c_ptr, c_i32, c_i32, c_f32, c_ptr, c_ptr],
    "vmm_cond_tokens": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_rows_layernorm_affine": [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_f32, c_ptr],
    "vmm_select_add": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    "vmm_select_concat": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_ptr],
    "vmm_gru_recurrent": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr],
    "vmm_tokens_select": [c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_gru_recurrent_bwd": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr],
    "vmm_tokens_select_bwd": [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_focus_rows": [c_i32, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_i32, c_i32, c_i32, c_ptr],
    "vmm_rotary_rows": [c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_relpos_bias": [c_ptr, c_ptr, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_conv1d_k4s2_silu": [c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr],
    "vmm_tokens_from_hidden": [c_ptr, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_ncthw_to_rows": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_i32, c_ptr],
    "vmm_rows_to_ncthw": [c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_pointwise_to_ncthw": [c_ptr, c_i32, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_ptr, c_ptr],
    "vmm_q_sample": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i64, c_ptr],
    "vmm_predict_x0": [c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i64, c_ptr],
    "vmm_quantile_rows": [c_ptr, c_i32, c_i64, c_i64, c_f32, c_f32, c_ptr, c_ptr, c_ptr],
    "vmm_posterior_step": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i64, c_ptr],
    "vmm_posterior_step_rng": [c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_i64, c_ptr, c_ptr],
    "vmm_ddim_step_rng": [c_ptr, c_ptr, c_ptr, c_f32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_i64, c_ptr, c_ptr],
    "vmm_step_inputs": [c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i64, c_ptr],
    "vmm_loss_reduce": [c_ptr, c_ptr, c_i64, c_i32, c_ptr, c_ptr, c_ptr],
    "vmm_cfg_combine": [c_ptr, c_ptr, c_f32, c_ptr, c_i64, c_ptr],
    "vmm_lincomb": [c_ptr, c_ptr, c_ptr, c_f32, c_f32, c_f32, c_f32, c_ptr, c_i64, c_ptr],
}

# include/vmm_experiments.h: entry points of libvmm_hip_exp.so only (VMM_EXPERIMENTS=1 build; bound when the loaded library exports them)
EXPERIMENT_SIGNATURES = {
    "vmm_conv3x3_wino_bf16x3": [C.POINTER(ConvDesc), c_ptr],
    "vmm_conv3x3_wino_fuses_gn": [C.POINTER(ConvDesc)],
    "vmm_conv3x3_wino_accepts": [C.POINTER(ConvDesc)],
    "vmm_temporal_block_f16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32,
                                  c_i32, c_f32, c_f32, c_ptr],
    "vmm_linattn_block_f16x3": [c_ptr, c_i32, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_ptr, c_i32, c_ptr, c_ptr, c_i32, c_i32, c_i32, c_i32, c_i32, c_i32,
                                 c_f32, c_ptr],
}

# include/vmm_dp.h: the data-parallel engine (RCCL bound at run time; nothing here runs unless a DP engine is created)
c_ptrp = C.POINTER(c_ptr)
DP_SIGNATURES = {
    "vmm_dp_get_unique_id": [C.c_char_p, c_ptr],
    "vmm_dp_init": [c_ptrp, C.c_char_p, c_i32, c_i32, c_ptr, c_i32],
    "vmm_dp_register_buckets": [c_ptr, C.POINTER(c_ptr), C.POINTER(c_i64), c_i32],
    "vmm_dp_allreduce_bucket_async": [c_ptr, c_i32, c_ptr],
    "vmm_dp_wait_all": [c_ptr, c_ptr],
    "vmm_dp_set_timing": [c_ptr, c_i32],
    "vmm_dp_window_mark": [c_ptr, c_i32, c_ptr],
    "vmm_dp_timing": [c_ptr, C.POINTER(c_f32)],
    "vmm_dp_bucket_timing": [c_ptr, C.POINTER(c_f32), c_i32],
    "vmm_dp_allreduce": [c_ptr, c_ptr, c_i64, c_i32, c_i32, c_ptr],
    "vmm_dp_broadcast": [c_ptr, c_ptr, c_i64, c_i32, c_ptr],
    "vmm_dp_all_gather": [c_ptr, c_ptr, c_ptr, c_i64, c_ptr],
    "vmm_dp_rank": [c_ptr],
    "vmm_dp_world": [c_ptr],
    "vmm_dp_rccl_version": [c_ptr],
    "vmm_dp_last_error": [c_ptr],
    "vmm_dp_finalize": [c_ptr],
}

RESTYPES = {"vmm_dp_last_error": C.c_char_p, "vmm_attention_bwd_scratch": c_i64, "vmm_cross_attention_bwd_scratch": c_i64, "vmm_linattn_block_workspace": c_i64, "vmm_conv3x3_wgrad_bf16x3_workspace": c_i64, "vmm_conv1x1_wgrad_bf16x3_workspace": c_i64, "vmm_conv_wgrad_tap_workspace": c_i64, "vmm_qkv_bwd_workspace": c_i64, "vmm_temporal_block_bwd_workspace": c_i64, "vmm_linattn_block_bwd_workspace": c_i64, "vmm_groupnorm_bwd_scratch": c_i64}  # everything else returns int (0 = ok)

_lib = None


class NativeError(RuntimeError):
    pass


def lib() -> C.CDLL:
    """Load (once) and type the shared library; raises if it has not been built."""
    global _lib
    if _lib is None:
        path = os.environ.get("VMM_LIB_PATH", LIB_PATH)  # (VMM_LIB_PATH: A/B measurements of two builds on one box; read once)
        if not os.path.exists(path):
            raise NativeError(
                f"{path} is missing: the HIP extension must be built (python -m videometamaterials_amd.build); "
                "there is no CPU/PyTorch fallback for the hot path"
            )
        handle = C.CDLL(path)
        for name, argtypes in list(SIGNATURES.items()) + list(DP_SIGNATURES.items()):
            fn = getattr(handle, name)  # AttributeError if the symbol is not exported
            fn.argtypes = argtypes
            fn.restype = RESTYPES.get(name, C.c_int)
        for name, argtypes in EXPERIMENT_SIGNATURES.items():
            fn = getattr(handle, name, None)
            if fn is not None:
                fn.argtypes = argtypes
                fn.restype = C.c_int
        _lib = handle
    return _lib


def experiments_built() -> bool:
    """True when the loaded library is the experiments build (libvmm_hip_exp.so via VMM_LIB_PATH)."""
    return all(hasattr(lib(), name) for name in EXPERIMENT_SIGNATURES)


def check(code: int, what: str) -> None:
    if code != 0:
        raise NativeError(f"{what} failed with code {code} (hipError_t if > 0, rejected argument if < 0)")
