/* Experiments: entry points that exist only in libvmm_hip_exp.so (VMM_EXPERIMENTS=1 python -m videometamaterials_amd.build).  Each was built to the
 * same parity bar as the product kernels, measured inside the captured step, and lost; no plan launches them (LABNOTES 8.4, 7.3 / 7.8, 10.1). */
#ifndef VMM_EXPERIMENTS_H
#define VMM_EXPERIMENTS_H
#include "vmm_kernels.h"
#ifdef __cplusplus
extern "C" {
#endif
/* The same convolution as Winograd F(2x2, 3x3) on the split-bf16 matrix cores (conv3x3_wino.hip): 16 transform-domain products per 2 x 2 output
 * tile and (cin, cout) instead of 36, input / output transforms in fp32 (exact constants), operands split after the transform: fp32-class
 * results (7e-6 relative against 4.7e-6 for the direct kernel on the same data).  d->w = fmt-8 output of vmm_pack_weights.  Envelope: 3 x 3 /
 * stride 1 / zero padding 1, even H and W whose tile grid (H / 2 x W / 2) divides into blocks of 32..64 tiles with an input patch of at most 324
 * pixels, C1 / C2 multiples of 16, Cout a multiple of 64; fused operand transform (a_mode 1), a_img_mod, bias, residual as in
 * vmm_conv3x3_bf16x3; GroupNorm partial sums in d->gn_part with n = vmm_conv3x3_wino_fuses_gn(d) slots per (sample, group) (0 = not produced).
 * Returns 1 (nothing launched) outside the envelope; vmm_conv3x3_wino_accepts is the host-only query for that. */
int vmm_conv3x3_wino_bf16x3(const vmm_conv_desc* d, vmm_stream_t stream);
int vmm_conv3x3_wino_fuses_gn(const vmm_conv_desc* d);
int vmm_conv3x3_wino_accepts(const vmm_conv_desc* d);
/* The sampler's fused attention blocks on IEEE-half hi | lo operands (round 6, LABNOTES 11.2): the same THREE passes per product as the `_bf16x3` entry points (fp32-class: x = hi + lo to
 * 2^-22 for O(1) operands, an absolute floor of 3e-8 from the half denormals), with a split of four vector instructions per pair instead of six (v_fma_mix_f32 subtracts hi
 * straight from its packed half; there is no such instruction for bf16) -- these kernels are bound by vector work in the matrix shadow.  Same arguments; weights =
 * vmm_pack_weights fmt 2 | 32 / 3 | 32 (fp16 hi plane, fp16 lo plane).  Measured (tools/bench_attn_split.py): 20x closer to the fp64 block than split-bf16 (1.2e-7
 * against 2.4e-6), 0.4 % faster at the sites with conditioning tokens and 4 % at the one without -- not enough to give the sampler two operand types. */
int vmm_temporal_block_f16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed,
                              const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                              const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                              float q_scale, float eps, vmm_stream_t stream);
int vmm_linattn_block_f16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                             const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                             int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream);
#ifdef __cplusplus
}
#endif
#endif
