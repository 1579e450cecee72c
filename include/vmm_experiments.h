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
#ifdef __cplusplus
}
#endif
#endif
