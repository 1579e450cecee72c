/* C ABI of libvmm_hip.so -- the MI355X (gfx950) hot path of VideoMetamaterials.
 *
 * Every entry point is a pure launcher: no allocation, no synchronisation, no global
 * state; work is enqueued on the hipStream_t passed in (the host passes PyTorch's
 * current stream so calls stay ordered with the surrounding torch ops).  Return value:
 * 0 on success, otherwise the hipError_t of the failed launch / a negative value for a
 * rejected argument.  All tensors are fp32 unless noted; "rows" are (b,t,h,w) positions
 * of the frame-major channels-last working layout, `ld*` are row strides in floats.
 *
 * Reference interface each entry point replaces (vddp.py =
 * /root/reference/denoising_diffusion_pytorch/video_denoising_diffusion_pytorch.py):
 * the reference has no native code, so these stand in for the ATen ops its Python
 * issues at the cited lines.
 */
#ifndef VMM_KERNELS_H
#define VMM_KERNELS_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef void* vmm_stream_t; /* hipStream_t */

/* ---- K1..K5, K8, K10: per-frame convolutions and projections as ONE implicit GEMM ----
 * out[orow, co] = epi( sum_{tap, ci} A[(img, a*stride + dh(tap), b*stride + dw(tap)), ci] * W[tap*Cin + ci, co] )
 * replaces nn.Conv3d (1,k,k) vddp.py:626,271,297,241,708; nn.ConvTranspose3d vddp.py:155 (as 4 phase
 * sub-convolutions); nn.Conv2d 1x1 vddp.py:319,325; nn.Linear vddp.py:413,421.
 * The input may be the channel-concatenation of two tensors (torch.cat skip, vddp.py:813,820).
 * Epilogue (in this order): + bias, * q_scale on the first q_ncols columns, rotary rotation of the first
 * rot_ncols columns (vddp.py:449,456,496; position = frame index of the row), + residual rows. */
typedef struct vmm_conv_desc {
  const float* a1;  const float* a2;      /* a2 may be NULL */
  int32_t C1, C2, lda1, lda2;             /* channels (multiples of 4) and row strides of the sources */
  const float* w;                          /* [ntaps*(C1+C2)][Cout], k-major */
  const float* bias;                       /* [Cout] or NULL */
  const float* res; int32_t ldres;         /* residual rows (indexed like out) or NULL */
  float* out; int32_t ldo;
  int32_t nimg, Hin, Win;                  /* B*T frames, input grid */
  int32_t Hv, Wv, stride;                  /* grid the M dimension runs over (per frame) */
  int32_t KH, KW, off_h, off_w, sgn_h, sgn_w; /* dh = off_h + sgn_h*kh, dw = off_w + sgn_w*kw */
  int32_t Hout, Wout, oscale, ooh, oow;    /* out row = ((img*Hout + a*oscale+ooh)*Wout + b*oscale+oow) */
  int32_t Cout;
  const float* rot_tab;                    /* [rot_T][rot_dh/2][2] (cos,sin) or NULL */
  int32_t rot_T, rot_HW, rot_ncols, rot_dh;
  float q_scale; int32_t q_ncols;
  /* A-operand transform fused into the tile load (0 = none, 1 = GroupNorm+FiLM+SiLU of the producer:
   * silu(x*ga[b,c] + gb[b,c]), coefficients per (sample, channel) of source a1; vddp.py:279-285) */
  int32_t a_mode; const float* a_coef; int32_t a_imgs_per_sample; /* frames per sample (T) */
  /* vmm_conv3x3_bf16x3 only: n_tickets zero-initialised ints that let few-row layers split the channel reduction over several
   * workgroups per output tile; the partial sums are added in a fixed order (ticket = next split), so results are bit-reproducible.
   * The kernel leaves them zero again.  NULL / too few: no split.  Launches sharing the array must be ordered on one stream. */
  int32_t* split_tickets; int32_t n_tickets;
  /* vmm_conv3x3_bf16x3 only: when non-NULL and n = vmm_conv3x3_fuses_gn(d) > 0 the kernel also leaves GroupNorm partial sums of its
   * output (+ bias) in gn_part[B * gn_groups][n][2] = n fp32 (sum x, sum x^2) pairs per (sample, group), every slot written exactly
   * once (vddp.py:274-279; vmm_groupnorm_coef totals them in a fixed order); samples are runs of a_imgs_per_sample frames. */
  float* gn_part; int32_t gn_groups;
  /* periodic ("circular") padding instead of zero padding along h / w (vddp.py:163-243: padding_mode 'circular' = both, 'circular_1d' =
   * w only): taps that leave the frame read the opposite border.  Honoured by the implicit-GEMM kernels, vmm_conv_wgrad_f32 and the 2-D-tiled
   * instances of the 3 x 3 halo kernels (flat row tiles, the persistent kernel, vmm_conv_s2 / vmm_stem_conv return 1 / are not used). */
  int32_t wrap_h, wrap_w;
  /* > 0: source frame i >= a_img_mod reads frame i - a_img_mod of a1 (the a_coef sample and the output row stay those of frame i): the two
   * halves of a guidance batch sharing ONE pre-norm tensor that the first convolution computed for half the batch (plan.py, mirrored
   * plans).  Honoured by the 2-D-tiled unsplit instances of the 3 x 3 halo kernels; every other kernel returns 1 / -1 for it. */
  int32_t a_img_mod;
  /* storage of the feature maps (the "bf16" throughput mode keeps the two upper levels' maps as bf16 in HBM): bit 0 = a1 / a2 point at bf16,
   * bit 1 = out (and res) point at bf16; 0 = fp32 everywhere.  Honoured by the single-pass entry points (vmm_conv3x3_bf16, vmm_conv_s2_acc_bf16,
   * vmm_proj_bf16, vmm_proj_bf16_res_silu); every other kernel returns -1 for a non-zero value.  ld* stay in ELEMENTS. */
  int32_t act_bf16;
  /* EXPERIMENTS library only (libvmm_hip_exp.so; the product library ignores both fields): workspace of sk_slots partial output tiles of 128 x 128
   * floats for the BALANCED launch of the few-tile 3 x 3 layers (a grid of two workgroups per CU shares the launch's (tile, channel chunk) iterations
   * evenly; the pieces of a tile are added in a fixed order, so results stay bit-reproducible).  Needs split_tickets with n_tickets >= 2048 + sk_slots
   * (entries 2048.. are the pieces' flags, zero before and after every launch).  NULL: one workgroup per tile. */
  float* sk_work; int32_t sk_slots;
  /* weight-gradient entry points with a workspace (vmm_conv3x3_wgrad_*, vmm_conv1x1_wgrad_*): != 0 = run the first stage only -- the partial blocks stay in the
   * workspace, which the caller keeps until it has totalled them with vmm_reduce_batch (jobs from vmm_conv3x3_wgrad_reduce_job / vmm_conv1x1_wgrad_reduce_job).
   * Every other entry point ignores the field. */
  int32_t defer_reduce;
} vmm_conv_desc;
int vmm_conv_igemm_f32(const vmm_conv_desc* d, vmm_stream_t stream);
/* Same contraction on the bf16 matrix cores with split-precision operands (a_hi*b_hi + a_hi*b_lo + a_lo*b_hi, fp32 accumulate;
 * ~1e-5 relative error): d->w must point to the fmt-1 output of vmm_pack_weights (pre-split, pre-transposed bf16 weights). */
int vmm_conv_igemm_bf16x3(const vmm_conv_desc* d, vmm_stream_t stream);
/* n <= 4 problems of identical shape (rows, taps, channels, output columns) in ONE launch, e.g. the four output phases of
 * ConvTranspose3d (1,4,4) stride 2 (vddp.py:155): descs[i] differ in weights / tap offsets / output phase only. */
int vmm_conv_igemm_bf16x3_batched(const vmm_conv_desc* descs, int32_t n, vmm_stream_t stream);
/* 3x3 / stride 1 / pad 1 specialisation of the above: LDS-resident halo patch (each input element is staged once per channel chunk
 * instead of once per tap) and weights read straight into registers in MFMA fragment order -- d->w must point to the fmt-2 output of
 * vmm_pack_weights.  Needs C1, C2 multiples of 32, Cout == 64 or a multiple of 128, and W <= 31 or (W % 16 == 0 and H % 16 == 0);
 * returns 1 (nothing launched) when the descriptor is outside that envelope. */
int vmm_conv3x3_bf16x3(const vmm_conv_desc* d, vmm_stream_t stream);
/* the "bf16" THROUGHPUT mode of the same kernel (BASELINE.json configs[3], `precision="bf16"`): same descriptor, same fmt-2 weights, same
 * envelope, tickets and GroupNorm partials; ONE matrix pass on the operands' bf16 roundings (2^-9 relative per operand), fp32 accumulation */
int vmm_conv3x3_bf16(const vmm_conv_desc* d, vmm_stream_t stream);
/* the same kernel on the exact-fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32; the "fp32" arithmetic mode): d->w = fmt-4 output of
 * vmm_pack_weights; same envelope, tickets and GroupNorm partials */
int vmm_conv3x3_f32(const vmm_conv_desc* d, vmm_stream_t stream);
/* host-only query: the number n of partial-sum pairs per (sample, group) vmm_conv3x3_bf16x3(d) will leave in d->gn_part
 * (unsplit 2-D-tiled layers), or 0 when it will not produce them */
int vmm_conv3x3_fuses_gn(const vmm_conv_desc* d);
/* host-only query: 1 when vmm_conv3x3_bf16x3 / vmm_conv3x3_f32 would take d as it stands (a_mode, a_img_mod, wrap_h / wrap_w, res ...), else 0 */
int vmm_conv3x3_accepts(const vmm_conv_desc* d);
/* (the Winograd F(2x2, 3x3) variant of this convolution -- built, parity-green, slower -- is declared in vmm_experiments.h and compiled only into
 * libvmm_hip_exp.so) */
/* 1x1 / Linear specialisation (to_qkv, to_out vddp.py:319,325,413,421; res_conv vddp.py:297): a workgroup stages its rows' full K
 * extent once in LDS and sweeps all output columns, weights read straight into registers in MFMA fragment order (d->w = fmt-2 output
 * of vmm_pack_weights), 16-byte epilogue stores; same epilogue options as vmm_conv_igemm_*.  ln_gamma != NULL: the rows pass through
 * the channel LayerNorm of PreNorm (vddp.py:245-254; gamma [K], eps inside the sqrt) while they are staged, which replaces a
 * separate vmm_channel_layernorm pass.  Envelope: KH = KW = 1, stride 1, identity row mapping, K = C1 + C2 padded to 32 in
 * {32, 64, 128, 256}; returns 1 (nothing launched) otherwise. */
int vmm_proj_bf16x3(const vmm_conv_desc* d, const float* ln_gamma, float ln_eps, vmm_stream_t stream);
/* the "bf16" throughput mode of the same kernel (and of its ResnetBlock-tail form below): one matrix pass on bf16-rounded operands */
int vmm_proj_bf16(const vmm_conv_desc* d, const float* ln_gamma, float ln_eps, vmm_stream_t stream);
int vmm_proj_bf16_res_silu(const vmm_conv_desc* d, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream);
/* the same, also leaving the LayerNorm statistics of every row in ln_stats [rows][2] = (mean, 1 / sqrt(var + eps)): the training forward of to_qkv
 * (the normalised rows are never materialised; vmm_conv1x1_wgrad_bf16x3_ln re-normalises from these) */
int vmm_proj_bf16x3_ln_stats(const vmm_conv_desc* d, const float* ln_gamma, float ln_eps, float* ln_stats, vmm_stream_t stream);
/* ResnetBlock tail (vddp.py:311) in one launch: out = silu(res * a + b') + proj(x); res = d->res = the pre-norm output of block2's
 * convolution (may alias d->out), (a, b') = res_coef [B][Cout][2] from vmm_groupnorm_coef, proj = res_conv.  Envelope of vmm_proj_bf16x3. */
int vmm_proj_bf16x3_res_silu(const vmm_conv_desc* d, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream);
/* wide contraction, 64 output columns (to_out K = 256 -> 64, the to_qkv data gradient K = 768 -> 64 at the 96 x 96 level: HBM-bound sweeps):
 * activations as the MFMA's B operand straight from global memory into registers (a lane = a row, 32 contiguous bytes per k16 step), fmt-2
 * weights straight from L1 / L2, a wave owns 64 rows x 64 columns, no LDS and no barrier (narrow_proj.hip).  Envelope: KH = KW = 1, stride 1,
 * identity rows, Cout == 64, C1 / C2 multiples of 16, K = C1 + C2 >= 64 and a multiple of 32, no fused operand transform, no rotary / q-scale;
 * bias / residual epilogue.  Returns 1 (nothing launched) outside it. */
int vmm_proj_narrow_bf16x3(const vmm_conv_desc* d, vmm_stream_t stream);
/* the ResnetBlock tail (vmm_proj_bf16x3_res_silu's contract: out = silu(res * a + b') + proj(x), res_coef [B][64][2]) on the same kernel */
int vmm_proj_narrow_bf16x3_res_silu(const vmm_conv_desc* d, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream);
/* exact-fp32 variant (d->w = fmt-4 output of vmm_pack_weights) */
int vmm_proj_f32(const vmm_conv_desc* d, const float* ln_gamma, float ln_eps, vmm_stream_t stream);

/* ---- training: weight gradient of the same contraction (autograd of vddp.py:155,241,271,297,319,325,413,421,626,708).
 * dw_packed[(tap, ci)][co] += sum_m A[m shifted by tap, ci] * dy[orow(m), co]; `d` is the FORWARD descriptor of the layer
 * (out/res/bias/rot fields ignored), the reduction is split over `nsplit` row slices combined with fp32 atomics, so
 * dw_packed must be zeroed by the caller.  dbias != NULL: dbias[co] += sum_m dy[orow(m), co] as well (the layer's bias gradient,
 * from the dY tiles the kernel stages anyway: one partial row per row slice in bias_scratch [nsplit][Cout], then vmm_sum_partials). */
int vmm_conv_wgrad_f32(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                       float* bias_scratch, vmm_stream_t stream);
/* the same contraction with split-bf16 (hi + lo) operands on the bf16 matrix cores, three passes per product: the "bf16x3" training
 * mode's weight gradient; same alignment requirements */
int vmm_conv_wgrad_bf16x3(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                          float* bias_scratch, vmm_stream_t stream);
/* out[c] += sum_{k < n} part[k * ld + c], fixed order: second stage of the reductions that leave one partial row per workgroup */
/* the 3 x 3 / stride 1 / pad 1 case of vmm_conv_wgrad_f32 with all nine taps of a (64 input channels) x (64 output channels) block in one
 * workgroup (one LDS patch of x and dY per row segment; wgrad3x3.hip).  vmm_conv_wgrad_f32 forwards to it; returns 1 (nothing launched)
 * outside its envelope: no fused operand transform, C1 / C2 / Cout multiples of 64, W a multiple of 24 or W = 12 with an even H. */
int vmm_conv3x3_wgrad_f32(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                          float* bias_scratch, vmm_stream_t stream);
/* the 3 x 3 / stride 1 / pad 1 case of vmm_conv_wgrad_bf16x3: all nine taps of a (64 x 64)-channel block per workgroup, x staged once in an
 * LDS ring of [channel][position] bf16 hi | lo fragments, the horizontal taps as register shifts of the dY fragment (wgrad3x3_bf16x3.hip).
 * vmm_conv_wgrad_bf16x3 forwards to it (without a workspace); returns 1 (nothing launched) outside its envelope: zero padding, no fused
 * operand transform, C1 / C2 / Cout multiples of 64.  workspace: vmm_conv3x3_wgrad_bf16x3_workspace(d, lddy) floats (0 = outside the
 * envelope), contents irrelevant -- the row slices leave partial blocks there and a second launch totals them in a fixed order
 * (bit-reproducible, and a third of the time fp32 atomics take); NULL: atomics straight into dw_packed / dbias. */
int64_t vmm_conv3x3_wgrad_bf16x3_workspace(const vmm_conv_desc* d, int32_t lddy);
int vmm_conv3x3_wgrad_bf16x3(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                             vmm_stream_t stream);
/* the 1 x 1 / Linear case (to_qkv, to_out, res_conv): dw_packed[ci][co] += sum_r x[r][ci] dY[r][co] on the split-bf16 matrix cores, a 128 x 128
 * channel block per workgroup, [channel][row] fragment images in LDS, partial blocks + fixed-order reduction (wgrad1x1_bf16x3.hip).  Envelope:
 * KH = KW = 1, stride 1, identity rows, no fused operand transform, C1 / C2 / Cout multiples of 64; workspace =
 * vmm_conv1x1_wgrad_bf16x3_workspace(d, lddy) floats (0 = outside the envelope).  Returns 1 (nothing launched) outside it or without workspace. */
int64_t vmm_conv1x1_wgrad_bf16x3_workspace(const vmm_conv_desc* d, int32_t lddy);
int vmm_conv1x1_wgrad_bf16x3(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                             vmm_stream_t stream);
/* the same for a PreNorm(to_qkv) whose forward fused the channel LayerNorm (vmm_proj_bf16x3_ln_stats): the layer's input is re-normalised while
 * it is staged, x[r][ci] = (a1[r][ci] - mean[r]) * rstd[r] * ln_gamma[ci] with (mean, rstd) = ln_stats[r][2]; single source (C2 == 0) */
int vmm_conv1x1_wgrad_bf16x3_ln(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* workspace, const float* ln_stats,
                                const float* ln_gamma, vmm_stream_t stream);
/* The same kernel for ANY convolution geometry the descriptor states (round 6; the exact-fp32 vmm_conv_wgrad_f32 before): kernel size, stride, tap direction (sgn: the phases
 * of the transposed convolutions), output scatter (oscale / ooh / oow), zero or periodic padding -- the 4 x 4 stride-2 layers (vddp.py:139-151), their transposed twins phase by
 * phase, the 7 x 7 stem.  dw_packed[(kh, kw, ci)][co] += sum over output rows; dbias[co] += column sums of the dY rows the call visits.  The loader decodes its eight
 * wave-uniform output rows on the scalar unit and offsets them by the lane's tap.  Envelope: no fused operand transform, even channel counts, KH KW (C1 + C2) a multiple
 * of 4, Hv Wv >= 2; workspace = vmm_conv_wgrad_tap_workspace floats (0 = outside); d->defer_reduce + vmm_conv_wgrad_tap_reduce_job as for the 1 x 1 form. */
int64_t vmm_conv_wgrad_tap_workspace(const vmm_conv_desc* d, int32_t lddy);
int vmm_conv_wgrad_tap_bf16x3(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_stream_t stream);
/* Backward of to_qkv at the C = 64 levels in ONE pass over the gradient g of the qkv rows (rows x 768): gy = g W (rows x 64, plain store) and
 * dw_packed[c][n] += sum_r g[r][n] y[r][c], y = x (ln_stats NULL) or the channel LayerNorm of x re-formed from ln_stats [rows][2] = (mean, rstd) and
 * ln_gamma [64] (vmm_proj_bf16x3_ln_stats).  w_frag = vmm_pack_weights fmt 2 of the (K = 768, N = 64) operand, i.e. the torch weight (768, 64) as the
 * data gradient's "input-major" matrix.  Both LDS images a piece of g needs ([column][row] for the weight gradient, [row][column] for the data
 * gradient) are written by the one loader pass (qkv_bwd.hip).  workspace = vmm_qkv_bwd_workspace(rows, C, Nq) floats (0 = outside the envelope:
 * C == 64, Nq == 768, rows a multiple of 64).  Returns 1 (nothing launched) outside the envelope. */
int64_t vmm_qkv_bwd_workspace(int64_t rows, int32_t C, int32_t Nq);
int vmm_qkv_bwd_bf16x3(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                       float* gy, int32_t ldgy, float* dw_packed, float* workspace, int64_t rows, int32_t C, int32_t Nq, vmm_stream_t stream);
/* The same pass with the backward of the PreNorm channel LayerNorm (vddp.py:245-264; vmm_channel_layernorm_bwd) as its epilogue -- gy never reaches
 * memory: dx[r][c] (= | +=, accumulate) rstd_r (gamma_c gy[r][c] - mean_c(gamma gy) - xhat[r][c] mean_c(gamma gy xhat)) and
 * dgamma[c] += sum_r gy[r][c] xhat[r][c] (per-workgroup partial rows inside the workspace + the fixed-order vmm_sum_partials).  ln_stats, ln_gamma and dx
 * are required; dx may be the buffer the residual gradient already lies in (accumulate = 1: read and written row by row).  Same envelope. */
int vmm_qkv_bwd_ln_bf16x3(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                          float* dx, int32_t lddx, int32_t accumulate, float* dgamma, float* dw_packed, float* workspace, int64_t rows, int32_t C,
                          int32_t Nq, vmm_stream_t stream);
int vmm_sum_partials(const float* part, int32_t n, int32_t ld, int32_t C, float* out, vmm_stream_t stream);
/* ---- deferred second stages (round 6).  The weight-gradient kernels leave one partial block per row slice and total them in a fixed order with a second launch of
 * 5-13 us; the training step's backward ran ~70 of those behind its 3 x 3 / 1 x 1 weight gradients.  A caller that sets vmm_conv_desc.defer_reduce keeps each
 * layer's workspace alive, asks for the layer's job (host-side, nothing launched; the arguments are those of the weight-gradient call) and totals many layers
 * at once: jobs_dev = the jobs in device memory with wg0 = the running sum of the preceding jobs' `wgs`, total_wgs = the sum of all.  Same summation order as
 * the per-layer launches (bit-identical results).  kind 0 = plain partial rows, out[c] += sum_{k < nz} part[k ld + c], c < Cout (what vmm_sum_partials does;
 * wgs = ceil(Cout / 16)). */
typedef struct vmm_reduce_job {
  const float* part; float* out;
  const float* bias_part; float* dbias;
  int32_t kind;                 /* 0 rows, 1 vmm_conv3x3_wgrad_* blocks, 2 vmm_conv1x1_wgrad_* blocks */
  int32_t nz, tiles_x, tiles_y, Cin, Cout, ld, n_main, gx;
  int32_t wgs;                  /* workgroups the job needs (filled by the *_reduce_job queries) */
  int32_t wg0;                  /* first workgroup of the job inside the batched launch (filled by the caller) */
  int32_t pad_;
} vmm_reduce_job;
int vmm_conv3x3_wgrad_reduce_job(const vmm_conv_desc* d, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job);
int vmm_conv1x1_wgrad_reduce_job(const vmm_conv_desc* d, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job);
int vmm_conv_wgrad_tap_reduce_job(const vmm_conv_desc* d, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job);
int vmm_reduce_batch(const vmm_reduce_job* jobs_dev, int32_t njobs, int32_t total_wgs, vmm_stream_t stream);
/* out[j] += sum_m x[m, j] (bias gradients and other per-channel reductions) */
int vmm_colsum_accumulate(const float* x, int32_t ldx, int64_t rows, int32_t C, float* out, vmm_stream_t stream);
/* batched weight packing: packed[(th*TW + tw)*Cp + c][n] <-> torch[n*sn + c*sc + (h0 + th*hs)*sh + (w0 + tw*ws)*sw];
 * direction 0: torch -> packed (channels c >= C are zero padding); 1: packed gradient -> torch (= or += per job). */
typedef struct vmm_pack_job {
  float* torch_w; float* packed;
  int32_t TH, TW, C, Cp, N;
  int32_t sn, sc, sh, sw, h0, hs, w0, ws;
  int32_t accumulate;
  int32_t fmt; /* 0: fp32 [K][N];  1: split bf16 for vmm_conv_igemm_bf16x3: [N][Kpad] hi plane, then lo plane (Kpad = K rounded up to 32);
                * 2: split bf16 in MFMA fragment order for vmm_conv3x3_bf16x3 / vmm_linattn_block_bf16x3: [N/32][Kpad/16][hi|lo][64 lanes][8],
                *    lane l = column (l & 31), k = step*16 + (l >> 5)*8 .. +7 (N and K padded to 32 with zeros);
                * 3: as 2 with k permuted inside every 32-block to the accumulator-register order (linattn_block.hip);
                * 4: fp32 in the fragment order of 2 for vmm_conv3x3_f32 / vmm_proj_f32: [N/32][Kpad/16][e 0..3 | e 4..7][64 lanes][4].
                * 5 / 6: the (1,4,4) stride-2 resampling kernels for vmm_conv_s2_bf16x3 in the fragment order of 2, as 3 x 3 convolutions
                *    (TH = TW = 4; strides of the torch tensor as usual; h0 / hs / w0 / ws / Cp unused):
                *    5 = Conv3d (N, C, 1, 4, 4) over 2 x 2 input cells: K = 9 taps x 4 C cell channels, [N/32][9*4C/16][hi|lo][64][8];
                *    6 = ConvTranspose3d (C, N, 1, 4, 4) with the four output phases as 4 N columns: [4N/32][9*C/16][hi|lo][64][8];
                * 7: the stem convolution (N, C <= 4, 1, k, k), k <= 8, for vmm_stem_conv_bf16x3: fragment order of 2 with
                *    K = (kernel row, tap 0..7, channel 0..3), zero beyond tap k - 1 and channel C - 1: [N/32][2k][hi|lo][64][8] (TH = TW = k);
                * 8: the 3 x 3 kernel (N, C, 1, 3, 3) as Winograd F(2x2, 3x3) weights U = G g G^T for vmm_conv3x3_wino_bf16x3, split, in "A" fragment order:
 *    [N/64][Cp/16][position xi*4+nu][column fragment 0..1][hi|lo][64 lanes][8], lane l = column nb*64 + mf*32 + (l & 31),
 *    channels ks*16 + (l >> 5)*8 .. +7 (TH = TW = 3, Cp a multiple of 16, N of 64; 64 (N/64) Cp bytes);
 *    1-8: direction 0 only.
 *    1 | 16, 2 | 16, 3 | 16, 5 | 16, 6 | 16: the same orders with IEEE-half values in the hi plane and a zero lo plane: operands of the `_fp16` entry points;
 *    ... | 32: IEEE-half hi plane AND IEEE-half lo plane (w - hi): operands of the `_f16x3` entry points of the experiments library (vmm_experiments.h). */
} vmm_pack_job;
int vmm_pack_weights(const vmm_pack_job* jobs_dev, int32_t njobs, int32_t max_elems, int32_t direction, vmm_stream_t stream);

/* ---- K3a: the stem, Conv3d(C <= 4, 64, (1,k,k)) pad k/2 (vddp.py:600 init_conv, k = 7): x rows [nimg*H*W][4] (vmm_ncthw_to_rows with
 * ldo = 4), weights = vmm_pack_weights fmt 7, out rows x 64 (ldo).  A 16 x 16 pixel tile's neighbourhood is staged once in LDS and a k16
 * step = four neighbouring taps = two 32-byte LDS reads.  Returns 1 (nothing launched) unless Cout == 64 and k is odd and <= 8. */
int vmm_stem_conv_bf16x3(const float* x, const float* w_packed, const float* bias, float* out, int32_t ldo, int32_t nimg, int32_t H, int32_t W,
                         int32_t Cout, int32_t k, vmm_stream_t stream);

/* ---- K3b: the resampling layers (vddp.py:155 Upsample = ConvTranspose3d (1,4,4) stride (1,2,2) pad (0,1,1); vddp.py:158 Downsample = Conv3d
 * of the same geometry) as tap-subset 3 x 3 convolutions with an LDS halo patch (conv3x3_bf16x3.hip): every input element is staged once
 * per channel chunk instead of gathered once per tap.  x rows [nimg][Hin][Win] x Cin (ldx), out rows [nimg][Hout][Wout] x Cout (ldo),
 * Hout = Hin / 2 (up == 0) or 2 Hin (up == 1); weights = vmm_pack_weights fmt 5 (up == 0) / fmt 6 (up == 1); bias [Cout] or NULL.
 * Returns 1 (nothing launched) outside the envelope -- Cin a power of two >= 32, Cout == 64 or Cout % 128 == 0, even Hin / Win for
 * up == 0, tile space (input pixels / 2 x 2 input cells) at most 31 wide or a multiple of 16 wide and of 8 (16 when Cout == 64 and
 * up == 0) high -- the caller then uses vmm_conv_igemm_bf16x3 (four phase launches for the transposed convolution). */
int vmm_conv_s2_supported(int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up); /* 1: inside the envelope */
int vmm_conv_s2_bf16x3(const float* x, int32_t ldx, const float* w_packed, const float* bias, float* out, int32_t ldo, int32_t nimg,
                       int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, vmm_stream_t stream);
/* the same with out = convolution (+ bias) + res (res rows indexed like out, may alias it; ldres a multiple of 4): the layers' data gradients --
 * d Downsample / dx = the Upsample kernel over dOut with the convolution's (Cout, Cin, 1, 4, 4) tensor read as a transposed-convolution weight
 * (fmt 6, C = Cout, N = Cin), d Upsample / dx = the Downsample kernel over dOut with the (Cin, Cout, 1, 4, 4) tensor read as a convolution weight
 * (fmt 5, N = Cin, C = Cout) -- accumulating into a gradient buffer that already holds the skip connection's share.  split_tickets / n_tickets
 * (NULL / 0: never split) as in vmm_conv_desc: few-tile up == 0 layers split their channel reduction over several workgroups per tile */
int vmm_conv_s2_acc_bf16x3(const float* x, int32_t ldx, const float* w_packed, const float* bias, const float* res, int32_t ldres, float* out,
                           int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                           int32_t n_tickets, vmm_stream_t stream);
/* the "bf16" throughput mode of the same layers: one matrix pass on bf16-rounded operands */
int vmm_conv_s2_acc_bf16(const float* x, int32_t ldx, const float* w_packed, const float* bias, const float* res, int32_t ldres, float* out,
                           int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                           int32_t n_tickets, vmm_stream_t stream);
/* ... over bf16-STORED feature maps: a16 = 1: x, res, out bf16; 2: x bf16, out (res) fp32 (the Downsample leaving the bf16 levels); 3: x fp32, out
 * (res) bf16 (the Upsample entering them).  Unsplit instances (the upper levels have tiles to spare); ld in elements. */
int vmm_conv_s2_acc_bf16_a16(const void* x, int32_t ldx, const float* w_packed, const float* bias, const void* res, int32_t ldres, void* out,
                             int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t a16,
                             vmm_stream_t stream);

/* ---- K6: GroupNorm(groups, C) statistics + fused affine/FiLM/SiLU (vddp.py:274-285) ---- */
/* sums[b, g] = (sum x, sum x^2) over (C/G channels, all rows of sample b), accumulated in fp64. */
int vmm_groupnorm_stats(const float* x, int32_t ldx, int32_t B, int32_t rows_per_sample, int32_t C, int32_t G,
                        double* sums /* [B*G*2], zeroed by the call */, vmm_stream_t stream);
/* the same statistics as per-workgroup fp32 (sum, sum of squares) slots part[B*G][n][2], n = vmm_groupnorm_stats_slots(B, rows_per_sample, C);
 * vmm_groupnorm_coef(sums = NULL, partials = part, n_contrib = n) adds them in a fixed order (bit-reproducible; no zero-fill, no atomics) */
int vmm_groupnorm_stats_slots(int32_t B, int32_t rows_per_sample, int32_t C);
int vmm_groupnorm_stats_partials(const float* x, int32_t ldx, int32_t B, int32_t rows_per_sample, int32_t C, int32_t G, float* part,
                                 vmm_stream_t stream);
/* mean/rstd from the sums, then coef[b,c] = (a, b'):  y = silu(x*a + b')  with a = rstd*gamma*(scale+1),
 * b' = (beta - mean*rstd*gamma)*(scale+1)+shift;  film = [B][ldfilm] rows (scale | shift) or NULL (vddp.py:283,306).
 * stats_out [B*G*2] = (mean, rstd) is kept for the backward pass (may be NULL).
 * Source of the moments, in this order of precedence: partials != NULL: the fixed-order total of n_contrib fp32 (sum x, sum x^2)
 * pairs per (sample, group) that vmm_conv3x3_bf16x3 left in d->gn_part; x != NULL: a direct fixed-order reduction of the group's
 * slice of x (rows [B * count_per_group / (C/G)][ldx]; for small layers, saves the vmm_groupnorm_stats launch); else sums. */
int vmm_groupnorm_coef(const double* sums, int64_t count_per_group, float eps, const float* gamma, const float* beta,
                       const float* film, int32_t ldfilm, int32_t B, int32_t C, int32_t G, float* coef /* [B][C][2] */,
                       float* stats_out, const float* partials, int32_t n_contrib, const float* x, int32_t ldx, vmm_stream_t stream);
/* y = silu(x*a + b') (+ res) ; in place allowed (vddp.py:285,311). */
int vmm_affine_silu(const float* x, int32_t ldx, const float* coef, const float* res, int32_t ldres, float* y, int32_t ldy,
                    int64_t rows, int32_t rows_per_sample, int32_t C, vmm_stream_t stream);
/* the same pass over bf16-STORED feature maps (x, res, y = bf16 bits, ld in elements; coefficients fp32): the "bf16" throughput mode */
int vmm_affine_silu_a16(const void* x, int32_t ldx, const float* coef, const void* res, int32_t ldres, void* y, int32_t ldy, int64_t rows,
                        int32_t rows_per_sample, int32_t C, vmm_stream_t stream);
/* ---- entry points over bf16-STORED feature maps (the "bf16" throughput mode keeps the two upper levels' maps as bf16 in HBM; same arguments as the
 * functions they are named after, activation pointers = bf16 bits, ld in elements, everything else fp32) */
int vmm_affine_silu_pointwise_to_ncthw_a16(const void* x, int32_t ldx, const float* coef, const void* res, int32_t ldres, int32_t C, const float* w,
                                           const float* bias, int32_t B, int32_t Cout, int32_t T, int32_t HW, float* out, vmm_stream_t stream);
int vmm_stem_conv_bf16x3_a16(const float* x, const float* w_frag, const float* bias, void* out, int32_t ldo, int32_t nimg, int32_t H, int32_t W,
                             int32_t Cout, int32_t k, vmm_stream_t stream);
int vmm_temporal_attention_a16(const void* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                               int32_t bias_on_cond, void* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, float* lse,
                               vmm_stream_t stream);
int vmm_temporal_block_bf16_a16(const void* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed, const float* ek,
                                const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond, const float* rot_tab, void* out, int32_t ldo,
                                int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float q_scale, float eps, vmm_stream_t stream);
int vmm_linattn_block_bf16_a16(const void* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag, const float* bias_out,
                               const float* ek, const float* ev, int32_t ntok, float* workspace, void* out, int32_t ldo, int32_t B, int32_t T,
                               int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream);
/* storage conversion of a dense feature map, n elements (a multiple of 4): src_bf16 / dst_bf16 = 0 fp32, 1 bf16 (round to nearest even) */
int vmm_convert_act(const void* src, int32_t src_bf16, void* dst, int32_t dst_bf16, int64_t n, vmm_stream_t stream);

/* the last ResnetBlock's output pass and final_conv.1 (vddp.py:311,729) in one kernel: out (B, Cout, T, HW) = bias + w (Cout, 64) .
 * (silu(x*a + b') + res) per row; the block's output is never stored.  Returns 1 (nothing launched) unless C == 64 and Cout <= 4. */
int vmm_affine_silu_pointwise_to_ncthw(const float* x, int32_t ldx, const float* coef, const float* res, int32_t ldres, int32_t C,
                                       const float* w, const float* bias, int32_t B, int32_t Cout, int32_t T, int32_t HW, float* out,
                                       vmm_stream_t stream);

/* ---- K7: channel LayerNorm, gamma only, eps inside sqrt (vddp.py:245-254) ---- */
int vmm_channel_layernorm(const float* x, int32_t ldx, const float* gamma, float* y, int32_t ldy, int64_t rows, int32_t C,
                          float eps, vmm_stream_t stream);

/* ---- K11/K13/K14: temporal softmax attention core (vddp.py:491-534), one (b, pixel, head) problem per 11 queries.
 * qkv rows [(b,t,hw)][3*heads*dh] with q pre-scaled and q,k pre-rotated by the projection epilogue;
 * ek/ev = conditioning keys/values [B][ntok][heads*dh] (ek pre-rotated when per-frame) or NULL;
 * bias = relative position bias [heads][T][T]; bias_on_cond: also add it to the token half (vddp.py:505-510). */
int vmm_temporal_attention(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                           const float* bias, int32_t bias_on_cond, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW,
                           int32_t heads, int32_t dh, float* lse /* [rows][heads] logsumexp for the backward, or NULL */,
                           vmm_stream_t stream);
/* ---- cond_attention = 'cross-attention' (vddp.py:354-363, 476-485): queries from to_q, keys / values = the conditioning tokens alone.
 * Softmax flavour (mid spatial site: bias NULL; temporal sites: bias [heads][T][T] added to the (frames x tokens) scores, ntok == T as in the
 * reference): out[row, head*dh + e] = sum_j softmax_j(q[row, head] . ek[b][j][head] (+ bias[head][t][j])) ev[b][j][head*dh + e]; q rows
 * [(b, t, pixel)] x heads*dh (ldq), pre-scaled / pre-rotated by the projection epilogue; ek / ev [B][ntok][heads*dh], ntok <= 64. */
int vmm_cross_attention(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, float* out, int32_t ldo,
                        int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
/* linear flavour: the context ctx[(b, t, head)][dh][dh] = softmax_j(ek[b][j])^T ev[b][j] / HW of the tokens alone (every frame of a sample gets
 * the same block); vmm_linattn_apply then runs on the q rows (ldqkv = heads*dh).  kstat as in vmm_linattn_context (or NULL). */
int vmm_linattn_cross_context(const float* ek, const float* ev, int32_t ntok, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, float* ctx,
                              float* kstat, vmm_stream_t stream);
/* backward of the two (training with cond_attention = 'cross-attention'; loss.backward() through vddp.py:354-363 / 476-485).
 * vmm_cross_attention_bwd: q = the rows the forward consumed, dout [rows][heads*dh]; writes dq = the gradient of the RAW to_q output (the
 * projection epilogue's rotation -- rot_tab [T][dh/2][2] (cos, sin) or NULL -- and q_scale are undone here), ADDS the token gradients into
 * dek / dev [B][ntok][heads*dh] and the bias gradient into dbias [heads][T][T] (NULL allowed).  Any number of heads <= 64, dh a multiple of 4
 * in 4..128 (the temporal sites follow attn_dim_head, vddp.py:615), ntok <= 64; scratch = vmm_cross_attention_bwd_scratch floats (0 -- and NULL
 * accepted -- where the one-pass kernel applies: 8 heads of 32, at most 16 tokens; otherwise the per-row ds / p records of the two-pass form).
 * vmm_linattn_cross_bwd: ctx / kstat as vmm_linattn_cross_context left them, dctx = [B*T*heads][dh*dh] scratch; writes dq, ADDS dek / dev. */
int64_t vmm_cross_attention_bwd_scratch(int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, int32_t ntok);
int vmm_cross_attention_bwd(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, const float* dout,
                            int32_t lddo, const float* rot_tab, float q_scale, float* dq, int32_t lddq, float* dek, float* dev, float* dbias,
                            float* scratch, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
int vmm_linattn_cross_bwd(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* ctx, const float* kstat,
                          const float* dout, int32_t lddo, float* dctx, float* dq, int32_t lddq, float* dek, float* dev, int32_t B, int32_t T, int32_t HW,
                          int32_t heads, int32_t dh, vmm_stream_t stream);
/* the path vmm_temporal_attention takes where it applies (heads = 8, dh = 32, T <= 16, ntok <= 16; returns 1 and launches nothing
 * otherwise): one workgroup per pixel, the T rows of k | v staged once in LDS */
int vmm_temporal_attention_staged(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                  int32_t bias_on_cond, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads,
                                  int32_t dh, float* lse, vmm_stream_t stream);

/* Temporal attention core + to_out + residual for the levels whose to_qkv is a separate projection (C = 128, 256, 512;
 * vddp.py:491-534, 421): out = x + to_out(attention(q, [ek|k], [ev|v])) with the scores and the value mix on the split-bf16 matrix
 * cores; the qkv rows (vmm_proj_bf16x3 output: q pre-scaled, q / k pre-rotated) are read once, the attention output never touches HBM.
 * wout_frag = vmm_pack_weights fmt 3 of to_out (C, 256).  Envelope: heads == 8, dim_head == 32, C % 128 == 0, T <= 16, ntok <= 16,
 * HW even; returns 1 (nothing launched) otherwise. */
int vmm_temporal_core_bf16x3(const float* qkv, int32_t ldqkv, const float* x, int32_t ldx, const float* wout_frag, const float* ek,
                             const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond, float* out, int32_t ldo, int32_t B,
                             int32_t T, int32_t HW, int32_t C, int32_t heads, vmm_stream_t stream);

/* Fused spatial linear-attention BLOCK for the full-resolution level (vddp.py:313-378 inside Residual(PreNorm(.)), vddp.py:613/628):
 * out = x + to_out(linear_attention(to_qkv(LayerNorm(x)))) + bias with the conditioning tokens ek/ev [B][ntok][256] stacked onto k, v.
 * q, k, v never touch HBM (x read twice, out written once).  wqkv_frag = vmm_pack_weights fmt 2 of to_qkv (768,64), wout_frag = fmt 3
 * of to_out (64,256); workspace = vmm_linattn_block_workspace(B,T,HW) floats.
 * Envelope: C == 64, heads == 8, dim_head == 32, HW % 32 == 0; returns 1 (nothing launched) otherwise. */
int64_t vmm_linattn_block_workspace(int32_t B, int32_t T, int32_t HW);
int vmm_linattn_block_bf16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                             const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                             int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream);
/* the "bf16" throughput mode of the same block: identical arguments and packed weights, one matrix pass per product on bf16-rounded operands
 * (LayerNorm, softmax, rotary, residual in fp32) */
int vmm_linattn_block_bf16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                             const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                             int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream);

/* Fused temporal-attention BLOCK for the full-resolution level (vddp.py:615,630,680: x + to_out(attn(rotary(to_qkv(LayerNorm(x)))))):
 * x is read once and out written once, qkv / attention outputs never touch HBM.  Projections, scores and value mix all run on the
 * split-bf16 matrix cores (wqkv_packed = vmm_pack_weights fmt 2 of to_qkv (768,64), wout_packed = fmt 3 of to_out (64,256)).
 * Envelope: C == 64, heads == 8, dim_head == 32, ntok <= 16, and T <= 16 with HW even (two pixels per 32-row tile) or T <= 32 (one pixel
 * per tile); returns 1 (nothing launched) otherwise.  vmm_temporal_block_supported: 0 = outside, 1 / 2 = which kernel would run. */
int vmm_temporal_block_supported(int32_t T, int32_t ntok, int32_t HW, int32_t C, int32_t heads);
int vmm_temporal_block_bf16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed,
                              const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                              const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                              float q_scale, float eps, vmm_stream_t stream);
/* the "bf16" throughput mode of the same block: identical arguments and packed weights, one matrix pass per product on bf16-rounded operands
 * (LayerNorm, softmax, rotary, residual in fp32) */
int vmm_temporal_block_bf16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed,
                              const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                              const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                              float q_scale, float eps, vmm_stream_t stream);

/* ---- K12: mid-level spatial softmax attention per frame (vddp.py:687-689): n = HW queries, keys = [tokens | HW].
 * tok_per_frame = 1: frame t sees only token t (vddp.py:459-462); 0: all ntok tokens. */
int vmm_spatial_attention(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                          int32_t tok_per_frame, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads,
                          int32_t dh, float* lse /* or NULL */, vmm_stream_t stream);

/* the same on the split-bf16 matrix cores (flash-attention forward, one wave per 32 queries; inference: no lse); returns 1 (nothing
 * launched) unless dh == 32 and ntok <= 32 */
int vmm_spatial_attention_bf16x3(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t tok_per_frame,
                                 float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);

/* ---- K9: spatial linear attention core (vddp.py:367-376), per (b*T frame, head):
 * ctx[d,e] = sum_n softmax_n(k)[d,n] * v[e,n]/HW over n = [tokens | HW pixels];  out[n, e] = sum_d ctx[d,e]*softmax_d(q[n,:])*scale */
int vmm_linattn_context(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t B, int32_t T,
                        int32_t HW, int32_t heads, int32_t dh, int32_t nsplit, float* part /* [B*T*heads*nsplit][dh*dh+2*dh] */,
                        float* ctx /* [B*T*heads][dh*dh] */, float* kstat /* [B*T*heads][2*dh] (max | 1/sum) or NULL */,
                        vmm_stream_t stream);
/* vmm_linattn_context with pass 1 (the per-slice partial contexts) on the split-bf16 matrix cores: one wave per (frame, head, slice),
 * operands loaded transposed, online softmax over the positions per key feature (inference); vmm_linattn_partial_bf16x3 is that pass alone
 * (rows_per_split a multiple of 32, the record layout [dh*dh + 2*dh] of `part`) */
int vmm_linattn_context_bf16x3(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t B, int32_t T,
                               int32_t HW, int32_t heads, int32_t dh, int32_t nsplit, float* part, float* ctx, float* kstat,
                               vmm_stream_t stream);
int vmm_linattn_partial_bf16x3(const float* qkv, int32_t ldqkv, int32_t frames, int32_t HW, int32_t heads, int32_t nsplit,
                               int32_t rows_per_split, float* part, vmm_stream_t stream);
int vmm_linattn_apply(const float* qkv, int32_t ldqkv, const float* ctx, float* out, int32_t ldo, int32_t B, int32_t T,
                      int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
/* the path vmm_linattn_apply takes for heads % 4 == 0 (returns 1 and launches nothing otherwise): out rows = softmax_d(q) scale . ctx on
 * the fp32 matrix cores, ctx^T resident as the MFMA "A" operand */
int vmm_linattn_apply_mfma(const float* qkv, int32_t ldqkv, const float* ctx, float* out, int32_t ldo, int32_t frames, int32_t HW,
                           int32_t heads, float scale, vmm_stream_t stream);

/* ---- K15: batched tiny dense layers (time_mlp, sign_emb, cond_token_to_hidden, ResnetBlock.mlp, to_k/to_v on tokens;
 * vddp.py:290-293,322-323,416-417,637-661).  One launch runs `njobs` independent jobs:
 * Y[r, o] = act_out( sum_k act_in(X[r,k]) * W[o,k] + b[o] ) (+ add[r,o]);  W is the torch (out,in) layout. */
typedef struct vmm_dense_job {
  const float* x; const float* w; const float* b; const float* add; float* y;
  int32_t rows, K, N, ldx, ldy, ldadd;
  int32_t act_in, act_out; /* 0 none, 1 SiLU, 2 GELU(erf) */
} vmm_dense_job;
/* max_n = max over jobs of N (one wave per job and output column; it reads its weight row once and walks the rows eight at a time) */
int vmm_dense_batched(const vmm_dense_job* jobs_dev, int32_t njobs, int32_t max_n, vmm_stream_t stream);

/* sinusoidal timestep embedding (vddp.py:139-151): out[b] = [sin(t*f_i) | cos(t*f_i)], f_i = exp(-i*ln(1e4)/(half-1)) */
int vmm_sinusoidal_embed(const int64_t* t, int32_t B, int32_t dim, float neg_step /* -ln(1e4)/(half-1) */, float* out,
                         vmm_stream_t stream);
/* per-frame conditioning (vddp.py:751-784): tokens[b,f,:] = cond[b,f]*w + bias; pooled = mean_f; then CFG replacement by the
 * null token where mask[b] != 0. */
int vmm_cond_tokens(const float* cond, const float* w, const float* bias, const float* null_token, const uint8_t* mask,
                    int32_t B, int32_t F, int32_t D, float* tokens, float* pooled, vmm_stream_t stream);
/* row-wise LayerNorm with affine (nn.LayerNorm, vddp.py:657) */
int vmm_rows_layernorm_affine(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t D, float eps,
                              vmm_stream_t stream);
/* out[b,:] = mask[b] ? null[:] : x[b,:]   then (+ add[b,:])   (vddp.py:784-788) */
int vmm_select_add(const float* x, const float* null_row, const uint8_t* mask, const float* add, float* out, int32_t B,
                   int32_t D, vmm_stream_t stream);
/* cond_to_time = 'concat' (vddp.py:788-789): out[b, :D] = t[b, :], out[b, D:2D] = mask[b] ? null[:] : x[b, :] */
int vmm_select_concat(const float* x, const float* null_row, const uint8_t* mask, const float* t, float* out, int32_t B, int32_t D,
                      vmm_stream_t stream);
/* focus_present_mask (vddp.py:431, 438-443, 514-524; Unet3D.forward's `focus_present_mask` / `prob_focus_present`): a sample that focuses on the
 * present attends to its own frame only, i.e. its attention output is its value row.  Row patches around the unchanged temporal attention
 * kernels; sample of a row = row / rows_per_sample; focus [B] bytes; ncols, lda, ldb multiples of 4:
 *   mode 0: b[row] = a[row] where focus        (forward: a = v third of the qkv rows, b = attention output)
 *   mode 1: b[row] = focus ? 0 : a[row]         (backward: the dO the core's backward sees)
 *   mode 2: b[row] += a[row] where focus        (backward: dv += dO) */
int vmm_focus_rows(int32_t mode, const float* a, int32_t lda, float* b, int32_t ldb, const uint8_t* focus, int32_t B, int32_t rows_per_sample,
                   int32_t ncols, vmm_stream_t stream);
/* cond_att_GRU (vddp.py:546-549, 567-571, 769-770: SignalEmbedding('GRU') = nn.GRU(1, cond_dim, num_layers=3, batch_first=True) over the signal):
 * the recurrence of ONE layer.  gi [B][L][3H] = W_ih x_t + b_ih for every step (a batched dense launch), whh_t = W_hh transposed [H][3H], bhh [3H];
 * y [B][L][H] = the layer's states; training also keeps hprev [B][L][H] (the state BEFORE each step) and gates [B][L][4][H] = (r, z, n, W_hn h + b_hn)
 * (both may be NULL).  Gate order r | z | n as in torch.  vmm_tokens_select: tokens[b, n, :] = mask[b] ? null_token[n, :] : g[b, n, :]. */
int vmm_gru_recurrent(const float* gi, const float* whh_t, const float* bhh, float* y, float* hprev, float* gates, int32_t B, int32_t L, int32_t H,
                      vmm_stream_t stream);
int vmm_tokens_select(const float* g, const float* null_token, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* tokens, vmm_stream_t stream);
/* rotate token keys for temporal attention: ek[b, n, h*dh + d] with position n (vddp.py:470-471) */
int vmm_rotary_rows(float* x, const float* rot_tab, int32_t B, int32_t N, int32_t heads, int32_t dh, vmm_stream_t stream);
/* relative position bias (vddp.py:70-108): embedding gather through the INTEGER T5 bucket table [n*n] that the host computes
 * once per n (bit-exact integer arithmetic, videometamaterials_amd/hostmath.py) -> out [heads][n][n] */
int vmm_relpos_bias(const float* emb /* [num_buckets][heads] */, const int32_t* buckets, int32_t n, int32_t heads, float* out,
                    vmm_stream_t stream);

/* SignalEmbedding 'CNN' stage (vddp.py:553-561): y = SiLU(Conv1d(k=4,s=2,p=1)(x)), x (B,Cin,Lin) -> y (B,Cout,Lin/2) */
int vmm_conv1d_k4s2_silu(const float* x, const float* w /* [Cout][Cin][4] */, const float* bias, float* y, int32_t B, int32_t Cin,
                         int32_t Cout, int32_t Lin, vmm_stream_t stream);
/* tokens[b,n,:] = mask[b] ? null_token[n,:] : hidden[b,:] (vddp.py:767,774-777) */
int vmm_tokens_from_hidden(const float* hidden, const float* null_token, const uint8_t* mask, int32_t B, int32_t N, int32_t D,
                           float* tokens, vmm_stream_t stream);

/* ---- layout edges: NCTHW (reference API) <-> channels-last rows ---- */
int vmm_ncthw_to_rows(const float* x, int32_t B, int32_t C, int32_t T, int32_t HW, float* rows, int32_t ld /* >= C, pad = 0 */,
                      vmm_stream_t stream);
int vmm_rows_to_ncthw(const float* rows, int32_t ld, int32_t B, int32_t C, int32_t T, int32_t HW, float* x, vmm_stream_t stream);
/* final 1x1x1 conv to a few output channels, written straight to NCTHW (vddp.py:708) */
int vmm_pointwise_to_ncthw(const float* rows, int32_t ld, int32_t Cin, const float* w /* [Cout][Cin] */, const float* bias,
                           int32_t B, int32_t Cout, int32_t T, int32_t HW, float* out, vmm_stream_t stream);

/* ---- K17/K18/K20: diffusion-step arithmetic (vddp.py:920-963,1036-1060) ---- */
/* x_t = a[t_b]*x0 + s[t_b]*noise  (q_sample, vddp.py:1036-1042); x0n = x*2-1 when normalize != 0 (vddp.py:1066,1109) */
int vmm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1macp,
                 int32_t normalize, float* out, int32_t B, int64_t per_sample, vmm_stream_t stream);
/* eps = null + (cond-null)*w (vddp.py:728) [null may be NULL => eps = cond]; x0 = c_recip[t]*x - c_recipm1[t]*eps (920-924);
 * writes x0 and |x0| (for the quantile). */
int vmm_predict_x0(const float* x, const float* eps_cond, const float* eps_null, float w, const int64_t* t,
                   const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, float* x0, float* absx0, int32_t B,
                   int64_t per_sample, vmm_stream_t stream);
/* per-sample linear-interpolated quantile of non-negative values (torch.quantile semantics, vddp.py:941-945):
 * exact radix select of order statistics k_lo and k_lo+1, s = lerp(v_lo, v_hi, frac), then max(s, floor_min). */
int vmm_quantile_rows(const float* absx, int32_t B, int64_t n, int64_t k_lo, float frac, float floor_min, float* s_out,
                      uint32_t* scratch /* [B][4112] */, vmm_stream_t stream);
/* clip_mode 0: x0c = x0; 1: clamp(x0,-1,1); 2: clamp(x0,-s[b],s[b])/s[b] (vddp.py:938-951);
 * mean = c1[t]*x0c + c2[t]*x (926-930); out = mean + [t>0]*exp(0.5*logvar[t])*noise (960-963), or just mean when noise == NULL. */
int vmm_posterior_step(const float* x0, const float* x, const float* noise, const float* s, const int64_t* t,
                       const float* coef1, const float* coef2, const float* logvar, int32_t clip_mode, float* out, int32_t B,
                       int64_t per_sample, vmm_stream_t stream);
/* the same step with the noise generated in the kernel (torch.randn_like of vddp.py:960 without the 4.9 MB noise tensor):
 * Philox4x32-10 keyed by rng_seed[0] (a 64-bit seed in device memory, drawn once per sample() call from torch's device generator), counter
 * = (element group, t[b], b), Box-Muller normals.  per_sample % 4 == 0, 16-byte aligned pointers.  t_next != NULL: t_next[b] = t[b] - 1
 * (the next replay of a captured step reads its timestep from there, vmm_step_inputs). */
int vmm_posterior_step_rng(const float* x0, const float* x, const int64_t* rng_seed, const float* s, const int64_t* t, const float* coef1,
                           const float* coef2, const float* logvar, int32_t clip_mode, float* out, int32_t B, int64_t per_sample,
                           int64_t* t_next, vmm_stream_t stream);
/* inputs of a captured sampling step in one launch: t[b] = time_in[b] = time_in[B + b] = t_src[b]; x_in[0 .. n) = img (n floats, a multiple
 * of 4).  copy_second_half bit 0: x_in[n .. 2n) = img too (guided plans of 2B rows that are not mirrored); bit 1: the plan has B rows (the
 * unguided step, guidance_scale == 1): time_in[B + b] is not written */
int vmm_step_inputs(const float* img, const int64_t* t_src, float* x_in, int32_t copy_second_half, int64_t* t, int64_t* time_in, int32_t B,
                    int64_t n, vmm_stream_t stream);
/* one DDIM step (vddp.py:986-1018) for the captured sampler: eps = null + (cond - null) w (eps_null may be NULL), x0 = c_recip[t] x - c_recipm1[t] eps,
 * out = coef[t][0] x0 + coef[t][1] eps + coef[t][2] noise (coef [T][4] made on the host: sqrt(alpha_next), c, sigma, last-step flag -> out = x0);
 * noise from the in-kernel Philox generator (only when sigma != 0); t_next[b] = next_of[t[b]].  per_sample % 4 == 0, 16-byte aligned pointers. */
int vmm_ddim_step_rng(const float* x, const float* eps_cond, const float* eps_null, float w, const int64_t* t, const float* c_recip,
                      const float* c_recipm1, const float* coef, const int64_t* next_of, const int64_t* rng_seed, float* out, int32_t B,
                      int64_t per_sample, int64_t* t_next, vmm_stream_t stream);
/* sum |a-b| or (a-b)^2 -> out[0] (fp64 accumulate, zeroed by the call); sign/diff for backward (vddp.py:1053-1056) */
int vmm_loss_reduce(const float* a, const float* b, int64_t n, int32_t squared, double* acc, float* out_mean,
                    vmm_stream_t stream);
/* classifier-free guidance: out = null + (cond - null) * w (vddp.py:728) */
int vmm_cfg_combine(const float* eps_cond, const float* eps_null, float w, float* out, int64_t n, vmm_stream_t stream);
/* out = a*x + b*y + c*z + d (y, z optional): DDIM update (vddp.py:1014-1016), un/normalize_img (1109-1113) */
int vmm_lincomb(const float* x, const float* y, const float* z, float a, float b, float c, float d, float* out, int64_t n,
                vmm_stream_t stream);

/* out_a[i] = out_b[i] = x[i] (n a multiple of 4, pointers 16-byte aligned): classifier-free guidance runs both branches as ONE batch whose
 * halves carry the same x; everything the network computes before the conditioning enters (the stem and init_temporal_attn, vddp.py:739-742)
 * is computed once for one half and handed to both */
int vmm_copy2(const float* x, float* out_a, float* out_b, int64_t n, vmm_stream_t stream);

/* ================================ training path (autograd of the calls above) ================================ */
/* GroupNorm(+FiLM)+SiLU backward: dz = grad of z = silu(a*h + b'); writes dh (= or +=), accumulates dgamma/dbeta [C],
 * writes dfilm [B][ldfilm] (scale | shift) when given.  stats = (mean, rstd) from vmm_groupnorm_coef. */
int vmm_groupnorm_bwd(const float* dz, int32_t lddz, const float* h, int32_t ldh, const float* coef, const float* stats,
                      const float* gamma, const float* beta, const float* film, int32_t ldfilm, int32_t B, int32_t rows_per_sample,
                      int32_t C, int32_t G, float* scratch /* vmm_groupnorm_bwd_scratch() floats */, float* dh, int32_t lddh, int32_t accumulate,
                      float* dgamma, float* dbeta, float* dfilm, vmm_stream_t stream);
/* floats of scratch vmm_groupnorm_bwd needs: one (P1, P2) partial row of C channels per workgroup of its reduce sweep + the group means */
int64_t vmm_groupnorm_bwd_scratch(int32_t B, int32_t rows_per_sample, int32_t C, int32_t G);
/* scratch: NULL (dgamma by atomics) or VMM_LN_BWD_MAX_BLOCKS * C floats (one partial row per workgroup + vmm_sum_partials) */
#define VMM_LN_BWD_MAX_BLOCKS 2048
int vmm_channel_layernorm_bwd(const float* x, int32_t ldx, const float* gamma, const float* dy, int32_t lddy, float* dx, int32_t lddx,
                              int32_t accumulate, float* dgamma, int64_t rows, int32_t C, float eps, float* scratch, vmm_stream_t stream);
/* softmax attention backward (mode 0 temporal, 1 mid spatial); qkv/out/lse as saved by the forward (q scaled+rotated, k rotated);
 * writes dqkv (gradient of the raw to_qkv output: rotation and q-scale undone), accumulates dek/dev [B][ntok][heads*dh] and
 * dbias [heads][T][T] (+=; caller zeroes them); dbuf = scratch of vmm_attention_bwd_scratch(...) floats. */
int vmm_attention_bwd(int32_t mode, const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                      int32_t tok_per_frame, const float* bias, int32_t bias_on_cond, const float* out, const float* dout, int32_t ldo,
                      const float* lse, const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev, float* dbias,
                      float* dbuf, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
/* mode-0 fast path that vmm_attention_bwd takes where it applies (heads = 8, dh = 32, T <= 16, ntok <= 16; returns 1 and launches
 * nothing otherwise): one workgroup per pixel stages the T rows once in LDS -- every qkv / dout / out element is read once */
int vmm_temporal_attention_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                               int32_t bias_on_cond, const float* out, const float* dout, int32_t ldo, const float* lse,
                               const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev, float* dbias, float* scratch,
                               int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
/* floats of scratch (dbuf) vmm_attention_bwd needs: rows * heads, or the fast path's per-workgroup partials if that is larger */
int64_t vmm_attention_bwd_scratch(int32_t mode, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t ntok);
/* linear attention backward; ctx and kstat (per (frame, head): max[32] | 1/sum[32] of the key softmax) saved by the forward */
int vmm_linattn_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* ctx,
                    const float* kstat, const float* dout, int32_t lddo, float* dctx /* [B*T*heads][32*32] scratch */, float* dqkv,
                    float* dek, float* dev, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream);
/* row pass of vmm_linattn_bwd on the fp32 matrix cores (heads % 4 == 0; returns 1 and launches nothing otherwise): dqkv rows from the
 * saved qkv, dout, the frame's ctx / dctx [frames*heads][32*32] and kstat */
int vmm_linattn_bwd_rows_mfma(const float* qkv, int32_t ldqkv, const float* dout, int32_t lddo, const float* ctx, const float* dctx,
                              const float* kstat, float* dqkv, int32_t frames, int32_t HW, int32_t heads, float scale, vmm_stream_t stream);
/* ---- backward of the FUSED attention blocks with recomputation (autograd of vddp.py:313-378 and 396-535 inside Residual(PreNorm(.))).
 * The training forward is vmm_temporal_block_bf16x3 / vmm_linattn_block_bf16x3: no qkv rows, attention outputs or softmax statistics are
 * stored.  The backward kernels re-form q, k, v and the probabilities from x on chip, take dO = dOut . W_out on chip, run the core's
 * backward on the split-bf16 matrix cores and write
 *   dqkv      rows x 768: gradient of the RAW to_qkv output (rotary and q-scale undone) -> vmm_qkv_bwd_bf16x3 (data + weight gradient)
 *   ln_stats  rows x 2: (mean, rstd) of the PreNorm LayerNorm of x, for the same kernel
 *   dwout_packed += [256][C] (packed-gradient layout [in][out] of to_out), dbias += [heads][T][T] (temporal), dek / dev += [B][ntok][256]
 * The residual path (dx += dOut) and the LayerNorm backward stay with the caller. */
typedef struct vmm_attn_block_bwd {
  const float* x; int32_t ldx;            /* block input, rows x C */
  const float* gamma;                     /* PreNorm LayerNorm weight [C] */
  const float* wqkv_frag;                 /* vmm_pack_weights fmt 2 of to_qkv (768, C) */
  const float* wout_t_frag;               /* fmt 2 of the (K = C, N = 256) operand: to_out (C, 256) read as [c][hd] (the data-gradient operand) */
  const float* ek; const float* ev; int32_t ntok;  /* conditioning keys / values [B][ntok][256] or NULL */
  const float* bias; int32_t bias_on_cond; /* temporal: relative-position bias [heads][T][T] */
  const float* rot_tab;                   /* temporal: [T][16][2] */
  const float* fwd_workspace;             /* linear: the workspace the forward vmm_linattn_block_bf16x3 call left (partials of the key softmax / context) */
  const float* dout; int32_t lddo;        /* gradient of the block output, rows x C */
  float* dqkv; int32_t lddqkv;            /* rows x 768 (`_bf16` / `_fp16` instances: 16-bit elements, see "the reduced-precision training leg" below) */
  float* ln_stats;
  float* dwout_packed;
  float* dbout;                           /* linear: += [C] gradient of the to_out bias, or NULL */
  float* dbias;
  float* dek; float* dev;
  float* workspace;                       /* vmm_*_block_bwd_workspace(...) floats */
  int32_t B, T, HW, C, heads;
  float q_scale, eps;
} vmm_attn_block_bwd;
/* floats of workspace; 0 outside the envelope (C == 64, heads == 8, dim_head == 32, T <= 16, ntok <= 16, even HW) */
int64_t vmm_temporal_block_bwd_workspace(int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, int32_t ntok);
/* returns 1 (nothing launched) outside the envelope */
int vmm_temporal_block_bwd_bf16x3(const vmm_attn_block_bwd* d, vmm_stream_t stream);
/* the linear-attention block (vddp.py:313-378): d->fwd_workspace = the workspace of the forward vmm_linattn_block_bf16x3 call on the same x
 * (its key-softmax partials and context fragments are read, not modified); envelope C == 64, heads == 8, dim_head == 32, HW % 32 == 0;
 * bias / bias_on_cond / rot_tab / dbias are ignored, dbout (+= [C]) receives the to_out bias gradient */
int64_t vmm_linattn_block_bwd_workspace(int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, int32_t ntok);
int vmm_linattn_block_bwd_bf16x3(const vmm_attn_block_bwd* d, vmm_stream_t stream);

/* ---- the reduced-precision training leg (`train_precision = "bf16"`; the counterpart of the reference's fp16 autocast, main.py:34): the same backward
 * kernels, same arguments, workspaces and envelopes as their _bf16x3 namesakes, with ONE matrix pass per product on the operands' bf16 roundings (fp32
 * accumulation; the second compilation of their sources with -DVMM_SINGLE_PASS=1).  The forward and data-gradient contractions of that leg are the
 * single-pass entry points of the sampling path (vmm_conv3x3_bf16, vmm_proj_bf16, vmm_temporal_block_bf16, vmm_linattn_block_bf16, ...).
 * ONE difference in the data they exchange (round 6): the gradient of the raw qkv rows -- `vmm_attn_block_bwd.dqkv` of the block backward, `g` of
 * vmm_qkv_bwd_* -- is rows of the operand's own 16-bit type in the single-pass instances (bf16 bits for `_bf16`, IEEE half for `_fp16`; `lddqkv` / `ldg`
 * count ELEMENTS, the pointers keep their `float*` spelling): it is a matrix operand of the to_qkv backward and nothing else, the producer's store
 * applies the rounding the consumer's operand conversion applied before, so the gradients are the same bit for bit and the 1.25 GB round trip per
 * 96 x 96 site is half the bytes.  The `_bf16x3` instances keep fp32 rows. */
int vmm_conv3x3_wgrad_bf16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                           vmm_stream_t stream);
int vmm_conv1x1_wgrad_bf16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                           vmm_stream_t stream);
int vmm_conv1x1_wgrad_bf16_ln(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* workspace, const float* ln_stats,
                              const float* ln_gamma, vmm_stream_t stream);
int vmm_qkv_bwd_bf16(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                     float* gy, int32_t ldgy, float* dw_packed, float* workspace, int64_t rows, int32_t C, int32_t Nq, vmm_stream_t stream);
int vmm_qkv_bwd_ln_bf16(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                        float* dx, int32_t lddx, int32_t accumulate, float* dgamma, float* dw_packed, float* workspace, int64_t rows, int32_t C,
                        int32_t Nq, vmm_stream_t stream);
int vmm_temporal_block_bwd_bf16(const vmm_attn_block_bwd* d, vmm_stream_t stream);
/* n (a multiple of 8) elements of such 16-bit rows widened to fp32, exactly: for the shapes the one-pass vmm_qkv_bwd_* does not take (its workspace query
 * returns 0), where the caller runs a separate weight- and data-gradient launch over fp32 rows (which round to the same 16 bits again) */
int vmm_dqkv_widen_bf16(const void* src, float* dst, int64_t n, vmm_stream_t stream);
/* the generic implicit GEMM of those legs (the layers no specialised kernel takes: to_qkv and its data gradient at C >= 128, res_conv, the transposed
 * convolutions' phases): vmm_conv_igemm_bf16x3 / _batched with ONE matrix pass; the same fmt-1 weight planes (the lo plane is not read) */
int vmm_conv_wgrad_tap_bf16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_stream_t stream);
int vmm_conv_igemm_bf16(const vmm_conv_desc* d, vmm_stream_t stream);
int vmm_conv_igemm_bf16_batched(const vmm_conv_desc* descs, int32_t n, vmm_stream_t stream);
int vmm_linattn_block_bwd_bf16(const vmm_attn_block_bwd* d, vmm_stream_t stream);

/* tiny dense layers: stage 1 writes g = dy*act_out'(z) over dy and dW/db (= or +=), stage 2 adds dx with atomics */
typedef struct vmm_dense_bwd_job {
  const float* x; const float* w; const float* b; float* dy; float* dx; float* dw; float* db;
  int32_t rows, K, N, ldx, lddy, lddx;
  int32_t act_in, act_out, accumulate;
} vmm_dense_bwd_job;
int vmm_dense_bwd_batched(const vmm_dense_bwd_job* jobs_dev, int32_t njobs, int32_t max_N, int32_t max_xunits, vmm_stream_t stream);
int vmm_cond_tokens_bwd(const float* cond, const uint8_t* mask, const float* dtokens, const float* dpooled, int32_t B, int32_t F, int32_t D,
                        float* dw, float* dbias, float* dnull_token, vmm_stream_t stream);
int vmm_rows_layernorm_affine_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t rows, int32_t D,
                                  float eps, vmm_stream_t stream);
int vmm_select_add_bwd(const float* dout, const uint8_t* mask, float* dx, float* dnull_row, float* dadd, int32_t B, int32_t D,
                       vmm_stream_t stream);
/* backward of vmm_select_concat: dout [B][2D]; ADDS into dx / dnull_row / dt (each may be NULL) */
int vmm_select_concat_bwd(const float* dout, const uint8_t* mask, float* dx, float* dnull_row, float* dt, int32_t B, int32_t D,
                          vmm_stream_t stream);
/* backward of vmm_gru_recurrent through time: dy [B][L][H] = gradient of the layer's states; whh = W_hh in torch layout (3H, H); WRITES dgi / dgh
 * [B][L][3H] (gradients of the input-side / hidden-side pre-activations: the weight, bias and input gradients are dense backward jobs over them).
 * vmm_tokens_select_bwd WRITES dg and ADDS into dnull_token. */
int vmm_gru_recurrent_bwd(const float* dy, const float* gates, const float* hprev, const float* whh, float* dgi, float* dgh, int32_t B, int32_t L, int32_t H,
                          vmm_stream_t stream);
int vmm_tokens_select_bwd(const float* dtokens, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* dg, float* dnull_token, vmm_stream_t stream);
int vmm_relpos_bias_bwd(const float* dbias, const int32_t* buckets, int32_t n, int32_t heads, float* demb, vmm_stream_t stream);
int vmm_tokens_from_hidden_bwd(const float* dtokens, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* dhidden,
                               float* dnull_token, vmm_stream_t stream);
int vmm_conv1d_k4s2_silu_bwd(const float* x, const float* w, const float* bias, const float* dy, float* dx, float* dw, float* db, int32_t B,
                             int32_t Cin, int32_t Cout, int32_t Lin, vmm_stream_t stream);
int vmm_pointwise_to_ncthw_bwd(const float* rows, int32_t ld, int32_t Cin, const float* w, const float* dout, int32_t B, int32_t Cout,
                               int32_t T, int32_t HW, float* drows, int32_t lddr, float* dw, float* db, vmm_stream_t stream);
/* data gradient of the stem (init_conv, vddp.py:600) into the NCTHW layout of the network input: dx[b, c, t, y, x] = sum_{kh, kw, co}
 * g[(b, t, y + k/2 - kh, x + k/2 - kw)][co] w[co][c][kh][kw]; g rows [B*T*H*W][Cout] (ldg), w in torch layout (Cout, Cx, 1, k, k); wrap_h / wrap_w: the
 * axis is periodic (padding_mode 'circular' / 'circular_1d', vddp.py:163-243: rows across the seam contribute), else zero padding.  Only
 * autograd users that ask for the gradient of the loss with respect to the input run it. */
int vmm_stem_conv_dgrad(const float* g, int32_t ldg, const float* w, float* dx, int32_t B, int32_t Cx, int32_t T, int32_t H, int32_t W, int32_t Cout,
                        int32_t k, int32_t wrap_h, int32_t wrap_w, vmm_stream_t stream);
/* d/dpred of mean|noise-pred| (or squared error) times the upstream scalar *gscale (NULL = 1) (vddp.py:1053-1056) */
int vmm_loss_grad(const float* noise, const float* pred, int64_t n, int32_t squared, const float* gscale, float* dpred, vmm_stream_t stream);

/* ---- K21: multi-tensor Adam (torch.optim.Adam defaults, vddp.py:1481,1633) and EMA (vddp.py:116-129) over a device job table */
typedef struct vmm_optim_job { float* p; const float* g; float* m; float* v; int64_t n; } vmm_optim_job;
int vmm_adam_step(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float lr, float beta1, float beta2, float eps, int32_t step,
                  float grad_scale, vmm_stream_t stream);
/* ---- loss scaling for the fp16-operand training leg, entirely on the device: what Accelerate(mixed_precision='fp16') wraps around the reference's step
 * (main.py:34; vddp.py:1629-1633 accelerator.backward / opt.step) -- torch.cuda.amp.GradScaler's state machine (scale 2^16, x backoff on an inf / nan
 * gradient with the optimiser step skipped, x growth after `interval` clean steps), no host round trip.
 * state: 5 floats -- [0] scale, [1] growth tracker, [2] found_inf of the step in flight, [3] steps skipped, [4] optimiser steps taken.
 *   vmm_scaler_init          state = {init_scale, 0, 0, 0, 0}
 *   vmm_loss_grad(..., gscale = state, ...)   the loss gradient times the scale (its upstream scalar)
 *   vmm_grad_nonfinite       state[2] = 1 when any of g[0 .. n) is inf / nan (g 16-byte aligned: the flat gradient buffer, after the all-reduce)
 *   vmm_adam_step_scaled     vmm_adam_step with grad_scale = extra_scale / state[0] and bias corrections for step state[4] + 1; a no-op when state[2] != 0
 *   vmm_scaler_update        GradScaler.update(): found_inf ? (scale *= backoff, tracker = 0, skipped += 1) : (steps += 1, tracker += 1,
 *                            tracker == interval ? scale *= growth, tracker = 0); found_inf = 0 */
int vmm_scaler_init(float* state, float init_scale, vmm_stream_t stream);
int vmm_grad_nonfinite(const float* g, int64_t n, float* state, vmm_stream_t stream);
int vmm_adam_step_scaled(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float lr, float beta1, float beta2, float eps, float extra_scale,
                         const float* state, vmm_stream_t stream);
int vmm_scaler_update(float* state, float growth, float backoff, int32_t interval, vmm_stream_t stream);
/* m (EMA weights) = copy_only ? p : beta*m + (1-beta)*p */
int vmm_ema_step(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float beta, int32_t copy_only, vmm_stream_t stream);

/* ---- geometry extraction (INT, bit-exact): the topology rule of Trainer.save_preds (vddp.py:1890-1913) followed by
 * clean_pred (src/utils.py:32-82), one workgroup per sample.  videos = the sampler's fp32 (N, C, T, P, P) output;
 * out[n][x * P/2 + y] in {0, 1}, the rows of the reference's geometries.csv.  lagrangian != 0 and T > 1: a pixel of the mirrored
 * upper-left quarter is void iff channel 1 (u_2) is within torch.isclose(atol = 0.02) of zero_u_2 in every frame; otherwise the
 * bottom-left quarter of channel 0 / frame 0 binarised at 0.5.  Then: transpose, drop pixels whose four (existing) neighbours are
 * empty, keep the largest 4-connected component (networkx's iteration order breaks ties).  P even, P <= 192. */
int vmm_extract_geometry(const float* videos, int32_t N, int32_t C, int32_t T, int32_t P, int32_t lagrangian, float zero_u_2,
                         int32_t* out, vmm_stream_t stream);

/* ---- training-sample assembly: Dataset.__getitem__ of the reference (vddp.py:1304-1397) for a minibatch gathered by
 * index from a dataset resident in HBM as decoded bytes.  frames [N][n_fields][f][HW] u8, field 0 = the topology; index [B] = dataset
 * rows; chan_src [nch]: source field of output channel c (| 0x100: pass the bytes / 255 through, i.e. the topology channel itself);
 * coef [N][nch][4] = (un_range, un_lo, g_lo, g_range) as fp32:  v = u8 / 255;  v = v * un_range + un_lo;  v = 0 where topology == 0;
 * out = (v - g_lo) / g_range, each operation rounded to fp32 in that order (bit-exact against the reference's torch expressions).
 * out [B][nch][T_out][HW] fp32; frames t >= f are zero (cast_num_frames pads, vddp.py:1115), t >= T_out dropped. */
int vmm_fields_to_samples(const uint8_t* frames, int32_t n_fields, int32_t f, int64_t HW, const int32_t* index, int32_t B,
                          const int32_t* chan_src, const float* coef, int32_t nch, int32_t T_out, float* out, vmm_stream_t stream);

/* ---- the fp16-operand training leg (`Unet3D.train_precision = "fp16"`): the reference's own training arithmetic -- Accelerate(mixed_precision='fp16'),
 * main.py:34; torch autocast runs every convolution / Linear / einsum of vddp.py:1044-1060 on fp16 operands with fp32 accumulation.  The single-pass
 * instances of the kernels above on IEEE-half operands (v_mfma_f32_32x32x16_f16; activations rounded to fp16 where the `_bf16` entry points round to bf16,
 * weight operands = vmm_pack_weights fmt | 16: fp16 planes), fp32 feature maps / master weights / accumulation / norms / softmax as everywhere else.
 * Same arguments, envelopes and return codes as their `_bf16` twins (fp32-stored maps only).  Loss scaling: vmm_scaler_* / vmm_adam_step_scaled. */
int vmm_conv3x3_fp16(const vmm_conv_desc* d, vmm_stream_t stream);
int vmm_conv_s2_acc_fp16(const float* x, int32_t ldx, const float* w_packed, const float* bias, const float* res, int32_t ldres, float* out,
                           int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                           int32_t n_tickets, vmm_stream_t stream);
int vmm_linattn_block_fp16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                             const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                             int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream);
int vmm_temporal_block_fp16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed,
                              const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                              const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                              float q_scale, float eps, vmm_stream_t stream);
int vmm_conv3x3_wgrad_fp16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                           vmm_stream_t stream);
int vmm_conv1x1_wgrad_fp16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                           vmm_stream_t stream);
int vmm_conv1x1_wgrad_fp16_ln(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* workspace, const float* ln_stats,
                              const float* ln_gamma, vmm_stream_t stream);
int vmm_qkv_bwd_fp16(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                     float* gy, int32_t ldgy, float* dw_packed, float* workspace, int64_t rows, int32_t C, int32_t Nq, vmm_stream_t stream);
int vmm_qkv_bwd_ln_fp16(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                        float* dx, int32_t lddx, int32_t accumulate, float* dgamma, float* dw_packed, float* workspace, int64_t rows, int32_t C,
                        int32_t Nq, vmm_stream_t stream);
int vmm_temporal_block_bwd_fp16(const vmm_attn_block_bwd* d, vmm_stream_t stream);
int vmm_linattn_block_bwd_fp16(const vmm_attn_block_bwd* d, vmm_stream_t stream);
int vmm_dqkv_widen_fp16(const void* src, float* dst, int64_t n, vmm_stream_t stream);
/* (weights: vmm_pack_weights fmt 1 | 16 -- IEEE-half hi plane) */
int vmm_conv_wgrad_tap_fp16(const vmm_conv_desc* d, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_stream_t stream);
int vmm_conv_igemm_fp16(const vmm_conv_desc* d, vmm_stream_t stream);
int vmm_conv_igemm_fp16_batched(const vmm_conv_desc* descs, int32_t n, vmm_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
