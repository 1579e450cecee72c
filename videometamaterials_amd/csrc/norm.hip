// GroupNorm(+FiLM+SiLU) and channel LayerNorm for channels-last rows on gfx950.
// All of these are HBM-bound sweeps: 16-byte loads along channels, wave-shuffle reductions,
// fp64 only for the cross-block GroupNorm sums (one atomic pair per block and group).
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

// ---------------------------------------------------------------- GroupNorm statistics
// grid = (blocks_per_sample, B); block = 256 threads; each thread owns one float4 column group and strides over rows.
__global__ __launch_bounds__(256) void gn_stats_kernel(const float* __restrict__ x, int ldx, int rows_per_sample, int C, int G,
                                                        int rows_per_block, double* __restrict__ sums, float* __restrict__ part) {
  const int b = blockIdx.y;
  const int c4n = C >> 2;                 // float4 columns per row
  const int tid = threadIdx.x;
  const int col4 = tid % c4n;
  const int rlane = tid / c4n;            // row slot of this thread
  const int rslots = 256 / c4n;           // rows covered per pass (c4n <= 256)
  const int r_begin = blockIdx.x * rows_per_block;
  const int r_end = min(r_begin + rows_per_block, rows_per_sample);
  float s[4] = {0, 0, 0, 0}, q[4] = {0, 0, 0, 0};
  if (rlane < rslots) {
    const float* base = x + ((long long)b * rows_per_sample) * ldx + col4 * 4;
    for (int r = r_begin + rlane; r < r_end; r += rslots) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)r * ldx);
      s[0] += v.x; s[1] += v.y; s[2] += v.z; s[3] += v.w;
      q[0] += v.x * v.x; q[1] += v.y * v.y; q[2] += v.z * v.z; q[3] += v.w * v.w;
    }
  }
  // per-channel partials -> LDS -> per-group fp64
  __shared__ float sh_s[256 * 4];
  __shared__ float sh_q[256 * 4];
#pragma unroll
  for (int j = 0; j < 4; ++j) { sh_s[tid * 4 + j] = s[j]; sh_q[tid * 4 + j] = q[j]; }
  __syncthreads();
  const int Cg = C / G;
  if (tid < G) {
    double ds = 0.0, dq = 0.0;
    for (int c = tid * Cg; c < (tid + 1) * Cg; ++c) {
      const int c4 = c >> 2, j = c & 3;
      for (int rl = 0; rl < rslots; ++rl) {
        ds += (double)sh_s[(rl * c4n + c4) * 4 + j];
        dq += (double)sh_q[(rl * c4n + c4) * 4 + j];
      }
    }
    if (part) {  // one (sum, sum of squares) slot per workgroup: summed in fixed order by gn_coef_kernel (no zero-fill, no atomics)
      float* pp = part + (((long long)(b * G + tid)) * gridDim.x + blockIdx.x) * 2;
      pp[0] = (float)ds;
      pp[1] = (float)dq;
    } else {
      atomicAdd(&sums[(b * G + tid) * 2 + 0], ds);
      atomicAdd(&sums[(b * G + tid) * 2 + 1], dq);
    }
  }
}

// coef[b][c] = (a, b') ; one workgroup per (sample, group): the group's two moments, then one thread per channel.  The moments come
// from (in this order of precedence) the per-workgroup fp32 partial sums the producing convolution left behind (conv3x3_bf16x3.hip),
// a direct fixed-order reduction of the group's slice of x (small layers: saves the statistics launch), or the fp64 sums of
// vmm_groupnorm_stats.
__global__ __launch_bounds__(256) void gn_coef_kernel(const double* __restrict__ sums, double inv_count, float eps, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const float* __restrict__ film, int ldfilm, int B, int C, int G,
                                                      float* __restrict__ coef, float* __restrict__ stats_out, const float* __restrict__ partials,
                                                      int n_contrib, const float* __restrict__ x, int ldx, int rows_per_sample) {
  __shared__ double red[2][4];
  const int b = blockIdx.x / G, g = blockIdx.x - b * G;
  const int tid = threadIdx.x;
  const int cpg = C / G;
  double s1 = 0.0, s2 = 0.0;
  if (partials || x) {
    if (partials) {
      const float* pp = partials + (long long)(b * G + g) * n_contrib * 2;
      for (int k = tid; k < n_contrib; k += 256) { s1 += (double)pp[2 * k]; s2 += (double)pp[2 * k + 1]; }
    } else {
      const int c4n = cpg >> 2;                       // float4 columns of the group
      const int col4 = tid % c4n, rslot = tid / c4n, rslots = 256 / c4n;
      if (rslot < rslots) {
        const float* base = x + ((long long)b * rows_per_sample) * ldx + g * cpg + col4 * 4;
        float a1 = 0.f, a2 = 0.f;
        int run = 0;
        for (int r = rslot; r < rows_per_sample; r += rslots) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(base + (long long)r * ldx);
          a1 += (v.x + v.y) + (v.z + v.w);
          a2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          if (++run == 64) { s1 += (double)a1; s2 += (double)a2; a1 = a2 = 0.f; run = 0; }  // bounded fp32 runs
        }
        s1 += (double)a1; s2 += (double)a2;
      }
    }
    // (wave shuffles, then the four wave sums through LDS: one barrier instead of the nine of a 256-wide LDS tree -- the launch is a pure latency
    // chain between a convolution and its consumer, 38 times per denoiser pass)
    s1 = wave_sum_d(s1); s2 = wave_sum_d(s2);
    if ((tid & 63) == 0) { red[0][tid >> 6] = s1; red[1][tid >> 6] = s2; }
    __syncthreads();
    s1 = (red[0][0] + red[0][1]) + (red[0][2] + red[0][3]);
    s2 = (red[1][0] + red[1][1]) + (red[1][2] + red[1][3]);
  } else {
    s1 = sums[(b * G + g) * 2];
    s2 = sums[(b * G + g) * 2 + 1];
  }
  const double mean = s1 * inv_count;
  double var = s2 * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float rstd = (float)(1.0 / sqrt(var + (double)eps));
  const float meanf = (float)mean;
  for (int cl = tid; cl < cpg; cl += 256) {
    const int c = g * cpg + cl;
    float a = rstd * gamma[c];
    float bb = beta[c] - meanf * a;
    if (film) {
      const float sc = film[(long long)b * ldfilm + c] + 1.0f;
      const float sh = film[(long long)b * ldfilm + C + c];
      a *= sc;
      bb = bb * sc + sh;
    }
    coef[((long long)b * C + c) * 2 + 0] = a;
    coef[((long long)b * C + c) * 2 + 1] = bb;
  }
  if (stats_out && tid == 0) {
    stats_out[(b * G + g) * 2 + 0] = meanf;
    stats_out[(b * G + g) * 2 + 1] = rstd;
  }
}

template <typename ST>
__global__ __launch_bounds__(256) void affine_silu_kernel(const ST* __restrict__ x, int ldx, const float* __restrict__ coef,
                                                          const ST* __restrict__ res, int ldres, ST* __restrict__ y, int ldy,
                                                          long long rows, int rows_per_sample, int C) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const int b = (int)(r / rows_per_sample);
    const f32x4 v = ld4(x + r * ldx + c);
    const float* cf = coef + ((long long)b * C + c) * 2;
    const f32x4 c0 = *reinterpret_cast<const f32x4*>(cf);
    const f32x4 c1 = *reinterpret_cast<const f32x4*>(cf + 4);
    f32x4 o;
    o.x = silu_f(v.x * c0.x + c0.y);
    o.y = silu_f(v.y * c0.z + c0.w);
    o.z = silu_f(v.z * c1.x + c1.y);
    o.w = silu_f(v.w * c1.z + c1.w);
    if (res) {
      const f32x4 rr = ld4(res + r * ldres + c);
      o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
    }
    st4(y + r * ldy + c, o);
  }
}

// storage conversion of a dense activation (fp32 <-> bf16): only where a kernel without a bf16-storage instance meets one with
template <typename SS, typename DS>
__global__ __launch_bounds__(256) void convert_act_kernel(const SS* __restrict__ src, DS* __restrict__ dst, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) st4(dst + 4 * i, ld4(src + 4 * i));
}

// ---------------------------------------------------------------- last block's output pass + final 1x1 convolution
// out[b][co][t][hw] = bias[co] + sum_c w[co][c] * (silu(x[r][c] * a + b') + res[r][c]),  C == 64, Cout <= 4 (vddp.py:311 then final_conv.1,
// vddp.py:729): the block's output never goes to memory.  16 lanes per row (one float4 each, 256-byte row reads), a wave walks 64
// consecutive rows, the per-row dot products are reduced inside the DPP row and leave through LDS so that lane r writes row r (256-byte
// runs of every output plane).
template <typename ST>
__global__ __launch_bounds__(256) void affine_silu_pointwise_kernel(const ST* __restrict__ x, int ldx, const float* __restrict__ coef,
                                                                    const ST* __restrict__ res, int ldres, int rows_per_sample,
                                                                    const float* __restrict__ w, const float* __restrict__ bias, int Cout,
                                                                    int T, int HW, float* __restrict__ out, long long nrows) {
  __shared__ float sm[4][64][4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int sub = lane & 15, grp = lane >> 4;
  const int c = sub * 4;
  f32x4 wv[4];
#pragma unroll
  for (int co = 0; co < 4; ++co) wv[co] = co < Cout ? *reinterpret_cast<const f32x4*>(w + co * 64 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
  const long long r0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (r0 >= nrows) return;
#pragma unroll 4
  for (int it = 0; it < 16; ++it) {
    const long long r = r0 + it * 4 + grp;
    float d[4] = {0.f, 0.f, 0.f, 0.f};
    if (r < nrows) {
      const int b = (int)(r / rows_per_sample);
      const f32x4 v = ld4(x + r * ldx + c);
      const float* cf = coef + ((long long)b * 64 + c) * 2;
      const f32x4 c0 = *reinterpret_cast<const f32x4*>(cf), c1 = *reinterpret_cast<const f32x4*>(cf + 4);
      f32x4 o = {silu_f(v.x * c0.x + c0.y), silu_f(v.y * c0.z + c0.w), silu_f(v.z * c1.x + c1.y), silu_f(v.w * c1.z + c1.w)};
      if (res) {
        const f32x4 rr = ld4(res + r * ldres + c);
        o.x += rr.x; o.y += rr.y; o.z += rr.z; o.w += rr.w;
      }
#pragma unroll
      for (int co = 0; co < 4; ++co) d[co] = (o.x * wv[co].x + o.y * wv[co].y) + (o.z * wv[co].z + o.w * wv[co].w);
    }
#pragma unroll
    for (int co = 0; co < 4; ++co) {
      float v = d[co];
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
      v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
      d[co] = v;
    }
    if (sub == 0) *reinterpret_cast<f32x4*>(&sm[wave][it * 4 + grp][0]) = f32x4{d[0], d[1], d[2], d[3]};
  }
  __builtin_amdgcn_s_waitcnt(0xc07f);  // lgkmcnt(0): this wave's own LDS writes (the rows of a wave are private to it)
  __builtin_amdgcn_wave_barrier();
  const long long r = r0 + lane;
  if (r < nrows) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(&sm[wave][lane][0]);
    const float vv[4] = {v.x, v.y, v.z, v.w};
    const int hw = (int)(r % HW);
    const int t = (int)((r / HW) % T);
    const long long b = r / ((long long)HW * T);
    for (int co = 0; co < Cout; ++co) out[((b * Cout + co) * T + t) * HW + hw] = vv[co] + (bias ? bias[co] : 0.f);
  }
}

// ---------------------------------------------------------------- channel LayerNorm
// GS lanes cooperate on one row (GS = power of two <= 64); each lane strides float4s over C.
template <int GS>
__global__ __launch_bounds__(256) void chan_ln_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                      float* __restrict__ y, int ldy, long long rows, int C, float eps) {
  const int tid = threadIdx.x;
  const int sub = tid % GS;
  const long long row = ((long long)blockIdx.x * 256 + tid) / GS;
  const bool valid = row < rows;
  const float* xr = x + (valid ? row : 0) * ldx;
  constexpr int MAXV = 8;  // up to GS*4*MAXV channels held in registers (2048 at GS=64)
  f32x4 v[MAXV];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c = (j * GS + sub) * 4;
    v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (valid && c < C) v[j] = *reinterpret_cast<const f32x4*>(xr + c);
    s += v[j].x + v[j].y + v[j].z + v[j].w;
  }
  s = group_sum(s, GS);
  const float mean = s / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c = (j * GS + sub) * 4;
    if (c < C) {
      const float d0 = v[j].x - mean, d1 = v[j].y - mean, d2 = v[j].z - mean, d3 = v[j].w - mean;
      q += d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3;
    }
  }
  q = group_sum(q, GS);
  const float rstd = 1.0f / sqrtf(q / (float)C + eps);
  if (!valid) return;
  float* yr = y + row * ldy;
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c = (j * GS + sub) * 4;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
      f32x4 o;
      o.x = (v[j].x - mean) * rstd * g.x;
      o.y = (v[j].y - mean) * rstd * g.y;
      o.z = (v[j].z - mean) * rstd * g.z;
      o.w = (v[j].w - mean) * rstd * g.w;
      *reinterpret_cast<f32x4*>(yr + c) = o;
    }
  }
}

}  // namespace

static int gn_stats_blocks(int rows_per_sample, int C, int B) {
  const int rslots = 256 / (C >> 2);
  // enough blocks to fill the chip (>= ~2048 in total) while keeping >= 8 rows per thread slot
  int blocks = max(1, min(cdiv(rows_per_sample, rslots * 8), max(1, 2048 / max(B, 1))));
  const int rows_per_block = cdiv(rows_per_sample, blocks);
  return cdiv(rows_per_sample, rows_per_block);
}

extern "C" int vmm_groupnorm_stats(const float* x, int32_t ldx, int32_t B, int32_t rows_per_sample, int32_t C, int32_t G,
                                   double* sums, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || C > 1024 || C % G || G > 256 || (ldx & 3)) return -1;
  if (int rc = vmm_zero_async(sums, sizeof(double) * B * G * 2, s)) return rc;  // (a kernel, not a memset node: see vmm_common.h)
  const int blocks = gn_stats_blocks(rows_per_sample, C, B);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(blocks, B), dim3(256), 0, s, x, ldx, rows_per_sample, C, G, cdiv(rows_per_sample, blocks), sums, nullptr);
  VMM_LAUNCH_CHECK();
  return 0;
}

// The same pass leaving one fp32 (sum, sum of squares) pair per workgroup in part[B*G][n][2], n = vmm_groupnorm_stats_slots(...):
// vmm_groupnorm_coef(partials = part, n_contrib = n) then adds them in a fixed order -- bit-reproducible, no zero-fill, no atomics.
extern "C" int vmm_groupnorm_stats_slots(int32_t B, int32_t rows_per_sample, int32_t C) {
  if ((C & 3) || C > 1024 || rows_per_sample <= 0) return 0;
  return gn_stats_blocks(rows_per_sample, C, B);
}
extern "C" int vmm_groupnorm_stats_partials(const float* x, int32_t ldx, int32_t B, int32_t rows_per_sample, int32_t C, int32_t G, float* part,
                                            vmm_stream_t stream) {
  if ((C & 3) || C > 1024 || C % G || G > 256 || (ldx & 3)) return -1;
  const int blocks = gn_stats_blocks(rows_per_sample, C, B);
  hipLaunchKernelGGL(gn_stats_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, x, ldx, rows_per_sample, C, G, cdiv(rows_per_sample, blocks),
                     nullptr, part);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_groupnorm_coef(const double* sums, int64_t count_per_group, float eps, const float* gamma, const float* beta,
                                  const float* film, int32_t ldfilm, int32_t B, int32_t C, int32_t G, float* coef,
                                  float* stats_out, const float* partials, int32_t n_contrib, const float* x, int32_t ldx, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (C % G) return -1;
  const int cpg = C / G;
  if (x && !partials && ((cpg & 3) || (ldx & 3) || (cpg >> 2) > 256 || count_per_group % cpg)) return -1;
  hipLaunchKernelGGL(gn_coef_kernel, dim3(B * G), dim3(256), 0, s, sums, 1.0 / (double)count_per_group, eps, gamma,
                     beta, film, ldfilm, B, C, G, coef, stats_out, partials, n_contrib, x, ldx, (int)(count_per_group / cpg));
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_affine_silu(const float* x, int32_t ldx, const float* coef, const float* res, int32_t ldres, float* y,
                               int32_t ldy, int64_t rows, int32_t rows_per_sample, int32_t C, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || (ldx & 3) || (ldy & 3) || (res && (ldres & 3))) return -1;
  const long long total = rows * (C >> 2);
  const int blocks = (int)min((long long)cdiv(total, 256), 8192LL);
  hipLaunchKernelGGL(affine_silu_kernel<float>, dim3(blocks), dim3(256), 0, s, x, ldx, coef, res, ldres, y, ldy, (long long)rows,
                     rows_per_sample, C);
  VMM_LAUNCH_CHECK();
  return 0;
}

// the same pass over bf16-stored feature maps (x, res, y: bf16 bits; the coefficients stay fp32)
extern "C" int vmm_affine_silu_a16(const void* x, int32_t ldx, const float* coef, const void* res, int32_t ldres, void* y, int32_t ldy, int64_t rows,
                                   int32_t rows_per_sample, int32_t C, vmm_stream_t stream) {
  if (rows <= 0) return 0;
  if ((C & 3) || (ldx & 3) || (ldy & 3) || (res && (ldres & 3))) return -1;
  const long long total = rows * (C >> 2);
  const int blocks = (int)min((long long)cdiv(total, 256), 8192LL);
  hipLaunchKernelGGL(affine_silu_kernel<bf16s>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const bf16s*>(x), ldx, coef,
                     static_cast<const bf16s*>(res), ldres, static_cast<bf16s*>(y), ldy, (long long)rows, rows_per_sample, C);
  VMM_LAUNCH_CHECK();
  return 0;
}

// dst = src with the storage type changed (n elements, a multiple of 4; 16-byte / 8-byte aligned): src_bf16 / dst_bf16 = 0 fp32, 1 bf16
extern "C" int vmm_convert_act(const void* src, int32_t src_bf16, void* dst, int32_t dst_bf16, int64_t n, vmm_stream_t stream) {
  if (n <= 0) return 0;
  if ((n & 3) || src_bf16 == dst_bf16) return -1;
  const int blocks = (int)min((long long)cdiv(n / 4, 256), 8192LL);
  if (src_bf16)
    hipLaunchKernelGGL((convert_act_kernel<bf16s, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const bf16s*>(src),
                       static_cast<float*>(dst), (long long)(n / 4));
  else
    hipLaunchKernelGGL((convert_act_kernel<float, bf16s>), dim3(blocks), dim3(256), 0, (hipStream_t)stream, static_cast<const float*>(src),
                       static_cast<bf16s*>(dst), (long long)(n / 4));
  VMM_LAUNCH_CHECK();
  return 0;
}

// x = the block's pre-norm convolution output (rows x 64), coef = vmm_groupnorm_coef's [B][64][2], res = the block's residual rows or NULL,
// w = (Cout, 64) torch weight of the 1x1 convolution, out = (B, Cout, T, HW).  Returns 1 (nothing launched) unless C == 64 and Cout <= 4.
extern "C" int vmm_affine_silu_pointwise_to_ncthw(const float* x, int32_t ldx, const float* coef, const float* res, int32_t ldres, int32_t C,
                                                  const float* w, const float* bias, int32_t B, int32_t Cout, int32_t T, int32_t HW, float* out,
                                                  vmm_stream_t stream) {
  if (C != 64 || Cout < 1 || Cout > 4 || (ldx & 3) || (res && (ldres & 3))) return 1;
  const long long nrows = (long long)B * T * HW;
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(affine_silu_pointwise_kernel<float>, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, (hipStream_t)stream, x, ldx, coef, res, ldres,
                     T * HW, w, bias, Cout, T, HW, out, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}
// ... over bf16-STORED maps (x, res = bf16 bits); the (B, Cout, T, HW) output stays fp32 (the API edge)
extern "C" int vmm_affine_silu_pointwise_to_ncthw_a16(const void* x, int32_t ldx, const float* coef, const void* res, int32_t ldres, int32_t C,
                                                      const float* w, const float* bias, int32_t B, int32_t Cout, int32_t T, int32_t HW, float* out,
                                                      vmm_stream_t stream) {
  if (C != 64 || Cout < 1 || Cout > 4 || (ldx & 3) || (res && (ldres & 3))) return 1;
  const long long nrows = (long long)B * T * HW;
  if (nrows <= 0) return 0;
  hipLaunchKernelGGL(affine_silu_pointwise_kernel<bf16s>, dim3((unsigned)cdiv(nrows, 256)), dim3(256), 0, (hipStream_t)stream,
                     static_cast<const bf16s*>(x), ldx, coef, static_cast<const bf16s*>(res), ldres, T * HW, w, bias, Cout, T, HW, out, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_channel_layernorm(const float* x, int32_t ldx, const float* gamma, float* y, int32_t ldy, int64_t rows,
                                     int32_t C, float eps, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || (ldx & 3) || (ldy & 3) || C > 2048) return -1;
  const int c4 = C >> 2;
  int gs = 1;
  while (gs < 64 && gs < c4) gs <<= 1;  // smallest power of two >= C/4, capped at 64
  const long long threads = rows * gs;
  const int blocks = cdiv(threads, 256);
#define LN_CASE(G)                                                                                                        \
  case G:                                                                                                                 \
    hipLaunchKernelGGL(chan_ln_kernel<G>, dim3(blocks), dim3(256), 0, s, x, ldx, gamma, y, ldy, (long long)rows, C, eps); \
    break;
  switch (gs) {
    LN_CASE(1) LN_CASE(2) LN_CASE(4) LN_CASE(8) LN_CASE(16) LN_CASE(32) LN_CASE(64)
  }
#undef LN_CASE
  VMM_LAUNCH_CHECK();
  return 0;
}
