// Diffusion-step arithmetic on (B, C, T, H, W) tensors viewed as B rows of `per_sample` floats
// (SURVEY.md K17, K18, K20): HBM-bound sweeps with the per-sample schedule coefficients gathered by the
// INTEGER timestep t[b], plus an exact radix select for the dynamic-threshold quantile.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

constexpr int EW_BLOCK = 256;
__host__ inline dim3 ew_grid(long long per_sample, int B) {
  long long blocks = (per_sample + EW_BLOCK * 4 - 1) / (EW_BLOCK * 4);
  if (blocks > 2048) blocks = 2048;
  if (blocks < 1) blocks = 1;
  return dim3((unsigned)blocks, B);
}

// grid (x, B): every kernel below grid-strides over one sample per blockIdx.y (coalesced 4-byte lanes)
#define EW_LOOP(body)                                                                                         \
  const long long base = (long long)blockIdx.y * per_sample;                                                  \
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample;                        \
       i += (long long)gridDim.x * blockDim.x) {                                                              \
    const long long o = base + i;                                                                             \
    body(o, 1)                                                                                                \
  }

__global__ void q_sample_kernel(const float* __restrict__ x0, const float* __restrict__ noise, const int64_t* __restrict__ t,
                                const float* __restrict__ ca, const float* __restrict__ cs, int normalize, float* __restrict__ out,
                                long long per_sample) {
  const float a = ca[t[blockIdx.y]], s = cs[t[blockIdx.y]];
#define BODY(o, W)                                              \
  for (int j = 0; j < W; ++j) {                                 \
    float v = x0[o + j];                                        \
    if (normalize) v = v * 2.0f - 1.0f;                         \
    out[o + j] = a * v + s * noise[o + j];                      \
  }
  EW_LOOP(BODY)
#undef BODY
}

__global__ void predict_x0_kernel(const float* __restrict__ x, const float* __restrict__ ec, const float* __restrict__ en, float w,
                                  const int64_t* __restrict__ t, const float* __restrict__ c_recip, const float* __restrict__ c_recipm1,
                                  float* __restrict__ x0, float* __restrict__ ax0, long long per_sample) {
  const float cr = c_recip[t[blockIdx.y]], cm = c_recipm1[t[blockIdx.y]];
#define BODY(o, W)                                              \
  for (int j = 0; j < W; ++j) {                                 \
    float e = ec[o + j];                                        \
    if (en) { const float nn = en[o + j]; e = nn + (e - nn) * w; } \
    const float v = cr * x[o + j] - cm * e;                     \
    x0[o + j] = v;                                              \
    if (ax0) ax0[o + j] = fabsf(v);                             \
  }
  EW_LOOP(BODY)
#undef BODY
}

__global__ void posterior_step_kernel(const float* __restrict__ x0, const float* __restrict__ x, const float* __restrict__ noise,
                                      const float* __restrict__ sthr, const int64_t* __restrict__ t, const float* __restrict__ c1,
                                      const float* __restrict__ c2, const float* __restrict__ logvar, int clip_mode,
                                      float* __restrict__ out, long long per_sample) {
  const int64_t tb = t[blockIdx.y];
  const float k1 = c1[tb], k2 = c2[tb];
  const float sig = (tb == 0) ? 0.f : expf(0.5f * logvar[tb]);
  const float s = (clip_mode == 2) ? sthr[blockIdx.y] : 1.0f;
#define BODY(o, W)                                              \
  for (int j = 0; j < W; ++j) {                                 \
    float c = x0[o + j];                                        \
    if (clip_mode) c = fminf(fmaxf(c, -s), s) / s;              \
    const float mean = k1 * c + k2 * x[o + j];                  \
    out[o + j] = noise ? mean + sig * noise[o + j] : mean;      \
  }
  EW_LOOP(BODY)
#undef BODY
}

// ---- step noise generated in the kernel (torch.randn_like of vddp.py:960): Philox4x32-10, key = the sample() call's 64-bit seed,
// counter = (group of four elements: low / high word, timestep, sample).  One block call gives four uniform words = four standard normals
// (two Box-Muller pairs).  The noise tensor (4.9 MB per step at 4 x 3 x 11 x 96 x 96) is never written or read.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0, unsigned k1, unsigned (&r)[4]) {
#pragma unroll
  for (int i = 0; i < 10; ++i) {
    const unsigned hi0 = __umulhi(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
    const unsigned hi1 = __umulhi(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
    c0 = hi1 ^ c1 ^ k0; c1 = lo1; c2 = hi0 ^ c3 ^ k1; c3 = lo0;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  r[0] = c0; r[1] = c1; r[2] = c2; r[3] = c3;
}
__device__ __forceinline__ void box_muller(unsigned a, unsigned b, float& z0, float& z1) {
  const float u1 = ((float)(a >> 8) + 0.5f) * (1.0f / 16777216.0f);  // (0, 1): 24 bits, never 0
  const float u2 = ((float)(b >> 8) + 0.5f) * (1.0f / 16777216.0f);
  const float r = sqrtf(-2.0f * logf(u1));
  float sn, cs;
  sincosf(6.283185307179586f * u2, &sn, &cs);
  z0 = r * cs;
  z1 = r * sn;
}

// posterior step with in-kernel noise; thread = four consecutive elements (per_sample % 4 == 0, 16-byte accesses).  Block (0, b) also leaves
// t_next[b] = t[b] - 1 for the next replay of a captured step (t itself stays untouched while other blocks read it).
__global__ __launch_bounds__(256) void posterior_step_rng_kernel(const float* __restrict__ x0, const float* __restrict__ x, const int64_t* __restrict__ rng,
                                                                 const float* __restrict__ sthr, const int64_t* __restrict__ t,
                                                                 const float* __restrict__ c1, const float* __restrict__ c2,
                                                                 const float* __restrict__ logvar, int clip_mode, float* __restrict__ out,
                                                                 long long per_sample, int64_t* __restrict__ t_next) {
  const int64_t tb = t[blockIdx.y];
  const float k1 = c1[tb], k2 = c2[tb];
  const float sig = (tb == 0) ? 0.f : expf(0.5f * logvar[tb]);
  const float s = (clip_mode == 2) ? sthr[blockIdx.y] : 1.0f;
  const unsigned long long seed = (unsigned long long)rng[0];
  const long long base = (long long)blockIdx.y * per_sample, n4 = per_sample >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 a = *reinterpret_cast<const f32x4*>(x0 + base + 4 * i), b = *reinterpret_cast<const f32x4*>(x + base + 4 * i);
    unsigned r[4];
    philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)tb, blockIdx.y, (unsigned)seed, (unsigned)(seed >> 32), r);
    float z[4];
    box_muller(r[0], r[1], z[0], z[1]);
    box_muller(r[2], r[3], z[2], z[3]);
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      float c = a[j];
      if (clip_mode) c = fminf(fmaxf(c, -s), s) / s;
      o[j] = (k1 * c + k2 * b[j]) + sig * z[j];
    }
    *reinterpret_cast<f32x4*>(out + base + 4 * i) = o;
  }
  if (t_next && blockIdx.x == 0 && threadIdx.x == 0) t_next[blockIdx.y] = tb - 1;
}

// inputs of a captured guided step, one launch (they were four torch elementwise kernels and a fill): t[b] = t_src[b]; time_in[b] (and
// time_in[B + b] when two_halves) = t_src[b]; x_in[first half] (and the second half unless the plan is mirrored) = img
__global__ __launch_bounds__(256) void step_inputs_kernel(const float* __restrict__ img, const int64_t* __restrict__ t_src, float* __restrict__ x_in,
                                                          int copy_second, int64_t* __restrict__ t, int64_t* __restrict__ time_in, int B, long long n4) {
  const long long i0 = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i0 < B) {
    const int64_t v = t_src[i0];
    t[i0] = v;
    time_in[i0] = v;
    if (!(copy_second & 2)) time_in[B + i0] = v;  // (bit 1: a plan of B rows -- the unguided step --, no second half anywhere)
  }
  for (long long i = i0; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(img)[i];
    reinterpret_cast<f32x4*>(x_in)[i] = v;
    if (copy_second & 1) reinterpret_cast<f32x4*>(x_in)[n4 + i] = v;
  }
}

// One DDIM step (vddp.py:986-1018) in one launch, for the captured sampler: eps = null + (cond - null) w, x0 = sqrt_recip[t] x - sqrt_recipm1[t] eps,
// x <- sqrt(a_next) x0 + c eps + sigma noise with the step's three coefficients from a host-made table indexed by t (coef[t] = {sqrt(a_next), c,
// sigma, last}: evaluated on the host exactly as the eager path does, fp32 0-d tensors; last != 0: x <- x0), the noise (sigma != 0 only) from the
// in-kernel Philox generator of the ancestral step; block (0, b) leaves the list's next timestep in t_next[b].
__global__ __launch_bounds__(256) void ddim_step_rng_kernel(const float* __restrict__ x, const float* __restrict__ ec, const float* __restrict__ en, float w,
                                                            const int64_t* __restrict__ t, const float* __restrict__ c_recip,
                                                            const float* __restrict__ c_recipm1, const float* __restrict__ coef,
                                                            const int64_t* __restrict__ next_of, const int64_t* __restrict__ rng, float* __restrict__ out,
                                                            long long per_sample, int64_t* __restrict__ t_next) {
  const int64_t tb = t[blockIdx.y];
  const float cr = c_recip[tb], cm = c_recipm1[tb];
  const float sa = coef[4 * tb], cc = coef[4 * tb + 1], sig = coef[4 * tb + 2];
  const bool last = coef[4 * tb + 3] != 0.f;
  const unsigned long long seed = (unsigned long long)rng[0];
  const long long base = (long long)blockIdx.y * per_sample, n4 = per_sample >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 xv = *reinterpret_cast<const f32x4*>(x + base + 4 * i);
    f32x4 e = *reinterpret_cast<const f32x4*>(ec + base + 4 * i);
    if (en) {
      const f32x4 nn = *reinterpret_cast<const f32x4*>(en + base + 4 * i);
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = nn[j] + (e[j] - nn[j]) * w;
    }
    float z[4] = {0.f, 0.f, 0.f, 0.f};
    if (sig != 0.f && !last) {
      unsigned r[4];
      philox4x32_10((unsigned)i, (unsigned)(i >> 32), (unsigned)tb, blockIdx.y, (unsigned)seed, (unsigned)(seed >> 32), r);
      box_muller(r[0], r[1], z[0], z[1]);
      box_muller(r[2], r[3], z[2], z[3]);
    }
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = cr * xv[j] - cm * e[j];
      float v = sa * x0;
      v += cc * e[j];
      if (sig != 0.f) v += sig * z[j];
      o[j] = last ? x0 : v;
    }
    *reinterpret_cast<f32x4*>(out + base + 4 * i) = o;
  }
  if (t_next && blockIdx.x == 0 && threadIdx.x == 0) t_next[blockIdx.y] = next_of[tb];
}

__global__ void lincomb_kernel(const float* __restrict__ x, const float* __restrict__ y, const float* __restrict__ z, float a, float b,
                               float c, float d, float* __restrict__ out, long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    float v = a * x[i];
    if (y) v += b * y[i];
    if (z) v += c * z[i];
    out[i] = v + d;
  }
}

// one read, two writes (16 bytes per lane): the shared prefix of the two guidance branches handed to both halves of the batch
__global__ __launch_bounds__(256) void copy2_kernel(const float4* __restrict__ x, float4* __restrict__ a, float4* __restrict__ b, long long n4) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const float4 v = x[i];
    a[i] = v;
    b[i] = v;
  }
}

// classifier-free guidance, same association as the reference: null + (cond - null) * w (vddp.py:728)
__global__ void cfg_combine_kernel(const float* __restrict__ ec, const float* __restrict__ en, float w, float* __restrict__ out,
                                   long long n) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float nn = en[i];
    out[i] = nn + (ec[i] - nn) * w;
  }
}

__global__ __launch_bounds__(256) void loss_reduce_kernel(const float* __restrict__ a, const float* __restrict__ b, long long n,
                                                          int squared, double* __restrict__ acc) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = a[i] - b[i];
    s += squared ? d * d : fabsf(d);
  }
  double ds = wave_sum_d((double)s);
  __shared__ double sh[4];
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = ds;
  __syncthreads();
  if (threadIdx.x == 0) atomicAdd(acc, sh[0] + sh[1] + sh[2] + sh[3]);
}
__global__ void loss_finalize_kernel(const double* __restrict__ acc, double inv_n, float* __restrict__ out) { out[0] = (float)(acc[0] * inv_n); }

// ---------------------------------------------------------------- exact order statistics by 12/12/8-bit radix select
// scratch per sample: hist[4096] | state { prefix, k_rem, count_le, min_gt }
constexpr int QH = 4096;
constexpr int QSTRIDE = QH + 16;

__global__ __launch_bounds__(256) void q_hist_kernel(const float* __restrict__ v, long long n, int pass, uint32_t* __restrict__ scratch) {
  __shared__ uint32_t h[QH];
  const int b = blockIdx.y;
  uint32_t* S = scratch + (long long)b * QSTRIDE;
  const uint32_t prefix = S[QH + 0];
  for (int i = threadIdx.x; i < QH; i += blockDim.x) h[i] = 0;
  __syncthreads();
  const uint32_t* u = reinterpret_cast<const uint32_t*>(v) + (long long)b * n;
  // pass 0: bits 31..20 ; pass 1: bits 19..8 within matching top-12 ; pass 2: bits 7..0 within matching top-24
  const int shift = (pass == 0) ? 20 : (pass == 1 ? 8 : 0);
  const uint32_t bmask = (pass == 2) ? 0xFFu : 0xFFFu;
  const uint32_t pmask = (pass == 0) ? 0u : (pass == 1 ? 0xFFF00000u : 0xFFFFFF00u);
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const uint32_t x = u[i];
    if ((x & pmask) == (prefix & pmask)) atomicAdd(&h[(x >> shift) & bmask], 1u);
  }
  __syncthreads();
  for (int i = threadIdx.x; i < QH; i += blockDim.x)
    if (h[i]) atomicAdd(&S[i], h[i]);
}

// one block per sample: locate the bin that holds rank k_rem, fold it into the prefix, clear the histogram
__global__ __launch_bounds__(256) void q_select_kernel(int pass, uint32_t* __restrict__ scratch) {
  uint32_t* S = scratch + (long long)blockIdx.x * QSTRIDE;
  __shared__ uint32_t part[256];
  __shared__ uint32_t found_bin, found_before;
  const int tid = threadIdx.x;
  const int per = QH / 256;
  uint32_t loc[QH / 256];
  uint32_t s = 0;
  for (int j = 0; j < per; ++j) { loc[j] = S[tid * per + j]; s += loc[j]; }
  part[tid] = s;
  __syncthreads();
  if (tid == 0) {
    uint32_t run = 0;
    for (int i = 0; i < 256; ++i) { const uint32_t c = part[i]; part[i] = run; run += c; }
  }
  __syncthreads();
  const uint32_t k = S[QH + 1];
  uint32_t run = part[tid];
  for (int j = 0; j < per; ++j) {
    if (k >= run && k < run + loc[j]) { found_bin = tid * per + j; found_before = run; }
    run += loc[j];
  }
  __syncthreads();
  for (int j = 0; j < per; ++j) S[tid * per + j] = 0;
  if (tid == 0) {
    const int shift = (pass == 0) ? 20 : (pass == 1 ? 8 : 0);
    S[QH + 0] |= found_bin << shift;
    S[QH + 1] = k - found_before;
    if (pass == 2) { S[QH + 2] = 0; S[QH + 3] = 0xFFFFFFFFu; }
  }
}

// count of elements <= v_k and the smallest element > v_k
__global__ __launch_bounds__(256) void q_next_kernel(const float* __restrict__ v, long long n, uint32_t* __restrict__ scratch) {
  const int b = blockIdx.y;
  uint32_t* S = scratch + (long long)b * QSTRIDE;
  const uint32_t vk = S[QH + 0];
  const uint32_t* u = reinterpret_cast<const uint32_t*>(v) + (long long)b * n;
  uint32_t cnt = 0, mn = 0xFFFFFFFFu;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const uint32_t x = u[i];
    if (x <= vk) ++cnt; else mn = min(mn, x);
  }
  for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); mn = min(mn, (uint32_t)__shfl_xor(mn, o, 64)); }
  if ((threadIdx.x & 63) == 0) { atomicAdd(&S[QH + 2], cnt); atomicMin(&S[QH + 3], mn); }
}

__global__ void q_finish_kernel(const uint32_t* __restrict__ scratch, long long k_lo, float frac, float floor_min, int B,
                                float* __restrict__ s_out) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const uint32_t* S = scratch + (long long)b * QSTRIDE;
  const float lo = __uint_as_float(S[QH + 0]);
  // sorted[k_lo+1] equals sorted[k_lo] when more than k_lo+1 elements are <= it
  const float hi = ((long long)S[QH + 2] > k_lo + 1 || S[QH + 3] == 0xFFFFFFFFu) ? lo : __uint_as_float(S[QH + 3]);
  // torch.lerp (ATen Lerp.h): weight < 0.5 ? a + w*(b-a) : b - (b-a)*(1-w)
  const float d = hi - lo;
  float r = (frac < 0.5f) ? lo + frac * d : hi - d * (1.0f - frac);
  s_out[b] = fmaxf(r, floor_min);
}

__global__ void q_init_kernel(uint32_t* __restrict__ scratch, uint32_t k) {
  uint32_t* S = scratch + (long long)blockIdx.x * QSTRIDE;
  for (int i = threadIdx.x; i < QSTRIDE; i += blockDim.x) S[i] = (i == QH + 1) ? k : 0u;
}

}  // namespace

extern "C" int vmm_q_sample(const float* x0, const float* noise, const int64_t* t, const float* sqrt_acp, const float* sqrt_1macp,
                            int32_t normalize, float* out, int32_t B, int64_t per_sample, vmm_stream_t stream) {
  hipLaunchKernelGGL(q_sample_kernel, ew_grid(per_sample, B), dim3(EW_BLOCK), 0, (hipStream_t)stream, x0, noise, t, sqrt_acp,
                     sqrt_1macp, normalize, out, (long long)per_sample);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_predict_x0(const float* x, const float* eps_cond, const float* eps_null, float w, const int64_t* t,
                              const float* sqrt_recip_acp, const float* sqrt_recipm1_acp, float* x0, float* absx0, int32_t B,
                              int64_t per_sample, vmm_stream_t stream) {
  hipLaunchKernelGGL(predict_x0_kernel, ew_grid(per_sample, B), dim3(EW_BLOCK), 0, (hipStream_t)stream, x, eps_cond, eps_null, w, t,
                     sqrt_recip_acp, sqrt_recipm1_acp, x0, absx0, (long long)per_sample);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_posterior_step(const float* x0, const float* x, const float* noise, const float* s, const int64_t* t,
                                  const float* coef1, const float* coef2, const float* logvar, int32_t clip_mode, float* out,
                                  int32_t B, int64_t per_sample, vmm_stream_t stream) {
  if (clip_mode == 2 && !s) return -1;
  hipLaunchKernelGGL(posterior_step_kernel, ew_grid(per_sample, B), dim3(EW_BLOCK), 0, (hipStream_t)stream, x0, x, noise, s, t, coef1,
                     coef2, logvar, clip_mode, out, (long long)per_sample);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_posterior_step_rng(const float* x0, const float* x, const int64_t* rng_seed, const float* s, const int64_t* t, const float* coef1,
                                      const float* coef2, const float* logvar, int32_t clip_mode, float* out, int32_t B, int64_t per_sample,
                                      int64_t* t_next, vmm_stream_t stream) {
  if ((clip_mode == 2 && !s) || !rng_seed || (per_sample & 3) || (((uintptr_t)x0 | (uintptr_t)x | (uintptr_t)out) & 15)) return -1;
  hipLaunchKernelGGL(posterior_step_rng_kernel, ew_grid(per_sample / 4, B), dim3(EW_BLOCK), 0, (hipStream_t)stream, x0, x, rng_seed, s, t, coef1, coef2,
                     logvar, clip_mode, out, (long long)per_sample, t_next);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_step_inputs(const float* img, const int64_t* t_src, float* x_in, int32_t copy_second_half, int64_t* t, int64_t* time_in, int32_t B,
                               int64_t n, vmm_stream_t stream) {
  if (!img || !t_src || !x_in || !t || !time_in || B < 1 || (n & 3) || (((uintptr_t)img | (uintptr_t)x_in) & 15)) return -1;
  const int blocks = (int)min((long long)cdiv(n / 4, 256), 4096LL);
  hipLaunchKernelGGL(step_inputs_kernel, dim3(max(blocks, cdiv(B, 256))), dim3(256), 0, (hipStream_t)stream, img, t_src, x_in, copy_second_half, t, time_in, B,
                     (long long)(n / 4));
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_ddim_step_rng(const float* x, const float* eps_cond, const float* eps_null, float w, const int64_t* t, const float* c_recip,
                                 const float* c_recipm1, const float* coef, const int64_t* next_of, const int64_t* rng_seed, float* out, int32_t B,
                                 int64_t per_sample, int64_t* t_next, vmm_stream_t stream) {
  if (B <= 0 || per_sample <= 0) return 0;
  if ((per_sample & 3) || (((uintptr_t)x | (uintptr_t)eps_cond | (uintptr_t)eps_null | (uintptr_t)out) & 15)) return -1;
  const int blocks = (int)max(1LL, min((long long)2048, (per_sample / 4 + 255) / 256));
  hipLaunchKernelGGL(ddim_step_rng_kernel, dim3(blocks, B), dim3(256), 0, (hipStream_t)stream, x, eps_cond, eps_null, w, t, c_recip, c_recipm1, coef, next_of,
                     rng_seed, out, (long long)per_sample, t_next);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_lincomb(const float* x, const float* y, const float* z, float a, float b, float c, float d, float* out, int64_t n,
                           vmm_stream_t stream) {
  const int blocks = (int)min((long long)cdiv(n, 256), 4096LL);
  hipLaunchKernelGGL(lincomb_kernel, dim3(max(blocks, 1)), dim3(256), 0, (hipStream_t)stream, x, y, z, a, b, c, d, out, (long long)n);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_copy2(const float* x, float* out_a, float* out_b, int64_t n, vmm_stream_t stream) {
  if (!x || !out_a || !out_b || n < 0 || (n & 3) || (((uintptr_t)x | (uintptr_t)out_a | (uintptr_t)out_b) & 15)) return -1;
  if (n == 0) return 0;
  const int blocks = (int)min((long long)cdiv(n / 4, 256), 8192LL);
  hipLaunchKernelGGL(copy2_kernel, dim3(max(blocks, 1)), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const float4*>(x),
                     reinterpret_cast<float4*>(out_a), reinterpret_cast<float4*>(out_b), (long long)(n / 4));
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_loss_reduce(const float* a, const float* b, int64_t n, int32_t squared, double* acc, float* out_mean,
                               vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if (int rc = vmm_zero_async(acc, sizeof(double), s)) return rc;
  const int blocks = (int)min((long long)cdiv(n, 256 * 8), 1024LL);
  hipLaunchKernelGGL(loss_reduce_kernel, dim3(max(blocks, 1)), dim3(256), 0, s, a, b, (long long)n, squared, acc);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(1), 0, s, acc, 1.0 / (double)n, out_mean);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_quantile_rows(const float* absx, int32_t B, int64_t n, int64_t k_lo, float frac, float floor_min, float* s_out,
                                 uint32_t* scratch, vmm_stream_t stream) {
  if (k_lo < 0 || k_lo >= n || n >= (1LL << 32)) return -1;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(q_init_kernel, dim3(B), dim3(256), 0, s, scratch, (uint32_t)k_lo);
  VMM_LAUNCH_CHECK();
  const int blocks = (int)max(1LL, min((long long)cdiv(n, 256 * 16), 256LL));
  for (int pass = 0; pass < 3; ++pass) {
    hipLaunchKernelGGL(q_hist_kernel, dim3(blocks, B), dim3(256), 0, s, absx, (long long)n, pass, scratch);
    VMM_LAUNCH_CHECK();
    hipLaunchKernelGGL(q_select_kernel, dim3(B), dim3(256), 0, s, pass, scratch);
    VMM_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(q_next_kernel, dim3(blocks, B), dim3(256), 0, s, absx, (long long)n, scratch);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(q_finish_kernel, dim3(cdiv(B, 64)), dim3(64), 0, s, scratch, (long long)k_lo, frac, floor_min, B, s_out);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_cfg_combine(const float* eps_cond, const float* eps_null, float w, float* out, int64_t n, vmm_stream_t stream) {
  const int blocks = (int)min((long long)cdiv(n, 256), 4096LL);
  hipLaunchKernelGGL(cfg_combine_kernel, dim3(max(blocks, 1)), dim3(256), 0, (hipStream_t)stream, eps_cond, eps_null, w, out, (long long)n);
  VMM_LAUNCH_CHECK();
  return 0;
}
