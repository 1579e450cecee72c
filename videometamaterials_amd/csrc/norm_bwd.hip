// Backward of GroupNorm(+FiLM)+SiLU and of the channel LayerNorm (training path), channels-last rows, gfx950.
//
// GroupNorm block  z = silu(y), y = a[b,c]*h + b'[b,c]  (a = rstd*gamma*(1+scale), see norm.hip):
//   dy = dz * silu'(y);  per (sample, channel):  P1 = sum_rows dy,  P2 = sum_rows dy * hhat,  hhat = (h - mean_g) * rstd_g
//   dh = a*dy - rstd_g * (m1_g + hhat * m2_g),   m1_g = sum_{c in g} gamma'_c P1_c / n,  m2_g = sum_{c in g} gamma'_c P2_c / n
//   dgamma_c += sum_b (1+scale_bc) P2_bc ; dbeta_c += sum_b (1+scale_bc) P1_bc ; dscale_bc = gamma_c P2_bc + beta_c P1_bc ; dshift_bc = P1_bc
// Two HBM sweeps over (dz, h): reduce, then apply.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

__device__ __forceinline__ float silu_grad(float y) {
  const float s = 1.0f / (1.0f + expf(-y));
  return s * (1.0f + y * (1.0f - s));
}

// grid (blocks_per_sample, B); P[b][c] = (P1, P2) accumulated with fp32 atomics (zeroed by the launcher)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ h, int ldh,
                                                            const float* __restrict__ coef, const float* __restrict__ stats, int rows_per_sample,
                                                            int C, int G, int rows_per_block, float* __restrict__ P) {
  const int b = blockIdx.y;
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int col4 = tid % c4n, rl = tid / c4n, rslots = 256 / c4n;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows_per_sample);
  float p1[4] = {0, 0, 0, 0}, p2[4] = {0, 0, 0, 0};
  if (rl < rslots) {
    const int c = col4 * 4;
    float a[4], bb[4], mu[4], rs[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      a[j] = coef[((long long)b * C + c + j) * 2];
      bb[j] = coef[((long long)b * C + c + j) * 2 + 1];
      const int g = (c + j) / (C / G);
      mu[j] = stats[(b * G + g) * 2];
      rs[j] = stats[(b * G + g) * 2 + 1];
    }
    const long long base = (long long)b * rows_per_sample;
    for (int r = r0 + rl; r < r1; r += rslots) {
      const f32x4 hv = *reinterpret_cast<const f32x4*>(h + (base + r) * ldh + c);
      const f32x4 gv = *reinterpret_cast<const f32x4*>(dz + (base + r) * lddz + c);
      const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float dy = gg[j] * silu_grad(a[j] * hh[j] + bb[j]);
        p1[j] += dy;
        p2[j] += dy * (hh[j] - mu[j]) * rs[j];
      }
    }
  }
  __shared__ float s1[1024], s2[1024];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[tid * 4 + j] = p1[j]; s2[tid * 4 + j] = p2[j]; }
  __syncthreads();
  for (int ch = tid; ch < C; ch += 256) {
    const int c4 = ch >> 2, j = ch & 3;
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < rslots; ++k) { t1 += s1[(k * c4n + c4) * 4 + j]; t2 += s2[(k * c4n + c4) * 4 + j]; }
    atomicAdd(&P[((long long)b * C + ch) * 2], t1);
    atomicAdd(&P[((long long)b * C + ch) * 2 + 1], t2);
  }
}

// one thread per (b, c): group means m1, m2 and the parameter / FiLM gradients
__global__ void gn_bwd_coef_kernel(const float* __restrict__ P, const float* __restrict__ gamma, const float* __restrict__ beta,
                                   const float* __restrict__ film, int ldfilm, float inv_n, int B, int C, int G, float* __restrict__ m12,
                                   float* __restrict__ dgamma, float* __restrict__ dbeta, float* __restrict__ dfilm) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  const int Cg = C / G, g = c / Cg;
  const float p1 = P[i * 2], p2 = P[i * 2 + 1];
  const float sc1 = film ? film[(long long)b * ldfilm + c] + 1.0f : 1.0f;
  atomicAdd(&dgamma[c], sc1 * p2);
  atomicAdd(&dbeta[c], sc1 * p1);
  if (dfilm) {
    dfilm[(long long)b * ldfilm + c] = gamma[c] * p2 + beta[c] * p1;
    dfilm[(long long)b * ldfilm + C + c] = p1;
  }
  if (c == g * Cg) {
    float m1 = 0.f, m2 = 0.f;
    for (int cc = c; cc < c + Cg; ++cc) {
      const float gp = gamma[cc] * (film ? film[(long long)b * ldfilm + cc] + 1.0f : 1.0f);
      m1 += gp * P[((long long)b * C + cc) * 2];
      m2 += gp * P[((long long)b * C + cc) * 2 + 1];
    }
    m12[(b * G + g) * 2] = m1 * inv_n;
    m12[(b * G + g) * 2 + 1] = m2 * inv_n;
  }
}

// dh (=|+=) a*dy - rstd*(m1 + hhat*m2)
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ coef, const float* __restrict__ stats,
                                                           const float* __restrict__ m12, float* __restrict__ dh, int lddh, long long rows,
                                                           int rows_per_sample, int C, int G, int accumulate) {
  const int c4n = C >> 2;
  const long long total = rows * c4n;
  const int Cg = C / G;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / c4n;
    const int c = (int)(i - r * c4n) * 4;
    const int b = (int)(r / rows_per_sample);
    const f32x4 hv = *reinterpret_cast<const f32x4*>(h + r * ldh + c);
    const f32x4 gv = *reinterpret_cast<const f32x4*>(dz + r * lddz + c);
    const float hh[4] = {hv.x, hv.y, hv.z, hv.w}, gg[4] = {gv.x, gv.y, gv.z, gv.w};
    float o[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = coef[((long long)b * C + c + j) * 2], bb = coef[((long long)b * C + c + j) * 2 + 1];
      const int g = (c + j) / Cg;
      const float mu = stats[(b * G + g) * 2], rs = stats[(b * G + g) * 2 + 1];
      const float dy = gg[j] * silu_grad(a * hh[j] + bb);
      o[j] = a * dy - rs * (m12[(b * G + g) * 2] + (hh[j] - mu) * rs * m12[(b * G + g) * 2 + 1]);
    }
    float* op = dh + r * lddh + c;
    if (accumulate) {
      const f32x4 old = *reinterpret_cast<const f32x4*>(op);
      o[0] += old.x; o[1] += old.y; o[2] += old.z; o[3] += old.w;
    }
    *reinterpret_cast<f32x4*>(op) = (f32x4){o[0], o[1], o[2], o[3]};
  }
}

// channel LayerNorm backward; GS lanes per row; every thread keeps the dgamma of its own channels in registers across its rows,
// then one LDS reduction and one atomic per channel per workgroup
template <int GS, int MAXV>
__global__ __launch_bounds__(256) void chan_ln_bwd_kernel(const float* __restrict__ x, int ldx, const float* __restrict__ gamma,
                                                          const float* __restrict__ dy, int lddy, float* __restrict__ dx, int lddx,
                                                          float* __restrict__ dgamma, long long rows, int C, float eps, int accumulate,
                                                          float* __restrict__ dgamma_part) {
  extern __shared__ float dg_sh[];  // [C]
  const int tid = threadIdx.x;
  for (int c = tid; c < C; c += 256) dg_sh[c] = 0.f;
  __syncthreads();
  const int sub = tid % GS;  // lane `sub` of a row owns channels (j * GS + sub) * 4 .. + 3, j < MAXV
  const long long rows_per_iter = (long long)gridDim.x * (256 / GS);
  f32x4 dga[MAXV];
#pragma unroll
  for (int j = 0; j < MAXV; ++j) dga[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  for (long long row0 = (long long)blockIdx.x * (256 / GS); row0 < rows; row0 += rows_per_iter) {
    const long long row = row0 + tid / GS;
    const bool valid = row < rows;
    const float* xr = x + (valid ? row : 0) * ldx;
    const float* gr = dy + (valid ? row : 0) * lddy;
    f32x4 v[MAXV], g[MAXV];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c = (j * GS + sub) * 4;
      v[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      g[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (valid && c < C) {
        v[j] = *reinterpret_cast<const f32x4*>(xr + c);
        const f32x4 d = *reinterpret_cast<const f32x4*>(gr + c);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + c);
        g[j] = (f32x4){d.x * gm.x, d.y * gm.y, d.z * gm.z, d.w * gm.w};  // gamma * dy
      }
      s += v[j].x + v[j].y + v[j].z + v[j].w;
    }
    const float mean = group_sum(s, GS) / (float)C;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
        v[j].x -= mean; v[j].y -= mean; v[j].z -= mean; v[j].w -= mean;
        q += v[j].x * v[j].x + v[j].y * v[j].y + v[j].z * v[j].z + v[j].w * v[j].w;
      }
    }
    const float rstd = 1.0f / sqrtf(group_sum(q, GS) / (float)C + eps);
    float a1 = 0.f, a2 = 0.f;  // sum g, sum g*xhat
#pragma unroll
    for (int j = 0; j < MAXV; ++j) {
      const int c = (j * GS + sub) * 4;
      if (c < C) {
        v[j].x *= rstd; v[j].y *= rstd; v[j].z *= rstd; v[j].w *= rstd;  // xhat
        a1 += g[j].x + g[j].y + g[j].z + g[j].w;
        a2 += g[j].x * v[j].x + g[j].y * v[j].y + g[j].z * v[j].z + g[j].w * v[j].w;
      }
    }
    a1 = group_sum(a1, GS) / (float)C;
    a2 = group_sum(a2, GS) / (float)C;
    if (valid) {
#pragma unroll
      for (int j = 0; j < MAXV; ++j) {
        const int c = (j * GS + sub) * 4;
        if (c < C) {
          f32x4 o = {rstd * (g[j].x - a1 - v[j].x * a2), rstd * (g[j].y - a1 - v[j].y * a2), rstd * (g[j].z - a1 - v[j].z * a2),
                     rstd * (g[j].w - a1 - v[j].w * a2)};
          float* op = dx + row * lddx + c;
          if (accumulate) {
            const f32x4 old = *reinterpret_cast<const f32x4*>(op);
            o.x += old.x; o.y += old.y; o.z += old.z; o.w += old.w;
          }
          *reinterpret_cast<f32x4*>(op) = o;
          // dgamma_c += dy_c * xhat_c
          const f32x4 d = *reinterpret_cast<const f32x4*>(gr + c);
          dga[j].x = fmaf(d.x, v[j].x, dga[j].x); dga[j].y = fmaf(d.y, v[j].y, dga[j].y);
          dga[j].z = fmaf(d.z, v[j].z, dga[j].z); dga[j].w = fmaf(d.w, v[j].w, dga[j].w);
        }
      }
    }
  }
#pragma unroll
  for (int j = 0; j < MAXV; ++j) {
    const int c = (j * GS + sub) * 4;
    if (c < C) {
      atomicAdd(&dg_sh[c], dga[j].x); atomicAdd(&dg_sh[c + 1], dga[j].y);
      atomicAdd(&dg_sh[c + 2], dga[j].z); atomicAdd(&dg_sh[c + 3], dga[j].w);
    }
  }
  __syncthreads();
  if (dgamma_part) {  // one partial row per workgroup, summed by vmm_sum_partials
    for (int c = tid; c < C; c += 256) dgamma_part[(long long)blockIdx.x * C + c] = dg_sh[c];
  } else {
    for (int c = tid; c < C; c += 256) atomicAdd(&dgamma[c], dg_sh[c]);
  }
}

}  // namespace

extern "C" int vmm_groupnorm_bwd(const float* dz, int32_t lddz, const float* h, int32_t ldh, const float* coef, const float* stats,
                                 const float* gamma, const float* beta, const float* film, int32_t ldfilm, int32_t B,
                                 int32_t rows_per_sample, int32_t C, int32_t G, float* scratch /* [B*C*2 + B*G*2] */, float* dh,
                                 int32_t lddh, int32_t accumulate, float* dgamma, float* dbeta, float* dfilm, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || C > 1024 || C % G || (lddz & 3) || (ldh & 3) || (lddh & 3)) return -1;
  float* P = scratch;
  float* m12 = scratch + (long long)B * C * 2;
  if (int rc = vmm_zero_async(P, sizeof(float) * B * C * 2, s)) return rc;
  const int rslots = 256 / (C >> 2);
  int blocks = max(1, min(cdiv(rows_per_sample, rslots * 8), max(1, 2048 / max(B, 1))));
  const int rpb = cdiv(rows_per_sample, blocks);
  blocks = cdiv(rows_per_sample, rpb);
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(blocks, B), dim3(256), 0, s, dz, lddz, h, ldh, coef, stats, rows_per_sample, C, G, rpb, P);
  VMM_LAUNCH_CHECK();
  const float inv_n = 1.0f / ((float)rows_per_sample * (float)(C / G));
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(cdiv(B * C, 256)), dim3(256), 0, s, P, gamma, beta, film, ldfilm, inv_n, B, C, G, m12, dgamma,
                     dbeta, dfilm);
  VMM_LAUNCH_CHECK();
  const long long rows = (long long)B * rows_per_sample;
  const int ab = (int)min((long long)cdiv(rows * (C >> 2), 256), 8192LL);
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(ab), dim3(256), 0, s, dz, lddz, h, ldh, coef, stats, m12, dh, lddh, rows, rows_per_sample, C, G,
                     accumulate);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_channel_layernorm_bwd(const float* x, int32_t ldx, const float* gamma, const float* dy, int32_t lddy, float* dx,
                                         int32_t lddx, int32_t accumulate, float* dgamma, int64_t rows, int32_t C, float eps,
                                         float* scratch, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || (ldx & 3) || (lddy & 3) || (lddx & 3) || C > 2048) return -1;
  const int c4 = C >> 2;
  int gs = 1;
  while (gs < 64 && gs < c4) gs <<= 1;
  // (at least four row groups per workgroup: the epilogue -- LDS atomics, a partial row of C floats, then vmm_sum_partials over all of them -- was
  // most of the 60 us the 12 x 12 level's launches took)
  const long long want = (rows * gs + 255) / 256;
  const int blocks = (int)max(1LL, min((want + 3) / 4, 2048LL));
#define LNB_LAUNCH(G, V)                                                                                                             \
  hipLaunchKernelGGL((chan_ln_bwd_kernel<G, V>), dim3(blocks), dim3(256), sizeof(float) * C, s, x, ldx, gamma, dy, lddy, dx, lddx, dgamma, \
                     (long long)rows, C, eps, accumulate, scratch)
#define LNB_CASE(G)                                                                                                                  \
  case G:                                                                                                                            \
    if (c4 <= G) LNB_LAUNCH(G, 1);                                                                                                   \
    else if (c4 <= 2 * G) LNB_LAUNCH(G, 2); /* (C = 512: two vectors per lane, not eight slots of which six are idle) */             \
    else if (c4 <= 4 * G) LNB_LAUNCH(G, 4);                                                                                          \
    else LNB_LAUNCH(G, 8);                                                                                                           \
    break;
  switch (gs) { LNB_CASE(1) LNB_CASE(2) LNB_CASE(4) LNB_CASE(8) LNB_CASE(16) LNB_CASE(32) LNB_CASE(64) }
#undef LNB_CASE
#undef LNB_LAUNCH
  VMM_LAUNCH_CHECK();
  if (scratch) return vmm_sum_partials(scratch, blocks, C, C, dgamma, stream);
  return 0;
}
