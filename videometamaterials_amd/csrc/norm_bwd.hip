// Backward of GroupNorm(+FiLM)+SiLU and of the channel LayerNorm (training path), channels-last rows, gfx950.
//
// GroupNorm block  z = silu(y), y = a[b,c]*h + b'[b,c]  (a = rstd*gamma*(1+scale), see norm.hip):
//   dy = dz * silu'(y);  per (sample, channel):  P1 = sum_rows dy,  P2 = sum_rows dy * hhat,  hhat = (h - mean_g) * rstd_g
//   dh = a*dy - rstd_g * (m1_g + hhat * m2_g),   m1_g = sum_{c in g} gamma'_c P1_c / n,  m2_g = sum_{c in g} gamma'_c P2_c / n
//   dgamma_c += sum_b (1+scale_bc) P2_bc ; dbeta_c += sum_b (1+scale_bc) P1_bc ; dscale_bc = gamma_c P2_bc + beta_c P1_bc ; dshift_bc = P1_bc
// Two HBM sweeps over (dz, h): reduce (per-workgroup partial rows, summed in a fixed order by the coefficient launch), then apply.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

__device__ __forceinline__ float silu_grad(float y) {
  const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-y));
  return s * (1.0f + y * (1.0f - s));
}

// Row unroll of the two sweeps: U rows of (dz, h) in flight per thread before the first exp.
constexpr int GNB_U = 4;

__host__ __device__ inline int gnb_blocks(int B, int rows_per_sample, int C) {
  const int rslots = 256 / (C >> 2) > 0 ? 256 / (C >> 2) : 1;
  int blocks = (rows_per_sample + rslots * GNB_U * 2 - 1) / (rslots * GNB_U * 2);
  const int cap = 2048 / (B > 0 ? B : 1) > 0 ? 2048 / (B > 0 ? B : 1) : 1;
  blocks = blocks < 1 ? 1 : (blocks > cap ? cap : blocks);
  const int rpb = (rows_per_sample + blocks - 1) / blocks;
  return (rows_per_sample + rpb - 1) / rpb;
}

// grid (blocks_per_sample, B); part[b][block][c] = (P1, P2) of the block's rows: one plain store per workgroup and channel, summed in a
// fixed order by gn_bwd_coef_kernel (no atomics, no zeroing launch)
__global__ __launch_bounds__(256) void gn_bwd_reduce_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ h, int ldh,
                                                            const float* __restrict__ coef, const float* __restrict__ stats, int rows_per_sample,
                                                            int C, int G, int rows_per_block, float* __restrict__ part) {
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
    const float* hp = h + (long long)b * rows_per_sample * ldh + c;
    const float* gp = dz + (long long)b * rows_per_sample * lddz + c;
    for (int r = r0 + rl; r < r1; r += rslots * GNB_U) {
      f32x4 hv[GNB_U], gv[GNB_U];
#pragma unroll
      for (int u = 0; u < GNB_U; ++u) {
        const int ru = r + u * rslots;
        const bool ok = ru < r1;
        hv[u] = ok ? *reinterpret_cast<const f32x4*>(hp + (long long)ru * ldh) : (f32x4){0.f, 0.f, 0.f, 0.f};
        gv[u] = ok ? *reinterpret_cast<const f32x4*>(gp + (long long)ru * lddz) : (f32x4){0.f, 0.f, 0.f, 0.f};  // dz = 0: no contribution
      }
#pragma unroll
      for (int u = 0; u < GNB_U; ++u) {
        const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w}, gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const float dy = gg[j] * silu_grad(a[j] * hh[j] + bb[j]);
          p1[j] += dy;
          p2[j] += dy * (hh[j] - mu[j]) * rs[j];
        }
      }
    }
  }
  __shared__ float s1[1024], s2[1024];
#pragma unroll
  for (int j = 0; j < 4; ++j) { s1[tid * 4 + j] = p1[j]; s2[tid * 4 + j] = p2[j]; }
  __syncthreads();
  float* pr = part + ((long long)b * gridDim.x + blockIdx.x) * C * 2;
  for (int ch = tid; ch < C; ch += 256) {
    const int c4 = ch >> 2, j = ch & 3;
    float t1 = 0.f, t2 = 0.f;
    for (int k = 0; k < rslots; ++k) { t1 += s1[(k * c4n + c4) * 4 + j]; t2 += s2[(k * c4n + c4) * 4 + j]; }
    *reinterpret_cast<float2*>(pr + ch * 2) = make_float2(t1, t2);
  }
}

// grid (channel chunks, B): sums the workgroup partials of CH channels (fixed order), then per (b, c): the parameter / FiLM gradients and,
// per group, the means m1, m2 the apply sweep needs.  CH is a multiple of the group width.
__global__ __launch_bounds__(256) void gn_bwd_coef_kernel(const float* __restrict__ part, int nblk, const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, const float* __restrict__ film, int ldfilm, float inv_n,
                                                          int C, int G, int CH, float* __restrict__ m12, float* __restrict__ dgamma,
                                                          float* __restrict__ dbeta, float* __restrict__ dfilm) {
  __shared__ f32x4 red[256];
  __shared__ float P[2048];
  const int b = blockIdx.y, c0 = blockIdx.x * CH, tid = threadIdx.x;
  const int nv = CH >> 1;  // f32x4 pieces of a partial row's CH (P1, P2) pairs
  const int nvl = min(nv, 256), nsl = 256 / nvl, slice = tid / nvl;
  const float* pb = part + ((long long)b * nblk * C + c0) * 2;
  // (uniform trip count: the barriers below must be reached by every thread -- with `v < nv` as the loop condition a CH whose piece count is not a
  // multiple of 256 (C = 640: nv = 320) sent the threads through different numbers of barriers; round-4 advisor)
  for (int v0 = 0; v0 < nv; v0 += nvl) {
    const int v = v0 + tid % nvl;
    const bool live = v < nv;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    if (slice < nsl && live)
      for (int k = slice; k < nblk; k += nsl) {
        const f32x4 x = *reinterpret_cast<const f32x4*>(pb + (long long)k * C * 2 + v * 4);
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
    red[tid] = acc;
    __syncthreads();
    if (slice == 0 && live) {
      for (int k = 1; k < nsl; ++k) {
        const f32x4 x = red[k * nvl + (tid % nvl)];
        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
      }
      P[v * 4] = acc.x; P[v * 4 + 1] = acc.y; P[v * 4 + 2] = acc.z; P[v * 4 + 3] = acc.w;
    }
    __syncthreads();
  }
  const int Cg = C / G;
  for (int ch = tid; ch < CH; ch += 256) {
    const int c = c0 + ch, g = c / Cg;
    const float p1 = P[ch * 2], p2 = P[ch * 2 + 1];
    const float sc1 = film ? film[(long long)b * ldfilm + c] + 1.0f : 1.0f;
    atomicAdd(&dgamma[c], sc1 * p2);
    atomicAdd(&dbeta[c], sc1 * p1);
    if (dfilm) {
      dfilm[(long long)b * ldfilm + c] = gamma[c] * p2 + beta[c] * p1;
      dfilm[(long long)b * ldfilm + C + c] = p1;
    }
    if (c == g * Cg) {
      float m1 = 0.f, m2 = 0.f;
      for (int cc = 0; cc < Cg; ++cc) {
        const float gp = gamma[c + cc] * (film ? film[(long long)b * ldfilm + c + cc] + 1.0f : 1.0f);
        m1 += gp * P[(ch + cc) * 2];
        m2 += gp * P[(ch + cc) * 2 + 1];
      }
      m12[(b * G + g) * 2] = m1 * inv_n;
      m12[(b * G + g) * 2 + 1] = m2 * inv_n;
    }
  }
}

// dh (=|+=) a*dy - rstd*(m1 + hhat*m2) = a*dy - (k0 + k1*h);  grid (blocks_per_sample, B), the per-channel constants in registers
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const float* __restrict__ dz, int lddz, const float* __restrict__ h, int ldh,
                                                           const float* __restrict__ coef, const float* __restrict__ stats,
                                                           const float* __restrict__ m12, float* __restrict__ dh, int lddh,
                                                           int rows_per_sample, int C, int G, int rows_per_block, int accumulate) {
  const int b = blockIdx.y;
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int col4 = tid % c4n, rl = tid / c4n, rslots = 256 / c4n;
  if (rl >= rslots) return;
  const int r0 = blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows_per_sample);
  const int c = col4 * 4;
  float a[4], bb[4], k0[4], k1[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    a[j] = coef[((long long)b * C + c + j) * 2];
    bb[j] = coef[((long long)b * C + c + j) * 2 + 1];
    const int g = (c + j) / (C / G);
    const float mu = stats[(b * G + g) * 2], rs = stats[(b * G + g) * 2 + 1];
    const float m1 = m12[(b * G + g) * 2], m2 = m12[(b * G + g) * 2 + 1];
    k1[j] = rs * rs * m2;
    k0[j] = rs * m1 - mu * k1[j];
  }
  const float* hp = h + (long long)b * rows_per_sample * ldh + c;
  const float* gp = dz + (long long)b * rows_per_sample * lddz + c;
  float* op = dh + (long long)b * rows_per_sample * lddh + c;
  for (int r = r0 + rl; r < r1; r += rslots * GNB_U) {
    f32x4 hv[GNB_U], gv[GNB_U], ov[GNB_U];
#pragma unroll
    for (int u = 0; u < GNB_U; ++u) {
      const int ru = min(r + u * rslots, r1 - 1);
      hv[u] = *reinterpret_cast<const f32x4*>(hp + (long long)ru * ldh);
      gv[u] = *reinterpret_cast<const f32x4*>(gp + (long long)ru * lddz);
      if (accumulate) ov[u] = *reinterpret_cast<const f32x4*>(op + (long long)ru * lddh);
    }
#pragma unroll
    for (int u = 0; u < GNB_U; ++u) {
      const int ru = r + u * rslots;
      if (ru >= r1) break;
      const float hh[4] = {hv[u].x, hv[u].y, hv[u].z, hv[u].w}, gg[4] = {gv[u].x, gv[u].y, gv[u].z, gv[u].w};
      float o[4];
#pragma unroll
      for (int j = 0; j < 4; ++j) o[j] = a[j] * (gg[j] * silu_grad(a[j] * hh[j] + bb[j])) - (k0[j] + k1[j] * hh[j]);
      if (accumulate) { o[0] += ov[u].x; o[1] += ov[u].y; o[2] += ov[u].z; o[3] += ov[u].w; }
      *reinterpret_cast<f32x4*>(op + (long long)ru * lddh) = (f32x4){o[0], o[1], o[2], o[3]};
    }
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

extern "C" int64_t vmm_groupnorm_bwd_scratch(int32_t B, int32_t rows_per_sample, int32_t C, int32_t G) {
  if (B <= 0 || rows_per_sample <= 0 || C <= 0 || (C & 3) || C > 1024 || G <= 0) return 0;
  return (int64_t)B * gnb_blocks(B, rows_per_sample, C) * C * 2 + (int64_t)B * G * 2;
}

extern "C" int vmm_groupnorm_bwd(const float* dz, int32_t lddz, const float* h, int32_t ldh, const float* coef, const float* stats,
                                 const float* gamma, const float* beta, const float* film, int32_t ldfilm, int32_t B,
                                 int32_t rows_per_sample, int32_t C, int32_t G, float* scratch /* vmm_groupnorm_bwd_scratch floats */, float* dh,
                                 int32_t lddh, int32_t accumulate, float* dgamma, float* dbeta, float* dfilm, vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  if ((C & 3) || C > 1024 || C % G || (lddz & 3) || (ldh & 3) || (lddh & 3)) return -1;
  const int blocks = gnb_blocks(B, rows_per_sample, C);
  const int rpb = cdiv(rows_per_sample, blocks);
  float* part = scratch;
  float* m12 = scratch + (long long)B * blocks * C * 2;
  hipLaunchKernelGGL(gn_bwd_reduce_kernel, dim3(blocks, B), dim3(256), 0, s, dz, lddz, h, ldh, coef, stats, rows_per_sample, C, G, rpb, part);
  VMM_LAUNCH_CHECK();
  const float inv_n = 1.0f / ((float)rows_per_sample * (float)(C / G));
  const int Cg = C / G;
  // channel chunk of a coefficient workgroup: whole groups, and as FEW of them as make eight channels -- the launch is a latency chain (a slice of a
  // workgroup adds its share of the `blocks` partial rows one after the other): with 64 channels per workgroup the 96 x 96 sites ran 4 workgroups of
  // 8 slices x 64 dependent loads (14 us a launch, 0.53 ms of the training step); with 8 channels it is 32 workgroups of 64 slices x 8 loads
  int CH = Cg;
  while (CH < 8 && CH * 2 <= C && C % (CH * 2) == 0) CH *= 2;
  if (C % CH || (CH & 1)) CH = C;
  hipLaunchKernelGGL(gn_bwd_coef_kernel, dim3(C / CH, B), dim3(256), 0, s, part, blocks, gamma, beta, film, ldfilm, inv_n, C, G, CH, m12, dgamma,
                     dbeta, dfilm);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(gn_bwd_apply_kernel, dim3(blocks, B), dim3(256), 0, s, dz, lddz, h, ldh, coef, stats, m12, dh, lddh, rows_per_sample, C, G,
                     rpb, accumulate);
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
