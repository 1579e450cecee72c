// Weight-gradient GEMM and weight (un)packing for the training path (gfx950, exact fp32 on the matrix cores).
//
// wgrad:  dWp[(tap, ci)][co] += sum_m A[(img, a*stride + dh(tap), b*stride + dw(tap)), ci] * dY[orow(m), co]
// i.e. the same gather as the forward implicit GEMM (igemm_conv.hip) with the reduction running over the
// rows m.  Each workgroup owns a 64 x 64 tile of dWp for one slice of the rows and adds it with fp32 atomics
// (the slices of one tile are spread over blockIdx.z).  LDS tiles are row-major by m, which is exactly the
// operand order of v_mfma_f32_32x32x2_f32 for A^T * dY: no transposes anywhere.
//
// pack:   one batched launch converts every torch-layout weight into the k-major operand layouts the forward /
// data-gradient GEMMs read, and (direction = 1) scatters packed weight gradients back into torch layout.
#include <stdlib.h>
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

constexpr int WK = 16;  // rows per K chunk (a multiple of 16: the loader walks groups of 16 rows)

// Workgroup tile = (64 AI) x (64 AJ) of dWp, 2 x 2 waves of (32 AI) x (32 AJ) each.  Measured on the Lagrangian training step (MI355X):
// 64 x 64 tiles with 16-row chunks (6 waves / SIMD) 18.6 ms, 128-wide tiles 19.8 ms, 128 x 128 with 32-row chunks 22.9 ms -- the
// kernel is bound by the latency of its gathered loads, which more resident waves hide better than more MFMAs per wave do; the host
// therefore launches <1, 1>.
template <int AI, int AJ>
__global__ __launch_bounds__(256) void wgrad_f32_kernel(const vmm_conv_desc p, const float* __restrict__ dy, int lddy,
                                                        float* __restrict__ dw, long long rows_per_split, float* __restrict__ bias_part) {
  constexpr int WI = 64 * AI, WJ = 64 * AJ;
  // rows padded by 32 floats: the two lane halves of an operand read (rows kk and kk + 1, same columns) land in different bank halves
  __shared__ __attribute__((aligned(16))) float As[2][WK][WI + 32];
  __shared__ __attribute__((aligned(16))) float Bs[2][WK][WJ + 32];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int Cin = p.C1 + p.C2;
  const int Ktot = p.KH * p.KW * Cin;
  const int i0 = blockIdx.x * WI, j0 = blockIdx.y * WJ;
  const long long M = (long long)p.nimg * p.Hv * p.Wv;
  const long long m_begin = (long long)blockIdx.z * rows_per_split;
  const long long m_end = min(m_begin + rows_per_split, M);
  if (m_begin >= m_end) return;
  const int nk = (int)((m_end - m_begin + WK - 1) / WK);

  // loader roles: 16 rows x 16 float4 per 64 columns of either tile
  const int lr = tid >> 4, l4 = tid & 15;
  // the A columns (tap, ci) of this thread are fixed
  bool ivalid[AI];
  int ci[AI], dh[AI], dwo[AI];
#pragma unroll
  for (int u = 0; u < AI; ++u) {
    const int ia = i0 + u * 64 + l4 * 4;
    ivalid[u] = ia < Ktot;
    const int tap = ivalid[u] ? ia / Cin : 0;
    ci[u] = ia - tap * Cin;
    const int kh = tap / p.KW, kw = tap - kh * p.KW;
    dh[u] = p.off_h + p.sgn_h * kh;
    dwo[u] = p.off_w + p.sgn_w * kw;
  }
  const bool identity_rows = (p.oscale == 1 && p.Hout == p.Hv && p.Wout == p.Wv && p.ooh == 0 && p.oow == 0);
  const int hw = p.Hv * p.Wv;

  constexpr int NG = WK / 16;  // row groups per chunk (rows lr, lr + 16, ...)
  f32x4 areg[NG][AI], breg[NG][AJ];
  f32x4 bsum[AJ];  // column sums of this thread's dY pieces (the bias gradient, taken by the blockIdx.x == 0 tiles)
#pragma unroll
  for (int v = 0; v < AJ; ++v) bsum[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // (img, a, b) of this thread's row of the next chunk, advanced by WK rows per chunk without divisions
  long long m = m_begin + lr;
  int img = (int)(m / hw);
  int a = (int)(m - (long long)img * hw) / p.Wv, b = (int)(m - (long long)img * hw) - a * p.Wv;
  // fused GroupNorm + SiLU of the producer: applied in store_chunk, not next to the load (the prefetch must stay in flight across the MFMAs)
  f32x4 fc0[NG][AI], fc1[NG][AI];
  unsigned fmask = 0;
  auto load_rows = [&](f32x4 (&areg)[AI], f32x4 (&breg)[AJ], int g) {
#pragma unroll
    for (int u = 0; u < AI; ++u) areg[u] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int v = 0; v < AJ; ++v) breg[v] = (f32x4){0.f, 0.f, 0.f, 0.f};
    if (m < m_end) {
#pragma unroll
      for (int u = 0; u < AI; ++u) {
        if (!ivalid[u]) continue;
        int ih = a * p.stride + dh[u], iw = b * p.stride + dwo[u];
        if (p.wrap_h) ih = ih < 0 ? ih + p.Hin : (ih >= p.Hin ? ih - p.Hin : ih);  // periodic padding (vddp.py:163-243)
        if (p.wrap_w) iw = iw < 0 ? iw + p.Win : (iw >= p.Win ? iw - p.Win : iw);
        if (ih >= 0 && ih < p.Hin && iw >= 0 && iw < p.Win) {
          const long long pix = ((long long)img * p.Hin + ih) * p.Win + iw;
          if (ci[u] < p.C1) {
            areg[u] = *reinterpret_cast<const f32x4*>(p.a1 + pix * p.lda1 + ci[u]);
            if (p.a_mode == 1) {  // coefficients fetched with the row, applied when it is consumed (store_chunk)
              const float* cf = p.a_coef + ((long long)(img / p.a_imgs_per_sample) * p.C1 + ci[u]) * 2;
              fc0[g][u] = *reinterpret_cast<const f32x4*>(cf);
              fc1[g][u] = *reinterpret_cast<const f32x4*>(cf + 4);
              fmask |= 1u << (g * AI + u);
            }
          } else {
            areg[u] = *reinterpret_cast<const f32x4*>(p.a2 + pix * p.lda2 + (ci[u] - p.C1));
          }
        }
      }
      long long orow = m;
      if (!identity_rows) orow = ((long long)img * p.Hout + a * p.oscale + p.ooh) * p.Wout + b * p.oscale + p.oow;
#pragma unroll
      for (int v = 0; v < AJ; ++v) {
        const int jb = j0 + v * 64 + l4 * 4;
        if (jb < p.Cout) breg[v] = *reinterpret_cast<const f32x4*>(dy + orow * lddy + jb);
      }
    }
    m += 16;
    b += 16;
    while (b >= p.Wv) {
      b -= p.Wv;
      if (++a == p.Hv) { a = 0; ++img; }
    }
  };
  auto load_chunk = [&]() {
    fmask = 0;
#pragma unroll
    for (int g = 0; g < NG; ++g) load_rows(areg[g], breg[g], g);
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int g = 0; g < NG; ++g) {
#pragma unroll
      for (int u = 0; u < AI; ++u) {
        if (fmask >> (g * AI + u) & 1) {
          const f32x4 c0 = fc0[g][u], c1 = fc1[g][u];
          areg[g][u].x = silu_rcp(areg[g][u].x * c0.x + c0.y);
          areg[g][u].y = silu_rcp(areg[g][u].y * c0.z + c0.w);
          areg[g][u].z = silu_rcp(areg[g][u].z * c1.x + c1.y);
          areg[g][u].w = silu_rcp(areg[g][u].w * c1.z + c1.w);
        }
        *reinterpret_cast<f32x4*>(&As[buf][g * 16 + lr][u * 64 + l4 * 4]) = areg[g][u];
      }
#pragma unroll
      for (int v = 0; v < AJ; ++v) {
        *reinterpret_cast<f32x4*>(&Bs[buf][g * 16 + lr][v * 64 + l4 * 4]) = breg[g][v];
        bsum[v] += breg[g][v];  // (here, not next to the load: the prefetch must stay in flight across the MFMAs)
      }
    }
  };

  f32x16 acc[AI][AJ];
#pragma unroll
  for (int u = 0; u < AI; ++u)
#pragma unroll
    for (int v = 0; v < AJ; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[u][v][r] = 0.f;
  load_chunk();
  store_chunk(0);
  __syncthreads();
  const int l31 = lane & 31, lk = lane >> 5;
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) load_chunk();
#pragma unroll
    for (int kk = 0; kk < WK; kk += 2) {
      float av[AI], bv[AJ];
#pragma unroll
      for (int u = 0; u < AI; ++u) av[u] = As[buf][kk + lk][(wi * AI + u) * 32 + l31];
#pragma unroll
      for (int v = 0; v < AJ; ++v) bv[v] = Bs[buf][kk + lk][(wj * AJ + v) * 32 + l31];
#pragma unroll
      for (int u = 0; u < AI; ++u)
#pragma unroll
        for (int v = 0; v < AJ; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[v], acc[u][v], 0, 0, 0);
    }
    if (kc + 1 < nk) store_chunk(buf ^ 1);
    __syncthreads();
  }
#pragma unroll
  for (int u = 0; u < AI; ++u)
#pragma unroll
    for (int v = 0; v < AJ; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + (wi * AI + u) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        const int j = j0 + (wj * AJ + v) * 32 + l31;
        if (i < Ktot && j < p.Cout) atomicAdd(&dw[(long long)i * p.Cout + j], acc[u][v][r]);
      }
  if (bias_part && blockIdx.x == 0) {  // (workgroup-uniform) column sums of this row slice of dY -> bias_part[slice][co]
    float* red = &As[0][0][0];         // [16 loader rows][64 columns] per 64-column group; the last chunk's barrier is behind us
#pragma unroll
    for (int v = 0; v < AJ; ++v) {
      if (v) __syncthreads();
      *reinterpret_cast<f32x4*>(red + lr * 64 + l4 * 4) = bsum[v];
      __syncthreads();
      if (tid < 64 && j0 + v * 64 + tid < p.Cout) {
        float t = 0.f;
#pragma unroll
        for (int k = 0; k < 16; ++k) t += red[k * 64 + tid];
        bias_part[(long long)blockIdx.z * p.Cout + j0 + v * 64 + tid] = t;
      }
    }
  }
}

// column sums: out[j] += sum_m x[m, j]   (bias gradients; also per-channel reductions elsewhere)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, long long rows, int C, float* __restrict__ out,
                                                     long long rows_per_block) {
  const int c4n = C >> 2;
  const int tid = threadIdx.x;
  const int col4 = tid % c4n, rl = tid / c4n, rslots = 256 / c4n;
  const long long r0 = (long long)blockIdx.x * rows_per_block, r1 = min(r0 + rows_per_block, rows);
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (rl < rslots) {
    long long r = r0 + rl;
    f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1, s3 = s1;
    for (; r + 3 * rslots < r1; r += 4 * rslots) {  // four independent loads in flight
      const f32x4 v0 = *reinterpret_cast<const f32x4*>(x + r * ldx + col4 * 4);
      const f32x4 v1 = *reinterpret_cast<const f32x4*>(x + (r + rslots) * ldx + col4 * 4);
      const f32x4 v2 = *reinterpret_cast<const f32x4*>(x + (r + 2 * rslots) * ldx + col4 * 4);
      const f32x4 v3 = *reinterpret_cast<const f32x4*>(x + (r + 3 * rslots) * ldx + col4 * 4);
      s += v0; s1 += v1; s2 += v2; s3 += v3;
    }
    for (; r < r1; r += rslots) s += *reinterpret_cast<const f32x4*>(x + r * ldx + col4 * 4);
    s += s1; s2 += s3; s += s2;
  }
  __shared__ f32x4 sh[256];
  sh[tid] = s;
  __syncthreads();
  if (tid < c4n) {
    f32x4 t = {0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < rslots; ++k) { const f32x4 v = sh[k * c4n + tid]; t.x += v.x; t.y += v.y; t.z += v.z; t.w += v.w; }
    atomicAdd(&out[tid * 4 + 0], t.x); atomicAdd(&out[tid * 4 + 1], t.y);
    atomicAdd(&out[tid * 4 + 2], t.z); atomicAdd(&out[tid * 4 + 3], t.w);
  }
}

// out[c] += sum_{k < n} part[k * ld + c]: the second stage of the reductions that leave one partial row per workgroup (same-address
// global atomics cost ~100 ns each on this part, so a thousand workgroups adding into one vector serialise for ~100 us).
// Workgroup = 16 columns x 16 slices of k; fixed summation order.
__global__ __launch_bounds__(256) void sum_partials_kernel(const float* __restrict__ part, int n, int ld, int C, float* __restrict__ out) {
  __shared__ float red[16][17];
  const int e = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* src = part + c;
    int k = kg;
    for (; k + 48 < n; k += 64) {
      s0 += src[(long long)k * ld];
      s1 += src[(long long)(k + 16) * ld];
      s2 += src[(long long)(k + 32) * ld];
      s3 += src[(long long)(k + 48) * ld];
    }
    for (; k < n; k += 16) s0 += src[(long long)k * ld];
  }
  red[kg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e];
    out[c] += t;
  }
}

// Gradient scatter (direction 1) of the plain layouts -- torch (N, C, TH, TW) contiguous, i.e. torch index n C T + c T + t with T = TH TW -- through
// an LDS tile: packed[(t, c)][n] is read in 256-byte runs along n, torch is written in runs of CB T floats along (c, t).  (The element-wise
// path below reads the packed buffer coalesced but scatters 4-byte read-modify-writes T or C T floats apart: 1.2 ms per training step.)
__device__ __forceinline__ int unpack_tile_cb(const vmm_pack_job& jb) {
  const int T = jb.TH * jb.TW;
  const bool plain = jb.fmt == 0 && jb.Cp == jb.C && jb.h0 == 0 && jb.w0 == 0 && jb.hs == 1 && jb.ws == 1 && jb.sw == 1 && jb.sh == jb.TW && jb.sc == T &&
                     jb.sn == jb.C * T && jb.N % 64 == 0 && (((uintptr_t)jb.torch_w | (uintptr_t)jb.packed) & 15) == 0;
  if (!plain) return 0;
  const int cb = T == 1 ? 64 : (T <= 16 ? 16 : 0);
  return (cb && jb.C % cb == 0) ? cb : 0;
}
constexpr int UNPACK_TILE_FLOATS = 16 * 16 * 64;  // T * CB * 64 <= this (T = 9 / 16 with CB = 16, T = 1 with CB = 64)

__global__ __launch_bounds__(256) void unpack_tiled_kernel(const vmm_pack_job* __restrict__ jobs) {
  __shared__ float tile[UNPACK_TILE_FLOATS + 64];
  const vmm_pack_job jb = jobs[blockIdx.y];
  const int CB = unpack_tile_cb(jb);
  if (!CB) return;
  const int T = jb.TH * jb.TW, run = CB * T;            // floats per (n, c block) run in torch layout; a multiple of 16
  const int nb = jb.N / 64, ntiles = nb * (jb.C / CB);
  for (int tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
    const int n0 = (tl % nb) * 64, c0 = (tl / nb) * CB;
    // packed rows (t, c0 + c): 64 floats at n0 -> tile[(c T + t)][n]  (+1 padding per 64: the transposed reads below walk the rows)
    for (int i = threadIdx.x; i < run * 16; i += 256) {
      const int n4 = i & 15, r = i >> 4;                // r = t * CB + c
      const int t = r / CB, c = r - t * CB;
      const f32x4 v = *reinterpret_cast<const f32x4*>(jb.packed + ((long long)t * jb.Cp + c0 + c) * jb.N + n0 + 4 * n4);
      float* d = tile + (c * T + t) * 65 + 4 * n4;
      d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
    }
    __syncthreads();
    // torch run of sample n: run floats at n sn + c0 T; thread = (n, 16-byte piece)
    const int pieces = run / 4;
    for (int i = threadIdx.x; i < 64 * pieces; i += 256) {
      const int n = i / pieces, q = i - n * pieces;
      float* o = jb.torch_w + (long long)(n0 + n) * jb.sn + (long long)c0 * T + 4 * q;
      f32x4 v = {tile[(4 * q) * 65 + n], tile[(4 * q + 1) * 65 + n], tile[(4 * q + 2) * 65 + n], tile[(4 * q + 3) * 65 + n]};
      if (jb.accumulate) v += *reinterpret_cast<const f32x4*>(o);
      *reinterpret_cast<f32x4*>(o) = v;
    }
    __syncthreads();
  }
}

// batched (un)pack: packed[(th*TW + tw)*Cp + c][n]  <->  torch[ n*sn + c*sc + (h0 + th*hs)*sh + (w0 + tw*ws)*sw ]
// the two 16-bit planes of an operand value: split bf16 hi | lo, or -- operand planes of the fp16-operand kernels, fmt | 16 -- IEEE half | 0
__device__ __forceinline__ void pack16(float v, int half, unsigned short& hi, unsigned short& lo) {
  if (half) {
    const _Float16 h = (_Float16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = half == 2 ? __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h)) : (unsigned short)0;  // (fmt | 32: fp16 hi | fp16 lo, the `_f16x3` entry points)
  } else {
    const __bf16 h = (__bf16)v;
    hi = __builtin_bit_cast(unsigned short, h);
    lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
  }
}

__global__ __launch_bounds__(256) void pack_kernel(const vmm_pack_job* __restrict__ jobs, int direction) {
  vmm_pack_job jb = jobs[blockIdx.y];
  const int half = (jb.fmt & 32) ? 2 : ((jb.fmt & 16) ? 1 : 0);  // formats 1, 2, 3, 5, 6 with fp16 planes: | 16 = fp16 | 0 (`_fp16` entry points), | 32 = fp16 hi | fp16 lo (`_f16x3`)
  jb.fmt &= 15;
  if (direction == 1 && unpack_tile_cb(jb)) return;  // (scattered by unpack_tiled_kernel)
  if (jb.fmt == 1) {
    // split-bf16 operand for igemm_bf16x3.hip: [N][Kpad] hi plane then lo plane, K = (th, tw, c) padded to a multiple of 32
    if (direction != 0) return;
    const int K = jb.TH * jb.TW * jb.Cp;
    const int Kpad = (K + 31) / 32 * 32;
    unsigned short* hi = reinterpret_cast<unsigned short*>(jb.packed);
    unsigned short* lo = hi + (long long)jb.N * Kpad;
    const long long tot = (long long)jb.N * Kpad;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int n = (int)(i / Kpad), k = (int)(i - (long long)n * Kpad);
      float v = 0.f;
      if (k < K) {
        const int c = k % jb.Cp, t = k / jb.Cp;
        const int tw = t % jb.TW, th = t / jb.TW;
        if (c < jb.C) v = jb.torch_w[(long long)n * jb.sn + (long long)c * jb.sc + (long long)(jb.h0 + th * jb.hs) * jb.sh + (long long)(jb.w0 + tw * jb.ws) * jb.sw];
      }
      pack16(v, half, hi[i], lo[i]);  // (fmt 1 | 16: IEEE-half hi plane, zero lo plane -- vmm_conv_igemm_fp16)
    }
    return;
  }
  if (jb.fmt == 7) {
    // stem convolution (stem_conv.hip): fmt-2 fragment planes with K = (kernel row, tap 0..7, channel 0..3); taps >= TW and channels >= C are zero
    if (direction != 0) return;
    const int KS = 2 * jb.TH, NT = jb.N / 32;
    unsigned short* dst = reinterpret_cast<unsigned short*>(jb.packed);
    const long long tot = (long long)NT * KS * 512;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int e = (int)(i & 7), l = (int)(i >> 3) & 63;
      const long long pl = i >> 9;
      const int ks = (int)(pl % KS), nt = (int)(pl / KS);
      const int n = nt * 32 + (l & 31);
      const int k = ks * 16 + (l >> 5) * 8 + e;
      const int th = k >> 5, tw = (k >> 2) & 7, c = k & 3;
      float v = 0.f;
      if (tw < jb.TW && c < jb.C) v = jb.torch_w[(long long)n * jb.sn + (long long)c * jb.sc + (long long)th * jb.sh + (long long)tw * jb.sw];
      const __bf16 h = (__bf16)v;
      const __bf16 lo = (__bf16)(v - (float)h);
      const long long o = pl * 1024 + l * 8 + e;
      dst[o] = __builtin_bit_cast(unsigned short, h);
      dst[o + 512] = __builtin_bit_cast(unsigned short, lo);
    }
    return;
  }
  if (jb.fmt == 5 || jb.fmt == 6) {
    // The (1,4,4) stride-2 resampling kernels as fmt-2 fragment planes of a 3 x 3 convolution (conv3x3_bf16x3.hip, TS variants); TH = TW = 4.
    //   5 (Downsample, torch (N, C, 1, 4, 4)): K = 9 taps x 4 C cell channels, cell channel = (2 sy + sx) C + c of sub-pixel (sy, sx);
    //     tap (th, tw) of sub-pixel (sy, sx) is kernel element (2 th + sy - 1, 2 tw + sx - 1), zero outside the kernel;
    //   6 (Upsample, torch (C, N, 1, 4, 4)): 4 N columns, column = (2 py + px) N + n of output phase (py, px), K = 9 taps x C;
    //     tap (th, tw) of phase (py, px) is kernel element ((py ? 4 : 3) - 2 th, (px ? 4 : 3) - 2 tw), zero outside the kernel.
    if (direction != 0) return;
    const int Cc = jb.fmt == 5 ? 4 * jb.C : jb.C, Nc = jb.fmt == 6 ? 4 * jb.N : jb.N;
    const int KS = 9 * Cc / 16, NT = Nc / 32;
    unsigned short* dst = reinterpret_cast<unsigned short*>(jb.packed);
    const long long tot = (long long)NT * KS * 512;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int e = (int)(i & 7), l = (int)(i >> 3) & 63;
      const long long pl = i >> 9;
      const int ks = (int)(pl % KS), nt = (int)(pl / KS);
      const int ncol = nt * 32 + (l & 31);
      const int k = ks * 16 + (l >> 5) * 8 + e;
      const int tap = k / Cc, cc = k - tap * Cc, th = tap / 3, tw = tap - th * 3;
      int n, c, kh, kw;
      if (jb.fmt == 5) {
        const int sub = cc / jb.C;
        n = ncol; c = cc - sub * jb.C;
        kh = 2 * th + (sub >> 1) - 1; kw = 2 * tw + (sub & 1) - 1;
      } else {
        const int ph = ncol / jb.N;
        n = ncol - ph * jb.N; c = cc;
        kh = ((ph >> 1) ? 4 : 3) - 2 * th; kw = ((ph & 1) ? 4 : 3) - 2 * tw;
      }
      float v = 0.f;
      if (kh >= 0 && kh < 4 && kw >= 0 && kw < 4) v = jb.torch_w[(long long)n * jb.sn + (long long)c * jb.sc + (long long)kh * jb.sh + (long long)kw * jb.sw];
      const long long o = pl * 1024 + l * 8 + e;
      pack16(v, half, dst[o], dst[o + 512]);
    }
    return;
  }
  if (jb.fmt == 8) {
    // Winograd F(2x2, 3x3) weights for conv3x3_wino.hip: U = G g G^T (G = [[1, 0, 0], [1/2, 1/2, 1/2], [1/2, -1/2, 1/2], [0, 0, 1]]) per (channel, column),
    // split, in "A" fragment order: plane (column block nb of 64, k16 step ks, position xi * 4 + nu, column fragment mf, hi | lo) = 64 lanes x 8 bf16,
    // lane l = column nb * 64 + mf * 32 + (l & 31), channels ks * 16 + (l >> 5) * 8 .. + 7.  thread = (nb, ks, mf, l): 72 loads, 32 16-byte stores.
    if (direction != 0) return;
    const int KS = jb.Cp / 16, NB = (jb.N + 63) / 64;
    uint4* dst = reinterpret_cast<uint4*>(jb.packed);
    const long long tot = (long long)NB * KS * 2 * 64;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int l = (int)(i & 63), mf = (int)(i >> 6) & 1;
      const long long pk = i >> 7;  // nb * KS + ks
      const int ks = (int)(pk % KS), nb = (int)(pk / KS);
      const int n = nb * 64 + mf * 32 + (l & 31), c0 = ks * 16 + (l >> 5) * 8;
      float tr[8][4][3];  // G g: [channel][xi][j]
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        float g[3][3];
#pragma unroll
        for (int th = 0; th < 3; ++th)
#pragma unroll
          for (int tw = 0; tw < 3; ++tw)
            g[th][tw] = (c0 + e < jb.C && n < jb.N) ? jb.torch_w[(long long)n * jb.sn + (long long)(c0 + e) * jb.sc + (long long)(jb.h0 + th * jb.hs) * jb.sh +
                                                                 (long long)(jb.w0 + tw * jb.ws) * jb.sw]
                                                    : 0.f;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          tr[e][0][j] = g[0][j];
          tr[e][1][j] = 0.5f * ((g[0][j] + g[2][j]) + g[1][j]);
          tr[e][2][j] = 0.5f * ((g[0][j] + g[2][j]) - g[1][j]);
          tr[e][3][j] = g[2][j];
        }
      }
#pragma unroll
      for (int xi = 0; xi < 4; ++xi)
#pragma unroll
        for (int nu = 0; nu < 4; ++nu) {
          unsigned h[4], lo[4];
#pragma unroll
          for (int e2 = 0; e2 < 4; ++e2) {
            float u[2];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
              const float* t = tr[2 * e2 + k][xi];
              u[k] = nu == 0 ? t[0] : nu == 3 ? t[2] : nu == 1 ? 0.5f * ((t[0] + t[2]) + t[1]) : 0.5f * ((t[0] + t[2]) - t[1]);
            }
            h[e2] = split_bf16_pair(u[0], u[1], lo[e2]);
          }
          uint4* o = dst + ((((pk * 16 + xi * 4 + nu) * 2 + mf) * 2) * 64 + l);
          o[0] = uint4{h[0], h[1], h[2], h[3]};
          o[64] = uint4{lo[0], lo[1], lo[2], lo[3]};
        }
    }
    return;
  }
  if (jb.fmt == 2 || jb.fmt == 3 || jb.fmt == 4) {
    // split-bf16 operand in MFMA fragment order for conv3x3_bf16x3.hip / proj_bf16x3.hip: plane (column tile nt of 32, k16 step ks, hi|lo)
    // = 64 lanes x 8 bf16; lane l holds column nt*32 + (l & 31), k = ks*16 + (l >> 5)*8 .. +7.  K = (th, tw, c) padded to 32, N to 32.
    // fmt 3: within every 32-block of k the contraction index follows the accumulator-register order of a preceding MFMA
    // (linattn_block.hip): element e of half lk in step s <-> k = (e & 3) + 8 (2 s + (e >> 2)) + 4 lk.
    // fmt 4: the same fragment order in fp32 for the exact variants (vmm_conv3x3_f32 / vmm_proj_f32): the two 1 KiB planes of a
    // (nt, ks) pair hold elements 0..3 and 4..7 of every lane's eight k values as float4.
    if (direction != 0) return;
    const int K = jb.TH * jb.TW * jb.Cp;
    const int KS = (K + 31) / 32 * 2;
    const int NT = (jb.N + 31) / 32;
    unsigned short* dst = reinterpret_cast<unsigned short*>(jb.packed);
    const long long tot = (long long)NT * KS * 512;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < tot; i += (long long)gridDim.x * blockDim.x) {
      const int e = (int)(i & 7), l = (int)(i >> 3) & 63;
      const long long pl = i >> 9;  // nt*KS + ks
      const int ks = (int)(pl % KS), nt = (int)(pl / KS);
      const int n = nt * 32 + (l & 31);
      const int k = jb.fmt != 3 ? ks * 16 + (l >> 5) * 8 + e : (ks >> 1) * 32 + (e & 3) + 8 * (2 * (ks & 1) + (e >> 2)) + 4 * (l >> 5);
      float v = 0.f;
      if (k < K && n < jb.N) {
        const int c = k % jb.Cp, t = k / jb.Cp;
        const int tw = t % jb.TW, th = t / jb.TW;
        if (c < jb.C) v = jb.torch_w[(long long)n * jb.sn + (long long)c * jb.sc + (long long)(jb.h0 + th * jb.hs) * jb.sh + (long long)(jb.w0 + tw * jb.ws) * jb.sw];
      }
      if (jb.fmt == 4) {
        jb.packed[pl * 512 + (e >> 2) * 256 + l * 4 + (e & 3)] = v;
        continue;
      }
      const long long o = pl * 1024 + l * 8 + e;
      pack16(v, half, dst[o], dst[o + 512]);
    }
    return;
  }
  const long long total = (long long)jb.TH * jb.TW * jb.Cp * jb.N;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const int n = (int)(i % jb.N);
    long long k = i / jb.N;
    const int c = (int)(k % jb.Cp);
    k /= jb.Cp;
    const int tw = (int)(k % jb.TW), th = (int)(k / jb.TW);
    const long long so = (long long)n * jb.sn + (long long)c * jb.sc + (long long)(jb.h0 + th * jb.hs) * jb.sh + (long long)(jb.w0 + tw * jb.ws) * jb.sw;
    if (direction == 0) {
      jb.packed[i] = (c < jb.C) ? jb.torch_w[so] : 0.f;
    } else if (c < jb.C) {
      if (jb.accumulate) jb.torch_w[so] += jb.packed[i]; else jb.torch_w[so] = jb.packed[i];
    }
  }
}

}  // namespace

extern "C" int vmm_sum_partials(const float* part, int32_t n, int32_t ld, int32_t C, float* out, vmm_stream_t stream) {
  if (n <= 0 || C <= 0) return 0;
  hipLaunchKernelGGL(sum_partials_kernel, dim3(cdiv(C, 16)), dim3(256), 0, (hipStream_t)stream, part, n, ld, C, out);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_conv_wgrad_f32(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                                  float* bias_scratch, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  if ((d.C1 & 3) || (d.C2 & 3) || (d.lda1 & 3) || (d.C2 && (d.lda2 & 3)) || (d.Cout & 3) || (lddy & 3) || nsplit < 1 || (dbias && !bias_scratch)) return -1;
  {  // the 3 x 3 "same" convolutions: all nine taps from one LDS patch (wgrad3x3.hip); VMM_WGRAD3X3=0 keeps the generic kernel (A/B runs)
    static const bool use3 = [] { const char* e = getenv("VMM_WGRAD3X3"); return !e || e[0] != '0'; }();
    if (use3) {
      const int rc = vmm_conv3x3_wgrad_f32(dp, dy, lddy, dw_packed, nsplit, dbias, bias_scratch, stream);
      if (rc != 1) return rc;
    }
  }
  const long long M = (long long)d.nimg * d.Hv * d.Wv;
  const int Ktot = d.KH * d.KW * (d.C1 + d.C2);
  const long long rps = (cdiv(M, nsplit) + WK - 1) / WK * WK;
  const int ai = 1, aj = 1;  // (see the note at the kernel)
  dim3 grid(cdiv(Ktot, 64 * ai), cdiv(d.Cout, 64 * aj), cdiv(M, rps));
  float* bp = dbias ? bias_scratch : nullptr;
  hipStream_t s = (hipStream_t)stream;
  if (ai == 2 && aj == 2) hipLaunchKernelGGL((wgrad_f32_kernel<2, 2>), grid, dim3(256), 0, s, d, dy, lddy, dw_packed, rps, bp);
  else if (ai == 2) hipLaunchKernelGGL((wgrad_f32_kernel<2, 1>), grid, dim3(256), 0, s, d, dy, lddy, dw_packed, rps, bp);
  else if (aj == 2) hipLaunchKernelGGL((wgrad_f32_kernel<1, 2>), grid, dim3(256), 0, s, d, dy, lddy, dw_packed, rps, bp);
  else hipLaunchKernelGGL((wgrad_f32_kernel<1, 1>), grid, dim3(256), 0, s, d, dy, lddy, dw_packed, rps, bp);
  VMM_LAUNCH_CHECK();
  if (dbias) return vmm_sum_partials(bias_scratch, (int)grid.z, d.Cout, d.Cout, dbias, stream);
  return 0;
}

extern "C" int vmm_colsum_accumulate(const float* x, int32_t ldx, int64_t rows, int32_t C, float* out, vmm_stream_t stream) {
  if ((C & 3) || (ldx & 3) || C > 1024) return -1;
  const int rslots = 256 / (C >> 2);
  // (at most one workgroup per CU: every workgroup ends with C same-address atomics, which serialise at ~100 ns each -- with 1 024 workgroups the
  // 96 x 96 launches took 127 us, most of it in that queue; 57 us with 256)
  long long blocks = min((long long)cdiv(rows, rslots * 16), 256LL);
  if (blocks < 1) blocks = 1;
  const long long rpb = cdiv(rows, blocks);
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(rows, rpb)), dim3(256), 0, (hipStream_t)stream, x, ldx, (long long)rows, C, out, rpb);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_pack_weights(const vmm_pack_job* jobs_dev, int32_t njobs, int32_t max_elems, int32_t direction, vmm_stream_t stream) {
  if (njobs <= 0) return 0;
  // few jobs (the per-bucket gradient scatters of the backward pass): more workgroups per job so that the launch still fills the chip
  const long long cap = max(64LL, 2048LL / njobs);
  const int bx = (int)max(1LL, min((long long)cdiv(max_elems, 256 * 4), cap));
  hipLaunchKernelGGL(pack_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, direction);
  VMM_LAUNCH_CHECK();
  if (direction == 1) {  // the plain conv / linear layouts leave through LDS tiles (jobs outside that envelope were handled above)
    const int tx = (int)max(1LL, min((long long)cdiv(max_elems, 64 * 144), cap));
    hipLaunchKernelGGL(unpack_tiled_kernel, dim3(tx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
