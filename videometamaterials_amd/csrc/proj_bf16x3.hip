// 1x1 convolution / Linear over rows with an A-stationary LDS tile and register-fed weights, split-bf16 MFMA, gfx950.
//
// The attention projections (to_qkv, to_out: vddp.py:319,325,413,421) and the ResnetBlock shortcut (res_conv, vddp.py:297) are
// skinny-K GEMMs: K = 64 ... 256 input channels against up to 768 output columns.  They are bound by their OUTPUT traffic, and the
// generic implicit-GEMM kernel (weights staged through LDS, one workgroup barrier per 32-k chunk, scalar epilogue stores) ran them
// at 3-5x their HBM time.  Here a workgroup owns a tile of rows for ALL output columns:
//   A  the rows' full K extent is staged ONCE in LDS as bf16 hi | lo -- optionally through the channel LayerNorm of PreNorm
//      (vddp.py:245-254, a_mode 2: the rows are complete here, so the separate LayerNorm pass and its round trip disappear);
//   B  weights in MFMA fragment order (vmm_pack_weights fmt 2) go straight from L2 into registers, one k16 step ahead;
//   the workgroup then sweeps the output columns in chunks with no barrier at all; each chunk's accumulators (output channels as
//   MFMA rows, so a lane owns 4 x 4 consecutive channels of one row) leave as 16-byte stores while the next chunk's MFMAs issue.
// Epilogue: bias, q-scale, rotary (temporal to_qkv, vddp.py:449,456), residual -- as in igemm_common.h.
// Wave tile 64 rows x 64 columns; workgroup 4 waves as 4x1 / 2x2 / 1x4 so that the LDS tile (rows x (4 K + 16) bytes) stays under
// 80 KB and two workgroups share a CU (one stages / stores while the other computes).
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct PJArgs {
  vmm_conv_desc p;
  const float* gamma;  // a_mode 2: LayerNorm weight [K]
  float eps;
  int M;               // rows
  int n_chunks;        // column chunks of WN*64
  int chunks_per_y;    // column chunks per blockIdx.y
  unsigned long long* trace = nullptr;  // VMM_PJ_TRACE=<launch>: wave 0 of every workgroup stamps s_memtime at its phase boundaries (18 slots)
  const float* res_coef;  // non-NULL: the residual enters as silu(res * a + b'), (a, b') = res_coef[sample][column][2] (vmm_proj_bf16x3_res_silu)
  int res_rps;            // rows per sample
  float* ln_stats = nullptr;  // a_mode 2: (mean, 1 / sqrt(var + eps)) of every row leave here as well [M][2] (the weight gradient normalises x with them)
};

__device__ __forceinline__ unsigned pj_split(float a, float b, unsigned& lo) { return split_bf16_pair(a, b, lo); }

// KS = k16 steps (K padded to 32 -> KS = Kpad / 16 in {2, 4, 8, 16})
// KSPLIT4: the four waves split the k16 steps of ONE column slice (Cout <= 64 under the 1 x 4 arrangement).  A template parameter, not a
// run-time branch: with both step loops in one kernel the accumulators live in different registers in the two copies and the K = 256
// instance spilled 102-109 registers to scratch (every launch of it, whichever path it took).
#ifndef VMM_PJ_DBG
#define VMM_PJ_DBG 0
#endif
// ONE: the "bf16" throughput mode (BASELINE.json configs[3]): hi planes only, one matrix pass per product
// A16 (single-pass instances): the rows, the residual and the output are bf16-STORED feature maps (the "bf16" mode's two upper levels): half-width
// staging loads, one rounding per stored output element; LayerNorm, q-scale, rotary, bias and the residual sum stay fp32.
template <int WM, int WN, int KS, bool F32, bool KSPLIT4 = false, bool ONE = false, bool A16 = false, bool KSPLIT2 = false>
__global__ __launch_bounds__(256, 2) void proj_x3_kernel(const PJArgs a) {
  static_assert(!KSPLIT2 || (WN == 2 && !KSPLIT4 && !A16 && KS % 4 == 0), "k split between the two column waves of the 2 x 2 arrangement");
  static_assert(!ONE || !F32, "single pass: bf16 operands");
  static_assert(!A16 || (ONE && !KSPLIT4), "bf16-stored maps: the single-pass instances without the k split");
  constexpr int BM = WM * 64, BN = WN * 64;
  constexpr int KP = KS * 16;         // padded K
  constexpr int PITCH = 2 * KP + 8;   // bf16 per LDS row: hi[KP] | lo[KP] | pad  ((4 KP + 16) / 16 is odd: conflict-free ds_read_b128)
  constexpr int IPL = (KP + 63) / 64; // float4 items per lane of a 16-lane row group
  constexpr int PASSES = BM / 16;     // 256 threads = 16 rows per pass
  extern __shared__ __attribute__((aligned(16))) unsigned short At[];
  const vmm_conv_desc& p = a.p;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn_id = wave % WN;
  const int lrow = lane & 31, lk = lane >> 5;
  const int m0 = blockIdx.x * BM;
  const int K = p.C1 + p.C2;
  const int nc_begin = blockIdx.y * a.chunks_per_y, nc_end = min(a.n_chunks, nc_begin + a.chunks_per_y);
  if (nc_begin >= nc_end) return;
  // (compiled in by -DVMM_PJ_TRACE_BUILD=1 only -- tools/build_ab.py proj_bf16x3 -DVMM_PJ_TRACE_BUILD=1 + VMM_LIB_PATH: the run-time pointer test keeps
  // the pointer and the slot index in registers for the whole kernel, see conv3x3_bf16x3.hip)
#ifndef VMM_PJ_TRACE_BUILD
#define VMM_PJ_TRACE_BUILD 0
#endif
  auto stamp = [&](int k) {  // measurement aid (tools/trace_proj.py)
    if constexpr (VMM_PJ_TRACE_BUILD)
      if (a.trace && tid == 0 && k < 18) a.trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 18 + k] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ---- stage the row tile: 16 lanes per row, lane l16 owns channels (l16 + 16 i) * 4 .. +3
  {
    const int l16 = tid & 15, r0 = tid >> 4;
    // Straight-line: every item is loaded (rows past M re-read the last row, channels past K re-read channel 0) and zeroed by a select
    // afterwards.  With `if (m < M && c < K)` around the loads and a branch per source the staging was ~50 exec-mask branches before the
    // first barrier (conv3x3_bf16x3.hip, same finding: tools/trace_c3.py).
    f32x4 v[PASSES][IPL];
    const bool two = p.C2 > 0;  // wave-uniform
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      const int m = m0 + ps * 16 + r0;
      const long long mc = min(m, a.M - 1);
#pragma unroll
      for (int i = 0; i < IPL; ++i) {
        const int c = (l16 + 16 * i) * 4;
        const bool okc = c < K;
        const int cc = okc ? c : 0;
        f32x4 t;
        if constexpr (A16) {
          const bf16s* q1 = reinterpret_cast<const bf16s*>(p.a1) + mc * p.lda1 + min(cc, p.C1 - 4);
          const bf16s* src = q1;
          if (two) {
            const bf16s* q2 = reinterpret_cast<const bf16s*>(p.a2) + mc * p.lda2 + max(cc - p.C1, 0);
            src = cc < p.C1 ? q1 : q2;
          }
          t = ld4(src);
        } else {
          const float* q1 = p.a1 + mc * p.lda1 + min(cc, p.C1 - 4);
          const float* src = q1;
          if (two) {  // (uniform branch; the select inside is per lane)
            const float* q2 = p.a2 + mc * p.lda2 + max(cc - p.C1, 0);
            src = cc < p.C1 ? q1 : q2;
          }
          t = *reinterpret_cast<const f32x4*>(src);
        }
        const bool ok = okc & (m < a.M);
        v[ps][i] = f32x4{ok ? t.x : 0.f, ok ? t.y : 0.f, ok ? t.z : 0.f, ok ? t.w : 0.f};
      }
    }
    stamp(1);
#pragma unroll
    for (int ps = 0; ps < PASSES; ++ps) {
      if (p.a_mode == 2) {  // channel LayerNorm: (x - mean) / sqrt(var + eps) * gamma, biased variance
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < IPL; ++i) s += (v[ps][i].x + v[ps][i].y) + (v[ps][i].z + v[ps][i].w);
        s = row_sum16(s);  // (the row's 16 lanes are one DPP row)
        const float mean = s / (float)K;
        float q = 0.f;
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
          const int c = (l16 + 16 * i) * 4;
          const float mk = c < K ? mean : 0.f;  // (padding channels hold 0 and stay 0)
          v[ps][i].x -= mk; v[ps][i].y -= mk; v[ps][i].z -= mk; v[ps][i].w -= mk;
          q += (v[ps][i].x * v[ps][i].x + v[ps][i].y * v[ps][i].y) + (v[ps][i].z * v[ps][i].z + v[ps][i].w * v[ps][i].w);
        }
        q = row_sum16(q);
        const float rstd = __builtin_amdgcn_rsqf(q / (float)K + a.eps);
        if (a.ln_stats && l16 == 0 && blockIdx.y == 0 && m0 + ps * 16 + r0 < a.M)
          *reinterpret_cast<float2*>(a.ln_stats + 2LL * (m0 + ps * 16 + r0)) = make_float2(mean, rstd);
#pragma unroll
        for (int i = 0; i < IPL; ++i) {
          const int c = (l16 + 16 * i) * 4;
          const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + (c < K ? c : 0));  // (padding channels: 0 * whatever)
          v[ps][i].x *= rstd * g.x; v[ps][i].y *= rstd * g.y; v[ps][i].z *= rstd * g.z; v[ps][i].w *= rstd * g.w;
        }
      }
      unsigned short* row = At + (ps * 16 + r0) * PITCH;
#pragma unroll
      for (int i = 0; i < IPL; ++i) {
        const int c = (l16 + 16 * i) * 4;
        if (c < KP) {
          if constexpr (F32) {  // exact-fp32 variant: the row is KP floats (the same bytes as hi | lo)
            *reinterpret_cast<f32x4*>(row + 2 * c) = v[ps][i];
          } else {
            unsigned l0, l1;
            const unsigned h0 = pj_split(v[ps][i].x, v[ps][i].y, l0);
            const unsigned h1 = pj_split(v[ps][i].z, v[ps][i].w, l1);
            *reinterpret_cast<uint2*>(row + c) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(row + KP + c) = make_uint2(l0, l1);
          }
        }
      }
    }
  }
  stamp(2);
  __syncthreads();
  stamp(3);

  // ---- sweep the output columns
  const uint4* wf = reinterpret_cast<const uint4*>(p.w);
  // Cout <= 64 under the 1 x 4 wave arrangement (K = 256: a 64-row tile is all the LDS holds) leaves one column slice: the four waves then
  // split the k16 steps of that slice instead of three of them idling, and wave 0 sums the partial accumulators through LDS.
  constexpr bool ksplit4 = KSPLIT4;
  // KSPLIT2: Cout <= 64 under the 2 x 2 arrangement (K = 128) leaves the two column waves of a row pair ONE column slice: they split its k16 steps
  // and then the epilogue (each takes one of the pair's two 32-row tiles) instead of one of them idling through products, residual transform and stores.
  const int wn = (ksplit4 || KSPLIT2) ? 0 : wn_id;
  auto load_b = [&](uint4 (&d)[4], int nc, int s) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      // column tiles past Cout are not in the packed operand: they re-read the last one that is (their accumulators are never stored);
      // no branch -- the step stays one basic block and its requests interleave with the MFMAs
      const int nt = min((nc * BN + wn * 64) / 32 + j, p.Cout / 32 - 1);
      const uint4* q = wf + ((long long)nt * KS + s) * 128 + lane;
      d[2 * j] = q[0];
      if constexpr (!ONE) d[2 * j + 1] = q[64];
    }
  };
  int abase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) abase[i] = (wm * 64 + i * 32 + lrow) * PITCH + lk * (F32 ? 16 : 8);
  auto load_a = [&](uint4 (&d)[4], int s) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned short* q = At + abase[i] + s * (F32 ? 32 : 16);
      d[2 * i] = *reinterpret_cast<const uint4*>(q);
      if constexpr (!ONE) d[2 * i + 1] = *reinterpret_cast<const uint4*>(q + (F32 ? 8 : KP));
    }
  };
  f32x16 acc[2][2];
  auto zero_acc = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  };
  auto mma_step = [&](const uint4 (&av)[4], const uint4 (&b)[4]) {
    if constexpr (F32) {  // exact fp32: eight K = 2 products per k16 step, e-major so consecutive MFMAs hit different accumulators
      const float pa[2][8] = {{__uint_as_float(av[0].x), __uint_as_float(av[0].y), __uint_as_float(av[0].z), __uint_as_float(av[0].w),
                               __uint_as_float(av[1].x), __uint_as_float(av[1].y), __uint_as_float(av[1].z), __uint_as_float(av[1].w)},
                              {__uint_as_float(av[2].x), __uint_as_float(av[2].y), __uint_as_float(av[2].z), __uint_as_float(av[2].w),
                               __uint_as_float(av[3].x), __uint_as_float(av[3].y), __uint_as_float(av[3].z), __uint_as_float(av[3].w)}};
      const float wb[2][8] = {{__uint_as_float(b[0].x), __uint_as_float(b[0].y), __uint_as_float(b[0].z), __uint_as_float(b[0].w),
                               __uint_as_float(b[1].x), __uint_as_float(b[1].y), __uint_as_float(b[1].z), __uint_as_float(b[1].w)},
                              {__uint_as_float(b[2].x), __uint_as_float(b[2].y), __uint_as_float(b[2].z), __uint_as_float(b[2].w),
                               __uint_as_float(b[3].x), __uint_as_float(b[3].y), __uint_as_float(b[3].z), __uint_as_float(b[3].w)}};
#pragma unroll
      for (int e = 0; e < 8; ++e)
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j][e], pa[i][e], acc[i][j], 0, 0, 0);
      return;
    }
    const bf16x8 ah0 = __builtin_bit_cast(bf16x8, av[0]), al0 = __builtin_bit_cast(bf16x8, av[1]);
    const bf16x8 ah1 = __builtin_bit_cast(bf16x8, av[2]), al1 = __builtin_bit_cast(bf16x8, av[3]);
    const bf16x8 bh0 = __builtin_bit_cast(bf16x8, b[0]), bl0 = __builtin_bit_cast(bf16x8, b[1]);
    const bf16x8 bh1 = __builtin_bit_cast(bf16x8, b[2]), bl1 = __builtin_bit_cast(bf16x8, b[3]);
    // weights = MFMA "A" (rows = output channels), rows of the tile = MFMA "B" (columns); pass-major order
#if VMM_PJ_DBG == 2   // measurement aid: operands loaded, no matrix work
    asm volatile("" :: "v"(bh0), "v"(bh1), "v"(bl0), "v"(bl1), "v"(ah0), "v"(ah1), "v"(al0), "v"(al1));
    return;
#endif
    if constexpr (ONE) {
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah0, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah1, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah1, acc[1][1], 0, 0, 0);
      return;
    }
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, al0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, al0, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, al1, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, al1, acc[1][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl0, ah0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl1, ah0, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl0, ah1, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bl1, ah1, acc[1][1], 0, 0, 0);
    acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah0, acc[0][0], 0, 0, 0);
    acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah0, acc[0][1], 0, 0, 0);
    acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh0, ah1, acc[1][0], 0, 0, 0);
    acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(bh1, ah1, acc[1][1], 0, 0, 0);
  };
  // Epilogue of one column chunk.  Everything a lane needs besides its accumulators is requested up front and consumed after a
  // single wait: the rotary factors depend only on (row, column mod 32) and are loaded once per workgroup, the bias / residual
  // pieces of the chunk are 16 independent 16-byte loads.  (The straightforward form -- load, wait, use, store per 16-byte piece,
  // with per-lane column tests -- serialised 16 memory latencies per chunk and left the kernel 3x short of its MFMA time.)
  // q_ncols, rot_ncols and Cout are multiples of 32 / 4, so the column tests are per 32-column tile and wave-uniform.
  const bool rotary = p.rot_ncols > 0;
  int mrow[2];
  f32x4 rot[2][4];  // [row tile i][g]: (cos, sin) of the two feature pairs at column offset 8 g + 4 lk of every 32-column head slice
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    mrow[i] = m0 + wm * 64 + i * 32 + lrow;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      rot[i][g] = f32x4{1.f, 0.f, 1.f, 0.f};
      if (rotary && mrow[i] < a.M) {
        // head width rot_dh (attn_dim_head of the temporal attentions): a multiple of 32 -- the first 32-column tile of every head rotates (the rotary
        // span is min(32, rot_dh), vddp.py:612), the others pass -- or a divisor of 32, where the head's pairs repeat inside the tile
        const int t = (mrow[i] / p.rot_HW) % p.rot_T, half = p.rot_dh >> 1;
        rot[i][g] = *reinterpret_cast<const f32x4*>(p.rot_tab + (t * half + ((4 * g + 2 * lk) & (min(half, 16) - 1))) * 2);
      }
    }
  }
  auto store_chunk = [&](int nc, int only_i = -1) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (only_i >= 0 && i != only_i) continue;  // (wave-uniform)
      const int m = mrow[i];
      if (m < a.M) {  // the only per-lane condition of the epilogue
        const float* resrow = p.res ? p.res + (long long)m * p.ldres : nullptr;
        float* outrow = p.out + (long long)m * p.ldo;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int tile0 = nc * BN + wn * 64 + j * 32;  // wave-uniform, as is everything tested on it (Cout, q_ncols, rot_ncols are multiples of 32)
          if (tile0 >= p.Cout) continue;
          const float scale = tile0 < p.q_ncols ? p.q_scale : 1.0f;
          const bool do_rot = tile0 < p.rot_ncols && (p.rot_dh <= 32 || tile0 % p.rot_dh == 0);
          const int c0 = tile0 + 4 * lk;
          f32x4 bv[4], rv[4];
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            bv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
            rv[g] = f32x4{0.f, 0.f, 0.f, 0.f};
          }
          if (p.bias) {
#pragma unroll
            for (int g = 0; g < 4; ++g) bv[g] = *reinterpret_cast<const f32x4*>(p.bias + c0 + 8 * g);
          }
          if (resrow) {
#pragma unroll
            for (int g = 0; g < 4; ++g)
              rv[g] = A16 ? ld4(reinterpret_cast<const bf16s*>(p.res) + (long long)m * p.ldres + c0 + 8 * g) : *reinterpret_cast<const f32x4*>(resrow + c0 + 8 * g);
            if (a.res_coef) {  // the ResnetBlock's main branch arrives pre-norm: GroupNorm * SiLU on the way in (vddp.py:311)
              const float* cf = a.res_coef + ((long long)(m / a.res_rps) * p.Cout + c0) * 2;
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                const f32x4 k0 = *reinterpret_cast<const f32x4*>(cf + 16 * g), k1 = *reinterpret_cast<const f32x4*>(cf + 16 * g + 4);
                rv[g].x = igemm::silu_fast(rv[g].x * k0.x + k0.y);
                rv[g].y = igemm::silu_fast(rv[g].y * k0.z + k0.w);
                rv[g].z = igemm::silu_fast(rv[g].z * k1.x + k1.y);
                rv[g].w = igemm::silu_fast(rv[g].w * k1.z + k1.w);
              }
            }
          }
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
            v.x = (v.x + bv[g].x) * scale; v.y = (v.y + bv[g].y) * scale; v.z = (v.z + bv[g].z) * scale; v.w = (v.w + bv[g].w) * scale;
            if (do_rot) {  // interleaved pairs (2i, 2i+1) of each head's 32 features, angle by the row's frame index (vddp.py:449,456)
              const f32x4 cs = rot[i][g], u = v;
              v.x = u.x * cs.x - u.y * cs.y; v.y = u.y * cs.x + u.x * cs.y;
              v.z = u.z * cs.z - u.w * cs.w; v.w = u.w * cs.z + u.z * cs.w;
            }
            v.x += rv[g].x; v.y += rv[g].y; v.z += rv[g].z; v.w += rv[g].w;
#if VMM_PJ_DBG == 1   // measurement aid (tools/build_ab.py proj_bf16x3 -DVMM_PJ_DBG=1): no output stores unless a value is NaN
            if (v.x != v.x) *reinterpret_cast<f32x4*>(outrow + c0 + 8 * g) = v;
#else
            if constexpr (A16) st4(reinterpret_cast<bf16s*>(p.out) + (long long)m * p.ldo + c0 + 8 * g, v);
            else *reinterpret_cast<f32x4*>(outrow + c0 + 8 * g) = v;
#endif
          }
        }
      }
    }
  };

  // software pipeline over the flattened (column chunk, k16 step) sequence: weights and tile fragments of the next step are
  // requested before the MFMAs of the current one; KS is even, so the two register buffers alternate cleanly across chunks
  uint4 bb[2][4], aa[2][4];
  if constexpr (ksplit4) {
    if constexpr (WN == 4 && KS % 4 == 0) {
      const int s_begin = wave * (KS / 4);
      zero_acc();
      load_b(bb[0], nc_begin, s_begin);
      load_a(aa[0], s_begin);
#pragma unroll
      for (int q = 0; q < KS / 4; ++q) {
        if (q + 1 < KS / 4) {
          load_b(bb[(q + 1) & 1], nc_begin, s_begin + q + 1);
          load_a(aa[(q + 1) & 1], s_begin + q + 1);
        }
        __builtin_amdgcn_sched_barrier(0);
        mma_step(aa[q & 1], bb[q & 1]);
        __builtin_amdgcn_sched_barrier(0);
      }
      // partial accumulators of waves 1..3 -> LDS (the row tile is no longer needed), summed by wave 0, which runs the epilogue
      __syncthreads();
      float* red = reinterpret_cast<float*>(At);
      if (wave > 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((wave - 1) * 4 + i * 2 + j) * 16 + r) * 64 + lane] = acc[i][j][r];
      }
      __syncthreads();
      if (wave == 0) {
#pragma unroll
        for (int w = 0; w < 3; ++w)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[i][j][r] += red[((w * 4 + i * 2 + j) * 16 + r) * 64 + lane];
        store_chunk(nc_begin);
      }
    }
    return;
  }
  if constexpr (KSPLIT2) {
    const int s_begin = wn_id * (KS / 2);
    zero_acc();
    load_b(bb[0], nc_begin, s_begin);
    load_a(aa[0], s_begin);
#pragma unroll
    for (int q = 0; q < KS / 2; ++q) {
      if (q + 1 < KS / 2) {
        load_b(bb[(q + 1) & 1], nc_begin, s_begin + q + 1);
        load_a(aa[(q + 1) & 1], s_begin + q + 1);
      }
      __builtin_amdgcn_sched_barrier(0);
      mma_step(aa[q & 1], bb[q & 1]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // every wave hands the row tile its partner owns (tile i belongs to column wave i) over through LDS -- the row tile is no longer needed
    __syncthreads();
    float* red = reinterpret_cast<float*>(At);
    const int give = wn_id ^ 1, partner = wm * WN + (wn_id ^ 1);
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) red[((wave * 2 + j) * 16 + r) * 64 + lane] = give ? acc[1][j][r] : acc[0][j][r];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float o = red[((partner * 2 + j) * 16 + r) * 64 + lane];
        if (wn_id) acc[1][j][r] += o; else acc[0][j][r] += o;
      }
    store_chunk(nc_begin, wn_id);
    return;
  }
  load_b(bb[0], nc_begin, 0);
  load_a(aa[0], 0);
  for (int nc = nc_begin; nc < nc_end; ++nc) {
    const bool more = nc + 1 < nc_end;
#pragma unroll
    for (int i = 0; i < 2; ++i) asm volatile("" : "+v"(abase[i]));  // keep tile addresses base + immediate
    zero_acc();
#pragma unroll
    for (int s = 0; s < KS; ++s) {
      if (s + 1 < KS) {
        load_b(bb[(s + 1) & 1], nc, s + 1);
        load_a(aa[(s + 1) & 1], s + 1);
      } else {
        load_b(bb[0], more ? nc + 1 : nc, 0);
        load_a(aa[0], 0);
      }
      mma_step(aa[s & 1], bb[s & 1]);
      {  // the next step's requests between this step's MFMAs (conv3x3_bf16x3.hip, same reasoning)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    stamp(4 + 2 * (nc - nc_begin));
    store_chunk(nc);
    stamp(5 + 2 * (nc - nc_begin));
  }
  stamp(16);
}

inline int* pj_launch_counter() { static int n = 0; return &n; }

template <int WM, int WN, int KS, bool F32, bool ONE>
int launch_pj_k2(const PJArgs& a, hipStream_t s) {  // the KSPLIT2 instance: one column slice, one workgroup row per row tile
  constexpr int BM = WM * 64;
  const size_t shm = sizeof(unsigned short) * (size_t)BM * (2 * KS * 16 + 8);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_x3_kernel<WM, WN, KS, F32, false, ONE, false, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((proj_x3_kernel<WM, WN, KS, F32, false, ONE, false, true>), dim3((unsigned)cdiv(a.M, BM), 1), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

template <int WM, int WN, int KS, bool F32, bool KSPLIT4 = false, bool ONE = false>
int launch_pj(const PJArgs& a, hipStream_t s) {
  constexpr int BM = WM * 64;
  const size_t shm = sizeof(unsigned short) * (size_t)BM * (2 * KS * 16 + 8);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_x3_kernel<WM, WN, KS, F32, KSPLIT4, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int mt = (int)cdiv(a.M, BM);
  const int ny = (int)cdiv(a.n_chunks, a.chunks_per_y);
  static const int trace_launch = [] { const char* e = getenv("VMM_PJ_TRACE"); return e ? atoi(e) : -1; }();
  static int* launch_no = pj_launch_counter();
  if (trace_launch >= 0 && (*launch_no)++ == trace_launch) {  // measurement aid: dump the workgroups' phase stamps (tools/trace_proj.py)
    PJArgs at = a;
    const size_t n = (size_t)mt * ny * 18;
    (void)hipMalloc(&at.trace, n * sizeof(unsigned long long));
    (void)hipMemsetAsync(at.trace, 0, n * sizeof(unsigned long long), s);
    hipLaunchKernelGGL((proj_x3_kernel<WM, WN, KS, F32, KSPLIT4, ONE>), dim3((unsigned)mt, (unsigned)ny), dim3(256), shm, s, at);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), at.trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(at.trace);
    const char* fn = getenv("VMM_PJ_TRACE_FILE");
    FILE* f = fopen(fn ? fn : "pj_trace.txt", "w");
    if (f) {
      fprintf(f, "# K %d Cout %d M %d WM %d WN %d KS %d chunks %d per_y %d a_mode %d res %d rot %d\n", a.p.C1 + a.p.C2, a.p.Cout, a.M, WM, WN, KS, a.n_chunks, a.chunks_per_y,
              a.p.a_mode, a.p.res != nullptr, a.p.rot_ncols);
      for (size_t w = 0; w < (size_t)mt * ny; ++w) {
        for (int k = 0; k < 18; ++k) fprintf(f, "%llu ", h[w * 18 + k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    return 0;
  }
  hipLaunchKernelGGL((proj_x3_kernel<WM, WN, KS, F32, KSPLIT4, ONE>), dim3((unsigned)mt, (unsigned)ny), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

template <int WM, int WN, int KS>
int launch_pj_a16(const PJArgs& a, hipStream_t s) {
  constexpr int BM = WM * 64;
  const size_t shm = sizeof(unsigned short) * (size_t)BM * (2 * KS * 16 + 8);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&proj_x3_kernel<WM, WN, KS, false, false, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((proj_x3_kernel<WM, WN, KS, false, false, true, true>), dim3((unsigned)cdiv(a.M, BM), (unsigned)cdiv(a.n_chunks, a.chunks_per_y)), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// 1x1 / Linear projection.  d->w = vmm_pack_weights fmt 2 of the (Cout, K) weight.  ln_gamma != NULL: the rows pass through the
// channel LayerNorm (gamma only, eps inside the sqrt, vddp.py:245-254) while they are staged.  Envelope: KH = KW = 1, stride 1,
// identity row mapping, K = C1 + C2 <= 256 with C1, C2 multiples of 4, Cout a multiple of 32; returns 1 (nothing launched) otherwise.
template <bool F32, bool ONE = false>
static int run_proj(const vmm_conv_desc& d, const float* ln_gamma, float ln_eps, vmm_stream_t stream, const float* res_coef = nullptr, int res_rps = 1,
                    float* ln_stats = nullptr) {
  const bool shape_ok = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.off_h == 0 && d.off_w == 0 && d.Hv == d.Hin && d.Wv == d.Win &&
                        d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && d.a_mode == 0;
  const int K = d.C1 + d.C2;
  const bool chan_ok = (d.C1 & 3) == 0 && (d.C2 & 3) == 0 && d.C1 > 0 && K <= 256 && (d.Cout & 31) == 0 && (d.lda1 & 3) == 0 &&
                       (!d.C2 || (d.lda2 & 3) == 0) && (d.ldo & 3) == 0 && (!d.res || (d.ldres & 3) == 0);
  if (!shape_ok || !chan_ok) return 1;
  if (d.rot_ncols > 0 && (!d.rot_tab || !(d.rot_dh >= 4 && (d.rot_dh % 32 == 0 || 32 % d.rot_dh == 0)) || (d.rot_ncols & 31) || d.rot_HW <= 0 || d.rot_T <= 0))
    return -2;
  if (d.q_ncols & 31) return -2;
  const long long M = (long long)d.nimg * d.Hv * d.Wv;
  if (M >= (1LL << 31)) return -4;
  if (M <= 0 || d.Cout <= 0) return 0;
  PJArgs a;
  a.p = d;
  a.p.a_mode = ln_gamma ? 2 : 0;
  a.gamma = ln_gamma;
  a.eps = ln_eps;
  a.res_coef = res_coef;
  a.res_rps = res_rps;
  a.ln_stats = ln_gamma ? ln_stats : nullptr;
  a.M = (int)M;
  const int KP = (K + 31) / 32 * 32;
  // workgroup shape: as many rows as fit 80 KB of LDS; column chunk = 256 / rows * 64
  const int wm = KP <= 64 ? 4 : KP <= 128 ? 2 : 1, wn = 4 / wm;
  a.n_chunks = (int)cdiv(d.Cout, wn * 64);
  // few rows: spread the column chunks over blockIdx.y until ~1000 workgroups exist (the row tile is then staged once per y)
  const long long mt = cdiv(M, wm * 64);
  int ny = (int)max(1LL, min((long long)a.n_chunks, 1024 / max(mt, 1LL)));
  a.chunks_per_y = (int)cdiv(a.n_chunks, ny);
  hipStream_t s = (hipStream_t)stream;
  if (d.act_bf16) {  // bf16-stored rows, residual and output (all three; the single-pass mode only)
    if (!ONE || d.act_bf16 != 3) return -1;
    if (KP == 32) return launch_pj_a16<4, 1, 2>(a, s);
    if (KP == 64) return launch_pj_a16<4, 1, 4>(a, s);
    if (KP == 128) return launch_pj_a16<2, 2, 8>(a, s);
    if (KP == 256 && d.Cout > 64) return launch_pj_a16<1, 4, 16>(a, s);
    return 1;
  }
  if (KP == 32) return launch_pj<4, 1, 2, F32, false, ONE>(a, s);
  if (KP == 64) return launch_pj<4, 1, 4, F32, false, ONE>(a, s);
  if (KP <= 128) {
    if (KP == 96) return 1;
    static const bool k2 = [] { const char* e = getenv("VMM_PJ_KSPLIT2"); return !e || e[0] != '0'; }();  // (0: A/B runs)
    if (KP == 128 && d.Cout <= 64 && a.n_chunks == 1 && k2) return launch_pj_k2<2, 2, 8, F32, ONE>(a, s);
    return launch_pj<2, 2, 8, F32, false, ONE>(a, s);
  }
  if (KP == 256) return d.Cout <= 64 ? launch_pj<1, 4, 16, F32, true, ONE>(a, s) : launch_pj<1, 4, 16, F32, false, ONE>(a, s);
  return 1;
}

extern "C" int vmm_proj_bf16x3(const vmm_conv_desc* dp, const float* ln_gamma, float ln_eps, vmm_stream_t stream) {
  return run_proj<false>(*dp, ln_gamma, ln_eps, stream);
}

// The same with the LayerNorm statistics of every row left in ln_stats [rows][2] = (mean, 1 / sqrt(var + eps)): the training forward of to_qkv --
// the normalised rows are never materialised, the weight gradient (vmm_conv1x1_wgrad_bf16x3_ln) re-normalises x from these while it stages it.
extern "C" int vmm_proj_bf16x3_ln_stats(const vmm_conv_desc* dp, const float* ln_gamma, float ln_eps, float* ln_stats, vmm_stream_t stream) {
  if (!ln_gamma || !ln_stats) return -1;
  return run_proj<false>(*dp, ln_gamma, ln_eps, stream, nullptr, 1, ln_stats);
}

// ResnetBlock tail (vddp.py:311): out = silu(res * a + b') + proj(x) in one launch -- res = the pre-norm output of block2's convolution
// (d->res, may be d->out: in place), (a, b') = vmm_groupnorm_coef's [B][Cout][2], proj = res_conv.  Replaces the res_conv launch, its
// output buffer and the vmm_affine_silu pass.  Same envelope as vmm_proj_bf16x3; d->res must be set.
extern "C" int vmm_proj_bf16x3_res_silu(const vmm_conv_desc* dp, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream) {
  if (!dp->res || !res_coef || rows_per_sample <= 0) return -1;
  return run_proj<false>(*dp, nullptr, 0.f, stream, res_coef, rows_per_sample);
}

// The same kernel on the exact-fp32 matrix-core instruction (d->w = vmm_pack_weights fmt 4); the "fp32" arithmetic mode's projections.
extern "C" int vmm_proj_f32(const vmm_conv_desc* dp, const float* ln_gamma, float ln_eps, vmm_stream_t stream) {
  return run_proj<true>(*dp, ln_gamma, ln_eps, stream);
}

// The "bf16" throughput mode of the projection kernel (BASELINE.json configs[3]): same descriptor, same fmt-2 weights, one matrix pass on the
// operands' bf16 roundings; the fused LayerNorm, the q-scale / rotary / bias / residual epilogue stay fp32.
extern "C" int vmm_proj_bf16(const vmm_conv_desc* dp, const float* ln_gamma, float ln_eps, vmm_stream_t stream) {
  return run_proj<false, true>(*dp, ln_gamma, ln_eps, stream);
}
extern "C" int vmm_proj_bf16_res_silu(const vmm_conv_desc* dp, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream) {
  if (!dp->res || !res_coef || rows_per_sample <= 0) return -1;
  return run_proj<false, true>(*dp, nullptr, 0.f, stream, res_coef, rows_per_sample);
}
