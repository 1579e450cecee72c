// Weight-gradient GEMM on the bf16 matrix cores with split (hi + lo) operands -- the "bf16x3" training mode's counterpart of
// vmm_conv_wgrad_f32 (wgrad.hip), gfx950.
//
//   dWp[(tap, ci)][co] += sum_m A[(img, a*stride + dh(tap), b*stride + dw(tap)), ci] * dY[orow(m), co]
//
// The contraction runs over the rows m, so both MFMA operands need eight consecutive rows of ONE column per lane.  The loader
// transposes in registers: thread = (4 consecutive columns, 8 consecutive rows) reads eight 16-byte row pieces (coalesced: 32 lanes
// cover 512 contiguous bytes of a row), splits the 4 x 8 values into bf16 hi / lo and writes one 16-byte fragment per column and
// plane, already in the order the MFMAs read.  Tile column 4 g + q lives at LDS position 32 q + g, so both the fragment writes and
// the operand reads are conflict-free; the accumulators' column <-> lane map is undone in the epilogue.  Waves 0-1 load A, waves 2-3 dY.
// (A first version with one 4-byte load per element and column spent 3x the fp32 kernel's time on address arithmetic.)
// Workgroup tile 128 x 128 (2 x 2 waves of 64 x 64 = four 32x32 accumulators), 32 rows per chunk, double-buffered LDS, the next
// chunk's loads in flight across the current chunk's 24 MFMAs.  Three passes per product (hi*lo + lo*hi + hi*hi, fp32 accumulate):
// ~1.5e-5 relative error per product, i.e. well inside the TF32 convolutions the reference trains with on its own hardware.
// STATUS: opt-in (Unet3D.use_x3_wgrad = True).  On the Lagrangian training step (MI355X, batch 4) it is SLOWER than the fp32 kernel:
// 31 ms vs 18.6 ms per step over the 100 weight gradients.  Ablation (ms per step removed): no global loads -13.5, no atomics -6.3,
// no MFMAs -2.0, remaining skeleton (split, LDS, barriers) 8.7: with 64 KB of LDS and 158 registers only two workgroups fit a CU and
// the gathered loads are not covered.  Kept as the starting point for a deeper-pipelined version.
// Row slices (blockIdx.z) combine with fp32 atomics, as in the fp32 kernel; the bias gradient (column sums of dY) leaves as one
// partial row per slice.
#include <stdlib.h>
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int TI = 128, TJ = 128, XK = 32;

// two floats -> packed bf16 hi pair and packed bf16 lo pair (lo = x - float(hi), both round-to-nearest-even)
__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) {
  const __bf16 h0 = (__bf16)x0, h1 = (__bf16)x1;
  const __bf16 l0 = (__bf16)(x0 - (float)h0), l1 = (__bf16)(x1 - (float)h1);
  hi = (unsigned)__builtin_bit_cast(unsigned short, h0) | ((unsigned)__builtin_bit_cast(unsigned short, h1) << 16);
  lo = (unsigned)__builtin_bit_cast(unsigned short, l0) | ((unsigned)__builtin_bit_cast(unsigned short, l1) << 16);
}

__global__ __launch_bounds__(256) void wgrad_x3_kernel(const vmm_conv_desc p, const float* __restrict__ dy, int lddy, float* __restrict__ dw,
                                                       long long rows_per_split, float* __restrict__ bias_part) {
  // [buffer][operand: A hi, A lo, B hi, B lo][step][k half][position] x 8 bf16 (16 bytes); position q*32 + g holds tile column 4 g + q
  __shared__ uint4 frag[2][4][2][2][128];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wi = wave >> 1, wj = wave & 1;
  const int Cin = p.C1 + p.C2;
  const int Ktot = p.KH * p.KW * Cin;
  const int i0 = blockIdx.x * TI, j0 = blockIdx.y * TJ;
  const long long M = (long long)p.nimg * p.Hv * p.Wv;
  const long long m_begin = (long long)blockIdx.z * rows_per_split;
  const long long m_end = min(m_begin + rows_per_split, M);
  if (m_begin >= m_end) return;
  const int nk = (int)((m_end - m_begin + XK - 1) / XK);

  // loader roles: waves 0-1 the A tile, waves 2-3 the dY tile; thread = (4 consecutive columns cg, 8 consecutive rows rg) of a chunk
  const bool a_role = tid < 128;
  const int cg = tid & 31, rg = (tid >> 5) & 3;
  const int col = (a_role ? i0 : j0) + cg * 4;
  const bool cvalid = col < (a_role ? Ktot : p.Cout);
  const int tap = (a_role && cvalid) ? col / Cin : 0;
  const int ci = col - tap * Cin;
  const int kh = tap / p.KW, kw = tap - kh * p.KW;
  const int dh = p.off_h + p.sgn_h * kh, dwo = p.off_w + p.sgn_w * kw;
  const bool from_a1 = ci < p.C1;
  const float* acol = from_a1 ? p.a1 + ci : p.a2 + (ci - p.C1);
  const int lda = from_a1 ? p.lda1 : p.lda2;
  const bool fused = a_role && from_a1 && p.a_mode == 1;  // GroupNorm + SiLU of the producer applied on the fly (per-sample scale / shift)
  const bool identity_rows = (p.oscale == 1 && p.Hout == p.Hv && p.Wout == p.Wv && p.ooh == 0 && p.oow == 0);
  const int hw = p.Hv * p.Wv;

  f32x4 rv[8];  // rows rg*8 .. +7 of the next chunk, this thread's four columns
  f32x4 bsum = {0.f, 0.f, 0.f, 0.f};
  // (img, a, b) of this thread's first row of the next chunk
  long long m = m_begin + rg * 8;
  int img = (int)(m / hw);
  int a = (int)(m - (long long)img * hw) / p.Wv, b = (int)(m - (long long)img * hw) - a * p.Wv;
  auto advance = [&](int n) {
    m += n;
    b += n;
    while (b >= p.Wv) {
      b -= p.Wv;
      if (++a == p.Hv) { a = 0; ++img; }
    }
  };
  // The fused GroupNorm + SiLU is applied when the chunk is consumed (store_chunk), never next to its load -- that would park the wave
  // on every single load.  A thread's eight rows touch at most two samples: their coefficients are fetched with the chunk, vmask /
  // smask remember per row whether it was read at all (conv zero padding stays zero) and which of the two samples it belongs to.
  f32x4 cf0[2], cf1[2];
  unsigned vmask = 0, smask = 0;
  auto load_chunk = [&]() {
    vmask = 0;
    smask = 0;
    int sample0 = 0;
    if (fused) {
      const int nsamp = p.nimg / p.a_imgs_per_sample;
      sample0 = min(img / p.a_imgs_per_sample, nsamp - 1);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const float* cf = p.a_coef + ((long long)min(sample0 + k, nsamp - 1) * p.C1 + ci) * 2;
        cf0[k] = *reinterpret_cast<const f32x4*>(cf);
        cf1[k] = *reinterpret_cast<const f32x4*>(cf + 4);
      }
    }
#pragma unroll
    for (int r = 0; r < 8; ++r) {
      f32x4 x = {0.f, 0.f, 0.f, 0.f};
      if (m < m_end && cvalid) {
        if (a_role) {
          int ih = a * p.stride + dh, iw = b * p.stride + dwo;
          if (p.wrap_h) ih = ih < 0 ? ih + p.Hin : (ih >= p.Hin ? ih - p.Hin : ih);  // periodic padding (vddp.py:163-243)
          if (p.wrap_w) iw = iw < 0 ? iw + p.Win : (iw >= p.Win ? iw - p.Win : iw);
          if ((unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
            x = *reinterpret_cast<const f32x4*>(acol + (((long long)img * p.Hin + ih) * p.Win + iw) * lda);
            vmask |= 1u << r;
            if (fused && img / p.a_imgs_per_sample != sample0) smask |= 1u << r;
          }
        } else {
          long long orow = m;
          if (!identity_rows) orow = ((long long)img * p.Hout + a * p.oscale + p.ooh) * p.Wout + b * p.oscale + p.oow;
          x = *reinterpret_cast<const f32x4*>(dy + orow * lddy + col);
        }
      }
      rv[r] = x;
      advance(1);
    }
    advance(XK - 8);
  };
  // register transpose: the eight rows of each of the four columns -> one 16-byte hi and one 16-byte lo fragment
  auto store_chunk = [&](int buf) {
    const int op = a_role ? 0 : 2, st = rg >> 1, kg = rg & 1;
    if (fused) {
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        if (vmask >> r & 1) {
          const int k = smask >> r & 1;
          const f32x4 c0 = k ? cf0[1] : cf0[0], c1 = k ? cf1[1] : cf1[0];
          rv[r].x = silu_rcp(rv[r].x * c0.x + c0.y);
          rv[r].y = silu_rcp(rv[r].y * c0.z + c0.w);
          rv[r].z = silu_rcp(rv[r].z * c1.x + c1.y);
          rv[r].w = silu_rcp(rv[r].w * c1.z + c1.w);
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      uint4 h, l;
      split_pair(rv[0][q], rv[1][q], h.x, l.x);
      split_pair(rv[2][q], rv[3][q], h.y, l.y);
      split_pair(rv[4][q], rv[5][q], h.z, l.z);
      split_pair(rv[6][q], rv[7][q], h.w, l.w);
      frag[buf][op][st][kg][q * 32 + cg] = h;
      frag[buf][op + 1][st][kg][q * 32 + cg] = l;
    }
    if (!a_role) {
#pragma unroll
      for (int r = 0; r < 8; ++r) bsum += rv[r];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
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
    for (int s = 0; s < 2; ++s) {
      bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        ah[u] = __builtin_bit_cast(bf16x8, frag[buf][0][s][lk][(wi * 2 + u) * 32 + l31]);
        al[u] = __builtin_bit_cast(bf16x8, frag[buf][1][s][lk][(wi * 2 + u) * 32 + l31]);
        bh[u] = __builtin_bit_cast(bf16x8, frag[buf][2][s][lk][(wj * 2 + u) * 32 + l31]);
        bl[u] = __builtin_bit_cast(bf16x8, frag[buf][3][s][lk][(wj * 2 + u) * 32 + l31]);
      }
      // pass-major: consecutive MFMAs write different accumulators
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bl[v], acc[u][v], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u], bh[v], acc[u][v], 0, 0, 0);
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int v = 0; v < 2; ++v) acc[u][v] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u], bh[v], acc[u][v], 0, 0, 0);
    }
    if (kc + 1 < nk) store_chunk(buf ^ 1);
    __syncthreads();
  }
  // accumulator (u, v): rows = tile columns 4 idx + (wi*2 + u) of A, lanes = tile columns 4 l31 + (wj*2 + v) of dY (position -> column map)
#pragma unroll
  for (int u = 0; u < 2; ++u)
#pragma unroll
    for (int v = 0; v < 2; ++v)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int i = i0 + 4 * ((r & 3) + 8 * (r >> 2) + 4 * lk) + (wi * 2 + u);
        const int j = j0 + 4 * l31 + (wj * 2 + v);
        if (i < Ktot && j < p.Cout) atomicAdd(&dw[(long long)i * p.Cout + j], acc[u][v][r]);
      }
  if (bias_part && blockIdx.x == 0) {  // (workgroup-uniform) column sums of this row slice of dY -> bias_part[slice][co]
    f32x4* red = reinterpret_cast<f32x4*>(&frag[0][0][0][0][0]);  // the last chunk's barrier is behind us
    if (!a_role) red[rg * 32 + cg] = bsum;
    __syncthreads();
    if (!a_role && rg == 0 && cvalid) {
      const f32x4 t = (red[cg] + red[32 + cg]) + (red[64 + cg] + red[96 + cg]);
      *reinterpret_cast<f32x4*>(bias_part + (long long)blockIdx.z * p.Cout + col) = t;
    }
  }
}

}  // namespace

// Same contract as vmm_conv_wgrad_f32 (wgrad.hip), products on the bf16 matrix cores with split operands.
extern "C" int vmm_conv_wgrad_bf16x3(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                                     float* bias_scratch, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  if (nsplit < 1 || (dbias && !bias_scratch) || (d.C1 & 3) || (d.C2 & 3) || (d.lda1 & 3) || (d.C2 && (d.lda2 & 3)) || (d.Cout & 3) || (lddy & 3)) return -1;
  {  // the 3 x 3 "same" convolutions: nine taps per workgroup, x staged once (wgrad3x3_bf16x3.hip); VMM_WGRAD3X3=0 keeps this kernel (A/B runs)
    static const bool use3 = [] { const char* e = getenv("VMM_WGRAD3X3"); return !e || e[0] != '0'; }();
    if (use3) {
      const int rc = vmm_conv3x3_wgrad_bf16x3(dp, dy, lddy, dw_packed, dbias, nullptr, stream);
      if (rc != 1) return rc;
    }
  }
  const long long M = (long long)d.nimg * d.Hv * d.Wv;
  if (M <= 0) return 0;
  const int Ktot = d.KH * d.KW * (d.C1 + d.C2);
  const long long rps = (cdiv(M, nsplit) + XK - 1) / XK * XK;
  dim3 grid(cdiv(Ktot, TI), cdiv(d.Cout, TJ), cdiv(M, rps));
  hipLaunchKernelGGL(wgrad_x3_kernel, grid, dim3(256), 0, (hipStream_t)stream, d, dy, lddy, dw_packed, rps, dbias ? bias_scratch : nullptr);
  VMM_LAUNCH_CHECK();
  if (dbias) return vmm_sum_partials(bias_scratch, (int)grid.z, d.Cout, d.Cout, dbias, stream);
  return 0;
}
