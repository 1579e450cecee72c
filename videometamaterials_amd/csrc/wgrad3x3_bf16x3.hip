// Weight gradient of the 3 x 3 "same" convolutions (ResnetBlock projections, vddp.py:268-285) on the bf16 matrix cores with split
// (hi + lo) operands, gfx950 -- the "bf16x3" training mode's counterpart of wgrad3x3.hip.
//
//   dWp[(kh, kw, ci)][co] += sum over pixels q   x[q + (kh-1) W + (kw-1)][ci] * dY[q][co]
//
// The contraction runs over PIXELS, so both operands of v_mfma_f32_32x32x16_bf16 need, per lane, eight consecutive pixels of one channel,
// while memory holds pixels x channels.  What this kernel is built around:
//
//  * Position space.  The frames are laid end to end as one line of positions P = (img (H + 1) + y) Wp + x with one zero row after every
//    frame (Wp = W when W is a multiple of 8, else W rounded up past a zero column): a tap is a constant position offset (kh-1) Wp + (kw-1),
//    vertical zero padding is the zero row, and a workgroup's work is a contiguous range of positions whatever the frame size.
//  * x is staged ONCE.  A workgroup owns all nine taps of a (64 input channels) x (64 output channels) block and walks its range in chunks
//    of 64 positions; x lives in an LDS ring [channel][position] (bf16 hi and lo planes) that always holds the current chunk plus one row
//    above and below, every element is loaded from L2, split and stored exactly once per workgroup (wgrad3x3.hip re-stages a 3.25x halo).
//  * The transposition is free.  A loader thread reads eight consecutive positions of two channels (eight 8-byte loads, 256 contiguous
//    bytes per position across 32 lanes), splits them and writes one 16-byte hi and lo fragment per channel: exactly the eight-pixel
//    operand piece a lane of the MFMA reads back with one ds_read_b128 (row pitch = odd multiple of 16 bytes: conflict-free both ways).
//  * The horizontal taps shift dY, not x:  sum_q x[q + (kh-1)Wp + (kw-1)] dY[q] = sum_q' x[q' + (kh-1)Wp] dY[q' - (kw-1)].  The x operand
//    of every tap is then an ALIGNED fragment at a row offset, and the three shifted versions of the one dY fragment a k-step needs are
//    made in registers: five v_alignbit per plane, the element shifted in from the neighbouring piece comes from a small side array the
//    loader fills (zero where the neighbour is across a frame border, which is the horizontal zero padding).
//  * 8 waves = 2 x (2 x 2 quadrants of the 64 x 64 block, nine 32 x 32 accumulators = 144 registers each).  The two wave groups take
//    alternate halves of every chunk and run half an iteration apart: group 0 splits / stores the next chunk's rows BEFORE its MFMAs,
//    group 1 AFTER, so on every SIMD one wave's loader phase sits under the other wave's matrix phase; one barrier per chunk.
//
// Row slices (blockIdx.z) combine with fp32 atomics like in the other weight-gradient kernels; the bias gradient (exact fp32 column sums of dY)
// is added by the ci-block-0 workgroups.
#include <stdlib.h>
#include "vmm_common.h"
#include "wgrad_reduce.h"
#include "../../include/vmm_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 64;                 // positions per chunk (four k16 steps, two per wave group)
constexpr int DPITCH = 2 * CH + 16;    // bytes per channel row of a dY plane (144 = 9 x 16)
constexpr int NPITCH = CH + 8;         // bytes per channel row of the neighbour array (8 bytes per eight-position piece)
constexpr int DY_LO = 64 * DPITCH, DY_NB = 2 * 64 * DPITCH, DY_BUF = DY_NB + 64 * NPITCH;
constexpr int PART_FLOATS = 9 * 64 * 64;  // one workgroup's partial block

struct W9Args {
  vmm_conv_desc p;
  const float* dy; int lddy;
  float* dw;
  float* dbias;                 // column sums of dY are ADDED here (or NULL)
  float* part;                  // workspace [gridDim.z][tiles][PART_FLOATS] (+ bias rows) or NULL: then atomics straight into dw / dbias
  float* bias_part;             // [gridDim.z][Cout] inside the workspace
  int Wp, rows_per_img;         // position pitch of an image row; H + 1
  unsigned wp_magic, rpi_magic; // floor(2^32 / d) + 1: n / d = mulhi(n, magic) for n d < 2^32 (checked by the launcher)
  int n_pos;                    // nimg * (H + 1) * Wp
  int nchunks, chunks_per_wg;
  int J0, R;                    // x look-ahead in chunk loads = ceil(2 Wp / CH); ring size in positions = (J0 + 2) * CH
};

__device__ __forceinline__ bf16x8 as_frag(unsigned a, unsigned b, unsigned c, unsigned d) {
  const uint4 v = {a, b, c, d};
  return __builtin_bit_cast(bf16x8, v);
}

__global__ __launch_bounds__(512) void wgrad9_x3_kernel(const W9Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const vmm_conv_desc& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3, wu = wq >> 1, wv = wq & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int R = a.R, RP = 2 * R + 16;          // ring row pitch in bytes: (R / 8 + 1) x 16, R / 8 even
  const int RING_LO = 64 * RP, DYB = 128 * RP;
  const int H = p.Hin, W = p.Win, Wp = a.Wp;
  const int Cin = p.C1 + p.C2;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const int c_begin = blockIdx.z * a.chunks_per_wg;
  const int n_it = min(a.chunks_per_wg, a.nchunks - c_begin);
  if (n_it <= 0) return;
  const int p0 = c_begin * CH;

  // ---------------------------------------------------------------- loader role: waves 0-3 x, waves 4-7 dY; thread = (piece, channel pair)
  const bool is_dy = wave >= 4;
  const int oct = (tid >> 5) & 7, cp = tid & 31;
  const bool src1 = ci0 < p.C1;
  const float* lsrc = is_dy ? a.dy + co0 + 2 * cp : (src1 ? p.a1 + ci0 : p.a2 + (ci0 - p.C1)) + 2 * cp;
  const int lld = is_dy ? a.lddy : (src1 ? p.lda1 : p.lda2);
  f32x2 rv[8], nv[2];
  unsigned lmask = 0;
  f32x2 bsum = {0.f, 0.f};
  // a_mode 1: x = silu(a1 * ga + gb), the GroupNorm * FiLM -> SiLU of the producing block (vddp.py:279-285) applied while the fragments are built --
  // every x element passes here once per (64 output channels), so the activated operand is never materialised.  (ga, gb) per (sample, channel):
  // an eight-position piece lies inside one frame, hence one sample: one 16-byte coefficient load per item.
  const bool fused = p.a_mode == 1 && src1;
  f32x4 cf = {1.f, 0.f, 1.f, 0.f};
  auto coef_of = [&](unsigned img) -> const f32x4* {
    return reinterpret_cast<const f32x4*>(p.a_coef + ((long long)(img / (unsigned)p.a_imgs_per_sample) * p.C1 + ci0 + 2 * cp) * 2);
  };

  // request the eight positions P0 .. P0 + 7 (dY: and the neighbours P0 - 1, P0 + 8) of this thread's channel pair; every load is
  // unconditional (a safe address where the position is padding): a lane-dependent branch around a load costs a vmcnt(0) at its first use
  auto request = [&](int P0) {
    bool ok = P0 >= 0 && P0 < a.n_pos;
    const unsigned Pu = ok ? (unsigned)P0 : 0u;
    const unsigned row = __umulhi(Pu, a.wp_magic);
    const int col0 = (int)(Pu - row * (unsigned)Wp);
    const unsigned img = __umulhi(row, a.rpi_magic);
    const int y = (int)(row - img * (unsigned)a.rows_per_img);
    ok = ok && y < H;
    const float* base = lsrc + ((long long)((int)img * H + y) * W + col0) * lld;
    unsigned m = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool v = ok && col0 + i < W;
      m |= (v ? 1u : 0u) << i;
      rv[i] = *reinterpret_cast<const f32x2*>(v ? base + (long long)i * lld : lsrc);
    }
    if (fused && !is_dy) cf = *coef_of(ok ? img : 0u);
    if (is_dy) {
      const bool vp = ok && col0 > 0 && col0 - 1 < W, vn = ok && col0 + 8 < W;
      m |= (vp ? 1u : 0u) << 8 | (vn ? 1u : 0u) << 9;
      nv[0] = *reinterpret_cast<const f32x2*>(vp ? base - lld : lsrc);
      nv[1] = *reinterpret_cast<const f32x2*>(vn ? base + 8LL * lld : lsrc);
    }
    lmask = m;
  };
  // split into bf16 hi | lo and store the fragments (x: ring slot `slot`; dY: chunk buffer `buf`)
  auto stage = [&](int slot, int buf) {
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = k;  // (the two-way bank conflict of these stores is covered by their 13-cycle data transfer)
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = (lmask >> i & 1) ? rv[i][c] : 0.f;
      if (fused && !is_dy) {
        const float ga = c ? cf.z : cf.x, gb = c ? cf.w : cf.y;
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (lmask >> i & 1) ? silu_rcp(e[i] * ga + gb) : 0.f;
      }
      uint4 h, l;
      h.x = split_bf16_pair(e[0], e[1], l.x);
      h.y = split_bf16_pair(e[2], e[3], l.y);
      h.z = split_bf16_pair(e[4], e[5], l.z);
      h.w = split_bf16_pair(e[6], e[7], l.w);
      const int ch = 2 * cp + c;
      if (!is_dy) {
        unsigned char* dst = sm + ch * RP + (slot + 8 * oct) * 2;
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + RING_LO) = l;
        if (slot + 8 * oct == 0) {  // positions 0-7 are mirrored behind the ring's end (the row's last 16 bytes): a k-step of 16 positions
          *reinterpret_cast<uint4*>(dst + 2 * R) = h;            // may start 8 before the end when Wp is 8 mod 16
          *reinterpret_cast<uint4*>(dst + 2 * R + RING_LO) = l;
        }
      } else {
        unsigned char* dst = sm + DYB + buf * DY_BUF + ch * DPITCH + oct * 16;
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + DY_LO) = l;
        // the elements the +-1 shifts pull in from the neighbouring pieces: low half = position P0 + 8, high half = position P0 - 1
        uint2 nb;
        nb.x = split_bf16_pair((lmask >> 9 & 1) ? nv[1][c] : 0.f, (lmask >> 8 & 1) ? nv[0][c] : 0.f, nb.y);
        *reinterpret_cast<uint2*>(sm + DYB + buf * DY_BUF + DY_NB + ch * NPITCH + oct * 8) = nb;
        bsum[c] += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
      }
    }
  };

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  // ---------------------------------------------------------------- prologue: x loads 0 .. J0 (all eight waves help), dY chunk 0
  // x load j covers positions [p0 - Wp + j CH, + CH) and lives at ring slot (j CH) mod R
  const float* xsrc = (src1 ? p.a1 + ci0 : p.a2 + (ci0 - p.C1)) + 2 * cp;
  const int xld = src1 ? p.lda1 : p.lda2;
  for (int j = wave >> 2; j <= a.J0; j += 2) {
    // (waves 4-7 act as x loaders here: same item geometry, the x source)
    bool ok;
    const int P0 = p0 - Wp + j * CH + 8 * oct;
    ok = P0 >= 0 && P0 < a.n_pos;
    const unsigned Pu = ok ? (unsigned)P0 : 0u;
    const unsigned row = __umulhi(Pu, a.wp_magic);
    const int col0 = (int)(Pu - row * (unsigned)Wp);
    const unsigned img = __umulhi(row, a.rpi_magic);
    const int y = (int)(row - img * (unsigned)a.rows_per_img);
    ok = ok && y < H;
    const float* base = xsrc + ((long long)((int)img * H + y) * W + col0) * xld;
    f32x2 t[8];
    const f32x4 pcf = fused ? *coef_of(ok ? img : 0u) : f32x4{1.f, 0.f, 1.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const bool v = ok && col0 + i < W;
      t[i] = *reinterpret_cast<const f32x2*>(v ? base + (long long)i * xld : xsrc);
      if (fused) t[i] = f32x2{silu_rcp(t[i][0] * pcf.x + pcf.y), silu_rcp(t[i][1] * pcf.z + pcf.w)};
      if (!v) t[i] = f32x2{0.f, 0.f};
    }
    const int slot = (j * CH) % R;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const int c = k;
      uint4 h, l;
      h.x = split_bf16_pair(t[0][c], t[1][c], l.x);
      h.y = split_bf16_pair(t[2][c], t[3][c], l.y);
      h.z = split_bf16_pair(t[4][c], t[5][c], l.z);
      h.w = split_bf16_pair(t[6][c], t[7][c], l.w);
      unsigned char* dst = sm + (2 * cp + c) * RP + (slot + 8 * oct) * 2;
      *reinterpret_cast<uint4*>(dst) = h;
      *reinterpret_cast<uint4*>(dst + RING_LO) = l;
      if (slot + 8 * oct == 0) {
        *reinterpret_cast<uint4*>(dst + 2 * R) = h;
        *reinterpret_cast<uint4*>(dst + 2 * R + RING_LO) = l;
      }
    }
  }
  if (is_dy) {
    request(p0 + 8 * oct);
    stage(0, 0);
    if (n_it > 1) request(p0 + CH + 8 * oct);
  } else if (n_it > 1) {
    request(p0 - Wp + (a.J0 + 1) * CH + 8 * oct);
  }
  __syncthreads();

  // ---------------------------------------------------------------- main loop: one chunk per iteration
  const unsigned char* a_lane = sm + (wu * 32 + l31) * RP + half * 16;          // + slot * 2 (+ RING_LO)
  const int b_lane = (wv * 32 + l31) * DPITCH + half * 16, n_lane = (wv * 32 + l31) * NPITCH + half * 8;
  int rb = 0;                                    // (it CH) mod R: ring slot of position p - Wp
  int xs = ((a.J0 + 1) * CH) % R;                // ring slot of the x load staged in this iteration
  auto loader_phase = [&](int it) {
    if (it + 1 < n_it) {
      stage(xs, (it + 1) & 1);
      if (it + 2 < n_it) request(is_dy ? p0 + (it + 2) * CH + 8 * oct : p0 - Wp + (it + 2 + a.J0) * CH + 8 * oct);
    }
  };
  for (int it = 0; it < n_it; ++it) {
    if (grp == 0) loader_phase(it);
    const unsigned char* dyb = sm + DYB + (it & 1) * DY_BUF;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int s = 2 * grp + ss;
      const uint4 bh = *reinterpret_cast<const uint4*>(dyb + b_lane + s * 32);
      const uint4 bl = *reinterpret_cast<const uint4*>(dyb + DY_LO + b_lane + s * 32);
      const uint2 nb = *reinterpret_cast<const uint2*>(dyb + DY_NB + n_lane + s * 16);
      int slot = rb + 16 * s;
      if (slot >= R) slot -= R;
      uint4 ah = *reinterpret_cast<const uint4*>(a_lane + slot * 2);
      uint4 al = *reinterpret_cast<const uint4*>(a_lane + RING_LO + slot * 2);
      // dY[q' + 1] (tap kw = 0), dY[q'] (kw = 1), dY[q' - 1] (kw = 2) as operand fragments
      const unsigned h1 = __builtin_amdgcn_alignbit(bh.y, bh.x, 16), h2 = __builtin_amdgcn_alignbit(bh.z, bh.y, 16),
                     h3 = __builtin_amdgcn_alignbit(bh.w, bh.z, 16);
      const unsigned g1 = __builtin_amdgcn_alignbit(bl.y, bl.x, 16), g2 = __builtin_amdgcn_alignbit(bl.z, bl.y, 16),
                     g3 = __builtin_amdgcn_alignbit(bl.w, bl.z, 16);
      bf16x8 Bh[3], Bl[3];
      Bh[0] = as_frag(h1, h2, h3, __builtin_amdgcn_alignbit(nb.x, bh.w, 16));
      Bh[1] = __builtin_bit_cast(bf16x8, bh);
      Bh[2] = as_frag(__builtin_amdgcn_alignbit(bh.x, nb.x, 16), h1, h2, h3);
      Bl[0] = as_frag(g1, g2, g3, __builtin_amdgcn_alignbit(nb.y, bl.w, 16));
      Bl[1] = __builtin_bit_cast(bf16x8, bl);
      Bl[2] = as_frag(__builtin_amdgcn_alignbit(bl.x, nb.y, 16), g1, g2, g3);
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const bf16x8 Ah = __builtin_bit_cast(bf16x8, ah), Al = __builtin_bit_cast(bf16x8, al);
        if (kh < 2) {  // the next row offset's fragments are requested before this one's nine MFMAs
          int sn = slot + Wp;
          if (sn >= R) sn -= R;
          slot = sn;
          ah = *reinterpret_cast<const uint4*>(a_lane + sn * 2);
          al = *reinterpret_cast<const uint4*>(a_lane + RING_LO + sn * 2);
        }
        // pass-major: consecutive MFMAs write different accumulators; lo products first
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) if constexpr (!VMM_SINGLE_PASS) acc[kh * 3 + kw] = vmm_mfma16(Ah, Bl[kw], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) if constexpr (!VMM_SINGLE_PASS) acc[kh * 3 + kw] = vmm_mfma16(Al, Bh[kw], acc[kh * 3 + kw]);
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) acc[kh * 3 + kw] = vmm_mfma16(Ah, Bh[kw], acc[kh * 3 + kw]);
      }
    }
    if (grp == 1) loader_phase(it);
    rb += CH;
    if (rb >= R) rb -= R;
    xs += CH;
    if (xs >= R) xs -= R;
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue
  // acc[tap][r]: row = input channel (r & 3) + 8 (r >> 2) + 4 half of this wave's 32, column = output channel l31 of its 32.
  // The two wave groups hold partial sums of the same block: group 1 hands its accumulators over through LDS (two rounds: the ring and the
  // chunk buffers are dead), group 0 adds them and stores the workgroup's partial block with 16-byte stores -- ONE plain store per element
  // and workgroup.  (fp32 atomics straight into dw cost 60-70 us per launch whatever the shape: 19 M atomic operations, the L2's atomic
  // throughput; 38 MB of stores + the reduction pass take a third of that.)
  uint4* xch = reinterpret_cast<uint4*>(sm);
  constexpr int ROUND[3] = {0, 5, 9};
#pragma unroll
  for (int rd = 0; rd < 2; ++rd) {
    if (grp == 1) {
#pragma unroll
      for (int t = ROUND[rd]; t < ROUND[rd + 1]; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint4 v = {__float_as_uint(acc[t][4 * j]), __float_as_uint(acc[t][4 * j + 1]), __float_as_uint(acc[t][4 * j + 2]), __float_as_uint(acc[t][4 * j + 3])};
          xch[(((t - ROUND[rd]) * 4 + wq) * 4 + j) * 64 + lane] = v;
        }
    }
    __syncthreads();
    if (grp == 0) {
#pragma unroll
      for (int t = ROUND[rd]; t < ROUND[rd + 1]; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const uint4 v = xch[(((t - ROUND[rd]) * 4 + wq) * 4 + j) * 64 + lane];
          acc[t][4 * j] += __uint_as_float(v.x);
          acc[t][4 * j + 1] += __uint_as_float(v.y);
          acc[t][4 * j + 2] += __uint_as_float(v.z);
          acc[t][4 * j + 3] += __uint_as_float(v.w);
        }
    }
    __syncthreads();
  }
  if (grp == 0) {
    if (a.part) {
      // partial block layout [tap][quadrant][j][lane] x 4 floats: coalesced 16-byte stores, the reduction pass undoes the order
      f32x4* dst = reinterpret_cast<f32x4*>(a.part) + ((long long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (PART_FLOATS / 4);
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int j = 0; j < 4; ++j) dst[((t * 4 + wq) * 4 + j) * 64 + lane] = f32x4{acc[t][4 * j], acc[t][4 * j + 1], acc[t][4 * j + 2], acc[t][4 * j + 3]};
    } else {
#pragma unroll
      for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int i = t * Cin + ci0 + wu * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          atomicAdd(&a.dw[(long long)i * p.Cout + co0 + wv * 32 + l31], acc[t][r]);
        }
    }
  }
  if (a.dbias && blockIdx.x == 0) {  // (workgroup-uniform) eight pieces x 64 channels of partial column sums -> one value per channel
    float* red = reinterpret_cast<float*>(sm);  // (the exchange rounds above ended with a barrier)
    if (is_dy) {
      red[oct * 64 + 2 * cp] = bsum[0];
      red[oct * 64 + 2 * cp + 1] = bsum[1];
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) s += red[o * 64 + tid];
      if (a.bias_part) a.bias_part[(long long)blockIdx.z * p.Cout + co0 + tid] = s;
      else atomicAdd(&a.dbias[co0 + tid], s);
    }
  }
}

// dw[(tap, ci)][co] += sum over row slices z of the partial blocks (fixed order: bit-reproducible); body in wgrad_reduce.h (shared with vmm_reduce_batch)
static_assert(PART_FLOATS == vmm_reduce::W9_PART_FLOATS, "partial block size");
__global__ __launch_bounds__(256) void wgrad9_reduce_kernel(const float* __restrict__ part, int nz, int tiles_x, int tiles_y, float* __restrict__ dw, int Cin,
                                                            int Cout, const float* __restrict__ bias_part, float* __restrict__ dbias, int n_main) {
  __shared__ f32x4 red[8][32];
  vmm_reduce::w9_body(part, nz, tiles_x, tiles_y, dw, Cin, Cout, bias_part, dbias, n_main, (int)blockIdx.x, (int)blockIdx.y, red);
}

}  // namespace

// Geometry shared by the launcher and the workspace query.  Returns false outside the envelope.
static bool w9_setup(const vmm_conv_desc& d, int32_t lddy, W9Args& a, int& gz) {
  const bool shape_ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.off_h == -1 && d.off_w == -1 && d.sgn_h == 1 && d.sgn_w == 1 && d.Hv == d.Hin &&
                        d.Wv == d.Win && d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && (d.a_mode == 0 || (d.a_mode == 1 && d.a_coef && d.a_imgs_per_sample > 0)) && !d.a_img_mod && !d.wrap_h &&
                        !d.wrap_w;
  const bool chan_ok = d.C1 > 0 && d.C1 % 64 == 0 && d.C2 % 64 == 0 && d.Cout % 64 == 0 && (d.lda1 & 1) == 0 && (!d.C2 || (d.lda2 & 1) == 0) && (lddy & 1) == 0;
  if (!shape_ok || !chan_ok || d.nimg <= 0 || d.Hin <= 0 || d.Win <= 0) return false;
  a.p = d;
  a.Wp = d.Win % 8 == 0 ? d.Win : (d.Win / 8 + 1) * 8;
  a.rows_per_img = d.Hin + 1;
  const long long n_pos = (long long)d.nimg * a.rows_per_img * a.Wp;
  if ((n_pos + 2 * CH) * a.Wp >= (1ll << 32) || n_pos / a.Wp * a.rows_per_img >= (1ll << 32)) return false;  // (the magic divisions' range)
  a.n_pos = (int)n_pos;
  a.wp_magic = (unsigned)(0x100000000ull / (unsigned)a.Wp) + 1u;
  a.rpi_magic = (unsigned)(0x100000000ull / (unsigned)a.rows_per_img) + 1u;
  a.nchunks = (int)((n_pos + CH - 1) / CH);
  a.J0 = (2 * a.Wp + CH - 1) / CH;
  a.R = (a.J0 + 2) * CH;
  if ((size_t)128 * (2 * a.R + 16) + 2 * (size_t)DY_BUF > 160 * 1024) return false;
  // row slices: one round of workgroups, one per CU, whatever the caller's nsplit (which is sized for the generic kernel's tiles)
  const int blocks_xy = ((d.C1 + d.C2) / 64) * (d.Cout / 64);
  const int nz = max(1, min(a.nchunks, 256 / blocks_xy));
  a.chunks_per_wg = (a.nchunks + nz - 1) / nz;
  gz = (a.nchunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  return true;
}

// floats of workspace vmm_conv3x3_wgrad_bf16x3 wants for this layer (partial blocks of every row slice + one bias row per slice); 0 = the
// layer is outside the kernel's envelope (3 x 3 / stride 1 / pad 1, zero padding, C1 / C2 / Cout multiples of 64; a_mode 1 = the producer's
// GroupNorm * FiLM -> SiLU on source a1 is applied in the loader)
#if !VMM_SINGLE_PASS
extern "C" int64_t vmm_conv3x3_wgrad_bf16x3_workspace(const vmm_conv_desc* dp, int32_t lddy) {
  W9Args a;
  int gz = 0;
  if (!w9_setup(*dp, lddy, a, gz)) return 0;
  const long long tiles = (long long)((dp->C1 + dp->C2) / 64) * (dp->Cout / 64);
  return (int64_t)gz * tiles * PART_FLOATS + (int64_t)gz * dp->Cout;
}

// The second stage of the call vmm_conv3x3_wgrad_*(d, dy, lddy, dw_packed, dbias, workspace) as a job of vmm_reduce_batch: a caller that sets
// d->defer_reduce runs the first stage alone, keeps the workspace and totals the pending blocks of many layers in one launch.  1 outside the envelope.
extern "C" int vmm_conv3x3_wgrad_reduce_job(const vmm_conv_desc* dp, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job) {
  W9Args a;
  int gz = 0;
  if (!workspace || !job || !w9_setup(*dp, lddy, a, gz)) return 1;
  const int tx = (dp->C1 + dp->C2) / 64, ty = dp->Cout / 64;
  const int n_main = PART_FLOATS / 4 / 32;
  job->part = workspace; job->out = dw_packed;
  job->bias_part = dbias ? workspace + (long long)gz * tx * ty * PART_FLOATS : nullptr; job->dbias = dbias;
  job->kind = 1; job->nz = gz; job->tiles_x = tx; job->tiles_y = ty; job->Cin = dp->C1 + dp->C2; job->Cout = dp->Cout; job->ld = 0;
  job->n_main = n_main; job->gx = n_main + (dbias ? cdiv(dp->Cout, 32) : 0); job->wgs = job->gx * tx * ty; job->wg0 = 0;
  return 0;
}

// Same contract as vmm_conv_wgrad_bf16x3 (which forwards the shapes inside this kernel's envelope here, without a workspace).  Returns 1
// (nothing launched) outside the envelope.  workspace = vmm_conv3x3_wgrad_bf16x3_workspace(d, lddy) floats (contents irrelevant): the row
// slices leave partial blocks there and a second launch totals them in a fixed order (bit-reproducible); NULL: fp32 atomics into dw_packed.
#endif
extern "C" int VMM_X3(vmm_conv3x3_wgrad_, )(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                                        vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  W9Args a;
  int gz = 0;
  if (!w9_setup(d, lddy, a, gz)) return (d.nimg <= 0 || d.Hin <= 0 || d.Win <= 0) && d.KH == 3 ? 0 : 1;
  a.dy = dy; a.lddy = lddy; a.dw = dw_packed; a.dbias = dbias;
  const int tx = (d.C1 + d.C2) / 64, ty = d.Cout / 64;
  a.part = workspace;
  a.bias_part = workspace ? workspace + (long long)gz * tx * ty * PART_FLOATS : nullptr;
  const size_t shm = (size_t)128 * (2 * a.R + 16) + 2 * (size_t)DY_BUF;  // (>= the 82 KB the accumulator exchange of the epilogue needs)
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad9_x3_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad9_x3_kernel, dim3(tx, ty, gz), dim3(512), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  if (workspace && !d.defer_reduce) {  // (defer_reduce: the caller totals the blocks later, vmm_conv3x3_wgrad_reduce_job + vmm_reduce_batch)
    const int n_main = PART_FLOATS / 4 / 32;
    hipLaunchKernelGGL(wgrad9_reduce_kernel, dim3(n_main + (dbias ? cdiv(d.Cout, 32) : 0), tx * ty), dim3(256), 0, (hipStream_t)stream, workspace, gz, tx, ty,
                       dw_packed, d.C1 + d.C2, d.Cout, dbias ? a.bias_part : nullptr, dbias, n_main);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
