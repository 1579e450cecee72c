// Fused temporal-attention block for the C = 64 levels, everything on the split-bf16 matrix cores, gfx950.
//
//   out = x + to_out( softmax_attention( rotary(to_qkv( LayerNorm(x) )) ) )        (vddp.py:615,630,680: Residual(PreNorm(Attention)))
//
// Attention runs along the frame axis of every pixel: T <= 16 frames plus <= 16 conditioning tokens per pixel and head.  The
// unfused path writes the 768-wide qkv rows and the 256-wide attention output to HBM (2.5 GB + 0.8 GB per site at batch 8);
// here x is read once and out written once, and -- unlike the first version of this kernel, which ran the 22-key softmax on the
// vector ALU with one thread per row -- the scores and the value mix are matrix-core work too.
//
// Tile = 2 pixels x 16 frame slots = 32 rows (row m = pixel a * 16 + frame t; slots t >= T are zero rows).  One wave = one head
// (512-thread workgroups), its q/k/v weight fragments stay in registers.  As in linattn_block.hip, every MFMA result is consumed as
// an operand of the next MFMA straight from the accumulator registers (lane = column, registers = rows; both operands of the
// next product enumerate the contraction index in register order), so nothing is transposed or staged:
//   q^T, k^T [d][m]   = W_q,k (A, fragment-order weights) . y^T (B, LayerNorm rows as fragments); rotary + scale in registers
//   v [m][d]          = y (A) . W_v^T (B)
//   s^T [key][query]  = k (A <- k^T accumulators) . q^T (B <- q^T accumulators); 32 x 32, the two 16 x 16 diagonal blocks are used
//   s_tok^T [tok][query] = ek (A, fragments built once per workgroup in LDS) . q^T
//   softmax over the 8 + 8 register-resident scores of a lane and its lane ^ 32 partner; relative-position bias from LDS
//   o^T [d][query]    = v^T (A <- v accumulators) . p^T (B <- s^T registers, zero outside the own pixel)  +  ev^T (A, LDS) . p_tok^T
//   part [m][c]       = o (A <- o^T accumulators) . W_out,h^T (B, fmt-3 fragments)
//   out               = sum over the 8 heads (LDS) + x
// 75 MFMA 32x32x16 per tile and head (3 split-bf16 passes each); padded frame slots and the off-diagonal score blocks are the price
// for needing no data movement between the products.
#include "igemm_common.h"
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TC = 64;    // channels
constexpr int HEADS = 8;  // = waves per workgroup
constexpr int DHd = 32;
constexpr int HID = HEADS * DHd;
constexpr int TB_TRACE_N = 40;  // intervals recorded by VMM_TB_TRACE

struct TBArgs {
  const float* x; int ldx;   // (the A16 instance of the second kernel reads / writes these as bf16s: bf16-stored maps of the "bf16" mode)
  const float* gamma;
  const uint4* wqkv;  // fmt 2 fragments of to_qkv (768, 64)
  const uint4* wout;  // fmt 3 fragments of to_out (64, 256)
  const float* ek; const float* ev; int ntok;
  const float* bias; int bias_on_cond;
  const float* rot;   // [T][16][2] cos, sin
  float* out; int ldo;
  int T, HW, nsplit, tps;  // tps: pixel pairs per split
  float q_scale, eps;
  int group_mode;  // second version: which heads form the late group (0: heads 4-7, 1: odd heads)
  unsigned long long* trace;  // VMM_TB_TRACE: s_memtime stamps of workgroup 0, [interval][wave][4]
};

__device__ __forceinline__ unsigned pack_split(float a, float b, unsigned& lo) { return split_bf16_pair(a, b, lo); }

__device__ __forceinline__ void split8(const f32x16& c, int r0, uint4& hi, uint4& lo) {
  hi.x = pack_split(c[r0 + 0], c[r0 + 1], lo.x);
  hi.y = pack_split(c[r0 + 2], c[r0 + 3], lo.y);
  hi.z = pack_split(c[r0 + 4], c[r0 + 5], lo.z);
  hi.w = pack_split(c[r0 + 6], c[r0 + 7], lo.w);
}

__device__ __forceinline__ void split8v(const float (&v)[8], uint4& hi, uint4& lo) {
  hi.x = pack_split(v[0], v[1], lo.x);
  hi.y = pack_split(v[2], v[3], lo.y);
  hi.z = pack_split(v[4], v[5], lo.z);
  hi.w = pack_split(v[6], v[7], lo.w);
}

// ONE ("bf16" throughput mode, BASELINE.json configs[3]): the hi planes only, one pass; the lo halves of every split in the kernel then have no
// reader and the compiler drops them (and the lo weight fragments' registers) with it
template <bool ONE>
__device__ __forceinline__ f32x16 mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 c) {
  if constexpr (ONE) return vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c);
  return c;
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  return c;
}

// index of the contraction / row slot that element j of lane half lk holds in k16 step s (accumulator register order)
__device__ __forceinline__ int slot(int s, int lk, int j) { return (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * lk; }

template <bool ONE>
__global__ __launch_bounds__(512, 2) void temporal_block_kernel(const TBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);                     // [8 heads][32 rows][64 channels]
  uint4* ekf = reinterpret_cast<uint4*>(red + HEADS * 32 * TC);        // [8 heads][2 steps][hi|lo][64 lanes]
  uint4* evf = ekf + HEADS * 4 * 64;                                   // [8 heads][hi|lo][64 lanes]
  float* biasf = reinterpret_cast<float*>(evf + HEADS * 2 * 64);       // [8 heads][2 halves][16 frames][8]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(biasf + HEADS * 2 * 16 * 8);  // [32 rows][hi 64 | lo 64 | pad 8]
  constexpr int YPITCH = 2 * TC + 8;                                   // 272 bytes = 17 x 16: conflict-free ds_read_b128 over consecutive rows
  float* rotf = reinterpret_cast<float*>(ytile + 32 * YPITCH);          // [2 halves][16 frames][8 pairs][cos, sin]

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int pa = lrow >> 4, ft = lrow & 15;  // pixel of the pair, frame slot
  const int T = a.T, HW = a.HW;
  const int b = blockIdx.x / a.nsplit, split = blockIdx.x - b * a.nsplit;
  const int pairs = HW / 2;
  const int p_begin = split * a.tps, p_end = min(pairs, p_begin + a.tps);
  const int ntok = a.ek ? a.ntok : 0;

  // ---- per-workgroup tables in LDS
  {
    // conditioning keys as an A operand (rows = tokens, contraction = d), values as an A operand (rows = d, contraction = tokens)
    uint4* eks = ekf + h * 4 * 64;
    uint4* evs = evf + h * 2 * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = lrow < ntok ? a.ek[((long long)b * ntok + lrow) * HID + h * DHd + slot(s, lk, j)] : 0.f;
      uint4 hi, lo;
      split8v(v, hi, lo);
      eks[(s * 2 + 0) * 64 + lane] = hi;
      eks[(s * 2 + 1) * 64 + lane] = lo;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      v[j] = tk < ntok ? a.ev[((long long)b * ntok + tk) * HID + h * DHd + lrow] : 0.f;
    }
    uint4 hi, lo;
    split8v(v, hi, lo);
    evs[lane] = hi;
    evs[64 + lane] = lo;
    // relative-position bias of query frame t against the 8 key frames a lane half holds: [h][lk][t][j]
    for (int i = tid; i < HEADS * 2 * 16 * 8; i += 512) {
      const int j = i & 7, t = (i >> 3) & 15, l2 = (i >> 7) & 1, hh = i >> 8;
      const int tk = slot(0, l2, j);
      biasf[i] = (t < T && tk < T) ? a.bias[(hh * T + t) * T + tk] : 0.f;
    }
  }

  // ---- per-wave constants in registers
  uint4 wqh[4], wql[4], wkh[4], wkl[4], wvh[4], wvl[4];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4* q = a.wqkv + (((long long)h * 4 + s) * 2) * 64 + lane;
    const uint4* k = a.wqkv + (((long long)(HEADS + h) * 4 + s) * 2) * 64 + lane;
    const uint4* v = a.wqkv + (((long long)(2 * HEADS + h) * 4 + s) * 2) * 64 + lane;
    wqh[s] = q[0]; wql[s] = q[64];
    wkh[s] = k[0]; wkl[s] = k[64];
    wvh[s] = v[0]; wvl[s] = v[64];
  }
  const uint4* wo_l = a.wout + ((long long)2 * h * 2) * 64 + lane;  // to_out fragments are re-read per tile (8 KB per head, L2-resident)
  // rotary factors of a (frame slot, lane half): (cos, sin) of the 8 (even, odd) feature pairs the lane holds in accumulator registers
  // (2i, 2i+1).  Kept in LDS, not in 16 registers: the q/k/v weight fragments already take 96 and the kernel sits at the 256 limit.
  for (int i = tid; i < 2 * 16 * 8; i += 512) {
    const int pr = i & 7, t = (i >> 3) & 15, l2 = i >> 7;
    const int d = slot(pr >> 2, l2, (2 * pr) & 7);  // feature index of register 2 pr
    const float2 cs = t < T ? *reinterpret_cast<const float2*>(a.rot + (t * 16 + (d >> 1)) * 2) : make_float2(1.f, 0.f);
    rotf[i * 2] = cs.x;
    rotf[i * 2 + 1] = cs.y;
  }
  const float* rot_l = rotf + ((lk * 16 + ft) * 8) * 2;
  auto rotate2 = [&](f32x16& u, f32x16& v) {  // both q^T and k^T with one read of the factors
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(rot_l + i4 * 4);  // pairs 2 i4, 2 i4 + 1
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = 2 * i4 + k;
        const float c = k ? cs.z : cs.x, sn = k ? cs.w : cs.y;
        float e = u[2 * i], o = u[2 * i + 1];
        u[2 * i] = e * c - o * sn;
        u[2 * i + 1] = o * c + e * sn;
        e = v[2 * i]; o = v[2 * i + 1];
        v[2 * i] = e * c - o * sn;
        v[2 * i + 1] = o * c + e * sn;
      }
    }
  };
  __syncthreads();

  const float* bias_l = biasf + ((h * 2 + lk) * 16 + ft) * 8;
  // reduction role
  const int rm = tid >> 4, rcol = (tid & 15) * 4;
  const int rpa = rm >> 4, rft = rm & 15;

  // The 32 x 64 input tile is read ONCE per workgroup: thread (row rm, channels rcol .. rcol+3) loads one float4 -- a tile ahead, so the
  // HBM latency hides under the previous tile's MFMAs -- the 16 lanes of a row normalise it, and the bf16 hi|lo rows go to LDS where
  // all eight head-waves pick up their fragments.  The same thread owns that element again in the head sum: x is never re-read.
  const f32x4 gam = *reinterpret_cast<const f32x4*>(a.gamma + rcol);
  auto load_x = [&](int pp) {
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (rft < T && pp < p_end) v = *reinterpret_cast<const f32x4*>(a.x + (((long long)b * T + rft) * HW + pp * 2 + rpa) * a.ldx + rcol);
    return v;
  };
  f32x4 x_next = load_x(p_begin);
  for (int pp = p_begin; pp < p_end; ++pp) {
    const f32x4 x_cur = x_next;
    x_next = load_x(pp + 1);
    // ---- LayerNorm of the 32 rows (16 lanes per row), rows to LDS as bf16 hi | lo
    {
      float s = (x_cur.x + x_cur.y) + (x_cur.z + x_cur.w);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) s += __shfl_xor(s, o, 64);
      const float mean = s * (1.0f / TC);
      const f32x4 c = {x_cur.x - mean, x_cur.y - mean, x_cur.z - mean, x_cur.w - mean};
      float q = (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) q += __shfl_xor(q, o, 64);
      const float rstd = 1.0f / sqrtf(q * (1.0f / TC) + a.eps);
      unsigned l0, l1;
      const unsigned h0 = pack_split(c.x * rstd * gam.x, c.y * rstd * gam.y, l0);
      const unsigned h1 = pack_split(c.z * rstd * gam.z, c.w * rstd * gam.w, l1);
      *reinterpret_cast<uint2*>(ytile + rm * YPITCH + rcol) = make_uint2(h0, h1);
      *reinterpret_cast<uint2*>(ytile + rm * YPITCH + TC + rcol) = make_uint2(l0, l1);
    }
    __syncthreads();  // tile rows visible; also: every wave has finished the previous tile's head sum (red is free again)
    uint4 yh[4], yl[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned short* q = ytile + lrow * YPITCH + i * 16 + lk * 8;
      yh[i] = *reinterpret_cast<const uint4*>(q);
      yl[i] = *reinterpret_cast<const uint4*>(q + TC);
    }
    // ---- projections
    f32x16 qt = zero16(), kt = zero16(), vt = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      qt = mfma3<ONE>(wqh[s], wql[s], yh[s], yl[s], qt);  // [d][m]
      kt = mfma3<ONE>(wkh[s], wkl[s], yh[s], yl[s], kt);  // [d][m]
      vt = mfma3<ONE>(yh[s], yl[s], wvh[s], wvl[s], vt);  // [m][d]
    }
    rotate2(qt, kt);
    uint4 qh[2], ql[2];
    split8(qt, 0, qh[0], ql[0]);
    split8(qt, 8, qh[1], ql[1]);
    // ---- scores: keys x queries
    f32x16 st = zero16(), sk = zero16();
    const uint4* eks = ekf + h * 4 * 64 + lane;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 kh, kl;
      split8(kt, s * 8, kh, kl);
      st = mfma3<ONE>(kh, kl, qh[s], ql[s], st);
      if (ntok) sk = mfma3<ONE>(eks[(s * 2) * 64], eks[(s * 2 + 1) * 64], qh[s], ql[s], sk);
    }
    // to_out fragments of this head (8 KB, L2-resident): requested here, a softmax / value phase before their use -- next to their use
    // each (column tile, step) pair paid an L2 round trip of its own; earlier than here the registers do not exist (q/k/v tiles live)
    uint4 woh[2][2], wol[2][2];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
        const uint4* q = wo_l + ((ct * 16 + s) * 2) * 64;
        woh[ct][s] = q[0];
        wol[ct][s] = q[64];
      }
    __builtin_amdgcn_sched_barrier(0);
    // ---- softmax over this query's 8 (+8) frame keys and 8 (+8) tokens: lane half lk holds slots {0-3, 8-11} + 4 lk
    const f32x4 bz0 = *reinterpret_cast<const f32x4*>(bias_l), bz1 = *reinterpret_cast<const f32x4*>(bias_l + 4);
    const float bz[8] = {bz0.x, bz0.y, bz0.z, bz0.w, bz1.x, bz1.y, bz1.z, bz1.w};
    const unsigned pmask = pa ? 0xffffffffu : 0u;
    float f[8], g[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      // bit blend, NOT `pa ? st[8 + j] : st[j]`: the compiler turns that select into st[(pa ? 8 : 0) + j], a run-time index into the
      // 16 accumulator registers, i.e. a 16-way compare / select cascade per element (~400 of the loop's ~700 VALU instructions)
      const float sv = __uint_as_float((__float_as_uint(st[8 + j]) & pmask) | (__float_as_uint(st[j]) & ~pmask));
      f[j] = tk < T ? sv * a.q_scale + bz[j] : -INFINITY;
      g[j] = tk < ntok ? sk[j] * a.q_scale + (a.bias_on_cond ? bz[j] : 0.f) : -INFINITY;
      mx = fmaxf(mx, fmaxf(f[j], g[j]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = __expf(f[j] - mx);
      g[j] = __expf(g[j] - mx);
      sum += f[j] + g[j];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float p0[8], p1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] *= inv;
      g[j] *= inv;
      p0[j] = pa ? 0.f : f[j];  // keys of pixel 0 <-> k16 step 0
      p1[j] = pa ? f[j] : 0.f;  // keys of pixel 1 <-> k16 step 1
    }
    // ---- o^T = v^T . p^T (+ token values)
    f32x16 ot = zero16();
    {
      uint4 vh, vl, ph, pl;
      split8(vt, 0, vh, vl);
      split8v(p0, ph, pl);
      ot = mfma3<ONE>(vh, vl, ph, pl, ot);
      split8(vt, 8, vh, vl);
      split8v(p1, ph, pl);
      ot = mfma3<ONE>(vh, vl, ph, pl, ot);
      if (ntok) {
        const uint4* evs = evf + h * 2 * 64 + lane;
        split8v(g, ph, pl);
        ot = mfma3<ONE>(evs[0], evs[64], ph, pl, ot);
      }
    }
    // ---- this head's share of to_out
    f32x16 pc[2] = {zero16(), zero16()};
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 oh, ol;
      split8(ot, s * 8, oh, ol);
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) pc[ct] = mfma3<ONE>(oh, ol, woh[ct][s], wol[ct][s], pc[ct]);
    }
    float* rb = red + (h * 32) * TC;  // (free: the barrier after the LayerNorm above came after everyone's previous head sum)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = (r & 3) + 8 * (r >> 2) + 4 * lk;
      rb[m * TC + lrow] = pc[0][r];
      rb[m * TC + 32 + lrow] = pc[1][r];
    }
    __syncthreads();
    if (rft < T) {
      f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int w = 0; w < HEADS; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * 32 + rm) * TC + rcol);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const long long row = ((long long)b * T + rft) * HW + pp * 2 + rpa;
      acc.x += x_cur.x; acc.y += x_cur.y; acc.z += x_cur.z; acc.w += x_cur.w;  // residual: the element this thread normalised
      *reinterpret_cast<f32x4*>(a.out + row * a.ldo + rcol) = acc;
    }
  }
}


// ---------------------------------------------------------------------------------------------------------------------------------
// The same block at the C = 128 level (48 x 48 / 24 x 24 frames of the Lagrangian configuration): the unfused path there writes and re-reads
// the 768-wide qkv rows (623 MB per site at batch 8) for a 128-wide input.  Structure of the first kernel above (eight head-waves in lock step,
// two pixels x 16 frame slots per tile), with what the doubled contraction changes:
//   * a head's q / k / v weight fragments are 8 k16 steps x 3 x (hi | lo) = 192 registers -- they do not fit.  All three projections STREAM
//     their fragments from L2 through a four-step register ring (the fragments of step s + 3 are requested before the MFMAs of step s), one
//     projection after the other so that only one 32 x 32 accumulator and one ring are live: 48 KB per head and tile, 30 B / clk per CU at
//     the kernel's pace, half of what the L1 delivers;
//   * to_out has four 32-channel column tiles: their fragments ride the same ring, and the head sum goes through the [8][32][64] LDS buffer in
//     two 64-channel halves;
//   * the LayerNorm row staging handles two float4 per thread.
constexpr int TC2 = 128;

template <bool ONE>
__global__ __launch_bounds__(512) void temporal_block128_kernel(const TBArgs a) {
  constexpr int NS = TC2 / 16, NCT = TC2 / 32, NV = TC2 / 64;
  constexpr int YPITCH = 2 * TC2 + 8;                                  // 528 bytes = 33 x 16: conflict-free ds_read_b128 over consecutive rows
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);                     // [8 heads][32 rows][64 channels] (one half of the channels at a time)
  uint4* ekf = reinterpret_cast<uint4*>(red + HEADS * 32 * 64);        // [8 heads][2 steps][hi|lo][64 lanes]
  uint4* evf = ekf + HEADS * 4 * 64;                                   // [8 heads][hi|lo][64 lanes]
  float* biasf = reinterpret_cast<float*>(evf + HEADS * 2 * 64);       // [8 heads][2 halves][16 frames][8]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(biasf + HEADS * 2 * 16 * 8);  // [32 rows][hi 128 | lo 128 | pad 8]
  float* rotf = reinterpret_cast<float*>(ytile + 32 * YPITCH);          // [2 halves][16 frames][8 pairs][cos, sin]

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int pa = lrow >> 4, ft = lrow & 15;  // pixel of the pair, frame slot
  const int T = a.T, HW = a.HW;
  const int b = blockIdx.x / a.nsplit, split = blockIdx.x - b * a.nsplit;
  const int pairs = HW / 2;
  const int p_begin = split * a.tps, p_end = min(pairs, p_begin + a.tps);
  const int ntok = a.ek ? a.ntok : 0;

  // ---- per-workgroup tables in LDS (as in the first kernel)
  {
    uint4* eks = ekf + h * 4 * 64;
    uint4* evs = evf + h * 2 * 64;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j)
        v[j] = lrow < ntok ? a.ek[((long long)b * ntok + lrow) * HID + h * DHd + slot(s, lk, j)] : 0.f;
      uint4 hi, lo;
      split8v(v, hi, lo);
      eks[(s * 2 + 0) * 64 + lane] = hi;
      eks[(s * 2 + 1) * 64 + lane] = lo;
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      v[j] = tk < ntok ? a.ev[((long long)b * ntok + tk) * HID + h * DHd + lrow] : 0.f;
    }
    uint4 hi, lo;
    split8v(v, hi, lo);
    evs[lane] = hi;
    evs[64 + lane] = lo;
    for (int i = tid; i < HEADS * 2 * 16 * 8; i += 512) {
      const int j = i & 7, t = (i >> 3) & 15, l2 = (i >> 7) & 1, hh = i >> 8;
      const int tk = slot(0, l2, j);
      biasf[i] = (t < T && tk < T) ? a.bias[(hh * T + t) * T + tk] : 0.f;
    }
  }
  for (int i = tid; i < 2 * 16 * 8; i += 512) {
    const int pr = i & 7, t = (i >> 3) & 15, l2 = i >> 7;
    const int d = slot(pr >> 2, l2, (2 * pr) & 7);  // feature index of register 2 pr
    const float2 cs = t < T ? *reinterpret_cast<const float2*>(a.rot + (t * 16 + (d >> 1)) * 2) : make_float2(1.f, 0.f);
    rotf[i * 2] = cs.x;
    rotf[i * 2 + 1] = cs.y;
  }
  const float* rot_l = rotf + ((lk * 16 + ft) * 8) * 2;
  auto rotate1 = [&](f32x16& u) {
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(rot_l + i4 * 4);  // pairs 2 i4, 2 i4 + 1
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = 2 * i4 + k;
        const float c = k ? cs.z : cs.x, sn = k ? cs.w : cs.y;
        const float e = u[2 * i], o = u[2 * i + 1];
        u[2 * i] = e * c - o * sn;
        u[2 * i + 1] = o * c + e * sn;
      }
    }
  };
  __syncthreads();

  // streamed weight fragments: plane pair (hi, lo) of column tile nt, k16 step ks of a fmt-2 / fmt-3 matrix with KS steps per column tile
  const uint4* wq_l = a.wqkv + ((long long)h * NS * 2) * 64 + lane;
  const uint4* wk_l = a.wqkv + ((long long)(HEADS + h) * NS * 2) * 64 + lane;
  const uint4* wv_l = a.wqkv + ((long long)(2 * HEADS + h) * NS * 2) * 64 + lane;
  const uint4* wo_l = a.wout + ((long long)2 * h * 2) * 64 + lane;  // + (ct * 16 + s) * 128
  constexpr int RING = 4, AHEAD = 3;

  const float* bias_l = biasf + ((h * 2 + lk) * 16 + ft) * 8;
  // reduction role
  const int rm = tid >> 4, rcol = (tid & 15) * 4;
  const int rpa = rm >> 4, rft = rm & 15;
  f32x4 gam[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) gam[v] = *reinterpret_cast<const f32x4*>(a.gamma + v * 64 + rcol);
  auto load_x = [&](int pp, f32x4 (&d)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      d[v] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (rft < T && pp < p_end) d[v] = *reinterpret_cast<const f32x4*>(a.x + (((long long)b * T + rft) * HW + pp * 2 + rpa) * a.ldx + v * 64 + rcol);
    }
  };
  // Projections with the weight fragments streamed through the ring.  The group loop stays a LOOP (ring slot = step within the group, RING ==
  // 4): unrolled, the scheduler hoists every fragment request of the projection to its top -- 64 registers of weights instead of 32 -- and the
  // kernel spills.  Two accumulators are always in flight (q^T with k^T; the even and the odd steps of v): the three passes of one mfma3 form a
  // dependent chain, and a single chain leaves the matrix pipe idle for half of every MFMA's latency.
  static_assert(RING == 4 && AHEAD == 3 && NS % 4 == 0, "ring slot = step & 3");
  auto read_y = [&](int g, uint4 (&yh)[4], uint4 (&yl)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned short* q = ytile + lrow * YPITCH + (4 * g + i) * 16 + lk * 8;
      yh[i] = *reinterpret_cast<const uint4*>(q);
      yl[i] = *reinterpret_cast<const uint4*>(q + TC2);
    }
  };
#ifndef VMM_TB128_HOLD_Q
#define VMM_TB128_HOLD_Q 0
#endif
  constexpr bool HOLD_Q = VMM_TB128_HOLD_Q;  // 1: the q fragments of the head stay in registers (64 of them; 254 registers + 36 bytes of scratch), k and v streamed -- measured equal to streaming all three (2.52 ms for the family either way): the kernel is bound by its lock-step phases, not by the fragment traffic
  uint4 wqh[HOLD_Q ? NS : 1], wql[HOLD_Q ? NS : 1];
  if constexpr (HOLD_Q) {
#pragma unroll
    for (int s = 0; s < NS; ++s) { wqh[s] = wq_l[(s * 2) * 64]; wql[s] = wq_l[(s * 2 + 1) * 64]; }
  }
  auto project_qk = [&](f32x16& qt, f32x16& kt) {  // both [d][m]: the weights are the A operand
    uint4 qh_[RING], ql_[RING], kh_[RING], kl_[RING];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) {
      if constexpr (!HOLD_Q) { qh_[s] = wq_l[(s * 2) * 64]; ql_[s] = wq_l[(s * 2 + 1) * 64]; }
      kh_[s] = wk_l[(s * 2) * 64]; kl_[s] = wk_l[(s * 2 + 1) * 64];
    }
    qt = zero16();
    kt = zero16();
    if constexpr (HOLD_Q) {
#pragma unroll
      for (int g = 0; g < NS / 4; ++g) {  // (unrolled: the held fragments are indexed statically; the k ring's requests are pinned per step)
        uint4 yh[4], yl[4];
        read_y(g, yh, yl);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int sn = min(4 * g + i + AHEAD, NS - 1);
          kh_[(i + AHEAD) & 3] = wk_l[(sn * 2) * 64]; kl_[(i + AHEAD) & 3] = wk_l[(sn * 2 + 1) * 64];
          qt = mfma3<ONE>(wqh[4 * g + i], wql[4 * g + i], yh[i], yl[i], qt);
          kt = mfma3<ONE>(kh_[i], kl_[i], yh[i], yl[i], kt);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      return;
    }
#pragma unroll 1
    for (int g = 0; g < NS / 4; ++g) {
      uint4 yh[4], yl[4];
      read_y(g, yh, yl);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sn = min(4 * g + i + AHEAD, NS - 1);  // (past the end: re-requests the last step, unused)
        qh_[(i + AHEAD) & 3] = wq_l[(sn * 2) * 64]; ql_[(i + AHEAD) & 3] = wq_l[(sn * 2 + 1) * 64];
        kh_[(i + AHEAD) & 3] = wk_l[(sn * 2) * 64]; kl_[(i + AHEAD) & 3] = wk_l[(sn * 2 + 1) * 64];
        qt = mfma3<ONE>(qh_[i], ql_[i], yh[i], yl[i], qt);
        kt = mfma3<ONE>(kh_[i], kl_[i], yh[i], yl[i], kt);
      }
    }
  };
  auto project_v = [&]() -> f32x16 {  // [m][d]: the rows are the A operand
    uint4 rh[RING], rl[RING];
#pragma unroll
    for (int s = 0; s < AHEAD; ++s) { rh[s] = wv_l[(s * 2) * 64]; rl[s] = wv_l[(s * 2 + 1) * 64]; }
    f32x16 a0 = zero16(), a1 = zero16();
#pragma unroll 1
    for (int g = 0; g < NS / 4; ++g) {
      uint4 yh[4], yl[4];
      read_y(g, yh, yl);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int sn = min(4 * g + i + AHEAD, NS - 1);
        rh[(i + AHEAD) & 3] = wv_l[(sn * 2) * 64];
        rl[(i + AHEAD) & 3] = wv_l[(sn * 2 + 1) * 64];
        if (i & 1) a1 = mfma3<ONE>(yh[i], yl[i], rh[i], rl[i], a1);
        else a0 = mfma3<ONE>(yh[i], yl[i], rh[i], rl[i], a0);
      }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) a0[r] += a1[r];
    return a0;
  };

  f32x4 x_next[NV];
  load_x(p_begin, x_next);
  for (int pp = p_begin; pp < p_end; ++pp) {
    f32x4 x_cur[NV];
#pragma unroll
    for (int v = 0; v < NV; ++v) x_cur[v] = x_next[v];
    load_x(pp + 1, x_next);
    // ---- LayerNorm of the 32 rows (16 lanes per row), rows to LDS as bf16 hi | lo
    {
      float s1 = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) s1 += (x_cur[v].x + x_cur[v].y) + (x_cur[v].z + x_cur[v].w);
      const float mean = row_sum16(s1) * (1.0f / TC2);
      f32x4 c[NV];
      float s2 = 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        c[v] = f32x4{x_cur[v].x - mean, x_cur[v].y - mean, x_cur[v].z - mean, x_cur[v].w - mean};
        s2 += (c[v].x * c[v].x + c[v].y * c[v].y) + (c[v].z * c[v].z + c[v].w * c[v].w);
      }
      const float rstd = __builtin_amdgcn_rsqf(row_sum16(s2) * (1.0f / TC2) + a.eps);
#pragma unroll
      for (int v = 0; v < NV; ++v) {
        unsigned l0, l1;
        const unsigned h0 = pack_split(c[v].x * rstd * gam[v].x, c[v].y * rstd * gam[v].y, l0);
        const unsigned h1 = pack_split(c[v].z * rstd * gam[v].z, c[v].w * rstd * gam[v].w, l1);
        *reinterpret_cast<uint2*>(ytile + rm * YPITCH + v * 64 + rcol) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(ytile + rm * YPITCH + TC2 + v * 64 + rcol) = make_uint2(l0, l1);
      }
    }
    __syncthreads();  // tile rows visible; also: every wave has finished the previous tile's head sum (red is free again)
    // ---- projections: q^T with k^T, then v
    f32x16 st = zero16(), sk = zero16();
    {
      f32x16 qt, kt;
      project_qk(qt, kt);
      rotate1(qt);
      rotate1(kt);
      const uint4* eks = ekf + h * 4 * 64 + lane;
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 qh, ql, kh, kl;
        split8(qt, s * 8, qh, ql);
        split8(kt, s * 8, kh, kl);
        st = mfma3<ONE>(kh, kl, qh, ql, st);
        if (ntok) sk = mfma3<ONE>(eks[(s * 2) * 64], eks[(s * 2 + 1) * 64], qh, ql, sk);
      }
    }
    const f32x16 vt = project_v();
    // ---- softmax over this query's 8 (+8) frame keys and 8 (+8) tokens: lane half lk holds slots {0-3, 8-11} + 4 lk
    const f32x4 bz0 = *reinterpret_cast<const f32x4*>(bias_l), bz1 = *reinterpret_cast<const f32x4*>(bias_l + 4);
    const float bz[8] = {bz0.x, bz0.y, bz0.z, bz0.w, bz1.x, bz1.y, bz1.z, bz1.w};
    const unsigned pmask = pa ? 0xffffffffu : 0u;
    float f[8], g[8];
    float mx = -INFINITY;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      const float sv = __uint_as_float((__float_as_uint(st[8 + j]) & pmask) | (__float_as_uint(st[j]) & ~pmask));  // (bit blend: see the first kernel)
      f[j] = tk < T ? sv * a.q_scale + bz[j] : -INFINITY;
      g[j] = tk < ntok ? sk[j] * a.q_scale + (a.bias_on_cond ? bz[j] : 0.f) : -INFINITY;
      mx = fmaxf(mx, fmaxf(f[j], g[j]));
    }
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = __expf(f[j] - mx);
      g[j] = __expf(g[j] - mx);
      sum += f[j] + g[j];
    }
    sum += __shfl_xor(sum, 32, 64);
    const float inv = 1.0f / sum;
    float p0[8], p1[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] *= inv;
      g[j] *= inv;
      p0[j] = pa ? 0.f : f[j];  // keys of pixel 0 <-> k16 step 0
      p1[j] = pa ? f[j] : 0.f;  // keys of pixel 1 <-> k16 step 1
    }
    // first to_out fragments into the ring while the value mix runs
    uint4 rh[2][2], rl[2][2];  // [ring slot][step]
#pragma unroll
    for (int s = 0; s < 2; ++s) { rh[0][s] = wo_l[(s * 2) * 64]; rl[0][s] = wo_l[(s * 2 + 1) * 64]; }
    // ---- o^T = v^T . p^T (+ token values)
    f32x16 ot = zero16();
    {
      uint4 vh, vl, ph, pl;
      split8(vt, 0, vh, vl);
      split8v(p0, ph, pl);
      ot = mfma3<ONE>(vh, vl, ph, pl, ot);
      split8(vt, 8, vh, vl);
      split8v(p1, ph, pl);
      ot = mfma3<ONE>(vh, vl, ph, pl, ot);
      if (ntok) {
        const uint4* evs = evf + h * 2 * 64 + lane;
        split8v(g, ph, pl);
        ot = mfma3<ONE>(evs[0], evs[64], ph, pl, ot);
      }
    }
    uint4 oh[2], ol[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) split8(ot, s * 8, oh[s], ol[s]);
    // ---- this head's share of to_out, 64 output channels at a time through the head-sum buffer
    float* rb = red + (h * 32) * 64;
#pragma unroll
    for (int half = 0; half < NV; ++half) {
      if (half) __syncthreads();  // the previous half's sums have been read
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int ct = half * 2 + c2;
        if (ct + 1 < NCT) {
#pragma unroll
          for (int s = 0; s < 2; ++s) {
            rh[(ct + 1) & 1][s] = wo_l[(((ct + 1) * 16 + s) * 2) * 64];
            rl[(ct + 1) & 1][s] = wo_l[(((ct + 1) * 16 + s) * 2 + 1) * 64];
          }
        }
        f32x16 pc = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) pc = mfma3<ONE>(oh[s], ol[s], rh[ct & 1][s], rl[ct & 1][s], pc);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int m = (r & 3) + 8 * (r >> 2) + 4 * lk;
          rb[m * 64 + c2 * 32 + lrow] = pc[r];
        }
      }
      __syncthreads();
      if (rft < T) {
        f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int w = 0; w < HEADS; ++w) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * 32 + rm) * 64 + rcol);
          acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        const long long row = ((long long)b * T + rft) * HW + pp * 2 + rpa;
        acc.x += x_cur[half].x; acc.y += x_cur[half].y; acc.z += x_cur[half].z; acc.w += x_cur[half].w;  // residual: the element this thread normalised
        *reinterpret_cast<f32x4*>(a.out + row * a.ldo + half * 64 + rcol) = acc;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Second version: the same products, two tiles in flight, the two waves of a SIMD half a tile apart.
//
// The kernel above runs its eight head-waves in lock step (two workgroup barriers per tile): ~69 MFMAs (2.2k matrix-pipe cycles per
// wave) and ~550 vector instructions (2.2k VALU cycles per wave) alternate, both waves of a SIMD are in the same phase at the same time,
// and the matrix pipes idle whenever the rotary / split / softmax arithmetic runs: measured 0.33 matrix-pipe utilisation, padding included.
// Here every tile's work is cut in two halves of similar length,
//   F: LayerNorm rows -> q^T, k^T, v (36 MFMAs), rotary, q / k split into bf16 hi | lo, frame and token scores (12), logits
//   S: softmax, value mix, to_out share, partial sums to LDS (21 MFMAs), and the workgroup-level roles below
// and the heads of group B (the second wave of every SIMD) run one interval behind group A: while an A wave streams its projections a
// B wave does its softmax and vice versa.  One workgroup barrier per interval; what crosses it through LDS is double-buffered where the
// two groups' life times overlap:
//   interval n = 2j     LayerNorm(tile j) -> ytile[j & 1]           (group B's threads, next to their S half)
//              2j + 2   A: F(j)     2j + 3  A: S(j) -> redA[j & 1]   B: F(j)
//              2j + 4   B: S(j) -> redB                              2j + 5  head sum(j) + residual -> out    (group A's threads, next to S)
// Frame slots: SLOTS = 16 (two pixels per 32-row tile, T <= 16) or 32 (one pixel, T <= 32: the 22-frame configuration).
template <int SLOTS, int PARK, bool ONE, typename ST = float>
__global__ __launch_bounds__(512, 2) void temporal_block2_kernel(const TBArgs a) {
  const ST* const xin = reinterpret_cast<const ST*>(a.x);
  ST* const outp = reinterpret_cast<ST*>(a.out);
  constexpr int NP = 32 / SLOTS;  // pixels per tile
  constexpr int NK = SLOTS / 2;   // frame keys per lane
  constexpr int YPITCH = 2 * TC + 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  const int T = a.T, HW = a.HW;
  const int ntok = a.ek ? a.ntok : 0;
  const int Tp = (T + 3) & ~3;                                           // frame slots kept in the partial sums: whole groups of 4 accumulator rows
  const int rows = NP * Tp;
  float* red = reinterpret_cast<float*>(smem_raw);                       // [2 x 4 group-A heads | 4 group-B heads][rows][64]
  uint4* ekf = reinterpret_cast<uint4*>(red + 12 * rows * TC);           // [8 heads][2 steps][hi|lo][16 token rows x 2 halves]   (ntok > 0)
  uint4* evf = ekf + (ntok ? HEADS * 4 * 32 : 0);                        // [8 heads][hi|lo][64 lanes]
  float* biasf = reinterpret_cast<float*>(evf + (ntok ? HEADS * 2 * 64 : 0));  // [8 heads][2 halves][T frames][NK]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(biasf + HEADS * 2 * T * NK);  // [2][32 rows][hi 64 | lo 64 | pad 8]
  float* rotf = reinterpret_cast<float*>(ytile + 2 * 32 * YPITCH);       // [2 halves][T frames][8 pairs][cos, sin]
  float* gamf = rotf + 2 * T * 8 * 2;                                    // [64]
  uint4* wvp = reinterpret_cast<uint4*>(gamf + TC);                      // [PARK][8 heads][64 lanes]: the weight fragments that do not fit in registers

  constexpr float LOG2E = 1.4426950408889634f;
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = a.group_mode ? (h & 1) : (h >> 2), hidx = a.group_mode ? (h >> 1) : (h & 3);
  const int lrow = lane & 31, lk = lane >> 5;
  const int pa = SLOTS == 16 ? (lrow >> 4) : 0, ft = lrow & (SLOTS - 1);
  const int ftc = min(ft, T - 1);
  const int b = blockIdx.x / a.nsplit, split = blockIdx.x - b * a.nsplit;
  const int units = HW / NP;
  const int p_begin = split * a.tps, p_end = min(units, p_begin + a.tps);
  const int nt = p_end - p_begin;

  if (ntok) {
    uint4* eks = ekf + h * 4 * 32;
    uint4* evs = evf + h * 2 * 64;
    if (lrow < 16) {  // token rows 16 .. 31 of the A operand only feed score rows 16 .. 31, which nobody reads: those lanes re-read rows 0 .. 15
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float v[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = lrow < ntok ? a.ek[((long long)b * ntok + lrow) * HID + h * DHd + slot(s, lk, j)] : 0.f;
        uint4 hi, lo;
        split8v(v, hi, lo);
        eks[(s * 2 + 0) * 32 + lk * 16 + lrow] = hi;
        eks[(s * 2 + 1) * 32 + lk * 16 + lrow] = lo;
      }
    }
    float v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      v[j] = tk < ntok ? a.ev[((long long)b * ntok + tk) * HID + h * DHd + lrow] : 0.f;
    }
    uint4 hi, lo;
    split8v(v, hi, lo);
    evs[lane] = hi;
    evs[64 + lane] = lo;
  }
  for (int i = tid; i < HEADS * 2 * T * NK; i += 512) {
    const int jj = i % NK, r1 = i / NK, t = r1 % T, r2 = r1 / T, l2 = r2 & 1, hh = r2 >> 1;
    const int tk = slot(jj >> 3, l2, jj & 7);
    biasf[i] = tk < T ? a.bias[(hh * T + t) * T + tk] * LOG2E : -INFINITY;  // logits in base 2; a key frame that does not exist: -inf
  }
  for (int i = tid; i < 2 * T * 8; i += 512) {
    const int pr = i & 7, r1 = i >> 3, t = r1 % T, l2 = r1 / T;
    const int d = slot(pr >> 2, l2, (2 * pr) & 7);
    const float2 cs = *reinterpret_cast<const float2*>(a.rot + (t * 16 + (d >> 1)) * 2);
    rotf[i * 2] = cs.x;
    rotf[i * 2 + 1] = cs.y;
  }

  // q / k / v weight fragments of this head: 24 - PARK stay in registers, PARK (1 or, LDS permitting, 3) are re-read from LDS every tile.
  // With all 24 in registers the compiler spills to scratch, and a scratch reload is a vmcnt(0) wait in the middle of the projections
  // -- behind the next tile's rows that were requested just before.
  constexpr int NREG = 4 - PARK;
  uint4 wqh[4], wql[4], wkh[4], wkl[4], wvh[4], wvl[NREG];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4* q = a.wqkv + (((long long)h * 4 + s) * 2) * 64 + lane;
    const uint4* k = a.wqkv + (((long long)(HEADS + h) * 4 + s) * 2) * 64 + lane;
    const uint4* v = a.wqkv + (((long long)(2 * HEADS + h) * 4 + s) * 2) * 64 + lane;
    wqh[s] = q[0]; wql[s] = q[64];
    wkh[s] = k[0]; wkl[s] = k[64];
    wvh[s] = v[0];
    if (s < NREG) wvl[s < NREG ? s : 0] = v[64];
    else wvp[(s - NREG) * 512 + tid] = v[64];
  }
  const uint4* wo_l = a.wout + ((long long)2 * h * 2) * 64 + lane;
  const uint4* eks_l = ekf + h * 4 * 32 + lk * 16 + (lrow & 15);
  const float* rot_l = rotf + ((lk * T + ftc) * 8) * 2;
  const float* bias_l = biasf + ((h * 2 + lk) * T + ftc) * NK;
  float* red_mine = red + ((grp ? 8 : 0) + hidx) * rows * TC;   // group A: + (j & 1) * 4 * rows * TC

  // Roles besides the head's matrix work.  They belong to the group that is in its (shorter) S half: group B's 256 threads normalise
  // the next tile's rows in the even intervals, group A's 256 threads sum the heads' shares of a finished tile in the odd ones.
  // Thread gt of its group: rows rm0 = gt >> 4 and rm0 + 16 of the tile, channels rcol .. rcol + 3.
  const int gt = hidx * 64 + lane;
  const int rm0 = gt >> 4, rcol = (gt & 15) * 4;
  if (tid < TC) gamf[tid] = a.gamma[tid];
  // (SLOTS == 16: the two rows are frame slot rm0 of the tile's two pixels; SLOTS == 32: frame slots rm0 and rm0 + 16 of its one pixel)
  const int rft[2] = {rm0, SLOTS == 16 ? rm0 : rm0 + 16};
  const int rpx[2] = {0, SLOTS == 16 ? 1 : 0};
  auto x_row = [&](int j, int i) { return (((long long)b * T + rft[i]) * HW + (long long)(p_begin + j) * NP + rpx[i]); };
  // Always two loads, no branch around them (frame slots / tiles that do not exist re-read one that does; the LayerNorm zeroes them):
  // with a lane-dependent branch around a load the compiler stops counting and waits for vmcnt(0) at the first use of ANY loaded
  // value, i.e. for a full HBM round trip of whatever was requested last.
  const int rfc[2] = {min(rft[0], T - 1), min(rft[1], T - 1)};
  using XRaw = decltype(ldraw4(xin));  // raw bits of four elements (bf16-stored maps: unpacked where they are USED, not where they are requested)
  auto load_x = [&](int j, XRaw (&v)[2]) {
    const int jc = min(max(j, 0), nt - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i)
      v[i] = ldraw4(xin + (((long long)b * T + rfc[i]) * HW + (long long)(p_begin + jc) * NP + rpx[i]) * a.ldx + rcol);
  };
  // sum over the 16 lanes of a row, every lane gets it: four rotate-and-add steps inside the DPP row (no LDS permutes, no waits)
  auto row_sum16 = [](float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));  // row_ror:8
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));  // row_ror:4
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));  // row_ror:2
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));  // row_ror:1
    return v;
  };

  // the value lane ^ 32 holds (v_permlane32_swap: no LDS permute, no wait)
  auto other_half = [](float v) {
    const unsigned u = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false);
    return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]);
  };

  // state a head carries from its F half to its S half: the logits of its query against its frame keys and the tokens, v
  f32x16 vt = zero16();
  float f[NK], g[8];
#pragma unroll
  for (int jj = 0; jj < NK; ++jj) f[jj] = 0.f;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) g[jj] = 0.f;
  XRaw xv[2];
  load_x(grp ? 0 : -1, xv);
  __syncthreads();

  // (compiled in by -DVMM_TB_TRACE_BUILD=1 only -- tools/build_ab.py temporal_block -DVMM_TB_TRACE_BUILD=1 + VMM_LIB_PATH; see conv3x3_bf16x3.hip)
#ifndef VMM_TB_TRACE_BUILD
#define VMM_TB_TRACE_BUILD 0
#endif
  const bool tracing = VMM_TB_TRACE_BUILD && a.trace && blockIdx.x == 0;
  auto stamp = [&](int n, int k) {
    if constexpr (VMM_TB_TRACE_BUILD)
      if (tracing && n < TB_TRACE_N && lane == 0) a.trace[(n * 8 + h) * 4 + k] = __builtin_amdgcn_s_memtime();
  };
  for (int n = 0; n < 2 * nt + 4; ++n) {
    stamp(n, 0);
    if (!(n & 1)) {
      const int j = n >> 1;
      if (grp) {
        if (j < nt) {  // LayerNorm of tile j: 16 lanes per row, rows to LDS as bf16 hi | lo (zero rows for the empty frame slots)
          const f32x4 gam = *reinterpret_cast<const f32x4*>(gamf + rcol);
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const f32x4 z4 = {0.f, 0.f, 0.f, 0.f};
            const f32x4 x = rft[i] < T ? unpack4(xv[i]) : z4;
            const float mean = row_sum16((x.x + x.y) + (x.z + x.w)) * (1.0f / TC);
            const f32x4 c = {x.x - mean, x.y - mean, x.z - mean, x.w - mean};
            const float rstd = __builtin_amdgcn_rsqf(row_sum16((c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w)) * (1.0f / TC) + a.eps);  // 1 ulp
            unsigned l0, l1;
            const unsigned h0 = pack_split(c.x * rstd * gam.x, c.y * rstd * gam.y, l0);
            const unsigned h1 = pack_split(c.z * rstd * gam.z, c.w * rstd * gam.w, l1);
            unsigned short* yt = ytile + (j & 1) * 32 * YPITCH + (rm0 + 16 * i) * YPITCH;
            *reinterpret_cast<uint2*>(yt + rcol) = make_uint2(h0, h1);
            *reinterpret_cast<uint2*>(yt + TC + rcol) = make_uint2(l0, l1);
          }
        }
      }
    } else if (!grp) {
      const int j = (n - 5) >> 1;
      if (n >= 5 && j < nt) {  // head sum of tile j, heads in order 0 .. 7 (group_mode 0), + residual
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (rft[i] < T) {
            const int rs = rpx[i] * Tp + rft[i];
            const float* ra = red + (((j & 1) * 4) * rows + rs) * TC + rcol;
            const float* rb = red + (8 * rows + rs) * TC + rcol;
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(ra + w * rows * TC);
              acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
#pragma unroll
            for (int w = 0; w < 4; ++w) {
              const f32x4 v = *reinterpret_cast<const f32x4*>(rb + w * rows * TC);
              acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
            }
            const f32x4 xr = unpack4(xv[i]);
            acc.x += xr.x; acc.y += xr.y; acc.z += xr.z; acc.w += xr.w;
            st4(outp + x_row(j, i) * a.ldo + rcol, acc);
          }
        }
      }
    }
    stamp(n, 1);

    // ---- the head's half tile
    const int m = n - 2 - grp;
    const int j = m >> 1;
    if (m >= 0 && j < nt) {
      if (!(m & 1)) {
        // F: projections, rotary, frame scores (keys x queries)
        const unsigned short* yt = ytile + (j & 1) * 32 * YPITCH + lrow * YPITCH + lk * 8;
        f32x16 qt = zero16(), kt = zero16();
        vt = zero16();
#pragma unroll
        for (int s = 0; s < 4; ++s) {
          const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
          qt = mfma3<ONE>(wqh[s], wql[s], yh, yl, qt);  // [d][m]
          kt = mfma3<ONE>(wkh[s], wkl[s], yh, yl, kt);  // [d][m]
          vt = mfma3<ONE>(yh, yl, wvh[s], s < NREG ? wvl[s < NREG ? s : 0] : wvp[(s < NREG ? 0 : s - NREG) * 512 + tid], vt);  // [m][d]
        }
#pragma unroll
        for (int i4 = 0; i4 < 4; ++i4) {
          const f32x4 cs = *reinterpret_cast<const f32x4*>(rot_l + i4 * 4);
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            const int i = 2 * i4 + k;
            const float c = k ? cs.z : cs.x, sn = k ? cs.w : cs.y;
            float e = qt[2 * i], o = qt[2 * i + 1];
            qt[2 * i] = e * c - o * sn;
            qt[2 * i + 1] = o * c + e * sn;
            e = kt[2 * i]; o = kt[2 * i + 1];
            kt[2 * i] = e * c - o * sn;
            kt[2 * i + 1] = o * c + e * sn;
          }
        }
        uint4 qh[2], ql[2];
        split8(qt, 0, qh[0], ql[0]);
        split8(qt, 8, qh[1], ql[1]);
        f32x16 st = zero16(), sk = zero16();
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          uint4 kh, kl;
          split8(kt, s * 8, kh, kl);
          st = mfma3<ONE>(kh, kl, qh[s], ql[s], st);
          if (ntok) sk = mfma3<ONE>(eks_l[(s * 2) * 32], eks_l[(s * 2 + 1) * 32], qh[s], ql[s], sk);
        }
        // logits of this lane's query: 8 (16) frame keys and 8 tokens per lane half, relative-position bias from LDS
        float bz[NK];
#pragma unroll
        for (int i = 0; i < NK / 4; ++i) {
          const f32x4 v = *reinterpret_cast<const f32x4*>(bias_l + i * 4);
          bz[4 * i] = v.x; bz[4 * i + 1] = v.y; bz[4 * i + 2] = v.z; bz[4 * i + 3] = v.w;
        }
        const unsigned pmask = pa ? 0xffffffffu : 0u;
        const float qs2 = a.q_scale * LOG2E;
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) {
          // (bit blend, not a select between st[8 + jj] and st[jj]: see the first kernel)
          const float sv = SLOTS == 16 ? __uint_as_float((__float_as_uint(st[(8 + jj) & 15]) & pmask) | (__float_as_uint(st[jj]) & ~pmask)) : st[jj];
          f[jj] = sv * qs2 + bz[jj];  // (scores of the empty frame slots are finite -- their k rows are zero -- and their table entry is -inf)
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          const int tk = slot(0, lk, jj);
          g[jj] = tk < ntok ? sk[jj] * qs2 + (a.bias_on_cond ? bz[jj] : 0.f) : -INFINITY;
        }
      } else {
        // S: softmax, value mix, this head's share of to_out
        // to_out fragments of this head (8 KB, L2-resident): requested a softmax / value phase before their use
        uint4 woh[2][2], wol[2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) {
            const uint4* q = wo_l + ((ct * 16 + s) * 2) * 64;
            woh[ct][s] = q[0];
            wol[ct][s] = q[64];
          }
        __builtin_amdgcn_sched_barrier(0);
        float mx = -INFINITY;
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) mx = fmaxf(mx, f[jj]);
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) mx = fmaxf(mx, g[jj]);
        mx = fmaxf(mx, other_half(mx));
        float sum = 0.f;
#pragma unroll
        for (int jj = 0; jj < NK; ++jj) {
          f[jj] = __builtin_amdgcn_exp2f(f[jj] - mx);
          sum += f[jj];
        }
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
          g[jj] = __builtin_amdgcn_exp2f(g[jj] - mx);
          sum += g[jj];
        }
        sum += other_half(sum);
        const float inv = __builtin_amdgcn_rcpf(sum);
        const float inv0 = (SLOTS == 16 && pa) ? 0.f : inv, inv1 = (SLOTS == 16 && !pa) ? 0.f : inv;
        f32x16 ot = zero16();
        {
          uint4 vh, vl, ph, pl;
          float p0[8], p1[8];
#pragma unroll
          for (int jj = 0; jj < 8; ++jj) {
            p0[jj] = f[jj] * inv0;                    // SLOTS == 16: keys of pixel 0 <-> k16 step 0, zero for the queries of pixel 1
            p1[jj] = f[(8 + jj) & (NK - 1)] * inv1;  //              keys of pixel 1 <-> k16 step 1, zero for the queries of pixel 0
            g[jj] *= inv;
          }
          split8(vt, 0, vh, vl);
          split8v(p0, ph, pl);
          ot = mfma3<ONE>(vh, vl, ph, pl, ot);
          split8(vt, 8, vh, vl);
          split8v(p1, ph, pl);
          ot = mfma3<ONE>(vh, vl, ph, pl, ot);
          if (ntok) {
            const uint4* evs = evf + h * 2 * 64 + lane;
            split8v(g, ph, pl);
            ot = mfma3<ONE>(evs[0], evs[64], ph, pl, ot);
          }
        }
        f32x16 pc[2] = {zero16(), zero16()};
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          uint4 oh, ol;
          split8(ot, s * 8, oh, ol);
#pragma unroll
          for (int ct = 0; ct < 2; ++ct) pc[ct] = mfma3<ONE>(oh, ol, woh[ct][s], wol[ct][s], pc[ct]);
        }
        // accumulator rows r of a lane half: frame slots 8 (r >> 2) + 4 lk + (r & 3) -- whole groups of four exist or do not (Tp)
        float* rb0 = red_mine + (grp ? 0 : (j & 1) * 4 * rows * TC) + (4 * lk) * TC + lrow;
        float* rb1 = rb0 + Tp * TC;  // second pixel of the tile (SLOTS == 16)
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
          const int fbase = SLOTS == 16 ? 8 * (g4 & 1) : 8 * g4;
          float* q = (SLOTS == 16 && g4 >= 2) ? rb1 : rb0;
          if (fbase + 4 * lk < Tp) {
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
              q[(fbase + r4) * TC] = pc[0][g4 * 4 + r4];
              q[(fbase + r4) * TC + 32] = pc[1][g4 * 4 + r4];
            }
          }
        }
      }
    }
    // The role's next input, requested at the END of the interval: the group's next interval is an F half (no global memory waits in
    // it), so the rows have a whole interval to arrive and are not live across this interval's S half.
    __builtin_amdgcn_sched_barrier(0);
    if (grp == ((n & 1) ^ 1)) load_x(grp ? (n >> 1) + 1 : ((n - 5) >> 1) + 1, xv);  // B: LayerNorm(n / 2 + 1) at n + 2; A: head sum at n + 2
    stamp(n, 2);
    __syncthreads();
    stamp(n, 3);
  }
}

size_t tb2_lds_bytes(int slots, int T, int ntok, int park) {
  const int np = 32 / slots, nk = slots / 2, rows = np * ((T + 3) & ~3);
  return sizeof(float) * 12 * rows * TC + (ntok ? sizeof(uint4) * (HEADS * 4 * 32 + HEADS * 2 * 64) : 0) + sizeof(float) * HEADS * 2 * T * nk +
         sizeof(unsigned short) * 2 * 32 * (2 * TC + 8) + sizeof(float) * (2 * T * 8 * 2 + TC) + sizeof(uint4) * 512 * park;
}

int tb_version() {  // read at every call (a host-side getenv per plan build / launch): the tests switch kernels inside one process
  const char* e = getenv("VMM_TB_VERSION");
  return e ? atoi(e) : 2;
}

}  // namespace

// Weights: wqkv_frag = vmm_pack_weights fmt 2 of to_qkv (768, 64), wout_frag = fmt 3 of to_out (64, 256).
// Returns 1 (nothing launched) when the shape is outside the envelope: C == 64, heads == 8, dim_head == 32, ntok <= 16, and T <= 16 with an
// even HW (two pixels per tile) or T <= 32 (one pixel per tile, second kernel only, LDS permitting -- vmm_temporal_block_supported); or C == 128
// with T <= 16 and an even HW (wqkv_frag then = fmt 2 of (768, 128), wout_frag = fmt 3 of (128, 256)).
// 0: outside the envelope of all kernels; 1: first kernel only (T <= 16); 2: two-tiles-in-flight kernel (LDS permitting, T <= 32); 3: C = 128
#if VMM_FP16_OPERANDS  // (the fp16-operand build of this file: the host query exists once, in the split-bf16 build; a private copy here)
#define vmm_temporal_block_supported vmm_temporal_block_supported_fp16_tu
static int vmm_temporal_block_supported(int32_t T, int32_t ntok, int32_t HW, int32_t C, int32_t heads) {
#else
extern "C" int vmm_temporal_block_supported(int32_t T, int32_t ntok, int32_t HW, int32_t C, int32_t heads) {
#endif
  if (C == TC2 && heads == HEADS && T >= 1 && T <= 16 && ntok >= 0 && ntok <= 16 && !(HW & 1)) return 3;  // the C = 128 kernel (streamed weights)
  if (C != TC || heads != HEADS || T < 1 || T > 32 || ntok < 0 || ntok > 16) return 0;
  const int slots = T <= 16 ? 16 : 32;
  if (slots == 16 && (HW & 1)) return 0;
  if (tb_version() >= 2 && tb2_lds_bytes(slots, T, ntok, 1) <= 160 * 1024) return 2;
  return T <= 16 ? 1 : 0;
}

template <bool ONE, typename ST = float>
static int tb_launch(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag, const float* ek, const float* ev,
                     int32_t ntok, const float* bias, int32_t bias_on_cond, const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW,
                     int32_t C, int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  if ((ldx & 3) || (ldo & 3)) return 1;
  const int kind = vmm_temporal_block_supported(T, ek ? ntok : 0, HW, C, heads);
  if (kind == 0) return 1;
  if (!std::is_same<ST, float>::value && kind != 2) return 1;  // bf16-stored maps: the two-tiles-in-flight kernel (C = 64)
  if (bias_on_cond && ek && ntok != T) return -2;
  if (B <= 0) return 0;
  TBArgs a;
  a.x = x; a.ldx = ldx; a.gamma = gamma;
  a.wqkv = reinterpret_cast<const uint4*>(wqkv_frag);
  a.wout = reinterpret_cast<const uint4*>(wout_frag);
  a.ek = ek; a.ev = ev; a.ntok = ek ? ntok : 0;
  a.bias = bias; a.bias_on_cond = bias_on_cond; a.rot = rot_tab;
  a.out = out; a.ldo = ldo; a.T = T; a.HW = HW; a.q_scale = q_scale; a.eps = eps;
  a.group_mode = getenv("VMM_TB_GROUP") ? atoi(getenv("VMM_TB_GROUP")) : 1;  // odd heads late: measured 2-4 % ahead of "heads 4-7 late"
  a.trace = nullptr;
  const bool want_trace = getenv("VMM_TB_TRACE") && kind == 2;
  if (want_trace) {
    hipMalloc(reinterpret_cast<void**>(&a.trace), sizeof(unsigned long long) * TB_TRACE_N * 8 * 4);
    hipMemset(a.trace, 0, sizeof(unsigned long long) * TB_TRACE_N * 8 * 4);
  }
  if (kind == 3) {  // C = 128: one 512-thread workgroup per CU, one round
    const int units = HW / 2;
    const int ns = max(1, min(units, 256 / B));
    a.tps = (units + ns - 1) / ns;
    a.nsplit = (units + a.tps - 1) / a.tps;
    static bool attr128 = false;
    if (!attr128) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block128_kernel<ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr128 = true;
    }
    const size_t shm = sizeof(float) * HEADS * 32 * 64 + sizeof(uint4) * HEADS * 6 * 64 + sizeof(float) * HEADS * 2 * 16 * 8 +
                       sizeof(unsigned short) * 32 * (2 * TC2 + 8) + sizeof(float) * 2 * 16 * 8 * 2;
    hipLaunchKernelGGL(temporal_block128_kernel<ONE>, dim3((unsigned)(B * a.nsplit)), dim3(512), shm, (hipStream_t)stream, a);
    VMM_LAUNCH_CHECK();
    return 0;
  }
  const int slots = (kind == 2 && T > 16) ? 32 : 16;
  const int units = HW / (32 / slots);
  int ns = max(1, min(units, 256 / B));  // one 512-thread workgroup per CU (LDS), one round of workgroups
  a.tps = (units + ns - 1) / ns;
  a.nsplit = (units + a.tps - 1) / a.tps;
  static bool attr_set = false;  // (one flag per instantiation of this launcher)
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel<ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block2_kernel<16, 1, ONE, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block2_kernel<32, 1, ONE, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block2_kernel<16, 3, ONE, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block2_kernel<32, 3, ONE, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const dim3 grid((unsigned)(B * a.nsplit));
  if (kind == 2) {
    const int park = tb2_lds_bytes(slots, T, a.ntok, 3) <= 160 * 1024 ? 3 : 1;
    const size_t shm = tb2_lds_bytes(slots, T, a.ntok, park);
    if (slots == 16 && park == 3) hipLaunchKernelGGL((temporal_block2_kernel<16, 3, ONE, ST>), grid, dim3(512), shm, (hipStream_t)stream, a);
    else if (slots == 16) hipLaunchKernelGGL((temporal_block2_kernel<16, 1, ONE, ST>), grid, dim3(512), shm, (hipStream_t)stream, a);
    else if (park == 3) hipLaunchKernelGGL((temporal_block2_kernel<32, 3, ONE, ST>), grid, dim3(512), shm, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((temporal_block2_kernel<32, 1, ONE, ST>), grid, dim3(512), shm, (hipStream_t)stream, a);
  } else {
    const size_t shm = sizeof(float) * HEADS * 32 * TC + sizeof(uint4) * HEADS * 6 * 64 + sizeof(float) * HEADS * 2 * 16 * 8 +
                       sizeof(unsigned short) * 32 * (2 * TC + 8) + sizeof(float) * 2 * 16 * 8 * 2;
    hipLaunchKernelGGL(temporal_block_kernel<ONE>, grid, dim3(512), shm, (hipStream_t)stream, a);
  }
  VMM_LAUNCH_CHECK();
  if (want_trace) {  // debugging aid: time line of workgroup 0 (ticks of s_memtime relative to the first stamp)
    static unsigned long long hbuf[TB_TRACE_N * 8 * 4];
    hipStreamSynchronize((hipStream_t)stream);
    hipMemcpy(hbuf, a.trace, sizeof(hbuf), hipMemcpyDeviceToHost);
    hipFree(a.trace);
    unsigned long long t0 = ~0ull;
    for (unsigned long long v : hbuf) if (v && v < t0) t0 = v;
    fprintf(stderr, "temporal_block2 trace (T=%d HW=%d ntok=%d): per interval and wave: start | role | half | barrier wait (ticks)\n", T, HW, a.ntok);
    for (int n = 0; n < TB_TRACE_N; ++n) {
      if (!hbuf[(n * 8) * 4]) break;
      fprintf(stderr, "n=%2d @%7llu:", n, hbuf[(n * 8) * 4] - t0);
      for (int w = 0; w < 8; ++w) {
        const unsigned long long* q = hbuf + (n * 8 + w) * 4;
        fprintf(stderr, "  w%d %4llu/%5llu/%5llu", w, q[1] - q[0], q[2] - q[1], q[3] - q[2]);
      }
      fprintf(stderr, "\n");
    }
  }
  return 0;
}

#if !VMM_FP16_OPERANDS
extern "C" int vmm_temporal_block_bf16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                         const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                         const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                         int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  return tb_launch<false>(x, ldx, gamma, wqkv_frag, wout_frag, ek, ev, ntok, bias, bias_on_cond, rot_tab, out, ldo, B, T, HW, C, heads, q_scale, eps, stream);
}
// the "bf16" throughput mode of the same block (BASELINE.json configs[3]): identical arguments and packed weights, one matrix pass per product on
// the operands' bf16 roundings; LayerNorm, softmax, rotary and the residual stay fp32
extern "C" int vmm_temporal_block_bf16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                       const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                       const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                       int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  return tb_launch<true>(x, ldx, gamma, wqkv_frag, wout_frag, ek, ev, ntok, bias, bias_on_cond, rot_tab, out, ldo, B, T, HW, C, heads, q_scale, eps, stream);
}
// ... over bf16-STORED feature maps (x, out = bf16 bits; ld in elements): the "bf16" mode's C = 64 level (the two-tiles-in-flight kernel)
extern "C" int vmm_temporal_block_bf16_a16(const void* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                           const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                           const float* rot_tab, void* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                           int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  return tb_launch<true, bf16s>(static_cast<const float*>(x), ldx, gamma, wqkv_frag, wout_frag, ek, ev, ntok, bias, bias_on_cond, rot_tab, static_cast<float*>(out), ldo,
                                B, T, HW, C, heads, q_scale, eps, stream);
}
#elif VMM_SPLIT_F16
// Three passes on IEEE-half hi | lo operands (vmm_common.h, VMM_SPLIT_F16: the split is four vector instructions per pair instead of six): the sampler's
// fp32-class block; identical arguments, weights = vmm_pack_weights fmt 2 | 32 / 3 | 32 (fp16 hi | fp16 lo planes)
extern "C" int vmm_temporal_block_f16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                        const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                        const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                        int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  return tb_launch<false>(x, ldx, gamma, wqkv_frag, wout_frag, ek, ev, ntok, bias, bias_on_cond, rot_tab, out, ldo, B, T, HW, C, heads, q_scale, eps, stream);
}
#else
// fp16 operands (`train_precision = "fp16"`: the reference's autocast dtype, main.py:34): the single-pass instance of this translation unit compiled with
// -DVMM_SINGLE_PASS=2; identical arguments, weights = vmm_pack_weights fmt 2 | 16 / 3 | 16 (fp16 planes)
extern "C" int vmm_temporal_block_fp16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                       const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                       const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                       int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  return tb_launch<true>(x, ldx, gamma, wqkv_frag, wout_frag, ek, ev, ntok, bias, bias_on_cond, rot_tab, out, ldo, B, T, HW, C, heads, q_scale, eps, stream);
}
#endif
