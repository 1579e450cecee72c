// Fused spatial linear-attention BLOCK for the two upper levels (C = 64 or 128, 8 heads x 32), split-bf16 MFMA, gfx950.
//
//   out = x + to_out( ctx^T . softmax_d(q) * scale ) + bias,   ctx = softmax_n([k_tok | k]) . ([v_tok | v] / HW)^T,   q,k,v = to_qkv(LayerNorm(x))
//   (vddp.py:313-378 SpatialLinearAttention inside Residual(PreNorm(.)), vddp.py:613/628)
//
// The unfused path writes the 768-wide qkv rows (2.5 GB per site at batch 8) and reads them back twice.  Here q, k, v, the
// softmax numerators and the per-head outputs never leave the registers of the wave that produced them:
//   pass A  (linattn_ctx_kernel)      x -> LayerNorm -> k, v tiles (MFMA) -> online softmax over pixels -> ctx^T += v^T . p (MFMA);
//                                     one partial (max, sum, 32x32 ctx^T) per (frame, split, head)
//   combine (linattn_combine_kernel)  merges the partials with the conditioning-token keys/values and writes ctx^T * scale /
//                                     (sum * HW) as ready-made MFMA operand fragments
//   pass B  (linattn_apply_kernel)    x -> LayerNorm -> q^T tile (MFMA) -> softmax over d -> o^T = ctx^T . q (MFMA) ->
//                                     partial to_out (MFMA) -> 8-head sum through LDS + bias + residual -> out
// HBM traffic per site: x read twice (+ once more from L2 for the residual), out written once.
//
// One wave = one head (512-thread workgroups): its weight fragments stay in registers for the whole kernel (one barrier per 32-pixel tile in
// pass A for the shared LayerNorm tile, two in pass B: tile and head sum).  The trick that removes every transposition: an MFMA 32x32
// accumulator holds, per lane, ONE column (lane & 31) and 16 rows; that is exactly an A or B operand of the next MFMA whose
// contraction runs over those rows, provided both operands enumerate the contraction index in the same (register) order:
//   k16 step s, lane half lk, element j  <->  row (j & 3) + 8 * (2 s + (j >> 2)) + 4 lk.
// Operands produced in-kernel (p, v, q, o) have that order by construction; the static ones that meet them (ctx^T fragments,
// to_out weights) are written in that order by the combine kernel / vmm_pack_weights fmt 3.
#include "igemm_common.h"
#include "linattn_split.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// CC = channels of the level: 64 (full resolution; every weight fragment of a head stays in its wave's registers) or 128 (the next level: the
// q / k / v fragments still do -- 128 registers in pass A, 64 in pass B --, the to_out fragments of pass B are streamed from L2 per tile through a
// two-plane ring, the head sum goes through LDS in two 64-channel halves of the [8][32][128] buffer's worth).
constexpr int LH = 8;             // heads (= waves per workgroup)
constexpr int LD = 32;            // dim_head
constexpr int PART = 64 + LD * LD;  // floats per partial: max[32] | sum[32] | ctx^T[e][d]

struct LAArgs {
  const void* x; int ldx;   // rows of ST: float, or bf16s when the maps are bf16-stored (the "bf16" mode; A16 instances of the two passes)
  const float* gamma; float eps;
  const uint4* wqkv;   // fmt 2 fragments, N = 768 (q | k | v), K = 64 (4 k16 steps)
  const uint4* wout;   // fmt 3 fragments, N = 64, K = 256 (16 k16 steps)
  const float* bias_out;
  const float* ek; const float* ev; int ntok;
  float* part;         // [frame][split][head][PART]
  uint4* ctxfrag;      // [frame][head][2 steps][hi|lo][64 lanes]
  void* out; int ldo;
  int T, HW, nsplit, sps;  // sps: 32-pixel tiles per split
  float q_scale;
};

__device__ __forceinline__ unsigned pack_split(float a, float b, unsigned& lo) { return split_bf16_pair(a, b, lo); }

// 8 consecutive accumulator registers -> one k16 operand fragment (hi, lo)
__device__ __forceinline__ void split8(const f32x16& c, int r0, uint4& hi, uint4& lo) {
  hi.x = pack_split(c[r0 + 0], c[r0 + 1], lo.x);
  hi.y = pack_split(c[r0 + 2], c[r0 + 3], lo.y);
  hi.z = pack_split(c[r0 + 4], c[r0 + 5], lo.z);
  hi.w = pack_split(c[r0 + 6], c[r0 + 7], lo.w);
}

// ONE ("bf16" throughput mode, BASELINE.json configs[3]): the hi planes only, one pass (the lo halves of the splits then have no reader)
template <bool ONE>
__device__ __forceinline__ f32x16 mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 c) {
  if constexpr (ONE) return vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), c);
  c = vmm_mfma16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c);
  return c;
}

// The 32 x CC input tile of a step is read ONCE per workgroup: thread (row tid >> 4, channels (tid & 15) * 4 .. +3 of every 64-channel group) loads
// its float4s a tile ahead (the HBM latency hides under the previous tile's MFMAs), the 16 lanes of a row normalise it (channel LayerNorm,
// vddp.py:245-254) and the bf16 hi | lo rows go to LDS, where the eight head-waves pick up their operand fragments: lane
// (pixel = lane & 31, lk) owns channels s*16 + lk*8 .. +7 for the CC / 16 k16 steps s.  The same registers serve as an A operand
// (rows = pixels) or a B operand (columns = pixels).
template <int CC> struct LAGeom {
  static constexpr int NV = CC / 64;          // float4s per staging thread
  static constexpr int NS = CC / 16;          // k16 steps of a projection from the channels
  static constexpr int YPITCH = 2 * CC + 8;   // bf16 per LDS row: hi CC | lo CC | pad (272 / 528 bytes = 17 / 33 x 16: conflict-free ds_read_b128)
};

template <int CC>
__device__ __forceinline__ void stage_norm_row(const LAArgs& a, const f32x4 (&x)[CC / 64], const f32x4 (&gam)[CC / 64], unsigned short* ytile, int tid) {
  // (row sums inside the DPP row, v_rsq_f32: the eight ds_bpermute round trips and the IEEE sqrt / divide were ~1k cycles of every tile,
  // in front of the barrier all eight head-waves wait at)
  constexpr int NV = CC / 64, YP = LAGeom<CC>::YPITCH;
  float s1 = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) s1 += (x[v].x + x[v].y) + (x[v].z + x[v].w);
  const float mean = row_sum16(s1) * (1.0f / CC);
  f32x4 c[NV];
  float s2 = 0.f;
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    c[v] = f32x4{x[v].x - mean, x[v].y - mean, x[v].z - mean, x[v].w - mean};
    s2 += (c[v].x * c[v].x + c[v].y * c[v].y) + (c[v].z * c[v].z + c[v].w * c[v].w);
  }
  const float rstd = __builtin_amdgcn_rsqf(row_sum16(s2) * (1.0f / CC) + a.eps);
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    unsigned l0, l1;
    const unsigned h0 = pack_split(c[v].x * rstd * gam[v].x, c[v].y * rstd * gam[v].y, l0);
    const unsigned h1 = pack_split(c[v].z * rstd * gam[v].z, c[v].w * rstd * gam[v].w, l1);
    unsigned short* row = ytile + (tid >> 4) * YP + v * 64 + (tid & 15) * 4;
    *reinterpret_cast<uint2*>(row) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(row + CC) = make_uint2(l0, l1);
  }
}

// four k16 steps s0 .. s0 + 3 of the row's fragments
template <int CC>
__device__ __forceinline__ void read_row_frags(const unsigned short* ytile, int lrow, int lk, int s0, uint4 (&yh)[4], uint4 (&yl)[4]) {
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const unsigned short* q = ytile + lrow * LAGeom<CC>::YPITCH + (s0 + i) * 16 + lk * 8;
    yh[i] = *reinterpret_cast<const uint4*>(q);
    yl[i] = *reinterpret_cast<const uint4*>(q + CC);
  }
}

__device__ __forceinline__ f32x16 zero16() {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  return c;
}

// ---------------------------------------------------------------------------------------------------------------- pass A
template <bool ONE, int CC, typename ST = float>
__global__ __launch_bounds__(512) void linattn_ctx_kernel(const LAArgs a) {
  constexpr int NV = CC / 64, NS = CC / 16, YP = LAGeom<CC>::YPITCH;
  const ST* const xin = static_cast<const ST*>(a.x);
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int frame = blockIdx.x / a.nsplit, split = blockIdx.x - frame * a.nsplit;
  const int tiles = a.HW / 32;
  const int t_begin = split * a.sps, t_end = min(tiles, t_begin + a.sps);

  uint4 wkh[NS], wkl[NS], wvh[NS], wvl[NS];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const uint4* qk = a.wqkv + (((long long)(LH + h) * NS + s) * 2) * 64 + lane;
    const uint4* qv = a.wqkv + (((long long)(2 * LH + h) * NS + s) * 2) * 64 + lane;
    wkh[s] = qk[0]; wkl[s] = qk[64];
    wvh[s] = qv[0]; wvl[s] = qv[64];
  }
  __shared__ __attribute__((aligned(16))) unsigned short ytile[2][32 * YP];  // double buffer: one barrier per tile
  f32x4 gam[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) gam[v] = *reinterpret_cast<const f32x4*>(a.gamma + v * 64 + (tid & 15) * 4);
  // (raw bits, unpacked where they are used: a prefetch unpacked where it is requested waits for its HBM round trip in front of the barrier)
  using XRaw = decltype(ldraw4(xin));
  auto load_x = [&](int t, XRaw (&d)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      d[v] = XRaw{};
      if (t < t_end) d[v] = ldraw4(xin + ((long long)frame * a.HW + t * 32 + (tid >> 4)) * a.ldx + v * 64 + (tid & 15) * 4);
    }
  };
  auto unpack = [&](const XRaw (&r)[NV], f32x4 (&d)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) d[v] = unpack4(r[v]);
  };
  float m = -INFINITY, ssum = 0.f;
  f32x16 ctx = zero16();  // rows e, column d = lrow
  XRaw x_next[NV];
  load_x(t_begin, x_next);
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    f32x4 x_cur[NV];
    unpack(x_next, x_cur);
    load_x(t + 1, x_next);
    stage_norm_row<CC>(a, x_cur, gam, ytile[buf], tid);
    __syncthreads();
    f32x16 kt = zero16(), vt = zero16();  // rows pixels, column d (resp. e) = lrow
#pragma unroll
    for (int g = 0; g < NS / 4; ++g) {
      uint4 yh[4], yl[4];
      read_row_frags<CC>(ytile[buf], lrow, lk, 4 * g, yh, yl);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        kt = mfma3<ONE>(yh[i], yl[i], wkh[4 * g + i], wkl[4 * g + i], kt);
        vt = mfma3<ONE>(yh[i], yl[i], wvh[4 * g + i], wvl[4 * g + i], vt);
      }
    }
    float tm = kt[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) tm = fmaxf(tm, kt[r]);
    tm = fmaxf(tm, lane_xor(tm, 5));
    const float mn = fmaxf(m, tm);
    const float f = __expf(m - mn);
    float ps = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      kt[r] = __expf(kt[r] - mn);
      ps += kt[r];
      ctx[r] *= f;
    }
    ssum = ssum * f + ps;
    m = mn;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 ph, pl, vh, vl;
      split8(kt, s * 8, ph, pl);
      split8(vt, s * 8, vh, vl);
      ctx = mfma3<ONE>(vh, vl, ph, pl, ctx);  // ctx^T[e][d] += sum_pixels v[pixel][e] p[pixel][d]
    }
  }
  ssum += lane_xor(ssum, 5);
  float* pp = a.part + (((long long)frame * a.nsplit + split) * LH + h) * PART;
  if (lk == 0) { pp[lrow] = m; pp[32 + lrow] = ssum; }
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int e = (r & 3) + 8 * (r >> 2) + 4 * lk;
    pp[64 + e * LD + lrow] = ctx[r];
  }
}

// --------------------------------------------------------------------------------------------------------------- combine
// one workgroup per (frame, head): thread (eg = tid >> 5, d = tid & 31) owns ctx^T[e][d] for e = eg*4 .. eg*4+3
__global__ __launch_bounds__(256) void linattn_combine_kernel(const LAArgs a) {
  const int frame = blockIdx.x / LH, h = blockIdx.x - frame * LH;
  const int d = threadIdx.x & 31, eg = threadIdx.x >> 5;
  const int b = frame / a.T;
  const float* pbase = a.part + ((long long)frame * a.nsplit * LH + h) * PART;
  const long long pstride = (long long)LH * PART;
  const float* ekb = a.ek ? a.ek + ((long long)b * a.ntok) * (LH * LD) + h * LD : nullptr;
  const float* evb = a.ev ? a.ev + ((long long)b * a.ntok) * (LH * LD) + h * LD : nullptr;
  const int ntok = a.ek ? a.ntok : 0;
  float M = -INFINITY;
  for (int s = 0; s < a.nsplit; ++s) M = fmaxf(M, pbase[s * pstride + d]);
  for (int j = 0; j < ntok; ++j) M = fmaxf(M, ekb[(long long)j * (LH * LD) + d]);
  float Z = 0.f, c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.nsplit; ++s) {
    const float* pp = pbase + s * pstride;
    const float f = __expf(pp[d] - M);
    Z += pp[32 + d] * f;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += pp[64 + (eg * 4 + i) * LD + d] * f;
  }
  for (int j = 0; j < ntok; ++j) {
    const float p = __expf(ekb[(long long)j * (LH * LD) + d] - M);
    Z += p;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += p * evb[(long long)j * (LH * LD) + eg * 4 + i];
  }
  const float sc = a.q_scale / (Z * (float)a.HW);
  // operand order: contraction index d <-> (step s, half lk, element j) with d = (j & 3) + 8 (2 s + (j >> 2)) + 4 lk
  const int s = d >> 4, lk = (d >> 2) & 1, j = (d & 3) + 4 * ((d >> 3) & 1);
  unsigned short* fb = reinterpret_cast<unsigned short*>(a.ctxfrag + ((long long)frame * LH + h) * 256);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float v = c[i] * sc;
    unsigned short lo;
    const unsigned short hi = vmm_split16(v, lo);
    const int lane = lk * 32 + eg * 4 + i;
    fb[((s * 2 + 0) * 64 + lane) * 8 + j] = hi;
    fb[((s * 2 + 1) * 64 + lane) * 8 + j] = lo;
  }
}

// ---------------------------------------------------------------------------------------------------------------- pass B
template <bool ONE, int CC, typename ST = float>
__global__ __launch_bounds__(512) void linattn_apply_kernel(const LAArgs a) {
  constexpr int NV = CC / 64, NS = CC / 16, NCT = CC / 32, YP = LAGeom<CC>::YPITCH;
  const ST* const xin = static_cast<const ST*>(a.x);
  ST* const outp = static_cast<ST*>(a.out);
  constexpr bool HOLD_WO = CC == 64;  // the to_out fragments of the head stay in registers (C = 64) or are streamed per tile (C = 128)
  extern __shared__ __attribute__((aligned(16))) float red[];  // [8 heads][32 pixels][64 channels] (one 64-channel half at a time), then the staged input tile
  unsigned short* ytile = reinterpret_cast<unsigned short*>(red + LH * 32 * 64);
  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int frame = blockIdx.x / a.nsplit, split = blockIdx.x - frame * a.nsplit;
  const int tiles = a.HW / 32;
  const int t_begin = split * a.sps, t_end = min(tiles, t_begin + a.sps);

  uint4 wqh[NS], wql[NS], woh[HOLD_WO ? NCT : 1][2], wol[HOLD_WO ? NCT : 1][2], ch[2], cl[2];
#pragma unroll
  for (int s = 0; s < NS; ++s) {
    const uint4* q = a.wqkv + (((long long)h * NS + s) * 2) * 64 + lane;
    wqh[s] = q[0]; wql[s] = q[64];
  }
  const uint4* wo_base = a.wout + ((long long)(2 * h) * 2) * 64 + lane;  // + (ct * 16 + s) * 128
  if constexpr (HOLD_WO) {
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const uint4* q = wo_base + (ct * 16 + s) * 128;
        woh[ct][s] = q[0]; wol[ct][s] = q[64];
      }
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint4* q = a.ctxfrag + ((long long)frame * LH + h) * 256 + (s * 2) * 64 + lane;
    ch[s] = q[0]; cl[s] = q[64];
  }
  // reduction role: pixel rp, channels rc .. rc+3 of every 64-channel half
  const int rp = tid >> 4, rc = (tid & 15) * 4;
  f32x4 bias[NV], gam[NV];
#pragma unroll
  for (int v = 0; v < NV; ++v) {
    bias[v] = a.bias_out ? *reinterpret_cast<const f32x4*>(a.bias_out + v * 64 + rc) : f32x4{0.f, 0.f, 0.f, 0.f};
    gam[v] = *reinterpret_cast<const f32x4*>(a.gamma + v * 64 + rc);
  }
  using XRaw = decltype(ldraw4(xin));  // raw bits, unpacked at the point of use (see the context kernel)
  auto load_x = [&](int t, XRaw (&d)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      d[v] = XRaw{};
      if (t < t_end) d[v] = ldraw4(xin + ((long long)frame * a.HW + t * 32 + rp) * a.ldx + v * 64 + rc);
    }
  };
  auto unpack = [&](const XRaw (&r)[NV], f32x4 (&d)[NV]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) d[v] = unpack4(r[v]);
  };
#ifndef VMM_LA_PIPE
#define VMM_LA_PIPE 1
#endif
  if constexpr (CC == 64 && VMM_LA_PIPE) {
    // C = 64: ONE barrier per tile.  The head-sum buffer and the staged rows are double-buffered (2 x 64 KB + 2 x 8.5 KB of LDS; the
    // workgroup is alone on its CU anyway): a tile's matrix phase writes red[buf], the NEXT tile's rows are normalised into ytile[buf ^ 1]
    // behind it, one barrier publishes both, then the head sum of this tile runs straight into the next tile's matrix phase.
    unsigned short* yt0 = reinterpret_cast<unsigned short*>(red + 2 * LH * 32 * 64);
    auto red_of = [&](int b2) { return red + b2 * (LH * 32 * 64); };
    auto yt_of = [&](int b2) { return yt0 + b2 * (32 * YP); };
    XRaw xa[NV], xb[NV];
    load_x(t_begin, xa);
    {
      f32x4 xu[NV];
      unpack(xa, xu);
      stage_norm_row<CC>(a, xu, gam, yt_of(0), tid);
    }
    load_x(t_begin + 1, xb);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
      const int buf = (t - t_begin) & 1;
      const long long row0 = (long long)frame * a.HW + t * 32;
      f32x16 qt = zero16();
      {
        uint4 yh[4], yl[4];
        read_row_frags<CC>(yt_of(buf), lrow, lk, 0, yh, yl);
#pragma unroll
        for (int i = 0; i < 4; ++i) qt = mfma3<ONE>(wqh[i], wql[i], yh[i], yl[i], qt);
      }
      float mx = qt[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qt[r]);
      mx = fmaxf(mx, lane_xor(mx, 5));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { qt[r] = __expf(qt[r] - mx); sum += qt[r]; }
      sum += lane_xor(sum, 5);
      const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
      for (int r = 0; r < 16; ++r) qt[r] *= inv;
      f32x16 ot = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 qh, ql;
        split8(qt, s * 8, qh, ql);
        ot = mfma3<ONE>(ch[s], cl[s], qh, ql, ot);
      }
      f32x16 pc[2] = {zero16(), zero16()};
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        uint4 oh, ol;
        split8(ot, s * 8, oh, ol);
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) pc[ct] = mfma3<ONE>(oh, ol, woh[ct][s], wol[ct][s], pc[ct]);
      }
      float* rb = red_of(buf) + (h * 32) * 64;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int px = (r & 3) + 8 * (r >> 2) + 4 * lk;
        rb[px * 64 + lrow] = pc[0][r];
        rb[px * 64 + 32 + lrow] = pc[1][r];
      }
      {
        f32x4 xu[NV];
        unpack(xb, xu);
        stage_norm_row<CC>(a, xu, gam, yt_of(buf ^ 1), tid);  // the next tile's rows (zeros past the end)
      }
      XRaw xc[NV];
      load_x(t + 2, xc);
      __syncthreads();
      f32x4 acc = bias[0];
#pragma unroll
      for (int w = 0; w < LH; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red_of(buf) + (w * 32 + rp) * 64 + rc);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      const f32x4 xr = unpack4(xa[0]);
      acc.x += xr.x; acc.y += xr.y; acc.z += xr.z; acc.w += xr.w;  // residual: the element this thread normalised
      st4(outp + (row0 + rp) * a.ldo + rc, acc);
#pragma unroll
      for (int v = 0; v < NV; ++v) { xa[v] = xb[v]; xb[v] = xc[v]; }
    }
    return;
  }
  XRaw x_next[NV];
  load_x(t_begin, x_next);
  for (int t = t_begin; t < t_end; ++t) {
    const long long row0 = (long long)frame * a.HW + t * 32;
    f32x4 x_cur[NV];
    unpack(x_next, x_cur);
    load_x(t + 1, x_next);
    stage_norm_row<CC>(a, x_cur, gam, ytile, tid);
    __syncthreads();  // tile rows visible; also: every wave has finished the previous tile's head sum (red is free again)
    // streamed to_out fragments: the first channel tile's two steps are requested now and land under the q projection
    uint4 sh[2][2], sl[2][2];  // [ring slot][step]
    if constexpr (!HOLD_WO) {
#pragma unroll
      for (int s = 0; s < 2; ++s) { const uint4* q = wo_base + s * 128; sh[0][s] = q[0]; sl[0][s] = q[64]; }
    }
    f32x16 qt = zero16();  // rows d, column pixel = lrow
#pragma unroll
    for (int g = 0; g < NS / 4; ++g) {
      uint4 yh[4], yl[4];
      read_row_frags<CC>(ytile, lrow, lk, 4 * g, yh, yl);
#pragma unroll
      for (int i = 0; i < 4; ++i) qt = mfma3<ONE>(wqh[4 * g + i], wql[4 * g + i], yh[i], yl[i], qt);
    }
    float mx = qt[0];
#pragma unroll
    for (int r = 1; r < 16; ++r) mx = fmaxf(mx, qt[r]);
    mx = fmaxf(mx, lane_xor(mx, 5));
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) { qt[r] = __expf(qt[r] - mx); sum += qt[r]; }
    sum += lane_xor(sum, 5);
    const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
    for (int r = 0; r < 16; ++r) qt[r] *= inv;
    f32x16 ot = zero16();  // rows e, column pixel
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 qh, ql;
      split8(qt, s * 8, qh, ql);
      ot = mfma3<ONE>(ch[s], cl[s], qh, ql, ot);
    }
    uint4 oh[2], ol[2];
#pragma unroll
    for (int s = 0; s < 2; ++s) split8(ot, s * 8, oh[s], ol[s]);
    float* rb = red + (h * 32) * 64;
#pragma unroll
    for (int half = 0; half < NV; ++half) {  // 64 output channels at a time through the head-sum buffer
      if (half) __syncthreads();             // the previous half's sums have been read
#pragma unroll
      for (int c2 = 0; c2 < 2; ++c2) {
        const int ct = half * 2 + c2;
        f32x16 pc = zero16();  // rows pixels, column channel ct*32 + lrow
        if constexpr (HOLD_WO) {
#pragma unroll
          for (int s = 0; s < 2; ++s) pc = mfma3<ONE>(oh[s], ol[s], woh[ct][s], wol[ct][s], pc);
        } else {
          if (ct + 1 < NCT) {  // next channel tile's fragments into the other ring slot
#pragma unroll
            for (int s = 0; s < 2; ++s) { const uint4* q = wo_base + ((ct + 1) * 16 + s) * 128; sh[(ct + 1) & 1][s] = q[0]; sl[(ct + 1) & 1][s] = q[64]; }
          }
#pragma unroll
          for (int s = 0; s < 2; ++s) pc = mfma3<ONE>(oh[s], ol[s], sh[ct & 1][s], sl[ct & 1][s], pc);
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int px = (r & 3) + 8 * (r >> 2) + 4 * lk;
          rb[px * 64 + c2 * 32 + lrow] = pc[r];
        }
      }
      __syncthreads();
      f32x4 acc = bias[half];
#pragma unroll
      for (int w = 0; w < LH; ++w) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * 32 + rp) * 64 + rc);
        acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
      }
      acc.x += x_cur[half].x; acc.y += x_cur[half].y; acc.z += x_cur[half].z; acc.w += x_cur[half].w;  // residual: the element this thread normalised
      st4(outp + (row0 + rp) * a.ldo + half * 64 + rc, acc);
    }
  }
}

int choose_split(int frames, int HW, int* sps) { return vmm_linattn_block_split(frames, HW, sps); }

}  // namespace

#if !VMM_FP16_OPERANDS
// floats of workspace vmm_linattn_block_bf16x3 needs (partials + context fragments)
extern "C" int64_t vmm_linattn_block_workspace(int32_t B, int32_t T, int32_t HW) {
  int sps;
  const int ns = choose_split(B * T, HW, &sps);
  return (int64_t)B * T * ns * LH * PART + (int64_t)B * T * LH * 1024;
}
#endif

template <bool ONE, int CC, typename ST = float>
static int la_run(const LAArgs& a, unsigned blocks, int frames, hipStream_t s) {
  hipLaunchKernelGGL((linattn_ctx_kernel<ONE, CC, ST>), dim3(blocks), dim3(512), 0, s, a);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(linattn_combine_kernel, dim3((unsigned)(frames * LH)), dim3(256), 0, s, a);
  VMM_LAUNCH_CHECK();
  const size_t shm = (CC == 64 && VMM_LA_PIPE ? 2 : 1) * (sizeof(float) * LH * 32 * 64 + sizeof(unsigned short) * 32 * LAGeom<CC>::YPITCH);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_apply_kernel<ONE, CC, ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((linattn_apply_kernel<ONE, CC, ST>), dim3(blocks), dim3(512), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// Returns 1 (nothing launched) outside the envelope: C == 64 or 128, heads == 8, dim_head == 32, HW % 32 == 0.
template <bool ONE, typename ST = float>
static int la_launch(const void* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag, const float* bias_out, const float* ek,
                     const float* ev, int32_t ntok, float* workspace, void* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                     float eps, vmm_stream_t stream) {
  if ((C != 64 && C != 128) || heads != LH || (HW % 32) || (ldx & 3) || (ldo & 3)) return 1;
  if (B * T <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  LAArgs a;
  a.x = x; a.ldx = ldx; a.gamma = gamma; a.eps = eps;
  a.wqkv = reinterpret_cast<const uint4*>(wqkv_frag);
  a.wout = reinterpret_cast<const uint4*>(wout_frag);
  a.bias_out = bias_out;
  a.ek = ek; a.ev = ev; a.ntok = ek ? ntok : 0;
  a.T = T; a.HW = HW;
  a.nsplit = choose_split(B * T, HW, &a.sps);
  a.part = workspace;
  a.ctxfrag = reinterpret_cast<uint4*>(workspace + (long long)B * T * a.nsplit * LH * PART);
  a.out = out; a.ldo = ldo;
  a.q_scale = 1.0f / sqrtf((float)LD);
  const unsigned blocks = (unsigned)(B * T * a.nsplit);
  return C == 64 ? la_run<ONE, 64, ST>(a, blocks, B * T, s) : la_run<ONE, 128, ST>(a, blocks, B * T, s);
}

#if !VMM_FP16_OPERANDS
extern "C" int vmm_linattn_block_bf16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                        const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                                        int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream) {
  return la_launch<false>(x, ldx, gamma, wqkv_frag, wout_frag, bias_out, ek, ev, ntok, workspace, out, ldo, B, T, HW, C, heads, eps, stream);
}
// the "bf16" throughput mode of the same block (BASELINE.json configs[3]): identical arguments, one matrix pass per product on bf16-rounded operands
extern "C" int vmm_linattn_block_bf16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                      const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                                      int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream) {
  return la_launch<true>(x, ldx, gamma, wqkv_frag, wout_frag, bias_out, ek, ev, ntok, workspace, out, ldo, B, T, HW, C, heads, eps, stream);
}
// ... over bf16-STORED feature maps (x, out = bf16 bits; ld in elements): the "bf16" mode's two upper levels
extern "C" int vmm_linattn_block_bf16_a16(const void* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                          const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, void* out,
                                          int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream) {
  return la_launch<true, bf16s>(x, ldx, gamma, wqkv_frag, wout_frag, bias_out, ek, ev, ntok, workspace, out, ldo, B, T, HW, C, heads, eps, stream);
}
#elif VMM_SPLIT_F16
// three passes on IEEE-half hi | lo operands (vmm_common.h, VMM_SPLIT_F16); identical arguments and workspace, weights = vmm_pack_weights fmt 2 | 32 / 3 | 32
extern "C" int vmm_linattn_block_f16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                       const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                                       int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream) {
  return la_launch<false>(x, ldx, gamma, wqkv_frag, wout_frag, bias_out, ek, ev, ntok, workspace, out, ldo, B, T, HW, C, heads, eps, stream);
}
#else
// fp16 operands (`train_precision = "fp16"`: the reference's autocast dtype, main.py:34): the single-pass instance of this translation unit compiled with
// -DVMM_SINGLE_PASS=2; identical arguments and workspace layout (the context fragments it leaves for the backward are fp16 too: vmm_linattn_block_bwd_fp16
// reads them), weights = vmm_pack_weights fmt 2 | 16 / 3 | 16
extern "C" int vmm_linattn_block_fp16(const float* x, int32_t ldx, const float* gamma, const float* wqkv_frag, const float* wout_frag,
                                      const float* bias_out, const float* ek, const float* ev, int32_t ntok, float* workspace, float* out,
                                      int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, float eps, vmm_stream_t stream) {
  return la_launch<true>(x, ldx, gamma, wqkv_frag, wout_frag, bias_out, ek, ev, ntok, workspace, out, ldo, B, T, HW, C, heads, eps, stream);
}
#endif
