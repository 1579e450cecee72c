// Temporal attention core + output projection + residual for the levels whose to_qkv runs as a separate projection
// (C = 128, 256, 512), on the split-bf16 matrix cores, gfx950.
//
//   out = x + to_out( softmax_attention(q, [ek | k], [ev | v]) )      (vddp.py:491-534 and 421; q pre-scaled, q / k / ek pre-rotated)
//
// The fully fused block (temporal_block.hip) keeps the q/k/v weight fragments of a head in registers, which only fits C = 64.  Here
// to_qkv stays the A-stationary projection kernel (proj_bf16x3.hip, with the LayerNorm, q-scale and rotary in its staging /
// epilogue) and this kernel replaces the VALU attention core and the to_out GEMM: the 768-wide qkv rows are read exactly once, the
// 256-wide attention output never exists in memory, x is read once and out written once.
//
// Same tile and the same accumulator-as-operand chaining as temporal_block.hip: tile = 2 pixels x 16 frame slots, wave = head.
//   q^T, k  : 16 + 16 floats per lane straight from the qkv rows in the operand layout (lane = row, contraction index d in register
//             order), split to bf16 hi | lo in registers
//   v^T     : lane = feature d, elements = key rows -- 16 dword loads whose 32 lanes read one row's 128 contiguous bytes
//   s^T = k . q^T, s_tok^T = ek . q^T -> softmax over the lane's 8 + 8 scores (+ lane ^ 32 partner) -> o^T = v^T . p^T + ev^T . p_tok^T
//   part[m][c] = o . W_out,h^T for 128 output channels per pass -> sum over the 8 heads through LDS -> + x -> out
// The next tile's q / k / v are requested before the current tile's MFMAs (48 registers; the q/k/v weights are gone).
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int HEADS = 8;  // = waves per workgroup
constexpr int DHd = 32;
constexpr int HID = HEADS * DHd;
constexpr int PC = 128;   // output channels per head-sum pass

struct TCArgs {
  const float* qkv; int ldqkv;
  const float* x; int ldx;
  const uint4* wout;  // fmt 3 fragments of to_out (C, 256)
  const float* ek; const float* ev; int ntok;
  const float* bias; int bias_on_cond;
  float* out; int ldo;
  int T, HW, C, nsplit, tps;
};

__device__ __forceinline__ unsigned pack_split(float a, float b, unsigned& lo) { return split_bf16_pair(a, b, lo); }

__device__ __forceinline__ void split8(const f32x16& c, int r0, uint4& hi, uint4& lo) {
  hi.x = pack_split(c[r0 + 0], c[r0 + 1], lo.x);
  hi.y = pack_split(c[r0 + 2], c[r0 + 3], lo.y);
  hi.z = pack_split(c[r0 + 4], c[r0 + 5], lo.z);
  hi.w = pack_split(c[r0 + 6], c[r0 + 7], lo.w);
}

__device__ __forceinline__ void split8v(const float* v, uint4& hi, uint4& lo) {
  hi.x = pack_split(v[0], v[1], lo.x);
  hi.y = pack_split(v[2], v[3], lo.y);
  hi.z = pack_split(v[4], v[5], lo.z);
  hi.w = pack_split(v[6], v[7], lo.w);
}

__device__ __forceinline__ f32x16 mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 c) {
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, al), __builtin_bit_cast(bf16x8, bh), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bl), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ah), __builtin_bit_cast(bf16x8, bh), c, 0, 0, 0);
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

struct QKV { float q[16], k[16], v[16]; };

__global__ __launch_bounds__(512, 2) void temporal_core_kernel(const TCArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);   // [8 heads][32 rows][128 channels]
  float* biasf = red + HEADS * 32 * PC;              // [8 heads][2 halves][16 frames][8]

  const int tid = threadIdx.x, lane = tid & 63, h = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int pa = lrow >> 4, ft = lrow & 15;  // pixel of the pair, frame slot
  const int T = a.T, HW = a.HW;
  const int b = blockIdx.x / a.nsplit, split = blockIdx.x - b * a.nsplit;
  const int pairs = HW / 2;
  const int p_begin = split * a.tps, p_end = min(pairs, p_begin + a.tps);
  const int ntok = a.ek ? a.ntok : 0;

  // relative-position bias of query frame t against the 8 key frames a lane half holds: [h][lk][t][j]
  for (int i = tid; i < HEADS * 2 * 16 * 8; i += 512) {
    const int j = i & 7, t = (i >> 3) & 15, l2 = (i >> 7) & 1, hh = i >> 8;
    const int tk = slot(0, l2, j);
    biasf[i] = (t < T && tk < T) ? a.bias[(hh * T + t) * T + tk] : 0.f;
  }
  // conditioning keys as an A operand (rows = tokens, contraction = d) and values (rows = d, contraction = tokens): per (sample, head),
  // constant for the whole workgroup -> registers
  uint4 ekh[2], ekl[2], evh, evl;
  {
    float v[8];
#pragma unroll
    for (int s = 0; s < 2; ++s) {
#pragma unroll
      for (int j = 0; j < 8; ++j) v[j] = lrow < ntok ? a.ek[((long long)b * ntok + lrow) * HID + h * DHd + slot(s, lk, j)] : 0.f;
      split8v(v, ekh[s], ekl[s]);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int tk = slot(0, lk, j);
      v[j] = tk < ntok ? a.ev[((long long)b * ntok + tk) * HID + h * DHd + lrow] : 0.f;
    }
    split8v(v, evh, evl);
  }
  __syncthreads();
  const float* bias_l = biasf + ((h * 2 + lk) * 16 + ft) * 8;

  // this wave's slice of the qkv rows of one tile, in operand layout
  // Straight-line: every request is issued whatever the lane's frame slot or the tile index (padding slots re-read frame 0, the look-ahead
  // past the last tile re-reads it) and padding is zeroed by selects.  With a lane-dependent `if` around each of the 20 loads the tile's
  // requests left one basic block at a time, each behind a wait for the previous one.
  auto load_tile = [&](int pp, QKV& d) {
    const int ppc = min(pp, p_end - 1);
    const bool ok = ft < T;
    const float* row = a.qkv + (((long long)b * T + (ok ? ft : 0)) * HW + ppc * 2 + pa) * a.ldqkv + h * DHd + 4 * lk;
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 2; ++g) {  // elements j = 4g .. 4g+3 of step s: features 16 s + 8 g + 4 lk + 0..3
        const f32x4 q4 = *reinterpret_cast<const f32x4*>(row + 16 * s + 8 * g);
        const f32x4 k4 = *reinterpret_cast<const f32x4*>(row + HID + 16 * s + 8 * g);
        d.q[s * 8 + g * 4 + 0] = ok ? q4.x : 0.f; d.q[s * 8 + g * 4 + 1] = ok ? q4.y : 0.f; d.q[s * 8 + g * 4 + 2] = ok ? q4.z : 0.f; d.q[s * 8 + g * 4 + 3] = ok ? q4.w : 0.f;
        d.k[s * 8 + g * 4 + 0] = ok ? k4.x : 0.f; d.k[s * 8 + g * 4 + 1] = ok ? k4.y : 0.f; d.k[s * 8 + g * 4 + 2] = ok ? k4.z : 0.f; d.k[s * 8 + g * 4 + 3] = ok ? k4.w : 0.f;
      }
    // v^T: lane = feature lrow, element (s, j) = key row slot(s, lk, j) = (pixel, frame) of the tile
    const float* vbase = a.qkv + ((long long)b * T * HW + ppc * 2) * a.ldqkv + 2 * HID + h * DHd + lrow;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int m = slot(e >> 3, lk, e & 7), va = m >> 4, vt = m & 15;
      const float v = vbase[((long long)min(vt, T - 1) * HW + va) * a.ldqkv];
      d.v[e] = vt < T ? v : 0.f;
    }
  };

  // reduction role: row rm of the tile, channels rcol .. rcol+3 and rcol+64 .. rcol+67 of each 128-channel pass
  const int rm = tid >> 4, rcol = (tid & 15) * 4;
  const int rpa = rm >> 4, rft = rm & 15;
  const int passes = a.C / PC;

  // the residual rows of the first 128-channel pass travel with the tile's q / k / v (requested one tile ahead): loaded where they are added,
  // behind the head-sum barrier, they were a full HBM round trip on every tile's critical path
  auto load_res = [&](int pp, f32x4 (&d)[2]) {
    const long long row = ((long long)b * T + min(rft, T - 1)) * HW + min(pp, p_end - 1) * 2 + rpa;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf) d[hf] = *reinterpret_cast<const f32x4*>(a.x + row * a.ldx + rcol + hf * 64);
  };
  QKV nx;
  f32x4 xn[2];
  load_tile(p_begin, nx);
  load_res(p_begin, xn);
  for (int pp = p_begin; pp < p_end; ++pp) {
    QKV cu = nx;
    f32x4 xc[2] = {xn[0], xn[1]};
    load_tile(pp + 1, nx);  // in flight under this tile's MFMAs
    load_res(pp + 1, xn);
    uint4 qh[2], ql[2];
    split8v(cu.q, qh[0], ql[0]);
    split8v(cu.q + 8, qh[1], ql[1]);
    // ---- scores: keys x queries
    f32x16 st = zero16(), sk = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 kh, kl;
      split8v(cu.k + 8 * s, kh, kl);
      st = mfma3(kh, kl, qh[s], ql[s], st);
      if (ntok) sk = mfma3(ekh[s], ekl[s], qh[s], ql[s], sk);
    }
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
      f[j] = tk < T ? sv + bz[j] : -INFINITY;
      g[j] = tk < ntok ? sk[j] + (a.bias_on_cond ? bz[j] : 0.f) : -INFINITY;
      mx = fmaxf(mx, fmaxf(f[j], g[j]));
    }
    mx = fmaxf(mx, lane_xor(mx, 5));
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f[j] = __expf(f[j] - mx);
      g[j] = __expf(g[j] - mx);
      sum += f[j] + g[j];
    }
    sum += lane_xor(sum, 5);
    const float inv = __builtin_amdgcn_rcpf(sum);
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
      split8v(cu.v, vh, vl);
      split8v(p0, ph, pl);
      ot = mfma3(vh, vl, ph, pl, ot);
      split8v(cu.v + 8, vh, vl);
      split8v(p1, ph, pl);
      ot = mfma3(vh, vl, ph, pl, ot);
      if (ntok) {
        split8v(g, ph, pl);
        ot = mfma3(evh, evl, ph, pl, ot);
      }
    }
    uint4 oh[2], ol[2];
    split8(ot, 0, oh[0], ol[0]);
    split8(ot, 8, oh[1], ol[1]);
    // ---- to_out, 128 output channels per pass: this head's share, sum over heads through LDS, residual, store
    for (int ps = 0; ps < passes; ++ps) {
      f32x16 pc[4] = {zero16(), zero16(), zero16(), zero16()};
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) {
          const uint4* q = a.wout + (((long long)(ps * 4 + ct) * 16 + 2 * h + s) * 2) * 64 + lane;  // 16 KB per head and pass, L2-resident
          pc[ct] = mfma3(oh[s], ol[s], q[0], q[64], pc[ct]);
        }
      __syncthreads();  // the previous head sum has been read by everyone
      float* rb = red + (h * 32) * PC;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = (r & 3) + 8 * (r >> 2) + 4 * lk;
#pragma unroll
        for (int ct = 0; ct < 4; ++ct) rb[m * PC + ct * 32 + lrow] = pc[ct][r];
      }
      __syncthreads();
      if (rft < T) {
        const long long row = ((long long)b * T + rft) * HW + pp * 2 + rpa;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
          const int c = rcol + hf * 64;
          f32x4 acc = ps == 0 ? xc[hf] : *reinterpret_cast<const f32x4*>(a.x + row * a.ldx + ps * PC + c);
#pragma unroll
          for (int w = 0; w < HEADS; ++w) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(red + (w * 32 + rm) * PC + c);
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
          }
          *reinterpret_cast<f32x4*>(a.out + row * a.ldo + ps * PC + c) = acc;
        }
      }
    }
  }
}


// ---- mid-level spatial softmax attention per frame (vddp.py:687-689, 491-534) on the split-bf16 matrix cores: a flash-attention forward
// for dim_head = 32.  One wave per (frame, head, 32 queries); keys in chunks of 32 (the conditioning tokens first, as one more chunk):
//   s^T = k . q^T          lane = query, accumulator rows = the chunk's keys (each lane half holds 16 of the 32, its partner lane ^ 32 the rest)
//   online softmax         per lane over its 16 scores + one exchange with the partner; running max / sum, the output rescaled per lane
//   o^T += v^T . p^T       the probabilities enter as the MFMA's second operand in the accumulator order they already have (slot())
// The thread-per-query VALU kernel this replaces for inference (vmm_spatial_attention) spent 0.13 ms per step on 2 GFLOP.
struct SAArgs {
  const float* qkv; int ldqkv;
  const float* ek; const float* ev; int ntok, tok_per_frame;
  float* out; int ldo;
  int T, HW, heads;
};

__global__ __launch_bounds__(64) void spatial_attn_mfma_kernel(const SAArgs a) {
  const int lane = threadIdx.x & 63, lrow = lane & 31, lk = lane >> 5;
  const int head = blockIdx.y % a.heads, bt = blockIdx.y / a.heads;
  const int b = bt / a.T, t = bt - b * a.T;
  const int hid = a.heads * DHd;
  const long long row0 = (long long)bt * a.HW;
  const int q0 = blockIdx.x * 32;
  // this lane's query row (rows past HW re-read the last one; their columns are never stored)
  const float* qrow = a.qkv + (row0 + min(q0 + lrow, a.HW - 1)) * a.ldqkv + head * DHd + 4 * lk;
  uint4 qh[2], ql[2];
  {
    float q[16];
#pragma unroll
    for (int s = 0; s < 2; ++s)
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qrow + 16 * s + 8 * g);
        q[s * 8 + g * 4 + 0] = v.x; q[s * 8 + g * 4 + 1] = v.y; q[s * 8 + g * 4 + 2] = v.z; q[s * 8 + g * 4 + 3] = v.w;
      }
    split8v(q, qh[0], ql[0]);
    split8v(q + 8, qh[1], ql[1]);
  }
  float m = -INFINITY, l = 0.f;
  f32x16 ot = zero16();  // o^T: lane = query, rows = features (r & 3) + 8 (r >> 2) + 4 lk
  const int ntok = a.ek ? a.ntok : 0;
  const int nchunks = (a.HW + 31) >> 5;
  for (int c = ntok ? -1 : 0; c < nchunks; ++c) {  // (wave-uniform trip count; chunk -1 = the conditioning tokens)
    const bool tok = c < 0;
    const int nkeys = tok ? ntok : min(32, a.HW - c * 32);
    // keys of the chunk as rows: lane = key row lrow (clamped), 16 features in operand order
    const float* krow = tok ? a.ek + ((long long)b * ntok + min(lrow, ntok - 1)) * hid + head * DHd + 4 * lk
                            : a.qkv + (row0 + min(c * 32 + lrow, a.HW - 1)) * a.ldqkv + hid + head * DHd + 4 * lk;
    f32x16 st = zero16();
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      float k[8];
#pragma unroll
      for (int g = 0; g < 2; ++g) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(krow + 16 * s + 8 * g);
        k[g * 4 + 0] = v.x; k[g * 4 + 1] = v.y; k[g * 4 + 2] = v.z; k[g * 4 + 3] = v.w;
      }
      uint4 kh, kl;
      split8v(k, kh, kl);
      st = mfma3(kh, kl, qh[s], ql[s], st);
    }
    // values transposed: lane = feature lrow, element e = key slot(e >> 3, lk, e & 7) of the chunk
    float v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int kk = min(slot(e >> 3, lk, e & 7), nkeys - 1);
      v[e] = tok ? a.ev[((long long)b * ntok + kk) * hid + head * DHd + lrow] : a.qkv[(row0 + c * 32 + kk) * a.ldqkv + 2 * hid + head * DHd + lrow];
    }
    // scores of this lane's query against keys (r & 3) + 8 (r >> 2) + 4 lk: mask, running max / sum
    float p[16];
    float mx = -INFINITY;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int kk = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const bool ok = kk < nkeys && (!tok || !a.tok_per_frame || kk == t);  // tok_per_frame: frame t sees token t only (vddp.py:459-462)
      p[r] = ok ? st[r] : -INFINITY;
      mx = fmaxf(mx, p[r]);
    }
    mx = fmaxf(mx, lane_xor(mx, 5));
    const float mn = fmaxf(m, mx);
    const float alpha = mn == -INFINITY ? 1.f : __expf(m - mn);  // (a chunk with no visible key for this query leaves everything as it is)
    float sum = 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      p[r] = mn == -INFINITY ? 0.f : __expf(p[r] - mn);
      sum += p[r];
    }
    sum += lane_xor(sum, 5);
    l = l * alpha + sum;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 vh, vl, ph, pl;
      // zero the value columns of masked keys: their probability is 0, but a stand-in value could be anything finite only by luck
      float vs[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) vs[j] = slot(s, lk, j) < nkeys ? v[8 * s + j] : 0.f;
      split8v(vs, vh, vl);
      split8v(p + 8 * s, ph, pl);
      ot = mfma3(vh, vl, ph, pl, ot);
    }
  }
  if (q0 + lrow < a.HW) {
    const float inv = __builtin_amdgcn_rcpf(l);
    float* o = a.out + (row0 + q0 + lrow) * a.ldo + head * DHd + 4 * lk;
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(o + 8 * g) = f32x4{ot[4 * g] * inv, ot[4 * g + 1] * inv, ot[4 * g + 2] * inv, ot[4 * g + 3] * inv};
  }
}


// ---- spatial linear attention, pass 1 (vddp.py:367-373): per (frame, head, slice of positions) the partial context
//   ctx[d][e] = sum_n exp(k[n][d] - max_n k[:, d]) v[n][e],  max[d],  sum[d]        (the record linattn_merge_kernel totals, attention.hip)
// on the split-bf16 matrix cores.  One wave per (frame, head, slice); positions in chunks of 32, both operands loaded transposed (lane =
// feature, elements = positions in accumulator order, 32 lanes reading one row's 128 bytes):  ctx^T[e][d] += v^T[e][n] . p[n][d], so that a
// lane's accumulators all belong to ONE key feature d and the online-softmax rescale exp(max_old - max_new) is a per-lane factor.
// The VALU version (a thread per (d, 4 e), 2 x 32 x 32 flops per position on the vector unit) was compute-bound at 3x its HBM time.
struct LPArgs {
  const float* qkv; int ldqkv;
  float* part;
  int HW, heads, nsplit, rows_per_split;
};

__global__ __launch_bounds__(64) void linattn_partial_mfma_kernel(const LPArgs a) {
  const int lane = threadIdx.x & 63, lrow = lane & 31, lk = lane >> 5;
  const int split = blockIdx.x, fh = blockIdx.y;
  const int head = fh % a.heads;
  const long long frame = fh / a.heads;
  const int hid = a.heads * DHd;
  const int n_begin = split * a.rows_per_split, n_end = min(n_begin + a.rows_per_split, a.HW);
  const float* kcol = a.qkv + frame * a.HW * a.ldqkv + hid + head * DHd + lrow;  // k[n][lrow] = kcol[n * ld], v[n][lrow] = kcol[n * ld + hid]
  float m = -INFINITY, ssum = 0.f;
  f32x16 ct = zero16();
  // the next chunk's 32 requests are in flight while this one is reduced (one wave per workgroup: nothing else hides the latency)
  float kn[16], vn[16];
  auto load_chunk = [&](int n0) {
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int n = n0 + slot(e >> 3, lk, e & 7);
      const float* q = kcol + (long long)min(n, n_end - 1) * a.ldqkv;
      const float kv = q[0], vv = q[hid];
      kn[e] = n < n_end ? kv : -INFINITY;
      vn[e] = n < n_end ? vv : 0.f;
    }
  };
  if (n_begin < n_end) load_chunk(n_begin);
  for (int n0 = n_begin; n0 < n_end; n0 += 32) {
    float kt[16], vt[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) { kt[e] = kn[e]; vt[e] = vn[e]; }
    load_chunk(min(n0 + 32, n_end - 1));  // (past the end: re-reads the last position, unused)
    float tm = kt[0];
#pragma unroll
    for (int e = 1; e < 16; ++e) tm = fmaxf(tm, kt[e]);
    tm = fmaxf(tm, lane_xor(tm, 5));
    const float mn = fmaxf(m, tm);  // (finite: the chunk holds at least one position)
    const float alpha = __expf(m - mn);
    float p[16], sum = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      p[e] = __expf(kt[e] - mn);
      sum += p[e];
    }
    sum += lane_xor(sum, 5);
    ssum = ssum * alpha + sum;
    m = mn;
#pragma unroll
    for (int r = 0; r < 16; ++r) ct[r] *= alpha;
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      uint4 vh, vl, ph, pl;
      split8v(vt + 8 * s, vh, vl);
      split8v(p + 8 * s, ph, pl);
      ct = mfma3(vh, vl, ph, pl, ct);
    }
  }
  // ct[r] = ctx[d = lrow][e = (r & 3) + 8 (r >> 2) + 4 lk]
  float* pp = a.part + ((long long)fh * a.nsplit + split) * (DHd * DHd + 2 * DHd);
#pragma unroll
  for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(pp + lrow * DHd + 8 * g + 4 * lk) = f32x4{ct[4 * g], ct[4 * g + 1], ct[4 * g + 2], ct[4 * g + 3]};
  if (lk == 0) {
    pp[DHd * DHd + lrow] = m;
    pp[DHd * DHd + DHd + lrow] = ssum;
  }
}

}  // namespace

// qkv rows [(b,t,hw)][768] with q pre-scaled and q, k pre-rotated (the to_qkv projection's epilogue), ek / ev [B][ntok][256] (ek
// pre-rotated when per-frame) or NULL, bias [heads][T][T], wout_frag = vmm_pack_weights fmt 3 of to_out (C, 256).
// Returns 1 (nothing launched) outside the envelope: heads == 8, dim_head == 32, C a multiple of 128, T <= 16, ntok <= 16, HW even.
extern "C" int vmm_temporal_core_bf16x3(const float* qkv, int32_t ldqkv, const float* x, int32_t ldx, const float* wout_frag,
                                        const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                        float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads,
                                        vmm_stream_t stream) {
  if (heads != HEADS || C < PC || (C % PC) || T > 16 || T < 1 || (HW & 1) || (ldqkv & 3) || (ldx & 3) || (ldo & 3) || (ek && ntok > 16)) return 1;
  if (bias_on_cond && ek && ntok != T) return -2;
  if (B <= 0) return 0;
  TCArgs a;
  a.qkv = qkv; a.ldqkv = ldqkv; a.x = x; a.ldx = ldx;
  a.wout = reinterpret_cast<const uint4*>(wout_frag);
  a.ek = ek; a.ev = ev; a.ntok = ek ? ntok : 0;
  a.bias = bias; a.bias_on_cond = bias_on_cond;
  a.out = out; a.ldo = ldo; a.T = T; a.HW = HW; a.C = C;
  const int pairs = HW / 2;
  int ns = max(1, min(pairs, 256 / B));  // one 512-thread workgroup per CU (LDS), one round of workgroups
  a.tps = (pairs + ns - 1) / ns;
  a.nsplit = (pairs + a.tps - 1) / a.tps;
  const size_t shm = sizeof(float) * HEADS * 32 * PC + sizeof(float) * HEADS * 2 * 16 * 8;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_core_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_core_kernel, dim3((unsigned)(B * a.nsplit)), dim3(512), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// The matrix-core (split-bf16) version of vmm_spatial_attention for inference: same arguments without the log-sum-exp output.  Returns 1
// (nothing launched) outside its envelope: dim_head 32, at most 32 conditioning tokens.
extern "C" int vmm_spatial_attention_bf16x3(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t tok_per_frame,
                                            float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (dh != DHd || (ldqkv & 3) || (ldo & 3) || HW < 1 || (ek && (ntok < 1 || ntok > 32)) || (ek && tok_per_frame && ntok < T)) return 1;
  if (B <= 0 || T <= 0) return 0;
  SAArgs a;
  a.qkv = qkv; a.ldqkv = ldqkv; a.ek = ek; a.ev = ev; a.ntok = ek ? ntok : 0; a.tok_per_frame = tok_per_frame;
  a.out = out; a.ldo = ldo; a.T = T; a.HW = HW; a.heads = heads;
  hipLaunchKernelGGL(spatial_attn_mfma_kernel, dim3((unsigned)((HW + 31) / 32), (unsigned)(B * T * heads)), dim3(64), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// pass 1 of vmm_linattn_context on the matrix cores (see linattn_partial_mfma_kernel); rows_per_split a multiple of 32.  attention.hip's
// vmm_linattn_context_bf16x3 follows it with the merge pass.
extern "C" int vmm_linattn_partial_bf16x3(const float* qkv, int32_t ldqkv, int32_t frames, int32_t HW, int32_t heads, int32_t nsplit,
                                          int32_t rows_per_split, float* part, vmm_stream_t stream) {
  if (nsplit < 1 || rows_per_split < 32 || (rows_per_split & 31) || HW < 1) return -1;
  if (frames <= 0) return 0;
  LPArgs a;
  a.qkv = qkv; a.ldqkv = ldqkv; a.part = part; a.HW = HW; a.heads = heads; a.nsplit = nsplit; a.rows_per_split = rows_per_split;
  hipLaunchKernelGGL(linattn_partial_mfma_kernel, dim3((unsigned)nsplit, (unsigned)(frames * heads)), dim3(64), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
