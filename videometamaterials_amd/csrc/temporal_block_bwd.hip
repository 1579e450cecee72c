// Backward of the fused temporal-attention block (temporal_block.hip) at the C = 64 levels WITH RECOMPUTATION, split-bf16 matrix cores, gfx950.
//
//   out = x + to_out( softmax_attention( rotary(to_qkv( LayerNorm(x) )) ) )        (vddp.py:396-535 inside Residual(PreNorm(.)), vddp.py:615,630,680)
//
// The training forward is the fused block: it stores nothing but its output.  The unfused training path wrote the 768-wide qkv rows (1.25 GB per
// 96 x 96 site at batch 4), the attention output and the softmax statistics, and its backward read them all back.  Here the backward re-forms
// q, k, v and the probabilities from x on chip, takes dO = dOut . W_out on chip too, and runs the attention core's backward on the matrix cores:
//
//   inputs   x, dOut (rows x 64), the weights, tokens, bias and rotary tables
//   outputs  dqkv (rows x 768: the gradient of the RAW to_qkv rows -- rotation and scale undone -- for the fused to_qkv backward, qkv_bwd.hip),
//            the LayerNorm statistics of x (rows x 2, for the same kernel), dW_out, dbias, d(ek), d(ev) as per-workgroup partials + a fixed-order sum
//
// Tile and roles as in the forward: 2 pixels x 16 frame slots = 32 rows (row m = pixel * 16 + frame), one wave = one head, every product's result
// is the next product's operand straight from the accumulator registers (chain_mfma.h).  The backward needs every matrix in BOTH orientations
// (a contraction runs over the register index: dq contracts over keys, dk over queries, dW over rows, ...); the second orientation is made on the
// matrix pipe by a product with the identity (four single-pass MFMAs) instead of a second projection (twelve) or an LDS round trip.  Per tile and head,
// with X{R, C} = rows R in registers, column C = lane:
//   qT, kT{d, m} = W . y^T, rotary                      sT{key, query} = k . qT          skT{tok, query} = ek . qT
//   softmax over a column's 32 key rows (own pixel, frames < T) and 16 token rows -> pT, pkT
//   vT{d, m} = W_v . y^T    doT{d, m} = W_out,h^T . dOut^T                              dPT{key, query} = v . doT     dPkT = ev . doT
//   dS = p (dP - sum p dP)            dbias += dS
//   o{m, d} = p . v (+ tokens)        dW_out,h{d, c} += o^T . dOut
//   dqT{d, query} = k^T . dST (+ ek^T . dSkT)   -> scale, inverse rotary, transpose -> dq rows
//   dkT{d, key}   = q^T . dS                    -> scale, inverse rotary, transpose -> dk rows
//   dv{key, d}    = p^T . dO                                                         -> dv rows
//   d(ek){tok, d} += dSk^T . q       d(ev){tok, d} += pk^T . dO
// 54 k16 steps (x 3 passes) per tile and head against the forward's 25.  Weight fragments are streamed from L2 (the registers hold the
// accumulators of d(ek), d(ev), dbias; those of dW_out live in LDS, every lane adding to its own slots: 72 persistent registers spilled).
#include "chain_mfma.h"
#include "../../include/vmm_kernels.h"
#include <math.h>

namespace {

using namespace chain;

constexpr int TC = 64;     // channels
constexpr int HEADS = 8;   // = waves per workgroup
constexpr int DHd = 32;
constexpr int HID = HEADS * DHd;
constexpr int YP = 2 * TC + 8;  // bf16 per row of a row image: hi 64 | lo 64 | pad 8 (272 bytes = 17 x 16: conflict-free ds_read_b128)
constexpr int GP = 40;          // bf16 per channel of the column image: 32 positions + pad (80 bytes = 5 x 16)

#ifndef VMM_TBB_SKIP
#define VMM_TBB_SKIP 0  // measurement builds only (tools/build_ab.py -DVMM_TBB_SKIP=n): bit 0 drops phase 3, bit 1 phase 4 of the tile loop,
                        // bit 2 the stores of the qkv-row gradients, bit 3 the loads of x / dOut (constant tiles), bit 4 phases 1 and 2
#endif

struct TBBArgs {
  const float* x; int ldx;
  const float* gamma;
  const uint4* wqkv;  // fmt 2 fragments of to_qkv (768, 64)
  const uint4* woT;   // fmt 2 fragments of the (K = 64 channels, N = 256) operand: to_out (64, 256) read as [c][hd]
  const float* ek; const float* ev; int ntok;
  const float* bias; int bias_on_cond;
  const float* rot;   // [T][16][2] cos, sin
  const float* gout; int ldg;
  vmm_dqkv_t* gqkv; int ldq;        // (16-bit rows in the single-pass builds, vmm_common.h: VMM_DQKV16)
  float* ln_stats;
  float* part_wo;     // [grid][256 * 64]: the workgroup's dW_out in the packed-gradient layout [hd][c]
  float* part_db;     // [grid][8 * T * T]
  float* part_ek; float* part_ev;  // [nsplit][B][ntok][256]
  int B, T, HW, nsplit, tps;
  float q_scale, eps;
};

__global__ __launch_bounds__(512, 2) void temporal_block_bwd_kernel(const TBBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  uint4* ekA = reinterpret_cast<uint4*>(smem_raw);                      // [8 heads][2 steps][hi|lo][16 token rows x 2 halves]: A operand, rows = tokens
  uint4* evA = ekA + HEADS * 4 * 32;
  f32x4* dwo = reinterpret_cast<f32x4*>(evA + HEADS * 4 * 32);          // [8 heads][2 channel tiles][4 register quads][64 lanes]: dW_out accumulators
  float* biasf = reinterpret_cast<float*>(dwo + HEADS * 8 * 64);        // [8 heads][2 halves][16 frames][8]
  float* rotf = biasf + HEADS * 2 * 16 * 8;                             // [2 halves][16 frames][8 pairs][cos, sin]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(rotf + 2 * 16 * 8 * 2);  // [32 rows][YP]  LayerNorm(x)
  unsigned short* gtile = ytile + 32 * YP;                              // [32 rows][YP]  dOut rows
  unsigned short* gcol = gtile + 32 * YP;                               // [hi|lo][64 channels][GP]  dOut columns, positions in slot order

  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lk = lane >> 5;
  const int pa = lrow >> 4, ft = lrow & 15;  // pixel of the pair, frame slot of the column this lane holds
  const int T = a.T, HW = a.HW;
  const int b = blockIdx.x / a.nsplit, split = blockIdx.x - b * a.nsplit;
  const int pairs = HW / 2;
  const int p_begin = split * a.tps, p_end = min(pairs, p_begin + a.tps);
  const int ntok = a.ek ? a.ntok : 0;

  // ---- per-workgroup tables in LDS
  if (ntok) {
    if (lrow < 16) {  // (token rows 16 .. 31 of the A operands only feed result rows nobody reads: those lanes re-read rows 0 .. 15)
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        float vk[8], vv[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const long long o = ((long long)b * ntok + lrow) * HID + h * DHd + slot(s, lk, j);
          vk[j] = lrow < ntok ? a.ek[o] : 0.f;
          vv[j] = lrow < ntok ? a.ev[o] : 0.f;
        }
        uint4 hi, lo;
        split8v(vk, hi, lo);
        ekA[((h * 2 + s) * 2 + 0) * 32 + lk * 16 + lrow] = hi;
        ekA[((h * 2 + s) * 2 + 1) * 32 + lk * 16 + lrow] = lo;
        split8v(vv, hi, lo);
        evA[((h * 2 + s) * 2 + 0) * 32 + lk * 16 + lrow] = hi;
        evA[((h * 2 + s) * 2 + 1) * 32 + lk * 16 + lrow] = lo;
      }
    }
  }
  f32x4* dwo_l = dwo + (h * 8) * 64 + lane;  // + (ct * 4 + quad) * 64: the lane's own accumulator registers 4 quad .. 4 quad + 3 of channel tile ct
#pragma unroll
  for (int i = 0; i < 8; ++i) dwo_l[i * 64] = f32x4{0.f, 0.f, 0.f, 0.f};
  // relative-position bias of query frame t against the 8 key frames a lane half holds: [h][lk][t][j]
  for (int i = tid; i < HEADS * 2 * 16 * 8; i += 512) {
    const int j = i & 7, t = (i >> 3) & 15, l2 = (i >> 7) & 1, hh = i >> 8;
    const int tk = slot(0, l2, j);
    biasf[i] = (t < T && tk < T) ? a.bias[(hh * T + t) * T + tk] : 0.f;
  }
  // rotary factors of a (frame slot, lane half): (cos, sin) of the 8 feature pairs a lane holds in accumulator registers (2i, 2i + 1)
  for (int i = tid; i < 2 * 16 * 8; i += 512) {
    const int pr = i & 7, t = (i >> 3) & 15, l2 = i >> 7;
    const int d = slot(pr >> 2, l2, (2 * pr) & 7);  // feature index of register 2 pr
    const float2 cs = t < T ? *reinterpret_cast<const float2*>(a.rot + (t * 16 + (d >> 1)) * 2) : make_float2(1.f, 0.f);
    rotf[i * 2] = cs.x;
    rotf[i * 2 + 1] = cs.y;
  }
  const float* rot_l = rotf + ((lk * 16 + ft) * 8) * 2;
  // u <- R u (sg = +1) or R^T u (sg = -1), rows = features in register pairs
  auto rotate = [&](f32x16& u, float sg) {
#pragma unroll
    for (int i4 = 0; i4 < 4; ++i4) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(rot_l + i4 * 4);
#pragma unroll
      for (int k = 0; k < 2; ++k) {
        const int i = 2 * i4 + k;
        const float c = k ? cs.z : cs.x, sn = sg * (k ? cs.w : cs.y);
        const float e = u[2 * i], o = u[2 * i + 1];
        u[2 * i] = e * c - o * sn;
        u[2 * i + 1] = o * c + e * sn;
      }
    }
  };
  uint4 I[2];
  identity_frags(lane, I);

  // streamed weight fragments: plane (step s, hi | lo) of the head's column tile; a wave-uniform base + the lane's 16 bytes
  const unsigned char* wq_b = reinterpret_cast<const unsigned char*>(a.wqkv + ((long long)h * 4 * 2) * 64);
  const unsigned char* wk_b = reinterpret_cast<const unsigned char*>(a.wqkv + ((long long)(HEADS + h) * 4 * 2) * 64);
  const unsigned char* wv_b = reinterpret_cast<const unsigned char*>(a.wqkv + ((long long)(2 * HEADS + h) * 4 * 2) * 64);
  const unsigned char* wo_b = reinterpret_cast<const unsigned char*>(a.woT + ((long long)h * 4 * 2) * 64);
  unsigned loff = (unsigned)lane * 16u;
  // the 16 fragments (4 steps x {matrix 0 hi, lo, matrix 1 hi, lo}) of two matrices into the weight buffer: requested a phase ahead of their use
  auto load_w2 = [&](const unsigned char* m0, const unsigned char* m1, uint4 (&wb)[16]) {
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      // (steps 2, 3 from a second uniform base: the instruction's immediate offset ends at 4095, and beyond it the compiler builds one 64-bit
      // address per load in vector registers)
      const unsigned char* b0 = s < 2 ? m0 : m0 + 4096;
      const unsigned char* b1 = s < 2 ? m1 : m1 + 4096;
      wb[4 * s + 0] = *reinterpret_cast<const uint4*>(b0 + ((s & 1) * 2) * 1024 + loff);
      wb[4 * s + 1] = *reinterpret_cast<const uint4*>(b0 + ((s & 1) * 2 + 1) * 1024 + loff);
      wb[4 * s + 2] = *reinterpret_cast<const uint4*>(b1 + ((s & 1) * 2) * 1024 + loff);
      wb[4 * s + 3] = *reinterpret_cast<const uint4*>(b1 + ((s & 1) * 2 + 1) * 1024 + loff);
    }
  };
  const float* bias_l = biasf + ((h * 2 + lk) * 16 + ft) * 8;
  const uint4* ekA_l = ekA + (h * 4) * 32 + lk * 16 + (lrow & 15);  // + (s * 2 + plane) * 32
  const uint4* evA_l = evA + (h * 4) * 32 + lk * 16 + (lrow & 15);
  // staging role: row rm of the tile, channels rcol .. rcol + 3
  const int rm = tid >> 4, rcol = (tid & 15) * 4;
  const int rpa = rm >> 4, rft = rm & 15;
  const int gpos = (rm >> 4) * 16 + ((rm >> 2) & 1) * 8 + (rm & 3) + 4 * ((rm >> 3) & 1);  // position of row rm in slot order: step | half | element
  const f32x4 gam = *reinterpret_cast<const f32x4*>(a.gamma + rcol);
  unsigned x_loff = (unsigned)((rft * HW + rpa) * a.ldx + rcol), g_loff = (unsigned)((rft * HW + rpa) * a.ldg + rcol);
  const unsigned st_loff = (unsigned)(rft * HW + rpa) * 2u;
  auto load_xg = [&](int pp, f32x4& xv, f32x4& gv) {
    xv = f32x4{0.f, 0.f, 0.f, 0.f};
    gv = xv;
    if (rft < T && pp < p_end && !(VMM_TBB_SKIP & 8)) {  // (wave-uniform base of the tile + the lane's 32-bit offset: no 64-bit address registers per load)
      const long long row0 = (long long)b * T * HW + pp * 2;
      xv = *reinterpret_cast<const f32x4*>(a.x + row0 * a.ldx + x_loff);
      gv = *reinterpret_cast<const f32x4*>(a.gout + row0 * a.ldg + g_loff);
    }
  };
  auto stage = [&](int pp, const f32x4& xv, const f32x4& gv) {
    float s = (xv.x + xv.y) + (xv.z + xv.w);
    const float mean = row_sum16(s) * (1.0f / TC);
    const f32x4 c = {xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean};
    const float q = (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
    const float rstd = 1.0f / sqrtf(row_sum16(q) * (1.0f / TC) + a.eps);
    if ((tid & 15) == 0 && rft < T && pp < p_end)
      *reinterpret_cast<float2*>(a.ln_stats + 2 * ((long long)b * T * HW + pp * 2) + st_loff) = make_float2(mean, rstd);
    unsigned l0, l1;
    unsigned h0 = split_bf16_pair(c.x * rstd * gam.x, c.y * rstd * gam.y, l0);
    unsigned h1 = split_bf16_pair(c.z * rstd * gam.z, c.w * rstd * gam.w, l1);
    unsigned short* yt = ytile + rm * YP + rcol;
    *reinterpret_cast<uint2*>(yt) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(yt + TC) = make_uint2(l0, l1);
    h0 = split_bf16_pair(gv.x, gv.y, l0);
    h1 = split_bf16_pair(gv.z, gv.w, l1);
    unsigned short* gt = gtile + rm * YP + rcol;
    *reinterpret_cast<uint2*>(gt) = make_uint2(h0, h1);
    *reinterpret_cast<uint2*>(gt + TC) = make_uint2(l0, l1);
    unsigned short* gc = gcol + rcol * GP + gpos;
    gc[0] = (unsigned short)(h0 & 0xffffu);
    gc[GP] = (unsigned short)(h0 >> 16);
    gc[2 * GP] = (unsigned short)(h1 & 0xffffu);
    gc[3 * GP] = (unsigned short)(h1 >> 16);
    gc += 64 * GP;
    gc[0] = (unsigned short)(l0 & 0xffffu);
    gc[GP] = (unsigned short)(l0 >> 16);
    gc[2 * GP] = (unsigned short)(l1 & 0xffffu);
    gc[3 * GP] = (unsigned short)(l1 >> 16);
  };

  // accumulators that live for the whole kernel
  float dEk[8], dEv[8], db[8];  // {tok, d} rows 0 .. 15 (registers 0 .. 7 of the product); dbias of (query frame, 8 key frames)
#pragma unroll
  for (int j = 0; j < 8; ++j) dEk[j] = dEv[j] = db[j] = 0.f;

  // rows of the gradient of the raw qkv: X{m, d} -> gqkv[row(m)][col0 + d]
  // Stores of the gradient rows: ONE wave-uniform pointer per tile (scalar registers) + a 32-bit offset per lane and register.  (A uniform pointer
  // per register row instead became sixteen loop-carried 64-bit induction variables, spilled and reloaded every tile.)
  unsigned qr_loff = (unsigned)(4 * lk * HW * a.ldq + lrow);  // lane part of a row-form store: frame 4 lk of the register's frame group, feature lrow
  const unsigned hwq = (unsigned)(HW * a.ldq);
  auto store_rows = [&](const f32x16& X, int pp, int col0) {
    vmm_dqkv_t* gq = a.gqkv + ((long long)b * T * HW + pp * 2) * a.ldq + col0 + h * DHd;  // wave-uniform
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int t0 = (r & 3) + 8 * ((r >> 2) & 1), pm = r >> 3;  // frame slot (+ 4 lk), pixel of row row_of(r, lk)
      if (t0 + 4 * lk < T && !((VMM_TBB_SKIP & 4) && X[r] != 12345.f)) st_dqkv1(gq + (qr_loff + (unsigned)t0 * hwq + (unsigned)(pm * a.ldq)), X[r]);
    }
  };
  // columns of the gradient of the raw qkv from a T-form matrix X{d, m}: the lane's row m gets four 16-byte pieces (features 8 q + 4 lk .. + 3)
  unsigned qc_loff = (unsigned)((ft * HW + pa) * a.ldq + 4 * lk);
  auto store_cols = [&](const f32x16& X, int pp, int col0) {
    if (ft < T && !((VMM_TBB_SKIP & 4) && X[0] != 12345.f)) {
      vmm_dqkv_t* gq = a.gqkv + ((long long)b * T * HW + pp * 2) * a.ldq + col0 + h * DHd;  // wave-uniform
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) st_dqkv4(gq + 8 * q4 + qc_loff, X[4 * q4], X[4 * q4 + 1], X[4 * q4 + 2], X[4 * q4 + 3]);
    }
  };

  f32x4 xv, gv;
  load_xg(p_begin, xv, gv);
  uint4 wb[16];  // weight buffer: (W_q, W_k) from phase 4 of a tile to phase 1 of the next, (W_v, W_out^T) from phase 1 to phase 2
  load_w2(wq_b, wk_b, wb);
  for (int pp = p_begin; pp < p_end; ++pp) {
    stage(pp, xv, gv);  // (one tile in LDS -- the dW_out accumulators take 64 KB of it --: a barrier on either side of the tile's products)
    __syncthreads();
    // The lanes' 32-bit offsets are made opaque once per iteration: as loop invariants the compiler adds them to every wave-uniform base OUTSIDE the
    // loop -- one 64-bit vector-register address per load / store, ~60 registers that it then spills -- instead of using the
    // scalar-base + vector-offset addressing mode of the instruction.
    asm volatile("" : "+v"(loff), "+v"(x_loff), "+v"(g_loff), "+v"(qr_loff), "+v"(qc_loff));
    load_xg(pp + 1, xv, gv);  // a tile ahead: the HBM latency hides under this tile's products
    const unsigned short* yt = ytile + lrow * YP + lk * 8;
    const unsigned short* gt = gtile + lrow * YP + lk * 8;

    // ================= phase 1: q^T, k^T (rotated), scores, softmax
    f32x16 pT;       // {key, query}: probabilities, zero outside the query's own pixel / beyond T
    float pk[8];     // {token, query}: rows 0 .. 15 <-> registers 0 .. 7
    F2 krf, qrf;     // k, q {m, d} (rotated) as fragments: carried to phase 4 (32 registers; the weight buffer idle over phases 3 and 4 was 64)
    {
      f32x16 qT = zero16(), kT = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
        qT = mfma3(wb[4 * s], wb[4 * s + 1], yh, yl, qT);
        kT = mfma3(wb[4 * s + 2], wb[4 * s + 3], yh, yl, kT);
      }
      __builtin_amdgcn_sched_barrier(0);
      load_w2(wv_b, wo_b, wb);  // for phase 2: a softmax ahead
      __builtin_amdgcn_sched_barrier(0);
      rotate(qT, 1.f);
      rotate(kT, 1.f);
      const F2 qf = tofrag(qT), kf = tofrag(kT);
      f32x16 sT = mmT(kf, qf, zero16()), skT = zero16();
      if (ntok) {
#pragma unroll
        for (int s = 0; s < 2; ++s) skT = mfma3(ekA_l[(s * 2) * 32], ekA_l[(s * 2 + 1) * 32], qf.h[s], qf.l[s], skT);
      }
      const f32x4 bz0 = *reinterpret_cast<const f32x4*>(bias_l), bz1 = *reinterpret_cast<const f32x4*>(bias_l + 4);
      const float bz[8] = {bz0.x, bz0.y, bz0.z, bz0.w, bz1.x, bz1.y, bz1.z, bz1.w};
      float mx = -INFINITY;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int tk = slot(0, lk, r & 7);
        const bool ok = ((r >> 3) == pa) && tk < T;
        pT[r] = ok ? sT[r] * a.q_scale + bz[r & 7] : -INFINITY;
        mx = fmaxf(mx, pT[r]);
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int tk = slot(0, lk, j);
        pk[j] = tk < ntok ? skT[j] * a.q_scale + (a.bias_on_cond ? bz[j] : 0.f) : -INFINITY;
        mx = fmaxf(mx, pk[j]);
      }
      mx = fmaxf(mx, lane_xor(mx, 5));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { pT[r] = __expf(pT[r] - mx); sum += pT[r]; }
#pragma unroll
      for (int j = 0; j < 8; ++j) { pk[j] = __expf(pk[j] - mx); sum += pk[j]; }
      sum += lane_xor(sum, 5);
      const float inv = 1.0f / sum;
#pragma unroll
      for (int r = 0; r < 16; ++r) pT[r] *= inv;
#pragma unroll
      for (int j = 0; j < 8; ++j) pk[j] *= inv;
      krf = tofrag(transp(kf, I));
      qrf = tofrag(transp(qf, I));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ================= phase 2: v^T, dO^T, dP, dS; row forms of v and dO
    F2 vrf, dorf, dsf, pf;   // v, dO {m, d}; dS^T, p^T {key, query}
    uint4 dskh, dskl, pkh, pkl;  // dS_tok^T, p_tok^T {tok, query} (one k16 step each)
    {
      f32x16 vT = zero16(), doT = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
        vT = mfma3(wb[4 * s], wb[4 * s + 1], yh, yl, vT);
        const uint4 gh = *reinterpret_cast<const uint4*>(gt + s * 16), gl = *reinterpret_cast<const uint4*>(gt + s * 16 + TC);
        doT = mfma3(wb[4 * s + 2], wb[4 * s + 3], gh, gl, doT);
      }
      __builtin_amdgcn_sched_barrier(0);
      const F2 vf = tofrag(vT), dof = tofrag(doT);
      f32x16 dPT = mmT(vf, dof, zero16()), dPk = zero16();
      if (ntok) {
#pragma unroll
        for (int s = 0; s < 2; ++s) dPk = mfma3(evA_l[(s * 2) * 32], evA_l[(s * 2 + 1) * 32], dof.h[s], dof.l[s], dPk);
      }
      float dl = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) dl += pT[r] * dPT[r];
#pragma unroll
      for (int j = 0; j < 8; ++j) dl += pk[j] * dPk[j];
      dl += lane_xor(dl, 5);
      float dsk[8];
#pragma unroll
      for (int r = 0; r < 16; ++r) dPT[r] = pT[r] * (dPT[r] - dl);  // dS^T
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        dsk[j] = pk[j] * (dPk[j] - dl);
        db[j] += dPT[j] + dPT[8 + j] + (a.bias_on_cond ? dsk[j] : 0.f);  // (the other pixel's eight are exact zeros)
      }
      dsf = tofrag(dPT);
      split8v(dsk, dskh, dskl);
      pf = tofrag(pT);
      split8v(pk, pkh, pkl);
      vrf = tofrag(transp(vf, I));
      dorf = tofrag(transp(dof, I));
    }
    __builtin_amdgcn_sched_barrier(0);

    // ================= phase 3: o and dW_out, dv, d(ev)
    if constexpr (!(VMM_TBB_SKIP & 1)) {
      f32x16 o = mmT(pf, vrf, zero16());
      if (ntok) {  // token values as a B operand {tok, d}: an A image times the identity is the image's matrix as an accumulator
        F2 e;
        e.h[0] = evA_l[0]; e.l[0] = evA_l[32]; e.h[1] = evA_l[64]; e.l[1] = evA_l[96];
        uint4 eh, el;
        split8(transp(e, I), 0, eh, el);
        o = mfma3(pkh, pkl, eh, el, o);
      }
      const F2 of = tofrag(o);
      const unsigned short* gc = gcol + lrow * GP + lk * 8;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {  // (accumulators in LDS: each lane adds to its own registers' slots)
        f32x16 acc;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
          const f32x4 v = dwo_l[(ct * 4 + q4) * 64];
          acc[4 * q4] = v.x; acc[4 * q4 + 1] = v.y; acc[4 * q4 + 2] = v.z; acc[4 * q4 + 3] = v.w;
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
          const uint4 ch = *reinterpret_cast<const uint4*>(gc + ct * 32 * GP + s * 16);
          const uint4 cl = *reinterpret_cast<const uint4*>(gc + 64 * GP + ct * 32 * GP + s * 16);
          acc = mfma3(of.h[s], of.l[s], ch, cl, acc);
        }
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) dwo_l[(ct * 4 + q4) * 64] = f32x4{acc[4 * q4], acc[4 * q4 + 1], acc[4 * q4 + 2], acc[4 * q4 + 3]};
      }
      const F2 prf = tofrag(transp(pf, I));  // p {query, key}
      store_rows(mmT(prf, dorf, zero16()), pp, 2 * HID);  // dv {key, d} = p^T . dO
      if (ntok) {
        const F2 pkr = tofrag(transp16(pkh, pkl, I));     // p_tok {query, tok}
        const f32x16 tv_ = mmT(pkr, dorf, zero16());
#pragma unroll
        for (int j = 0; j < 8; ++j) dEv[j] += tv_[j];
      }
    }
    __builtin_amdgcn_sched_barrier(0);

    // ================= phase 4: dq, dk, d(ek); (W_q, W_k) of the next tile requested at its start (phases 2 and 3 need the registers)
    load_w2(wq_b, wk_b, wb);
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (!(VMM_TBB_SKIP & 2)) {
      // dq^T {d, query} = k^T . dS^T (+ tokens), scale, inverse rotary
      f32x16 dqT = mmT(krf, dsf, zero16());
      if (ntok) {  // token keys as an A operand [d][tok] = the fragments of {tok, d}
        F2 e;
        e.h[0] = ekA_l[0]; e.l[0] = ekA_l[32]; e.h[1] = ekA_l[64]; e.l[1] = ekA_l[96];
        uint4 eh, el;
        split8(transp(e, I), 0, eh, el);
        dqT = mfma3(eh, el, dskh, dskl, dqT);
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) dqT[r] *= a.q_scale;
      rotate(dqT, -1.f);
      store_cols(dqT, pp, 0);
      // dk^T {d, key} = q^T . dS
      const F2 dsrf = tofrag(transp(dsf, I));
      f32x16 dkT = mmT(qrf, dsrf, zero16());
#pragma unroll
      for (int r = 0; r < 16; ++r) dkT[r] *= a.q_scale;
      rotate(dkT, -1.f);
      store_cols(dkT, pp, HID);
      if (ntok) {
        const F2 dskr = tofrag(transp16(dskh, dskl, I));  // dS_tok {query, tok}
        const f32x16 tk_ = mmT(dskr, qrf, zero16());
#pragma unroll
        for (int j = 0; j < 8; ++j) dEk[j] += tk_[j];
      }
    }
    __syncthreads();
  }

  // ---- per-workgroup partials
  {
    float* pw = a.part_wo + (long long)blockIdx.x * (HID * TC);
#pragma unroll
    for (int ct = 0; ct < 2; ++ct)
#pragma unroll
      for (int r = 0; r < 16; ++r) pw[(h * DHd + row_of(r, lk)) * TC + ct * 32 + lrow] = reinterpret_cast<const float*>(dwo_l + (ct * 4 + (r >> 2)) * 64)[r & 3];
    float* pd = a.part_db + (long long)blockIdx.x * (HEADS * T * T);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float v = db[j] + lane_xor(db[j], 4);  // the two pixels of the tile
      const int tk = slot(0, lk, j);
      if (pa == 0 && ft < T && tk < T) pd[(h * T + ft) * T + tk] = v;
    }
    if (ntok) {
      float* pe = a.part_ek + (((long long)split * a.B + b) * ntok) * HID;
      float* pv = a.part_ev + (((long long)split * a.B + b) * ntok) * HID;
#pragma unroll
      for (int r = 0; r < 8; ++r) {
        const int tk = row_of(r, lk);
        if (tk < ntok) {
          pe[tk * HID + h * DHd + lrow] = dEk[r] * a.q_scale;
          pv[tk * HID + h * DHd + lrow] = dEv[r];
        }
      }
    }
  }
}

struct Split { int nsplit, tps; };
Split choose_split(int B, int HW) {
  const int units = HW / 2;
  const int ns = max(1, min(units, 256 / max(B, 1)));  // one 512-thread workgroup per CU (LDS), one round of workgroups
  Split s;
  s.tps = (units + ns - 1) / ns;
  s.nsplit = (units + s.tps - 1) / s.tps;
  return s;
}

bool supported(int T, int ntok, int HW, int C, int heads) {
  return C == TC && heads == HEADS && T >= 1 && T <= 16 && ntok >= 0 && ntok <= 16 && !(HW & 1);
}

}  // namespace

// floats of workspace vmm_temporal_block_bwd_bf16x3 needs; 0 outside its envelope (C == 64, heads == 8, T <= 16, ntok <= 16, even HW)
#if !VMM_SINGLE_PASS
extern "C" int64_t vmm_temporal_block_bwd_workspace(int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, int32_t ntok) {
  if (!supported(T, ntok, HW, C, heads) || B <= 0) return 0;
  const Split s = choose_split(B, HW);
  const long long G = (long long)B * s.nsplit;
  return G * (HID * TC) + G * (HEADS * T * T) + 2LL * s.nsplit * B * ntok * HID;
}

#endif
extern "C" int VMM_X3(vmm_temporal_block_bwd_, )(const vmm_attn_block_bwd* d, vmm_stream_t stream) {
  const int ntok = d->ek ? d->ntok : 0;
  if (!supported(d->T, ntok, d->HW, d->C, d->heads) || (d->ldx & 3) || (d->lddo & 3) || !d->workspace) return 1;
  if (d->bias_on_cond && ntok && ntok != d->T) return -2;
  if (d->B <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const Split sp = choose_split(d->B, d->HW);
  const long long G = (long long)d->B * sp.nsplit;
  TBBArgs a;
  a.x = d->x; a.ldx = d->ldx; a.gamma = d->gamma;
  a.wqkv = reinterpret_cast<const uint4*>(d->wqkv_frag);
  a.woT = reinterpret_cast<const uint4*>(d->wout_t_frag);
  a.ek = d->ek; a.ev = d->ev; a.ntok = ntok;
  a.bias = d->bias; a.bias_on_cond = d->bias_on_cond; a.rot = d->rot_tab;
  a.gout = d->dout; a.ldg = d->lddo;
  a.gqkv = reinterpret_cast<vmm_dqkv_t*>(d->dqkv); a.ldq = d->lddqkv; a.ln_stats = d->ln_stats;
  a.part_wo = d->workspace;
  a.part_db = a.part_wo + G * (HID * TC);
  a.part_ek = a.part_db + G * (HEADS * d->T * d->T);
  a.part_ev = a.part_ek + (long long)sp.nsplit * d->B * ntok * HID;
  a.B = d->B; a.T = d->T; a.HW = d->HW; a.nsplit = sp.nsplit; a.tps = sp.tps;
  a.q_scale = d->q_scale; a.eps = d->eps;
  const size_t shm = sizeof(uint4) * (2 * HEADS * 4 * 32 + HEADS * 8 * 64) + sizeof(float) * (HEADS * 2 * 16 * 8 + 2 * 16 * 8 * 2) +
                     sizeof(unsigned short) * (2 * 32 * YP + 2 * 64 * GP);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_block_bwd_kernel, dim3((unsigned)G), dim3(512), shm, s, a);
  VMM_LAUNCH_CHECK();
  int rc = vmm_sum_partials(a.part_wo, (int)G, HID * TC, HID * TC, d->dwout_packed, stream);
  if (rc) return rc;
  if (d->dbias) {
    rc = vmm_sum_partials(a.part_db, (int)G, HEADS * d->T * d->T, HEADS * d->T * d->T, d->dbias, stream);
    if (rc) return rc;
  }
  if (ntok) {
    const int n = d->B * ntok * HID;
    if (d->dek) { rc = vmm_sum_partials(a.part_ek, sp.nsplit, n, n, d->dek, stream); if (rc) return rc; }
    if (d->dev) { rc = vmm_sum_partials(a.part_ev, sp.nsplit, n, n, d->dev, stream); if (rc) return rc; }
  }
  return 0;
}
