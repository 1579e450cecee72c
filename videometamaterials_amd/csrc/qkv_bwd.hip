// Backward of to_qkv at the C = 64 levels (autograd of vddp.py:319, 413: qkv = LayerNorm(x) W^T, W (768, 64)) in ONE pass over the gradient of
// the qkv rows, split-bf16 matrix cores, gfx950:
//
//   gy[r][c]  = sum_n g[r][n] W[n][c]            (data gradient, 768 -> 64)
//   dW[n][c] += sum_r g[r][n] y[r][c]            (weight gradient; y = LayerNorm(x) re-formed from the forward's row statistics, or y itself)
//
// Run separately (narrow_proj.hip + wgrad1x1_bf16x3.hip) each of the two reads the 3 KB rows of g once: 2 x 1.25 GB at the 96 x 96 level, and
// both are HBM-bound.  Here a workgroup streams 64-row chunks of g in eight pieces of 96 columns; a loader thread (8 rows x 2 columns) splits
// its values once and writes them into BOTH LDS images a piece needs:
//   GT [column][row]  (16-byte fragments, rows contiguous)  -> "A" operand of the weight gradient  (contraction over rows)
//   GR [row][column]  (4-byte pairs, columns contiguous)    -> "B" operand of the data gradient    (contraction over columns)
// The weight gradient's 768 x 64 accumulators live in registers for the whole kernel: six 32 x 32 tiles per wave, one of which is at work in a piece
// (tile t of piece p belongs to wave (p + t) mod 8: every wave has at most 12 MFMAs of weight gradient per piece; round 5 -- before, two owner waves did
// 36 each and six waited at the barrier).  The data gradient's 64 x 64 tile of a chunk is four 32 x 32 accumulators, each shared by the two wave groups,
// which split a piece's six k16 steps (W^T fragments straight from L1 / L2 as the A operand); the halves meet in LDS when the chunk's eight pieces are
// done.  One barrier per piece, double-buffered images, the two wave groups half an iteration apart, two pieces of g in flight per loader thread (see
// qkv_bwd_body); per workgroup partial weight-gradient blocks + a fixed-order reduction, like the other weight-gradient kernels.
// 0.573 -> 0.50 ms per 96 x 96 site (batch 4) in round 5; what the knock-out builds (VMM_QB_SKIP) say about the rest: LABNOTES 10.9.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

#ifndef VMM_QB_SKIP
#define VMM_QB_SKIP 0  // measurement builds only (tools/build_ab.py qkv_bwd -DVMM_QB_SKIP=n): bit 0 no weight-gradient MFMAs, bit 1 no data-gradient MFMAs,
#endif                 // bit 2 no LDS image writes of the g pieces, bit 3 no global loads of g, bit 4 the staged values do not depend on the loads
constexpr int CH = 64;                  // rows per chunk
constexpr int NP = 96;                  // columns of g per piece (three 32-column fragments)
constexpr int NPIECE = 8;               // 768 / 96
constexpr int NQ = 768, CC = 64;
constexpr int TP = 2 * CH + 16;         // GT / YT: bytes per column (channel) row of a plane: 144 = 9 x 16
constexpr int RP = 2 * NP + 16;         // GR: bytes per row of a plane: 208 = 13 x 16
constexpr int GT_PLANE = NP * TP, GR_PLANE = CH * RP;
constexpr int PIECE_BUF = 2 * GT_PLANE + 2 * GR_PLANE;   // GT hi | GT lo | GR hi | GR lo = 54 272 bytes
constexpr int YT_PLANE = CC * TP, YT_BUF = 2 * YT_PLANE;  // 18 432 bytes
constexpr int LDS_BYTES = 2 * PIECE_BUF + 2 * YT_BUF;     // 145 408 bytes (+ GY_SCRATCH behind them)
constexpr int GY_SCRATCH = 4 * 4 * 64 * 16;              // 16 384 bytes: group 1's halves of the chunk's data-gradient tiles
constexpr int LN_SUMS = 2 * 2 * 64 * 8;                  // 2 048 bytes: the four half-tile contributions to a row's LayerNorm-backward sums (163 840 in all)
constexpr int PART_FLOATS = NQ * CC;

struct QBArgs {
  const float* x; int ldx;           // rows x 64: y itself, or (ln_stats != NULL) the un-normalised x
  const float* ln_stats;             // [rows][2] (mean, rstd) or NULL
  const float* ln_gamma;             // [64]
  const vmm_dqkv_t* g; int ldg;      // rows x 768 (single-pass builds: the operand's 16-bit type, vmm_common.h VMM_DQKV16; ldg in elements)
  const unsigned char* wfrag;        // vmm_pack_weights fmt 2 of the (K = 768, N = 64) operand W
  float* gy; int ldgy;               // rows x 64 (=)
  float* part;                       // [gridDim.x][PART_FLOATS]
  long long rows;
  int nchunks, chunks_per_wg;
  // LayerNorm backward as the epilogue (vmm_qkv_bwd_ln_*): dx (=|+=) the gradient of the block INPUT through PreNorm, gy itself is not written
  float* dx; int lddx; int accumulate;
  float* dgamma_part;                // [2 gridDim.x][64]: partial rows of the gamma gradient
};

// The body of the kernel for one wave group and loader role (four instances, selected once per wave: no role branches inside the loops, so the
// compiler's per-register load tracking sees one straight instruction stream per wave).
//
// Schedule.  Two wave groups (waves 0-3 / 4-7; a SIMD hosts one wave of each) run half an iteration apart: group 0 does its loader step BEFORE its
// products of a piece, group 1 AFTER, so that on every SIMD one wave's loader phase -- the wait for its global loads, the splits, the LDS writes --
// lies under the other wave's matrix phase.  A loader step = stage the next piece (in its register set since two steps: pieces of even / odd index own
// one set each, so two pieces per thread are in flight and a load has two iterations to land) into the buffer the previous iteration read, request the
// piece three ahead into the same registers.  The W^T fragments of the next piece are requested behind this piece's data-gradient products.  Loads
// return in order, and the compiler sizes every wait for the worst path into the loop: the prologue therefore leaves its requests pending in exactly the
// order the loop does, every request of the loop is unconditional, and the staged set's registers are pinned at the staging (see the comments there).
// Per group: three g-loader waves (192 threads = 4 row groups x 48 column pairs) and one y-loader wave (64 threads x 2 items per chunk).
template <int GRP, bool GROLE, bool LNB>
__device__ __forceinline__ void qkv_bwd_body(const QBArgs& a, unsigned char* sm) {
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int l31 = lane & 31, half = lane >> 5;
  const int c_begin = blockIdx.x * a.chunks_per_wg;
  const int n_ch = min(a.chunks_per_wg, a.nchunks - c_begin);
  const long long r_begin = (long long)c_begin * CH;
  unsigned char* const ybuf = sm + 2 * PIECE_BUF;

  // ---------------------------------------------------------------- loaders
  const int gt = tid - 256 * GRP;                                        // 0 .. 191 in the g role
  const int go = GROLE ? 4 * GRP + gt / 48 : 0, gp = GROLE ? gt % 48 : 0;  // 8 row groups x 48 column pairs
  const int yt = 64 * GRP + lane, yo0 = (yt >> 5) & 3, ycp = yt & 31;    // y item k (0 / 1): row group yo0 + 4 k, channel pair ycp
  f32x2 gvA[8];  // the role's rows in flight: the next g piece, or a y item
#if VMM_DQKV16
  unsigned gvG[8];  // (16-bit rows of g: a column pair per row is one dword; the y role keeps gvA)
#else
  f32x2 (&gvG)[8] = gvA;
#endif
  const f32x2 lg = (!GROLE && a.ln_stats) ? *reinterpret_cast<const f32x2*>(a.ln_gamma + 2 * ycp) : f32x2{1.f, 1.f};
  // (rows is a multiple of 64: no tail.  Addresses = a wave-uniform row base (scalar registers) + ONE 32-bit per-thread offset: eight 64-bit
  // vector-register addresses per role cost 16 registers each and pushed the first version into scratch, whose reloads sit on the same in-order
  // counter as the prefetched rows)
  unsigned g_toff = (unsigned)((8 * go) * a.ldg + 2 * gp), y_toff0 = (unsigned)(8 * yo0 * a.ldx + 2 * ycp);
#if VMM_DQKV16
  // 16-bit rows of g: a row's column pair is ONE dword -- already the GR image's dword (column 2 gp in the low half); the GT fragments are the same
  // halves regrouped by column (one v_perm per dword).  No conversion, no lo planes.
  typedef unsigned GV;
  auto g_request = [&](long long r0, int piece, GV (&gv)[8]) {  // rows r0 + 8 go .. + 7, columns piece * 96 + 2 gp
    const vmm_dqkv_t* gb = (VMM_QB_SKIP & 32) ? a.g + piece * NP : a.g + r0 * a.ldg + piece * NP;  // wave-uniform
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const vmm_dqkv_t* rowp = gb + (long long)i * a.ldg;
      gv[i] = (VMM_QB_SKIP & 8) ? 0x3c003c00u : *reinterpret_cast<const unsigned*>(rowp + g_toff);
    }
  };
  auto g_stage = [&](int buf, const GV (&gv)[8]) {
    unsigned char* base = sm + buf * PIECE_BUF;
    if ((VMM_QB_SKIP & 4) && gv[0] != 12345u) return;
    uint4 h, h1;
    h.x = __builtin_amdgcn_perm(gv[1], gv[0], 0x05040100u); h1.x = __builtin_amdgcn_perm(gv[1], gv[0], 0x07060302u);
    h.y = __builtin_amdgcn_perm(gv[3], gv[2], 0x05040100u); h1.y = __builtin_amdgcn_perm(gv[3], gv[2], 0x07060302u);
    h.z = __builtin_amdgcn_perm(gv[5], gv[4], 0x05040100u); h1.z = __builtin_amdgcn_perm(gv[5], gv[4], 0x07060302u);
    h.w = __builtin_amdgcn_perm(gv[7], gv[6], 0x05040100u); h1.w = __builtin_amdgcn_perm(gv[7], gv[6], 0x07060302u);
    *reinterpret_cast<uint4*>(base + (2 * gp) * TP + go * 16) = h;
    *reinterpret_cast<uint4*>(base + (2 * gp + 1) * TP + go * 16) = h1;
    unsigned char* gr = base + 2 * GT_PLANE + (8 * go) * RP + gp * 4;
#pragma unroll
    for (int i = 0; i < 8; ++i) *reinterpret_cast<unsigned*>(gr + i * RP) = gv[i];
  };
#else
  typedef f32x2 GV;
  auto g_request = [&](long long r0, int piece, f32x2 (&gv)[8]) {  // rows r0 + 8 go .. + 7, columns piece * 96 + 2 gp
    const float* gb = (VMM_QB_SKIP & 32) ? a.g + piece * NP : a.g + r0 * a.ldg + piece * NP;  // wave-uniform (bit 5: every chunk re-reads the first one -- L2 hits)
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* rowp = gb + (long long)i * a.ldg;
      gv[i] = (VMM_QB_SKIP & 8) ? f32x2{1.f, 2.f} : *reinterpret_cast<const f32x2*>(rowp + g_toff);
    }
  };
  auto g_stage = [&](int buf, const f32x2 (&gv)[8]) {
    unsigned char* base = sm + buf * PIECE_BUF;
    if ((VMM_QB_SKIP & 4) && gv[0][0] != 12345.f) return;
    float e0[8], e1[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      e0[i] = gv[i][0];
      e1[i] = gv[i][1];
    }
    // GT: one 16-byte fragment (eight rows) per column and plane
    uint4 h, l;
    h.x = split_bf16_pair(e0[0], e0[1], l.x); h.y = split_bf16_pair(e0[2], e0[3], l.y);
    h.z = split_bf16_pair(e0[4], e0[5], l.z); h.w = split_bf16_pair(e0[6], e0[7], l.w);
    *reinterpret_cast<uint4*>(base + (2 * gp) * TP + go * 16) = h;
    *reinterpret_cast<uint4*>(base + GT_PLANE + (2 * gp) * TP + go * 16) = l;
    uint4 h1, l1;
    h1.x = split_bf16_pair(e1[0], e1[1], l1.x); h1.y = split_bf16_pair(e1[2], e1[3], l1.y);
    h1.z = split_bf16_pair(e1[4], e1[5], l1.z); h1.w = split_bf16_pair(e1[6], e1[7], l1.w);
    *reinterpret_cast<uint4*>(base + (2 * gp + 1) * TP + go * 16) = h1;
    *reinterpret_cast<uint4*>(base + GT_PLANE + (2 * gp + 1) * TP + go * 16) = l1;
    // GR: per row the column pair as one dword per plane (the same hi / lo values regrouped by row: one v_perm per dword)
    unsigned char* gr = base + 2 * GT_PLANE + (8 * go) * RP + gp * 4;
    const unsigned hq[4] = {h.x, h.y, h.z, h.w}, h1q[4] = {h1.x, h1.y, h1.z, h1.w}, lq[4] = {l.x, l.y, l.z, l.w}, l1q[4] = {l1.x, l1.y, l1.z, l1.w};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const unsigned sel = (i & 1) ? 0x07060302u : 0x05040100u;  // row i of column 2 gp in the low half, of column 2 gp + 1 in the high half
      *reinterpret_cast<unsigned*>(gr + i * RP) = __builtin_amdgcn_perm(h1q[i >> 1], hq[i >> 1], sel);
      *reinterpret_cast<unsigned*>(gr + GR_PLANE + i * RP) = __builtin_amdgcn_perm(l1q[i >> 1], lq[i >> 1], sel);
    }
  };
#endif
  auto y_request = [&](long long r0, int k) {
    f32x2 (&gv)[8] = gvA;
    const float* yb = a.x + (r0 + 32 * k) * a.ldx;
#pragma unroll
    for (int i = 0; i < 8; ++i) gv[i] = *reinterpret_cast<const f32x2*>(yb + (long long)i * a.ldx + y_toff0);
  };
  auto y_stage = [&](long long r0, int k, int buf) {
    const f32x2 (&gv)[8] = gvA;
    unsigned char* base = ybuf + buf * YT_BUF + (yo0 + 4 * k) * 16;
    f32x2 st[8];
    if (a.ln_stats) {
      const float* sb = a.ln_stats + 2 * (r0 + 32 * k);
#pragma unroll
      for (int i = 0; i < 8; ++i) st[i] = *reinterpret_cast<const f32x2*>(sb + 2 * (8 * yo0 + i));
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      float e[8];
#pragma unroll
      for (int i = 0; i < 8; ++i) e[i] = a.ln_stats ? (gv[i][c] - st[i][0]) * (st[i][1] * lg[c]) : gv[i][c];
      uint4 h, l;
      h.x = split_bf16_pair(e[0], e[1], l.x); h.y = split_bf16_pair(e[2], e[3], l.y);
      h.z = split_bf16_pair(e[4], e[5], l.z); h.w = split_bf16_pair(e[6], e[7], l.w);
      *reinterpret_cast<uint4*>(base + (2 * ycp + c) * TP) = h;
      *reinterpret_cast<uint4*>(base + YT_PLANE + (2 * ycp + c) * TP) = l;
    }
  };

  // ---------------------------------------------------------------- accumulators
  // Weight gradient: a piece's 96 x 64 block is six 32 x 32 tiles t = (column fragment j = t % 3, channel fragment c = t / 3); in piece p wave w
  // works on tile t = (w - p) mod 8 (two waves sit out): every piece costs a wave at most ONE tile (12 MFMAs) instead of costing two owner waves 36 each
  // with six waiting at the barrier, and over a chunk every wave meets each of its six accumulators once (accumulator t of wave w = piece (w - t) mod 8).
  f32x16 dw[6];
#pragma unroll
  for (int t = 0; t < 6; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) dw[t][r] = 0.f;
  // Data gradient: the chunk's 64 x 64 tile of gy^T as four 32 x 32 tiles (rt = rows, ct = channels), each shared by the two wave groups, which split
  // a piece's six k16 steps (kh = group: steps 3 kh .. 3 kh + 2); the halves meet in LDS when the chunk's eight pieces are done.  v_mfma_f32_32x32x16_bf16
  // with W^T fragments straight from L1 / L2 as the A operand: 9 MFMAs on one accumulator (back-to-back issue) and 6 ds_read_b128 per piece and wave,
  // where sixteen 16 x 16 accumulators cost 18 MFMAs in two dependent chains and 12 reads.
  const int rt = wave & 1, ct = (wave >> 1) & 1;
  const uint4* wq = reinterpret_cast<const uint4*>(a.wfrag);
  constexpr int KS = NQ / 16;  // k16 planes per column tile of the fmt-2 weights
  // W^T fragment of k16 step ks for channels ct * 32 .. + 31: fmt-2 plane (nt = ct, ks), lane = (8-column group) * 32 + channel % 32
  const uint4* wbase = wq + ((long long)(ct * KS + 3 * GRP) * 2) * 64 + lane;  // + ((piece * 6 + s) * 2 + plane) * 64
  uint4 wf[3][2];
  auto w_request = [&](int piece) {
#pragma unroll
    for (int sx = 0; sx < 3; ++sx) {
      wf[sx][0] = wbase[(long long)((piece * 6 + sx) * 2) * 64];
      wf[sx][1] = wbase[(long long)((piece * 6 + sx) * 2 + 1) * 64];
    }
  };
  float* const gscr = reinterpret_cast<float*>(sm + LDS_BYTES);  // [4 tiles][4 register quads][64 lanes] x 16 bytes: group 1's halves of gy^T

  // ---------------------------------------------------------------- prologue: piece 0 of chunk 0 staged, piece 1 requested; y of chunk 0
  if (GROLE) {
    g_request(r_begin, 0, gvG);
    g_stage(0, gvG);
  } else {
    y_request(r_begin, 0);
    y_stage(r_begin, 0, 0);
    y_request(r_begin, 1);
    y_stage(r_begin, 1, 0);
  }
  // (The requests go out in the order in which the loop leaves them pending at its top -- group 0: the next piece, then the fragments; group 1: the
  // fragments, then the next piece -- : the compiler sizes the waits inside the loop for the worst path into it, and a prologue that leaves the loads
  // in another order tightens every wait of the steady state.  Compiler-level memory barriers between them: a scheduling barrier alone does not keep
  // two runs of loads from being merged.)
  asm volatile("" ::: "memory");
  if (GRP == 1) w_request(0);
  asm volatile("" ::: "memory");
  if (GROLE) g_request(r_begin, 1, gvG);
  asm volatile("" ::: "memory");
  if (GRP == 0) w_request(0);
  asm volatile("" ::: "memory");
  __syncthreads();
  float dgam[16];  // LNB, group 0: sum over this lane's rows of gy * xhat for its sixteen channels
#pragma unroll
  for (int r = 0; r < 16; ++r) dgam[r] = 0.f;
  float* const lsum = reinterpret_cast<float*>(sm + LDS_BYTES + GY_SCRATCH);  // LNB: [group][ct][64 rows] (sum g, sum g xhat) of a wave's half tile
  // LNB: the lane's 32-bit offsets into the chunk's rows (wave-uniform chunk base + ONE vector offset per array; made opaque once per chunk, else the
  // compiler forms a 64-bit vector address per 16-byte piece outside the loops and spills them)
  unsigned lx_off = (unsigned)((rt * 32 + l31) * a.ldx + ct * 32 + 4 * half), ld_off = (unsigned)((rt * 32 + l31) * a.lddx + ct * 32 + 4 * half);
  unsigned lg_off = (unsigned)(ct * 32 + 4 * half), ls_off = (unsigned)(2 * (rt * 32 + l31));

  for (int ch = 0; ch < n_ch; ++ch) {
    const long long r0 = r_begin + (long long)ch * CH;
    const bool more_ch = ch + 1 < n_ch;
    f32x16 gyacc;  // gy^T{channel ct * 32 + (r & 3) + 8 (r >> 2) + 4 half, row rt * 32 + l31}: this group's half of the sum over the 768 columns
#pragma unroll
    for (int r = 0; r < 16; ++r) gyacc[r] = 0.f;
    const unsigned char* yb = ybuf + (ch & 1) * YT_BUF;
    f32x2 st_ln;   // LNB: (mean, rstd) of this lane's row (xhat is formed twice, a quad at a time: sixteen registers across the barrier did not fit)

    // One loader step of piece p: the next piece (in registers since the previous step) goes to the other buffer (read last in iteration p - 1: the
    // barrier that ended it makes the buffer free for both groups) and the one after it is requested; the y loaders stage the next chunk's rows.
    // One loader step of piece p: piece p + 1 (in registers since the previous step) goes to the other buffer (read last in iteration p - 1: the
    // barrier that ended it makes the buffer free for both groups) and piece p + 2 is requested into the same registers.  (Two pieces in flight per
    // thread were measured: no faster, 16 registers more.)  The y loaders stage the next chunk's rows one item at a time.  The g request is
    // unconditional -- past the last piece it repeats rows of the current chunk and is never staged: a skipped request would give the compiler a path
    // with fewer loads behind the ones it waits for, i.e. a tighter wait on every path.
    auto loader_step = [&](int p) {
      const bool last_piece = p == NPIECE - 1;
      if (GROLE) {
        // (the registers are pinned HERE: the splits are plain vector arithmetic, which the scheduler otherwise lifts across the barrier into the
        // previous step -- and with them the wait for these rows, a whole matrix phase early)
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(gvG[i]));
        if (!last_piece || more_ch) g_stage((p & 1) ^ 1, gvG);
        __builtin_amdgcn_sched_barrier(0);  // (the request stays behind the staging)
        const bool next_chunk = p + 2 >= NPIECE;
        g_request(next_chunk && more_ch ? r0 + CH : r0, (p + 2) % NPIECE, gvG);
      } else if (more_ch) {
        if (p == 2) y_stage(r0 + CH, 0, (ch + 1) & 1);
        if (p == 5) y_stage(r0 + CH, 1, (ch + 1) & 1);
        if (p == 0) y_request(r0 + CH, 0);
        if (p == 3) y_request(r0 + CH, 1);
      }
    };
    auto piece_step = [&](int p) {
      const unsigned char* pb = sm + (p & 1) * PIECE_BUF;  // (NPIECE is even: piece p of every chunk lives in buffer p & 1)
      // (the lanes' 32-bit offsets are made opaque once per iteration: as loop invariants the compiler adds them to every wave-uniform row base
      // outside the loop -- one 64-bit vector-register address per load -- instead of using the scalar-base + vector-offset addressing mode)
      asm volatile("" : "+v"(g_toff), "+v"(y_toff0));
      if (GRP == 0) {
        loader_step(p);
        __builtin_amdgcn_sched_barrier(0);
      }
      // ---- weight gradient: this wave's tile of the piece (12 MFMAs; fragments of step s + 1 requested before the products of step s)
      auto wg_tile = [&](f32x16& acc, int j, int c) {
        const unsigned char* bp = yb + (c * 32 + l31) * TP + half * 16;
        const unsigned char* ap = pb + (j * 32 + l31) * TP + half * 16;
        uint4 F[2][4];  // [step parity][A hi, A lo, B hi, B lo]
        auto rd = [&](int sx, uint4 (&f)[4]) {
          f[0] = *reinterpret_cast<const uint4*>(ap + sx * 32);
          f[1] = *reinterpret_cast<const uint4*>(ap + sx * 32 + GT_PLANE);
          f[2] = *reinterpret_cast<const uint4*>(bp + sx * 32);
          f[3] = *reinterpret_cast<const uint4*>(bp + sx * 32 + YT_PLANE);
        };
        rd(0, F[0]);
#pragma unroll
        for (int sx = 0; sx < 4; ++sx) {
          if (sx < 3) rd(sx + 1, F[(sx + 1) & 1]);
          const bf16x8 Ah = __builtin_bit_cast(bf16x8, F[sx & 1][0]), Al = __builtin_bit_cast(bf16x8, F[sx & 1][1]);
          const bf16x8 Bh = __builtin_bit_cast(bf16x8, F[sx & 1][2]), Bl = __builtin_bit_cast(bf16x8, F[sx & 1][3]);
          if constexpr (!VMM_SINGLE_PASS) {
            acc = vmm_mfma16(Ah, Bl, acc);
            acc = vmm_mfma16(Al, Bh, acc);
          }
          acc = vmm_mfma16(Ah, Bh, acc);
        }
      };
      if (!(VMM_QB_SKIP & 1)) {
        switch ((wave - p) & 7) {  // (wave-uniform: a scalar branch; the accumulator of every case is static)
          case 0: wg_tile(dw[0], 0, 0); break;
          case 1: wg_tile(dw[1], 1, 0); break;
          case 2: wg_tile(dw[2], 2, 0); break;
          case 3: wg_tile(dw[3], 0, 1); break;
          case 4: wg_tile(dw[4], 1, 1); break;
          case 5: wg_tile(dw[5], 2, 1); break;
          default: break;
        }
      }
      // ---- data gradient: gy^T[channel][row] += W^T[channel][n] g^T[n][row] over this group's three k16 steps of the piece
      if (!(VMM_QB_SKIP & 2)) {
        const unsigned char* gp_ = pb + 2 * GT_PLANE + (rt * 32 + l31) * RP + (3 * GRP * 16 + half * 8) * 2;
        uint4 G[2][2];  // (two steps of fragments in flight)
        G[0][0] = *reinterpret_cast<const uint4*>(gp_);
        G[0][1] = *reinterpret_cast<const uint4*>(gp_ + GR_PLANE);
        G[1][0] = *reinterpret_cast<const uint4*>(gp_ + 32);
        G[1][1] = *reinterpret_cast<const uint4*>(gp_ + 32 + GR_PLANE);
#pragma unroll
        for (int sx = 0; sx < 3; ++sx) {
          const bf16x8 Wh = __builtin_bit_cast(bf16x8, wf[sx][0]), Wl = __builtin_bit_cast(bf16x8, wf[sx][1]);
          const bf16x8 Gh = __builtin_bit_cast(bf16x8, G[sx & 1][0]), Gl = __builtin_bit_cast(bf16x8, G[sx & 1][1]);
          if (sx == 0) {
            G[0][0] = *reinterpret_cast<const uint4*>(gp_ + 64);
            G[0][1] = *reinterpret_cast<const uint4*>(gp_ + 64 + GR_PLANE);
          }
          if constexpr (!VMM_SINGLE_PASS) {
            gyacc = vmm_mfma16(Wh, Gl, gyacc);
            gyacc = vmm_mfma16(Wl, Gh, gyacc);
          }
          gyacc = vmm_mfma16(Wh, Gh, gyacc);
        }
      }
      if (LNB && p == NPIECE - 1) {
        // LayerNorm backward, first half (every wave, on its OWN half tile -- the row sums are linear in gy): xhat of this lane's row and sixteen
        // channels, the half tile's contribution to (sum_c g, sum_c g xhat), g = gamma gy; the four contributions of a row meet in LDS behind the barrier
        asm volatile("" : "+v"(lx_off), "+v"(ld_off), "+v"(lg_off), "+v"(ls_off));
        const float* xb = a.x + r0 * a.ldx;  // wave-uniform
        st_ln = *reinterpret_cast<const f32x2*>(a.ln_stats + 2 * r0 + ls_off);
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const f32x4 xv = *reinterpret_cast<const f32x4*>(xb + lx_off + 8 * q);
          const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_gamma + lg_off + 8 * q);
          const f32x4 xq = {(xv.x - st_ln[0]) * st_ln[1], (xv.y - st_ln[0]) * st_ln[1], (xv.z - st_ln[0]) * st_ln[1], (xv.w - st_ln[0]) * st_ln[1]};
          const float g0 = gyacc[4 * q] * gm.x, g1 = gyacc[4 * q + 1] * gm.y, g2 = gyacc[4 * q + 2] * gm.z, g3 = gyacc[4 * q + 3] * gm.w;
          s1 += (g0 + g1) + (g2 + g3);
          s2 += (g0 * xq.x + g1 * xq.y) + (g2 * xq.z + g3 * xq.w);
        }
        s1 += lane_xor(s1, 5);
        s2 += lane_xor(s2, 5);
        if (half == 0) *reinterpret_cast<f32x2*>(lsum + ((GRP * 2 + ct) * 64 + rt * 32 + l31) * 2) = f32x2{s1, s2};
      }
      if (GRP == 1 && p == NPIECE - 1) {  // this group's half of the chunk's gy^T tile: read by group 0 behind this piece's barrier
#pragma unroll
        for (int q = 0; q < 4; ++q)
          *reinterpret_cast<f32x4*>(gscr + (((wave & 3) * 4 + q) * 64 + lane) * 4) = f32x4{gyacc[4 * q], gyacc[4 * q + 1], gyacc[4 * q + 2], gyacc[4 * q + 3]};
      }
      // the W^T fragments of the next piece: behind this piece's data-gradient products (they overwrite the fragment registers)
      __builtin_amdgcn_sched_barrier(0);
      w_request((p + 1) % NPIECE);
      if (GRP == 1) {
        __builtin_amdgcn_sched_barrier(0);
        loader_step(p);
      }
      __syncthreads();
    };
#pragma unroll 1
    for (int p = 0; p < NPIECE; ++p) piece_step(p);
    // ---- the chunk's data gradient: group 0 adds group 1's half (fixed order) and stores; lane = row rt * 32 + l31, register quad q = channels
    // ct * 32 + 8 q + 4 half .. + 3.  (Group 1 writes its next half seven barriers from here.)
    if (GRP == 0 && !LNB) {
      float* gyr = a.gy + (r0 + rt * 32 + l31) * a.ldgy + ct * 32 + 4 * half;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(gscr + (((wave & 3) * 4 + q) * 64 + lane) * 4);
        *reinterpret_cast<f32x4*>(gyr + 8 * q) = f32x4{gyacc[4 * q] + o.x, gyacc[4 * q + 1] + o.y, gyacc[4 * q + 2] + o.z, gyacc[4 * q + 3] + o.w};
      }
    }
    if (GRP == 0 && LNB) {
      // LayerNorm backward, second half: dx = rstd (g - mean_c g - xhat mean_c (g xhat)) (vddp.py:245-254 differentiated), added to the gradient that
      // reached the block input past the attention (the residual) when the caller says so; dgamma_c += gy_c xhat_c (this lane's row)
      float a1 = 0.f, a2 = 0.f;
#pragma unroll
      for (int k = 0; k < 4; ++k) {  // (fixed order: group 0 ct 0, ct 1, group 1 ct 0, ct 1)
        const f32x2 v = *reinterpret_cast<const f32x2*>(lsum + (k * 64 + rt * 32 + l31) * 2);
        a1 += v[0];
        a2 += v[1];
      }
      a1 *= 1.0f / CC;
      a2 *= 1.0f / CC;
      float* dxr = a.dx + r0 * a.lddx + ld_off;  // (wave-uniform base + the lane's offset)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const f32x4 o = *reinterpret_cast<const f32x4*>(gscr + (((wave & 3) * 4 + q) * 64 + lane) * 4);
        const f32x4 gm = *reinterpret_cast<const f32x4*>(a.ln_gamma + lg_off + 8 * q);
        const f32x4 xv = *reinterpret_cast<const f32x4*>(a.x + r0 * a.ldx + lx_off + 8 * q);
        const f32x4 xq = {(xv.x - st_ln[0]) * st_ln[1], (xv.y - st_ln[0]) * st_ln[1], (xv.z - st_ln[0]) * st_ln[1], (xv.w - st_ln[0]) * st_ln[1]};
        const float gy0 = gyacc[4 * q] + o.x, gy1 = gyacc[4 * q + 1] + o.y, gy2 = gyacc[4 * q + 2] + o.z, gy3 = gyacc[4 * q + 3] + o.w;
        dgam[4 * q] += gy0 * xq.x; dgam[4 * q + 1] += gy1 * xq.y; dgam[4 * q + 2] += gy2 * xq.z; dgam[4 * q + 3] += gy3 * xq.w;
        f32x4 d = {st_ln[1] * (gy0 * gm.x - a1 - xq.x * a2), st_ln[1] * (gy1 * gm.y - a1 - xq.y * a2),
                   st_ln[1] * (gy2 * gm.z - a1 - xq.z * a2), st_ln[1] * (gy3 * gm.w - a1 - xq.w * a2)};
        if (a.accumulate) d += *reinterpret_cast<const f32x4*>(dxr + 8 * q);
        *reinterpret_cast<f32x4*>(dxr + 8 * q) = d;
      }
    }
  }

  // ---------------------------------------------------------------- the workgroup's partial weight-gradient block: [wave][f][c][q][lane] x 4 floats
  f32x4* dst = reinterpret_cast<f32x4*>(a.part) + (long long)blockIdx.x * (PART_FLOATS / 4);
  if (LNB && GRP == 0) {  // the gamma gradient of this workgroup: sum over the 32 rows (lanes) of a half, one partial row per (workgroup, rt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float v = dgam[r];
#pragma unroll
      for (int bit = 4; bit >= 0; --bit) v += lane_xor(v, bit);
      dgam[r] = v;
    }
    if (l31 == 0) {
      float* pg = a.dgamma_part + ((long long)blockIdx.x * 2 + rt) * CC + ct * 32 + 4 * half;
#pragma unroll
      for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(pg + 8 * q) = f32x4{dgam[4 * q], dgam[4 * q + 1], dgam[4 * q + 2], dgam[4 * q + 3]};
    }
  }
#pragma unroll
  for (int t = 0; t < 6; ++t) {
    const int pc = (wave - t) & 7, j = t % 3, c = t / 3;  // accumulator t of this wave: piece pc, column fragment j, channel fragment c
#pragma unroll
    for (int q = 0; q < 4; ++q)
      dst[(((pc * 3 + j) * 2 + c) * 4 + q) * 64 + lane] = f32x4{dw[t][4 * q], dw[t][4 * q + 1], dw[t][4 * q + 2], dw[t][4 * q + 3]};
  }
}

template <bool LNB>
__global__ __launch_bounds__(512) void qkv_bwd_x3_kernel(const QBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  if (a.nchunks - (int)blockIdx.x * a.chunks_per_wg <= 0) return;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  if (wave < 4) {
    if ((wave & 3) != 3) qkv_bwd_body<0, true, LNB>(a, sm); else qkv_bwd_body<0, false, LNB>(a, sm);
  } else {
    if ((wave & 3) != 3) qkv_bwd_body<1, true, LNB>(a, sm); else qkv_bwd_body<1, false, LNB>(a, sm);
  }
}

// dw_packed[c][n] += sum over workgroups of the partial blocks (fixed order).  thread = (16-byte piece, slice lane)
__global__ __launch_bounds__(256) void qkv_bwd_reduce_kernel(const float* __restrict__ part, int nz, float* __restrict__ dwp) {
  __shared__ f32x4 red[8][32];
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int piece = blockIdx.x * 32 + e;  // (((wave * 3 + f) * 2 + c) * 4 + q) * 64 + lane
  const f32x4* src = reinterpret_cast<const f32x4*>(part) + piece;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = zl; z < nz; z += 8) s += src[(long long)z * (PART_FLOATS / 4)];
  red[zl][e] = s;
  __syncthreads();
  if (zl == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][e];
    const int lane = piece & 63, q = (piece >> 6) & 3, c = (piece >> 8) & 1, wf = piece >> 9;  // wf = wave * 3 + f
    const int wave = wf / 3, f = wf - wave * 3;
    // accumulator register 4 q + k of the lane: row (column n of W) = (3 wave + f) * 32 + k + 8 q + 4 (lane >> 5), column (channel) = c * 32 + (lane & 31)
    const int n = (wave * 3 + f) * 32 + 8 * q + 4 * (lane >> 5), cc = c * 32 + (lane & 31);
    f32x4* o = reinterpret_cast<f32x4*>(dwp + (long long)cc * NQ + n);
    *o += s;
  }
}

}  // namespace

#if !VMM_SINGLE_PASS
extern "C" int64_t vmm_qkv_bwd_workspace(int64_t rows, int32_t C, int32_t Nq) {
  if (C != CC || Nq != NQ || rows <= 0 || rows % CH) return 0;
  const long long nchunks = (rows + CH - 1) / CH;
  const long long nwg = nchunks < 256 ? nchunks : 256;
  return nwg * (PART_FLOATS + 2 * CC);  // partial weight-gradient blocks + (vmm_qkv_bwd_ln_*) two partial rows of the gamma gradient per workgroup
}

// gy = g W (rows x 64, plain store) and dw_packed[c][n] += g^T y in one pass over g (rows x 768).  x / ln_stats / ln_gamma: y = x when ln_stats is
// NULL, else y = (x - mean) rstd gamma with (mean, rstd) = ln_stats[r][2].  w_frag = vmm_pack_weights fmt 2 of the (K = 768, N = 64) operand
// (the to_qkv weight (768, 64) as it lies in torch).  Returns 1 (nothing launched) unless C == 64, Nq == 768 and rows is a multiple of 64.
//
// vmm_qkv_bwd_ln_*: the same pass with the backward of the PreNorm LayerNorm (vddp.py:245-264) as its epilogue -- gy never reaches memory:
//   dx (=|+=, `accumulate`) rstd (gamma gy - mean_c(gamma gy) - xhat mean_c(gamma gy xhat)),   dgamma[c] += sum_rows gy[r][c] xhat[r][c]
// (ln_stats and ln_gamma required; dgamma leaves as two partial rows per workgroup + the fixed-order vmm_sum_partials).  Replaces
// vmm_channel_layernorm_bwd behind the fused to_qkv backward: one 16-byte-per-element sweep and a launch less per attention block.
#endif
namespace {
int qkv_bwd_launch(QBArgs& a, float* dw_packed, float* workspace, float* dgamma, vmm_stream_t stream) {
  a.nchunks = (int)((a.rows + CH - 1) / CH);
  const int nwg = a.nchunks < 256 ? a.nchunks : 256;
  a.chunks_per_wg = (a.nchunks + nwg - 1) / nwg;
  const int gx = (a.nchunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  a.part = workspace;
  a.dgamma_part = workspace + (long long)nwg * PART_FLOATS;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_bwd_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&qkv_bwd_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  if (a.dx) hipLaunchKernelGGL(qkv_bwd_x3_kernel<true>, dim3(gx), dim3(512), LDS_BYTES + GY_SCRATCH + LN_SUMS, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(qkv_bwd_x3_kernel<false>, dim3(gx), dim3(512), LDS_BYTES + GY_SCRATCH, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(qkv_bwd_reduce_kernel, dim3(PART_FLOATS / 4 / 32), dim3(256), 0, (hipStream_t)stream, workspace, gx, dw_packed);
  VMM_LAUNCH_CHECK();
  if (a.dx && dgamma) return vmm_sum_partials(a.dgamma_part, 2 * gx, CC, CC, dgamma, stream);
  return 0;
}
}  // namespace

#if VMM_SINGLE_PASS  // (also in -DVMM_DQKV16=0 measurement builds: the library exports the same symbols)
// Rows of the 16-bit qkv-row gradient widened to fp32 (exact): for the shapes this file's one-pass kernel does not take (rows no multiple of 64 -- small
// test geometries), where the plan runs the separate weight- and data-gradient launches, which read fp32 rows and round them to the same 16 bits again.
namespace {
__global__ __launch_bounds__(256) void dqkv_widen_kernel(const uint4* __restrict__ src, f32x4* __restrict__ dst, long long n8) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n8; i += (long long)gridDim.x * 256) {
    const uint4 u = src[i];
    const unsigned w[4] = {u.x, u.y, u.z, u.w};
    float f[8];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#if VMM_FP16_OPERANDS
      f[2 * k] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] & 0xffffu));
      f[2 * k + 1] = (float)__builtin_bit_cast(_Float16, (unsigned short)(w[k] >> 16));
#else
      f[2 * k] = __uint_as_float(w[k] << 16);
      f[2 * k + 1] = __uint_as_float(w[k] & 0xffff0000u);
#endif
    }
    dst[2 * i] = f32x4{f[0], f[1], f[2], f[3]};
    dst[2 * i + 1] = f32x4{f[4], f[5], f[6], f[7]};
  }
}
}  // namespace
extern "C" int VMM_X3(vmm_dqkv_widen_, )(const void* src, float* dst, int64_t n, vmm_stream_t stream) {
  if (n & 7) return 1;
  if (n <= 0) return 0;
  const long long n8 = n / 8;
  const int blocks = (int)((n8 + 255) / 256 < 2048 ? (n8 + 255) / 256 : 2048);
  hipLaunchKernelGGL(dqkv_widen_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const uint4*>(src), reinterpret_cast<f32x4*>(dst), n8);
  VMM_LAUNCH_CHECK();
  return 0;
}
#endif

extern "C" int VMM_X3(vmm_qkv_bwd_, )(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg, const float* w_frag,
                                  float* gy, int32_t ldgy, float* dw_packed, float* workspace, int64_t rows, int32_t C, int32_t Nq, vmm_stream_t stream) {
  if (C != CC || Nq != NQ || !workspace || (rows % CH) || (ldx & 1) || (ldg & 1) || (ldgy & 3) || (ln_stats && !ln_gamma)) return 1;
  if (rows <= 0) return 0;
  QBArgs a;
  a.x = x; a.ldx = ldx; a.ln_stats = ln_stats; a.ln_gamma = ln_gamma; a.g = reinterpret_cast<const vmm_dqkv_t*>(g); a.ldg = ldg;
  a.wfrag = reinterpret_cast<const unsigned char*>(w_frag);
  a.gy = gy; a.ldgy = ldgy; a.rows = rows;
  a.dx = nullptr; a.lddx = 0; a.accumulate = 0;
  return qkv_bwd_launch(a, dw_packed, workspace, nullptr, stream);
}

extern "C" int VMM_X3(vmm_qkv_bwd_ln_, )(const float* x, int32_t ldx, const float* ln_stats, const float* ln_gamma, const float* g, int32_t ldg,
                                     const float* w_frag, float* dx, int32_t lddx, int32_t accumulate, float* dgamma, float* dw_packed, float* workspace,
                                     int64_t rows, int32_t C, int32_t Nq, vmm_stream_t stream) {
  if (C != CC || Nq != NQ || !workspace || (rows % CH) || (ldx & 3) || (ldg & 1) || (lddx & 3) || !ln_stats || !ln_gamma || !dx) return 1;
  if (rows <= 0) return 0;
  QBArgs a;
  a.x = x; a.ldx = ldx; a.ln_stats = ln_stats; a.ln_gamma = ln_gamma; a.g = reinterpret_cast<const vmm_dqkv_t*>(g); a.ldg = ldg;
  a.wfrag = reinterpret_cast<const unsigned char*>(w_frag);
  a.gy = nullptr; a.ldgy = 0; a.rows = rows;
  a.dx = dx; a.lddx = lddx; a.accumulate = accumulate;
  return qkv_bwd_launch(a, dw_packed, workspace, dgamma, stream);
}
