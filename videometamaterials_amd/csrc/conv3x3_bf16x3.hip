// 3x3 stride-1 "same" convolution with an LDS-resident halo patch and register-fed weights, split-bf16 MFMA, gfx950.
//
// The ResnetBlock projections (vddp.py:268-285, Conv3d (1,3,3) pad (0,1,1)) are 60 % of the denoiser's flops.  The generic
// implicit-GEMM kernel gathers every input element nine times (once per tap), stages the weights through LDS and synchronises
// the workgroup once per K chunk.  Here, per 32-channel chunk:
//   A  the tile's pixels plus a one-pixel halo are staged ONCE in LDS as bf16 hi | lo; the nine taps are nine row offsets into
//      that patch.  Big frames use 2-D pixel tiles (TH x 16 pixels, patch (TH+2) x 18, 27-41 % halo, out-of-image patch rows are
//      zeros); small frames use flat row tiles over the whole [frame][h][w] row space (patch = tile +- (W+1) rows, no partial
//      tiles at frame ends) with per-pixel tap masks for the image borders.
//   B  weights come pre-split and pre-arranged in MFMA fragment order (vmm_pack_weights fmt 2): one 32-column x 16-k fragment
//      plane is 1 KB contiguous, so each wave loads its B operands with fully coalesced 16-byte-per-lane global loads straight
//      into registers, one k16 step ahead of use.  No LDS traffic, no LDS footprint and NO workgroup barrier for the weights:
//      the only barriers are the two around the patch refresh at a channel-chunk boundary (every 18 k16 steps).
// LDS traffic per MFMA drops to the A fragments only (1/3 of the LDS read bandwidth at full MFMA rate), the weights ride the
// vector-memory/L1 path in parallel, and at <= 52 KB LDS / <= 256 VGPRs two workgroups share a CU so that one computes while the
// other refreshes its patch or drains its epilogue.
//
// Wave tile 64 rows x 64 columns (2 x 2 MFMA 32x32x16 tiles, 3 passes: lo*hi + hi*lo + hi*hi, see igemm_bf16x3.hip); workgroup =
// 4 waves as 4 x 1 (256 pixels x 64 columns, Cout == 64) or 2 x 2 (128 pixels x 128 columns).  Few-row layers (12 x 12 level) split
// the channel chunks over blockIdx.y and add their partial sums in a fixed (ticketed) order.  Fused input transform (GroupNorm * FiLM -> SiLU, a_mode 1), bias,
// two-source channel concat and the residual add are the same as in the generic kernel.
#include <stdio.h>
#include <stdlib.h>
#include <type_traits>
#include <vector>
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

// VMM_EXPERIMENTS (python -m videometamaterials_amd.build with VMM_EXPERIMENTS=1 in the environment -> libvmm_hip_exp.so): the kernels that were built,
// are parity-green and measured SLOWER inside the captured step -- the persistent wave-specialised kernel (LABNOTES 7.3, 7.8) and the 32-column-wave-tile
// instance at three workgroups per CU (LABNOTES 10.1), the balanced ("stream-K") launch of the few-tile layers (LABNOTES 10.4: faster launch by launch on the
// layers it was made for, no faster inside the captured step).  The product library contains none of them.
#ifndef VMM_EXPERIMENTS
#define VMM_EXPERIMENTS 0
#endif
#ifndef VMM_C3_INTERLEAVE
#define VMM_C3_INTERLEAVE 1
#endif
constexpr bool C3_INTERLEAVE = VMM_C3_INTERLEAVE;  // memory requests of step q + 1 between the MFMAs of step q (-DVMM_C3_INTERLEAVE=0: A/B builds)
#ifndef VMM_C3_XCD_ORDER
#define VMM_C3_XCD_ORDER 1
#endif
constexpr bool C3_XCD_ORDER = VMM_C3_XCD_ORDER;  // XCD-contiguous tile numbering (-DVMM_C3_XCD_ORDER=0: A/B builds)
constexpr int CK = 32;    // channels per chunk
#ifndef VMM_C3_ONE_LO
#define VMM_C3_ONE_LO 0
#endif
// Measurement build (-DVMM_C3_PAIR=1; built, parity-green, 2-4 % SLOWER: LABNOTES 11.7): the single-pass unsplit 3 x 3 instances stage channel chunks in PAIRS -- chunk c in
// the hi half of a patch row, chunk c + 1 in the (otherwise dead) lo half -- so that both patches of a pair are requested together (one exposed HBM round trip per 64 channels
// instead of two, half the chunk boundaries and barriers) and the prefetch of the next pair has 36 instead of 18 k16 steps to land.
#ifndef VMM_C3_PAIR
#define VMM_C3_PAIR 0
#endif
constexpr int CROW = 72;  // LDS patch row pitch in bf16: 32 hi | 32 lo | 8 pad = 144 bytes (9 x 16 B: ds_read_b128 over consecutive rows is conflict-free)

struct C3Args {
  vmm_conv_desc p;
  int n_tiles;                   // column tiles
  int mode;                      // 0: flat row tiles across frames, 1: 2-D TH x 16 pixel tiles
  int tiles_x, tiles_per_frame;  // 2-D mode
  int PR;                        // patch rows
  int pitch;                     // patch rows per image row step (2-D: 18, flat: W)
  int halo;                      // flat mode: W + 1
  int KS;                        // k16 steps per 32-column tile in the packed weights
  int chunks_per_split;
  int total_rows;                // nimg * H * W
  int cps_shift;                 // TS == 2: log2(channel chunks per sub-pixel)
  unsigned tpf_magic, tx_magic;  // floor(2^32 / d) + 1 for d = tiles_per_frame, tiles_x: n / d = mulhi(n, magic) for n d < 2^32 -- a run-time
                                 // division is expanded on the VECTOR unit (v_rcp_iflag) and leaves the wave-uniform tile coordinates in VGPRs
  int sk_tiles = 0;              // balanced launch: output tiles of the launch (row tiles x column tiles)
  int gn_fine = 0;               // 2-D tiles of 16 pixel rows writing GroupNorm slots in the 8-pixel-row layout of the 32-column-wave-tile instance (slot 2k: sums, 2k + 1: zeros)
  unsigned long long* trace = nullptr;  // VMM_C3_TRACE=<launch>: wave 0 of every workgroup stamps s_memtime at its phase boundaries (16 slots per workgroup)
};

__device__ __forceinline__ void split2c(float x0, float x1, unsigned& hi, unsigned& lo) { hi = split_bf16_pair(x0, x1, lo); }

// TS ("tap subsets"): the stride-2 resampling layers as 3 x 3 neighbourhood convolutions of which every unit uses a 2 x 2 corner:
//   TS == 1  ConvTranspose (1,4,4) stride 2 pad 1 (Upsample, vddp.py:155): output pixel (2y + py, 2x + px) is a 2 x 2 convolution over the
//            input neighbourhood rows {y - 1 + py, y + py}, i.e. the four output phases are 4 x Cout output columns of ONE 3 x 3
//            convolution over the input tile, each phase with its own corner of the nine taps (weights: vmm_pack_weights fmt 6);
//   TS == 2  Conv (1,4,4) stride 2 pad 1 (Downsample, vddp.py:158) over 2 x 2 input cells: a 3 x 3 convolution over the cell grid with
//            4 x Cin cell channels; the channels of sub-pixel (sy, sx) meet the taps of the corner (1 - sy, 1 - sx) only (fmt 5).  The
//            tile space (p.Hin x p.Win, patch rows) is the CELL grid, the source image is twice as large.
// Eight k16 steps per chunk instead of eighteen; every input element is still staged once (the generic kernel gathers it four times).
// ONE: the "bf16" throughput mode (BASELINE.json configs[3]): the operands' hi planes only, ONE MFMA pass per product (the lo planes of the patch and of
// the fmt-2 weights are neither read nor multiplied): bf16-rounded operands, fp32 accumulation.
// A16 (single-pass mode only): storage of the feature maps -- 0: fp32 in, fp32 out; 1: bf16 in, bf16 out; 2: bf16 in, fp32 out; 3: fp32 in, bf16 out
// (the level boundaries of the resampling layers).  bf16 in: a patch item is 8 bytes, and without the fused transform the stored bits ARE the hi
// plane (no conversion at all between HBM and the LDS patch); bf16 out: the epilogue rounds once, the GroupNorm sums come from the fp32 accumulators.
// NJ: 32-column MFMA tiles per wave -- 2: the 64 x 64 wave tile; 1 (unsplit bf16x3 layers with 64 output channels on 2-D tiles): a 64-pixel x 32-column
// wave tile, 2 x 2 waves = 128 pixels x 64 columns per workgroup, half the accumulators and half the patch items per thread: <= 168 registers, i.e.
// THREE workgroups per CU (launch bound 3) to cover each other's memory latencies (LABNOTES 10.1).
// One segment of work: channel chunks [c_begin, c_end) of output tile `tile_id`.  The one-tile-per-workgroup kernels call it once; the balanced ("stream-K")
// instances (SK, below) call it for every piece of a workgroup's share of the launch's (tile, chunk) iterations: sk_mode 0 = the segment is a whole tile
// (ordinary epilogue), 1 = a tile's tail or middle (the accumulators are PUBLISHED as a partial tile and the segment ends), 2 = a tile's head (the partial
// tiles of the workgroups that own the rest of the tile are added, in iteration order, then the ordinary epilogue runs).
struct SKSeg {
  int mode = 0;
  int self = 0;     // partial-tile slot / flag of this workgroup (its index in iteration order)
  int first = 0;    // mode 2: slots first .. first + count - 1 hold the rest of the tile
  int count = 0;
};

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32, int TS, bool ONE = false, int A16 = 0, int NJ = 2, bool SK = false>
__device__ __forceinline__ void conv3x3_x3_tile(const C3Args& a, const int tile_id, const int c_begin, const int c_end, const SKSeg sk) {
  static_assert(!SK || (NJ == 2 && !SPLIT && !TS && A16 == 0 && WM == 2 && WN == 2), "balanced instances: the unsplit 128 x 128 tiles");
  static_assert(NJ == 2 || (NJ == 1 && !SPLIT && !F32 && !TS && !ONE && A16 == 0), "32-column wave tiles: the unsplit split-bf16 3 x 3 instances");
  constexpr int WCOLS = NJ * 32;  // output columns per wave
  static_assert(!ONE || !F32, "single pass: bf16 operands");
  static_assert(A16 == 0 || (ONE && !SPLIT), "bf16-stored maps: the unsplit single-pass instances");
  constexpr bool IN16 = A16 == 1 || A16 == 2, OUT16 = A16 == 1 || A16 == 3;
  constexpr int BM = WM * 64;
  constexpr int NQ = TS ? 8 : 18;  // k16 steps per chunk
  static_assert(!TS || !F32, "tap-subset layers: split-bf16 only");
  static_assert(TS != 1 || !SPLIT, "the transposed convolution's phase scatter has no split epilogue");
  static_assert(MAXP <= 16, "the source-row exchange below has every lane of an 8-lane group compute two of the MAXP patch items (ps = k4, k4 + 8)");
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ph = smem;
  const vmm_conv_desc& p = a.p;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 31, lk = lane >> 5;
  const int W = p.Win, H = p.Hin, HW = H * W;
  // The phase stamps are compiled in only by -DVMM_C3_TRACE_BUILD=1 (tools/build_ab.py conv3x3_bf16x3 -DVMM_C3_TRACE_BUILD=1, then VMM_LIB_PATH): the
  // run-time test `a.trace != nullptr` alone kept the pointer and the workgroup's slot index alive through the whole kernel -- 15-35 registers in
  // every instantiation, and 36 bytes of scratch in the 256 x 64 instance of the 96 x 96 layers (256 -> 237 registers without them; the family
  // 5.93 -> 5.87 ms per guided step on one box).
#ifndef VMM_C3_TRACE_BUILD
#define VMM_C3_TRACE_BUILD 0
#endif
  auto stamp = [&](int k) {  // measurement aid (outside the step loop only)
    if constexpr (VMM_C3_TRACE_BUILD)
      if (a.trace && tid == 0 && k < 14) a.trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + k] = __builtin_readcyclecounter();
  };
  if constexpr (VMM_C3_TRACE_BUILD)
    if (a.trace && tid == 0) {
      a.trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 14] = __builtin_amdgcn_s_getreg(63492);  // HW_ID
      a.trace[(size_t)(blockIdx.y * gridDim.x + blockIdx.x) * 16 + 15] = __builtin_amdgcn_s_getreg(63508);  // XCC_ID
    }
  stamp(0);
  // Workgroup b runs on XCD b % 8 (dispatch order), each XCD has its own L2.  Tiles are numbered so that an XCD works on a CONTIGUOUS
  // range of them: the column tiles of one row tile (which read the same patch) and neighbouring row tiles (which share halo rows) meet in
  // one L2 instead of being fetched once per XCD.
  const int mtile = tile_id / a.n_tiles;
  const int n0 = (tile_id % a.n_tiles) * (WN * WCOLS);
  const int Cin = p.C1 + p.C2;
  const int nchunks = (TS == 2 ? 4 : 1) * Cin / CK;
  if (c_begin >= c_end) return;

  int img = 0, ty0 = 0, tx0 = 0, g0 = 0;
  if (MODE) {
    img = (int)__umulhi((unsigned)mtile, a.tpf_magic);
    const int t = mtile - img * a.tiles_per_frame;
    const int tyi = (int)__umulhi((unsigned)t, a.tx_magic);
    ty0 = tyi * (BM / 16);
    tx0 = (t - tyi * a.tiles_x) * 16;
  } else {
    g0 = mtile * BM;
  }

  // p.a_img_mod (2-D tiles): frames of the batch's second half read the first half's rows of a1 (one pre-norm tensor shared by both guidance
  // branches); coefficients and output rows stay those of the tile's own frame
  const int simg = (!TS && MODE && p.a_img_mod > 0 && img >= p.a_img_mod) ? img - p.a_img_mod : img;
  // this lane's two output pixels: patch row of the centre tap and the mask of taps that read inside the image
  int prow[2];
  unsigned tapmask[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wm * 64 + i * 32 + lrow;
    if (MODE) {
      prow[i] = ((m >> 4) + 1) * a.pitch + (m & 15) + 1;
      tapmask[i] = 0x1FFu;  // out-of-image patch rows hold zeros
    } else {
      prow[i] = m + a.halo;
      unsigned msk = 0;
      const int g = g0 + m;
      if (g < a.total_rows) {
        const int pix = g % HW;
        const int h = pix / W, w = pix - h * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
          if (hh >= 0 && hh < H && ww >= 0 && ww < W) msk |= 1u << t;
        }
      }
      tapmask[i] = msk;
    }
  }

  // patch staging: item (row r, 4 channels k4); source row of every item is chunk-invariant
  const int k4 = tid & 7;
  // TS == 2, flat tiles: source pixel of the (0, 0) sub-pixel of every patch cell, kept (decoding a cell index costs two divisions)
  int cpix[(TS == 2 && !MODE) ? MAXP : 1];
  if (TS == 2 && !MODE) {
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int g = g0 - a.halo + (tid >> 3) + ps * 32;
      int s = -1;
      if ((tid >> 3) + ps * 32 < a.PR && g >= 0 && g < a.total_rows) {
        const int im = g / HW, rem = g - im * HW, h = rem / W, w = rem - h * W;
        s = im * 4 * HW + 4 * h * W + 2 * w;
      }
      cpix[(TS == 2 && !MODE) ? ps : 0] = s;
    }
  }
  const int PRc = MODE ? 18 * (BM / 16 + 2) : a.PR;  // 2-D tiles: the patch geometry is a compile-time constant
  // Straight-line selects, no branches: this runs MAXP times before the first request of a workgroup can leave (measured: with nested ifs --
  // 110 exec-mask branches in the prologue -- 10k cycles passed between entry and the first patch load, a fifth of the workgroup's life).
  auto src_row_of = [&](int ps, int sub) -> int {  // sub = 2 sy + sx (TS == 2)
    const int r = (tid >> 3) + ps * 32;
    if (TS == 2 && !MODE) {
      const int s = cpix[(TS == 2 && !MODE) ? ps : 0];
      return s >= 0 ? s + (sub >> 1) * 2 * W + (sub & 1) : s;
    }
    bool ok = r < PRc;
    int s;
    if (MODE) {
      const int py = r / 18, px = r - py * 18;
      int h = ty0 - 1 + py, w = tx0 - 1 + px;
      // periodic padding (vddp.py:163-243; 2-D tiles only, plan_c3): the halo rows / columns that leave the frame come from the opposite border
      const int hwrap = h + (h < 0 ? H : 0) - (h >= H ? H : 0), wwrap = w + (w < 0 ? W : 0) - (w >= W ? W : 0);
      h = p.wrap_h ? hwrap : h;
      w = p.wrap_w ? wwrap : w;
      ok = ok & ((unsigned)h < (unsigned)H) & ((unsigned)w < (unsigned)W);
      s = TS == 2 ? img * 4 * HW + (2 * h + (sub >> 1)) * 2 * W + 2 * w + (sub & 1) : simg * HW + h * W + w;
    } else {
      s = g0 - a.halo + r;
      ok = ok & (s >= 0) & (s < a.total_rows);
    }
    return ok ? s : -1;
  };
  // 3 x 3 layers: the eight threads that stage the same patch rows (channel groups k4 = 0 .. 7 of rows (tid >> 3) + 32 ps) would each compute
  // all MAXP source rows -- ~35 vector instructions apiece, at the half rate the set-up gets in the sibling's MFMA shadow.  Each computes
  // two of them (ps = k4 and k4 + 8) and the group exchanges them (consecutive lanes of one wave: ds_bpermute).
  int srow[TS ? 1 : MAXP];
  if (!TS) {
    const int mine0 = src_row_of(tid & 7, 0);
    const int mine1 = MAXP > 8 ? src_row_of((tid & 7) + 8 < MAXP ? (tid & 7) + 8 : 0, 0) : 0;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps)
      srow[TS ? 0 : ps] = __builtin_amdgcn_ds_bpermute(((lane & ~7) | (ps & 7)) << 2, ps < 8 ? mine0 : mine1);
  }
  auto src_row = [&](int ps, int sub) -> int { return TS ? src_row_of(ps, sub) : srow[TS ? 0 : ps]; };
  // which of this thread's patch items are real rows (chunk-invariant).  The loads below are UNCONDITIONAL (padding / beyond-the-patch items
  // re-read row 0 and are zeroed when the patch is stored): a lane-dependent branch around a load splits the step's basic block, and
  // then neither the compiler's waits are counted nor can the requests be interleaved with the MFMAs.
  unsigned vmask = 0;
#pragma unroll
  for (int ps = 0; ps < MAXP; ++ps)
    if (src_row(ps, 0) >= 0) vmask |= 1u << ps;
  // channel chunk -> (sub-pixel, first channel): TS == 2 walks the four sub-pixels of the cell, Cin channels each
  auto sub_of = [&](int cc) { return TS == 2 ? cc >> a.cps_shift : 0; };
  auto chan_of = [&](int cc) { return (TS == 2 ? cc & ((1 << a.cps_shift) - 1) : cc) * CK; };
  constexpr bool PAIR = VMM_C3_PAIR && ONE && !SPLIT && !TS && !SK && NJ == 2 && !(MODE == 0 && MAXP > 8);  // (the flat 256-row tile would spill with two register sets)
  f32x4 pregs[PAIR ? 2 : 1][MAXP];  // patch items in flight (PAIR: set 1 = the second chunk of a pair); every index below is a compile-time constant
  f32x4 (&preg)[MAXP] = pregs[0];
  f32x4 cfa2 = {1.f, 0.f, 1.f, 0.f}, cfb2 = cfa2;
  using Set0 = std::integral_constant<int, 0>;
  using Set1 = std::integral_constant<int, PAIR ? 1 : 0>;  // (never instantiated with 1 in the chunk-by-chunk instances)
  // GroupNorm * FiLM coefficients of this thread's four channels for the chunk in flight (2-D tiles: one sample per tile).  Requested with
  // the chunk's patch loads, not when the patch is stored: a dependent L2 round trip would otherwise sit on the critical path.
  f32x4 cfa = {1.f, 0.f, 1.f, 0.f}, cfb = cfa;
  auto load_coef = [&](int cc, int set = 0) {  // (set: compile-time constant at every call site -- PAIR's second chunk keeps its own coefficients)
    const int c0 = cc * CK;
    if (!TS && MODE && p.a_mode == 1 && c0 < p.C1) {
      const float* cf = p.a_coef + ((long long)(img / p.a_imgs_per_sample) * p.C1 + c0 + k4 * 4) * 2;
      if (set) {
        cfa2 = *reinterpret_cast<const f32x4*>(cf);
        cfb2 = *reinterpret_cast<const f32x4*>(cf + 4);
      } else {
        cfa = *reinterpret_cast<const f32x4*>(cf);
        cfb = *reinterpret_cast<const f32x4*>(cf + 4);
      }
    }
  };
  auto load_patch = [&](int cc, auto SETT) {  // raw loads only, so that they stay in flight under the MFMAs; the operand transform runs at store time
    constexpr int SET = decltype(SETT)::value;
    f32x4 (&preg)[MAXP] = pregs[SET];
    load_coef(cc, SET);
    const int c0 = chan_of(cc);
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = (src1 ? c0 : c0 - p.C1) + k4 * 4;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int sr = max(src_row(ps, sub_of(cc)), 0);
      if constexpr (IN16) {  // four bf16 = 8 bytes, carried as bits in the item's first two registers
        const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16s*>(src) + (long long)sr * ld + cb);
        preg[ps] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
      } else {
        preg[ps] = *reinterpret_cast<const f32x4*>(src + (long long)sr * ld + cb);
      }
    }
  };
  // One item of the next chunk's patch.  vmcnt retires in order, so an HBM-latency load blocks every younger weight-fragment wait
  // until it lands: the prefetch is issued one item per k16 step, AFTER that step's weight loads, which leaves it PFB + 1 steps to
  // arrive before anything waits on it (all MAXP items at the chunk top parked the waves for ~40 % of their cycles).
  // `last` (wave-uniform; unsplit 3 x 3 layers): the tile's last chunk has nothing to prefetch, and the request cannot be skipped without a
  // branch in the step -- it used to re-read its own patch (41 KB per workgroup for nothing).  It fetches the epilogue's bias pieces instead:
  // item ps < 8 = the 16 bytes that bias piece (j, g) = (ps >> 2, ps & 3) of this lane needs, so the epilogue finds them in registers.
  constexpr int NBIAS = (!SPLIT && !TS && !IN16) ? (MAXP < 4 * NJ ? MAXP : 4 * NJ) : 0;  // bias pieces that ride in the prefetch registers
  auto load_patch_item = [&](int cc, int ps, bool last, auto SETT) {
    f32x4 (&preg)[MAXP] = pregs[decltype(SETT)::value];
    const int c0 = chan_of(cc);
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = (src1 ? c0 : c0 - p.C1) + k4 * 4;
    const int sr = max(src_row(ps, sub_of(cc)), 0);
    if constexpr (IN16) {
      const uint2 u = *reinterpret_cast<const uint2*>(reinterpret_cast<const bf16s*>(src) + (long long)sr * ld + cb);
      preg[ps] = f32x4{__uint_as_float(u.x), __uint_as_float(u.y), 0.f, 0.f};
      return;
    }
    const float* q = src + (long long)sr * ld + cb;
    if (NBIAS) {
      // (lane offset recomputed here, two instructions: as a loop-invariant 64-bit pointer it cost the 256 x 64 kernel a spill whose reload in
      // the epilogue waited, vmcnt being in order, for every output store)
      int lko = lane;
      asm volatile("" : "+v"(lko));
      const float* qb = p.bias ? p.bias + (n0 + wn * WCOLS + 4 * (lko >> 5) + (ps < NBIAS ? (ps >> 2) * 32 + 8 * (ps & 3) : 0)) : p.a1;
      q = last ? qb : q;
    }
    preg[ps] = *reinterpret_cast<const f32x4*>(q);
  };
  // (2-D tiles: the patch geometry is a compile-time constant, so only the last item keeps a row test; everything up to the LDS store is
  // selects -- the nested ifs this replaces compiled to three exec-mask branches per item, between the arrival of the patch and the barrier)
  auto store_patch = [&](int cc, auto SETT) {
    constexpr int SET = decltype(SETT)::value;
    constexpr int HOFF = SET * CK;  // PAIR: the pair's second chunk lives in the lo half of the patch rows
    f32x4 (&preg)[MAXP] = pregs[SET];
    const int c0 = chan_of(cc);
    const bool xform = !TS && c0 < p.C1 && p.a_mode == 1;
    const int rows_per_sample = HW * p.a_imgs_per_sample;
    // GroupNorm * FiLM coefficients of this thread's four channels: one sample per 2-D tile -> fetched once per chunk, not per item
    f32x4 ca = SET ? cfa2 : cfa, cb4 = SET ? cfb2 : cfb;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int r = (tid >> 3) + ps * 32;
      f32x4 v = preg[ps];
      const bool real = (vmask >> ps) & 1u;
      if constexpr (IN16) {
        unsigned b0 = __float_as_uint(v.x), b1 = __float_as_uint(v.y);
        if (!xform) {  // (wave-uniform) the stored bits are the operand
          if ((ps + 1) * 32 <= PRc || r < PRc) *reinterpret_cast<uint2*>(&Ph[r * CROW + HOFF + k4 * 4]) = make_uint2(real ? b0 : 0u, real ? b1 : 0u);
          continue;
        }
        v = f32x4{__uint_as_float(b0 << 16), __uint_as_float(b0 & 0xffff0000u), __uint_as_float(b1 << 16), __uint_as_float(b1 & 0xffff0000u)};
      }
      if (!MODE && xform) {  // flat row tiles run across samples (wave-uniform condition; padding items read sample 0's coefficients, unused)
        const int sr = max(src_row(ps, 0), 0);
        const float* cf = p.a_coef + ((long long)(sr / rows_per_sample) * p.C1 + c0 + k4 * 4) * 2;
        ca = *reinterpret_cast<const f32x4*>(cf);
        cb4 = *reinterpret_cast<const f32x4*>(cf + 4);
      }
      if (xform) {  // wave-uniform
        v.x = igemm::silu_fast(v.x * ca.x + ca.y);
        v.y = igemm::silu_fast(v.y * ca.z + ca.w);
        v.z = igemm::silu_fast(v.z * cb4.x + cb4.y);
        v.w = igemm::silu_fast(v.w * cb4.z + cb4.w);
      }
      // zero padding is applied AFTER the activation (vddp.py:268-285): padded items are 0 whatever the transform made of the stand-in row
      v.x = real ? v.x : 0.f; v.y = real ? v.y : 0.f; v.z = real ? v.z : 0.f; v.w = real ? v.w : 0.f;
      if ((ps + 1) * 32 <= PRc || r < PRc) {
        if constexpr (F32) {  // exact-fp32 variant: the patch row is 32 floats (the same 128 + 16 bytes as bf16 hi | lo)
          *reinterpret_cast<f32x4*>(&Ph[r * CROW + k4 * 8]) = v;
        } else {
          unsigned h0, l0, h1, l1;
          split2c(v.x, v.y, h0, l0);
          split2c(v.z, v.w, h1, l1);
          *reinterpret_cast<uint2*>(&Ph[r * CROW + HOFF + k4 * 4]) = make_uint2(h0, h1);
          // (single-pass instances never read the lo half of a patch row: neither stored nor -- its value being dead -- formed.  -DVMM_C3_ONE_LO=1: stored, for A/B)
          if constexpr (!ONE || (VMM_C3_ONE_LO && !PAIR)) *reinterpret_cast<uint2*>(&Ph[r * CROW + CK + k4 * 4]) = make_uint2(l0, l1);
        }
      }
    }
  };

  // weight fragments: plane (column tile nt, k16 step ks, hi|lo) = 64 lanes x 16 bytes contiguous
  const uint4* wf = reinterpret_cast<const uint4*>(p.w);
  const uint4* bbase[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) bbase[j] = wf + (long long)((n0 + wn * WCOLS) / 32 + j) * a.KS * 128 + lane;
  const int cin16 = (TS == 2 ? 4 : 1) * Cin / 16;
  // TS: the unit's corner of the 3 x 3 taps -- rows tr0, tr0 + 1 and columns tc0, tc0 + 1.  TS == 1: the output phase of this wave's 64
  // columns (phase-major columns); TS == 2: the sub-pixel of the channel chunk.
  const int phase = TS == 1 ? (n0 + wn * WCOLS) / p.Cout : 0;
  auto corner = [&](int cc, int& tr0, int& tc0) {
    if (TS == 1) { tr0 = phase >> 1; tc0 = phase & 1; }
    else { const int sub = sub_of(cc); tr0 = 1 - (sub >> 1); tc0 = 1 - (sub & 1); }
  };
  auto load_b = [&](uint4 (&d)[4], int ks) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const uint4* q = bbase[j] + (long long)ks * 128;
      d[2 * j] = q[0];
      if constexpr (!ONE) d[2 * j + 1] = q[64];
    }
  };

  f32x16 acc[2][NJ];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // A fragments of one k16 step: [i*2 + plane], double-buffered like the weights so that the LDS reads of step q+1 are in
  // flight while the twelve MFMAs of step q issue
  // patch addressing: one base per (pixel, kernel row); the kernel column, k16 half and hi|lo plane are immediate offsets
  const int pitch = MODE ? 18 : a.pitch;
  int abase[2][3];  // element offsets into Ph
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) abase[i][kh] = (prow[i] + (kh - 1) * pitch - 1) * CROW + lk * (F32 ? 16 : 8);
  // Flat row tiles (MODE 0): a tap that leaves the image reads a real patch row of ANOTHER pixel, so it is masked per pixel.  The masked lanes
  // read three rows of zeros behind the patch instead (one select on the ADDRESS per pixel block; it was eight selects on the fragment's
  // registers -- 16 v_cndmask per k16 step between the MFMAs, 546 in the kernel).  Immediate offsets stay below 3 * CROW.
#ifndef VMM_C3_ZERO_ROWS
#define VMM_C3_ZERO_ROWS 1
#endif
  constexpr bool ZROWS = VMM_C3_ZERO_ROWS && !MODE;
  const int zb = PRc * CROW;  // element offset of the zero rows
  auto load_a = [&](uint4 (&d)[4], int tap, int s, int hoff = 0) {  // hoff (PAIR): CK for the pair's second chunk
    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // bf16x3: 8 hi then 8 lo values of channels s*16 + lk*8 .. +7;  fp32: those eight channels as two float4
      // (TS: the tap is a run-time, wave-uniform value: the centre-row base plus ONE scalar offset.  Selecting among the three row bases
      // -- `kh == 0 ? abase[i][0] : ...` -- was turned back into a run-time index by the compiler: the six bases went to scratch and every k16
      // step paid two scratch loads on the vector-memory counter its weight fragments wait on; 32 bytes of scratch in every conv_s2 instance)
      // (family 0.775 -> 0.64 ms per guided step on one box)
      const bool ok = MODE || ((tapmask[i] >> tap) & 1u);
      const unsigned short* q;
      if constexpr (ZROWS) {
        if (TS) q = Ph + (ok ? abase[i][1] + (((kh - 1) * pitch + kw) * CROW + s * 16) : zb);
        else q = Ph + (ok ? abase[i][kh] : zb) + (kw * CROW + s * (F32 ? 32 : 16));
      } else {
        q = TS ? Ph + abase[i][1] + (((kh - 1) * pitch + kw) * CROW + s * 16) : Ph + abase[i][kh] + (kw * CROW + s * (F32 ? 32 : 16));
      }
      uint4 vh = *reinterpret_cast<const uint4*>(q + hoff);
      if constexpr (ONE) {
        if (!ZROWS && !ok) vh = make_uint4(0u, 0u, 0u, 0u);
        d[2 * i] = vh;
      } else {
        uint4 vl = *reinterpret_cast<const uint4*>(q + (F32 ? 8 : CK));
        if (!ZROWS && !ok) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
        d[2 * i] = vh;
        d[2 * i + 1] = vl;
      }
    }
  };
  auto mma_step = [&](const uint4 (&av)[4], const uint4 (&b)[4]) {
    if constexpr (F32) {
      // exact fp32: v_mfma_f32_32x32x2_f32, eight K = 2 products per k16 step (the lane halves hold channels lk*8 + e, e = 0..7 of both
      // operands); e-major order keeps consecutive MFMAs on different accumulators
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
          for (int j = 0; j < NJ; ++j) {
            if constexpr (SPLIT) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(pa[i][e], wb[j][e], acc[i][j], 0, 0, 0);
            else acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[j][e], pa[i][e], acc[i][j], 0, 0, 0);
          }
      return;
    }
    const bf16x8 ah0 = __builtin_bit_cast(bf16x8, av[0]), al0 = __builtin_bit_cast(bf16x8, av[1]);
    const bf16x8 ah1 = __builtin_bit_cast(bf16x8, av[2]), al1 = __builtin_bit_cast(bf16x8, av[3]);
    const bf16x8 bh0 = __builtin_bit_cast(bf16x8, b[0]), bl0 = __builtin_bit_cast(bf16x8, b[1]);
    const bf16x8 bh1 = __builtin_bit_cast(bf16x8, b[2]), bl1 = __builtin_bit_cast(bf16x8, b[3]);
    // Unsplit layers: the weights are the MFMA "A" (rows = output channels), the pixels the "B" (columns): a lane then holds 4 x 4
    // CONSECUTIVE output channels of ONE pixel and the epilogue writes 16-byte pieces (4x fewer store instructions; the store tail
    // was issue-bound).  Split layers keep pixels as rows: their epilogue is atomics, which want 32 lanes on one 128-byte line.
    // pass-major: consecutive MFMAs never share an accumulator
    auto mm = [&](const bf16x8& pix, const bf16x8& wgt, f32x16 c) {
      if constexpr (SPLIT) return vmm_mfma16(pix, wgt, c);
      else return vmm_mfma16(wgt, pix, c);
    };
    if constexpr (ONE && NJ == 2) {
      acc[0][0] = mm(ah0, bh0, acc[0][0]);
      acc[0][1] = mm(ah0, bh1, acc[0][1]);
      acc[1][0] = mm(ah1, bh0, acc[1][0]);
      acc[1][1] = mm(ah1, bh1, acc[1][1]);
      return;
    }
    if constexpr (NJ == 1) {  // 64 x 32 wave tile: six MFMAs per k16 step, consecutive ones on different accumulators
      acc[0][0] = mm(al0, bh0, acc[0][0]);
      acc[1][0] = mm(al1, bh0, acc[1][0]);
      acc[0][0] = mm(ah0, bl0, acc[0][0]);
      acc[1][0] = mm(ah1, bl0, acc[1][0]);
      acc[0][0] = mm(ah0, bh0, acc[0][0]);
      acc[1][0] = mm(ah1, bh0, acc[1][0]);
    } else {
      acc[0][0] = mm(al0, bh0, acc[0][0]);
      acc[0][1] = mm(al0, bh1, acc[0][1]);
      acc[1][0] = mm(al1, bh0, acc[1][0]);
      acc[1][1] = mm(al1, bh1, acc[1][1]);
      acc[0][0] = mm(ah0, bl0, acc[0][0]);
      acc[0][1] = mm(ah0, bl1, acc[0][1]);
      acc[1][0] = mm(ah1, bl0, acc[1][0]);
      acc[1][1] = mm(ah1, bl1, acc[1][1]);
      acc[0][0] = mm(ah0, bh0, acc[0][0]);
      acc[0][1] = mm(ah0, bh1, acc[0][1]);
      acc[1][0] = mm(ah1, bh0, acc[1][0]);
      acc[1][1] = mm(ah1, bh1, acc[1][1]);
    }
  };

  // software pipeline over the 18 k16 steps q = 2 tap + half of a chunk: the weight fragments of step q + PFB and the patch fragments
  // of step q + 1 are requested before the MFMAs of step q (PFB = 2 covers an L2 hit even when the sibling wave does not leave the
  // line in L1).  sched_barrier pins that order (the scheduler otherwise sinks the loads next to their uses to save registers).
  constexpr int NB = PFB + 1;
  static_assert(NQ % NB == 0, "the weight-fragment ring must close at a chunk boundary");
  uint4 bb[NB][4], aa[2][4];
  // step q of a chunk: tap (q >> 1) and channel half (q & 1); TS: the (q >> 1)-th tap of the unit's corner
  auto tap_of = [&](int cc, int q) {
    if (!TS) return q >> 1;
    int tr0, tc0;
    corner(cc, tr0, tc0);
    return (tr0 + (q >> 2)) * 3 + tc0 + ((q >> 1) & 1);
  };
  auto ks_of = [&](int cc, int q) { return tap_of(cc, q) * cin16 + cc * 2 + (q & 1); };
  if constexpr (PAIR) {
    // ---- chunk pairs (see VMM_C3_PAIR): chunk cc in the hi half of the patch rows, chunk cc + 1 in the lo half; both requested together, one barrier pair per PAIR
    const bool has2_first = c_begin + 1 < c_end;
    load_patch(c_begin, Set0{});
    if (has2_first) load_patch(c_begin + 1, Set1{});
#pragma unroll
    for (int q = 0; q < PFB; ++q) load_b(bb[q], ks_of(c_begin, q));
    stamp(1);
    store_patch(c_begin, Set0{});
    if (has2_first) store_patch(c_begin + 1, Set1{});
    if constexpr (ZROWS)
      for (int i = tid; i < 3 * CROW / 2; i += 256) reinterpret_cast<unsigned*>(Ph + zb)[i] = 0u;  // (stays zero: no patch store reaches row PR)
    stamp(2);
    __syncthreads();
    stamp(3);
    load_a(aa[0], tap_of(c_begin, 0), 0);
    // the 18 k16 steps of chunk `cc` (half HALF of the patch rows); during them: the weight fragments run on into chunk `nxt_w`, one patch item per step of chunk
    // `nxt_p` is requested into register set HALF (`last_p`: nothing left to prefetch -- the epilogue's bias pieces instead), and the last step requests the first
    // patch fragments of the pair's second chunk when it follows without a barrier (`a_next_tap` >= 0)
    auto steps = [&](auto HT, int cc, int nxt_w, int nxt_p, bool last_p, int a_next_tap) {
      constexpr int HALF = decltype(HT)::value;
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) asm volatile("" : "+v"(abase[i][kh]));
#pragma unroll
      for (int q = 0; q < NQ; ++q) {
        if (q + PFB < NQ) load_b(bb[(q + PFB) % NB], ks_of(cc, q + PFB));
        else load_b(bb[(q + PFB) % NB], ks_of(nxt_w, q + PFB - NQ));
        if (q == 0) load_coef(nxt_p, HALF);
        if (q < MAXP) load_patch_item(nxt_p, q, last_p, HT);
        if (q + 1 < NQ) load_a(aa[(q + 1) & 1], tap_of(cc, q + 1), (q + 1) & 1, HALF * CK);
        else if (HALF == 0 && a_next_tap >= 0) load_a(aa[(q + 1) & 1], a_next_tap, 0, CK);  // (NQ is even: slot 0, as the chunk top expects)
        if constexpr (!C3_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mma_step(aa[q & 1], bb[q % NB]);
        if constexpr (C3_INTERLEAVE) {  // single pass: four MFMAs, two weight-fragment requests, two LDS fragment reads per step
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
#pragma unroll
          for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
    };
    static_assert(NQ % 2 == 0, "the patch-fragment double buffer must close at a chunk boundary");
    for (int cc = c_begin; cc < c_end; cc += 2) {
      const bool has2 = cc + 1 < c_end, more = cc + 2 < c_end, more2 = cc + 3 < c_end;
      steps(Set0{}, cc, has2 ? cc + 1 : (more ? cc + 2 : cc), more ? cc + 2 : cc, !more, has2 ? tap_of(cc + 1, 0) : -1);
      if (has2) steps(Set1{}, cc + 1, more ? cc + 2 : cc + 1, more2 ? cc + 3 : cc + 1, !more2, -1);
      stamp(4 + (cc - c_begin));
      if (more) {
        __syncthreads();  // every wave is done reading the patches of the pair
        store_patch(cc + 2, Set0{});
        if (more2) store_patch(cc + 3, Set1{});
        __syncthreads();
        stamp(5 + (cc - c_begin));
        load_a(aa[0], tap_of(cc + 2, 0), 0);
      }
    }
  } else {
    load_patch(c_begin, Set0{});
  #pragma unroll
    for (int q = 0; q < PFB; ++q) load_b(bb[q], ks_of(c_begin, q));
    stamp(1);
    store_patch(c_begin, Set0{});
    if constexpr (ZROWS)
      for (int i = tid; i < 3 * CROW / 2; i += 256) reinterpret_cast<unsigned*>(Ph + zb)[i] = 0u;  // (stays zero: no patch store reaches row PR)
    stamp(2);
    __syncthreads();
    stamp(3);
    load_a(aa[0], tap_of(c_begin, 0), 0);
    for (int cc = c_begin; cc < c_end; ++cc) {
      const bool more = cc + 1 < c_end;
      const int nxt = more ? cc + 1 : cc;
      // keep the six patch bases opaque per chunk: the 18 per-step addresses then stay base + immediate (ds_read offset field)
      // instead of being hoisted out of the loop into 36 address registers
  #pragma unroll
      for (int i = 0; i < 2; ++i)
  #pragma unroll
        for (int kh = 0; kh < 3; ++kh) asm volatile("" : "+v"(abase[i][kh]));
  #pragma unroll
      for (int q = 0; q < NQ; ++q) {
        // (the prefetches are issued for the last chunk too -- they re-read it: a wave-uniform branch around them is still a basic-block
        // boundary in the middle of the step, i.e. requests in one lump between the MFMA groups)
        if (q + PFB < NQ) load_b(bb[(q + PFB) % NB], ks_of(cc, q + PFB));
        else load_b(bb[(q + PFB) % NB], ks_of(nxt, q + PFB - NQ));
        if (q == 0) load_coef(nxt);
        if (TS) {  // eight steps for up to eleven patch items: two per step
          if (2 * q < MAXP) load_patch_item(nxt, 2 * q, false, Set0{});
          if (2 * q + 1 < MAXP) load_patch_item(nxt, 2 * q + 1, false, Set0{});
        } else if (q < MAXP) {
          load_patch_item(nxt, q, !more, Set0{});
        }
        if (q + 1 < NQ) load_a(aa[(q + 1) & 1], tap_of(cc, q + 1), (q + 1) & 1);
        if constexpr (!C3_INTERLEAVE) __builtin_amdgcn_sched_barrier(0);
        mma_step(aa[q & 1], bb[q % NB]);
        if constexpr (C3_INTERLEAVE && NJ == 1) {  // six MFMAs: two weight-fragment requests, four LDS fragment reads, the step's patch item
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        if constexpr (C3_INTERLEAVE && !ONE && NJ == 2) {  // (fp32 variant: 32 MFMAs of 64 cycles per step, the requests go between the first nine)
          // nothing queues behind the MFMA in flight: the requests above go BETWEEN this step's MFMAs (weight fragments first: they have the
          // longest way), not in front of them as one block during which the matrix pipe runs dry
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);  // one vector-memory read
          }
  #pragma unroll
          for (int k = 0; k < 4; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
          }
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);    // the patch prefetch item(s) of this step, if any
          __builtin_amdgcn_sched_group_barrier(0x008, 3, 0);
        }
        if constexpr (C3_INTERLEAVE && ONE) {  // single pass: four MFMAs, two weight-fragment requests, two LDS fragment reads per step
  #pragma unroll
          for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x020, 1, 0);
          }
  #pragma unroll
          for (int k = 0; k < 2; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
          }
          __builtin_amdgcn_sched_group_barrier(0x020, 2, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      }
      stamp(4 + 2 * (cc - c_begin));
      if (more) {
        __syncthreads();  // every wave is done reading the patch of chunk cc
        store_patch(cc + 1, Set0{});
        __syncthreads();
        stamp(5 + 2 * (cc - c_begin));
        load_a(aa[0], tap_of(cc + 1, 0), 0);
      }
    }

  }

  // epilogue.  Split channel reduction (gridDim.y > 1): the splits of one output tile add their partial sums in split order, so the
  // result does not depend on scheduling.  Split 0 writes through to memory; split y waits for ticket == y (acquire), adds with device-scope
  // fp32 atomics (performed at the memory side, coherent across the XCDs' L2s), and passes the ticket on behind an agent-scope release
  // (the last one resets it to 0).  Workgroups are dispatched y-major, i.e. split y - 1 is
  // always resident before split y.
  if constexpr (SPLIT) {
    // acc[i][j]: rows = pixels (r & 3) + 8 (r >> 2) + 4 lk of row tile i, column = output channel j*32 + lrow
    const bool first = blockIdx.y == 0;
    int* ticket = p.split_tickets + blockIdx.x;
    if (!first) {
      if (tid == 0)
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)blockIdx.y) __builtin_amdgcn_s_sleep(8);
      __syncthreads();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // (pairs with the release below)
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        long long orow;
        if (MODE) {
          orow = (long long)img * HW + (ty0 + (m >> 4)) * W + tx0 + (m & 15);
        } else {
          if (g0 + m >= a.total_rows) continue;
          orow = g0 + m;
        }
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
          const int col = n0 + wn * WCOLS + j * 32 + lrow;
          float v = acc[i][j][r];
          float* o = p.out + orow * p.ldo + col;
          if (first) {
            if (p.bias) v += p.bias[col];
            if (p.res) v += p.res[orow * p.ldres + col];
            __hip_atomic_store(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          } else {
            __hip_atomic_fetch_add(o, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          }
        }
      }
    }
    // Release at AGENT scope before the ticket is passed on.  With a workgroup-scope fence (a bare s_waitcnt) the ticket -- another address, another
    // memory channel -- could become visible to the next split on another XCD before this split's write-through stores had been performed at the
    // memory side; its atomic adds then landed first and were overwritten.  Seen as one training step in ~800 with a handful of gradients off by 1e-2
    // of their scale, only with several processes on the GPU (LABNOTES 9.8); the few-tile layers that split are too small for the fence to show.
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    // (MI355X_MICROARCH.md "Compiler hazard (ROCm 7.2, gfx950)": hipcc may drop the s_waitcnt vmcnt(0) behind buffer_wbl2 once it can prove the wave's
    // vmcnt scoreboard empty; inline asm is invisible to that pass, so the wait between the write-back and the ticket is spelled out.  tools/scan_isa.py
    // checks the built code object for it.)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ticket, blockIdx.y + 1 == gridDim.y ? 0 : (int)blockIdx.y + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    if constexpr (SK) {
      // Partial tiles travel through memory as this workgroup's own register image (16 pieces of 16 bytes per thread, piece q of all threads
      // contiguous: 4 KB per store instruction), written THROUGH to memory and read past the caches (sc0 sc1 on both sides: the two workgroups sit on
      // different CUs, often different XCDs, whose L1 / L2 are not coherent with each other -- MI355X_MICROARCH.md, "Valid forms").  One flag per
      // publisher: stores drained (vmcnt(0)) by every wave, barrier, relaxed agent-scope flag store; the reader polls with one lane, and resets the
      // flag (the array is all zero again when the launch ends).
      constexpr int SLOT = WM * 64 * WN * 64;  // floats per partial tile
      int* flags = a.p.split_tickets + 2048;
      int lt = tid;
      asm volatile("" : "+v"(lt));  // (the piece addresses are formed HERE: hoisted above the step loops they were spilled to scratch)
      if (sk.mode == 1) {
        float* slot = a.p.sk_work + (long long)sk.self * SLOT + lt * 4;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
              const f32x4 v = {acc[i][j][4 * g], acc[i][j][4 * g + 1], acc[i][j][4 * g + 2], acc[i][j][4 * g + 3]};
              asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(slot + ((i * 2 + j) * 4 + g) * 1024), "v"(v) : "memory");
            }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) __hip_atomic_store(flags + sk.self, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return;
      }
      if (sk.mode == 2) {
        for (int pj = sk.first; pj < sk.first + sk.count; ++pj) {  // fixed order: the sum does not depend on timing
          if (tid == 0)
            while (__hip_atomic_load(flags + pj, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) __builtin_amdgcn_s_sleep(8);
          __syncthreads();
          const float* slot = a.p.sk_work + (long long)pj * SLOT + lt * 4;
          // (four pieces at a time: sixteen in flight cost 64 registers on top of the accumulators -- scratch)
#pragma unroll
          for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
              f32x4 v[4];
#pragma unroll
              for (int g = 0; g < 4; ++g)
                asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(v[g]) : "v"(slot + ((i * 2 + j) * 4 + g) * 1024) : "memory");
              asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
              for (int g = 0; g < 4; ++g) {
                acc[i][j][4 * g] += v[g].x; acc[i][j][4 * g + 1] += v[g].y; acc[i][j][4 * g + 2] += v[g].z; acc[i][j][4 * g + 3] += v[g].w;
              }
            }
          if (tid == 0) __hip_atomic_store(flags + pj, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
      }
    }
    // acc[i][j]: rows = output channels (r & 3) + 8 (r >> 2) + 4 lk of column tile j, column = pixel i*32 + lrow of this wave.
    // The eight bias pieces of the lane are requested together and folded into the accumulators after ONE wait (a load -> wait ->
    // add -> store chain per 16-byte piece serialised sixteen L2 latencies per tile); the residual pieces likewise per pixel.
    {
      f32x4 bv[NJ][4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bv[j][g] = !p.bias                ? f32x4{0.f, 0.f, 0.f, 0.f}
                     : (j * 4 + g < NBIAS) ? preg[(j * 4 + g < NBIAS) ? j * 4 + g : 0]  // (requested during the last chunk's steps)
                                           : *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * WCOLS - (TS == 1 ? phase * p.Cout : 0) + j * 32 + 8 * g + 4 * lk);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            acc[i][j][4 * g] += bv[j][g].x; acc[i][j][4 * g + 1] += bv[j][g].y; acc[i][j][4 * g + 2] += bv[j][g].z; acc[i][j][4 * g + 3] += bv[j][g].w;
          }
    }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = wm * 64 + i * 32 + lrow;
      long long orow;
      if (MODE) {
        orow = (long long)img * HW + (ty0 + (m >> 4)) * W + tx0 + (m & 15);
        // TS == 1: input pixel (y, x), phase (py, px) -> output pixel (2y + py, 2x + px) of the 2H x 2W frame
        if (TS == 1) orow = (long long)img * 4 * HW + (2 * (ty0 + (m >> 4)) + (phase >> 1)) * 2 * W + 2 * (tx0 + (m & 15)) + (phase & 1);
      } else {
        if (g0 + m >= a.total_rows) continue;
        orow = g0 + m;
        if (TS == 1) {
          const int g = g0 + m, im = g / HW, rem = g - im * HW, y = rem / W, x = rem - y * W;
          orow = (long long)im * 4 * HW + (2 * y + (phase >> 1)) * 2 * W + 2 * x + (phase & 1);
        }
      }
      const int c0 = n0 + wn * WCOLS - (TS == 1 ? phase * p.Cout : 0) + 4 * lk;
      f32x4 rv[NJ][4];
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          rv[j][g] = !p.res ? f32x4{0.f, 0.f, 0.f, 0.f}
                     : OUT16 ? ld4(reinterpret_cast<const bf16s*>(p.res) + orow * p.ldres + c0 + j * 32 + 8 * g)
                             : *reinterpret_cast<const f32x4*>(p.res + orow * p.ldres + c0 + j * 32 + 8 * g);
      float* orp = p.out + orow * p.ldo + c0;
#pragma unroll
      for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[i][j][4 * g] + rv[j][g].x, acc[i][j][4 * g + 1] + rv[j][g].y, acc[i][j][4 * g + 2] + rv[j][g].z,
                           acc[i][j][4 * g + 3] + rv[j][g].w};
          if constexpr (OUT16) st4(reinterpret_cast<bf16s*>(p.out) + orow * p.ldo + c0 + j * 32 + 8 * g, v);
          else *reinterpret_cast<f32x4*>(orp + j * 32 + 8 * g) = v;
        }
    }
    stamp(12);
    if (p.gn_part) {
      // GroupNorm statistics of the output (vddp.py:274-279) while it is still in registers: per run of 8 consecutive output channels
      // the sum and the sum of squares over this wave's 64 pixels, combined over the workgroup's waves in LDS and written as this
      // tile's own slot of the (sample, group)'s contribution list (every slot exactly once: no zero-fill, no atomics; summed in a
      // fixed order by vmm_groupnorm_coef) -- the separate statistics pass over the output disappears.
      // 2-D tiles lie inside one frame, hence one sample.  Flat row tiles run across frames and samples: a tile (at most as long as a
      // sample, plan_c3) touches at most two samples, A = that of its first row and B = the next one, and keeps two sets of sums.
      const int R = HW * p.a_imgs_per_sample;                      // rows per sample
      const int smpA = MODE ? img / p.a_imgs_per_sample : g0 / R;
      const int bnd = MODE ? 0x7fffffff : (smpA + 1) * R;           // first row of sample B
      const bool straddle = !MODE && g0 + BM > bnd && bnd < a.total_rows;
      // 16 values per lane: (sum, sum of squares) x 8 runs; wave totals by a reduce-scatter butterfly -- every exchange halves the
      // values a lane still carries (8 + 4 + 2 + 1 shuffles), two plain steps finish: 17 shuffles instead of 96
      auto wave_sums = [&](bool setB) -> float {
        float gv[16];
#pragma unroll
        for (int k = 8 * NJ; k < 16; ++k) gv[k] = 0.f;  // (32-column wave tiles: runs 4 .. 7 do not exist)
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            float s1 = 0.f, s2 = 0.f;  // (the bias is already in the accumulators)
#pragma unroll
            for (int i = 0; i < 2; ++i) {
              bool mine = true;
              if (!MODE) {
                const int row = g0 + wm * 64 + i * 32 + lrow;
                mine = row < a.total_rows && ((row >= bnd) == setB);
              }
              if (mine) {
                const float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
                s1 += (v0 + v1) + (v2 + v3);
                s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
              }
            }
            gv[(j * 4 + g) * 2] = s1;
            gv[(j * 4 + g) * 2 + 1] = s2;
          }
        }
#pragma unroll
        for (int bit = 5, n = 8; bit >= 2; --bit, n >>= 1) {
          const bool hi = (lane >> bit) & 1;
#pragma unroll
          for (int k = 0; k < n; ++k) {
            const float send = hi ? gv[k] : gv[k + n], keep = hi ? gv[k + n] : gv[k];
            gv[k] = keep + lane_xor(send, bit);
          }
        }
        gv[0] += lane_xor(gv[0], 1);
        gv[0] += lane_xor(gv[0], 0);  // lane L now holds the wave total of value (L >> 2) & 15
        return gv[0];
      };
      const float totA = wave_sums(false);
      const float totB = straddle ? wave_sums(true) : 0.f;
      __syncthreads();  // every wave is done with the patch: reuse its LDS
      float* sc = reinterpret_cast<float*>(smem);
      if ((lane & 3) == 0) {
        sc[wave * 16 + (lane >> 2)] = totA;
        sc[64 + wave * 16 + (lane >> 2)] = totB;
      }
      __syncthreads();
      auto put_slot = [&](float* q, float v) { *q = v; };
      if (tid < WN * 16 && (NJ == 2 || (tid & 15) < 8)) {  // (32-column wave tiles: runs 0 .. 3)
        const int wn2 = tid >> 4, slot = tid & 15, run = slot >> 1;
        float vA = 0.f, vB = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) {
          vA += sc[(w2 * WN + wn2) * 16 + slot];
          vB += sc[64 + (w2 * WN + wn2) * 16 + slot];
        }
        const int cout0 = n0 + wn2 * WCOLS + (run >> 2) * 32 + (run & 3) * 8;
        const int cpg = p.Cout / p.gn_groups, rpg = cpg >> 3;
        const int grp = cout0 / cpg, rig = (cout0 - grp * cpg) >> 3;
        if (MODE) {
          const int fr = img - smpA * p.a_imgs_per_sample;
          if (NJ == 2 && a.gn_fine) {
            // the slot list is laid out for tiles of 8 pixel rows (vmm_conv3x3_fuses_gn does not know which arithmetic will run): this tile of 16 rows
            // covers the fine tiles (2 ty, tx) and (2 ty + 1, tx); the first gets the sums, the second zeros
            const int t = mtile - img * a.tiles_per_frame, tyi = t / a.tiles_x, txi = t - tyi * a.tiles_x;
            const int n_contrib = p.a_imgs_per_sample * 2 * a.tiles_per_frame * rpg;
            float* base = p.gn_part + (((long long)smpA * p.gn_groups + grp) * n_contrib) * 2 + (slot & 1);
            const int k0 = (fr * 2 * a.tiles_per_frame + (2 * tyi) * a.tiles_x + txi) * rpg + rig;
            put_slot(base + k0 * 2, vA);
            put_slot(base + (k0 + a.tiles_x * rpg) * 2, 0.f);
          } else {
          const int n_contrib = p.a_imgs_per_sample * a.tiles_per_frame * rpg;
          const int k = (fr * a.tiles_per_frame + (mtile - img * a.tiles_per_frame)) * rpg + rig;
          put_slot(p.gn_part + (((long long)smpA * p.gn_groups + grp) * n_contrib + k) * 2 + (slot & 1), vA);
          }
        } else {
          // sample s owns the tiles floor(s R / BM) .. floor(((s + 1) R - 1) / BM): ceil(R / BM) or one more of them.  Its slot list has
          // room for the larger count; the sample's last tile zeroes the spare slot when the count is the smaller one.
          const int n_max = (R + BM - 1) / BM + 1, n_contrib = n_max * rpg;
          auto put = [&](int smp, float v) {
            const int first_t = (int)(((long long)smp * R) / BM), last_t = (int)((((long long)smp + 1) * R - 1) / BM);
            float* base = p.gn_part + (((long long)smp * p.gn_groups + grp) * n_contrib) * 2 + (slot & 1);
            put_slot(base + ((mtile - first_t) * rpg + rig) * 2, v);
            if (mtile == last_t && last_t - first_t + 1 < n_max) put_slot(base + ((n_max - 1) * rpg + rig) * 2, 0.f);
          };
          put(smpA, vA);
          if (straddle) put(smpA + 1, vB);
        }
      }
    }
    stamp(13);
  }
}

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32, int TS, bool ONE = false, int A16 = 0, int NJ = 2>
__device__ __forceinline__ void conv3x3_x3_body(const C3Args& a) {
  // Workgroup b runs on XCD b % 8 (dispatch order), each XCD has its own L2.  Tiles are numbered so that an XCD works on a CONTIGUOUS
  // range of them: the column tiles of one row tile (which read the same patch) and neighbouring row tiles (which share halo rows) meet in
  // one L2 instead of being fetched once per XCD.
  const int G = gridDim.x, xcd = blockIdx.x & 7, jx = blockIdx.x >> 3;
  const int tile_id = C3_XCD_ORDER ? xcd * (G >> 3) + min(xcd, G & 7) + jx : (int)blockIdx.x;
  const int nchunks = (TS == 2 ? 4 : 1) * (a.p.C1 + a.p.C2) / CK;
  const int c_begin = blockIdx.y * a.chunks_per_split;
  conv3x3_x3_tile<WM, WN, MAXP, MODE, PFB, SPLIT, F32, TS, ONE, A16, NJ, false>(a, tile_id, c_begin, min(nchunks, c_begin + a.chunks_per_split), SKSeg{});
}

#if VMM_EXPERIMENTS
// Balanced launch of the few-tile layers (12 x 12 and 24 x 24 levels: 198 .. 792 tiles of 128 x 128 for 512 workgroup slots, i.e. 140 CUs with two
// long-lived workgroups and 116 with one that finishes early; LABNOTES 10.4).  A grid of exactly two workgroups per CU; workgroup g (in ITERATION order:
// XCD b % 8 owns a contiguous eighth, so that a tile's pieces and the column tiles sharing a patch meet in one L2) owns iterations
// [g total / G, (g + 1) total / G) of the launch's tile-major list of (tile, 32-channel chunk) iterations: the tail of one tile, whole tiles, the head of
// another.  At most its FIRST segment is a tail / middle piece (published early, never waited for); a head is always its last, so the pieces it
// collects were published long before, or -- a piece that is some workgroup's whole share -- at the same moment.  No workgroup waits before it has
// published, hence no cycle of waits whatever the dispatch order or residency; sums are added in iteration order (bit-reproducible).
template <int MAXP, int MODE, int PFB, bool F32, bool ONE>
__global__ __launch_bounds__(256, 2) void conv3x3_sk_kernel(const C3Args a) {
  const int G = gridDim.x, b = blockIdx.x;
  const int gi = (b & 7) * (G >> 3) + (b >> 3);  // (G is a multiple of 8)
  const int nchunks = (a.p.C1 + a.p.C2) / CK;
  const long long total = (long long)a.sk_tiles * nchunks;
  int it = (int)(total * gi / G);
  const int it1 = (int)(total * (gi + 1) / G);
  bool first = true;
  while (it < it1) {
    const int tile = it / nchunks, c0 = it - tile * nchunks, c1 = min(nchunks, c0 + (it1 - it));
    SKSeg sk;
    sk.self = gi;
    if (c0 > 0) {
      sk.mode = 1;
    } else if (c1 < nchunks) {
      sk.mode = 2;
      sk.first = gi + 1;
      const long long tile_end = (long long)(tile + 1) * nchunks;
      int n = 0;
      while (gi + 1 + n < G && total * (gi + 1 + n) / G < tile_end) ++n;
      sk.count = n;
    }
    if (!first) __syncthreads();  // every wave is done with the previous segment's patch (and epilogue scratch) in LDS
    first = false;
    conv3x3_x3_tile<2, 2, MAXP, MODE, PFB, false, F32, 0, ONE, 0, 2, true>(a, tile, c0, c1, sk);
    it += c1 - c0;
  }
}
#endif  // VMM_EXPERIMENTS (balanced launch)

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32, bool ONE = false, int A16 = 0>
#ifndef VMM_C3_WGS
#define VMM_C3_WGS 2
#endif
// (single-pass instances: with a third of the matrix work per chunk the kernel is bound by the latency of its patch loads, not by the matrix pipe -- VMM_C3_ONE_WGS
// workgroups per CU, i.e. waves per SIMD, for them)
#ifndef VMM_C3_ONE_WGS
#define VMM_C3_ONE_WGS 2
#endif
__global__ __launch_bounds__(256, (ONE ? VMM_C3_ONE_WGS : VMM_C3_WGS)) void conv3x3_x3_kernel(const C3Args a) {
  conv3x3_x3_body<WM, WN, MAXP, MODE, PFB, SPLIT, F32, 0, ONE, A16>(a);
}

#if VMM_EXPERIMENTS
// 32-column wave tiles (NJ = 1): 2 x 2 waves = 128 pixels (8 x 16) x 64 columns, three workgroups per CU
template <int MAXP, int MODE, int PFB>
__global__ __launch_bounds__(256, 3) void conv3x3_x3n_kernel(const C3Args a) {
  conv3x3_x3_body<2, 2, MAXP, MODE, PFB, false, false, 0, false, 0, 1>(a);
}
#endif

// the resampling layers (TS = 1: Upsample, 2: Downsample) under their own kernel name, so that profiles keep them apart from the 3 x 3 family
template <int WM, int WN, int MAXP, int MODE, int TS, bool SPLIT = false, bool ONE = false, int A16 = 0>
__global__ __launch_bounds__(256, 2) void conv_s2_kernel(const C3Args a) {
  conv3x3_x3_body<WM, WN, MAXP, MODE, 1, SPLIT, false, TS, ONE, A16>(a);
}

#if VMM_EXPERIMENTS
// ---------------------------------------------------------------------------------------------------------------------------------
// Persistent, wave-specialised variant of the same contraction (unsplit layers).  One workgroup of EIGHT waves per CU walks a list of
// 256-pixel x 64-column output tiles in 16-channel chunks (nine k16 steps, one per tap):
//   waves 0-3  "matrix" waves, one per SIMD: a 64 x 64 output tile each.  Both operands come from LDS (eight ds_read_b128 per twelve
//              MFMAs); the only vector-memory instructions they ever issue are the epilogue's -- nothing can sit in front of an
//              operand on vmcnt (in the kernel above the weight fragments of four waves went through the L1 / texture path, 16 KB per
//              k16 step and CU, behind the patch prefetch: every layer sat at 260-340 TFLOP/s whatever its shape).
//   waves 4-6  "patch" waves: global loads of the NEXT-BUT-ONE chunk's patch rows into registers, then GroupNorm * FiLM -> SiLU, the
//              bf16 hi | lo split and the LDS stores of the NEXT chunk -- VALU, LDS-store and VMEM pipes, beside the MFMAs.
//   wave  7    "weight" wave: the next chunk's 36 fragment planes (64 columns x nine k16 steps, hi | lo, 1 KB each, contiguous in the
//              fmt-2 packing) by LDS-DMA (global_load_lds_dwordx4): once per workgroup instead of once per wave, no registers.
// Patch and weight stage are double-buffered in LDS; ONE s_barrier per chunk (about 3.5k cycles of MFMAs) is the only synchronisation
// and also carries the workgroup from tile to tile: the first chunk of tile n + 1 is staged while the last chunk of tile n is
// multiplied, the epilogue's stores drain under the next tile's MFMAs, there is no prologue latency after the first tile.  Workgroup b
// takes tiles from the contiguous range of XCD b % 8 (dispatch order, MI355X_MICROARCH.md), column tiles of one row tile adjacent, so
// that halo rows and the row tile's re-reads by its column tiles are served by that XCD's L2.
constexpr int CK2 = 16;                    // channels per chunk
constexpr int CROW2 = 40;                  // patch row pitch in 2-byte units: 16 hi | 16 lo | 8 pad = 80 bytes
constexpr int PW_MAXP = 7;                 // patch items per patch-wave thread: 192 threads x 4 channels = 48 rows per pass
constexpr bool PW_KEEP_SD = false;        // keep the drained halves' store-data registers untouched for a whole tile (64 VGPRs: spills)
constexpr int PW_STAGE = 9 * 4 * 1024;     // bytes of one weight stage: nine k16 steps x (2 column tiles x hi | lo) planes of 1 KB

struct PWArgs {
  vmm_conv_desc p;
  int n_tiles;                   // 64-column tiles
  int mode, tiles_x, tiles_per_frame, PR, pitch, halo, KS, total_rows, n_units;
  int patch_bytes;               // one patch buffer, rounded up to 1 KB
  int dbg;                       // measurement knobs (VMM_PW_DBG): 2 no patch work, 4 no weight DMA, 8 no epilogue, 32 no output stores, 64 no GroupNorm sums
  int stagger;                   // cycles between the start of consecutive workgroup quarters (0 = all together)
  unsigned long long* trace;     // VMM_PW_TRACE: workgroup 0 stamps s_memtime at every barrier (arrive, leave) per wave
};

template <int MODE, bool F32>
__global__ __launch_bounds__(512, 2) void conv3x3_pw_kernel(const PWArgs a_in) {
  constexpr int BM = 256;
  extern __shared__ __attribute__((aligned(1024))) unsigned char smem_b[];
  // a private copy of the arguments: the barriers and the hand-placed loads below are asm statements with a "memory" clobber, after each
  // of which the compiler would re-read every argument it needs from the kernel-argument segment (an s_load + wait of ~200 cycles per
  // dependent round; the patch waves spent 3.7k cycles per phase "issuing" nine loads)
  const PWArgs a = a_in;
  const vmm_conv_desc& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int W = p.Win, H = p.Hin, HW = H * W;
  const int Cin = p.C1 + p.C2;
  const int nchunks = Cin / CK2;
  const int cin16 = Cin / 16;
  unsigned char* const wstage = smem_b + 2 * a.patch_bytes;

  // this workgroup's tiles: u = first + i * J, i < cnt; u = mtile * n_tiles + ntile
  const int J = gridDim.x >> 3, xcd = blockIdx.x & 7, j_in = blockIdx.x >> 3;
  const int upx = (a.n_units + 7) >> 3;
  const int u_lo = xcd * upx, u_hi = min(a.n_units, u_lo + upx);
  const int first = u_lo + j_in;
  const int cnt = first < u_hi ? (u_hi - first + J - 1) / J : 0;
  if (cnt == 0) return;           // (the whole workgroup: no wave skips a barrier the others wait at)
  const int P = cnt * nchunks;    // phases = (tile, chunk) pairs; every wave executes exactly 1 + P barriers

  struct Geo { int img, ty0, tx0, g0, n0, mtile; };
  auto geo_of = [&](int i) -> Geo {
    const int u = first + i * J;
    Geo g;
    g.mtile = u / a.n_tiles;
    g.n0 = (u - g.mtile * a.n_tiles) * 64;
    g.img = g.ty0 = g.tx0 = g.g0 = 0;
    if (MODE) {
      g.img = g.mtile / a.tiles_per_frame;
      const int t = g.mtile - g.img * a.tiles_per_frame;
      const int tyi = t / a.tiles_x;
      g.ty0 = tyi * 16;
      g.tx0 = (t - tyi * a.tiles_x) * 16;
    } else {
      g.g0 = g.mtile * BM;
    }
    return g;
  };
  // barriers are raw: what each role must have retired before it arrives differs (LDS stores / LDS-DMA / LDS reads), and a
  // matrix wave must NOT wait for its epilogue stores (vmcnt counts stores on gfx9)
  int tr_n = 0;
  const bool tracing = a.trace && blockIdx.x == 0 && lane == 0;
  auto stamp = [&](int which) {
    if (tracing && tr_n < 128) {
      a.trace[(tr_n * 8 + wave) * 2 + which] = __builtin_readcyclecounter();
      if (which) ++tr_n;
    }
  };
#define PW_BARRIER_LDS() do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); stamp(0); asm volatile("s_barrier" ::: "memory"); stamp(1); } while (0)
#define PW_BARRIER_DMA() do { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); stamp(0); asm volatile("s_barrier" ::: "memory"); stamp(1); } while (0)

  // Epilogue staging: 4 waves x 32 pixel rows x (256 + 16 pad) bytes behind the weight stages, private to each matrix wave (see there).
  constexpr int EP_PITCH = 272, EP_WAVE = 32 * EP_PITCH;
  const int ep_base = 2 * a.patch_bytes + 2 * PW_STAGE;
  const int bias_off = PW_MAXP * 48 * CROW2 * 2;  // 64 floats in the slack at the end of patch buffer 0: the bias of this workgroup's column tile

  if (wave >= 4 && (a.dbg & 256)) __builtin_amdgcn_s_setprio(2);
  if (wave == 7) {
    // ======================================================================================================== weight wave
    const uint4* wf = reinterpret_cast<const uint4*>(p.w);
    auto stage = [&](const Geo& g, int cc, unsigned char* dst) {
      const uint4* base = wf + (long long)(g.n0 / 32) * a.KS * 128 + lane;
#pragma unroll
      for (int idx = 0; idx < 36; ++idx) {
        const int t = idx >> 2, j = (idx >> 1) & 1, hl = idx & 1;
        const uint4* src = base + ((long long)j * a.KS + t * cin16 + cc) * 128 + hl * 64;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src, (__attribute__((address_space(3))) void*)(dst + idx * 1024), 16, 0, 0);
      }
    };
    Geo g = geo_of(0);   // tile being STAGED (one phase ahead)
    int si = 0, sc = 0;
    // the column tile of a workgroup never changes (grid 256: J = 32 is a multiple of the number of column tiles; smaller grids run one
    // tile per workgroup): its bias is parked in LDS once
    reinterpret_cast<float*>(smem_b + bias_off)[lane] = p.bias ? p.bias[g.n0 + lane] : 0.f;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    stage(g, 0, wstage);
    PW_BARRIER_DMA();
    for (int ph = 0; ph < P; ++ph) {
      if (ph + 1 < P) {
        if (++sc == nchunks) { sc = 0; ++si; g = geo_of(si); }
        if (!(a.dbg & 4)) stage(g, sc, wstage + ((ph + 1) & 1) * PW_STAGE);
      }
      PW_BARRIER_DMA();  // (this wave's only vector-memory operations are the stage's loads)
    }
    return;
  }

  if (wave >= 4) {
    // ======================================================================================================== patch waves
    const int pt = tid - 256;  // 0 .. 191
    const int k4 = pt & 3;
    const int rows_per_sample = HW * p.a_imgs_per_sample;
    // Per register set: the raw rows of one phase, plus what depends on the TILE only (source row of every item; flat tiles, which
    // run across samples: the item's offset into the GroupNorm coefficient table) -- recomputed when the set moves to another tile,
    // not per chunk (an integer division per item and chunk was a third of these waves' instructions).
    f32x4 preg[2][PW_MAXP];
    unsigned roff[2][PW_MAXP];   // byte offset of the item's 16 bytes from (source 1 + first channel of the chunk): row * lda1 * 4 + k4 * 16
    unsigned valid[2] = {0, 0};  // bit ps: the item is a real row (else zero padding / beyond the patch)
    unsigned second[2] = {0, 0}; // flat tiles: bit ps: the item belongs to the SECOND sample the tile touches (a patch spans at most two)
    int smp0[2] = {0, 0};        // flat tiles: first sample of the patch
    int tile_of[2] = {-1, -1};
    // GroupNorm * FiLM coefficients of this thread's four channels: [0..1] of the (first) sample, [2..3] flat tiles: of the second one
    f32x4 cf[2][4];
    Geo gq[2];
    int cq[2];
    auto set_tile = [&](auto set, int i) {
      constexpr int S = decltype(set)::value;
      if (tile_of[S] == i) return;
      tile_of[S] = i;
      const Geo g = geo_of(i);
      gq[S] = g;
      unsigned vmask = 0, smask = 0;
      if (!MODE) smp0[S] = max(g.g0 - a.halo, 0) / rows_per_sample;
      const int row_s1 = MODE ? 0 : (smp0[S] + 1) * rows_per_sample;  // first row of the second sample
#pragma unroll
      for (int ps = 0; ps < PW_MAXP; ++ps) {
        const int r = (pt >> 2) + ps * 48;
        int sr = -1;
        if (r < a.PR) {
          if (MODE) {
            const int py = r / 18, px = r - py * 18;
            const int h = g.ty0 - 1 + py, w = g.tx0 - 1 + px;
            if (h >= 0 && h < H && w >= 0 && w < W) sr = g.img * HW + h * W + w;
          } else {
            const int gg = g.g0 - a.halo + r;
            if (gg >= 0 && gg < a.total_rows) sr = gg;
          }
        }
        if (sr >= 0) vmask |= 1u << ps;
        if (!MODE && sr >= row_s1) smask |= 1u << ps;
        roff[S][ps] = ((unsigned)max(sr, 0) * (unsigned)p.lda1 + k4 * 4) * 4u;
      }
      valid[S] = vmask;
      second[S] = smask;
    };
    // Every global load of these waves is an inline-asm statement, every wait is placed by hand, and the code is straight-line (padded
    // items load row 0 and are zeroed by a select; rows beyond the patch land in the buffer's slack).  The rows of phase q + 2 must
    // stay in flight across the LDS stores of phase q + 1 AND across the barrier; hipcc's own bookkeeping cannot express that here:
    // with a branch per item it waited vmcnt(0) before every LDS store, with straight-line loads it folded the two register sets into
    // one body joined by register copies (which wait for the rows just requested), and once the bodies were kept apart it still
    // drained the queue before re-using the address temporaries (the .s of each attempt; cdna_hip_programming.md 5.7, form (ii)).
#define PW_GLOAD(dst, ptr) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(dst) : "v"(ptr) : "memory")
#define PW_GLOAD_S(dst, off, base) asm volatile("global_load_dwordx4 %0, %1, %2" : "=v"(dst) : "v"(off), "s"(base) : "memory")
    auto load_patch = [&](auto set, int cc) {  // raw loads only; the operand transform runs at store time, one phase later
      constexpr int S = decltype(set)::value;
      cq[S] = cc;
      const int c0 = cc * CK2;
      const bool src1 = c0 < p.C1;
      // (always the same number of loads -- PW_LOADS -- whatever the layer: the waits below count on it)
      const bool xf = p.a_mode == 1 && src1;
      const int smp = MODE ? gq[S].img / p.a_imgs_per_sample : smp0[S];
      const float* cfp = xf ? p.a_coef + ((long long)smp * p.C1 + c0 + k4 * 4) * 2 : p.a1;
      f32x4 &c0r = cf[S][0], &c1r = cf[S][1];
      const float* cfp4 = cfp + 4;
      PW_GLOAD(c0r, cfp);
      PW_GLOAD(c1r, cfp4);
      if (!MODE) {  // the second sample a flat tile may run into (clamped to the last sample: never selected then)
        const int nsmp = a.total_rows / rows_per_sample;
        const float* cfq = xf ? p.a_coef + ((long long)min(smp + 1, nsmp - 1) * p.C1 + c0 + k4 * 4) * 2 : p.a1;
        f32x4 &c2r = cf[S][2], &c3r = cf[S][3];
        const float* cfq4 = cfq + 4;
        PW_GLOAD(c2r, cfq);
        PW_GLOAD(c3r, cfq4);
      }
      if (a.dbg & 512) {
      } else if (src1) {  // scalar base + the tile's per-item byte offsets: one instruction per item
        const float* base = p.a1 + c0;
#pragma unroll
        for (int k = 0; k < PW_MAXP; ++k) {
          const int ps = S ? PW_MAXP - 1 - k : k;  // (set 1 walks its items backwards: the two instantiations must not be isomorphic, see phase())
          f32x4& dst = preg[S][ps];
          const unsigned off = roff[S][ps];
          PW_GLOAD_S(dst, off, base);
        }
      } else {     // second (concatenated) source: its own row pitch
        const float* base = p.a2 + (c0 - p.C1);
#pragma unroll
        for (int k = 0; k < PW_MAXP; ++k) {
          const int ps = S ? PW_MAXP - 1 - k : k;
          f32x4& dst = preg[S][ps];
          const unsigned off = roff[S][ps];  // (plan_pw: both sources share one row pitch)
          PW_GLOAD_S(dst, off, base);
        }
      }
    };
    constexpr int PW_LOADS = PW_MAXP + (MODE ? 2 : 4);  // vector-memory operations of one load_patch
    // the rows (and coefficients) of set S have landed once at most `younger` later vector-memory operations are outstanding; naming
    // every destination "+v" keeps the compiler from touching them above this statement
    auto wait_set = [&](auto set, auto younger) {
      constexpr int S = decltype(set)::value;
      constexpr int N = decltype(younger)::value;
      static_assert(PW_MAXP == 7, "operand list below");
      // (references first: clang does not capture a variable that only appears in an asm operand list of a generic lambda)
      f32x4 &r0 = preg[S][0], &r1 = preg[S][1], &r2 = preg[S][2], &r3 = preg[S][3], &r4 = preg[S][4], &r5 = preg[S][5], &r6 = preg[S][6];
      f32x4 &c0r = cf[S][0], &c1r = cf[S][1], &c2r = cf[S][2], &c3r = cf[S][3];
      asm volatile("s_waitcnt vmcnt(%c0)" ::"n"(N) : "memory");
      __builtin_amdgcn_sched_barrier(0);  // nothing -- in particular no register copy of a destination -- moves above the wait
      asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3), "+v"(r4), "+v"(r5), "+v"(r6), "+v"(c0r), "+v"(c1r));
      if (!MODE) asm volatile("" : "+v"(c2r), "+v"(c3r));
    };
    auto store_patch = [&](auto set, unsigned char* dst) {
      constexpr int S = decltype(set)::value;
      unsigned short* Ph = reinterpret_cast<unsigned short*>(dst);
      const int c0 = cq[S] * CK2;
      const bool xform = c0 < p.C1 && p.a_mode == 1;
#pragma unroll
      for (int k = 0; k < PW_MAXP; ++k) {
        const int ps = S ? PW_MAXP - 1 - k : k;
        const int r = (pt >> 2) + ps * 48;
        f32x4 v = preg[S][ps];
        if (xform) {  // (wave-uniform)
          f32x4 ca = cf[S][0], cb4 = cf[S][1];
          if (!MODE && ((second[S] >> ps) & 1u)) { ca = cf[S][2]; cb4 = cf[S][3]; }
          v.x = igemm::silu_fast(v.x * ca.x + ca.y);
          v.y = igemm::silu_fast(v.y * ca.z + ca.w);
          v.z = igemm::silu_fast(v.z * cb4.x + cb4.y);
          v.w = igemm::silu_fast(v.w * cb4.z + cb4.w);
        }
        if (!((valid[S] >> ps) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};  // zero padding is applied AFTER the activation (vddp.py:268-285)
        if constexpr (F32) {  // 16 floats per row (the same 64 + 16 bytes as 16 bf16 hi | 16 lo)
          *reinterpret_cast<f32x4*>(&Ph[r * CROW2 + k4 * 8]) = v;
        } else {
          unsigned h0, l0, h1, l1;
          split2c(v.x, v.y, h0, l0);
          split2c(v.z, v.w, h1, l1);
          *reinterpret_cast<uint2*>(&Ph[r * CROW2 + k4 * 4]) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(&Ph[r * CROW2 + CK2 + k4 * 4]) = make_uint2(l0, l1);
        }
      }
    };
    using S0 = std::integral_constant<int, 0>;
    using S1 = std::integral_constant<int, 1>;
    unsigned long long tsum[4] = {0, 0, 0, 0};  // tracing: cycles in (drain of half 1, requests, wait for the rows, transform + LDS stores)
    // phase q = (tile q / nchunks, chunk q % nchunks).  Register set q & 1 holds the raw rows of phase q: requested during phase q - 2,
    // stored (to patch buffer q & 1) during phase q - 1 -- a full phase in flight.
    int li = 0, lc = 0;      // (tile, chunk) of the most recently REQUESTED phase
    using N0 = std::integral_constant<int, 0>;
    using NL = std::integral_constant<int, PW_LOADS>;
    set_tile(S0{}, 0);
    load_patch(S0{}, 0);
    wait_set(S0{}, N0{});
    store_patch(S0{}, smem_b);
    if (P > 1) {
      if (++lc == nchunks) { lc = 0; ++li; }
    }
    set_tile(S1{}, li);
    load_patch(S1{}, lc);
    PW_BARRIER_LDS();
    auto phase = [&](auto cur, auto nxt, int q) {
      // during phase q: request phase q + 2 into the set phase q just vacated (first: in flight while the other set is transformed),
      // then store phase q + 1
      // (the two instantiations differ only in which register set plays which part: without a distinguishing marker the compiler folds
      // them into one body and swaps the sets with register copies -- copies that wait for the rows requested a moment ago)
      asm volatile("; patch phase, rows in flight -> set %0" ::"n"(decltype(cur)::value));
      // One shape for every phase: past the end the last chunk is simply requested again, so that the store below always finds exactly
      // PW_LOADS younger operations in the queue (a second variant of the wait made the compiler copy the destinations ABOVE it).
      // The two instantiations (register sets swapped) walk their items in opposite order: isomorphic bodies get folded into one that
      // swaps the sets with register copies.
      const unsigned long long ts0 = tracing ? __builtin_readcyclecounter() : 0;
      const unsigned long long ts1 = ts0;
      if (!(a.dbg & 2)) {
        if (q + 2 < P) {
          if (++lc == nchunks) { lc = 0; ++li; }
          set_tile(cur, li);
        }
        load_patch(cur, lc);
        const unsigned long long ts2 = tracing ? __builtin_readcyclecounter() : 0;
        if (tracing) { tsum[0] += ts1 - ts0; tsum[1] += ts2 - ts1; }
        // At most PW_LOADS operations outstanding <=> the rows of set nxt have landed: these waves issue loads only, loads retire in
        // order, and PW_LOADS of them were requested after the ones waited for.
        const unsigned long long ts3 = tracing ? __builtin_readcyclecounter() : 0;
        wait_set(nxt, NL{});
        const unsigned long long ts4 = tracing ? __builtin_readcyclecounter() : 0;
        store_patch(nxt, smem_b + ((q + 1) & 1) * a.patch_bytes);
        if (tracing) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); tsum[2] += ts4 - ts3; tsum[3] += __builtin_readcyclecounter() - ts4; }
      }
      PW_BARRIER_LDS();
    };
    for (int q = 0; q < P; q += 2) {
      phase(S0{}, S1{}, q);
      if (q + 1 < P) phase(S1{}, S0{}, q + 1);
    }
    if (tracing) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      for (int k = 0; k < 4; ++k) a.trace[128 * 16 + (wave - 4) * 4 + k] = tsum[k];
    }
#undef PW_GLOAD
#undef PW_GLOAD_S
    return;
  }

  // ============================================================================================================ matrix waves
  if (a.dbg & 128) __builtin_amdgcn_s_setprio(2);
  const int wm = wave;
  const int lrow = lane & 31, lk = lane >> 5;
  // 2-D tiles: lane -> pixel of the wave's 4 x 16 pixel strip.  Lanes 16-31 of an MFMA tile take the second pixel row ROTATED by two
  // columns: with the 80-byte patch pitch the sixteen lanes of every ds_read_b128 service group then sit on sixteen different patch
  // rows modulo 16, i.e. on disjoint bank quads (identity mapping: two-way conflicts on every A-operand read).
  const int tr_l = lrow >> 4, pc_l = lrow < 16 ? lrow : ((lrow - 2) & 15);
  Geo g = geo_of(0);
  unsigned tapmask[2] = {0x1FFu, 0x1FFu};
  auto set_masks = [&](const Geo& gg) {
    if (MODE) return;  // out-of-image patch rows hold zeros
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int m = wm * 64 + i * 32 + lrow;
      unsigned msk = 0;
      const int gr = gg.g0 + m;
      if (gr < a.total_rows) {
        const int pix = gr % HW;
        const int h = pix / W, w = pix - h * W;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
          if (hh >= 0 && hh < H && ww >= 0 && ww < W) msk |= 1u << t;
        }
      }
      tapmask[i] = msk;
    }
  };
  set_masks(g);

  // `acc` collects the tile being multiplied.  A finished tile (bias added) is written out one small step per k16 step of the NEXT tile,
  // through this wave's private 32-pixel LDS staging area, so that a store instruction covers four whole 256-byte rows (an accumulator
  // lane holds 4 x 4 channels of ONE pixel: stored directly that is 64 write requests of 16 bytes per instruction): its first half
  // (pixels i = 0) goes to the staging area at the tile's end, its second half waits in `prev1` until the first has left.  Nothing is
  // ever waited for (vmcnt counts stores; these waves never wait on vmcnt), and a CU's 64 KB of output leave spread over 24 of the
  // next tile's 36+ steps instead of as one burst at every tile end.
  f32x16 acc[2][2], prev1[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) { acc[0][j][r] = 0.f; acc[1][j][r] = 0.f; prev1[j][r] = 0.f; }

  const int pitch = MODE ? 18 : a.pitch;
  int abase0[2][3];  // byte offsets into patch buffer 0; the kernel column and the hi | lo plane are immediate offsets
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int m = wm * 64 + i * 32 + lrow;
    const int prow = MODE ? (wm * 4 + i * 2 + tr_l + 1) * 18 + pc_l + 1 : m + a.halo;
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) abase0[i][kh] = ((prow + (kh - 1) * pitch - 1) * CROW2 + lk * (F32 ? 16 : 8)) * 2;
  }
  int abase[2][3];
  const int woff0 = 2 * a.patch_bytes + lane * 16;  // byte offset of this lane's 16 bytes in plane 0 of weight stage 0
  int woff = woff0;
  auto load_ab = [&](uint4 (&av)[4], uint4 (&bv)[4], int tap) {
    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const unsigned char* q = smem_b + abase[i][kh] + kw * (CROW2 * 2);
      uint4 vh = *reinterpret_cast<const uint4*>(q);
      uint4 vl = *reinterpret_cast<const uint4*>(q + (F32 ? 16 : CK2 * 2));
      if (!MODE && !((tapmask[i] >> tap) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
      av[2 * i] = vh;
      av[2 * i + 1] = vl;
    }
#pragma unroll
    for (int pl = 0; pl < 4; ++pl) bv[pl] = *reinterpret_cast<const uint4*>(smem_b + woff + (tap * 4 + pl) * 1024);
  };
  auto mma_step = [&](const uint4 (&av)[4], const uint4 (&b)[4]) {
    if constexpr (F32) {
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
    // weights are the MFMA "A" (rows = output channels), pixels the "B" (columns): a lane holds 4 x 4 consecutive output channels of
    // one pixel.  Pass-major order: consecutive MFMAs never share an accumulator.
    auto mm = [&](const bf16x8& pix, const bf16x8& wgt, f32x16 c) { return vmm_mfma16(wgt, pix, c); };
    acc[0][0] = mm(al0, bh0, acc[0][0]);
    acc[0][1] = mm(al0, bh1, acc[0][1]);
    acc[1][0] = mm(al1, bh0, acc[1][0]);
    acc[1][1] = mm(al1, bh1, acc[1][1]);
    acc[0][0] = mm(ah0, bl0, acc[0][0]);
    acc[0][1] = mm(ah0, bl1, acc[0][1]);
    acc[1][0] = mm(ah1, bl0, acc[1][0]);
    acc[1][1] = mm(ah1, bl1, acc[1][1]);
    acc[0][0] = mm(ah0, bh0, acc[0][0]);
    acc[0][1] = mm(ah0, bh1, acc[0][1]);
    acc[1][0] = mm(ah1, bh0, acc[1][0]);
    acc[1][1] = mm(ah1, bh1, acc[1][1]);
  };

  // ---- writing a finished tile out, 17 units: 0-7 read row 4 k + er of the staged first half back (16 lanes per 256-byte row) and
  // store it, 8 stages the second half (eight 16-byte pieces per lane), 9-16 read / store its rows.  ONE copy of the code, unit index
  // at run time: specialised copies of the step loop made the accumulators change registers between copies (hundreds of spills).
  const int ep_off = ep_base + wm * EP_WAVE;
  const int er = lane >> 4, ec = lane & 15;
  const unsigned half_step = (MODE ? 2u * (unsigned)W : 32u) * (unsigned)p.ldo * 4u;  // half 1 starts 2 image rows (flat: 32 rows) further
  Geo gp = g;          // geometry of the tile being written out
  int uidx = 17;       // next unit (17 = nothing pending)
  char* obase = nullptr;
  // A unit is split around the step's MFMAs: its staging read is requested BEFORE the step's operand reads (LDS answers in order, so
  // it is back long before them), its store is issued AFTER the step's MFMAs have been issued -- placed between the operand reads and
  // the MFMAs, the store's wait for its data held up the MFMAs by an LDS round trip at every unit (+500 cycles per step, measured).
  f32x4 ustage = {0.f, 0.f, 0.f, 0.f};
  auto unit_read = [&]() {
    if (uidx >= 17 || uidx == 8) return;  // (wave-uniform)
    const int k = uidx > 8 ? uidx - 9 : uidx;
    ustage = *reinterpret_cast<const f32x4*>(smem_b + ep_off + (4 * k + er) * EP_PITCH + ec * 16);
  };
  // one bookkeeping region per step, behind the step's MFMAs: finish unit uidx (store the row its read fetched / stage half 1), then
  // request the next unit's row -- one scalar test and one branch per step when nothing is pending
  auto unit_write = [&]() {
    if (uidx >= 17) return;
    if (uidx == 8) {
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int gq2 = 0; gq2 < 4; ++gq2) {
          const f32x4 v = {prev1[j][4 * gq2], prev1[j][4 * gq2 + 1], prev1[j][4 * gq2 + 2], prev1[j][4 * gq2 + 3]};
          *reinterpret_cast<f32x4*>(smem_b + ep_off + lrow * EP_PITCH + (j * 32 + 8 * gq2 + 4 * lk) * 4) = v;
        }
    } else {
      const int h = uidx > 8, k = h ? uidx - 9 : uidx;
      const int r = 4 * k + er;  // MFMA column (pixel) r of the half
      const int rel = MODE ? (wm * 4 + (r >> 4)) * W + (r < 16 ? r : ((r - 2) & 15)) : wm * 64 + r;
      bool ok = true;
      if (!MODE) ok = gp.g0 + wm * 64 + h * 32 + r < a.total_rows;
      if (ok && !(a.dbg & 32)) *reinterpret_cast<f32x4*>(obase + ((unsigned)rel * (unsigned)p.ldo + ec * 4) * 4u + (h ? half_step : 0u)) = ustage;
    }
    ++uidx;
    unit_read();
  };

  // Tile end: bias (from LDS) added, first half staged, second half parked in prev1, accumulators cleared; and the GroupNorm statistics
  // of the tile (vddp.py:274-279) while it is in registers: per run of 8 consecutive output channels the sum and the sum of squares over
  // this wave's 64 pixels -- reduce-scatter butterfly (8 + 4 + 2 + 1 shuffles, then two plain steps) -- written as this WAVE's own slot
  // of the (sample, group)'s contribution list: no barrier, no atomics.  The previous tile's 24 units are all done by now (plan_pw: at
  // least four chunks, i.e. 36 steps, per tile).
  auto finish_tile = [&](const Geo& gg) {
    float gv[16];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int gq2 = 0; gq2 < 4; ++gq2) {
        const f32x4 b4 = *reinterpret_cast<const f32x4*>(smem_b + bias_off + (j * 32 + 8 * gq2 + 4 * lk) * 4);
        const f32x4 v0 = {acc[0][j][4 * gq2] + b4.x, acc[0][j][4 * gq2 + 1] + b4.y, acc[0][j][4 * gq2 + 2] + b4.z, acc[0][j][4 * gq2 + 3] + b4.w};
        const f32x4 v1 = {acc[1][j][4 * gq2] + b4.x, acc[1][j][4 * gq2 + 1] + b4.y, acc[1][j][4 * gq2 + 2] + b4.z, acc[1][j][4 * gq2 + 3] + b4.w};
        *reinterpret_cast<f32x4*>(smem_b + ep_off + lrow * EP_PITCH + (j * 32 + 8 * gq2 + 4 * lk) * 4) = v0;
        prev1[j][4 * gq2] = v1.x; prev1[j][4 * gq2 + 1] = v1.y; prev1[j][4 * gq2 + 2] = v1.z; prev1[j][4 * gq2 + 3] = v1.w;
        gv[(j * 4 + gq2) * 2] = ((v0.x + v0.y) + (v0.z + v0.w)) + ((v1.x + v1.y) + (v1.z + v1.w));
        gv[(j * 4 + gq2) * 2 + 1] = ((v0.x * v0.x + v0.y * v0.y) + (v0.z * v0.z + v0.w * v0.w)) + ((v1.x * v1.x + v1.y * v1.y) + (v1.z * v1.z + v1.w * v1.w));
#pragma unroll
        for (int i = 0; i < 2; ++i) { acc[i][j][4 * gq2] = 0.f; acc[i][j][4 * gq2 + 1] = 0.f; acc[i][j][4 * gq2 + 2] = 0.f; acc[i][j][4 * gq2 + 3] = 0.f; }
      }
    gp = gg;
    uidx = 0;
    unit_read();  // (after the staging stores above: a wave's LDS operations execute in order)
    const long long row0 = MODE ? (long long)gg.img * HW + (long long)gg.ty0 * W + gg.tx0 : (long long)gg.g0;
    obase = reinterpret_cast<char*>(p.out + row0 * p.ldo + gg.n0);
    if (MODE && p.gn_part && !(a.dbg & 64)) {
#pragma unroll
      for (int bit = 5, n = 8; bit >= 2; --bit, n >>= 1) {
        const bool hi = (lane >> bit) & 1;
#pragma unroll
        for (int k = 0; k < n; ++k) {
          const float send = hi ? gv[k] : gv[k + n], keep = hi ? gv[k + n] : gv[k];
          gv[k] = keep + lane_xor(send, bit);
        }
      }
      gv[0] += lane_xor(gv[0], 1);
      gv[0] += lane_xor(gv[0], 0);  // lane L now holds the wave total of value (L >> 2) & 15
      if ((lane & 3) == 0) {
        const int slot = lane >> 2, run = slot >> 1;
        const int cout0 = gg.n0 + (run >> 2) * 32 + (run & 3) * 8;
        const int cpg = p.Cout / p.gn_groups, rpg = cpg >> 3;
        const int grp = cout0 / cpg, rig = (cout0 - grp * cpg) >> 3;
        const int smp = gg.img / p.a_imgs_per_sample, fr = gg.img - smp * p.a_imgs_per_sample;
        const int n_contrib = p.a_imgs_per_sample * a.tiles_per_frame * 4 * rpg;
        const int k = ((fr * a.tiles_per_frame + (gg.mtile - gg.img * a.tiles_per_frame)) * 4 + wm) * rpg + rig;
        p.gn_part[(((long long)smp * p.gn_groups + grp) * n_contrib + k) * 2 + (slot & 1)] = gv[0];
      }
    }
  };

  // nine k16 steps (taps) per chunk; the operands of step t + 1 are requested before the MFMAs of step t; one unit of the previous
  // tile's write-out rides along with each of the first 17 steps of a tile
  uint4 aa[2][4], bb[2][4];
  auto chunk_steps = [&]() {
    load_ab(aa[0], bb[0], 0);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      if (t + 1 < 9) load_ab(aa[(t + 1) & 1], bb[(t + 1) & 1], t + 1);
      mma_step(aa[t & 1], bb[t & 1]);
      // The matrix pipe takes a new MFMA every 32 cycles and nothing queues behind the one in flight: whatever else this wave issues between
      // two MFMA groups is a bubble.  The next step's eight operand reads therefore go BETWEEN this step's MFMAs, one per MFMA (all of them
      // in front of the group cost ~50 cycles per step).
      if (t + 1 < 9) {
#pragma unroll
        for (int k = 0; k < 8; ++k) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);  // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // one LDS read
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      unit_write();
      __builtin_amdgcn_sched_barrier(0);
    }
  };
  PW_BARRIER_LDS();  // phase 0 is staged (and the bias is parked)
  int ph = 0;
  for (int i = 0; i < cnt; ++i) {
    for (int cc = 0; cc < nchunks; ++cc, ++ph) {
      const int poff = (ph & 1) * a.patch_bytes;
#pragma unroll
      for (int ii = 0; ii < 2; ++ii)
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
          abase[ii][kh] = abase0[ii][kh] + poff;
          asm volatile("" : "+v"(abase[ii][kh]));  // keep the six bases opaque: the per-step addresses stay base + immediate
        }
      woff = woff0 + (ph & 1) * PW_STAGE;
      asm volatile("" : "+v"(woff));
      chunk_steps();
      if (cc + 1 == nchunks) finish_tile(g);
      PW_BARRIER_LDS();  // phase ph + 1 is staged in the other buffers; this phase's buffers may be overwritten
    }
    if (i + 1 < cnt) {
      g = geo_of(i + 1);
      set_masks(g);
    }
  }
  // the last tile's write-out
  while (uidx < 17) unit_write();
#undef PW_BARRIER_LDS
#undef PW_BARRIER_DMA
}

template <int MODE, bool F32>
int launch_pw(const PWArgs& a, hipStream_t s) {
  const size_t shm = (size_t)2 * a.patch_bytes + 2 * PW_STAGE + 4 * 32 * 272;  // patches, weight stages, epilogue staging = 160 KB
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_pw_kernel<MODE, F32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // one workgroup per CU (256 CUs, 8 XCDs); fewer tiles than CUs: one tile each, rounded up to whole XCD groups
  const int grid = a.n_units >= 256 ? 256 : ((a.n_units + 7) / 8) * 8;
  static const int trace_launch = [] { const char* e = getenv("VMM_PW_TRACE"); return e ? atoi(e) : -1; }();
  static int launch_no = 0;
  static unsigned long long* trace_buf = nullptr;
  PWArgs aa = a;
  const bool tracing = trace_launch >= 0 && launch_no++ == trace_launch;
  if (tracing) {
    if (!trace_buf) (void)hipMalloc(&trace_buf, (128 * 16 + 16) * sizeof(unsigned long long));
    (void)hipMemset(trace_buf, 0, (128 * 16 + 16) * sizeof(unsigned long long));
    aa.trace = trace_buf;
  }
  hipLaunchKernelGGL((conv3x3_pw_kernel<MODE, F32>), dim3((unsigned)grid), dim3(512), shm, s, aa);
  VMM_LAUNCH_CHECK();
  if (tracing) {  // measurement aid only: synchronises the device and prints workgroup 0's barrier time line
    (void)hipDeviceSynchronize();
    static unsigned long long h[128 * 16 + 16];
    (void)hipMemcpy(h, trace_buf, sizeof(h), hipMemcpyDeviceToHost);
    fprintf(stderr, "[pw trace] Cin %d Cout %d HxW %dx%d mode %d units %d grid %d a_mode %d\n", a.p.C1 + a.p.C2, a.p.Cout, a.p.Hin, a.p.Win, a.mode, a.n_units, grid, a.p.a_mode);
    for (int w = 0; w < 3; ++w)
      fprintf(stderr, "  patch wave %d: drain-1 %lld, requests %lld, wait for rows %lld, transform + LDS stores %lld cycles in total\n", w, (long long)h[128 * 16 + w * 4],
              (long long)h[128 * 16 + w * 4 + 1], (long long)h[128 * 16 + w * 4 + 2], (long long)h[128 * 16 + w * 4 + 3]);
    const unsigned long long t0 = h[0];
    for (int q = 0; q < 128 && h[(q * 8) * 2 + 1]; ++q) {
      fprintf(stderr, "  barrier %3d:", q);
      for (int w = 0; w < 8; ++w) fprintf(stderr, " w%d %7lld+%-6lld", w, (long long)(h[(q * 8 + w) * 2] - t0), (long long)(h[(q * 8 + w) * 2 + 1] - h[(q * 8 + w) * 2]));
      fprintf(stderr, "\n");
    }
  }
  return 0;
}

// geometry of the persistent variant: 256-pixel x 64-column tiles for every layer; false when the shape is outside its envelope
bool plan_pw(const vmm_conv_desc& d, PWArgs& a) {
  if (d.wrap_h || d.wrap_w || d.a_img_mod) return false;  // (periodic padding, shared source frames: the one-tile kernel's 2-D instances)
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  a.p = d;
  a.n_tiles = d.Cout / 64;
  a.total_rows = (int)M;
  a.KS = 9 * (d.C1 + d.C2) / 16;
  int mtiles;
  if (d.Win >= 32 && d.Win % 16 == 0 && d.Hin % 16 == 0) {
    a.mode = 1;
    a.tiles_x = d.Win / 16;
    a.tiles_per_frame = a.tiles_x * (d.Hin / 16);
    a.pitch = 18;
    a.halo = 0;
    a.PR = 18 * 18;
    mtiles = d.nimg * a.tiles_per_frame;
  } else {
    a.mode = 0;
    a.tiles_x = a.tiles_per_frame = 1;
    a.pitch = d.Win;
    a.halo = d.Win + 1;
    a.PR = 256 + 2 * a.halo;
    mtiles = (int)cdiv(M, 256);
  }
  if (a.PR > PW_MAXP * 48 || d.Cout % 64) return false;
  if ((d.C1 + d.C2) / CK2 < 2) return false;  // a tile's 17 write-out units ride on the next tile's steps: at least 18 of them
  if (d.res) return false;                     // (gradient accumulation into an existing tensor stays on the one-tile-per-workgroup kernel)
  if (d.C2 > 0 && d.lda2 != d.lda1) return false;  // the patch waves keep ONE byte offset per row
  // flat tiles with the fused transform: a patch may touch two samples, not three
  if (a.mode == 0 && d.a_mode == 1 && (long long)d.Hin * d.Win * d.a_imgs_per_sample < a.PR) return false;
  a.patch_bytes = (PW_MAXP * 48 * CROW2 * 2 + 1023) / 1024 * 1024;  // room for every item of every patch thread: stores need no row check
  a.n_units = mtiles * a.n_tiles;
  static const int dbg = [] { const char* e = getenv("VMM_PW_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  a.trace = nullptr;
  static const int stagger_env = [] { const char* e = getenv("VMM_PW_STAGGER"); return e ? atoi(e) : -1; }();
  const int nchunks = (d.C1 + d.C2) / CK2;
  // VMM_PW_STAGGER = cycles per 16-channel chunk between the starts of consecutive workgroup quarters (measurement aid)
  a.stagger = stagger_env > 0 ? stagger_env * nchunks : 0;  // (measured without effect on the 96 x 96 layers: off unless asked for)
  return true;
}

#endif  // VMM_EXPERIMENTS (persistent kernel)

inline int* c3_launch_counter() { static int n = 0; return &n; }  // one count over all instances

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32, bool ONE = false>
int launch_c3(const C3Args& a, int mtiles, int ksplit, hipStream_t s) {
  const size_t shm = sizeof(unsigned short) * ((size_t)a.PR + 3) * CROW;  // (+ three rows of zeros for the masked taps of flat row tiles)
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, SPLIT, F32, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  // VMM_C3_TRACE=<k>: the k-th launch of this process dumps its workgroups' phase stamps to VMM_C3_TRACE_FILE (tools/trace_c3.py reads them)
  static const int trace_launch = [] { const char* e = getenv("VMM_C3_TRACE"); return e ? atoi(e) : -1; }();
  static int* launch_no = c3_launch_counter();
  const unsigned nwg = (unsigned)(mtiles * a.n_tiles);
  if (trace_launch >= 0 && (*launch_no)++ == trace_launch) {
    C3Args at = a;
    const size_t n = (size_t)nwg * ksplit * 16;
    (void)hipMalloc(&at.trace, n * sizeof(unsigned long long));
    (void)hipMemsetAsync(at.trace, 0, n * sizeof(unsigned long long), s);
    hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, SPLIT, F32, ONE>), dim3(nwg, ksplit), dim3(256), shm, s, at);
    (void)hipStreamSynchronize(s);
    std::vector<unsigned long long> h(n);
    (void)hipMemcpy(h.data(), at.trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(at.trace);
    const char* fn = getenv("VMM_C3_TRACE_FILE");
    FILE* f = fopen(fn ? fn : "c3_trace.txt", "w");
    if (f) {
      fprintf(f, "# Cin %d Cout %d HxW %dx%d mode %d WM %d WN %d nwg %u ksplit %d a_mode %d\n", a.p.C1 + a.p.C2, a.p.Cout, a.p.Hin, a.p.Win, MODE, WM, WN, nwg, ksplit, a.p.a_mode);
      for (size_t w = 0; w < (size_t)nwg * ksplit; ++w) {
        for (int k = 0; k < 16; ++k) fprintf(f, "%llu ", h[w * 16 + k]);
        fprintf(f, "\n");
      }
      fclose(f);
    }
    return 0;
  }
  hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, SPLIT, F32, ONE>), dim3(nwg, ksplit), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

#if VMM_EXPERIMENTS
template <int MAXP, int MODE, int PFB, bool F32, bool ONE>
int launch_c3_sk(const C3Args& a, int grid, hipStream_t s) {
  const size_t shm = sizeof(unsigned short) * ((size_t)a.PR + 3) * CROW;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_sk_kernel<MAXP, MODE, PFB, F32, ONE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_sk_kernel<MAXP, MODE, PFB, F32, ONE>), dim3((unsigned)grid, 1), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

template <int MAXP, int MODE, int PFB>
int launch_c3n(const C3Args& a, int mtiles, hipStream_t s) {
  const size_t shm = sizeof(unsigned short) * ((size_t)a.PR + 3) * CROW;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3n_kernel<MAXP, MODE, PFB>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_x3n_kernel<MAXP, MODE, PFB>), dim3((unsigned)(mtiles * a.n_tiles), 1), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
#endif

template <int WM, int WN, int MAXP, int MODE, int TS, bool SPLIT = false, bool ONE = false, int A16 = 0>
int launch_s2(const C3Args& a, int mtiles, hipStream_t s, int ksplit = 1) {
  const size_t shm = sizeof(unsigned short) * ((size_t)a.PR + 3) * CROW;  // (+ three rows of zeros for the masked taps of flat row tiles)
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_s2_kernel<WM, WN, MAXP, MODE, TS, SPLIT, ONE, A16>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv_s2_kernel<WM, WN, MAXP, MODE, TS, SPLIT, ONE, A16>), dim3((unsigned)(mtiles * a.n_tiles), (unsigned)ksplit), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// the single-pass 3 x 3 instances over bf16-stored maps (A16 = 1: in and out)
template <int WM, int WN, int MAXP, int MODE, int PFB>
int launch_c3_a16(const C3Args& a, int mtiles, hipStream_t s) {
  const size_t shm = sizeof(unsigned short) * ((size_t)a.PR + 3) * CROW;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, false, false, true, 1>), hipFuncAttributeMaxDynamicSharedMemorySize,
                        160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, false, false, true, 1>), dim3((unsigned)(mtiles * a.n_tiles), 1), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// shape planning shared by the launcher and the query below; returns 0 and fills a / mtiles / ksplit / gn, or the launcher's status
// 64-column layers on 2-D tiles: the 32-column-wave-tile instance (128-pixel tiles, three workgroups per CU) for the split-bf16 arithmetic, and its
// GroupNorm slot layout (tiles of 8 pixel rows) for every arithmetic.  A function of the descriptor (and the process environment) alone.
static bool c3_nj1(const vmm_conv_desc& d) {
#if VMM_EXPERIMENTS
  static const int on = [] {
    const char* e = getenv("VMM_C3_NJ1");
    const char* pw = getenv("VMM_C3_PERSISTENT");
    return (e ? atoi(e) : 0) && !(pw && atoi(pw));
  }();
  return on && d.Cout == 64 && d.Win >= 32 && d.Win % 16 == 0 && d.Hin % 16 == 0;
#else
  (void)d;
  return false;
#endif
}

int plan_c3(const vmm_conv_desc& d, C3Args& a, int& mtiles, int& ksplit, bool& gn, bool x3 = false) {
  const bool shape_ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.off_h == -1 && d.off_w == -1 && d.sgn_h == 1 && d.sgn_w == 1 &&
                        d.Hv == d.Hin && d.Wv == d.Win && d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 &&
                        d.rot_ncols == 0 && d.q_ncols == 0;
  const bool chan_ok = (d.C1 % CK == 0) && (d.C2 % CK == 0) && d.C1 > 0 && (d.Cout == 64 || d.Cout % 128 == 0) && (d.lda1 & 3) == 0 &&
                       (!d.C2 || (d.lda2 & 3) == 0) && (d.ldo & 3) == 0 && (!d.res || (d.ldres & 3) == 0);
  if (!shape_ok || !chan_ok) return 1;
  if (d.a_mode == 1 && (!d.a_coef || d.a_imgs_per_sample <= 0)) return -3;
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  if (M >= (1LL << 31)) return -4;
  const bool wide = d.Cout >= 128;  // 2 x 2 waves, 128 pixels x 128 columns; else 4 x 1 waves, 256 pixels x 64 columns
  const int BM = wide ? 128 : 256, TH = BM / 16;
  a.p = d;
  a.cps_shift = 0;
  a.n_tiles = wide ? d.Cout / 128 : 1;
  a.total_rows = (int)M;
  a.KS = 9 * (d.C1 + d.C2) / 16;
  if (d.Win >= 32 && d.Win % 16 == 0 && d.Hin % TH == 0) {
    a.mode = 1;
    a.tiles_x = d.Win / 16;
    a.tiles_per_frame = a.tiles_x * (d.Hin / TH);
    a.tpf_magic = (unsigned)(0x100000000ull / (unsigned)a.tiles_per_frame) + 1u;
    a.tx_magic = (unsigned)(0x100000000ull / (unsigned)a.tiles_x) + 1u;
    a.pitch = 18;
    a.halo = 0;
    a.PR = (TH + 2) * 18;
    mtiles = d.nimg * a.tiles_per_frame;
  } else {
    a.mode = 0;
    a.tiles_x = a.tiles_per_frame = 1;
    a.tpf_magic = a.tx_magic = 0;
    a.pitch = d.Win;
    a.halo = d.Win + 1;
    a.PR = BM + 2 * a.halo;
    mtiles = (int)cdiv(M, BM);
  }
  if (a.PR > (wide ? 6 : 11) * 32) return 1;
  if ((d.wrap_h || d.wrap_w) && a.mode == 0) return 1;  // periodic padding: the flat row tiles' neighbourhood is a contiguous row range
  // few-row layers (12 x 12 level): split the channel chunks so that both workgroup slots of every CU are filled a few times over;
  // the partial sums are added in a fixed order (tickets, see the kernel epilogue)
  const int nch = (d.C1 + d.C2) / CK;
  const long long blocks = (long long)mtiles * a.n_tiles;
  ksplit = 1;
  // split the channel reduction only when the tiles cannot even half-fill the chip: measured on the Lagrangian sampler (batch 8), splitting
  // below 1024 workgroups cost 1.15 ms per step against splitting below 128 (ordered atomics epilogue, no fused GroupNorm sums)
  // How far: a split keeps at least four 32-channel chunks and there are at most four of them.  (Round 3, tools/bench_wino.py with WINO_TICKETS=1,
  // 11 / 22 / 44 frames: the earlier "up to eight" was never the best choice -- 256 -> 256 at 12 x 12, 44 frames: 71 us with eight splits, 50 with
  // two or none; 512 -> 512, 22 frames: 84 / 74 / 92 us with eight / four / none -- the ordered ticket epilogue serialises a tile's splits.)
  // Balanced launch (conv3x3_sk_kernel): 128 x 128 tiles that fill the two workgroup slots per CU unevenly.  slots = 2 x CUs; a plain launch runs
  // ceil(tiles / slots) rounds, of which the last is partly empty.  Measured on the Lagrangian sampler (LABNOTES 10.4): it pays where the tiles are more
  // than one round (24 x 24 level: 792 tiles, 0.183 -> 0.168 ms) or very long (1024 -> 512 at 12 x 12: 0.172 -> 0.162 ms); one under-full round of
  // 512 -> 512 tiles is as fast unbalanced (0.159 vs 0.161 ms: the CUs with ONE resident workgroup finish early and the chip's clock, not the slot
  // count, sets the pace), and layers of a few chunks per workgroup lose to the partial-tile traffic.  A caller that passes FEWER slots than two per
  // CU has chosen the grid itself (the kernel tests do, to reach every segment pattern on small shapes).
  a.sk_tiles = 0;
  int sk_grid = 0;
#if VMM_EXPERIMENTS
  if (wide && d.sk_work && d.split_tickets && !d.a_img_mod && !d.act_bf16) {
    static const int sk_env = [] { const char* e = getenv("VMM_C3_SK"); return e ? atoi(e) : 1; }();  // 0: off (A/B runs)
    static const int n_cu = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
      return n;
    }();
    const bool chosen = d.sk_slots < 2 * n_cu;
    const int grid = min(2 * n_cu, (int)d.sk_slots) & ~7;
    const long long rounds = grid ? (blocks + grid - 1) / grid : 0;
    const bool uneven = blocks * 100 < rounds * grid * 85;
    if (sk_env && grid >= 8 && nch >= 4 && blocks * nch >= 2LL * grid && d.n_tickets >= 2048 + grid &&
        (chosen || (uneven && blocks <= 3LL * grid && (blocks >= grid || nch >= 32)))) {
      a.sk_tiles = (int)blocks;
      sk_grid = grid;
    }
  }
#endif
  static const int ksplit_max = [] { const char* e = getenv("VMM_C3_KSPLIT_MAX"); return e ? atoi(e) : 4; }();  // (measurement aid)
  if (!a.sk_tiles && blocks < 128 && d.split_tickets && d.n_tickets >= blocks)
    ksplit = (int)max(1LL, min((long long)min(nch / 4, ksplit_max), 2048 / max(blocks, 1LL)));
  a.chunks_per_split = (int)cdiv(nch, ksplit);
  ksplit = (int)cdiv(nch, a.chunks_per_split);
  if (a.sk_tiles) a.chunks_per_split = sk_grid;  // (carries the grid size to the dispatcher; the balanced kernel does not use the field)
  if (d.a_img_mod && (a.mode == 0 || ksplit > 1 || d.a_img_mod < 0 || d.a_img_mod >= d.nimg)) return 1;  // shared source frames: unsplit 2-D tiles only
  // GroupNorm statistics of the output in the epilogue: unsplit 2-D tiles (one frame, hence one sample, per workgroup), groups of whole
  // 8-channel runs, no residual in the output
  // (flat row tiles: a tile may touch two samples, not three)
  gn = d.gn_part && d.gn_groups > 0 && ksplit == 1 && !d.res && d.a_imgs_per_sample > 0 && d.Cout % d.gn_groups == 0 &&
       (d.Cout / d.gn_groups) % 8 == 0 && d.nimg % d.a_imgs_per_sample == 0 &&
       (a.mode == 1 || (long long)d.Hin * d.Win * d.a_imgs_per_sample >= BM);
  if (!gn) a.p.gn_part = nullptr;
  a.gn_fine = 0;
  if (a.mode == 1 && ksplit == 1 && c3_nj1(d)) {
    if (x3) {  // 8 x 16 pixel tiles, 2 x 2 waves of 64 pixels x 32 columns
      a.tiles_per_frame = a.tiles_x * (d.Hin / 8);
      a.tpf_magic = (unsigned)(0x100000000ull / (unsigned)a.tiles_per_frame) + 1u;
      a.PR = (8 + 2) * 18;
      mtiles = d.nimg * a.tiles_per_frame;
      a.n_tiles = 1;
      a.gn_fine = 2;  // (marks the instance for the dispatcher)
    } else {
      a.gn_fine = 1;
    }
  }
  return 0;
}

}  // namespace

#if VMM_EXPERIMENTS
// unsplit layers run the persistent wave-specialised kernel (VMM_C3_LEGACY=1 in the environment keeps the one-tile-per-workgroup
// kernel for A/B measurements; read once)
// 0: never (default), 1: 2-D tiles of 64-column layers, 2: every shape inside its envelope.
// Default 0 although the persistent kernel is the faster KERNEL on those layers (185 against 205 us, launches timed one by one): inside the
// captured sampling step the chip's clock governor answers its denser matrix-pipe activity with a lower sustained clock for the whole step
// (rocm-smi during bench.py on one box: 2.16-2.17 GHz at 1.11 kW with it, 2.27-2.30 GHz at 1.20-1.24 kW without; staggering the
// workgroups' starts changes nothing), and the step as a whole came out 0.15-0.4 ms SLOWER on three of four boxes (0.3 ms faster on
// the fourth).  DESIGN.md section 7.
static int c3_persistent() {
  static const int mode = [] {
    const char* l = getenv("VMM_C3_LEGACY");
    if (l && l[0] == '1') return 0;
    const char* e = getenv("VMM_C3_PERSISTENT");
    return e ? atoi(e) : 0;
  }();
  return mode;
}
static bool c3_use_pw(const vmm_conv_desc& d, int ksplit, PWArgs& pa) {
  const int mode = c3_persistent();
  if (mode == 0 || ksplit != 1 || !plan_pw(d, pa)) return false;
  return mode >= 2 || (pa.mode == 1 && d.Cout == 64);
}
// measurement aid: VMM_PW_ONLY=k keeps the persistent kernel for the k-th eligible launch of the process only (layer bisection)
static bool c3_pw_this_launch() {
  static const int only = [] { const char* e = getenv("VMM_PW_ONLY"); return e ? atoi(e) : -1; }();
  static int n = 0;
  return only < 0 || n++ == only;
}
#endif

template <bool F32, bool ONE = false>
int dispatch_c3(const C3Args& a, int mtiles, int ksplit, bool wide, hipStream_t s) {
#if VMM_EXPERIMENTS
  PWArgs pa;
  if (!ONE && a.gn_fine != 2 && c3_use_pw(a.p, ksplit, pa) && c3_pw_this_launch()) {
    pa.p.gn_part = pa.mode == 1 ? a.p.gn_part : nullptr;  // (cleared by plan_c3 when the statistics are not fused; the persistent kernel fuses them for 2-D tiles only)
    if (getenv("VMM_PW_LOG")) fprintf(stderr, "[pw] Cin %d+%d Cout %d %dx%d nimg %d mode %d units %d a_mode %d gn %p res %p lda %d %d ldo %d\n", a.p.C1, a.p.C2, a.p.Cout, a.p.Hin,
                                      a.p.Win, a.p.nimg, pa.mode, pa.n_units, a.p.a_mode, (void*)pa.p.gn_part, (void*)a.p.res, a.p.lda1, a.p.lda2, a.p.ldo);
    return pa.mode ? launch_pw<1, F32>(pa, s) : launch_pw<0, F32>(pa, s);
  }
#endif
  if (ksplit > 1) {
    if (wide) return a.mode ? launch_c3<2, 2, 6, 1, 2, true, F32, ONE>(a, mtiles, ksplit, s) : launch_c3<2, 2, 6, 0, 2, true, F32, ONE>(a, mtiles, ksplit, s);
    return a.mode ? launch_c3<4, 1, 11, 1, 1, true, F32, ONE>(a, mtiles, ksplit, s) : launch_c3<4, 1, 11, 0, 1, true, F32, ONE>(a, mtiles, ksplit, s);
  }
#if VMM_EXPERIMENTS
  if (wide && a.sk_tiles)
    return a.mode ? launch_c3_sk<6, 1, 2, F32, ONE>(a, a.chunks_per_split, s) : launch_c3_sk<6, 0, 2, F32, ONE>(a, a.chunks_per_split, s);
#endif
  if (wide) return a.mode ? launch_c3<2, 2, 6, 1, 2, false, F32, ONE>(a, mtiles, 1, s) : launch_c3<2, 2, 6, 0, 2, false, F32, ONE>(a, mtiles, 1, s);
#if VMM_EXPERIMENTS
  if constexpr (!F32 && !ONE)
    if (a.gn_fine == 2) return launch_c3n<6, 1, 2>(a, mtiles, s);
#endif
  // (weight fragments two steps ahead where the registers allow it: with the requests interleaved into the MFMA stream one step is less
  // than an L2 round trip)
  constexpr int PF = F32 ? 1 : 2;
  return a.mode ? launch_c3<4, 1, 11, 1, PF, false, F32, ONE>(a, mtiles, 1, s) : launch_c3<4, 1, 11, 0, PF, false, F32, ONE>(a, mtiles, 1, s);
}

#if !VMM_FP16_OPERANDS  // (the fp16-operand build of this file exports its single-pass instances alone: the end of the file)
// Number of GroupNorm partial-sum pairs per (sample, group) vmm_conv3x3_bf16x3(d) will leave in d->gn_part (the caller then skips
// vmm_groupnorm_stats and hands them to vmm_groupnorm_coef), 0 when it will not.  Pure host logic.
extern "C" int vmm_conv3x3_fuses_gn(const vmm_conv_desc* dp) {
  if (getenv("VMM_NO_GN_FUSE")) return 0;  // measurement aid
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  if (plan_c3(*dp, a, mtiles, ksplit, gn) != 0 || !gn) return 0;
  const int rpg = (dp->Cout / dp->gn_groups) >> 3;
#if VMM_EXPERIMENTS
  PWArgs pa;
  if (c3_use_pw(*dp, ksplit, pa))  // the persistent kernel (2-D tiles only) leaves one slot per 64-pixel wave tile of its 256-pixel tiles
    return pa.mode == 1 ? dp->a_imgs_per_sample * pa.tiles_per_frame * 4 * rpg : 0;
#endif
  if (a.mode == 1) return dp->a_imgs_per_sample * a.tiles_per_frame * rpg * (a.gn_fine ? 2 : 1);
  const int BM = dp->Cout >= 128 ? 128 : 256;
  const long long R = (long long)dp->Hin * dp->Win * dp->a_imgs_per_sample;
  return (int)((R + BM - 1) / BM + 1) * rpg;  // flat row tiles: the most tiles a sample can touch
}

// host-only query: 1 when vmm_conv3x3_bf16x3 / vmm_conv3x3_f32 would take this descriptor as it stands (fused operand transform, a_img_mod, periodic
// padding, residual ...), 0 when the launcher would return non-zero.  Lets a plan builder decide on optional features (shared source frames)
// without restating the kernel's envelope.
extern "C" int vmm_conv3x3_accepts(const vmm_conv_desc* dp) {
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  return plan_c3(*dp, a, mtiles, ksplit, gn) == 0 ? 1 : 0;
}

// Weights: vmm_pack_weights fmt 2 (MFMA fragment order).  Returns 1 (nothing launched) when the descriptor is outside this
// kernel's envelope: 3x3 / stride 1 / pad 1, C1 % 32 == C2 % 32 == 0, Cout == 64 or Cout % 128 == 0, W <= 31 or (W % 16 == 0 and
// H % 16 == 0) -- the caller then uses vmm_conv_igemm_bf16x3 with fmt-1 weights.
extern "C" int vmm_conv3x3_bf16x3(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  if (d.act_bf16) return -1;  // (bf16-stored maps: the single-pass entry point only)
  const int rc = plan_c3(d, a, mtiles, ksplit, gn, true);
  if (rc != 0) return rc;
  if (a.total_rows <= 0) return 0;
  const bool wide = d.Cout >= 128;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_c3<false>(a, mtiles, ksplit, wide, s);
}

// The "bf16" throughput mode of the same kernel (BASELINE.json configs[3]): same descriptor, same fmt-2 weights, ONE matrix pass on the operands'
// bf16 roundings (2^-9 relative per operand; fp32 accumulation, bias / residual / GroupNorm sums in fp32 as before).
extern "C" int vmm_conv3x3_bf16(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  const int rc = plan_c3(d, a, mtiles, ksplit, gn);
  if (rc != 0) return rc;
  if (a.total_rows <= 0) return 0;
  if (d.act_bf16) {  // bf16-stored maps: input(s), residual and output alike; the unsplit instances (the upper levels have tiles to spare)
    if (d.act_bf16 != 3 || ksplit != 1) return -1;
    hipStream_t s = (hipStream_t)stream;
    if (d.Cout >= 128) return a.mode ? launch_c3_a16<2, 2, 6, 1, 2>(a, mtiles, s) : launch_c3_a16<2, 2, 6, 0, 2>(a, mtiles, s);
    return a.mode ? launch_c3_a16<4, 1, 11, 1, 2>(a, mtiles, s) : launch_c3_a16<4, 1, 11, 0, 2>(a, mtiles, s);
  }
  return dispatch_c3<false, true>(a, mtiles, ksplit, d.Cout >= 128, (hipStream_t)stream);
}

#endif  // !VMM_FP16_OPERANDS

// The resampling layers on the tap-subset variants of the kernel (TS, see there):
//   up == 0: Conv3d (1,4,4) stride (1,2,2) pad (0,1,1) (Downsample, vddp.py:158): x [nimg][Hin][Win][Cin] -> out [nimg][Hin/2][Win/2][Cout],
//            weights = vmm_pack_weights fmt 5 of the (Cout, Cin, 1, 4, 4) tensor;
//   up == 1: ConvTranspose3d (1,4,4) stride (1,2,2) pad (0,1,1) (Upsample, vddp.py:155): -> out [nimg][2 Hin][2 Win][Cout], fmt 6 of (Cin, Cout, 1, 4, 4).
// Returns 1 (nothing launched) outside the envelope: Cin a power of two >= 32, Cout == 64 or Cout % 128 == 0, even Hin / Win for up == 0, and the
// tile space (input pixels for up == 1, 2 x 2 input cells for up == 0) 2-D-tileable (W >= 32, W % 16 == 0, H % 8 (16 for 64 output columns) == 0)
// or at most 31 wide.
static int s2_plan(const float* x, int32_t ldx, const float* w_frag, const float* bias, float* out, int32_t ldo, int32_t nimg, int32_t Hin,
                   int32_t Win, int32_t Cin, int32_t Cout, int32_t up, C3Args& a, int& mtiles, bool& wide) {
  if (Cin < 32 || (Cin & (Cin - 1)) || !(Cout == 64 || Cout % 128 == 0) || (ldx & 3) || (ldo & 3) || Hin <= 0 || Win <= 0 || nimg <= 0) return 1;
  if (!up && ((Hin | Win) & 1)) return 1;
  const int Ht = up ? Hin : Hin / 2, Wt = up ? Win : Win / 2;
  const long long M = (long long)nimg * Ht * Wt;
  if (M * 4 >= (1LL << 31)) return 1;
  const int ncols = up ? 4 * Cout : Cout;
  wide = ncols >= 128;
  const int BM = wide ? 128 : 256, TH = BM / 16;
  a = C3Args{};
  a.p.a1 = x; a.p.C1 = Cin; a.p.lda1 = ldx; a.p.w = w_frag; a.p.bias = bias; a.p.out = out; a.p.ldo = ldo;
  a.p.nimg = nimg; a.p.Hin = Ht; a.p.Win = Wt; a.p.Cout = Cout; a.p.a_imgs_per_sample = 1;
  a.n_tiles = ncols / (wide ? 128 : 64);
  a.total_rows = (int)M;
  a.KS = 9 * (up ? 1 : 4) * Cin / 16;
  a.chunks_per_split = (up ? 1 : 4) * Cin / CK;
  a.cps_shift = 0;
  while ((CK << a.cps_shift) < Cin) ++a.cps_shift;
  if (Wt >= 32 && Wt % 16 == 0 && Ht % TH == 0) {
    a.mode = 1;
    a.tiles_x = Wt / 16;
    a.tiles_per_frame = a.tiles_x * (Ht / TH);
    a.tpf_magic = (unsigned)(0x100000000ull / (unsigned)a.tiles_per_frame) + 1u;
    a.tx_magic = (unsigned)(0x100000000ull / (unsigned)a.tiles_x) + 1u;
    a.pitch = 18;
    a.PR = (TH + 2) * 18;
    mtiles = nimg * a.tiles_per_frame;
  } else {
    a.mode = 0;
    a.tiles_x = a.tiles_per_frame = 1;
    a.tpf_magic = a.tx_magic = 0;
    a.pitch = Wt;
    a.halo = Wt + 1;
    a.PR = BM + 2 * a.halo;
    mtiles = (int)cdiv(M, BM);
  }
  if (a.PR > (wide ? 6 : 11) * 32) return 1;
  return 0;
}

#if !VMM_FP16_OPERANDS
// 1 when vmm_conv_s2_bf16x3 takes the shape, 0 when it would return 1 (pure host logic)
extern "C" int vmm_conv_s2_supported(int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up) {
  C3Args a;
  int mtiles;
  bool wide;
  return s2_plan(nullptr, 0, nullptr, nullptr, nullptr, 0, nimg, Hin, Win, Cin, Cout, up, a, mtiles, wide) == 0 ? 1 : 0;
}

#endif

// res != NULL: out = convolution (+ bias) + res, res rows indexed like out (may alias it): the layers' DATA gradients accumulating into a gradient
// buffer that already holds the skip connection's share (autograd of vddp.py:155,158).  split_tickets / n_tickets as in vmm_conv_desc: the
// Downsample form (up == 0) of a few-tile layer (12 x 12 outputs: 100 workgroups walking K = 9216) splits its channel reduction like the 3 x 3 kernel.
template <bool ONE>
static int s2_run(const float* x, int32_t ldx, const float* w_frag, const float* bias, const float* res, int32_t ldres, float* out, int32_t ldo, int32_t nimg,
                  int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets, int32_t n_tickets, vmm_stream_t stream) {
  C3Args a;
  int mtiles;
  bool wide;
  const int rc = s2_plan(x, ldx, w_frag, bias, out, ldo, nimg, Hin, Win, Cin, Cout, up, a, mtiles, wide);
  if (rc) return rc;
  if (res && (ldres & 3)) return 1;
  a.p.res = res;
  a.p.ldres = ldres;
  hipStream_t s = (hipStream_t)stream;
  if (up) return a.mode ? launch_s2<2, 2, 6, 1, 1, false, ONE>(a, mtiles, s) : launch_s2<2, 2, 6, 0, 1, false, ONE>(a, mtiles, s);
  const long long blocks = (long long)mtiles * a.n_tiles;
  const int nch = a.chunks_per_split;  // (s2_plan: all of them)
  static const int ksplit_max = [] { const char* e = getenv("VMM_C3_KSPLIT_MAX"); return e ? atoi(e) : 4; }();  // (measurement aid)
  int ksplit = 1;
  // (round 5, same box, environments alternating: splitting below 128 tiles instead of below 256 -- i.e. NOT splitting the 198-tile layers of the
  // sampler's batch -- conv_s2 family 0.685 -> 0.638 ms per guided step, 0.90 -> 0.79 ms per training step: the ordered ticket epilogue costs more than
  // the half-empty second workgroup slot)
  static const int s2_blocks = [] { const char* e = getenv("VMM_S2_SPLIT_BELOW"); return e ? atoi(e) : 128; }();  // (measurement aid)
  if (blocks < s2_blocks && split_tickets && n_tickets >= blocks) ksplit = (int)max(1LL, min((long long)min(nch / 4, ksplit_max), 2048 / max(blocks, 1LL)));
  if (ksplit > 1) {
    a.chunks_per_split = (int)cdiv(nch, ksplit);
    ksplit = (int)cdiv(nch, a.chunks_per_split);
    a.p.split_tickets = split_tickets;
    a.p.n_tickets = n_tickets;
    if (wide) return a.mode ? launch_s2<2, 2, 6, 1, 2, true, ONE>(a, mtiles, s, ksplit) : launch_s2<2, 2, 6, 0, 2, true, ONE>(a, mtiles, s, ksplit);
    return a.mode ? launch_s2<4, 1, 11, 1, 2, true, ONE>(a, mtiles, s, ksplit) : launch_s2<4, 1, 11, 0, 2, true, ONE>(a, mtiles, s, ksplit);
  }
  if (wide) return a.mode ? launch_s2<2, 2, 6, 1, 2, false, ONE>(a, mtiles, s) : launch_s2<2, 2, 6, 0, 2, false, ONE>(a, mtiles, s);
  return a.mode ? launch_s2<4, 1, 11, 1, 2, false, ONE>(a, mtiles, s) : launch_s2<4, 1, 11, 0, 2, false, ONE>(a, mtiles, s);
}
#if !VMM_FP16_OPERANDS
extern "C" int vmm_conv_s2_acc_bf16x3(const float* x, int32_t ldx, const float* w_frag, const float* bias, const float* res, int32_t ldres, float* out,
                                      int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                                      int32_t n_tickets, vmm_stream_t stream) {
  return s2_run<false>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, split_tickets, n_tickets, stream);
}
// the "bf16" throughput mode of the same layers (BASELINE.json configs[3]): one matrix pass on bf16-rounded operands
extern "C" int vmm_conv_s2_acc_bf16(const float* x, int32_t ldx, const float* w_frag, const float* bias, const float* res, int32_t ldres, float* out,
                                    int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                                    int32_t n_tickets, vmm_stream_t stream) {
  return s2_run<true>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, split_tickets, n_tickets, stream);
}

// The single-pass resampling layers over bf16-STORED maps: a16 = 1: input, residual and output bf16; 2: bf16 in, fp32 out (the Downsample that leaves the
// bf16 levels); 3: fp32 in, bf16 out (the Upsample that enters them).  Unsplit instances only (returns -1 if the shape would need the channel split).
template <int A16>
static int s2_run_a16(const void* x, int32_t ldx, const float* w_frag, const float* bias, const void* res, int32_t ldres, void* out, int32_t ldo, int32_t nimg,
                      int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, hipStream_t s) {
  C3Args a;
  int mtiles;
  bool wide;
  const int rc = s2_plan(static_cast<const float*>(x), ldx, w_frag, bias, static_cast<float*>(out), ldo, nimg, Hin, Win, Cin, Cout, up, a, mtiles, wide);
  if (rc) return rc;
  if (res && (ldres & 3)) return 1;
  a.p.res = static_cast<const float*>(res);
  a.p.ldres = ldres;
  if (up) return a.mode ? launch_s2<2, 2, 6, 1, 1, false, true, A16>(a, mtiles, s) : launch_s2<2, 2, 6, 0, 1, false, true, A16>(a, mtiles, s);
  if (wide) return a.mode ? launch_s2<2, 2, 6, 1, 2, false, true, A16>(a, mtiles, s) : launch_s2<2, 2, 6, 0, 2, false, true, A16>(a, mtiles, s);
  return a.mode ? launch_s2<4, 1, 11, 1, 2, false, true, A16>(a, mtiles, s) : launch_s2<4, 1, 11, 0, 2, false, true, A16>(a, mtiles, s);
}
extern "C" int vmm_conv_s2_acc_bf16_a16(const void* x, int32_t ldx, const float* w_frag, const float* bias, const void* res, int32_t ldres, void* out,
                                        int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t a16,
                                        vmm_stream_t stream) {
  hipStream_t s = (hipStream_t)stream;
  switch (a16) {
    case 1: return s2_run_a16<1>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, s);
    case 2: return s2_run_a16<2>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, s);
    case 3: return s2_run_a16<3>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, s);
    default: return -1;
  }
}
extern "C" int vmm_conv_s2_bf16x3(const float* x, int32_t ldx, const float* w_frag, const float* bias, float* out, int32_t ldo, int32_t nimg,
                                  int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, vmm_stream_t stream) {
  return vmm_conv_s2_acc_bf16x3(x, ldx, w_frag, bias, nullptr, 0, out, ldo, nimg, Hin, Win, Cin, Cout, up, nullptr, 0, stream);
}

// The same kernel on the exact-fp32 matrix-core instruction (v_mfma_f32_32x32x2_f32, 1e-6 parity): weights = vmm_pack_weights fmt 4
// (fragment order, fp32).  Used by the "fp32" arithmetic mode (training default) for the forward and the data-gradient 3x3 convolutions.
extern "C" int vmm_conv3x3_f32(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  const int rc = plan_c3(d, a, mtiles, ksplit, gn);
  if (rc != 0) return rc;
  if (a.total_rows <= 0) return 0;
  return dispatch_c3<true>(a, mtiles, ksplit, d.Cout >= 128, (hipStream_t)stream);
}
#else  // VMM_FP16_OPERANDS: this translation unit compiled with -DVMM_SINGLE_PASS=2 -- every 16-bit operand is IEEE half (vmm_common.h)

// The single-pass instances on fp16 operands (`train_precision = "fp16"`: the reference's own autocast dtype, main.py:34): same descriptors and launch
// geometry as vmm_conv3x3_bf16 / vmm_conv_s2_acc_bf16; weights = vmm_pack_weights fmt 2 | 16 (5 | 16, 6 | 16): fp16 planes.  fp32-stored maps only.
extern "C" int vmm_conv3x3_fp16(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  if (d.act_bf16) return -1;
  const int rc = plan_c3(d, a, mtiles, ksplit, gn);
  if (rc != 0) return rc;
  if (a.total_rows <= 0) return 0;
  return dispatch_c3<false, true>(a, mtiles, ksplit, d.Cout >= 128, (hipStream_t)stream);
}
extern "C" int vmm_conv_s2_acc_fp16(const float* x, int32_t ldx, const float* w_frag, const float* bias, const float* res, int32_t ldres, float* out,
                                    int32_t ldo, int32_t nimg, int32_t Hin, int32_t Win, int32_t Cin, int32_t Cout, int32_t up, int32_t* split_tickets,
                                    int32_t n_tickets, vmm_stream_t stream) {
  return s2_run<true>(x, ldx, w_frag, bias, res, ldres, out, ldo, nimg, Hin, Win, Cin, Cout, up, split_tickets, n_tickets, stream);
}
#endif
