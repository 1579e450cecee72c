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
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CK = 32;    // channels per chunk
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
};

__device__ __forceinline__ void split2c(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32>
__global__ __launch_bounds__(256, 2) void conv3x3_x3_kernel(const C3Args a) {
  constexpr int BM = WM * 64;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ph = smem;
  const vmm_conv_desc& p = a.p;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const int lrow = lane & 31, lk = lane >> 5;
  const int W = p.Win, H = p.Hin, HW = H * W;
  const int mtile = blockIdx.x / a.n_tiles;
  const int n0 = (blockIdx.x % a.n_tiles) * (WN * 64);
  const int Cin = p.C1 + p.C2;
  const int nchunks = Cin / CK;
  const int c_begin = blockIdx.y * a.chunks_per_split;
  const int c_end = min(nchunks, c_begin + a.chunks_per_split);
  if (c_begin >= c_end) return;

  int img = 0, ty0 = 0, tx0 = 0, g0 = 0;
  if (MODE) {
    img = mtile / a.tiles_per_frame;
    const int t = mtile - img * a.tiles_per_frame;
    const int tyi = t / a.tiles_x;
    ty0 = tyi * (BM / 16);
    tx0 = (t - tyi * a.tiles_x) * 16;
  } else {
    g0 = mtile * BM;
  }

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
  auto src_row = [&](int ps) -> int {  // recomputed per chunk rather than kept in MAXP registers
    const int r = (tid >> 3) + ps * 32;
    int s = -1;
    if (r < a.PR) {
      if (MODE) {
        const int py = r / 18, px = r - py * 18;
        const int h = ty0 - 1 + py, w = tx0 - 1 + px;
        if (h >= 0 && h < H && w >= 0 && w < W) s = img * HW + h * W + w;
      } else {
        const int g = g0 - a.halo + r;
        if (g >= 0 && g < a.total_rows) s = g;
      }
    }
    return s;
  };
  f32x4 preg[MAXP];
  // GroupNorm * FiLM coefficients of this thread's four channels for the chunk in flight (2-D tiles: one sample per tile).  Requested with
  // the chunk's patch loads, not when the patch is stored: a dependent L2 round trip would otherwise sit on the critical path.
  f32x4 cfa = {1.f, 0.f, 1.f, 0.f}, cfb = cfa;
  auto load_coef = [&](int cc) {
    const int c0 = cc * CK;
    if (MODE && p.a_mode == 1 && c0 < p.C1) {
      const float* cf = p.a_coef + ((long long)(img / p.a_imgs_per_sample) * p.C1 + c0 + k4 * 4) * 2;
      cfa = *reinterpret_cast<const f32x4*>(cf);
      cfb = *reinterpret_cast<const f32x4*>(cf + 4);
    }
  };
  auto load_patch = [&](int cc) {  // raw loads only, so that they stay in flight under the MFMAs; the operand transform runs at store time
    load_coef(cc);
    const int c0 = cc * CK;
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = (src1 ? c0 : c0 - p.C1) + k4 * 4;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int sr = src_row(ps);
      if (sr >= 0) v = *reinterpret_cast<const f32x4*>(src + (long long)sr * ld + cb);
      preg[ps] = v;
    }
  };
  // One item of the next chunk's patch.  vmcnt retires in order, so an HBM-latency load blocks every younger weight-fragment wait
  // until it lands: the prefetch is issued one item per k16 step, AFTER that step's weight loads, which leaves it PFB + 1 steps to
  // arrive before anything waits on it (all MAXP items at the chunk top parked the waves for ~40 % of their cycles).
  auto load_patch_item = [&](int cc, int ps) {
    const int c0 = cc * CK;
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = (src1 ? c0 : c0 - p.C1) + k4 * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    const int sr = src_row(ps);
    if (sr >= 0) v = *reinterpret_cast<const f32x4*>(src + (long long)sr * ld + cb);
    preg[ps] = v;
  };
  auto store_patch = [&](int cc) {
    const int c0 = cc * CK;
    const bool xform = c0 < p.C1 && p.a_mode == 1;
    const int rows_per_sample = HW * p.a_imgs_per_sample;
    // GroupNorm * FiLM coefficients of this thread's four channels: one sample per 2-D tile -> fetched once per chunk, not per item
    f32x4 ca = cfa, cb4 = cfb;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int r = (tid >> 3) + ps * 32;
      if (r < a.PR) {
        f32x4 v = preg[ps];
        const int sr = xform ? src_row(ps) : -1;
        if (sr >= 0) {  // zero padding is applied AFTER the activation (vddp.py:268-285), so padded items stay 0
          if (!MODE) {  // flat row tiles run across samples
            const float* cf = p.a_coef + ((long long)(sr / rows_per_sample) * p.C1 + c0 + k4 * 4) * 2;
            ca = *reinterpret_cast<const f32x4*>(cf);
            cb4 = *reinterpret_cast<const f32x4*>(cf + 4);
          }
          v.x = igemm::silu_fast(v.x * ca.x + ca.y);
          v.y = igemm::silu_fast(v.y * ca.z + ca.w);
          v.z = igemm::silu_fast(v.z * cb4.x + cb4.y);
          v.w = igemm::silu_fast(v.w * cb4.z + cb4.w);
        }
        if constexpr (F32) {  // exact-fp32 variant: the patch row is 32 floats (the same 128 + 16 bytes as bf16 hi | lo)
          *reinterpret_cast<f32x4*>(&Ph[r * CROW + k4 * 8]) = v;
        } else {
          unsigned h0, l0, h1, l1;
          split2c(v.x, v.y, h0, l0);
          split2c(v.z, v.w, h1, l1);
          *reinterpret_cast<uint2*>(&Ph[r * CROW + k4 * 4]) = make_uint2(h0, h1);
          *reinterpret_cast<uint2*>(&Ph[r * CROW + CK + k4 * 4]) = make_uint2(l0, l1);
        }
      }
    }
  };

  // weight fragments: plane (column tile nt, k16 step ks, hi|lo) = 64 lanes x 16 bytes contiguous
  const uint4* wf = reinterpret_cast<const uint4*>(p.w);
  const uint4* bbase[2];
#pragma unroll
  for (int j = 0; j < 2; ++j) bbase[j] = wf + (long long)((n0 + wn * 64) / 32 + j) * a.KS * 128 + lane;
  const int cin16 = Cin / 16;
  auto load_b = [&](uint4 (&d)[4], int ks) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const uint4* q = bbase[j] + (long long)ks * 128;
      d[2 * j] = q[0];
      d[2 * j + 1] = q[64];
    }
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
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
  auto load_a = [&](uint4 (&d)[4], int tap, int s) {
    const int kh = tap / 3, kw = tap % 3;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      // bf16x3: 8 hi then 8 lo values of channels s*16 + lk*8 .. +7;  fp32: those eight channels as two float4
      const unsigned short* q = Ph + abase[i][kh] + (kw * CROW + s * (F32 ? 32 : 16));
      uint4 vh = *reinterpret_cast<const uint4*>(q);
      uint4 vl = *reinterpret_cast<const uint4*>(q + (F32 ? 8 : CK));
      if (!MODE && !((tapmask[i] >> tap) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
      d[2 * i] = vh;
      d[2 * i + 1] = vl;
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
          for (int j = 0; j < 2; ++j) {
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
      if constexpr (SPLIT) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(pix, wgt, c, 0, 0, 0);
      else return __builtin_amdgcn_mfma_f32_32x32x16_bf16(wgt, pix, c, 0, 0, 0);
    };
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

  // software pipeline over the 18 k16 steps q = 2 tap + half of a chunk: the weight fragments of step q + PFB and the patch fragments
  // of step q + 1 are requested before the MFMAs of step q (PFB = 2 covers an L2 hit even when the sibling wave does not leave the
  // line in L1).  sched_barrier pins that order (the scheduler otherwise sinks the loads next to their uses to save registers).
  constexpr int NB = PFB + 1;
  uint4 bb[NB][4], aa[2][4];
  auto ks_of = [&](int cc, int q) { return (q >> 1) * cin16 + cc * 2 + (q & 1); };
  load_patch(c_begin);
#pragma unroll
  for (int q = 0; q < PFB; ++q) load_b(bb[q], ks_of(c_begin, q));
  store_patch(c_begin);
  __syncthreads();
  load_a(aa[0], 0, 0);
  for (int cc = c_begin; cc < c_end; ++cc) {
    const bool more = cc + 1 < c_end;
    // keep the six patch bases opaque per chunk: the 18 per-step addresses then stay base + immediate (ds_read offset field)
    // instead of being hoisted out of the loop into 36 address registers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) asm volatile("" : "+v"(abase[i][kh]));
#pragma unroll
    for (int q = 0; q < 18; ++q) {
      if (q + PFB < 18) load_b(bb[(q + PFB) % NB], ks_of(cc, q + PFB));
      else if (more) load_b(bb[(q + PFB) % NB], ks_of(cc + 1, q + PFB - 18));
      if (q == 0 && more) load_coef(cc + 1);
      if (q < MAXP && more) load_patch_item(cc + 1, q);
      if (q + 1 < 18) load_a(aa[(q + 1) & 1], (q + 1) >> 1, (q + 1) & 1);
      __builtin_amdgcn_sched_barrier(0);
      mma_step(aa[q & 1], bb[q % NB]);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more) {
      __syncthreads();  // every wave is done reading the patch of chunk cc
      store_patch(cc + 1);
      __syncthreads();
      load_a(aa[0], 0, 0);
    }
  }

  // epilogue.  Split channel reduction (gridDim.y > 1): the splits of one output tile add their partial sums in split order, so the
  // result does not depend on scheduling.  Split 0 writes through to memory; split y waits for ticket == y, adds with device-scope
  // fp32 atomics (performed at the memory side, coherent across the XCDs' L2s without cache write-backs), waits for their
  // completion and passes the ticket on (the last one resets it to 0).  Workgroups are dispatched y-major, i.e. split y - 1 is
  // always resident before split y.
  if constexpr (SPLIT) {
    // acc[i][j]: rows = pixels (r & 3) + 8 (r >> 2) + 4 lk of row tile i, column = output channel j*32 + lrow
    const bool first = blockIdx.y == 0;
    int* ticket = p.split_tickets + blockIdx.x;
    if (!first) {
      if (tid == 0)
        while (__hip_atomic_load(ticket, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (int)blockIdx.y) __builtin_amdgcn_s_sleep(8);
      __syncthreads();
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
        for (int j = 0; j < 2; ++j) {
          const int col = n0 + wn * 64 + j * 32 + lrow;
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
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt: this wave's stores / atomics have been performed
    __syncthreads();
    if (tid == 0) __hip_atomic_store(ticket, blockIdx.y + 1 == gridDim.y ? 0 : (int)blockIdx.y + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  } else {
    // acc[i][j]: rows = output channels (r & 3) + 8 (r >> 2) + 4 lk of column tile j, column = pixel i*32 + lrow of this wave.
    // The eight bias pieces of the lane are requested together and folded into the accumulators after ONE wait (a load -> wait ->
    // add -> store chain per 16-byte piece serialised sixteen L2 latencies per tile); the residual pieces likewise per pixel.
    {
      f32x4 bv[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          bv[j][g] = p.bias ? *reinterpret_cast<const f32x4*>(p.bias + n0 + wn * 64 + j * 32 + 8 * g + 4 * lk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
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
      } else {
        if (g0 + m >= a.total_rows) continue;
        orow = g0 + m;
      }
      const int c0 = n0 + wn * 64 + 4 * lk;
      f32x4 rv[2][4];
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g)
          rv[j][g] = p.res ? *reinterpret_cast<const f32x4*>(p.res + orow * p.ldres + c0 + j * 32 + 8 * g) : f32x4{0.f, 0.f, 0.f, 0.f};
      float* orp = p.out + orow * p.ldo + c0;
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          const f32x4 v = {acc[i][j][4 * g] + rv[j][g].x, acc[i][j][4 * g + 1] + rv[j][g].y, acc[i][j][4 * g + 2] + rv[j][g].z,
                           acc[i][j][4 * g + 3] + rv[j][g].w};
          *reinterpret_cast<f32x4*>(orp + j * 32 + 8 * g) = v;
        }
    }
    if (MODE && p.gn_part) {
      // GroupNorm statistics of the output (vddp.py:274-279) while it is still in registers: per run of 8 consecutive output channels
      // the sum and the sum of squares over this wave's 64 pixels, combined over the workgroup's waves in LDS, then one fp64
      // atomic per (run, moment) into sums[sample][group] -- the separate statistics pass over the output disappears.
      // 16 values per lane: (sum, sum of squares) x 8 runs; wave totals by a reduce-scatter butterfly -- every exchange halves the
      // values a lane still carries (8 + 4 + 2 + 1 shuffles), two plain steps finish: 17 shuffles instead of 96
      float gv[16];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
          float s1 = 0.f, s2 = 0.f;  // (the bias is already in the accumulators)
#pragma unroll
          for (int i = 0; i < 2; ++i) {
            const float v0 = acc[i][j][4 * g], v1 = acc[i][j][4 * g + 1], v2 = acc[i][j][4 * g + 2], v3 = acc[i][j][4 * g + 3];
            s1 += (v0 + v1) + (v2 + v3);
            s2 += (v0 * v0 + v1 * v1) + (v2 * v2 + v3 * v3);
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
          gv[k] = keep + __shfl_xor(send, 1 << bit, 64);
        }
      }
      gv[0] += __shfl_xor(gv[0], 2, 64);
      gv[0] += __shfl_xor(gv[0], 1, 64);  // lane L now holds the wave total of value (L >> 2) & 15
      __syncthreads();  // every wave is done with the patch: reuse its LDS
      float* sc = reinterpret_cast<float*>(smem);
      if ((lane & 3) == 0) sc[wave * 16 + (lane >> 2)] = gv[0];
      __syncthreads();
      if (tid < WN * 16) {
        const int wn2 = tid >> 4, slot = tid & 15, run = slot >> 1;
        float v = 0.f;
#pragma unroll
        for (int w2 = 0; w2 < WM; ++w2) v += sc[(w2 * WN + wn2) * 16 + slot];
        // slot of this (frame, tile, 8-channel run) among the contributions to its (sample, group): written exactly once, no atomics
        const int cout0 = n0 + wn2 * 64 + (run >> 2) * 32 + (run & 3) * 8;
        const int cpg = p.Cout / p.gn_groups, rpg = cpg >> 3;
        const int grp = cout0 / cpg, rig = (cout0 - grp * cpg) >> 3;
        const int smp = img / p.a_imgs_per_sample, fr = img - smp * p.a_imgs_per_sample;
        const int n_contrib = p.a_imgs_per_sample * a.tiles_per_frame * rpg;
        const int k = (fr * a.tiles_per_frame + (mtile - img * a.tiles_per_frame)) * rpg + rig;
        p.gn_part[(((long long)smp * p.gn_groups + grp) * n_contrib + k) * 2 + (slot & 1)] = v;
      }
    }
  }
}

template <int WM, int WN, int MAXP, int MODE, int PFB, bool SPLIT, bool F32>
int launch_c3(const C3Args& a, int mtiles, int ksplit, hipStream_t s) {
  const size_t shm = sizeof(unsigned short) * (size_t)a.PR * CROW;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, SPLIT, F32>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_x3_kernel<WM, WN, MAXP, MODE, PFB, SPLIT, F32>), dim3((unsigned)(mtiles * a.n_tiles), ksplit), dim3(256), shm, s, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// shape planning shared by the launcher and the query below; returns 0 and fills a / mtiles / ksplit / gn, or the launcher's status
int plan_c3(const vmm_conv_desc& d, C3Args& a, int& mtiles, int& ksplit, bool& gn) {
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
  a.n_tiles = wide ? d.Cout / 128 : 1;
  a.total_rows = (int)M;
  a.KS = 9 * (d.C1 + d.C2) / 16;
  if (d.Win >= 32 && d.Win % 16 == 0 && d.Hin % TH == 0) {
    a.mode = 1;
    a.tiles_x = d.Win / 16;
    a.tiles_per_frame = a.tiles_x * (d.Hin / TH);
    a.pitch = 18;
    a.halo = 0;
    a.PR = (TH + 2) * 18;
    mtiles = d.nimg * a.tiles_per_frame;
  } else {
    a.mode = 0;
    a.tiles_x = a.tiles_per_frame = 1;
    a.pitch = d.Win;
    a.halo = d.Win + 1;
    a.PR = BM + 2 * a.halo;
    mtiles = (int)cdiv(M, BM);
  }
  if (a.PR > (wide ? 6 : 11) * 32) return 1;
  // few-row layers (12 x 12 level): split the channel chunks so that both workgroup slots of every CU are filled a few times over;
  // the partial sums are added in a fixed order (tickets, see the kernel epilogue)
  const int nch = (d.C1 + d.C2) / CK;
  const long long blocks = (long long)mtiles * a.n_tiles;
  ksplit = 1;
  // split the channel reduction only when the tiles cannot even half-fill the chip: measured on the Lagrangian sampler (batch 8), splitting
  // below 1024 workgroups cost 1.15 ms per step against splitting below 128 (ordered atomics epilogue, no fused GroupNorm sums)
  if (blocks < 128 && d.split_tickets && d.n_tickets >= blocks) ksplit = (int)max(1LL, min((long long)min(nch, 8), 2048 / max(blocks, 1LL)));
  a.chunks_per_split = (int)cdiv(nch, ksplit);
  ksplit = (int)cdiv(nch, a.chunks_per_split);
  // GroupNorm statistics of the output in the epilogue: unsplit 2-D tiles (one frame, hence one sample, per workgroup), groups of whole
  // 8-channel runs, no residual in the output
  gn = d.gn_part && d.gn_groups > 0 && a.mode == 1 && ksplit == 1 && !d.res && d.a_imgs_per_sample > 0 && d.Cout % d.gn_groups == 0 &&
       (d.Cout / d.gn_groups) % 8 == 0 && d.nimg % d.a_imgs_per_sample == 0;
  if (!gn) a.p.gn_part = nullptr;
  return 0;
}

}  // namespace

template <bool F32>
int dispatch_c3(const C3Args& a, int mtiles, int ksplit, bool wide, hipStream_t s) {
  if (ksplit > 1) {
    if (wide) return a.mode ? launch_c3<2, 2, 6, 1, 2, true, F32>(a, mtiles, ksplit, s) : launch_c3<2, 2, 6, 0, 2, true, F32>(a, mtiles, ksplit, s);
    return a.mode ? launch_c3<4, 1, 11, 1, 1, true, F32>(a, mtiles, ksplit, s) : launch_c3<4, 1, 11, 0, 1, true, F32>(a, mtiles, ksplit, s);
  }
  if (wide) return a.mode ? launch_c3<2, 2, 6, 1, 2, false, F32>(a, mtiles, 1, s) : launch_c3<2, 2, 6, 0, 2, false, F32>(a, mtiles, 1, s);
  return a.mode ? launch_c3<4, 1, 11, 1, 1, false, F32>(a, mtiles, 1, s) : launch_c3<4, 1, 11, 0, 1, false, F32>(a, mtiles, 1, s);
}

// Number of GroupNorm partial-sum pairs per (sample, group) vmm_conv3x3_bf16x3(d) will leave in d->gn_part (the caller then skips
// vmm_groupnorm_stats and hands them to vmm_groupnorm_coef), 0 when it will not.  Pure host logic.
extern "C" int vmm_conv3x3_fuses_gn(const vmm_conv_desc* dp) {
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  if (plan_c3(*dp, a, mtiles, ksplit, gn) != 0 || !gn) return 0;
  return dp->a_imgs_per_sample * a.tiles_per_frame * ((dp->Cout / dp->gn_groups) >> 3);
}

// Weights: vmm_pack_weights fmt 2 (MFMA fragment order).  Returns 1 (nothing launched) when the descriptor is outside this
// kernel's envelope: 3x3 / stride 1 / pad 1, C1 % 32 == C2 % 32 == 0, Cout == 64 or Cout % 128 == 0, W <= 31 or (W % 16 == 0 and
// H % 16 == 0) -- the caller then uses vmm_conv_igemm_bf16x3 with fmt-1 weights.
extern "C" int vmm_conv3x3_bf16x3(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  C3Args a;
  int mtiles, ksplit;
  bool gn = false;
  const int rc = plan_c3(d, a, mtiles, ksplit, gn);
  if (rc != 0) return rc;
  if (a.total_rows <= 0) return 0;
  const bool wide = d.Cout >= 128;
  hipStream_t s = (hipStream_t)stream;
  return dispatch_c3<false>(a, mtiles, ksplit, wide, s);
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
