// 3x3 stride-1 "same" convolution as Winograd F(2x2, 3x3) on the split-bf16 matrix cores, gfx950.
//
// The ResnetBlock projections (vddp.py:268-285, Conv3d (1,3,3) pad (0,1,1)) are 60 % of the denoiser's flops.  The direct kernel
// (conv3x3_bf16x3.hip) spends 9 taps x 3 split passes = 27 MFMA passes per (pixel, cin, cout); here a 2 x 2 output tile is computed from
// the 4 x 4 input tile around it with 16 products per (cin, cout) instead of 36:
//
//   Y = A^T [ sum_cin (G g G^T) .* (B^T d B) ] A          (Lavin & Gray; B^T, A^T have entries 0 / +-1, G has 0 / +-1/2 / 1)
//
// i.e. sixteen independent GEMMs  M_pos[tile][cout] = sum_cin V_pos[tile][cin] U_pos[cin][cout]  (pos = the 4 x 4 transform-domain
// position), 2.25 x fewer matrix passes.  The transforms are exact-constant fp32 additions; with the operands split AFTER the
// transform (hi + lo bf16, three passes, fp32 accumulate) the result stays fp32-class: 7e-6 relative against 4.7e-6 for the direct
// split-bf16 kernel on the same data (tests/test_gpu_kernels.py compares both with F.conv2d at 5e-5).
//
// One workgroup = 8 waves = (up to) 64 tiles (TBH x TBW tiles = 2 TBH x 2 TBW output pixels of one frame) x 64 output channels x all 16
// positions, walking the input channels 16 at a time:
//   patch   the (2 TBH + 2) x (2 TBW + 2) input pixels x 16 channels as fp32 in LDS (two half patches of 8 channels), requested a step ahead
//           into registers; the fused operand transform (GroupNorm * FiLM -> SiLU, a_mode 1) and the zero padding are applied on the way in,
//           once per element;
//   V       thread (tile, channel pair) reads its 4 x 4 x 2 window, forms B^T d B with packed fp32 adds, splits the sixteen pairs and
//           writes them as dwords into the MFMA "B" fragment image  V[pos][hi|lo][k octet][tile][8 bf16]  (double-buffered, 64 KB each).
//           Waves 0-3 do this for channels 0..7 of a chunk while waves 4-7 run the previous chunk's matrix instructions, then they swap:
//           the VALU / LDS work of one group hides under the MFMAs of the other (one wave of each group per SIMD);
//   U       = G g G^T, pre-split and in "A" fragment order (vmm_pack_weights fmt 8), straight from L2 into registers: wave w owns the
//           positions 2 w, 2 w + 1, whose fragments of a step are 8 KB contiguous;
//   MFMA    wave w: M_pos[64 cout][64 tiles] for its two positions = 8 accumulators of 32 x 32, 24 MFMAs per step, four V fragment
//           reads per twelve MFMAs;
//   A^T.A   the positions live in different waves, so the output transform goes through LDS once per workgroup (two rounds of 32
//           output channels over the V buffers: every wave leaves its two nu-partial sums, a thread = (tile, four channels) gathers
//           the eight waves' pieces, adds bias / residual, stores 16-byte pieces and -- for the GroupNorm that follows every
//           ResnetBlock convolution (vddp.py:274-279) -- leaves the per-workgroup partial sums of its outputs).
// Per 16-channel step and workgroup: 48 MFMAs per SIMD (1536 matrix cycles) against ~900 VALU cycles for the transform, where the direct
// kernel needs 108 MFMAs for the same 256 pixels x 64 channels.
#include <stdio.h>
#include <stdlib.h>
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int NTHR = 512;
constexpr int V_OCT = 64 * 16, V_PLANE = 2 * V_OCT, V_POS = 2 * V_PLANE, V_BUF = 16 * V_POS;  // 1024, 2048, 4096, 65536 bytes
constexpr int HPITCH = 48;        // bytes per pixel of a half patch: 8 channels fp32 + 16 (3 x 16: eight tiles two pixels apart start 24 banks apart -- conflict-free ds_read_b64)
constexpr int PATCH_PIX = 324;    // 18 x 18
constexpr int PATCH_OFF = 2 * V_BUF;
constexpr int HALF_BYTES = PATCH_PIX * HPITCH;             // 15 552
constexpr int LDS_BYTES = PATCH_OFF + 2 * HALF_BYTES;      // 162 176 of 163 840
constexpr int PITEMS = 2;         // half-patch items (pixel, 4 channels) per thread: 2 x 512 >= 324 x 2

// V stores: ds_write_addtid_b32 (address = M0 + offset + 4 lane, no address register: 128 B/clk/CU, twice ds_write_b32) -- a wave's sixty-four
// (tile, channel pair) dwords of one (position, plane, k octet) are 256 contiguous bytes of the fragment image
#define VMM_WINO_ST8(base, v, o)                                                                                                                   \
  asm volatile("s_mov_b32 %[t], m0\n\ts_mov_b32 m0, %[b]\n\ts_nop 0\n\t"                                                                          \
               "ds_write_addtid_b32 %[v0] offset:%[o0]\n\tds_write_addtid_b32 %[v1] offset:%[o1]\n\t"                                               \
               "ds_write_addtid_b32 %[v2] offset:%[o2]\n\tds_write_addtid_b32 %[v3] offset:%[o3]\n\t"                                               \
               "ds_write_addtid_b32 %[v4] offset:%[o4]\n\tds_write_addtid_b32 %[v5] offset:%[o5]\n\t"                                               \
               "ds_write_addtid_b32 %[v6] offset:%[o6]\n\tds_write_addtid_b32 %[v7] offset:%[o7]\n\ts_mov_b32 m0, %[t]"                             \
               : [t] "=&s"(m0_keep)                                                                                                                \
               : [b] "s"(base), [v0] "v"(v[0]), [v1] "v"(v[1]), [v2] "v"(v[2]), [v3] "v"(v[3]), [v4] "v"(v[4]), [v5] "v"(v[5]), [v6] "v"(v[6]),   \
                 [v7] "v"(v[7]), [o0] "n"((o)), [o1] "n"((o) + V_PLANE), [o2] "n"((o) + V_POS), [o3] "n"((o) + V_POS + V_PLANE),                   \
                 [o4] "n"((o) + 2 * V_POS), [o5] "n"((o) + 2 * V_POS + V_PLANE), [o6] "n"((o) + 3 * V_POS), [o7] "n"((o) + 3 * V_POS + V_PLANE)    \
               : "memory")

struct WArgs {
  vmm_conv_desc p;
  int TBH, TBW, PH, PW;   // tile block (tiles), patch (pixels)
  int ntile;              // TBH * TBW <= 64
  int nbx, bpf;           // tile blocks across a frame, per frame
  int KS, ncb;            // 16-channel steps, 64-column output blocks
  int gn_run, gn_rpg;     // GroupNorm sums: channels per partial sum (min(channels per group, 32)), partial sums per group and workgroup row
  int dbg = 0;            // measurement aid (VMM_WINO_DBG): 1 no transform, 2 no matrix instructions, 4 no V stores, 8 no patch stores, 16 no U loads, 32 no patch loads
  unsigned long long* trace = nullptr;  // VMM_WINO_TRACE=<launch>: wave 0 of every workgroup stamps s_memtime at its phase boundaries (32 slots per workgroup)
};

__global__ __launch_bounds__(NTHR) void conv3x3_wino_kernel(const WArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const vmm_conv_desc& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int H = p.Hin, W = p.Win;
  auto stamp = [&](int k) {  // measurement aid
    if (a.trace && tid == 0 && k < 32) a.trace[(size_t)blockIdx.x * 32 + k] = __builtin_readcyclecounter();
  };
  stamp(0);

  // ---------------------------------------------------------------- which block
  const int id = blockIdx.x;
  const int cb = id % a.ncb, blk_all = id / a.ncb;
  const int img = blk_all / a.bpf, blk = blk_all - img * a.bpf;
  const int by = blk / a.nbx, bx = blk - by * a.nbx;
  const int Y0 = by * 2 * a.TBH, X0 = bx * 2 * a.TBW;   // first output pixel of the block
  // p.a_img_mod: frames of the batch's second half read the first half's rows of a1 (one pre-norm tensor shared by both guidance branches)
  const int simg = (p.a_img_mod > 0 && img >= p.a_img_mod) ? img - p.a_img_mod : img;
  const int smp = p.a_imgs_per_sample > 0 ? img / p.a_imgs_per_sample : 0;

  // ---------------------------------------------------------------- roles
  // waves 0-3 ("A") turn channels 0..7 of every 16-channel chunk into k octet 0 of V, waves 4-7 ("B") channels 8..15 into octet 1; while one
  // group transforms, the other runs its matrix instructions (waves w and w + 4 share a SIMD), and they swap every half step
  const bool roleA = wave < 4;
  const int oct = wave >> 2;

  // ---------------------------------------------------------------- half-patch loader: item k = (patch pixel, channel quad q2 of the half)
  // q2 is wave-uniform (odd / even waves), so the fused transform's coefficients of a wave's four channels are eight SGPRs (s_load), not registers
  const int q2 = wave & 1;
  int prow[PITEMS], pdst[PITEMS];
  unsigned pmask = 0;  // bit k: inside the image, bit 8 + k: the item exists
  const int npix = a.PH * a.PW;
#pragma unroll
  for (int k = 0; k < PITEMS; ++k) {
    const int ppix = (wave >> 1) * 64 + lane + 256 * k;
    const int py = ppix / a.PW, px = ppix - py * a.PW;
    const int y = Y0 - 1 + py, x = X0 - 1 + px;
    const bool exists = ppix < npix;
    const bool inside = exists && y >= 0 && y < H && x >= 0 && x < W;
    prow[k] = inside ? (simg * H + y) * W + x : 0;
    pdst[k] = PATCH_OFF + ppix * HPITCH + q2 * 16;
    pmask |= (inside ? 1u : 0u) << k | (exists ? 1u : 0u) << (8 + k);
  }
  const int row2_delta = (img - simg) * H * W;  // the second source is always the frame's own
  typedef float f32x8 __attribute__((ext_vector_type(8)));
  struct Half {
    f32x4 pv[PITEMS];
    f32x8 cf;  // (a, b) x the wave's four channels: SGPRs
    bool xform;
  } hA, hB;
  auto gload = [&](Half& h, int c, int half) {  // chunk c = channels 16 c .. + 15; half 0 / 1 = its channels 0..7 / 8..15
    if ((a.dbg & 32) && c > 1) return;
    const int c0 = c * 16 + half * 8 + q2 * 4;
    const bool src1 = c0 < p.C1;
    const float* base = src1 ? p.a1 + c0 : p.a2 + (c0 - p.C1);
    const int ld = src1 ? p.lda1 : p.lda2, dr = src1 ? 0 : row2_delta;
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) h.pv[k] = *reinterpret_cast<const f32x4*>(base + (long long)((prow[k] + dr) * ld));
    h.xform = p.a_mode == 1 && src1;
    if (h.xform) {
      const float* cf = p.a_coef + ((long long)smp * p.C1 + c0) * 2;
      asm volatile("s_load_dwordx8 %0, %1, 0x0" : "=s"(h.cf) : "s"(cf));
    }
  };
  auto pstore = [&](Half& h, int half) {
    if (a.dbg & 8) return;
    if (h.xform) asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(h.cf));  // (the scalar load above is invisible to the compiler's wait-count bookkeeping)
#pragma unroll
    for (int k = 0; k < PITEMS; ++k) {
      f32x4 v = h.pv[k];
      if (h.xform) {  // (workgroup-uniform)
        v.x = igemm::silu_fast(v.x * h.cf[0] + h.cf[1]);
        v.y = igemm::silu_fast(v.y * h.cf[2] + h.cf[3]);
        v.z = igemm::silu_fast(v.z * h.cf[4] + h.cf[5]);
        v.w = igemm::silu_fast(v.w * h.cf[6] + h.cf[7]);
      }
      // zero padding is applied AFTER the activation (vddp.py:268-285)
      if (!((pmask >> k) & 1u)) v = f32x4{0.f, 0.f, 0.f, 0.f};
      if ((pmask >> (8 + k)) & 1u) *reinterpret_cast<f32x4*>(sm + pdst[k] + half * HALF_BYTES) = v;
    }
  };

  // ---------------------------------------------------------------- input transform: thread = (tile slot, channel pair cp4 of the group's octet)
  const int slot = (wave & 3) * 16 + (lane >> 2), cp4 = lane & 3;
  const int rslot = min(slot, a.ntile - 1);  // (slots past the block read a real tile's window; their columns of M are never stored)
  const int tty = rslot / a.TBW, ttx = rslot - tty * a.TBW;
  const int trd = PATCH_OFF + oct * HALF_BYTES + ((2 * tty) * a.PW + 2 * ttx) * HPITCH + cp4 * 8;  // + (i PW + j) HPITCH
  const int prs = a.PW * HPITCH;
  const int twr0 = __builtin_amdgcn_readfirstlane(oct * V_OCT + (wave & 3) * 256);              // + buf V_BUF (+ 4 lane by the instruction)
  // (plain v_add / v_sub through asm: hipcc turns <2 x float> arithmetic -- and adjacent scalar adds, by SLP -- into v_pk_add_f32, which runs on the matrix
  // datapath and stalls against the other wave's MFMAs: the transform took 2.1-2.5 k cycles beside them with packed adds)
  auto fadd = [](float x, float y) { float r; asm("v_add_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto fsub = [](float x, float y) { float r; asm("v_sub_f32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(y)); return r; };
  auto transform = [&](int buf) {
    if (a.dbg & 1) return;
    f32x2 d[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j) d[i][j] = *reinterpret_cast<const f32x2*>(sm + trd + i * prs + j * HPITCH);
    // B^T d: rows (d0 - d2, d1 + d2, d2 - d1, d1 - d3)
    float t[4][4][2];
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        t[0][j][c] = fsub(d[0][j][c], d[2][j][c]);
        t[1][j][c] = fadd(d[1][j][c], d[2][j][c]);
        t[2][j][c] = fsub(d[2][j][c], d[1][j][c]);
        t[3][j][c] = fsub(d[1][j][c], d[3][j][c]);
      }
    const unsigned base = (unsigned)(buf * V_BUF + twr0);
    unsigned m0_keep;
#pragma unroll
    for (int xi = 0; xi < 4; ++xi) {
      // one wave per SIMD transforms at a time: the four splits of a row are written stage by stage, so that every instruction has three
      // independent neighbours (as four serial cvt -> shift -> sub -> cvt chains the transform ran at ~10 cycles per instruction)
      float v[4][2];
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        v[0][c] = fsub(t[xi][0][c], t[xi][2][c]);
        v[1][c] = fadd(t[xi][1][c], t[xi][2][c]);
        v[2][c] = fsub(t[xi][2][c], t[xi][1][c]);
        v[3][c] = fsub(t[xi][1][c], t[xi][3][c]);
      }
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      unsigned w[8];  // (hi, lo) x nu
      float r[4][2];
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) w[2 * nu] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{v[nu][0], v[nu][1]}), bf16x2_t));
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) r[nu][0] = fsub(v[nu][0], __uint_as_float(w[2 * nu] << 16));
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) r[nu][1] = fsub(v[nu][1], __uint_as_float(w[2 * nu] & 0xFFFF0000u));
#pragma unroll
      for (int nu = 0; nu < 4; ++nu) w[2 * nu + 1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2{r[nu][0], r[nu][1]}), bf16x2_t));
      if (a.dbg & 4) {
        if (w[0] + w[1] + w[2] + w[3] + w[4] + w[5] + w[6] + w[7] == 0x12345u) sm[0] = 1;
        continue;
      }
      switch (xi) {  // (the offsets are instruction immediates)
        case 0: VMM_WINO_ST8(base, w, 0); break;
        case 1: VMM_WINO_ST8(base, w, 4 * V_POS); break;
        case 2: VMM_WINO_ST8(base, w, 8 * V_POS); break;
        default: VMM_WINO_ST8(base, w, 12 * V_POS); break;
      }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // (the stores above are invisible to the compiler's wait-count bookkeeping)
  };

  // ---------------------------------------------------------------- matrix part: wave w = positions 2 w, 2 w + 1
  f32x16 acc[2][2][2];  // [position][cout fragment][tile fragment]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int k = 0; k < 2; ++k)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][k][r] = 0.f;
  const uint4* ubase = reinterpret_cast<const uint4*>(p.w) + ((long long)cb * a.KS * 16 + 2 * wave) * 4 * 64 + lane;  // + s * 16 * 256 + (pos * 4 + mf * 2 + plane) * 64
  uint4 uf[2][2][2];  // [position][cout fragment][hi | lo]
  auto uload = [&](int s, int pos) {
    if (a.dbg & 16) return;
    const uint4* q = ubase + (long long)s * (16 * 4 * 64) + pos * 4 * 64;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int pl = 0; pl < 2; ++pl) uf[pos][mf][pl] = q[(mf * 2 + pl) * 64];
  };
  const int vrd = (lane >> 5) * V_OCT + (lane & 31) * 16;  // + buf V_BUF + pos V_POS + plane V_PLANE + nf 512
  auto mma = [&](int buf, int pos) {
    if (a.dbg & 2) return;
    const unsigned char* vb = sm + buf * V_BUF + (2 * wave + pos) * V_POS + vrd;
    bf16x8 Bh[2], Bl[2], Ah[2], Al[2];
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      Bh[nf] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(vb + nf * 512));
      Bl[nf] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(vb + V_PLANE + nf * 512));
    }
#pragma unroll
    for (int mf = 0; mf < 2; ++mf) {
      Ah[mf] = __builtin_bit_cast(bf16x8, uf[pos][mf][0]);
      Al[mf] = __builtin_bit_cast(bf16x8, uf[pos][mf][1]);
    }
    // pass-major, lo products first
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[pos][mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[mf], Bl[nf], acc[pos][mf][nf], 0, 0, 0);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[pos][mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Al[mf], Bh[nf], acc[pos][mf][nf], 0, 0, 0);
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int nf = 0; nf < 2; ++nf) acc[pos][mf][nf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(Ah[mf], Bh[nf], acc[pos][mf][nf], 0, 0, 0);
  };

  // ---------------------------------------------------------------- prologue
  const int KS = a.KS;
  gload(hA, 0, 0);
  gload(hB, 0, 1);
  pstore(hA, 0);
  pstore(hB, 1);
  if (KS > 1) {
    gload(hA, 1, 0);
    gload(hB, 1, 1);
  }
  __syncthreads();
  transform(0);  // both groups, their own octet
  if (!roleA) uload(0, 0);
  __syncthreads();
  if (KS > 1) pstore(hA, 0);
  if (KS > 2) gload(hA, 2, 0);
  __syncthreads();
  stamp(1);

  // ---------------------------------------------------------------- steps: two half steps, a barrier after each
  // at the top of step s: V[s & 1] holds chunk s; half patch A holds chunk s + 1, hB's registers hold half B of chunk s + 1, hA's half A of chunk s + 2
  // (in flight); uf = step s.
  //   first half:   everyone: half patch B <- chunk s + 1, request half B of chunk s + 2;   A: transform chunk s + 1 (octet 0);   B: the step's 24 MFMAs
  //   second half:  everyone: half patch A <- chunk s + 2, request half A of chunk s + 3;   A: the step's 24 MFMAs;   B: transform chunk s + 1 (octet 1)
  // A position's U fragments of step s + 1 are requested as soon as its twelve MFMAs of step s are issued: a whole step of flight (requested at the
  // start of the matrix half they were an L2 round trip in front of it: 24 MFMAs took 1.5-2.8 k cycles).
  // (one loop per group, each a straight line of half steps: with the groups' bodies as branches inside ONE loop the accumulators met at the joins and the
  // compiler copied / spilled them every iteration)
  auto common = [&](int s, bool first) {
    if (first) {
      if (s + 1 < KS) pstore(hB, 1);
      if (s + 2 < KS) gload(hB, s + 2, 1);
    } else {
      if (s + 2 < KS) pstore(hA, 0);
      if (s + 3 < KS) gload(hA, s + 3, 0);
    }
  };
  if (roleA) {
#pragma unroll 1
    for (int s = 0; s < KS; ++s) {
      const int buf = s & 1;
      common(s, true);
      if (s + 1 < KS) {
        transform(buf ^ 1);
      }
      uload(s, 0);
      __syncthreads();
      stamp(2 + 2 * s);
      common(s, false);
      uload(s, 1);
      mma(buf, 0);
      mma(buf, 1);
      __syncthreads();
      stamp(3 + 2 * s);
    }
  } else {
#pragma unroll 1
    for (int s = 0; s < KS; ++s) {
      const int buf = s & 1;
      common(s, true);
      uload(s, 1);
      mma(buf, 0);
      mma(buf, 1);
      __syncthreads();
      common(s, false);
      if (s + 1 < KS) {
        transform(buf ^ 1);
        uload(s + 1, 0);
      }
      __syncthreads();
    }
  }

  // ---------------------------------------------------------------- output transform through LDS, 32 output channels per round
  // wave w: xi = w >> 1, nu pair = w & 1.  Y[a][b] = sum_xi At[a][xi] sum_nu At[b][nu] M[xi][nu], At = [[1, 1, 1, 0], [0, 1, -1, -1]]:
  // the wave's nu-partials are  b = 0: M0 + M1 | M2,   b = 1: M1 | -(M2 + M3)
  // X[(wave, b)][tile][32 channels], 128 bytes per tile with the 16-byte columns XOR-swizzled by the tile index (both the writers -- lane = tile --
  // and the readers -- eight lanes = one tile's 128 bytes -- are conflict-free)
  const int r_tile = tid >> 3, r_cq = tid & 7;   // reader: tile slot, channel quad of the round
  const int r_ty = r_tile / a.TBW, r_tx = r_tile - r_ty * a.TBW;
  const bool r_active = r_tile < a.ntile;
  const long long orow = ((long long)img * H + Y0 + 2 * r_ty) * W + X0 + 2 * r_tx;
  float* sc = reinterpret_cast<float*>(sm + PATCH_OFF);  // [wave][run of the round (<= 4)][2] (the half patches are free now)
#pragma unroll 1
  for (int r = 0; r < 2; ++r) {
#pragma unroll
    for (int nf = 0; nf < 2; ++nf) {
      f32x16 q0, q1;
      const f32x16 m0 = r ? acc[0][1][nf] : acc[0][0][nf], m1 = r ? acc[1][1][nf] : acc[1][0][nf];
      if (wave & 1) {
        q0 = m0;
        q1 = -(m0 + m1);
      } else {
        q0 = m0 + m1;
        q1 = m1;
      }
      const int tile = nf * 32 + (lane & 31);
      unsigned char* x0 = sm + ((wave * 2) * 64 + tile) * 128;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int col = ((2 * q + (lane >> 5)) ^ (tile & 7)) * 16;
        *reinterpret_cast<f32x4*>(x0 + col) = f32x4{q0[4 * q], q0[4 * q + 1], q0[4 * q + 2], q0[4 * q + 3]};
        *reinterpret_cast<f32x4*>(x0 + 64 * 128 + col) = f32x4{q1[4 * q], q1[4 * q + 1], q1[4 * q + 2], q1[4 * q + 3]};
      }
    }
    __syncthreads();
    float s1 = 0.f, s2 = 0.f;
    const int c0 = cb * 64 + r * 32 + r_cq * 4;
    if (r_active) {
      f32x4 y[2][2];  // [a][b]
      const unsigned char* xr = sm + r_tile * 128 + ((r_cq ^ (r_tile & 7)) * 16);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        f32x4 R[4];
#pragma unroll
        for (int xi = 0; xi < 4; ++xi)
          R[xi] = *reinterpret_cast<const f32x4*>(xr + (((2 * xi) * 2 + b) * 64) * 128) + *reinterpret_cast<const f32x4*>(xr + (((2 * xi + 1) * 2 + b) * 64) * 128);
        y[0][b] = (R[0] + R[1]) + R[2];
        y[1][b] = (R[1] - R[2]) - R[3];
      }
      f32x4 bias = {0.f, 0.f, 0.f, 0.f};
      if (p.bias) bias = *reinterpret_cast<const f32x4*>(p.bias + c0);
#pragma unroll
      for (int ya = 0; ya < 2; ++ya)
#pragma unroll
        for (int xb = 0; xb < 2; ++xb) {
          const long long row = orow + ya * W + xb;
          f32x4 v = y[ya][xb] + bias;
          s1 += (v.x + v.y) + (v.z + v.w);
          s2 += (v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w);
          if (p.res) v += *reinterpret_cast<const f32x4*>(p.res + row * p.ldres + c0);
          *reinterpret_cast<f32x4*>(p.out + row * p.ldo + c0) = v;
        }
    }
    if (p.gn_part) {
      // wave totals per run of gn_run channels: over the wave's eight tiles (lane bits 3-5), then over the run's channel quads (lane bits 0-2)
#pragma unroll
      for (int bit = 5; bit >= 3; --bit) {
        s1 += lane_xor(s1, bit);
        s2 += lane_xor(s2, bit);
      }
      s1 += lane_xor(s1, 0); s2 += lane_xor(s2, 0);                        // 8 channels
      if (a.gn_run >= 16) { s1 += lane_xor(s1, 1); s2 += lane_xor(s2, 1); }
      if (a.gn_run >= 32) { s1 += lane_xor(s1, 2); s2 += lane_xor(s2, 2); }
      const int qpr = a.gn_run >> 2;  // channel quads per run
      if (lane < 8 && (lane % qpr) == 0) {
        sc[(wave * 4 + lane / qpr) * 2] = s1;
        sc[(wave * 4 + lane / qpr) * 2 + 1] = s2;
      }
    }
    __syncthreads();
    if (p.gn_part) {
      const int nrun = 32 / a.gn_run;
      if (tid < nrun * 2) {
        const int run = tid >> 1, which = tid & 1;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < 8; ++w) v += sc[(w * 4 + run) * 2 + which];
        const int ch = cb * 64 + r * 32 + run * a.gn_run;
        const int cpg = p.Cout / p.gn_groups;
        const int grp = ch / cpg, rig = (ch - grp * cpg) / a.gn_run;
        const int fr = img - smp * p.a_imgs_per_sample;
        const long long n = (long long)p.a_imgs_per_sample * a.bpf * a.gn_rpg;
        const long long k = ((long long)fr * a.bpf + blk) * a.gn_rpg + rig;
        p.gn_part[(((long long)smp * p.gn_groups + grp) * n + k) * 2 + which] = v;
      }
    }
    stamp(30 + r);
  }
}

// shape planning shared by the launcher and the queries; 0 = inside the envelope
int plan_wino(const vmm_conv_desc& d, WArgs& a, bool& gn) {
  const bool shape_ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.off_h == -1 && d.off_w == -1 && d.sgn_h == 1 && d.sgn_w == 1 && d.Hv == d.Hin &&
                        d.Wv == d.Win && d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && d.rot_ncols == 0 &&
                        d.q_ncols == 0 && !d.wrap_h && !d.wrap_w && (d.Hin & 1) == 0 && (d.Win & 1) == 0 && d.Hin > 0 && d.Win > 0 && d.nimg > 0;
  const bool chan_ok = d.C1 > 0 && d.C1 % 16 == 0 && d.C2 % 16 == 0 && d.Cout % 64 == 0 && (d.lda1 & 3) == 0 && (!d.C2 || (d.lda2 & 3) == 0) &&
                       (d.ldo & 3) == 0 && (!d.res || (d.ldres & 3) == 0);
  if (!shape_ok || !chan_ok) return 1;
  if (d.a_mode != 0 && d.a_mode != 1) return 1;
  if (d.a_mode == 1 && (!d.a_coef || d.a_imgs_per_sample <= 0)) return -3;
  if (d.a_img_mod < 0 || d.a_img_mod >= d.nimg) return 1;
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  const int ldmax = max(d.lda1, d.C2 ? d.lda2 : 0);
  if (M * ldmax >= (1LL << 31)) return 1;  // (32-bit element offsets in the patch loader)
  // tile block: TBH x TBW tiles dividing the frame's tile grid, at most 64 of them, patch at most 324 pixels; the fullest block wins
  const int th = d.Hin / 2, tw = d.Win / 2;
  int best = 0, bh = 0, bw = 0;
  for (int h = 1; h <= th && h <= 64; ++h) {
    if (th % h) continue;
    for (int w = 1; w <= tw && h * w <= 64; ++w) {
      if (tw % w || (2 * h + 2) * (2 * w + 2) > PATCH_PIX) continue;
      const int n = h * w;
      // (ties: the wider block -- longer contiguous runs per patch row)
      if (n > best || (n == best && w > bw)) best = n, bh = h, bw = w;
    }
  }
  if (best < 32) return 1;  // less than half of the matrix tile's columns would be real
  a.p = d;
  a.TBH = bh; a.TBW = bw; a.PH = 2 * bh + 2; a.PW = 2 * bw + 2; a.ntile = best;
  a.nbx = tw / bw; a.bpf = a.nbx * (th / bh);
  a.KS = (d.C1 + d.C2) / 16;
  a.ncb = d.Cout / 64;
  if ((long long)d.nimg * a.bpf * a.ncb >= (1LL << 31)) return 1;
  a.gn_run = 32; a.gn_rpg = 1;
  gn = false;
  if (d.gn_part && d.gn_groups > 0 && !d.res && d.a_imgs_per_sample > 0 && d.Cout % d.gn_groups == 0 && d.nimg % d.a_imgs_per_sample == 0) {
    const int cpg = d.Cout / d.gn_groups;
    if (cpg == 8 || cpg == 16 || cpg == 32) gn = true, a.gn_run = cpg, a.gn_rpg = 1;
    else if (cpg % 32 == 0) gn = true, a.gn_run = 32, a.gn_rpg = cpg / 32;
  }
  if (!gn) a.p.gn_part = nullptr;
  return 0;
}

}  // namespace

// Number of GroupNorm partial-sum pairs per (sample, group) vmm_conv3x3_wino_bf16x3(d) will leave in d->gn_part, 0 when it will not (or when the
// descriptor is outside the kernel's envelope).  Pure host logic.
extern "C" int vmm_conv3x3_wino_fuses_gn(const vmm_conv_desc* dp) {
  WArgs a;
  bool gn = false;
  if (plan_wino(*dp, a, gn) != 0 || !gn) return 0;
  return dp->a_imgs_per_sample * a.bpf * a.gn_rpg;
}

// host-only query: 1 when vmm_conv3x3_wino_bf16x3 would take the descriptor as it stands
extern "C" int vmm_conv3x3_wino_accepts(const vmm_conv_desc* dp) {
  WArgs a;
  bool gn = false;
  return plan_wino(*dp, a, gn) == 0 ? 1 : 0;
}

// Weights: vmm_pack_weights fmt 8 (G g G^T, split, "A" fragment order).  Returns 1 (nothing launched) outside the envelope: 3 x 3 / stride 1 /
// zero padding 1, even H and W whose tile grid (H / 2 x W / 2) divides into blocks of 32..64 tiles with a patch of at most 324 pixels, C1 and C2
// multiples of 16, Cout a multiple of 64.  Fused operand transform (a_mode 1), a_img_mod, bias, residual and the GroupNorm partial sums
// (vmm_conv3x3_wino_fuses_gn slots) as in vmm_conv3x3_bf16x3; no split_tickets (the channel reduction is never split).
extern "C" int vmm_conv3x3_wino_bf16x3(const vmm_conv_desc* dp, vmm_stream_t stream) {
  WArgs a;
  bool gn = false;
  const int rc = plan_wino(*dp, a, gn);
  if (rc != 0) return rc;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const unsigned grid = (unsigned)((long long)dp->nimg * a.bpf * a.ncb);
  static const int dbg = [] { const char* e = getenv("VMM_WINO_DBG"); return e ? atoi(e) : 0; }();
  a.dbg = dbg;
  // VMM_WINO_TRACE=<k>: the k-th launch of this process prints where its workgroups' time went (mean stamp deltas, s_memtime ticks)
  static const int trace_launch = [] { const char* e = getenv("VMM_WINO_TRACE"); return e ? atoi(e) : -1; }();
  static int launch_no = 0;
  if (trace_launch >= 0 && launch_no++ == trace_launch) {
    const size_t n = (size_t)grid * 32;
    (void)hipMalloc(&a.trace, n * sizeof(unsigned long long));
    (void)hipMemset(a.trace, 0, n * sizeof(unsigned long long));
    hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, a);
    (void)hipStreamSynchronize((hipStream_t)stream);
    unsigned long long* h = (unsigned long long*)malloc(n * sizeof(unsigned long long));
    (void)hipMemcpy(h, a.trace, n * sizeof(unsigned long long), hipMemcpyDeviceToHost);
    (void)hipFree(a.trace);
    // slots 0..9 and 30..31: wave 0's phase boundaries (deltas in slot order); 10..16 / 20..26: step 1 in detail for wave 0 / wave 4 (deltas from the group's first stamp)
    double sum[32] = {0};
    for (size_t g = 0; g < grid; ++g) {
      unsigned long long prev = h[g * 32];
      for (int k = 1; k < 32; ++k) {
        if (!h[g * 32 + k]) continue;
        if (k >= 10 && k < 30) {
          const int k0 = k < 20 ? 10 : 20;
          sum[k] += (double)((long long)(h[g * 32 + k] - h[g * 32 + k0]));
        } else {
          sum[k] += (double)(h[g * 32 + k] - prev), prev = h[g * 32 + k];
        }
      }
    }
    fprintf(stderr, "[wino trace] %dx%d %d+%d->%d a_mode %d grid %u KS %d tiles %d; mean ticks:", dp->Hin, dp->Win, dp->C1, dp->C2, dp->Cout, dp->a_mode, grid, a.KS, a.ntile);
    for (int k = 1; k < 32; ++k)
      if (sum[k] != 0) fprintf(stderr, " [%d] %.0f", k, sum[k] / grid);
    fprintf(stderr, "\n");
    free(h);
    return 0;
  }
  hipLaunchKernelGGL(conv3x3_wino_kernel, dim3(grid), dim3(NTHR), LDS_BYTES, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
