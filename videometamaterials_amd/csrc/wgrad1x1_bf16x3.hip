// Weight gradient of the 1 x 1 convolutions / Linear layers (to_qkv, to_out, res_conv: autograd of vddp.py:297, 319, 325, 413, 421) on the bf16
// matrix cores with split (hi + lo) operands, gfx950 -- the one-tap sibling of wgrad3x3_bf16x3.hip.
//
//   dWp[ci][co] += sum over rows r   x[r][ci] * dY[r][co]
//
// The contraction runs over rows, so both MFMA operands need eight consecutive rows of one channel per lane while memory holds rows x
// channels.  As in the nine-tap kernel the loader's thread = (8 rows, 2 channels) reads eight 8-byte pieces (64 lanes cover 512 contiguous
// bytes of a row), splits them and writes one 16-byte hi and lo fragment per channel into an LDS image [channel][row] whose row pitch is an
// odd multiple of 16 bytes: exactly what a lane of the MFMA reads back with one ds_read_b128, conflict-free both ways.  A workgroup owns a
// 128 x 128 channel block and walks its row slice in chunks of 64 rows (double-buffered LDS, the next chunk's rows in flight in registers);
// 8 waves = 2 groups x 2 x 2 quadrants of 64 x 64 (four 32 x 32 accumulators each); the groups take alternate halves of a chunk and stage the
// next chunk before (group 0) / after (group 1) their MFMAs, one barrier per chunk.  Row slices leave partial blocks in a workspace with
// 16-byte stores, a second launch totals them in a fixed order (bit-reproducible; no atomics).  Channel counts need only be multiples of 64:
// the half of a 128-wide block that lies beyond C1 + C2 / Cout is neither loaded nor stored.
// The bias gradient (exact fp32 column sums of dY) leaves as one partial row per slice from the ci-block-0 workgroups.
#include "vmm_common.h"
#include "wgrad_reduce.h"
#include "../../include/vmm_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CH = 64;               // rows per chunk (four k16 steps, two per wave group)
constexpr int PITCH = 2 * CH + 16;   // bytes per channel row of a plane (144 = 9 x 16)
constexpr int PLANE = 128 * PITCH;   // one plane (hi or lo) of one operand: 128 channels
constexpr int BUF = 4 * PLANE;       // x hi | x lo | dY hi | dY lo
constexpr int BLOCK_FLOATS = 128 * 128;

struct W1Args {
  vmm_conv_desc p;
  const float* dy; int lddy;
  float* part;        // [gridDim.z][tiles][BLOCK_FLOATS]
  float* bias_part;   // [gridDim.z][Cout] or NULL
  long long rows;
  int nchunks, chunks_per_wg;
  const float* ln_stats;  // non-NULL: x = (a1 - mean[r]) * rstd[r] * ln_gamma[ci], (mean, rstd) = ln_stats[r][2] (PreNorm LayerNorm, vddp.py:245-254)
  const float* ln_gamma;
  // TAP instances (vmm_conv_wgrad_tap_*): the K axis is (tap, input channel), K = KH KW (C1 + C2); output row r = (img, a, b) meets the input pixel
  // (a stride + off_h + sgn_h kh, b stride + off_w + sgn_w kw) of tap (kh, kw) (zero or periodic padding) and the dY row of the descriptor's output scatter
  int Ktot, hw, wv;
  unsigned hw_magic, wv_magic;  // floor(2^32 / d) + 1: n / d = mulhi(n, magic) while n d < 2^32 (checked by the launcher)
};

template <bool TAP>
__global__ __launch_bounds__(512) void wgrad1_x3_kernel(const W1Args a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char sm[];
  const vmm_conv_desc& p = a.p;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int grp = wave >> 2, wq = wave & 3, wu = wq >> 1, wv = wq & 1;
  const int l31 = lane & 31, half = lane >> 5;
  const int Cin = p.C1 + p.C2;
  const int ci0 = blockIdx.x * 128, co0 = blockIdx.y * 128;
  const int c_begin = blockIdx.z * a.chunks_per_wg;
  const int n_it = min(a.chunks_per_wg, a.nchunks - c_begin);
  if (n_it <= 0) return;
  const long long r_begin = (long long)c_begin * CH;

  // ---------------------------------------------------------------- loader: thread = (8-row piece `oct`, channel pair `cp`) of BOTH operands
  const int oct = wave, cp = tid & 63;  // (oct = tid >> 6 is the wave: wave-uniform, so the per-row LayerNorm statistics below are scalar loads)
  const int xc = ci0 + 2 * cp, yc = co0 + 2 * cp;
  const bool x_ok = xc < (TAP ? a.Ktot : Cin), y_ok = yc < p.Cout;
  // TAP: this thread's column pair of K = (tap, channel): the tap's offset into the input grid, the channel inside its source (a pair never straddles taps: Cin is even)
  const int tap = TAP ? (x_ok ? xc / Cin : 0) : 0;
  const int xci = TAP ? xc - tap * Cin : xc;
  const int tkh = TAP ? tap / p.KW : 0, tkw = TAP ? tap - tkh * p.KW : 0;
  const int dih = TAP ? p.off_h + p.sgn_h * tkh : 0, diw = TAP ? p.off_w + p.sgn_w * tkw : 0;
  const bool x_src1 = xci < p.C1;
  const float* xsrc = x_src1 ? p.a1 + xci : p.a2 + (xci - p.C1);
  const int xld = x_src1 ? p.lda1 : p.lda2;
  const float* ysrc = a.dy + yc;
  f32x2 xv[8], yv[8], sv[8];
  unsigned xmask = 0xffu;  // TAP: which of the eight rows in flight read inside the image (the others are padding: staged as zeros)
  f32x2 bsum = {0.f, 0.f};
  const bool ln = a.ln_stats != nullptr;  // (workgroup-uniform)
  const f32x2 lg = (ln && x_ok) ? *reinterpret_cast<const f32x2*>(a.ln_gamma + xc) : f32x2{1.f, 1.f};
  // (every load is unconditional: rows past the end and channels past the layer re-read row 0 / channel 0 and are zeroed when staged)
  auto request = [&](long long r0) __attribute__((always_inline)) {  // (the TAP body is long: not inlined, its captures -- the rows in flight -- would live in scratch)
    if constexpr (TAP) {
      unsigned m = 0;
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        const long long r = r0 + i;
        const bool v = r < a.rows;
        // (r is wave-uniform -- a wave owns eight rows, its lanes the column pairs --: the decode runs on the scalar unit)
        const unsigned ru = v ? (unsigned)r : 0u;
        const unsigned img = __umulhi(ru, a.hw_magic);
        const unsigned rem = ru - img * (unsigned)a.hw;
        const unsigned oa = __umulhi(rem, a.wv_magic);
        const unsigned ob = rem - oa * (unsigned)a.wv;
        int ih = (int)oa * p.stride + dih, iw = (int)ob * p.stride + diw;
        if (p.wrap_h) ih = ih < 0 ? ih + p.Hin : (ih >= p.Hin ? ih - p.Hin : ih);  // periodic padding (vddp.py:163-243)
        if (p.wrap_w) iw = iw < 0 ? iw + p.Win : (iw >= p.Win ? iw - p.Win : iw);
        const bool okx = v && x_ok && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win;
        const long long xrow = ((long long)img * p.Hin + ih) * p.Win + iw;
        const long long orow = ((long long)img * p.Hout + (int)oa * p.oscale + p.ooh) * p.Wout + (int)ob * p.oscale + p.oow;
        xv[i] = *reinterpret_cast<const f32x2*>(okx ? xsrc + xrow * xld : p.a1);
        yv[i] = *reinterpret_cast<const f32x2*>((v && y_ok) ? ysrc + orow * a.lddy : a.dy);
        m |= (okx ? 1u : 0u) << i;
      }
      xmask = m;
      return;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const long long r = r0 + i;
      const bool v = r < a.rows;
      xv[i] = *reinterpret_cast<const f32x2*>((v && x_ok) ? xsrc + r * xld : p.a1);
      yv[i] = *reinterpret_cast<const f32x2*>((v && y_ok) ? ysrc + r * a.lddy : a.dy);
      if (ln) sv[i] = *reinterpret_cast<const f32x2*>(a.ln_stats + 2 * (v ? r : 0));
    }
  };
  auto stage = [&](long long r0, int buf) __attribute__((always_inline)) {
    unsigned char* base = sm + buf * BUF + oct * 16;
#pragma unroll
    for (int op = 0; op < 2; ++op) {
      const bool ok = op ? y_ok : x_ok;
#pragma unroll
      for (int c = 0; c < 2; ++c) {
        float e[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) e[i] = (ok && r0 + i < a.rows && (!TAP || op || ((xmask >> i) & 1u))) ? (op ? yv[i][c] : xv[i][c]) : 0.f;
        if (ln && !op) {
#pragma unroll
          for (int i = 0; i < 8; ++i) e[i] = (ok && r0 + i < a.rows) ? (e[i] - sv[i][0]) * (sv[i][1] * lg[c]) : 0.f;
        }
        uint4 h, l;
        h.x = split_bf16_pair(e[0], e[1], l.x);
        h.y = split_bf16_pair(e[2], e[3], l.y);
        h.z = split_bf16_pair(e[4], e[5], l.z);
        h.w = split_bf16_pair(e[6], e[7], l.w);
        unsigned char* dst = base + op * 2 * PLANE + (2 * cp + c) * PITCH;
        *reinterpret_cast<uint4*>(dst) = h;
        *reinterpret_cast<uint4*>(dst + PLANE) = l;
        if (op) bsum[c] += ((e[0] + e[1]) + (e[2] + e[3])) + ((e[4] + e[5]) + (e[6] + e[7]));
      }
    }
  };

  f32x16 acc[2][2];  // [ci fragment of the quadrant][co fragment]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  request(r_begin + 8 * oct);
  stage(r_begin + 8 * oct, 0);
  if (n_it > 1) request(r_begin + CH + 8 * oct);
  __syncthreads();

  const int a_lane = (wu * 64 + l31) * PITCH + half * 16, b_lane = 2 * PLANE + (wv * 64 + l31) * PITCH + half * 16;
  auto loader_phase = [&](int it) __attribute__((always_inline)) {
    if (it + 1 < n_it) {
      stage(r_begin + (long long)(it + 1) * CH + 8 * oct, (it + 1) & 1);
      if (it + 2 < n_it) request(r_begin + (long long)(it + 2) * CH + 8 * oct);
    }
  };
  for (int it = 0; it < n_it; ++it) {
    if (grp == 0) loader_phase(it);
    const unsigned char* buf = sm + (it & 1) * BUF;
#pragma unroll
    for (int ss = 0; ss < 2; ++ss) {
      const int s = 2 * grp + ss;
      bf16x8 Ah[2], Al[2], Bh[2], Bl[2];
#pragma unroll
      for (int f = 0; f < 2; ++f) {
        Ah[f] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + a_lane + f * 32 * PITCH + s * 32));
        Al[f] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + PLANE + a_lane + f * 32 * PITCH + s * 32));
        Bh[f] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + b_lane + f * 32 * PITCH + s * 32));
        Bl[f] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(buf + PLANE + b_lane + f * 32 * PITCH + s * 32));
      }
      // pass-major, lo products first
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) if constexpr (!VMM_SINGLE_PASS) acc[i][j] = vmm_mfma16(Ah[i], Bl[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) if constexpr (!VMM_SINGLE_PASS) acc[i][j] = vmm_mfma16(Al[i], Bh[j], acc[i][j]);
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = vmm_mfma16(Ah[i], Bh[j], acc[i][j]);
    }
    if (grp == 1) loader_phase(it);
    __syncthreads();
  }

  // ---------------------------------------------------------------- epilogue: group 1 hands its accumulators over through LDS, group 0 adds them
  // and stores the workgroup's partial block: layout [quadrant][i][j][q][lane] x 4 floats (coalesced 16-byte stores; the reduction undoes it)
  uint4* xch = reinterpret_cast<uint4*>(sm);
  if (grp == 1) {
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q)
          xch[(((wq * 2 + i) * 2 + j) * 4 + q) * 64 + lane] = uint4{__float_as_uint(acc[i][j][4 * q]), __float_as_uint(acc[i][j][4 * q + 1]),
                                                                    __float_as_uint(acc[i][j][4 * q + 2]), __float_as_uint(acc[i][j][4 * q + 3])};
  }
  __syncthreads();
  if (grp == 0) {
    f32x4* dst = reinterpret_cast<f32x4*>(a.part) + ((long long)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * (BLOCK_FLOATS / 4);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int k = (((wq * 2 + i) * 2 + j) * 4 + q) * 64 + lane;
          const uint4 v = xch[k];
          dst[k] = f32x4{acc[i][j][4 * q] + __uint_as_float(v.x), acc[i][j][4 * q + 1] + __uint_as_float(v.y), acc[i][j][4 * q + 2] + __uint_as_float(v.z),
                         acc[i][j][4 * q + 3] + __uint_as_float(v.w)};
        }
  }
  if (a.bias_part && blockIdx.x == 0) {  // (workgroup-uniform) eight pieces x 128 channels of partial column sums -> one value per channel
    __syncthreads();
    float* red = reinterpret_cast<float*>(sm);
    red[oct * 128 + 2 * cp] = bsum[0];
    red[oct * 128 + 2 * cp + 1] = bsum[1];
    __syncthreads();
    if (tid < 128 && co0 + tid < p.Cout) {
      float s = 0.f;
#pragma unroll
      for (int o = 0; o < 8; ++o) s += red[o * 128 + tid];
      a.bias_part[(long long)blockIdx.z * p.Cout + co0 + tid] = s;
    }
  }
}

// dw[ci][co] += sum over row slices z of the partial blocks (fixed order); body in wgrad_reduce.h (shared with vmm_reduce_batch)
static_assert(BLOCK_FLOATS == vmm_reduce::W1_BLOCK_FLOATS, "partial block size");
__global__ __launch_bounds__(256) void wgrad1_reduce_kernel(const float* __restrict__ part, int nz, int tiles_x, int tiles_y, float* __restrict__ dw, int Cin,
                                                            int Cout, const float* __restrict__ bias_part, float* __restrict__ dbias, int n_main) {
  __shared__ f32x4 red[8][32];
  vmm_reduce::w1_body(part, nz, tiles_x, tiles_y, dw, Cin, Cout, bias_part, dbias, n_main, (int)blockIdx.x, (int)blockIdx.y, red);
}

static bool w1_setup(const vmm_conv_desc& d, int32_t lddy, W1Args& a, int& gz, int& tx, int& ty) {
  const bool shape_ok = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.off_h == 0 && d.off_w == 0 && d.Hv == d.Hin && d.Wv == d.Win && d.oscale == 1 &&
                        d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && d.a_mode == 0 && !d.a_img_mod;
  const bool chan_ok = d.C1 > 0 && d.C1 % 64 == 0 && d.C2 % 64 == 0 && d.Cout % 64 == 0 && (d.lda1 & 1) == 0 && (!d.C2 || (d.lda2 & 1) == 0) && (lddy & 1) == 0;
  if (!shape_ok || !chan_ok || d.nimg <= 0 || d.Hin <= 0 || d.Win <= 0) return false;
  a.p = d;
  a.rows = (long long)d.nimg * d.Hv * d.Wv;
  if (a.rows >= (1ll << 31) * CH) return false;
  a.nchunks = (int)((a.rows + CH - 1) / CH);
  tx = (d.C1 + d.C2 + 127) / 128;
  ty = (d.Cout + 127) / 128;
  const int nz = max(1, min(a.nchunks, 256 / (tx * ty)));  // one round of workgroups, one per CU
  a.chunks_per_wg = (a.nchunks + nz - 1) / nz;
  gz = (a.nchunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  return true;
}

// TAP form: any kernel size / stride / output scatter of the descriptor (the 4 x 4 stride-2 convolutions, the four phases of the transposed ones, the 7 x 7 stem),
// zero or periodic padding; no fused operand transform; channel counts even, K = KH KW (C1 + C2) a multiple of four
static bool w1_setup_tap(const vmm_conv_desc& d, int32_t lddy, W1Args& a, int& gz, int& tx, int& ty) {
  const int Cin = d.C1 + d.C2;
  const bool shape_ok = d.KH >= 1 && d.KW >= 1 && d.KH * d.KW <= 1024 && d.stride >= 1 && d.oscale >= 1 && d.a_mode == 0 && !d.a_img_mod && d.Hv > 0 && d.Wv > 0 && d.Hout > 0 &&
                        d.Wout > 0;
  const bool chan_ok = d.C1 > 0 && (d.C1 & 1) == 0 && (d.C2 & 1) == 0 && (d.Cout & 1) == 0 && (d.lda1 & 1) == 0 && (!d.C2 || (d.lda2 & 1) == 0) && (lddy & 1) == 0 &&
                       ((long long)d.KH * d.KW * Cin) % 4 == 0;
  if (!shape_ok || !chan_ok || d.nimg <= 0 || d.Hin <= 0 || d.Win <= 0) return false;
  a.p = d;
  a.rows = (long long)d.nimg * d.Hv * d.Wv;
  a.hw = d.Hv * d.Wv;
  a.wv = d.Wv;
  // (the magic divisions: n / d = mulhi(n, floor(2^32 / d) + 1) holds while n d < 2^32)
  if (a.rows * a.hw >= (1ll << 32) || (long long)a.hw * a.wv >= (1ll << 32) || (long long)d.nimg * d.Hin * d.Win >= (1ll << 31) ||
      (long long)d.nimg * d.Hout * d.Wout >= (1ll << 31))
    return false;
  if (a.hw < 2 || a.wv < 2) return false;  // (a divisor of 1 has no 32-bit magic; such layers stay on the generic kernel)
  a.hw_magic = (unsigned)(0x100000000ull / (unsigned)a.hw) + 1u;
  a.wv_magic = (unsigned)(0x100000000ull / (unsigned)a.wv) + 1u;
  a.Ktot = d.KH * d.KW * Cin;
  a.nchunks = (int)((a.rows + CH - 1) / CH);
  tx = (a.Ktot + 127) / 128;
  ty = (d.Cout + 127) / 128;
  const int nz = max(1, min(a.nchunks, 256 / (tx * ty)));
  a.chunks_per_wg = (a.nchunks + nz - 1) / nz;
  gz = (a.nchunks + a.chunks_per_wg - 1) / a.chunks_per_wg;
  return true;
}

}  // namespace

// floats of workspace vmm_conv1x1_wgrad_bf16x3 wants for this layer; 0 = outside the kernel's envelope (1 x 1, stride 1, identity rows, no fused
// operand transform, C1 / C2 / Cout multiples of 64)
#if !VMM_SINGLE_PASS
extern "C" int64_t vmm_conv1x1_wgrad_bf16x3_workspace(const vmm_conv_desc* dp, int32_t lddy) {
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!w1_setup(*dp, lddy, a, gz, tx, ty)) return 0;
  return (int64_t)gz * tx * ty * BLOCK_FLOATS + (int64_t)gz * dp->Cout;
}
// the second stage of vmm_conv1x1_wgrad_*(d, dy, lddy, dw_packed, dbias, workspace) (and of the _ln form) as a job of vmm_reduce_batch (d->defer_reduce)
extern "C" int vmm_conv1x1_wgrad_reduce_job(const vmm_conv_desc* dp, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job) {
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!workspace || !job || !w1_setup(*dp, lddy, a, gz, tx, ty)) return 1;
  const int n_main = BLOCK_FLOATS / 4 / 32;
  job->part = workspace; job->out = dw_packed;
  job->bias_part = dbias ? workspace + (long long)gz * tx * ty * BLOCK_FLOATS : nullptr; job->dbias = dbias;
  job->kind = 2; job->nz = gz; job->tiles_x = tx; job->tiles_y = ty; job->Cin = dp->C1 + dp->C2; job->Cout = dp->Cout; job->ld = 0;
  job->n_main = n_main; job->gx = n_main + (dbias ? cdiv(dp->Cout, 32) : 0); job->wgs = job->gx * tx * ty; job->wg0 = 0;
  return 0;
}
#endif

#if !VMM_SINGLE_PASS
// ... and of the TAP form (vmm_conv_wgrad_tap_*): workspace floats (0 = outside its envelope), the deferred second stage as a vmm_reduce_batch job
extern "C" int64_t vmm_conv_wgrad_tap_workspace(const vmm_conv_desc* dp, int32_t lddy) {
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!w1_setup_tap(*dp, lddy, a, gz, tx, ty)) return 0;
  return (int64_t)gz * tx * ty * BLOCK_FLOATS + (int64_t)gz * dp->Cout;
}
extern "C" int vmm_conv_wgrad_tap_reduce_job(const vmm_conv_desc* dp, int32_t lddy, float* dw_packed, float* dbias, float* workspace, vmm_reduce_job* job) {
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!workspace || !job || !w1_setup_tap(*dp, lddy, a, gz, tx, ty)) return 1;
  const int n_main = BLOCK_FLOATS / 4 / 32;
  job->part = workspace; job->out = dw_packed;
  job->bias_part = dbias ? workspace + (long long)gz * tx * ty * BLOCK_FLOATS : nullptr; job->dbias = dbias;
  job->kind = 2; job->nz = gz; job->tiles_x = tx; job->tiles_y = ty; job->Cin = a.Ktot; job->Cout = dp->Cout; job->ld = 0;
  job->n_main = n_main; job->gx = n_main + (dbias ? cdiv(dp->Cout, 32) : 0); job->wgs = job->gx * tx * ty; job->wg0 = 0;
  return 0;
}
#endif

// dw_packed[(kh, kw, ci)][co] += sum over output rows of x[input pixel of the tap][ci] dY[output row][co] for ANY convolution geometry the descriptor states -- kernel
// size, stride, tap direction (sgn: the transposed convolutions' phases), output scatter (oscale / ooh / oow), zero or periodic padding -- on the matrix cores
// (the 4 x 4 stride-2 layers, the four 2 x 2 phases of the transposed ones, the 7 x 7 stem: exact-fp32 kernel before round 6).  The 1 x 1 kernel with a loader that
// decodes its eight wave-uniform rows once (scalar unit) and offsets them by the lane's tap.  dbias[co] += column sums of the dY rows the call visits.
extern "C" int VMM_X3(vmm_conv_wgrad_tap_, )(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                                         vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!workspace || !w1_setup_tap(d, lddy, a, gz, tx, ty)) return 1;
  a.ln_stats = nullptr;
  a.ln_gamma = nullptr;
  a.dy = dy; a.lddy = lddy;
  a.part = workspace;
  a.bias_part = dbias ? workspace + (long long)gz * tx * ty * BLOCK_FLOATS : nullptr;
  const size_t shm = 2 * (size_t)BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1_x3_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad1_x3_kernel<true>, dim3(tx, ty, gz), dim3(512), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  if (d.defer_reduce) return 0;
  const int n_main = BLOCK_FLOATS / 4 / 32;
  hipLaunchKernelGGL(wgrad1_reduce_kernel, dim3(n_main + (dbias ? cdiv(d.Cout, 32) : 0), tx * ty), dim3(256), 0, (hipStream_t)stream, workspace, gz, tx, ty, dw_packed,
                     a.Ktot, d.Cout, a.bias_part, dbias, n_main);
  VMM_LAUNCH_CHECK();
  return 0;
}

// dw_packed[ci][co] += sum_r x[r][ci] dY[r][co] (and dbias[co] += sum_r dY[r][co] when dbias != NULL); d = the FORWARD descriptor of the layer;
// workspace = vmm_conv1x1_wgrad_bf16x3_workspace(d, lddy) floats (contents irrelevant).  Returns 1 (nothing launched) outside the envelope.
static int w1_launch(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace, const float* ln_stats,
                     const float* ln_gamma, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  W1Args a;
  int gz = 0, tx = 0, ty = 0;
  if (!workspace || !w1_setup(d, lddy, a, gz, tx, ty)) return 1;
  a.ln_stats = ln_stats;
  a.ln_gamma = ln_gamma;
  a.dy = dy; a.lddy = lddy;
  a.part = workspace;
  a.bias_part = dbias ? workspace + (long long)gz * tx * ty * BLOCK_FLOATS : nullptr;
  const size_t shm = 2 * (size_t)BUF;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad1_x3_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad1_x3_kernel<false>, dim3(tx, ty, gz), dim3(512), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  if (d.defer_reduce) return 0;  // (the caller totals the blocks later: vmm_conv1x1_wgrad_reduce_job + vmm_reduce_batch)
  const int n_main = BLOCK_FLOATS / 4 / 32;
  hipLaunchKernelGGL(wgrad1_reduce_kernel, dim3(n_main + (dbias ? cdiv(d.Cout, 32) : 0), tx * ty), dim3(256), 0, (hipStream_t)stream, workspace, gz, tx, ty, dw_packed,
                     d.C1 + d.C2, d.Cout, a.bias_part, dbias, n_main);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int VMM_X3(vmm_conv1x1_wgrad_, )(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, float* dbias, float* workspace,
                                        vmm_stream_t stream) {
  return w1_launch(dp, dy, lddy, dw_packed, dbias, workspace, nullptr, nullptr, stream);
}

// The same with the layer's input normalised on the way in: x[r][ci] = (a1[r][ci] - mean[r]) * rstd[r] * ln_gamma[ci], (mean, rstd) = ln_stats[r][2] as
// vmm_proj_bf16x3_ln_stats left them -- the weight gradient of a PreNorm(to_qkv) whose forward fused the LayerNorm (single source: C2 == 0).
extern "C" int VMM_X3(vmm_conv1x1_wgrad_, _ln)(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, float* workspace, const float* ln_stats,
                                           const float* ln_gamma, vmm_stream_t stream) {
  if (!ln_stats || !ln_gamma || dp->C2) return -1;
  return w1_launch(dp, dy, lddy, dw_packed, nullptr, workspace, ln_stats, ln_gamma, stream);
}
