// Backward of the fused spatial linear-attention block (linattn_block.hip) at the C = 64 levels WITH RECOMPUTATION, split-bf16 matrix cores, gfx950.
//
//   out = x + to_out( ctx^T . softmax_d(q) * scale ) + bias,   ctx = softmax_n([k_tok | k]) . ([v_tok | v] / HW)^T,   q,k,v = to_qkv(LayerNorm(x))
//   (vddp.py:313-378 SpatialLinearAttention inside Residual(PreNorm(.)), vddp.py:613/628)
//
// The training forward is the fused block: it keeps its small workspace (per frame / slice / head the key softmax's max, sum and 32 x 32 context)
// and stores no qkv rows (1.25 GB per 96 x 96 site at batch 4), no attention output.  With sm = softmax_d(q), ks = softmax_n(k), ctxn = the
// normalised context, dO = dOut . W_out:
//   dctx[d][e] = scale sum_n sm[d][n] dO[n][e]                                   (needs all pixels of a frame: pass 1)
//   dq0 = sm (dqs - sum_d sm dqs),  dqs[d][n] = scale sum_e ctxn[d][e] dO[n][e]
//   dk0 = ks (dks - delta[d]),      dks[d][n] = sum_e dctx[d][e] v[e][n] / HW,   delta[d] = sum_e dctx[d][e] ctxn[d][e]
//   dv0[e][n] = sum_d ks[d][n] dctx[d][e] / HW                                   (pass 2, rows of the raw qkv gradient)
//   dW_out[e][c] += sum_n out[n][e] dOut[n][c],  out[n][e] = sum_d sm[d][n] scale ctxn[d][e]   (pass 1)
//   d(ek), d(ev): the token columns of dk0 / dv0                                 (combine kernel, per frame)
// Kernels (one wave = one head, 32-pixel tiles, products chained through the accumulator registers, chain_mfma.h):
//   pass 1   la_bwd_ctx_kernel    x, dOut -> per (frame, slice, head) partial dctx (registers) and the workgroup's dW_out (LDS accumulators)
//   combine  la_bwd_combine_kernel forward partials + tokens -> max / 1/sum / delta per feature, scale ctxn and dctx / HW as operand fragments, token gradients
//   pass 2   la_bwd_rows_kernel   x, dOut -> dq0^T, dk0^T, dv0^T {feature, pixel}: every product lands in the orientation the store wants, no transposition
// HBM traffic per site: x and dOut read twice, the qkv gradient written once (it feeds the fused to_qkv backward, qkv_bwd.hip).
#include "chain_mfma.h"
#include "linattn_split.h"
#include "../../include/vmm_kernels.h"
#include <math.h>

namespace {

using namespace chain;

constexpr int TC = 64;
constexpr int LH = 8;    // heads = waves per workgroup
constexpr int LD = 32;   // dim_head
constexpr int HID = LH * LD;
constexpr int YP = 2 * TC + 8;   // bf16 per row of a row image (hi 64 | lo 64 | pad 8)
constexpr int GP = 40;           // bf16 per channel of the column image (32 positions + pad)
constexpr int TAB = 3 * 32 + 3 * 1024;  // floats per (frame, head) of the pass-2 tables: max | 1/sum | delta | CA | DA | DB (each 256 uint4 = 1024 floats)

struct LBArgs {
  const float* x; int ldx;
  const float* gamma; float eps;
  const uint4* wqkv;   // fmt 2 of to_qkv (768, 64)
  const uint4* woT;    // fmt 2 of the (K = 64, N = 256) operand of to_out
  const float* ek; const float* ev; int ntok;
  const float* fpart;        // forward partials [frame][fsplit][head][LA_PART]
  const uint4* fctx;         // forward context fragments [frame][head][256]: scale ctxn as B[k = d][n = e]
  int fnsplit;
  const float* gout; int ldg;
  vmm_dqkv_t* gqkv; int ldq;        // (16-bit rows in the single-pass builds, vmm_common.h: VMM_DQKV16)
  float* ln_stats;
  float* p1;           // [frame][nsplit][head][1024]: partial dctx[d][e] (without the scale)
  float* part_wo;      // [frame * nsplit][256 * 64]
  float* tab;          // [frame][head][TAB]
  float* part_ek; float* part_ev;  // [T][B][ntok][256]
  int B, T, HW, nsplit, sps;
  float q_scale;
};

// staging of one 32-pixel tile by the whole workgroup: thread (row tid >> 4, channels (tid & 15) * 4 .. + 3)
struct Stager {
  int rm, rcol, gpos;
  f32x4 gam;
};

__device__ __forceinline__ void stage_tile(const LBArgs& a, const Stager& st, const f32x4& xv, const f32x4& gv, unsigned short* ytile, unsigned short* gtile,
                                           unsigned short* gcol, float* stats_row) {
  float s = (xv.x + xv.y) + (xv.z + xv.w);
  const float mean = row_sum16(s) * (1.0f / TC);
  const f32x4 c = {xv.x - mean, xv.y - mean, xv.z - mean, xv.w - mean};
  const float q = (c.x * c.x + c.y * c.y) + (c.z * c.z + c.w * c.w);
  const float rstd = 1.0f / sqrtf(row_sum16(q) * (1.0f / TC) + a.eps);
  if (stats_row && (threadIdx.x & 15) == 0) *reinterpret_cast<float2*>(stats_row) = make_float2(mean, rstd);
  unsigned l0, l1;
  unsigned h0 = split_bf16_pair(c.x * rstd * st.gam.x, c.y * rstd * st.gam.y, l0);
  unsigned h1 = split_bf16_pair(c.z * rstd * st.gam.z, c.w * rstd * st.gam.w, l1);
  unsigned short* yt = ytile + st.rm * YP + st.rcol;
  *reinterpret_cast<uint2*>(yt) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(yt + TC) = make_uint2(l0, l1);
  h0 = split_bf16_pair(gv.x, gv.y, l0);
  h1 = split_bf16_pair(gv.z, gv.w, l1);
  unsigned short* gt = gtile + st.rm * YP + st.rcol;
  *reinterpret_cast<uint2*>(gt) = make_uint2(h0, h1);
  *reinterpret_cast<uint2*>(gt + TC) = make_uint2(l0, l1);
  if (gcol) {
    unsigned short* gc = gcol + st.rcol * GP + st.gpos;
    gc[0] = (unsigned short)(h0 & 0xffffu);
    gc[GP] = (unsigned short)(h0 >> 16);
    gc[2 * GP] = (unsigned short)(h1 & 0xffffu);
    gc[3 * GP] = (unsigned short)(h1 >> 16);
    gc += 64 * GP;
    gc[0] = (unsigned short)(l0 & 0xffffu);
    gc[GP] = (unsigned short)(l0 >> 16);
    gc[2 * GP] = (unsigned short)(l1 & 0xffffu);
    gc[3 * GP] = (unsigned short)(l1 >> 16);
  }
}

// softmax over the rows (features) of a T-form accumulator {d, n}: 16 registers + the lane ^ 32 partner
__device__ __forceinline__ void softmax_rows(f32x16& q) {
  float mx = q[0];
#pragma unroll
  for (int r = 1; r < 16; ++r) mx = fmaxf(mx, q[r]);
  mx = fmaxf(mx, lane_xor(mx, 5));
  float sum = 0.f;
#pragma unroll
  for (int r = 0; r < 16; ++r) { q[r] = __expf(q[r] - mx); sum += q[r]; }
  sum += lane_xor(sum, 5);
  const float inv = __builtin_amdgcn_rcpf(sum);
#pragma unroll
  for (int r = 0; r < 16; ++r) q[r] *= inv;
}

// ---------------------------------------------------------------------------------------------------------------- pass 1
__global__ __launch_bounds__(512, 2) void la_bwd_ctx_kernel(const LBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  f32x4* dwo = reinterpret_cast<f32x4*>(smem_raw);                                   // [8 heads][2 channel tiles][4 quads][64 lanes]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(dwo + LH * 8 * 64);      // [2][32 * YP]
  unsigned short* gtile = ytile + 2 * 32 * YP;                                       // [2][32 * YP]
  unsigned short* gcol = gtile + 2 * 32 * YP;                                        // [2][hi|lo][64 * GP]
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lk = lane >> 5;
  const int frame = blockIdx.x / a.nsplit, split = blockIdx.x - frame * a.nsplit;
  const int tiles = a.HW / 32;
  const int t_begin = split * a.sps, t_end = min(tiles, t_begin + a.sps);

  uint4 wq[4][2], wo[4][2], cb[2][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4* q = a.wqkv + (((long long)h * 4 + s) * 2) * 64 + lane;
    const uint4* o = a.woT + (((long long)h * 4 + s) * 2) * 64 + lane;
    wq[s][0] = q[0]; wq[s][1] = q[64];
    wo[s][0] = o[0]; wo[s][1] = o[64];
  }
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    const uint4* q = a.fctx + ((long long)frame * LH + h) * 256 + (s * 2) * 64 + lane;
    cb[s][0] = q[0]; cb[s][1] = q[64];
  }
  uint4 I[2];
  identity_frags(lane, I);
  f32x4* dwo_l = dwo + (h * 8) * 64 + lane;
#pragma unroll
  for (int i = 0; i < 8; ++i) dwo_l[i * 64] = f32x4{0.f, 0.f, 0.f, 0.f};

  Stager st;
  st.rm = tid >> 4; st.rcol = (tid & 15) * 4;
  st.gpos = (st.rm >> 4) * 16 + ((st.rm >> 2) & 1) * 8 + (st.rm & 3) + 4 * ((st.rm >> 3) & 1);
  st.gam = *reinterpret_cast<const f32x4*>(a.gamma + st.rcol);
  unsigned x_loff = (unsigned)(st.rm * a.ldx + st.rcol), g_loff = (unsigned)(st.rm * a.ldg + st.rcol);
  auto load_xg = [&](int t, f32x4& xv, f32x4& gv) {
    xv = f32x4{0.f, 0.f, 0.f, 0.f};
    gv = xv;
    if (t < t_end) {
      const long long row0 = (long long)frame * a.HW + t * 32;
      xv = *reinterpret_cast<const f32x4*>(a.x + row0 * a.ldx + x_loff);
      gv = *reinterpret_cast<const f32x4*>(a.gout + row0 * a.ldg + g_loff);
    }
  };

  f32x16 dctx = zero16();  // {d, e}
  f32x4 xv, gv;
  load_xg(t_begin, xv, gv);
  stage_tile(a, st, xv, gv, ytile, gtile, gcol, nullptr);
  load_xg(t_begin + 1, xv, gv);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    asm volatile("" : "+v"(x_loff), "+v"(g_loff));
    const unsigned short* yt = ytile + buf * 32 * YP + lrow * YP + lk * 8;
    const unsigned short* gt = gtile + buf * 32 * YP + lrow * YP + lk * 8;
    // q^T {d, n} -> softmax over d
    f32x16 qT = zero16(), dor = zero16();
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
      qT = mfma3(wq[s][0], wq[s][1], yh, yl, qT);
      const uint4 gh = *reinterpret_cast<const uint4*>(gt + s * 16), gl = *reinterpret_cast<const uint4*>(gt + s * 16 + TC);
      dor = mfma3(gh, gl, wo[s][0], wo[s][1], dor);  // dO {n, e}: the rows are the A operand
    }
    softmax_rows(qT);
    const F2 smf = tofrag(qT);
    // out {n, e} = sm^T . (scale ctxn) and this head's rows of dW_out
    {
      f32x16 o = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) o = mfma3(smf.h[s], smf.l[s], cb[s][0], cb[s][1], o);
      const F2 of = tofrag(o);
      const unsigned short* gc = gcol + buf * 2 * 64 * GP + lrow * GP + lk * 8;
#pragma unroll
      for (int ct = 0; ct < 2; ++ct) {
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
    }
    // dctx {d, e} += sm {n, d}^T . dO {n, e}
    {
      const F2 smr = tofrag(transp(smf, I));
      const F2 dorf = tofrag(dor);
      dctx = mmT(smr, dorf, dctx);
    }
    // the next tile's rows into the other buffer, the one after it requested
    stage_tile(a, st, xv, gv, ytile + (buf ^ 1) * 32 * YP, gtile + (buf ^ 1) * 32 * YP, gcol + (buf ^ 1) * 2 * 64 * GP, nullptr);
    load_xg(t + 2, xv, gv);
    __syncthreads();
  }
  float* pp = a.p1 + (((long long)frame * a.nsplit + split) * LH + h) * 1024;
#pragma unroll
  for (int r = 0; r < 16; ++r) pp[row_of(r, lk) * LD + lrow] = dctx[r];
  float* pw = a.part_wo + (long long)blockIdx.x * (HID * TC);
#pragma unroll
  for (int ct = 0; ct < 2; ++ct)
#pragma unroll
    for (int r = 0; r < 16; ++r) pw[(h * LD + row_of(r, lk)) * TC + ct * 32 + lrow] = reinterpret_cast<const float*>(dwo_l + (ct * 4 + (r >> 2)) * 64)[r & 3];
}

// --------------------------------------------------------------------------------------------------------------- combine
// one workgroup per (frame, head): thread (eg = tid >> 5, d = tid & 31) owns [d][e] for e = eg * 4 .. eg * 4 + 3
__global__ __launch_bounds__(256) void la_bwd_combine_kernel(const LBArgs a) {
  __shared__ float red[8][32];
  __shared__ float dsh[32][33];  // dctx / HW [d][e]
  const int frame = blockIdx.x / LH, h = blockIdx.x - frame * LH;
  const int d = threadIdx.x & 31, eg = threadIdx.x >> 5;
  const int b = frame / a.T, t = frame - b * a.T;
  const int ntok = a.ek ? a.ntok : 0;
  const float* ekb = ntok ? a.ek + ((long long)b * ntok) * HID + h * LD : nullptr;
  const float* evb = ntok ? a.ev + ((long long)b * ntok) * HID + h * LD : nullptr;
  // the forward's key softmax: max, sum, context (as linattn_combine_kernel)
  const float* pbase = a.fpart + ((long long)frame * a.fnsplit * LH + h) * LA_PART;
  const long long pstride = (long long)LH * LA_PART;
  float M = -INFINITY;
  for (int s = 0; s < a.fnsplit; ++s) M = fmaxf(M, pbase[s * pstride + d]);
  for (int j = 0; j < ntok; ++j) M = fmaxf(M, ekb[(long long)j * HID + d]);
  float Z = 0.f, c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < a.fnsplit; ++s) {
    const float* pp = pbase + s * pstride;
    const float f = __expf(pp[d] - M);
    Z += pp[32 + d] * f;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += pp[64 + (eg * 4 + i) * LD + d] * f;
  }
  for (int j = 0; j < ntok; ++j) {
    const float p = __expf(ekb[(long long)j * HID + d] - M);
    Z += p;
#pragma unroll
    for (int i = 0; i < 4; ++i) c[i] += p * evb[(long long)j * HID + eg * 4 + i];
  }
  const float Zinv = 1.0f / Z, hwinv = 1.0f / (float)a.HW;
  // dctx = scale * sum of the slices' partials
  float dc[4] = {0.f, 0.f, 0.f, 0.f};
  const float* qb = a.p1 + ((long long)frame * a.nsplit * LH + h) * 1024 + d * LD + eg * 4;
  for (int s = 0; s < a.nsplit; ++s) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(qb + (long long)s * LH * 1024);
    dc[0] += v.x; dc[1] += v.y; dc[2] += v.z; dc[3] += v.w;
  }
  float ctxn[4], dl = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    dc[i] *= a.q_scale;
    ctxn[i] = c[i] * Zinv * hwinv;
    dl += dc[i] * ctxn[i];
    dsh[d][eg * 4 + i] = dc[i] * hwinv;
  }
  red[eg][d] = dl;
  __syncthreads();
  float delta = 0.f;
#pragma unroll
  for (int k = 0; k < 8; ++k) delta += red[k][d];
  __syncthreads();
  // tables for pass 2
  float* tb = a.tab + ((long long)frame * LH + h) * TAB;
  if (eg == 0) { tb[d] = M; tb[32 + d] = Zinv; tb[64 + d] = delta; }
  unsigned short* CA = reinterpret_cast<unsigned short*>(tb + 96);          // A[i = d][k = e]: scale ctxn
  unsigned short* DA = reinterpret_cast<unsigned short*>(tb + 96 + 1024);   // A[i = d][k = e]: dctx / HW
  unsigned short* DB = reinterpret_cast<unsigned short*>(tb + 96 + 2048);   // A[i = e][k = d]: dctx / HW
  auto put = [](unsigned short* f, int i, int k, float v) {  // element (row i, contraction index k) of an A image in slot order
    const int s = k >> 4, lk = (k >> 2) & 1, j = (k & 3) + 4 * ((k >> 3) & 1);
    unsigned short lo;
    const unsigned short hi = vmm_split16(v, lo);
    f[((s * 2 + 0) * 64 + lk * 32 + i) * 8 + j] = hi;
    f[((s * 2 + 1) * 64 + lk * 32 + i) * 8 + j] = lo;
  };
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int e = eg * 4 + i;
    put(CA, d, e, ctxn[i] * a.q_scale);
    put(DA, d, e, dc[i] * hwinv);
    put(DB, e, d, dc[i] * hwinv);
  }
  // token keys / values: the token columns of dk0 / dv0
  if (ntok) {
    float* pe = a.part_ek + (((long long)t * a.B + b) * ntok) * HID + h * LD;
    float* pv = a.part_ev + (((long long)t * a.B + b) * ntok) * HID + h * LD;
    for (int j = 0; j < ntok; ++j) {
      const float kst = __expf(ekb[(long long)j * HID + d] - M) * Zinv;
      // dks[d][j] = sum_e dctx[d][e] ev[j][e] / HW: this thread's four e, then over the eight groups
      float part = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) part += dc[i] * hwinv * evb[(long long)j * HID + eg * 4 + i];
      red[eg][d] = part;
      __syncthreads();
      if (eg == 0) {
        float dks = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dks += red[k][d];
        pe[(long long)j * HID + d] = kst * (dks - delta);
      }
      __syncthreads();
      // dev[j][e] = sum_d kst[d][j] dctx[d][e] / HW: thread (eg, d) now acts as e = d, partial over d' = eg * 4 .. + 3
      red[eg][d] = 0.f;
      float pe2 = 0.f;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int dd = eg * 4 + i;
        const float ks2 = __expf(ekb[(long long)j * HID + dd] - __shfl(M, dd, 32)) * __shfl(Zinv, dd, 32);
        pe2 += ks2 * dsh[dd][d];
      }
      red[eg][d] = pe2;
      __syncthreads();
      if (eg == 0) {
        float dv = 0.f;
#pragma unroll
        for (int k = 0; k < 8; ++k) dv += red[k][d];
        pv[(long long)j * HID + d] = dv;
      }
      __syncthreads();
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------- pass 2
__global__ __launch_bounds__(512, 2) void la_bwd_rows_kernel(const LBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* tabs = reinterpret_cast<float*>(smem_raw);                                  // [8 heads][TAB]
  unsigned short* ytile = reinterpret_cast<unsigned short*>(tabs + LH * TAB);        // [2][32 * YP]
  unsigned short* gtile = ytile + 2 * 32 * YP;
  const int tid = threadIdx.x, lane = tid & 63;
  const int h = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lrow = lane & 31, lk = lane >> 5;
  const int frame = blockIdx.x / a.nsplit, split = blockIdx.x - frame * a.nsplit;
  const int tiles = a.HW / 32;
  const int t_begin = split * a.sps, t_end = min(tiles, t_begin + a.sps);

  // the frame's tables of all heads: a straight copy
  {
    const f32x4* src = reinterpret_cast<const f32x4*>(a.tab + (long long)frame * LH * TAB);
    f32x4* dst = reinterpret_cast<f32x4*>(tabs);
    for (int i = tid; i < LH * TAB / 4; i += 512) dst[i] = src[i];
  }
  uint4 wq[4][2], wk[4][2], wv[4][2], wo[4][2];
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const uint4* q = a.wqkv + (((long long)h * 4 + s) * 2) * 64 + lane;
    const uint4* k = a.wqkv + (((long long)(LH + h) * 4 + s) * 2) * 64 + lane;
    const uint4* v = a.wqkv + (((long long)(2 * LH + h) * 4 + s) * 2) * 64 + lane;
    const uint4* o = a.woT + (((long long)h * 4 + s) * 2) * 64 + lane;
    wq[s][0] = q[0]; wq[s][1] = q[64];
    wk[s][0] = k[0]; wk[s][1] = k[64];
    wv[s][0] = v[0]; wv[s][1] = v[64];
    wo[s][0] = o[0]; wo[s][1] = o[64];
  }
  const float* tb = tabs + h * TAB;
  const uint4* CA = reinterpret_cast<const uint4*>(tb + 96) + lane;          // + (s * 2 + plane) * 64
  const uint4* DA = reinterpret_cast<const uint4*>(tb + 96 + 1024) + lane;
  const uint4* DB = reinterpret_cast<const uint4*>(tb + 96 + 2048) + lane;

  Stager st;
  st.rm = tid >> 4; st.rcol = (tid & 15) * 4; st.gpos = 0;
  st.gam = *reinterpret_cast<const f32x4*>(a.gamma + st.rcol);
  unsigned x_loff = (unsigned)(st.rm * a.ldx + st.rcol), g_loff = (unsigned)(st.rm * a.ldg + st.rcol);
  auto load_xg = [&](int t, f32x4& xv, f32x4& gv) {
    xv = f32x4{0.f, 0.f, 0.f, 0.f};
    gv = xv;
    if (t < t_end) {
      const long long row0 = (long long)frame * a.HW + t * 32;
      xv = *reinterpret_cast<const f32x4*>(a.x + row0 * a.ldx + x_loff);
      gv = *reinterpret_cast<const f32x4*>(a.gout + row0 * a.ldg + g_loff);
    }
  };
  // T-form matrix X{feature, pixel}: the lane's pixel row gets four 16-byte pieces (features 8 q + 4 lk .. + 3)
  unsigned q_loff = (unsigned)(lrow * a.ldq + 4 * lk);
  auto store_cols = [&](const f32x16& X, int t, int col0) {
    vmm_dqkv_t* gq = a.gqkv + ((long long)frame * a.HW + t * 32) * a.ldq + col0 + h * LD;  // wave-uniform
#pragma unroll
    for (int q4 = 0; q4 < 4; ++q4) st_dqkv4(gq + 8 * q4 + q_loff, X[4 * q4], X[4 * q4 + 1], X[4 * q4 + 2], X[4 * q4 + 3]);
  };

  f32x4 xv, gv;
  load_xg(t_begin, xv, gv);
  stage_tile(a, st, xv, gv, ytile, gtile, nullptr, t_begin < t_end ? a.ln_stats + 2 * ((long long)frame * a.HW + t_begin * 32 + st.rm) : nullptr);
  load_xg(t_begin + 1, xv, gv);
  __syncthreads();
  for (int t = t_begin; t < t_end; ++t) {
    const int buf = (t - t_begin) & 1;
    asm volatile("" : "+v"(x_loff), "+v"(g_loff), "+v"(q_loff));
    const unsigned short* yt = ytile + buf * 32 * YP + lrow * YP + lk * 8;
    const unsigned short* gt = gtile + buf * 32 * YP + lrow * YP + lk * 8;
    // ---- dq0^T = sm (dqs - sum_d sm dqs),  dqs^T {d, n} = (scale ctxn) . dO^T
    {
      f32x16 qT = zero16(), doT = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
        qT = mfma3(wq[s][0], wq[s][1], yh, yl, qT);
        const uint4 gh = *reinterpret_cast<const uint4*>(gt + s * 16), gl = *reinterpret_cast<const uint4*>(gt + s * 16 + TC);
        doT = mfma3(wo[s][0], wo[s][1], gh, gl, doT);
      }
      softmax_rows(qT);
      const F2 dof = tofrag(doT);
      f32x16 dqs = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) dqs = mfma3(CA[(s * 2) * 64], CA[(s * 2 + 1) * 64], dof.h[s], dof.l[s], dqs);
      float tt = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) tt += qT[r] * dqs[r];
      tt += lane_xor(tt, 5);
#pragma unroll
      for (int r = 0; r < 16; ++r) dqs[r] = qT[r] * (dqs[r] - tt);
      store_cols(dqs, t, 0);
    }
    // ---- dk0^T = ks (dks - delta),  dks^T {d, n} = (dctx / HW) . v^T;   dv0^T {e, n} = (dctx / HW)^T . ks^T
    {
      f32x16 kT = zero16(), vT = zero16();
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        const uint4 yh = *reinterpret_cast<const uint4*>(yt + s * 16), yl = *reinterpret_cast<const uint4*>(yt + s * 16 + TC);
        kT = mfma3(wk[s][0], wk[s][1], yh, yl, kT);
        vT = mfma3(wv[s][0], wv[s][1], yh, yl, vT);
      }
      const F2 vf = tofrag(vT);
      f32x16 dks = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) dks = mfma3(DA[(s * 2) * 64], DA[(s * 2 + 1) * 64], vf.h[s], vf.l[s], dks);
#pragma unroll
      for (int q4 = 0; q4 < 4; ++q4) {  // per-feature scalars of registers 4 q .. 4 q + 3: features 8 q + 4 lk .. + 3
        const f32x4 M4 = *reinterpret_cast<const f32x4*>(tb + 8 * q4 + 4 * lk);
        const f32x4 Z4 = *reinterpret_cast<const f32x4*>(tb + 32 + 8 * q4 + 4 * lk);
        const f32x4 D4 = *reinterpret_cast<const f32x4*>(tb + 64 + 8 * q4 + 4 * lk);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const float ks = __expf(kT[4 * q4 + i] - M4[i]) * Z4[i];
          kT[4 * q4 + i] = ks;
          dks[4 * q4 + i] = ks * (dks[4 * q4 + i] - D4[i]);
        }
      }
      store_cols(dks, t, HID);
      const F2 ksf = tofrag(kT);
      f32x16 dv = zero16();
#pragma unroll
      for (int s = 0; s < 2; ++s) dv = mfma3(DB[(s * 2) * 64], DB[(s * 2 + 1) * 64], ksf.h[s], ksf.l[s], dv);
      store_cols(dv, t, 2 * HID);
    }
    stage_tile(a, st, xv, gv, ytile + (buf ^ 1) * 32 * YP, gtile + (buf ^ 1) * 32 * YP, nullptr,
               t + 1 < t_end ? a.ln_stats + 2 * ((long long)frame * a.HW + (t + 1) * 32 + st.rm) : nullptr);
    load_xg(t + 2, xv, gv);
    __syncthreads();
  }
}

int choose_bwd_split(int frames, int HW, int* sps) {
  const int tiles = HW / 32;
  int ns = max(1, min(tiles / 4, 512 / max(frames, 1)));
  *sps = (tiles + ns - 1) / ns;
  return (tiles + *sps - 1) / *sps;
}

bool supported(int HW, int C, int heads, int ntok) { return C == TC && heads == LH && HW > 0 && HW % 32 == 0 && ntok >= 0; }

}  // namespace

// floats of workspace vmm_linattn_block_bwd_bf16x3 needs; 0 outside its envelope (C == 64, heads == 8, dim_head == 32, HW % 32 == 0)
#if !VMM_SINGLE_PASS
extern "C" int64_t vmm_linattn_block_bwd_workspace(int32_t B, int32_t T, int32_t HW, int32_t C, int32_t heads, int32_t ntok) {
  if (!supported(HW, C, heads, ntok) || B <= 0 || T <= 0) return 0;
  int sps;
  const int ns = choose_bwd_split(B * T, HW, &sps);
  const long long frames = (long long)B * T;
  return frames * ns * LH * 1024 + frames * ns * (HID * TC) + frames * LH * TAB + 2LL * T * B * ntok * HID;
}

#endif
extern "C" int VMM_X3(vmm_linattn_block_bwd_, )(const vmm_attn_block_bwd* d, vmm_stream_t stream) {
  const int ntok = d->ek ? d->ntok : 0;
  if (!supported(d->HW, d->C, d->heads, ntok) || (d->ldx & 3) || (d->lddo & 3) || (d->lddqkv & 3) || !d->workspace || !d->fwd_workspace) return 1;
  if (d->B <= 0 || d->T <= 0) return 0;
  hipStream_t s = (hipStream_t)stream;
  const int frames = d->B * d->T;
  LBArgs a;
  a.x = d->x; a.ldx = d->ldx; a.gamma = d->gamma; a.eps = d->eps;
  a.wqkv = reinterpret_cast<const uint4*>(d->wqkv_frag);
  a.woT = reinterpret_cast<const uint4*>(d->wout_t_frag);
  a.ek = d->ek; a.ev = d->ev; a.ntok = ntok;
  int fsps;
  a.fnsplit = vmm_linattn_block_split(frames, d->HW, &fsps);
  a.fpart = d->fwd_workspace;
  a.fctx = reinterpret_cast<const uint4*>(d->fwd_workspace + (long long)frames * a.fnsplit * LH * LA_PART);
  a.gout = d->dout; a.ldg = d->lddo;
  a.gqkv = reinterpret_cast<vmm_dqkv_t*>(d->dqkv); a.ldq = d->lddqkv; a.ln_stats = d->ln_stats;
  a.B = d->B; a.T = d->T; a.HW = d->HW;
  a.nsplit = choose_bwd_split(frames, d->HW, &a.sps);
  a.q_scale = d->q_scale;
  a.p1 = d->workspace;
  a.part_wo = a.p1 + (long long)frames * a.nsplit * LH * 1024;
  a.tab = a.part_wo + (long long)frames * a.nsplit * (HID * TC);
  a.part_ek = a.tab + (long long)frames * LH * TAB;
  a.part_ev = a.part_ek + (long long)d->T * d->B * ntok * HID;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&la_bwd_ctx_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&la_bwd_rows_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const unsigned blocks = (unsigned)(frames * a.nsplit);
  const size_t shm1 = sizeof(f32x4) * LH * 8 * 64 + sizeof(unsigned short) * (4 * 32 * YP + 2 * 2 * 64 * GP);
  hipLaunchKernelGGL(la_bwd_ctx_kernel, dim3(blocks), dim3(512), shm1, s, a);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(la_bwd_combine_kernel, dim3((unsigned)(frames * LH)), dim3(256), 0, s, a);
  VMM_LAUNCH_CHECK();
  const size_t shm2 = sizeof(float) * LH * TAB + sizeof(unsigned short) * (4 * 32 * YP);
  hipLaunchKernelGGL(la_bwd_rows_kernel, dim3(blocks), dim3(512), shm2, s, a);
  VMM_LAUNCH_CHECK();
  int rc = vmm_sum_partials(a.part_wo, (int)blocks, HID * TC, HID * TC, d->dwout_packed, stream);
  if (rc) return rc;
  if (d->dbout) {
    rc = vmm_colsum_accumulate(d->dout, d->lddo, (int64_t)frames * d->HW, TC, d->dbout, stream);
    if (rc) return rc;
  }
  if (ntok) {
    const int n = d->B * ntok * HID;
    if (d->dek) { rc = vmm_sum_partials(a.part_ek, d->T, n, n, d->dek, stream); if (rc) return rc; }
    if (d->dev) { rc = vmm_sum_partials(a.part_ev, d->T, n, n, d->dev, stream); if (rc) return rc; }
  }
  return 0;
}
