// Backward of the attention cores (training path), gfx950.  dim_head = 32 for the mid spatial and the linear attention (vddp.py:314, 401, 679, 687);
// the generic softmax passes A / B are templated on the head slice (attn_dim_head of the temporal attentions, vddp.py:615).
//
// Softmax attention (temporal: keys = frames of one pixel; mid spatial: keys = pixels of one frame), flash-style:
//   forward saved  L = logsumexp per (query row, head) and O;   D = dO . O
//   pass A (thread per query):  ds_ij = p_ij (dO_i.v_j - D_i),  dq_i = sum_j ds_ij k_j
//   pass B (thread per key):    dk_j = sum_i ds_ij q_i,  dv_j = sum_i p_ij dO_i,  dbias[h,i,j] = sum ds_ij
// The projection epilogue's q-scale and rotary rotation are undone here (R^T and * scale), so the kernels emit the
// gradient of the raw to_qkv output; gradients of the conditioning keys/values (shared by every pixel of a sample) are
// reduced inside the wave before they touch memory (64 atomics per wave-group).
//
// Linear attention: see the formulas next to each kernel.
#include "vmm_common.h"
#include "head_vec.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32;

__device__ __forceinline__ void ld32(float (&dst)[DH], const float* src) {
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
    dst[i * 4] = v.x; dst[i * 4 + 1] = v.y; dst[i * 4 + 2] = v.z; dst[i * 4 + 3] = v.w;
  }
}
__device__ __forceinline__ void st32(float* dst, const float (&src)[DH]) {
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) *reinterpret_cast<f32x4*>(dst + i * 4) = (f32x4){src[i * 4], src[i * 4 + 1], src[i * 4 + 2], src[i * 4 + 3]};
}
__device__ __forceinline__ float dot32r(const float (&a)[DH], const float (&b)[DH]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int i = 0; i < DH; i += 4) { s0 = fmaf(a[i], b[i], s0); s1 = fmaf(a[i + 1], b[i + 1], s1); s2 = fmaf(a[i + 2], b[i + 2], s2); s3 = fmaf(a[i + 3], b[i + 3], s3); }
  return (s0 + s1) + (s2 + s3);
}
struct AttnGeom {
  int mode;  // 0 temporal (batch = (b, pix), n = T, stride = HW), 1 spatial (batch = (b, t), n = HW, stride = 1)
  int B, T, HW, heads, ntok, tok_per_frame, bias_on_cond;
};
__device__ __forceinline__ long long geom_row0(const AttnGeom& g, int b, int inner) {
  return g.mode == 0 ? (long long)b * g.T * g.HW + inner : ((long long)b * g.T + inner) * g.HW;
}

// ---------------------------------------------------------------- pass A: thread per (batch element, head, query)
// (templated on the head slice, head_vec.h: the temporal attentions follow attn_dim_head, vddp.py:582, 615; dh = 32 is the shipped instance)
template <int DM, bool EX>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnGeom g, int dh_, const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                         const float* __restrict__ ev, const float* __restrict__ bias,
                                                         const float* __restrict__ O, const float* __restrict__ dO, int ldo,
                                                         const float* __restrict__ lse, const float* __restrict__ rot, float q_scale,
                                                         float* __restrict__ dqkv, float* __restrict__ Dbuf) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  const int n = g.mode == 0 ? g.T : g.HW;
  const int ninner = g.mode == 0 ? g.HW : g.T;
  const long long total = (long long)g.B * ninner * g.heads * n;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= total) return;
  const int i = (int)(gid % n);
  const int head = (int)((gid / n) % g.heads);
  const long long bi = gid / ((long long)n * g.heads);
  const int inner = (int)(bi % ninner), b = (int)(bi / ninner);
  const long long row0 = geom_row0(g, b, inner);
  const long long stride = g.mode == 0 ? g.HW : 1;
  const long long rq = row0 + i * stride;
  const int hid = g.heads * dh;
  float q[DM], go[DM], dq[DM], tmp[DM];
  HV::ld(q, qkv + rq * ldqkv + head * dh, dh);
  HV::ld(go, dO + rq * ldo + head * dh, dh);
  HV::ld(tmp, O + rq * ldo + head * dh, dh);
  const float Dv = HV::dotr(go, tmp);
  const float L = lse[rq * g.heads + head];
  Dbuf[rq * g.heads + head] = Dv;
  HV::zero(dq);
  const float* brow = bias ? bias + ((long long)head * n + i) * n : nullptr;
  if (ek) {
    const int t = g.mode == 1 ? inner : 0;
    const int j0 = g.tok_per_frame ? t : 0, j1 = g.tok_per_frame ? t + 1 : g.ntok;
    for (int j = j0; j < j1; ++j) {
      HV::ld(tmp, ek + ((long long)b * g.ntok + j) * hid + head * dh, dh);
      float s = HV::dotr(q, tmp);
      if (brow && g.bias_on_cond) s += brow[j];
      const float p = __expf(s - L);
      float vv[DM];
      HV::ld(vv, ev + ((long long)b * g.ntok + j) * hid + head * dh, dh);
      const float ds = p * (HV::dotr(go, vv) - Dv);
#pragma unroll
      for (int d = 0; d < DM; ++d) dq[d] = fmaf(ds, tmp[d], dq[d]);
    }
  }
  for (int j = 0; j < n; ++j) {
    const float* r = qkv + (row0 + j * stride) * ldqkv + head * dh;
    HV::ld(tmp, r + hid, dh);
    float s = HV::dotr(q, tmp);
    if (brow) s += brow[j];
    const float p = __expf(s - L);
    float vv[DM];
    HV::ld(vv, r + 2 * hid, dh);
    const float ds = p * (HV::dotr(go, vv) - Dv);
#pragma unroll
    for (int d = 0; d < DM; ++d) dq[d] = fmaf(ds, tmp[d], dq[d]);
  }
  if (rot) HV::unrotate(dq, rot, i, dh);
#pragma unroll
  for (int d = 0; d < DM; ++d) dq[d] *= q_scale;
  HV::st(dqkv + rq * ldqkv + head * dh, dq, dh);
}

// ---------------------------------------------------------------- pass B: wave per (b, head, key j, group of 64-lane chunks of the inner index)
constexpr int CHUNKS_PER_WAVE = 8;
template <int DM, bool EX>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnGeom g, int dh_, const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                          const float* __restrict__ ev, const float* __restrict__ bias,
                                                          const float* __restrict__ dO, int ldo, const float* __restrict__ lse,
                                                          const float* __restrict__ Dbuf, const float* __restrict__ rot,
                                                          float* __restrict__ dqkv, float* __restrict__ dek, float* __restrict__ dev,
                                                          float* __restrict__ dbias) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  const int n = g.mode == 0 ? g.T : g.HW;
  const int ninner = g.mode == 0 ? g.HW : g.T;
  const int nchunks = (ninner + 63) / 64;
  const int ngroups = (nchunks + CHUNKS_PER_WAVE - 1) / CHUNKS_PER_WAVE;
  const int nkeys = g.ntok + n;
  const long long wave_id = ((long long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  const long long nwaves = (long long)g.B * g.heads * nkeys * ngroups;
  if (wave_id >= nwaves) return;
  const int grp = (int)(wave_id % ngroups);
  const int j = (int)((wave_id / ngroups) % nkeys);
  const int head = (int)((wave_id / ((long long)ngroups * nkeys)) % g.heads);
  const int b = (int)(wave_id / ((long long)ngroups * nkeys * g.heads));
  const bool is_tok = j < g.ntok;
  const int jj = j - g.ntok;  // frame/pixel key index
  const int hid = g.heads * dh;
  const long long stride = g.mode == 0 ? g.HW : 1;
  float kacc[DM], vacc[DM];  // token-key accumulators over chunks (per lane partial)
  HV::zero(kacc);
  HV::zero(vacc);
  float bias_acc = 0.f;  // lane i (< n, temporal only) accumulates dbias[h, i, j]
  for (int c = grp * CHUNKS_PER_WAVE; c < min((grp + 1) * CHUNKS_PER_WAVE, nchunks); ++c) {
    const int inner = c * 64 + lane;
    bool active = inner < ninner;
    if (is_tok && g.tok_per_frame && active) active = (g.mode == 1) ? (inner == j) : true;
    const long long row0 = geom_row0(g, b, active ? inner : 0);
    float kk[DM], vv[DM], dk[DM], dv[DM];
    if (is_tok) {
      HV::ld(kk, ek + ((long long)b * g.ntok + j) * hid + head * dh, dh);
      HV::ld(vv, ev + ((long long)b * g.ntok + j) * hid + head * dh, dh);
    } else {
      const float* r = qkv + (row0 + jj * stride) * ldqkv + head * dh;
      HV::ld(kk, r + hid, dh);
      HV::ld(vv, r + 2 * hid, dh);
    }
    HV::zero(dk);
    HV::zero(dv);
    const bool use_bias = bias && (!is_tok || g.bias_on_cond);
    const int jb = is_tok ? j : jj;
    for (int i = 0; i < n; ++i) {
      const long long rq = row0 + i * stride;
      float q[DM], go[DM];
      HV::ld(q, qkv + rq * ldqkv + head * dh, dh);
      HV::ld(go, dO + rq * ldo + head * dh, dh);
      float s = HV::dotr(q, kk);
      if (use_bias) s += bias[((long long)head * n + i) * n + jb];
      const float p = active ? __expf(s - lse[rq * g.heads + head]) : 0.f;
      const float ds = p * (HV::dotr(go, vv) - Dbuf[rq * g.heads + head]);
#pragma unroll
      for (int d = 0; d < DM; ++d) { dk[d] = fmaf(ds, q[d], dk[d]); dv[d] = fmaf(p, go[d], dv[d]); }
      if (use_bias && dbias) {
        const float tot = wave_sum(ds);
        if (lane == (i & 63)) bias_acc += tot;  // n <= 64 whenever a bias exists (temporal: n = T)
      }
    }
    if (is_tok) {
#pragma unroll
      for (int d = 0; d < DM; ++d) { kacc[d] += dk[d]; vacc[d] += dv[d]; }
    } else if (active) {
      if (rot) HV::unrotate(dk, rot, jj, dh);
      float* o = dqkv + (row0 + jj * stride) * ldqkv + head * dh;
      HV::st(o + hid, dk, dh);
      HV::st(o + 2 * hid, dv, dh);
    }
  }
  if (is_tok) {
    // rotary on the token keys is undone by the caller (vmm_rotary_rows with the transposed table) -- gradients here are
    // with respect to the rotated ek that the forward kernels consumed.  Lanes 0..31 carry 32 columns of dek, lanes 32..63 of dev.
#pragma unroll
    for (int d0 = 0; d0 < DM; d0 += 32) {
      if (!EX && d0 >= dh) break;
      float mine = 0.f;
#pragma unroll
      for (int d = d0; d < (d0 + 32 < DM ? d0 + 32 : DM); ++d) {
        const float tk = wave_sum(kacc[d]);
        const float tv = wave_sum(vacc[d]);
        if (lane == d - d0) mine = tk;
        if (lane == 32 + d - d0) mine = tv;
      }
      const int col = d0 + (lane & 31);
      if (col < dh) atomicAdd((lane < 32 ? dek : dev) + ((long long)b * g.ntok + j) * hid + head * dh + col, mine);
    }
  }
  if (bias && dbias && (!is_tok || g.bias_on_cond) && lane < n) atomicAdd(&dbias[((long long)head * n + lane) * n + (is_tok ? j : jj)], bias_acc);
}

// ---------------------------------------------------------------- pass A for the mid spatial attention: the frame's keys and values staged in LDS
// The thread-per-query kernel above walks its keys through two dependent global loads each (k_j, then v_j): with 144 keys and about one wave per
// SIMD that chain of 288 L2 round trips was the whole 150 us of the launch.  Same arithmetic in the same order (bit-identical results), the
// key / value rows of the (sample, frame, head) read from LDS (uniform address: broadcast).
__global__ __launch_bounds__(256) void spatial_attn_bwd_q_kernel(AttnGeom g, const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                                 const float* __restrict__ ev, const float* __restrict__ O,
                                                                 const float* __restrict__ dO, int ldo, const float* __restrict__ lse, float q_scale,
                                                                 float* __restrict__ dqkv, float* __restrict__ Dbuf) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // K[HW][32] | V[HW][32]
  const int HW = g.HW, hid = g.heads * DH;
  float* Ks = sm;
  float* Vs = Ks + HW * DH;
  const int tid = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  const int bt = blockIdx.x / g.heads, t = bt % g.T, b = bt / g.T;
  const long long row0 = (long long)bt * HW;
  for (int e = tid; e < HW * (DH / 4); e += 256) {
    const int r = e >> 3, c = (e & 7) * 4;
    *reinterpret_cast<f32x4*>(Ks + r * DH + c) = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ldqkv + hid + head * DH + c);
    *reinterpret_cast<f32x4*>(Vs + r * DH + c) = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ldqkv + 2 * hid + head * DH + c);
  }
  __syncthreads();
  if (tid >= HW) return;
  const long long rq = row0 + tid;
  float q[DH], go[DH], dq[DH], tmp[DH];
  ld32(q, qkv + rq * ldqkv + head * DH);
  ld32(go, dO + rq * ldo + head * DH);
  ld32(tmp, O + rq * ldo + head * DH);
  const float Dv = dot32r(go, tmp);
  const float L = lse[rq * g.heads + head];
  Dbuf[rq * g.heads + head] = Dv;
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[d] = 0.f;
  if (ek) {
    const int j0 = g.tok_per_frame ? t : 0, j1 = g.tok_per_frame ? t + 1 : g.ntok;
    for (int j = j0; j < j1; ++j) {
      ld32(tmp, ek + ((long long)b * g.ntok + j) * hid + head * DH);
      const float p = __expf(dot32r(q, tmp) - L);
      float vv[DH];
      ld32(vv, ev + ((long long)b * g.ntok + j) * hid + head * DH);
      const float ds = p * (dot32r(go, vv) - Dv);
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, tmp[d], dq[d]);
    }
  }
  for (int j = 0; j < HW; ++j) {
    ld32(tmp, Ks + j * DH);
    const float p = __expf(dot32r(q, tmp) - L);
    float vv[DH];
    ld32(vv, Vs + j * DH);
    const float ds = p * (dot32r(go, vv) - Dv);
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, tmp[d], dq[d]);
  }
#pragma unroll
  for (int d = 0; d < DH; ++d) dq[d] *= q_scale;
  st32(dqkv + rq * ldqkv + head * DH, dq);
}

// ---------------------------------------------------------------- pass B for the mid spatial attention (mode 1, HW <= 256 keys per frame)
// The generic pass B maps lanes to the inner index, which here is the frame (T = 11 of 64 lanes busy, one wave per key).  This
// one takes a workgroup per (sample, frame, head): q, dO, logsumexp and D of the frame's HW queries are staged once in LDS, thread j
// = key j (thread HW = the frame's conditioning token) sweeps the queries with its k_j / v_j in registers.
__global__ __launch_bounds__(256) void spatial_attn_bwd_kv_kernel(AttnGeom g, const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                                  const float* __restrict__ ev, const float* __restrict__ dO, int ldo,
                                                                  const float* __restrict__ lse, const float* __restrict__ Dbuf,
                                                                  float* __restrict__ dqkv, float* __restrict__ dek, float* __restrict__ dev) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // Q[HW][32] | G[HW][32] | L[HW] | D[HW]
  const int HW = g.HW, hid = g.heads * DH;
  float* Qs = sm;
  float* Gs = Qs + HW * DH;
  float* Ls = Gs + HW * DH;
  float* Dsh = Ls + HW;
  const int tid = threadIdx.x;
  const int head = blockIdx.x % g.heads;
  const int bt = blockIdx.x / g.heads, t = bt % g.T, b = bt / g.T;
  const long long row0 = (long long)bt * HW;
  for (int e = tid; e < HW * (DH / 4); e += 256) {
    const int r = e >> 3, c = (e & 7) * 4;
    *reinterpret_cast<f32x4*>(Qs + r * DH + c) = *reinterpret_cast<const f32x4*>(qkv + (row0 + r) * ldqkv + head * DH + c);
    *reinterpret_cast<f32x4*>(Gs + r * DH + c) = *reinterpret_cast<const f32x4*>(dO + (row0 + r) * ldo + head * DH + c);
  }
  for (int r = tid; r < HW; r += 256) {
    Ls[r] = lse[(row0 + r) * g.heads + head];
    Dsh[r] = Dbuf[(row0 + r) * g.heads + head];
  }
  __syncthreads();
  const bool is_tok = tid == HW && g.ntok > 0;
  if (tid >= HW && !is_tok) return;
  float kk[DH], vv[DH], dk[DH], dv[DH];
  if (is_tok) {
    ld32(kk, ek + ((long long)b * g.ntok + t) * hid + head * DH);
    ld32(vv, ev + ((long long)b * g.ntok + t) * hid + head * DH);
  } else {
    ld32(kk, qkv + (row0 + tid) * ldqkv + hid + head * DH);
    ld32(vv, qkv + (row0 + tid) * ldqkv + 2 * hid + head * DH);
  }
#pragma unroll
  for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
  for (int i = 0; i < HW; ++i) {
    float q[DH], go[DH];
    ld32(q, Qs + i * DH);  // (uniform address: LDS broadcast)
    ld32(go, Gs + i * DH);
    const float p = __expf(dot32r(q, kk) - Ls[i]);
    const float ds = p * (dot32r(go, vv) - Dsh[i]);
#pragma unroll
    for (int d = 0; d < DH; ++d) { dk[d] = fmaf(ds, q[d], dk[d]); dv[d] = fmaf(p, go[d], dv[d]); }
  }
  if (is_tok) {  // one writer per (sample, frame, head)
    float* kd = dek + ((long long)b * g.ntok + t) * hid + head * DH;
    float* vd = dev + ((long long)b * g.ntok + t) * hid + head * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { kd[d] += dk[d]; vd[d] += dv[d]; }
  } else {
    float* o = dqkv + (row0 + tid) * ldqkv + head * DH;
    st32(o + hid, dk);
    st32(o + 2 * hid, dv);
  }
}

// ================================================================ linear attention backward
// forward: kt = softmax_n(k), ctx[d,e] = sum_n kt[d,n] v[e,n] / HW, qt = softmax_d(q)*scale, out[n,e] = sum_d ctx[d,e] qt[n,d]
// dctx[d,e] = sum_n qt[n,d] dout[n,e]                                       (kernel 1, reduction over rows, atomics)
// g[d] = sum_e ctx[d,e] dout[n,e];  dq[n,d] = scale * p[d] (g[d] - sum_d p g),  p = softmax_d(q[n,:])     (kernel 2)
// dv[e,n] = sum_d kt[d,n] dctx[d,e]/HW;  dkt[d,n] = sum_e dctx[d,e] v[e,n]/HW;  dk[d,n] = kt[d,n] (dkt[d,n] - R[d]),
// R[d] = sum_e dctx[d,e] ctx[d,e]                                           (kernel 2; tokens: kernel 3)
constexpr int LB_TILE = 64;
__global__ __launch_bounds__(256) void linattn_bwd_dctx_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ dout,
                                                               int lddo, int HW, int heads, int rows_per_split, float scale,
                                                               float* __restrict__ dctx) {
  __shared__ __attribute__((aligned(16))) float Qs[LB_TILE][DH + 1];
  __shared__ __attribute__((aligned(16))) float Gs[LB_TILE][DH];
  const int tid = threadIdx.x;
  const int fh = blockIdx.y, head = fh % heads;
  const long long frame = fh / heads;
  const int d = tid >> 3, e0 = (tid & 7) * 4;
  const int n_begin = blockIdx.x * rows_per_split, n_end = min(n_begin + rows_per_split, HW);
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = n_begin; n0 < n_end; n0 += LB_TILE) {
    const int nrows = min(LB_TILE, n_end - n0);
    __syncthreads();
    for (int e = tid; e < LB_TILE * (DH / 4); e += 256) {
      const int r = e / (DH / 4), cc = (e % (DH / 4)) * 4;
      f32x4 qv = {0.f, 0.f, 0.f, 0.f}, gv = {0.f, 0.f, 0.f, 0.f};
      if (r < nrows) {
        qv = *reinterpret_cast<const f32x4*>(qkv + (frame * HW + n0 + r) * ldqkv + head * DH + cc);
        gv = *reinterpret_cast<const f32x4*>(dout + (frame * HW + n0 + r) * lddo + head * DH + cc);
      }
      Qs[r][cc] = qv.x; Qs[r][cc + 1] = qv.y; Qs[r][cc + 2] = qv.z; Qs[r][cc + 3] = qv.w;
      *reinterpret_cast<f32x4*>(&Gs[r][cc]) = gv;
    }
    __syncthreads();
    if (tid < LB_TILE) {  // row softmax over d, scaled, in place
      float mx = Qs[tid][0];
      for (int k = 1; k < DH; ++k) mx = fmaxf(mx, Qs[tid][k]);
      float sum = 0.f;
      for (int k = 0; k < DH; ++k) { const float ex = __expf(Qs[tid][k] - mx); Qs[tid][k] = ex; sum += ex; }
      const float sc = (tid < nrows) ? scale / sum : 0.f;
      for (int k = 0; k < DH; ++k) Qs[tid][k] *= sc;
    }
    __syncthreads();
    for (int r = 0; r < nrows; ++r) {
      const float qd = Qs[r][d];
      const f32x4 gv = *reinterpret_cast<const f32x4*>(&Gs[r][e0]);
      c[0] = fmaf(qd, gv.x, c[0]); c[1] = fmaf(qd, gv.y, c[1]); c[2] = fmaf(qd, gv.z, c[2]); c[3] = fmaf(qd, gv.w, c[3]);
    }
  }
  float* o = dctx + (long long)fh * DH * DH + d * DH + e0;
  atomicAdd(o, c[0]); atomicAdd(o + 1, c[1]); atomicAdd(o + 2, c[2]); atomicAdd(o + 3, c[3]);
}

// thread per (row, head); ctx / dctx of a group of `hpb` heads of the frame in LDS (blockIdx.z = head group)
__global__ __launch_bounds__(256) void linattn_bwd_rows_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ dout,
                                                               int lddo, const float* __restrict__ ctx, const float* __restrict__ dctx,
                                                               const float* __restrict__ kstat, int HW, int heads, int hpb, float scale,
                                                               float* __restrict__ dqkv) {
  extern __shared__ __attribute__((aligned(16))) float sm[];  // ctx[hpb][CSTR] | dctx[hpb][CSTR] | R[hpb][32]
  constexpr int CSTR = DH * DH + 4;
  float* cs = sm;
  float* ds_ = sm + hpb * CSTR;
  float* Rs = ds_ + hpb * CSTR;
  const int tid = threadIdx.x;
  const long long frame = blockIdx.y;
  const int h0 = blockIdx.z * hpb;
  for (int e = tid; e < hpb * DH * DH / 4; e += 256) {
    const int h = e / (DH * DH / 4), r = e % (DH * DH / 4);
    *reinterpret_cast<f32x4*>(&cs[h * CSTR + r * 4]) = *reinterpret_cast<const f32x4*>(ctx + (frame * heads + h0 + h) * DH * DH + r * 4);
    *reinterpret_cast<f32x4*>(&ds_[h * CSTR + r * 4]) = *reinterpret_cast<const f32x4*>(dctx + (frame * heads + h0 + h) * DH * DH + r * 4);
  }
  __syncthreads();
  for (int e = tid; e < hpb * DH; e += 256) {
    const int h = e / DH, d = e % DH;
    float r = 0.f;
    for (int k = 0; k < DH; ++k) r = fmaf(ds_[h * CSTR + d * DH + k], cs[h * CSTR + d * DH + k], r);
    Rs[e] = r;
  }
  __syncthreads();
  const int rows_per_block = 256 / hpb;
  const int hl = tid % hpb;
  const int head = h0 + hl;
  const int n = blockIdx.x * rows_per_block + tid / hpb;
  if (n >= HW || tid / hpb >= rows_per_block) return;
  const int hid = heads * DH;
  const float* src = qkv + (frame * HW + n) * ldqkv + head * DH;
  float* dst = dqkv + (frame * HW + n) * ldqkv + head * DH;
  const float* ch = cs + hl * CSTR;
  const float* dh = ds_ + hl * CSTR;
  float go[DH], x[DH], o[DH];
  ld32(go, dout + (frame * HW + n) * lddo + head * DH);
  // ---- q
  ld32(x, src);
  float mx = x[0];
#pragma unroll
  for (int d = 1; d < DH; ++d) mx = fmaxf(mx, x[d]);
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) { x[d] = __expf(x[d] - mx); sum += x[d]; }
  const float inv = 1.0f / sum;
  float pg = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    float gd = 0.f;
#pragma unroll
    for (int e = 0; e < DH; ++e) gd = fmaf(ch[d * DH + e], go[e], gd);
    x[d] *= inv;  // p[d]
    o[d] = gd;
    pg = fmaf(x[d], gd, pg);
  }
#pragma unroll
  for (int d = 0; d < DH; ++d) o[d] = scale * x[d] * (o[d] - pg);
  st32(dst, o);
  // ---- k, v
  const float invHW = 1.0f / (float)HW;
  float kt[DH], vv[DH];
  ld32(kt, src + hid);
  ld32(vv, src + 2 * hid);
  const float* ks = kstat + ((frame * heads + head) * 2) * DH;  // max | 1/sum
#pragma unroll
  for (int d = 0; d < DH; ++d) kt[d] = __expf(kt[d] - ks[d]) * ks[DH + d];
  float dvv[DH];
#pragma unroll
  for (int e = 0; e < DH; ++e) dvv[e] = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    float dkt = 0.f;
#pragma unroll
    for (int e = 0; e < DH; ++e) {
      const float dc = dh[d * DH + e];
      dkt = fmaf(dc, vv[e], dkt);
      dvv[e] = fmaf(kt[d], dc, dvv[e]);
    }
    o[d] = kt[d] * (dkt * invHW - Rs[hl * DH + d]);
  }
#pragma unroll
  for (int e = 0; e < DH; ++e) dvv[e] *= invHW;
  st32(dst + hid, o);
  st32(dst + 2 * hid, dvv);
}

// tokens: thread per (frame*heads + head, token j, d): dek[b,j,h,d] += kt (dkt - R), dev[b,j,h,e] += sum_d kt dctx / HW
__global__ void linattn_bwd_tokens_kernel(const float* __restrict__ ek, const float* __restrict__ ev, int ntok, const float* __restrict__ ctx,
                                          const float* __restrict__ dctx, const float* __restrict__ kstat, int T, int HW, int heads, int nfh,
                                          float* __restrict__ dek, float* __restrict__ dev) {
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (gid >= (long long)nfh * ntok * DH) return;
  const int d = (int)(gid % DH);
  const int j = (int)((gid / DH) % ntok);
  const int fh = (int)(gid / ((long long)DH * ntok));
  const int head = fh % heads, b = (fh / heads) / T;
  const int hid = heads * DH;
  const float* c = ctx + (long long)fh * DH * DH;
  const float* dc = dctx + (long long)fh * DH * DH;
  const float* ks = kstat + (long long)fh * 2 * DH;
  const float* kr = ek + ((long long)b * ntok + j) * hid + head * DH;
  const float* vr = ev + ((long long)b * ntok + j) * hid + head * DH;
  const float invHW = 1.0f / (float)HW;
  // dk for channel d
  float R = 0.f, dkt = 0.f;
  for (int e = 0; e < DH; ++e) { R = fmaf(dc[d * DH + e], c[d * DH + e], R); dkt = fmaf(dc[d * DH + e], vr[e], dkt); }
  const float kt = __expf(kr[d] - ks[d]) * ks[DH + d];
  atomicAdd(&dek[((long long)b * ntok + j) * hid + head * DH + d], kt * (dkt * invHW - R));
  // dv for channel e = d
  float acc = 0.f;
  for (int dd = 0; dd < DH; ++dd) acc = fmaf(__expf(kr[dd] - ks[dd]) * ks[DH + dd], dc[dd * DH + d], acc);
  atomicAdd(&dev[((long long)b * ntok + j) * hid + head * DH + d], acc * invHW);
}

// cond_attention = 'cross-attention': only the queries are rows.  thread per (row, head): g[d] = sum_e ctx[d,e] dout[n,e],
// dq[n,d] = scale p[d] (g[d] - sum_d' p g), p = softmax_d(q[n,:]); ctx of the frame's heads in LDS as [head][d][33]
__global__ __launch_bounds__(256) void linattn_bwd_q_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ dout, int lddo,
                                                            const float* __restrict__ ctx, int HW, int heads, float scale, float* __restrict__ dq,
                                                            int lddq) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x;
  const long long frame = blockIdx.y;
  for (int i = tid; i < heads * DH * DH; i += 256) sm[(i / DH) * (DH + 1) + i % DH] = ctx[frame * heads * DH * DH + i];
  __syncthreads();
  const int rows_per_block = 256 / heads;
  const int head = tid % heads, n = blockIdx.x * rows_per_block + tid / heads;
  if (n >= HW || tid / heads >= rows_per_block) return;
  const long long row = frame * HW + n;
  float qv[DH], go[DH];
  ld32(qv, q + row * ldq + head * DH);
  ld32(go, dout + row * lddo + head * DH);
  float mx = qv[0];
#pragma unroll
  for (int d = 1; d < DH; ++d) mx = fmaxf(mx, qv[d]);
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) { qv[d] = __expf(qv[d] - mx); sum += qv[d]; }
  const float inv = 1.0f / sum;
  float g[DH], pg = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) {
    const float* c = sm + (head * DH + d) * (DH + 1);
    float a = 0.f;
#pragma unroll
    for (int e = 0; e < DH; ++e) a = fmaf(c[e], go[e], a);
    g[d] = a;
    qv[d] *= inv;
    pg = fmaf(qv[d], a, pg);
  }
#pragma unroll
  for (int d = 0; d < DH; ++d) g[d] = scale * qv[d] * (g[d] - pg);
  st32(dq + row * lddq + head * DH, g);
}

}  // namespace

// Backward of the linear cross-attention core (vddp.py:354-363; forward = vmm_linattn_cross_context + vmm_linattn_apply on the q rows): dq
// [rows][heads*32] is written, the token gradients are ADDED into dek / dev [B][ntok][heads*32].  ctx / kstat as the forward left them
// ([B*T*heads][32*32], [B*T*heads][2][32]); dctx: [B*T*heads][32*32] scratch.
extern "C" int vmm_linattn_cross_bwd(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* ctx, const float* kstat,
                                     const float* dout, int32_t lddo, float* dctx, float* dq, int32_t lddq, float* dek, float* dev, int32_t B, int32_t T,
                                     int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (dh != DH || (ldq & 3) || (lddo & 3) || (lddq & 3) || !ek || !ev || ntok < 1 || !kstat || heads < 1 || heads > 32) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nfh = B * T * heads;
  if (int rc = vmm_zero_async(dctx, sizeof(float) * nfh * DH * DH, s)) return rc;
  const float scale = 0.17677669529663687f;
  const int nsplit = max(1, min((HW + LB_TILE - 1) / LB_TILE, cdiv(2048, nfh)));
  const int rps = cdiv(cdiv(HW, nsplit), LB_TILE) * LB_TILE;
  hipLaunchKernelGGL(linattn_bwd_dctx_kernel, dim3(cdiv(HW, rps), nfh), dim3(256), 0, s, q, ldq, dout, lddo, HW, heads, rps, scale, dctx);
  VMM_LAUNCH_CHECK();
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(linattn_bwd_q_kernel, dim3(cdiv(HW, 256 / heads), B * T), dim3(256), sizeof(float) * heads * DH * (DH + 1), s, q, ldq, dout, lddo, ctx,
                     HW, heads, scale, dq, lddq);
  VMM_LAUNCH_CHECK();
  const long long tot = (long long)nfh * ntok * DH;
  hipLaunchKernelGGL(linattn_bwd_tokens_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, s, ek, ev, ntok, ctx, dctx, kstat, T, HW, heads, nfh, dek, dev);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_attention_bwd(int32_t mode, const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                                 int32_t tok_per_frame, const float* bias, int32_t bias_on_cond, const float* out, const float* dout,
                                 int32_t ldo, const float* lse, const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev,
                                 float* dbias, float* dbuf, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh,
                                 vmm_stream_t stream) {
  if (!vmm_head_dim_ok(dh) || heads < 1 || (ldqkv & 3) || (ldo & 3)) return -1;
  if (mode == 0 && !tok_per_frame) {  // temporal attention: LDS-staged workgroup-per-pixel kernel (temporal_attn_bwd.hip) where it applies
    const int rc = vmm_temporal_attention_bwd(qkv, ldqkv, ek, ev, ntok, bias, bias_on_cond, out, dout, ldo, lse, rot_tab, q_scale, dqkv, dek, dev,
                                              dbias, dbuf, B, T, HW, heads, dh, stream);
    if (rc != 1) return rc;
  }
  const int n = mode == 0 ? T : HW, ninner = mode == 0 ? HW : T;
  if (bias && n > 64) return -2;
  hipStream_t s = (hipStream_t)stream;
  AttnGeom g{mode, B, T, HW, heads, ek ? ntok : 0, tok_per_frame, bias_on_cond};
  const long long total = (long long)B * ninner * heads * n;
  if (total <= 0) return 0;
  const bool mid_spatial = dh == DH && mode == 1 && HW < 256 && !bias && !rot_tab && (g.ntok == 0 || (tok_per_frame && g.ntok == T));
  if (mid_spatial) {
    static bool attr_q = false;
    if (!attr_q) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_attn_bwd_q_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_q = true;
    }
    hipLaunchKernelGGL(spatial_attn_bwd_q_kernel, dim3((unsigned)(B * T * heads)), dim3(256), sizeof(float) * (size_t)HW * 2 * DH, s, g, qkv, ldqkv, ek, ev, out,
                       dout, ldo, lse, q_scale, dqkv, dbuf);
  } else {
#define VMM_CALL(DM, EX)                                                                                                                             \
  hipLaunchKernelGGL((attn_bwd_q_kernel<DM, EX>), dim3(cdiv(total, 256)), dim3(256), 0, s, g, dh, qkv, ldqkv, ek, ev, bias, out, dout, ldo, lse, rot_tab, \
                     q_scale, dqkv, dbuf)
    VMM_HEADVEC_DISPATCH(dh, VMM_CALL);
#undef VMM_CALL
  }
  VMM_LAUNCH_CHECK();
  if (mid_spatial) {
    static bool attr_set = false;
    if (!attr_set) {
      hipFuncSetAttribute(reinterpret_cast<const void*>(&spatial_attn_bwd_kv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
      attr_set = true;
    }
    const size_t shm = sizeof(float) * (size_t)HW * (2 * DH + 2);
    hipLaunchKernelGGL(spatial_attn_bwd_kv_kernel, dim3((unsigned)(B * T * heads)), dim3(256), shm, s, g, qkv, ldqkv, ek, ev, dout, ldo, lse, dbuf, dqkv, dek,
                       dev);
    VMM_LAUNCH_CHECK();
    return 0;
  }
  const int nchunks = (ninner + 63) / 64, ngroups = (nchunks + CHUNKS_PER_WAVE - 1) / CHUNKS_PER_WAVE;
  const long long nwaves = (long long)B * heads * (g.ntok + n) * ngroups;
#define VMM_CALL(DM, EX)                                                                                                                                \
  hipLaunchKernelGGL((attn_bwd_kv_kernel<DM, EX>), dim3(cdiv(nwaves, 4)), dim3(256), 0, s, g, dh, qkv, ldqkv, ek, ev, bias, dout, ldo, lse, dbuf, rot_tab, \
                     dqkv, dek, dev, dbias)
  VMM_HEADVEC_DISPATCH(dh, VMM_CALL);
#undef VMM_CALL
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_linattn_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* ctx,
                               const float* kstat, const float* dout, int32_t lddo, float* dctx /* [B*T*heads][32*32] scratch */,
                               float* dqkv, float* dek, float* dev, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh,
                               vmm_stream_t stream) {
  if (dh != DH || (ldqkv & 3) || (lddo & 3)) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int nfh = B * T * heads;
  if (int rc = vmm_zero_async(dctx, sizeof(float) * nfh * DH * DH, s)) return rc;
  const float scale = 0.17677669529663687f;
  const int nsplit = max(1, min((HW + LB_TILE - 1) / LB_TILE, cdiv(2048, nfh)));
  const int rps = cdiv(cdiv(HW, nsplit), LB_TILE) * LB_TILE;
  hipLaunchKernelGGL(linattn_bwd_dctx_kernel, dim3(cdiv(HW, rps), nfh), dim3(256), 0, s, qkv, ldqkv, dout, lddo, HW, heads, rps, scale, dctx);
  VMM_LAUNCH_CHECK();
  // row pass: fp32 matrix-core kernel (linattn_rows.hip) where it applies, else the thread-per-(row, head) kernel below
  int rc_rows = vmm_linattn_bwd_rows_mfma(qkv, ldqkv, dout, lddo, ctx, dctx, kstat, dqkv, B * T, HW, heads, scale, stream);
  if (rc_rows < 0 || rc_rows > 1) return rc_rows;
  const int hpb = (heads % 4 == 0) ? 4 : ((heads % 2 == 0) ? 2 : 1);
  const int rows_per_block = 256 / hpb;
  const size_t shm = sizeof(float) * (2 * hpb * (DH * DH + 4) + hpb * DH);
  if (rc_rows == 1) {
    hipLaunchKernelGGL(linattn_bwd_rows_kernel, dim3(cdiv(HW, rows_per_block), B * T, heads / hpb), dim3(256), shm, s, qkv, ldqkv, dout, lddo, ctx, dctx,
                       kstat, HW, heads, hpb, scale, dqkv);
    VMM_LAUNCH_CHECK();
  }
  if (ek && ntok > 0) {
    const long long tot = (long long)nfh * ntok * DH;
    hipLaunchKernelGGL(linattn_bwd_tokens_kernel, dim3(cdiv(tot, 256)), dim3(256), 0, s, ek, ev, ntok, ctx, dctx, kstat, T, HW, heads, nfh, dek, dev);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
