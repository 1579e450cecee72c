// cond_attention = 'cross-attention' (vddp.py:354-363 SpatialLinearAttention, vddp.py:476-485 Attention; north_star's "cross-attention on the
// stress-strain conditioning"): the queries come from their own projection (to_q), keys and values are the conditioning tokens ALONE -- a
// handful of rows per sample, the same for every frame and pixel.  Both flavours are therefore HBM-bound sweeps over the q rows with the
// sample's token keys / values resident in LDS:
//   * softmax attention (mid spatial site, temporal sites): out[row, head] = softmax_j(q . ek[j] (+ bias[head][t][j])) . ev   (this file);
//     the temporal sites add the (frames x frames) relative-position bias to the (frames x tokens) scores as the reference does, which
//     requires tokens == frames;
//   * linear attention: the context softmax_n(k)^T v / (h w) depends on the tokens only -- the merge kernel of the self-stacked path with no
//     pixel partials (attention.hip: vmm_linattn_cross_context) -- and vmm_linattn_apply runs unchanged on the q rows.
// Exact fp32 on the vector unit in both arithmetic modes: 2 x tokens x 32 flops per (row, head) against 256 bytes of q and out.
#include "vmm_common.h"
#include "head_vec.h"
#include "../../include/vmm_kernels.h"

namespace {

constexpr int DH = 32, TOK_MAX = 64, HSTR = DH + 1;  // (head stride 33 floats: the eight heads of a row hit eight banks)

// thread = (row, head); grid (row blocks of a sample, B); ek / ev of the sample in LDS as [token][head][dh + 1]
// (templated on the head slice, head_vec.h: the temporal sites follow attn_dim_head, vddp.py:615; 32 everywhere else)
template <int DM, bool EX>
__global__ __launch_bounds__(256) void cross_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ ek, const float* __restrict__ ev,
                                                         int ntok, const float* __restrict__ bias, int T, int HW, int heads, int dh_, float* __restrict__ out,
                                                         int ldo) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  const int hstr = dh + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int hid = heads * dh;
  float* ks = sm;
  float* vs = sm + ntok * heads * hstr;
  for (int i = tid; i < ntok * hid; i += 256) {
    const int j = i / hid, c = i - j * hid;
    const int h = c / dh, d = c - h * dh;
    ks[(j * heads + h) * hstr + d] = ek[((long long)b * ntok + j) * hid + c];
    vs[(j * heads + h) * hstr + d] = ev[((long long)b * ntok + j) * hid + c];
  }
  __syncthreads();
  const int rows_per_block = 256 / heads;
  const int head = tid % heads;
  const int r = blockIdx.x * rows_per_block + tid / heads;  // row inside the sample: t * HW + pixel
  if (r >= T * HW || tid / heads >= rows_per_block) return;
  const long long row = (long long)b * T * HW + r;
  float qv[DM];
  HV::ld(qv, q + row * ldq + head * dh, dh);
  const float* bp = bias ? bias + ((long long)head * T + r / HW) * T : nullptr;
  float s[TOK_MAX];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    float a = -INFINITY;
    if (j < ntok) {
      const float* kp = ks + (j * heads + head) * hstr;
      a = 0.f;
#pragma unroll
      for (int d = 0; d < DM; ++d)
        if (EX || d < dh) a = fmaf(qv[d], kp[d], a);
      if (bp) a += bp[j];
    }
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f, o[DM];
  HV::zero(o);
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    if (j < ntok) {
      const float p = __expf(s[j] - mx);
      sum += p;
      const float* vp = vs + (j * heads + head) * hstr;
#pragma unroll
      for (int e = 0; e < DM; ++e)
        if (EX || e < dh) o[e] = fmaf(p, vp[e], o[e]);
    }
  }
  const float inv = 1.0f / sum;
#pragma unroll
  for (int e = 0; e < DM; ++e) o[e] *= inv;
  HV::st(out + row * ldo + head * dh, o, dh);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// Backward of the softmax cross-attention core (training with cond_attention = 'cross-attention').  With p_j = softmax_j(q . k_j + bias_j):
//   dp_j = dO . v_j,  D = sum_j p_j dp_j,  ds_j = p_j (dp_j - D),
//   dq = sum_j ds_j k_j   (then R^T and * scale: the projection epilogue's rotation and q-scale are undone here, the kernel emits the
//                          gradient of the RAW to_q output, like vmm_attention_bwd),
//   dk_j = sum_rows ds_j q,  dv_j = sum_rows p_j dO,  dbias[h][t][j] = sum_{b, pixels} ds_j.
// The token gradients are sums over every row of a sample: a workgroup walks `chunks` runs of 32 rows; phase 1 (thread = (row, head)) leaves
// ds and p of the run in LDS and writes dq, phase 2 (thread = (head, d)) accumulates its column of dk / dv for all tokens in registers over
// the whole walk; one atomic per (token, column) and workgroup at the end.  heads * 32 == 256, tokens <= 16.
constexpr int BT_MAX = 16, RB = 32;

__global__ __launch_bounds__(256) void cross_attn_bwd_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ ek, const float* __restrict__ ev,
                                                             int ntok, const float* __restrict__ bias, const float* __restrict__ dout, int lddo,
                                                             const float* __restrict__ rot_tab, float q_scale, float* __restrict__ dq, int lddq,
                                                             float* __restrict__ dek, float* __restrict__ dev, float* __restrict__ dbias, int T, int HW,
                                                             int chunks) {
  constexpr int heads = 8, hid = 256;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, b = blockIdx.y;
  float* ks = sm;                                // [ntok][8][33]
  float* vs = ks + ntok * heads * HSTR;          // [ntok][8][33]
  float* dsb = vs + ntok * heads * HSTR;         // [RB][8][BT_MAX]
  float* ppb = dsb + RB * heads * BT_MAX;        // [RB][8][BT_MAX]
  float* dbl = ppb + RB * heads * BT_MAX;        // [8][T][ntok] (bias only)
  for (int i = tid; i < ntok * hid; i += 256) {
    const int j = i / hid, c = i - j * hid;
    const int h = c >> 5, d = c & 31;
    ks[(j * heads + h) * HSTR + d] = ek[((long long)b * ntok + j) * hid + c];
    vs[(j * heads + h) * HSTR + d] = ev[((long long)b * ntok + j) * hid + c];
  }
  if (bias)
    for (int i = tid; i < heads * T * ntok; i += 256) dbl[i] = 0.f;
  const int rows_s = T * HW;                     // rows of the sample
  const long long row_b = (long long)b * rows_s;
  const int r1 = tid >> 3, h1 = tid & 7;         // phase 1: (row of the run, head)
  const int h2 = tid >> 5, d2 = tid & 31;        // phase 2: (head, column)
  float gk[BT_MAX], gv[BT_MAX];
#pragma unroll
  for (int j = 0; j < BT_MAX; ++j) gk[j] = gv[j] = 0.f;
  __syncthreads();
  for (int c = 0; c < chunks; ++c) {
    const int r0 = (blockIdx.x * chunks + c) * RB;
    if (r0 >= rows_s) break;  // (uniform)
    {  // ---- phase 1
      const int r = r0 + r1;
      const bool valid = r < rows_s;
      const long long row = row_b + (valid ? r : rows_s - 1);
      float qv[DH], go[DH];
      const float* qp = q + row * ldq + h1 * DH;
      const float* gp = dout + row * lddo + h1 * DH;
#pragma unroll
      for (int d4 = 0; d4 < DH / 4; ++d4) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(qp + 4 * d4), g = *reinterpret_cast<const f32x4*>(gp + 4 * d4);
        qv[4 * d4] = a.x; qv[4 * d4 + 1] = a.y; qv[4 * d4 + 2] = a.z; qv[4 * d4 + 3] = a.w;
        go[4 * d4] = g.x; go[4 * d4 + 1] = g.y; go[4 * d4 + 2] = g.z; go[4 * d4 + 3] = g.w;
      }
      const int t = (valid ? r : rows_s - 1) / HW;
      const float* bp = bias ? bias + ((long long)h1 * T + t) * T : nullptr;
      float s[BT_MAX], dp[BT_MAX];
      float mx = -INFINITY;
#pragma unroll
      for (int j = 0; j < BT_MAX; ++j) {
        float a = -INFINITY, g = 0.f;
        if (j < ntok) {
          const float* kp = ks + (j * heads + h1) * HSTR;
          const float* vp = vs + (j * heads + h1) * HSTR;
          a = 0.f;
#pragma unroll
          for (int d = 0; d < DH; ++d) { a = fmaf(qv[d], kp[d], a); g = fmaf(go[d], vp[d], g); }
          if (bp) a += bp[j];
        }
        s[j] = a;
        dp[j] = g;
        mx = fmaxf(mx, a);
      }
      float sum = 0.f;
#pragma unroll
      for (int j = 0; j < BT_MAX; ++j) { s[j] = j < ntok ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
      const float inv = 1.0f / sum;
      float D = 0.f;
#pragma unroll
      for (int j = 0; j < BT_MAX; ++j) { s[j] *= inv; D = fmaf(s[j], dp[j], D); }
      float gq[DH];
#pragma unroll
      for (int d = 0; d < DH; ++d) gq[d] = 0.f;
#pragma unroll
      for (int j = 0; j < BT_MAX; ++j) {
        const float pj = valid ? s[j] : 0.f;
        const float ds = pj * (dp[j] - D);
        dsb[(r1 * heads + h1) * BT_MAX + j] = ds;
        ppb[(r1 * heads + h1) * BT_MAX + j] = pj;
        if (j < ntok) {
          const float* kp = ks + (j * heads + h1) * HSTR;
#pragma unroll
          for (int d = 0; d < DH; ++d) gq[d] = fmaf(ds, kp[d], gq[d]);
          if (bias && valid) atomicAdd(&dbl[(h1 * T + t) * ntok + j], ds);
        }
      }
      if (valid) {
        if (rot_tab) {  // transpose of the interleaved-pair rotation by the row's frame
#pragma unroll
          for (int f = 0; f < DH / 2; ++f) {
            const float cs = rot_tab[(t * (DH / 2) + f) * 2], sn = rot_tab[(t * (DH / 2) + f) * 2 + 1];
            const float a = gq[2 * f], bb = gq[2 * f + 1];
            gq[2 * f] = a * cs + bb * sn;
            gq[2 * f + 1] = bb * cs - a * sn;
          }
        }
        float* op = dq + row * lddq + h1 * DH;
#pragma unroll
        for (int d4 = 0; d4 < DH / 4; ++d4)
          *reinterpret_cast<f32x4*>(op + 4 * d4) = f32x4{gq[4 * d4] * q_scale, gq[4 * d4 + 1] * q_scale, gq[4 * d4 + 2] * q_scale, gq[4 * d4 + 3] * q_scale};
      }
    }
    __syncthreads();
    {  // ---- phase 2: this thread's column of dk / dv, all tokens (q and dO of the run come back from L2, 128 bytes per head and row)
      const int nr = min(RB, rows_s - r0);
      for (int r = 0; r < nr; ++r) {
        const long long row = row_b + r0 + r;
        const float qd = q[row * ldq + h2 * DH + d2], gd = dout[row * lddo + h2 * DH + d2];
        const float* dsr = dsb + (r * heads + h2) * BT_MAX;
        const float* ppr = ppb + (r * heads + h2) * BT_MAX;
#pragma unroll
        for (int j4 = 0; j4 < BT_MAX / 4; ++j4) {
          const f32x4 a = *reinterpret_cast<const f32x4*>(dsr + 4 * j4), p4 = *reinterpret_cast<const f32x4*>(ppr + 4 * j4);
          gk[4 * j4] = fmaf(a.x, qd, gk[4 * j4]); gk[4 * j4 + 1] = fmaf(a.y, qd, gk[4 * j4 + 1]);
          gk[4 * j4 + 2] = fmaf(a.z, qd, gk[4 * j4 + 2]); gk[4 * j4 + 3] = fmaf(a.w, qd, gk[4 * j4 + 3]);
          gv[4 * j4] = fmaf(p4.x, gd, gv[4 * j4]); gv[4 * j4 + 1] = fmaf(p4.y, gd, gv[4 * j4 + 1]);
          gv[4 * j4 + 2] = fmaf(p4.z, gd, gv[4 * j4 + 2]); gv[4 * j4 + 3] = fmaf(p4.w, gd, gv[4 * j4 + 3]);
        }
      }
    }
    __syncthreads();
  }
#pragma unroll
  for (int j = 0; j < BT_MAX; ++j)
    if (j < ntok) {
      atomicAdd(&dek[((long long)b * ntok + j) * hid + h2 * DH + d2], gk[j]);
      atomicAdd(&dev[((long long)b * ntok + j) * hid + h2 * DH + d2], gv[j]);
    }
  if (bias && dbias)
    for (int i = tid; i < heads * T * ntok; i += 256) atomicAdd(&dbias[i], dbl[i]);
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The same backward for every other shape (any number of heads, any head slice of head_vec.h, up to 32 tokens): two passes over the q rows.
//   pass A, thread = (row, head): p, dp, D, ds as above; dq written (rotation and q-scale undone); ds and p of the row left in a scratch
//           [rows][heads][ntok] each (the only intermediate that goes through memory);
//   pass B, workgroup = (sample, head, token): dk_j = sum_rows ds_j q, dv_j = sum_rows p_j dO -- every thread walks a stride of the
//           sample's rows with the head slice in registers, the workgroup reduces in LDS and ADDS its (token, head) slice into dek / dev (one
//           writer per element: no atomics); the bias gradient dbias[h][t][j] += sum_pixels ds_j per frame (one atomic per frame: samples share it).
template <int DM, bool EX>
__global__ __launch_bounds__(256) void cross_attn_bwd_rows_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ ek, const float* __restrict__ ev,
                                                                  int ntok, const float* __restrict__ bias, const float* __restrict__ dout, int lddo,
                                                                  const float* __restrict__ rot_tab, float q_scale, float* __restrict__ dq, int lddq,
                                                                  float* __restrict__ ds_out, float* __restrict__ p_out, int T, int HW, int heads, int dh_) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  const int hstr = dh + 1;
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int hid = heads * dh;
  float* ks = sm;
  float* vs = sm + ntok * heads * hstr;
  for (int i = tid; i < ntok * hid; i += 256) {
    const int j = i / hid, c = i - j * hid;
    const int h = c / dh, d = c - h * dh;
    ks[(j * heads + h) * hstr + d] = ek[((long long)b * ntok + j) * hid + c];
    vs[(j * heads + h) * hstr + d] = ev[((long long)b * ntok + j) * hid + c];
  }
  __syncthreads();
  const int rows_per_block = 256 / heads;
  const int head = tid % heads;
  const int r = blockIdx.x * rows_per_block + tid / heads;
  if (r >= T * HW || tid / heads >= rows_per_block) return;
  const long long row = (long long)b * T * HW + r;
  const int t = r / HW;
  float qv[DM], go[DM], gq[DM];
  HV::ld(qv, q + row * ldq + head * dh, dh);
  HV::ld(go, dout + row * lddo + head * dh, dh);
  HV::zero(gq);
  const float* bp = bias ? bias + ((long long)head * T + t) * T : nullptr;
  float s[TOK_MAX], dp[TOK_MAX];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    float a = -INFINITY, g = 0.f;
    if (j < ntok) {
      const float* kp = ks + (j * heads + head) * hstr;
      const float* vp = vs + (j * heads + head) * hstr;
      a = 0.f;
#pragma unroll
      for (int d = 0; d < DM; ++d)
        if (EX || d < dh) { a = fmaf(qv[d], kp[d], a); g = fmaf(go[d], vp[d], g); }
      if (bp) a += bp[j];
    }
    s[j] = a;
    dp[j] = g;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) { s[j] = j < ntok ? __expf(s[j] - mx) : 0.f; sum += s[j]; }
  const float inv = 1.0f / sum;
  float D = 0.f;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) { s[j] *= inv; D = fmaf(s[j], dp[j], D); }
  float* dsr = ds_out + (row * heads + head) * ntok;
  float* ppr = p_out + (row * heads + head) * ntok;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    if (j < ntok) {
      const float ds = s[j] * (dp[j] - D);
      dsr[j] = ds;
      ppr[j] = s[j];
      const float* kp = ks + (j * heads + head) * hstr;
#pragma unroll
      for (int d = 0; d < DM; ++d)
        if (EX || d < dh) gq[d] = fmaf(ds, kp[d], gq[d]);
    }
  }
  if (rot_tab) HV::unrotate(gq, rot_tab, t, dh);
#pragma unroll
  for (int d = 0; d < DM; ++d) gq[d] *= q_scale;
  HV::st(dq + row * lddq + head * dh, gq, dh);
}

template <int DM, bool EX>
__global__ __launch_bounds__(256) void cross_attn_bwd_tokens_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ dout, int lddo,
                                                                    const float* __restrict__ ds_in, const float* __restrict__ p_in, int ntok,
                                                                    float* __restrict__ dek, float* __restrict__ dev, float* __restrict__ dbias, int T, int HW,
                                                                    int heads, int dh_) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  __shared__ float red[4][2 * DM];
  __shared__ float bred[4];
  const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
  const int j = blockIdx.x % ntok, head = (blockIdx.x / ntok) % heads, b = blockIdx.x / (ntok * heads);
  const int hid = heads * dh;
  const long long row_b = (long long)b * T * HW;
  float gk[DM], gv[DM];
  HV::zero(gk);
  HV::zero(gv);
  for (int t = 0; t < T; ++t) {
    float bsum = 0.f;
    for (int pix = tid; pix < HW; pix += 256) {
      const long long row = row_b + (long long)t * HW + pix;
      const float ds = ds_in[(row * heads + head) * ntok + j], pj = p_in[(row * heads + head) * ntok + j];
      float qv[DM], go[DM];
      HV::ld(qv, q + row * ldq + head * dh, dh);
      HV::ld(go, dout + row * lddo + head * dh, dh);
#pragma unroll
      for (int d = 0; d < DM; ++d) { gk[d] = fmaf(ds, qv[d], gk[d]); gv[d] = fmaf(pj, go[d], gv[d]); }
      bsum += ds;
    }
    if (dbias) {  // (workgroup-uniform)
      bsum = wave_sum(bsum);
      if (lane == 0) bred[wv] = bsum;
      __syncthreads();
      if (tid == 0) atomicAdd(&dbias[((long long)head * T + t) * ntok + j], (bred[0] + bred[1]) + (bred[2] + bred[3]));
      __syncthreads();
    }
  }
#pragma unroll
  for (int d = 0; d < DM; ++d) {
    const float a = wave_sum(gk[d]), c = wave_sum(gv[d]);
    if (lane == 0) { red[wv][d] = a; red[wv][DM + d] = c; }
  }
  __syncthreads();
  if (tid < 2 * DM) {
    const int d = tid < DM ? tid : tid - DM;
    if (d < dh) {
      const float v = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
      float* dst = (tid < DM ? dek : dev) + ((long long)b * ntok + j) * hid + head * dh + d;
      *dst += v;
    }
  }
}

}  // namespace

// Backward of vmm_cross_attention.  q: the rows the forward consumed (scaled, rotated); dout [rows][heads*dh]; writes dq = gradient of the RAW
// to_q output (rot_tab [T][dh/2][2] (cos, sin) or NULL, q_scale: the projection epilogue undone); ADDS the token gradients into dek / dev
// [B][ntok][heads*dh] and the bias gradient into dbias [heads][T][T] (may be NULL; only with bias).  scratch: vmm_cross_attention_bwd_scratch
// floats (0 inside the fused kernel's envelope -- 8 heads of 32, at most 16 tokens -- where NULL is accepted).  -1: dh not a multiple of 4
// in 4..128, ntok outside 1..64, bias with ntok != T, misaligned rows, a missing scratch.
extern "C" int64_t vmm_cross_attention_bwd_scratch(int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, int32_t ntok) {
  if (dh == DH && heads == 8 && ntok <= BT_MAX) return 0;
  return 2 * (int64_t)B * T * HW * heads * ntok;
}

extern "C" int vmm_cross_attention_bwd(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, const float* dout,
                                       int32_t lddo, const float* rot_tab, float q_scale, float* dq, int32_t lddq, float* dek, float* dev, float* dbias,
                                       float* scratch, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (!vmm_head_dim_ok(dh) || heads < 1 || heads > 64 || ntok < 1 || ntok > TOK_MAX || (bias && ntok != T) || (ldq & 3) || (lddo & 3) || (lddq & 3) || !dek || !dev)
    return -1;
  if (B <= 0 || T <= 0 || HW <= 0) return 0;
  const long long rows_s = (long long)T * HW;
  if (dh == DH && heads == 8 && ntok <= BT_MAX) {
    const int nruns = (int)cdiv(rows_s, RB);
    const int nblk = nruns < 96 ? nruns : 96;  // workgroups per sample: each ends with 2 * ntok * 256 atomics
    const int chunks = (int)cdiv(nruns, nblk);
    const size_t shm = sizeof(float) * (2 * (size_t)ntok * heads * HSTR + 2 * (size_t)RB * heads * BT_MAX + (bias ? (size_t)heads * T * ntok : 0));
    static bool attr_set = false;
    if (!attr_set) {
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
      attr_set = true;
    }
    hipLaunchKernelGGL(cross_attn_bwd_kernel, dim3((unsigned)cdiv(nruns, chunks), (unsigned)B), dim3(256), shm, (hipStream_t)stream, q, ldq, ek, ev, ntok, bias,
                       dout, lddo, rot_tab, q_scale, dq, lddq, dek, dev, dbias, T, HW, chunks);
    VMM_LAUNCH_CHECK();
    return 0;
  }
  if (!scratch) return -1;
  float* ds_buf = scratch;
  float* p_buf = scratch + (long long)B * rows_s * heads * ntok;
  const int rows_per_block = 256 / heads;
  const size_t shm = sizeof(float) * 2 * (size_t)ntok * heads * (dh + 1);
  if (shm > 160 * 1024) return -1;
  hipStream_t s = (hipStream_t)stream;
#define VMM_CALL(DM, EX)                                                                                                                                     \
  do {                                                                                                                                                       \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_bwd_rows_kernel<DM, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);    \
    hipLaunchKernelGGL((cross_attn_bwd_rows_kernel<DM, EX>), dim3(cdiv(rows_s, rows_per_block), B), dim3(256), shm, s, q, ldq, ek, ev, ntok, bias, dout, lddo, \
                       rot_tab, q_scale, dq, lddq, ds_buf, p_buf, T, HW, heads, dh);                                                                          \
    hipLaunchKernelGGL((cross_attn_bwd_tokens_kernel<DM, EX>), dim3((unsigned)(B * heads * ntok)), dim3(256), 0, s, q, ldq, dout, lddo, ds_buf, p_buf, ntok,  \
                       dek, dev, bias ? dbias : nullptr, T, HW, heads, dh);                                                                                   \
  } while (0)
  VMM_HEADVEC_DISPATCH(dh, VMM_CALL);
#undef VMM_CALL
  VMM_LAUNCH_CHECK();
  return 0;
}

// out[row, head*dh + e] = sum_j softmax_j(q[row, head] . ek[b][j][head] (+ bias[head][t(row)][j])) ev[b][j][head*dh + e]; q rows [(b, t, pixel)] x
// heads*dh (ldq), already scaled (and rotated for the temporal sites) by the projection's epilogue; ek / ev [B][ntok][heads*dh]; bias [heads][T][T]
// or NULL (then T only sizes the sample: rows per sample = T * HW).  -1: dh not a multiple of 4 in 4..128, ntok outside 1..64, bias with ntok != T,
// misaligned rows, more than 64 heads.
extern "C" int vmm_cross_attention(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, float* out, int32_t ldo,
                                   int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (!vmm_head_dim_ok(dh) || ntok < 1 || ntok > TOK_MAX || (bias && ntok != T) || (ldq & 3) || (ldo & 3) || heads < 1 || heads > 64) return -1;
  if (B <= 0 || T <= 0 || HW <= 0) return 0;
  const int rows_per_block = 256 / heads;
  const size_t shm = sizeof(float) * 2 * (size_t)ntok * heads * (dh + 1);
  if (shm > 160 * 1024) return -1;
#define VMM_CALL(DM, EX)                                                                                                                                  \
  do {                                                                                                                                                    \
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_kernel<DM, EX>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);          \
    hipLaunchKernelGGL((cross_attn_kernel<DM, EX>), dim3(cdiv((long long)T * HW, rows_per_block), B), dim3(256), shm, (hipStream_t)stream, q, ldq, ek, ev, \
                       ntok, bias, T, HW, heads, dh, out, ldo);                                                                                            \
  } while (0)
  VMM_HEADVEC_DISPATCH(dh, VMM_CALL);
#undef VMM_CALL
  VMM_LAUNCH_CHECK();
  return 0;
}
