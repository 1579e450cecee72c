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
#include "../../include/vmm_kernels.h"

namespace {

constexpr int DH = 32, TOK_MAX = 32, HSTR = DH + 1;  // (head stride 33 floats: the eight heads of a row hit eight banks)

// thread = (row, head); grid (row blocks of a sample, B); ek / ev of the sample in LDS as [token][head][33]
__global__ __launch_bounds__(256) void cross_attn_kernel(const float* __restrict__ q, int ldq, const float* __restrict__ ek, const float* __restrict__ ev,
                                                         int ntok, const float* __restrict__ bias, int T, int HW, int heads, float* __restrict__ out,
                                                         int ldo) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const int tid = threadIdx.x, b = blockIdx.y;
  const int hid = heads * DH;
  float* ks = sm;
  float* vs = sm + ntok * heads * HSTR;
  for (int i = tid; i < ntok * hid; i += 256) {
    const int j = i / hid, c = i - j * hid;
    const int h = c >> 5, d = c & 31;
    ks[(j * heads + h) * HSTR + d] = ek[((long long)b * ntok + j) * hid + c];
    vs[(j * heads + h) * HSTR + d] = ev[((long long)b * ntok + j) * hid + c];
  }
  __syncthreads();
  const int rows_per_block = 256 / heads;
  const int head = tid % heads;
  const int r = blockIdx.x * rows_per_block + tid / heads;  // row inside the sample: t * HW + pixel
  if (r >= T * HW || tid / heads >= rows_per_block) return;
  const long long row = (long long)b * T * HW + r;
  float qv[DH];
  const float* qp = q + row * ldq + head * DH;
#pragma unroll
  for (int d4 = 0; d4 < DH / 4; ++d4) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(qp + 4 * d4);
    qv[4 * d4] = v.x; qv[4 * d4 + 1] = v.y; qv[4 * d4 + 2] = v.z; qv[4 * d4 + 3] = v.w;
  }
  const float* bp = bias ? bias + ((long long)head * T + r / HW) * T : nullptr;
  float s[TOK_MAX];
  float mx = -INFINITY;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    float a = -INFINITY;
    if (j < ntok) {
      const float* kp = ks + (j * heads + head) * HSTR;
      a = 0.f;
#pragma unroll
      for (int d = 0; d < DH; ++d) a = fmaf(qv[d], kp[d], a);
      if (bp) a += bp[j];
    }
    s[j] = a;
    mx = fmaxf(mx, a);
  }
  float sum = 0.f, o[DH];
#pragma unroll
  for (int e = 0; e < DH; ++e) o[e] = 0.f;
#pragma unroll
  for (int j = 0; j < TOK_MAX; ++j) {
    if (j < ntok) {
      const float p = __expf(s[j] - mx);
      sum += p;
      const float* vp = vs + (j * heads + head) * HSTR;
#pragma unroll
      for (int e = 0; e < DH; ++e) o[e] = fmaf(p, vp[e], o[e]);
    }
  }
  const float inv = 1.0f / sum;
  float* op = out + row * ldo + head * DH;
#pragma unroll
  for (int e4 = 0; e4 < DH / 4; ++e4)
    *reinterpret_cast<f32x4*>(op + 4 * e4) = f32x4{o[4 * e4] * inv, o[4 * e4 + 1] * inv, o[4 * e4 + 2] * inv, o[4 * e4 + 3] * inv};
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

}  // namespace

// Backward of vmm_cross_attention.  q: the rows the forward consumed (scaled, rotated); dout [rows][heads*32]; writes dq = gradient of the RAW
// to_q output (rot_tab [T][16][2] (cos, sin) or NULL, q_scale: the projection epilogue undone); ADDS the token gradients into dek / dev
// [B][ntok][heads*32] and the bias gradient into dbias [heads][T][T] (may be NULL; only with bias).  -1: dh != 32, heads != 8, ntok outside
// 1..16, bias with ntok != T, misaligned rows.
extern "C" int vmm_cross_attention_bwd(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, const float* dout,
                                       int32_t lddo, const float* rot_tab, float q_scale, float* dq, int32_t lddq, float* dek, float* dev, float* dbias,
                                       int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (dh != DH || heads != 8 || ntok < 1 || ntok > BT_MAX || (bias && ntok != T) || (ldq & 3) || (lddo & 3) || (lddq & 3) || !dek || !dev) return -1;
  if (B <= 0 || T <= 0 || HW <= 0) return 0;
  const long long rows_s = (long long)T * HW;
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

// out[row, head*32 + e] = sum_j softmax_j(q[row, head] . ek[b][j][head] (+ bias[head][t(row)][j])) ev[b][j][head*32 + e]; q rows [(b, t, pixel)] x
// heads*32 (ldq), already scaled (and rotated for the temporal sites) by the projection's epilogue; ek / ev [B][ntok][heads*32]; bias [heads][T][T]
// or NULL (then T only sizes the sample: rows per sample = T * HW).  -1: dh != 32, ntok outside 1..32, bias with ntok != T, misaligned rows.
extern "C" int vmm_cross_attention(const float* q, int32_t ldq, const float* ek, const float* ev, int32_t ntok, const float* bias, float* out, int32_t ldo,
                                   int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (dh != DH || ntok < 1 || ntok > TOK_MAX || (bias && ntok != T) || (ldq & 3) || (ldo & 3) || heads < 1 || heads > 64 || 256 % heads) return -1;
  if (B <= 0 || T <= 0 || HW <= 0) return 0;
  const int rows_per_block = 256 / heads;
  const size_t shm = sizeof(float) * 2 * (size_t)ntok * heads * HSTR;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&cross_attn_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(cross_attn_kernel, dim3(cdiv((long long)T * HW, rows_per_block), B), dim3(256), shm, (hipStream_t)stream, q, ldq, ek, ev, ntok, bias, T, HW,
                     heads, out, ldo);
  VMM_LAUNCH_CHECK();
  return 0;
}
