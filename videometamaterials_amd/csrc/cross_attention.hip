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

}  // namespace

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
