// Temporal attention core, exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32), for the path that keeps qkv / out / logsumexp
// (training; inference outside the fused-kernel envelopes), gfx950.  heads = 8, dim_head = 32, T <= 16 frames, <= 16 tokens.
//
// A wave owns one head and walks the workgroup's pixels (no LDS, no barriers).  Per pixel:
//   S^T = K Q^T (and the token keys' EK Q^T)      accumulator layout: register = key 4 g + r, lane = query c
//   softmax over the keys of a query              = over a lane's registers and its 4 lane groups (two shuffles)
//   O = P V (+ P_tok EV)                          the normalised probabilities ARE the "A" operand (lane = query, contraction = key)
// 32 MFMAs per pixel and head; q / k rows are read in row layout (lane = frame, 8 channels per lane group), v in column layout
// (lane = channel, register = frame).  (Replaced an LDS-staged packed-FMA kernel of the same blocking: 2.6 -> 1.8 ms per training
// step over the 10 sites; the thread-per-query kernel of attention.hip, 3.4 ms, remains the fallback outside the envelope.)
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32, HEADS = 8, HID = HEADS * DH;

struct TFMArgs {
  const float *qkv, *ek, *ev, *bias;
  float *out, *lse;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample;
};

__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void load_row8(float (&dst)[8], const float* p, bool ok) {
  f32x4 u = {0.f, 0.f, 0.f, 0.f}, w = u;
  if (ok) {
    u = *reinterpret_cast<const f32x4*>(p);
    w = *reinterpret_cast<const f32x4*>(p + 4);
  }
  dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
  dst[4] = w.x; dst[5] = w.y; dst[6] = w.z; dst[7] = w.w;
}

__global__ __launch_bounds__(512) void temporal_attn_fwd_mfma_kernel(const TFMArgs a) {
  const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, c = lane & 15, g = lane >> 4;
  const int T = a.T, ntok = a.ntok;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool tok_bias = a.bias && a.bias_on_cond;
  const bool cT = c < T;
  // bias in the score layout: register r <-> key j = 4 g + r, lane <-> query i = c
  float bB[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) bB[r] = (a.bias && cT && 4 * g + r < T) ? a.bias[((long long)head * T + c) * T + 4 * g + r] : 0.f;
  float ekr[8], evc[2][4];
  load_row8(ekr, a.ek + ((long long)b * ntok + c) * HID + head * DH + 8 * g, c < ntok);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) evc[h][r] = (4 * g + r < ntok) ? a.ev[((long long)b * ntok + 4 * g + r) * HID + head * DH + c + 16 * h] : 0.f;
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};

  for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
    const long long row0 = (long long)b * T * a.HW + pix;
    const long long rc = row0 + (long long)c * a.HW;
    const float* qrow = a.qkv + rc * a.ldqkv + head * DH + 8 * g;
    float qr[8], kr[8], vc[2][4];
    load_row8(qr, qrow, cT);
    load_row8(kr, qrow + HID, cT);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool ok = 4 * g + r < T;
      const float* vrow = a.qkv + (row0 + (long long)(4 * g + r) * a.HW) * a.ldqkv + 2 * HID + head * DH + c;
      vc[0][r] = ok ? vrow[0] : 0.f;
      vc[1][r] = ok ? vrow[16] : 0.f;
    }
    f32x4 S = zero4, St = zero4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      S = mm(kr[s], qr[s], S);     // S^T[j][i]
      St = mm(ekr[s], qr[s], St);  // token keys
    }
    float sv[4], st[4];
    float m = -INFINITY;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int j = 4 * g + r;
      sv[r] = j < T ? S[r] + bB[r] : -INFINITY;
      st[r] = j < ntok ? St[r] + (tok_bias ? bB[r] : 0.f) : -INFINITY;
      m = fmaxf(m, fmaxf(sv[r], st[r]));
    }
    m = fmaxf(m, __shfl_xor(m, 16, 64));
    m = fmaxf(m, __shfl_xor(m, 32, 64));
    float l = 0.f;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      sv[r] = __expf(sv[r] - m);
      st[r] = __expf(st[r] - m);
      l += sv[r] + st[r];
    }
    l += __shfl_xor(l, 16, 64);
    l += __shfl_xor(l, 32, 64);
    const float inv = 1.0f / l;
    f32x4 O[2] = {zero4, zero4};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        O[h] = mm(sv[r] * inv, vc[h][r], O[h]);
        O[h] = mm(st[r] * inv, evc[h][r], O[h]);
      }
    if (a.lse && g == 0 && cT) a.lse[rc * HEADS + head] = m + logf(l);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (4 * g + r < T) a.out[(row0 + (long long)(4 * g + r) * a.HW) * a.ldo + head * DH + c + 16 * h] = O[h][r];
  }
}

}  // namespace

// Fast path of vmm_temporal_attention (same arguments and results).  Returns 1 (nothing launched) outside its envelope: heads = 8,
// dim_head = 32, T <= 16, ntok <= 16, ntok <= T when the bias also covers the tokens.
extern "C" int vmm_temporal_attention_staged(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                             int32_t bias_on_cond, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads,
                                             int32_t dh, float* lse, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 16 || T < 1 || ntok > 16 || (ldqkv & 3) || (ldo & 3)) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (B <= 0 || HW <= 0) return 0;
  TFMArgs a{qkv, ek, ev, bias, out, lse, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0};
  a.blocks_per_sample = (int)max(1LL, min((long long)HW, cdiv(1024, B)));
  hipLaunchKernelGGL(temporal_attn_fwd_mfma_kernel, dim3((unsigned)(B * a.blocks_per_sample)), dim3(512), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
