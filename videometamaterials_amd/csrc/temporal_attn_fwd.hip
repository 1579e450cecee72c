// Temporal attention core, exact fp32, for the path that keeps qkv / out / logsumexp for the backward pass (training; inference
// outside the fused-kernel envelopes), gfx950.  heads = 8, dim_head = 32, T <= 16 frames, <= 16 conditioning tokens.
//
// Same shape as temporal_attn_bwd.hip: one workgroup = one pixel at a time, 8 heads x 16 lanes, lane i = query frame i.  The pixel's
// T rows of k | v are read once, coalesced, into LDS (the thread-per-query kernel of attention.hip leaves that reuse to L1: every
// thread walks all T key rows itself); the sample's conditioning keys / values are staged once per workgroup.  Online softmax on
// packed fp32 FMAs; writes out rows and the logsumexp per (row, head).
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

#include <cstdlib>

// the matrix-core version (temporal_attn_fwd_mfma.hip)
int vmm_temporal_attention_fwd_mfma_launch(const float* qkv, int ldqkv, const float* ek, const float* ev, int ntok, const float* bias, int bias_on_cond,
                                           float* out, int ldo, float* lse, int B, int T, int HW, hipStream_t s);

namespace {
constexpr int DH = 32, HEADS = 8, HID = HEADS * DH, NTH = HEADS * 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TFArgs {
  const float *qkv, *ek, *ev, *bias;
  float *out, *lse;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample;
};

// rows of 32 floats, 16-byte chunk c of row r stored at chunk c ^ (r & 7)
__device__ __forceinline__ int sw(int r, int c) { return r * DH + ((c ^ (r & 7)) << 2); }

__device__ __forceinline__ void lds_row(f32x2 (&dst)[16], const float* base, int r) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + sw(r, c));
    dst[2 * c] = (f32x2){v.x, v.y};
    dst[2 * c + 1] = (f32x2){v.z, v.w};
  }
}
__device__ __forceinline__ float dot32(const f32x2 (&a)[16], const f32x2 (&b)[16]) {
  f32x2 s0 = {0.f, 0.f}, s1 = {0.f, 0.f};
#pragma unroll
  for (int c = 0; c < 16; c += 2) { s0 = a[c] * b[c] + s0; s1 = a[c + 1] * b[c + 1] + s1; }
  s0 += s1;
  return s0.x + s0.y;
}

__global__ __launch_bounds__(NTH) void temporal_attn_fwd_kernel(const TFArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, ntok = a.ntok;
  float* Ks = smem;                      // [HEADS][T][32] (swizzled chunks)
  float* Vs = Ks + HEADS * T * DH;
  float* EK = Vs + HEADS * T * DH;       // [HEADS][ntok][32] (swizzled), the sample's conditioning keys / values
  float* EV = EK + HEADS * ntok * DH;
  float* Bs = EV + HEADS * ntok * DH;    // bias [HEADS][T][T]
  const int tid = threadIdx.x, head = tid >> 4, i = tid & 15;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool act = i < T;
  const bool tok_bias = a.bias && a.bias_on_cond;
  for (int e = tid; e < HEADS * T * T; e += NTH) Bs[e] = a.bias ? a.bias[e] : 0.f;
  for (int e = tid; e < ntok * (HID / 4); e += NTH) {
    const int j = e >> 6, c4 = e & 63, h = c4 >> 3;
    const long long off = ((long long)b * ntok + j) * HID + c4 * 4;
    *reinterpret_cast<f32x4*>(EK + sw(h * ntok + j, c4 & 7)) = *reinterpret_cast<const f32x4*>(a.ek + off);
    *reinterpret_cast<f32x4*>(EV + sw(h * ntok + j, c4 & 7)) = *reinterpret_cast<const f32x4*>(a.ev + off);
  }
  const float* Bh = Bs + (head * T + i) * T;

  for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
    const long long row0 = (long long)b * T * a.HW + pix;  // row of frame t = row0 + t * HW
    __syncthreads();  // the previous pixel is done with the tiles (first pass: the token / bias tiles are complete)
#pragma unroll 4
    for (int e = tid; e < T * 128; e += NTH) {  // k | v columns of the T rows
      const int t = e >> 7, c4 = e & 127;
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.qkv + (row0 + (long long)t * a.HW) * a.ldqkv + HID + c4 * 4);
      *reinterpret_cast<f32x4*>((c4 < 64 ? Ks : Vs) + sw(((c4 >> 3) & 7) * T + t, c4 & 7)) = v;
    }
    f32x2 q[16];
    if (act) {
      const float* qr = a.qkv + (row0 + (long long)i * a.HW) * a.ldqkv + head * DH;
#pragma unroll
      for (int c = 0; c < 8; ++c) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(qr + c * 4);
        q[2 * c] = (f32x2){v.x, v.y};
        q[2 * c + 1] = (f32x2){v.z, v.w};
      }
    }
    __syncthreads();
    if (act) {
      f32x2 acc[16];
#pragma unroll
      for (int c = 0; c < 16; ++c) acc[c] = (f32x2){0.f, 0.f};
      float m = -INFINITY, l = 0.f;
      auto key = [&](const float* kt, const float* vt, int r, float bias) {
        f32x2 kk[16];
        lds_row(kk, kt, r);
        const float s = dot32(q, kk) + bias;
        const float mn = fmaxf(m, s);
        const float corr = __expf(m - mn), p = __expf(s - mn);  // (first key: exp(-inf) = 0)
        l = l * corr + p;
        m = mn;
        f32x2 vv[16];
        lds_row(vv, vt, r);
        const f32x2 c2 = {corr, corr}, p2 = {p, p};
#pragma unroll
        for (int c = 0; c < 16; ++c) acc[c] = acc[c] * c2 + p2 * vv[c];
      };
      for (int j = 0; j < ntok; ++j) key(EK, EV, head * ntok + j, tok_bias ? Bh[j] : 0.f);
      for (int j = 0; j < T; ++j) key(Ks, Vs, head * T + j, Bh[j]);
      const float inv = 1.0f / l;
      const long long rq = row0 + (long long)i * a.HW;
      if (a.lse) a.lse[rq * HEADS + head] = m + logf(l);
      float* o = a.out + rq * a.ldo + head * DH;
#pragma unroll
      for (int c = 0; c < 8; ++c)
        *reinterpret_cast<f32x4*>(o + c * 4) = (f32x4){acc[2 * c].x * inv, acc[2 * c].y * inv, acc[2 * c + 1].x * inv, acc[2 * c + 1].y * inv};
    }
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
  static const bool use_valu = getenv("VMM_TEMPORAL_FWD_VALU") != nullptr;  // A/B switch: the VALU / LDS kernel of this file
  if (!use_valu) return vmm_temporal_attention_fwd_mfma_launch(qkv, ldqkv, ek, ev, ntok, bias, bias_on_cond, out, ldo, lse, B, T, HW, (hipStream_t)stream);
  TFArgs a{qkv, ek, ev, bias, out, lse, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0};
  a.blocks_per_sample = (int)max(1LL, min((long long)HW, cdiv(1536, B)));
  const size_t shm = sizeof(float) * (size_t)(2 * HEADS * T * DH + 2 * HEADS * ntok * DH + HEADS * T * T);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_fwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_attn_fwd_kernel, dim3((unsigned)(B * a.blocks_per_sample)), dim3(NTH), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
