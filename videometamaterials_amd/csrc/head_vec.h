// A head slice (dim_head floats of one attention head) in registers, for the thread-per-query / thread-per-key attention kernels that
// serve every dim_head the reference's constructor accepts here (attn_dim_head, vddp.py:582, 615: the temporal attentions; a multiple of 4,
// at most 128).  DM = compile-time capacity of the register arrays, EX = the slice fills it exactly (dh == DM: no predicate anywhere, the
// instruction stream of the dim_head = 32 kernels these helpers were lifted from).  Otherwise the 16-byte pieces beyond dh are never
// loaded (the registers hold zeros, which every product and sum below passes through unchanged) and never stored.
#pragma once
#include "vmm_common.h"

template <int DM, bool EX>
struct HeadVec {
  static_assert(DM % 4 == 0, "head slices move in 16-byte pieces");
  static __device__ __forceinline__ bool on(int i4, int dh) { return EX || 4 * i4 < dh; }

  static __device__ __forceinline__ void ld(float (&dst)[DM], const float* src, int dh) {
#pragma unroll
    for (int i = 0; i < DM / 4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (on(i, dh)) v = *reinterpret_cast<const f32x4*>(src + i * 4);
      dst[i * 4] = v.x; dst[i * 4 + 1] = v.y; dst[i * 4 + 2] = v.z; dst[i * 4 + 3] = v.w;
    }
  }
  static __device__ __forceinline__ void st(float* dst, const float (&src)[DM], int dh) {
#pragma unroll
    for (int i = 0; i < DM / 4; ++i)
      if (on(i, dh)) *reinterpret_cast<f32x4*>(dst + i * 4) = (f32x4){src[i * 4], src[i * 4 + 1], src[i * 4 + 2], src[i * 4 + 3]};
  }
  static __device__ __forceinline__ void zero(float (&dst)[DM]) {
#pragma unroll
    for (int i = 0; i < DM; ++i) dst[i] = 0.f;
  }
  // registers . memory (four partial sums, as the dim_head = 32 kernels always summed)
  static __device__ __forceinline__ float dot(const float (&a)[DM], const float* b, int dh) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < DM / 4; ++i) {
      if (on(i, dh)) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(b + i * 4);
        s0 = fmaf(a[i * 4 + 0], v.x, s0); s1 = fmaf(a[i * 4 + 1], v.y, s1);
        s2 = fmaf(a[i * 4 + 2], v.z, s2); s3 = fmaf(a[i * 4 + 3], v.w, s3);
      }
    }
    return (s0 + s1) + (s2 + s3);
  }
  // registers . registers (the pieces beyond dh are zeros)
  static __device__ __forceinline__ float dotr(const float (&a)[DM], const float (&b)[DM]) {
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
    for (int i = 0; i < DM; i += 4) {
      s0 = fmaf(a[i], b[i], s0); s1 = fmaf(a[i + 1], b[i + 1], s1); s2 = fmaf(a[i + 2], b[i + 2], s2); s3 = fmaf(a[i + 3], b[i + 3], s3);
    }
    return (s0 + s1) + (s2 + s3);
  }
  // transpose of the interleaved-pair rotation by position `pos`; tab [positions][dh/2][2] (cos, sin), identity pairs beyond the rotary span
  static __device__ __forceinline__ void unrotate(float (&g)[DM], const float* __restrict__ tab, int pos, int dh) {
    const int half = EX ? DM / 2 : dh >> 1;
#pragma unroll
    for (int f = 0; f < DM / 2; ++f) {
      if (EX || 2 * f < dh) {
        const float c = tab[(pos * half + f) * 2], s = tab[(pos * half + f) * 2 + 1];
        const float a = g[2 * f], b = g[2 * f + 1];
        g[2 * f] = a * c + b * s;
        g[2 * f + 1] = b * c - a * s;
      }
    }
  }
};

// Instances: the reference's default (32), its two neighbours exactly, and masked catch-alls for every other multiple of 4 up to 128.
#define VMM_HEADVEC_DISPATCH(dh, CALL)                   \
  do {                                                   \
    if ((dh) == 32) { CALL(32, true); }                  \
    else if ((dh) == 16) { CALL(16, true); }             \
    else if ((dh) == 64) { CALL(64, true); }             \
    else if ((dh) < 32) { CALL(32, false); }             \
    else if ((dh) < 64) { CALL(64, false); }             \
    else { CALL(128, false); }                           \
  } while (0)

static inline bool vmm_head_dim_ok(int dh) { return dh >= 4 && dh <= 128 && (dh & 3) == 0; }
