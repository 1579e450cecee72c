// Temporal attention core, exact fp32 on the matrix cores (v_mfma_f32_16x16x4_f32), for the path that keeps qkv / out / logsumexp
// (training; inference outside the fused-kernel envelopes), gfx950.  heads = 8, dim_head = 32, T <= 32 frames (one or two tiles of 16),
// <= 16 tokens.
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
  const void* qkv;  // rows of ST: float, or bf16s when the feature maps are bf16-stored (the "bf16" mode's upper levels)
  const float *ek, *ev, *bias;
  void* out;
  float* lse;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample;
};

__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

__device__ __forceinline__ void load_row8(float (&dst)[8], const bf16s* p, bool ok) {
  uint4 u = make_uint4(0u, 0u, 0u, 0u);
  if (ok) u = *reinterpret_cast<const uint4*>(p);
  dst[0] = __uint_as_float(u.x << 16); dst[1] = __uint_as_float(u.x & 0xffff0000u);
  dst[2] = __uint_as_float(u.y << 16); dst[3] = __uint_as_float(u.y & 0xffff0000u);
  dst[4] = __uint_as_float(u.z << 16); dst[5] = __uint_as_float(u.z & 0xffff0000u);
  dst[6] = __uint_as_float(u.w << 16); dst[7] = __uint_as_float(u.w & 0xffff0000u);
}
__device__ __forceinline__ void load_row8(float (&dst)[8], const float* p, bool ok) {
  f32x4 u = {0.f, 0.f, 0.f, 0.f}, w = u;
  if (ok) {
    u = *reinterpret_cast<const f32x4*>(p);
    w = *reinterpret_cast<const f32x4*>(p + 4);
  }
  dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
  dst[4] = w.x; dst[5] = w.y; dst[6] = w.z; dst[7] = w.w;
}

// Raw forms for values requested one pixel ahead (NT == 1): the bits travel through the prefetch registers and are unpacked where the pixel is
// computed -- an unpack at the request is where the compiler waits for the load (LABNOTES 9.11).
template <typename ST> struct Raw8;
template <> struct Raw8<float> { f32x4 a, b; };
template <> struct Raw8<bf16s> { uint4 u; };
__device__ __forceinline__ void ldraw8(Raw8<float>& r, const float* p, bool ok) {
  r.a = f32x4{0.f, 0.f, 0.f, 0.f}; r.b = r.a;
  if (ok) { r.a = *reinterpret_cast<const f32x4*>(p); r.b = *reinterpret_cast<const f32x4*>(p + 4); }
}
__device__ __forceinline__ void ldraw8(Raw8<bf16s>& r, const bf16s* p, bool ok) {
  r.u = make_uint4(0u, 0u, 0u, 0u);
  if (ok) r.u = *reinterpret_cast<const uint4*>(p);
}
__device__ __forceinline__ void unpack8(const Raw8<float>& r, float (&d)[8]) {
  d[0] = r.a.x; d[1] = r.a.y; d[2] = r.a.z; d[3] = r.a.w; d[4] = r.b.x; d[5] = r.b.y; d[6] = r.b.z; d[7] = r.b.w;
}
__device__ __forceinline__ void unpack8(const Raw8<bf16s>& r, float (&d)[8]) {
  d[0] = __uint_as_float(r.u.x << 16); d[1] = __uint_as_float(r.u.x & 0xffff0000u);
  d[2] = __uint_as_float(r.u.y << 16); d[3] = __uint_as_float(r.u.y & 0xffff0000u);
  d[4] = __uint_as_float(r.u.z << 16); d[5] = __uint_as_float(r.u.z & 0xffff0000u);
  d[6] = __uint_as_float(r.u.w << 16); d[7] = __uint_as_float(r.u.w & 0xffff0000u);
}
__device__ __forceinline__ float2 ldraw2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ unsigned ldraw2(const bf16s* p) { return *reinterpret_cast<const unsigned*>(p); }
__device__ __forceinline__ float2 unpack2(const float2& v) { return v; }
__device__ __forceinline__ float2 unpack2(unsigned u) { return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u)); }

// NT = frame tiles of 16 (T <= 16 NT): queries and frame keys are walked tile by tile, the conditioning tokens are one more key tile
template <int NT, typename ST = float>
__global__ __launch_bounds__(512) void temporal_attn_fwd_mfma_kernel(const TFMArgs a) {
  const ST* const qkv = static_cast<const ST*>(a.qkv);
  ST* const outp = static_cast<ST*>(a.out);
  const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, c = lane & 15, g = lane >> 4;
  const int T = a.T, ntok = a.ntok;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool tok_bias = a.bias && a.bias_on_cond;
  float ekr[8], evc[2][4];
  load_row8(ekr, a.ek + ((long long)b * ntok + c) * HID + head * DH + 8 * g, c < ntok);
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) evc[h][r] = (4 * g + r < ntok) ? a.ev[((long long)b * ntok + 4 * g + r) * HID + head * DH + 2 * c + h] : 0.f;
  // (value / output columns: lane c of the two 16-column halves owns channels 2 c and 2 c + 1 of the head, so that a value is ONE 8-byte load
  // -- 4-byte for bf16-stored rows -- and an output ONE store per row, 16 lanes covering the head's 32 contiguous channels; two separate
  // element loads at c and c + 16 were half-line requests, and 2-byte ones with bf16 storage: 2.4 vs 1.6 ms at configs[3]'s C = 128 level)
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  // bias in the score layout, per (query tile, key tile): register r <-> key 16 jk + 4 g + r, lane <-> query 16 iq + c
  float bB[NT][NT][4];
#pragma unroll
  for (int iq = 0; iq < NT; ++iq)
#pragma unroll
    for (int jk = 0; jk < NT; ++jk)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int i = 16 * iq + c, j = 16 * jk + 4 * g + r;
        bB[iq][jk][r] = (a.bias && i < T && j < T) ? a.bias[((long long)head * T + i) * T + j] : 0.f;
      }

  // T <= 16 (the training shapes): the next pixel's k / v / q are requested during this pixel's arithmetic (a pixel was load latency + 50 MFMAs +
  // stores in sequence, the same for both waves of a SIMD: 3.85 TB/s)
  Raw8<ST> nk, nqr;
  decltype(ldraw2(qkv)) nv[4];
  bool nvok[4] = {false, false, false, false};
  auto request = [&](int pn) {
    const bool okp = pn < a.HW;
    const long long r0 = (long long)b * T * a.HW + min(pn, a.HW - 1);
    ldraw8(nk, qkv + (r0 + (long long)c * a.HW) * a.ldqkv + HID + head * DH + 8 * g, okp && c < T);
    ldraw8(nqr, qkv + (r0 + (long long)c * a.HW) * a.ldqkv + head * DH + 8 * g, okp && c < T);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int tv = 4 * g + r;
      nvok[r] = okp && tv < T;
      nv[r] = ldraw2(qkv + (r0 + (long long)min(tv, T - 1) * a.HW) * a.ldqkv + 2 * HID + head * DH + 2 * c);
    }
  };
  if constexpr (NT == 1) request(blk);

  for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
    const long long row0 = (long long)b * T * a.HW + pix;
    // keys / values of every frame tile: k in row layout (lane = frame 16 jk + c), v in column layout (register = frame 16 jk + 4 g + r)
    float kr[NT][8], vc[NT][2][4];
    float qpre[8];
    if constexpr (NT == 1) {
      unpack8(nk, kr[0]);
      unpack8(nqr, qpre);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float2 vp = unpack2(nv[r]);
        vc[0][0][r] = nvok[r] ? vp.x : 0.f;
        vc[0][1][r] = nvok[r] ? vp.y : 0.f;
      }
      request(pix + a.blocks_per_sample);
    }
#pragma unroll
    for (int jk = 0; jk < (NT == 1 ? 0 : NT); ++jk) {
      const int tk = 16 * jk + c;
      load_row8(kr[jk], qkv + (row0 + (long long)tk * a.HW) * a.ldqkv + HID + head * DH + 8 * g, tk < T);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int tv = 16 * jk + 4 * g + r;
        const bool ok = tv < T;
        const float2 vp = ld2(qkv + (row0 + (long long)(ok ? tv : 0) * a.HW) * a.ldqkv + 2 * HID + head * DH + 2 * c);
        vc[jk][0][r] = ok ? vp.x : 0.f;
        vc[jk][1][r] = ok ? vp.y : 0.f;
      }
    }
#pragma unroll
    for (int iq = 0; iq < NT; ++iq) {
      const int ti = 16 * iq + c;  // this lane's query frame
      const bool qok = ti < T;
      if (NT > 1 && 16 * iq >= T) break;
      const long long rq = row0 + (long long)ti * a.HW;
      float qr[8];
      if constexpr (NT == 1) {
#pragma unroll
        for (int j = 0; j < 8; ++j) qr[j] = qpre[j];
      } else {
        load_row8(qr, qkv + rq * a.ldqkv + head * DH + 8 * g, qok);
      }
      f32x4 S[NT], St = zero4;
#pragma unroll
      for (int jk = 0; jk < NT; ++jk) S[jk] = zero4;
#pragma unroll
      for (int s = 0; s < 8; ++s) {
#pragma unroll
        for (int jk = 0; jk < NT; ++jk) S[jk] = mm(kr[jk][s], qr[s], S[jk]);  // S^T[j][i]: register = key 16 jk + 4 g + r, lane = query
        St = mm(ekr[s], qr[s], St);
      }
      float sv[NT][4], st[4];
      float m = -INFINITY;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int jk = 0; jk < NT; ++jk) {
          const int j = 16 * jk + 4 * g + r;
          sv[jk][r] = j < T ? S[jk][r] + bB[iq][jk][r] : -INFINITY;
          m = fmaxf(m, sv[jk][r]);
        }
        const int tt = 4 * g + r;
        st[r] = tt < ntok ? St[r] + (tok_bias ? bB[iq][0][r] : 0.f) : -INFINITY;  // (token t shares the bias column of frame t: ntok <= 16)
        m = fmaxf(m, st[r]);
      }
      m = fmaxf(m, __shfl_xor(m, 16, 64));
      m = fmaxf(m, __shfl_xor(m, 32, 64));
      float l = 0.f;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int jk = 0; jk < NT; ++jk) { sv[jk][r] = __expf(sv[jk][r] - m); l += sv[jk][r]; }
        st[r] = __expf(st[r] - m);
        l += st[r];
      }
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
      const float inv = 1.0f / l;
      f32x4 O[2] = {zero4, zero4};
#pragma unroll
      for (int h = 0; h < 2; ++h)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int jk = 0; jk < NT; ++jk) O[h] = mm(sv[jk][r] * inv, vc[jk][h][r], O[h]);
          O[h] = mm(st[r] * inv, evc[h][r], O[h]);
        }
      if (a.lse && g == 0 && qok) a.lse[rq * HEADS + head] = m + logf(l);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int to = 16 * iq + 4 * g + r;
        if (to < T) st2(outp + (row0 + (long long)to * a.HW) * a.ldo + head * DH + 2 * c, O[0][r], O[1][r]);
      }
    }
  }
}

}  // namespace

// Fast path of vmm_temporal_attention (same arguments and results).  Returns 1 (nothing launched) outside its envelope: heads = 8,
// dim_head = 32, T <= 32, ntok <= 16, ntok <= T when the bias also covers the tokens.
extern "C" int vmm_temporal_attention_staged(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                             int32_t bias_on_cond, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads,
                                             int32_t dh, float* lse, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 32 || T < 1 || ntok > 16 || (ldqkv & 3) || (ldo & 3)) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (B <= 0 || HW <= 0) return 0;
  TFMArgs a{qkv, ek, ev, bias, out, lse, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0};
  a.blocks_per_sample = (int)max(1LL, min((long long)HW, cdiv(1024, B)));
  if (T <= 16) hipLaunchKernelGGL(temporal_attn_fwd_mfma_kernel<1>, dim3((unsigned)(B * a.blocks_per_sample)), dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(temporal_attn_fwd_mfma_kernel<2>, dim3((unsigned)(B * a.blocks_per_sample)), dim3(512), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

// The same core over bf16-STORED qkv rows and output (the "bf16" mode's C = 128 level with more than 16 frames): fp32 arithmetic, half the bytes.
// Envelope of the fast path above (returns 1 outside it; no thread-per-query fallback for this storage).
extern "C" int vmm_temporal_attention_a16(const void* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                          int32_t bias_on_cond, void* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh,
                                          float* lse, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 32 || T < 1 || ntok > 16 || (ldqkv & 7) || (ldo & 3)) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (bias_on_cond && ek && ntok != T) return -2;
  if (B <= 0 || HW <= 0) return 0;
  TFMArgs a{qkv, ek, ev, bias, out, lse, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0};
  a.blocks_per_sample = (int)max(1LL, min((long long)HW, cdiv(1024, B)));
  if (T <= 16) hipLaunchKernelGGL((temporal_attn_fwd_mfma_kernel<1, bf16s>), dim3((unsigned)(B * a.blocks_per_sample)), dim3(512), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL((temporal_attn_fwd_mfma_kernel<2, bf16s>), dim3((unsigned)(B * a.blocks_per_sample)), dim3(512), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
