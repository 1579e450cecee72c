// Backward of the temporal attention core (training path, vddp.py:397-466 under autograd), gfx950.  heads = 8, dim_head = 32,
// T <= 16 frames, <= 16 conditioning tokens.
//
// One workgroup = one pixel at a time, 8 heads x 16 lanes; lane i of a head group is query i in pass A and key i in pass B:
//   staging   the pixel's T rows of qkv, dO and O are read once, coalesced (3 KB / 1 KB contiguous per row), into LDS; D_i = dO_i . O_i
//             is reduced while staging (8 lanes per head slice)
//   pass A    p_ij = exp(q_i . k_j + bias_ij - L_i),  ds_ij = p_ij (dO_i . v_j - D_i),  dq_i = sum_j ds_ij k_j      (k_j, v_j: LDS broadcast)
//   pass B    dk_j = sum_i ds_ij q_i,  dv_j = sum_i p_ij dO_i                                                       (q_i, dO_i: LDS broadcast)
// with p / ds handed from A to B through a 16 x 16 LDS tile per head.  The conditioning tokens go through the same two passes first
// (keys from global memory, shared by every pixel of the sample); their key / value gradients stay in registers across the
// workgroup's pixels and reach memory as one set of atomics per workgroup, the bias gradient likewise through an LDS tile.
// Every element of qkv / dO / O is read from HBM once and every element of dqkv written once (the thread-per-query / wave-per-key
// kernels of attention_bwd.hip re-read each row T times); the rotary rotation and q-scale of the projection epilogue are undone
// on the way out, so dqkv is the gradient of the raw to_qkv output.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32, HEADS = 8, HID = HEADS * DH, NTH = HEADS * 16;

struct TBArgs {
  const float *qkv, *ek, *ev, *bias, *O, *dO, *lse, *rot;
  float *dqkv, *dek, *dev, *dbias;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample;
  float q_scale;
};

// rows of 32 floats, 16-byte chunk c of row r stored at chunk c ^ (r & 7): own-row reads of the 16 lanes of a group spread over the banks
__device__ __forceinline__ int sw(int r, int c) { return r * DH + ((c ^ (r & 7)) << 2); }

__device__ __forceinline__ void lds_row(float (&dst)[DH], const float* base, int r) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + sw(r, c));
    dst[c * 4] = v.x; dst[c * 4 + 1] = v.y; dst[c * 4 + 2] = v.z; dst[c * 4 + 3] = v.w;
  }
}
__device__ __forceinline__ void glb_row(float (&dst)[DH], const float* src) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * 4);
    dst[c * 4] = v.x; dst[c * 4 + 1] = v.y; dst[c * 4 + 2] = v.z; dst[c * 4 + 3] = v.w;
  }
}
__device__ __forceinline__ float dot32(const float (&a)[DH], const float (&b)[DH]) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int i = 0; i < DH; i += 4) { s0 = fmaf(a[i], b[i], s0); s1 = fmaf(a[i + 1], b[i + 1], s1); s2 = fmaf(a[i + 2], b[i + 2], s2); s3 = fmaf(a[i + 3], b[i + 3], s3); }
  return (s0 + s1) + (s2 + s3);
}
// transpose of the interleaved-pair rotation by position pos, then scale; 128-byte row store
__device__ __forceinline__ void unrotate_store(float* dst, float (&g)[DH], const float* __restrict__ tab, int pos, float scale) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f32x4 o = {g[c * 4], g[c * 4 + 1], g[c * 4 + 2], g[c * 4 + 3]};
    if (tab) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + (pos * (DH / 2) + c * 2) * 2);  // cos0 sin0 cos1 sin1
      o = (f32x4){g[c * 4] * cs.x + g[c * 4 + 1] * cs.y, g[c * 4 + 1] * cs.x - g[c * 4] * cs.y,
                  g[c * 4 + 2] * cs.z + g[c * 4 + 3] * cs.w, g[c * 4 + 3] * cs.z - g[c * 4 + 2] * cs.w};
    }
    o.x *= scale; o.y *= scale; o.z *= scale; o.w *= scale;
    *reinterpret_cast<f32x4*>(dst + c * 4) = o;
  }
}

__global__ __launch_bounds__(NTH) void temporal_attn_bwd_kernel(const TBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, ntok = a.ntok;
  float* Qs = smem;                       // [HEADS][T][32] (swizzled chunks)
  float* Ks = Qs + HEADS * T * DH;
  float* Vs = Ks + HEADS * T * DH;
  float* Gs = Vs + HEADS * T * DH;        // dO
  float* Pm = Gs + HEADS * T * DH;        // [HEADS][16 queries][16 keys]
  float* Dm = Pm + HEADS * 256;
  float* Bacc = Dm + HEADS * 256;         // bias-gradient accumulator [HEADS][16][16]
  float* Dsum = Bacc + HEADS * 256;       // [HEADS][16]
  float* Bs = Dsum + HEADS * 16;          // bias [HEADS][T][T]
  const int tid = threadIdx.x, head = tid >> 4, i = tid & 15;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool act = i < T, tact = i < ntok;
  const bool tok_bias = a.bias && a.bias_on_cond;
  for (int e = tid; e < HEADS * 256; e += NTH) Bacc[e] = 0.f;
  if (a.bias)
    for (int e = tid; e < HEADS * T * T; e += NTH) Bs[e] = a.bias[e];
  float kacc[DH], vacc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { kacc[d] = 0.f; vacc[d] = 0.f; }
  float* Pme = Pm + (head * 16 + i) * 16;   // row i as a query
  float* Dme = Dm + (head * 16 + i) * 16;
  float* Bme = Bacc + (head * 16 + i) * 16;

  for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
    const long long row0 = (long long)b * T * a.HW + pix;  // row of frame t = row0 + t * HW
    __syncthreads();  // the previous pixel's pass B is done with the tiles
    // ---- stage q | k | v
    constexpr int NQ4 = 3 * HID / 4;
#pragma unroll 4
    for (int e = tid; e < T * NQ4; e += NTH) {
      const int t = e / NQ4, c4 = e - t * NQ4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.qkv + (row0 + (long long)t * a.HW) * a.ldqkv + c4 * 4);
      const int which = c4 >> 6, h = (c4 >> 3) & 7;
      float* base = which == 0 ? Qs : which == 1 ? Ks : Vs;
      *reinterpret_cast<f32x4*>(base + sw(h * T + t, c4 & 7)) = v;
    }
    // ---- stage dO, D = dO . O
#pragma unroll 2
    for (int e = tid; e < T * (HID / 4); e += NTH) {
      const int t = e >> 6, c4 = e & 63, h = c4 >> 3;
      const long long off = (row0 + (long long)t * a.HW) * a.ldo + c4 * 4;
      const f32x4 g = *reinterpret_cast<const f32x4*>(a.dO + off);
      const f32x4 o = *reinterpret_cast<const f32x4*>(a.O + off);
      *reinterpret_cast<f32x4*>(Gs + sw(h * T + t, c4 & 7)) = g;
      float s = (g.x * o.x + g.y * o.y) + (g.z * o.z + g.w * o.w);
      s += __shfl_xor(s, 1, 64);
      s += __shfl_xor(s, 2, 64);
      s += __shfl_xor(s, 4, 64);
      if ((c4 & 7) == 0) Dsum[h * 16 + t] = s;
    }
    __syncthreads();
    float q[DH], go[DH], dq[DH];
    float L = 0.f, Dv = 0.f;
    if (act) {
      lds_row(q, Qs, head * T + i);
      lds_row(go, Gs, head * T + i);
      L = a.lse[(row0 + (long long)i * a.HW) * HEADS + head];
      Dv = Dsum[head * 16 + i];
    } else {
#pragma unroll
      for (int d = 0; d < DH; ++d) { q[d] = 0.f; go[d] = 0.f; }
    }
#pragma unroll
    for (int d = 0; d < DH; ++d) dq[d] = 0.f;
    // ---- conditioning tokens: pass A (lane = query), pass B (lane = token)
    if (ntok > 0) {
      for (int j = 0; j < ntok; ++j) {
        float kk[DH], vv[DH];
        glb_row(kk, a.ek + ((long long)b * ntok + j) * HID + head * DH);
        glb_row(vv, a.ev + ((long long)b * ntok + j) * HID + head * DH);
        float s = dot32(q, kk);
        if (tok_bias && act) s += Bs[(head * T + i) * T + j];
        const float p = act ? __expf(s - L) : 0.f;
        const float ds = p * (dot32(go, vv) - Dv);
#pragma unroll
        for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, kk[d], dq[d]);
        Pme[j] = p;
        Dme[j] = ds;
        if (tok_bias) Bme[j] += ds;
      }
      __syncthreads();
      if (tact) {
        for (int ii = 0; ii < T; ++ii) {
          const float p = Pm[(head * 16 + ii) * 16 + i], ds = Dm[(head * 16 + ii) * 16 + i];
          float r[DH];
          lds_row(r, Qs, head * T + ii);
#pragma unroll
          for (int d = 0; d < DH; ++d) kacc[d] = fmaf(ds, r[d], kacc[d]);
          lds_row(r, Gs, head * T + ii);
#pragma unroll
          for (int d = 0; d < DH; ++d) vacc[d] = fmaf(p, r[d], vacc[d]);
        }
      }
      __syncthreads();
    }
    // ---- frames: pass A
    for (int j = 0; j < T; ++j) {
      float r[DH];
      lds_row(r, Ks, head * T + j);
      float s = dot32(q, r);
      if (a.bias && act) s += Bs[(head * T + i) * T + j];
      const float p = act ? __expf(s - L) : 0.f;
      float vv[DH];
      lds_row(vv, Vs, head * T + j);
      const float ds = p * (dot32(go, vv) - Dv);
#pragma unroll
      for (int d = 0; d < DH; ++d) dq[d] = fmaf(ds, r[d], dq[d]);
      Pme[j] = p;
      Dme[j] = ds;
      if (a.bias) Bme[j] += ds;
    }
    if (act) unrotate_store(a.dqkv + (row0 + (long long)i * a.HW) * a.ldqkv + head * DH, dq, a.rot, i, a.q_scale);
    __syncthreads();
    // ---- frames: pass B (lane = key)
    if (act) {
      float dk[DH], dv[DH];
#pragma unroll
      for (int d = 0; d < DH; ++d) { dk[d] = 0.f; dv[d] = 0.f; }
      for (int ii = 0; ii < T; ++ii) {
        const float p = Pm[(head * 16 + ii) * 16 + i], ds = Dm[(head * 16 + ii) * 16 + i];
        float r[DH];
        lds_row(r, Qs, head * T + ii);
#pragma unroll
        for (int d = 0; d < DH; ++d) dk[d] = fmaf(ds, r[d], dk[d]);
        lds_row(r, Gs, head * T + ii);
#pragma unroll
        for (int d = 0; d < DH; ++d) dv[d] = fmaf(p, r[d], dv[d]);
      }
      float* o = a.dqkv + (row0 + (long long)i * a.HW) * a.ldqkv + head * DH;
      unrotate_store(o + HID, dk, a.rot, i, 1.0f);
      unrotate_store(o + 2 * HID, dv, nullptr, 0, 1.0f);
    }
  }
  // ---- gradients shared by the workgroup's pixels: one set of atomics.  (The rotation of the token keys is undone by the caller --
  // these are gradients of the rotated ek the forward consumed, as in attention_bwd.hip.)
  if (tact && a.dek) {
    float* kd = a.dek + ((long long)b * ntok + i) * HID + head * DH;
    float* vd = a.dev + ((long long)b * ntok + i) * HID + head * DH;
#pragma unroll
    for (int d = 0; d < DH; ++d) { atomicAdd(kd + d, kacc[d]); atomicAdd(vd + d, vacc[d]); }
  }
  if (a.bias && a.dbias && act) {  // (own accumulator row)
    for (int j = 0; j < T; ++j) atomicAdd(&a.dbias[((long long)head * T + i) * T + j], Bme[j]);
  }
}

}  // namespace

// Fast path of vmm_attention_bwd for mode 0 (same arguments and results; dbuf is not needed).  Returns 1 (nothing launched) outside
// its envelope: heads = 8, dim_head = 32, T <= 16, ntok <= 16 shared tokens, ntok <= T when the bias also covers the tokens.
extern "C" int vmm_temporal_attention_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                          int32_t bias_on_cond, const float* out, const float* dout, int32_t ldo, const float* lse,
                                          const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev, float* dbias, int32_t B,
                                          int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 16 || T < 1 || ntok > 16 || (ldqkv & 3) || (ldo & 3)) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (ntok > 0 && (!dek || !dev)) return -1;
  if (B <= 0 || HW <= 0) return 0;
  TBArgs a{qkv, ek, ev, bias, out, dout, lse, rot_tab, dqkv, dek, dev, dbias, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0, q_scale};
  a.blocks_per_sample = (int)max(1LL, min((long long)HW, cdiv(1024, B)));
  const size_t shm = sizeof(float) * (size_t)(4 * HEADS * T * DH + 3 * HEADS * 256 + HEADS * 16 + HEADS * T * T);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_attn_bwd_kernel, dim3((unsigned)(B * a.blocks_per_sample)), dim3(NTH), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
