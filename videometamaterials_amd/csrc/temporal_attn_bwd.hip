// Backward of the temporal attention core (training path, vddp.py:397-466 under autograd), gfx950.  heads = 8, dim_head = 32,
// T <= 16 frames, <= 16 conditioning tokens.
//
// One workgroup = one pixel at a time; two waves = 8 heads x 16 lanes for the frames, two more for the conditioning tokens; lane i of a
// head group is key i in pass 1 and query i in pass 2:
//   staging   the pixel's T rows of q, k, dO and O are read once, coalesced, into LDS; D_i = dO_i . O_i is reduced while staging
//             (8 lanes per head slice); v_j goes straight into its key lane's registers
//   pass 1    lane = key j (own k_j, v_j in registers; q_i, dO_i: LDS broadcast), one sweep over the queries:
//             p_ij = exp(q_i . k_j + bias_ij - L_i),  ds_ij = p_ij (dO_i . v_j - D_i),  dk_j += ds_ij q_i,  dv_j += p_ij dO_i
//   pass 2    lane = query i:  dq_i = sum_j ds_ij k_j   (k_j: LDS broadcast, ds through a T x (ntok + T) LDS tile per head)
// The conditioning tokens are keys of pass 1 as well (lane = token, keys from global memory, shared by every pixel of the sample);
// their key / value gradients stay in registers across the workgroup's pixels and leave as one partial per workgroup, the bias
// gradient likewise through an LDS tile; a small second kernel sums the partials in a fixed order (no atomics).
// Every element of qkv / dO / O is read from HBM once and every element of dqkv written once (the thread-per-query / wave-per-key
// kernels of attention_bwd.hip re-read each row T times); the rotary rotation and q-scale of the projection epilogue are undone
// on the way out, so dqkv is the gradient of the raw to_qkv output.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

#include <cstdlib>

// the matrix-core version (temporal_attn_bwd_mfma.hip); same partial-buffer layout, reduced by the kernel at the end of this file
int vmm_temporal_attention_bwd_mfma_launch(const float* qkv, int ldqkv, const float* ek, const float* ev, int ntok, const float* bias, int bias_on_cond,
                                           const float* out, const float* dout, int ldo, const float* lse, const float* rot_tab, float q_scale,
                                           float* dqkv, float* scratch, int B, int T, int HW, int blocks_per_sample, int pstride, hipStream_t s);

namespace {
constexpr int DH = 32, HEADS = 8, HID = HEADS * DH, NTH = HEADS * 16;
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct TBArgs {
  const float *qkv, *ek, *ev, *bias, *O, *dO, *lse, *rot;
  float *dqkv, *part;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample, pstride;
  float q_scale;
};

// rows of 32 floats, 16-byte chunk c of row r stored at chunk c ^ (r & 7): own-row reads of the 16 lanes of a group spread over the banks
__device__ __forceinline__ int sw(int r, int c) { return r * DH + ((c ^ (r & 7)) << 2); }

// a row as 16 float pairs: the dot products and axpys below are written on pairs so that they compile to v_pk_fma_f32
__device__ __forceinline__ void lds_row(f32x2 (&dst)[16], const float* base, int r) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(base + sw(r, c));
    dst[2 * c] = (f32x2){v.x, v.y};
    dst[2 * c + 1] = (f32x2){v.z, v.w};
  }
}
__device__ __forceinline__ void glb_row(f32x2 (&dst)[16], const float* src) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + c * 4);
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
__device__ __forceinline__ void axpy32(f32x2 (&acc)[16], float w, const f32x2 (&x)[16]) {
  const f32x2 w2 = {w, w};
#pragma unroll
  for (int c = 0; c < 16; ++c) acc[c] = w2 * x[c] + acc[c];
}
__device__ __forceinline__ void zero32(f32x2 (&x)[16]) {
#pragma unroll
  for (int c = 0; c < 16; ++c) x[c] = (f32x2){0.f, 0.f};
}
// transpose of the interleaved-pair rotation by position pos, then scale; 128-byte row store
__device__ __forceinline__ void unrotate_store(float* dst, const f32x2 (&g)[16], const float* __restrict__ tab, int pos, float scale) {
#pragma unroll
  for (int c = 0; c < 8; ++c) {
    f32x4 o = {g[2 * c].x, g[2 * c].y, g[2 * c + 1].x, g[2 * c + 1].y};
    if (tab) {
      const f32x4 cs = *reinterpret_cast<const f32x4*>(tab + (pos * (DH / 2) + c * 2) * 2);  // cos0 sin0 cos1 sin1
      o = (f32x4){g[2 * c].x * cs.x + g[2 * c].y * cs.y, g[2 * c].y * cs.x - g[2 * c].x * cs.y,
                  g[2 * c + 1].x * cs.z + g[2 * c + 1].y * cs.w, g[2 * c + 1].y * cs.z - g[2 * c + 1].x * cs.w};
    }
    o.x *= scale; o.y *= scale; o.z *= scale; o.w *= scale;
    *reinterpret_cast<f32x4*>(dst + c * 4) = o;
  }
}
__device__ __forceinline__ void store32(float* dst, const f32x2 (&g)[16]) {
#pragma unroll
  for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(dst + c * 4) = (f32x4){g[2 * c].x, g[2 * c].y, g[2 * c + 1].x, g[2 * c + 1].y};
}

__global__ __launch_bounds__(2 * NTH) void temporal_attn_bwd_kernel(const TBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int T = a.T, ntok = a.ntok, NK = ntok + T;
  float* Qs = smem;                       // [HEADS][T][32] (swizzled chunks); after pass 1: the token part of dq
  float* Ks = Qs + HEADS * T * DH;
  float* Gs = Ks + HEADS * T * DH;        // dO
  float* DSm = Gs + HEADS * T * DH;       // ds [HEADS][T queries][ntok + T keys]
  float* Bacc = DSm + HEADS * T * NK;     // bias-gradient accumulator [HEADS][T][T]
  float* Bs = Bacc + HEADS * T * T;       // bias [HEADS][T][T]
  float* Ls = Bs + HEADS * T * T;         // logsumexp [HEADS][16]
  float* Dsum = Ls + HEADS * 16;          // D = dO . O [HEADS][16]
  // waves 0-1: frame keys (and the final dq); waves 2-3 (launched only when there are conditioning tokens): token keys
  const int tid = threadIdx.x, nth = blockDim.x, role = tid >> 7, head = (tid >> 4) & 7, i = tid & 15;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool act = i < T, tact = role == 1 && i < ntok;
  const bool tok_bias = a.bias && a.bias_on_cond;
  for (int e = tid; e < HEADS * T * T; e += nth) {
    Bacc[e] = 0.f;
    Bs[e] = a.bias ? a.bias[e] : 0.f;
  }
  const float* Bh = Bs + head * T * T;
  float* Bah = Bacc + head * T * T;
  float* DSh = DSm + head * T * NK;
  auto stage = [&](long long row0) {
    // ---- q | k (v is only ever needed by its own key lane: read straight into registers)
#pragma unroll 4
    for (int e = tid; e < T * 128; e += nth) {
      const int t = e >> 7, c4 = e & 127;
      const f32x4 v = *reinterpret_cast<const f32x4*>(a.qkv + (row0 + (long long)t * a.HW) * a.ldqkv + c4 * 4);
      *reinterpret_cast<f32x4*>((c4 < 64 ? Qs : Ks) + sw(((c4 >> 3) & 7) * T + t, c4 & 7)) = v;
    }
    // ---- dO, D = dO . O
#pragma unroll 2
    for (int e = tid; e < T * (HID / 4); e += nth) {
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
  };
  // The two roles run the same sequence of barriers per pixel (stage | pass 1 | pass 2 | hand-over of the token part of dq).
  if (role == 1) {
    // ================= token waves: lane = (head, token i) in pass 1, (head, query i) in pass 2
    f32x2 ekk[16], evv[16], kacc[16], vacc[16];  // own key / value (the same for every pixel), gradients summed over this workgroup's pixels
    zero32(kacc);
    zero32(vacc);
    if (tact) {
      glb_row(ekk, a.ek + ((long long)b * ntok + i) * HID + head * DH);
      glb_row(evv, a.ev + ((long long)b * ntok + i) * HID + head * DH);
    }
    for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
      const long long row0 = (long long)b * T * a.HW + pix;  // row of frame t = row0 + t * HW
      __syncthreads();  // the previous pixel is done with the tiles
      stage(row0);
      __syncthreads();
      if (tact) {
        for (int ii = 0; ii < T; ++ii) {
          f32x2 r[16], g[16];
          lds_row(r, Qs, head * T + ii);
          float s = dot32(r, ekk);
          if (tok_bias) s += Bh[ii * T + i];
          const float p = __expf(s - Ls[head * 16 + ii]);
          lds_row(g, Gs, head * T + ii);
          const float ds = p * (dot32(g, evv) - Dsum[head * 16 + ii]);
          axpy32(kacc, ds, r);
          axpy32(vacc, p, g);
          DSh[ii * NK + i] = ds;
        }
      }
      __syncthreads();
      if (act) {  // token part of dq_i, left in the (now free) q tile
        f32x2 dq[16];
        zero32(dq);
        for (int j = 0; j < ntok; ++j) {
          f32x2 r[16];
          glb_row(r, a.ek + ((long long)b * ntok + j) * HID + head * DH);
          axpy32(dq, DSh[i * NK + j], r);
        }
#pragma unroll
        for (int c = 0; c < 8; ++c) *reinterpret_cast<f32x4*>(Qs + sw(head * T + i, c)) = (f32x4){dq[2 * c].x, dq[2 * c].y, dq[2 * c + 1].x, dq[2 * c + 1].y};
      }
      __syncthreads();
    }
    // gradients shared by the workgroup's pixels: one partial per workgroup, summed in a fixed order by the reduce kernel below
    if (tact) {
      float* part = a.part + (long long)blockIdx.x * a.pstride;
      store32(part + i * HID + head * DH, kacc);
      store32(part + ntok * HID + i * HID + head * DH, vacc);
    }
  } else {
    // ================= frame waves: lane = (head, frame i): key i in pass 1, query i in pass 2
    for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
      const long long row0 = (long long)b * T * a.HW + pix;
      __syncthreads();
      stage(row0);
      if (act) Ls[head * 16 + i] = a.lse[(row0 + (long long)i * a.HW) * HEADS + head];
      __syncthreads();
      // pass 1: s, p, ds against every query (q_i, dO_i: LDS broadcast) and the key's own gradients in one sweep
      if (act) {
        f32x2 kk[16], vv[16], dk[16], dv[16];
        lds_row(kk, Ks, head * T + i);
        float* orow = a.dqkv + (row0 + (long long)i * a.HW) * a.ldqkv + head * DH;
        glb_row(vv, a.qkv + (row0 + (long long)i * a.HW) * a.ldqkv + 2 * HID + head * DH);
        zero32(dk);
        zero32(dv);
        for (int ii = 0; ii < T; ++ii) {
          f32x2 r[16], g[16];
          lds_row(r, Qs, head * T + ii);
          float s = dot32(r, kk);
          if (a.bias) s += Bh[ii * T + i];
          const float p = __expf(s - Ls[head * 16 + ii]);
          lds_row(g, Gs, head * T + ii);
          const float ds = p * (dot32(g, vv) - Dsum[head * 16 + ii]);
          axpy32(dk, ds, r);
          axpy32(dv, p, g);
          DSh[ii * NK + ntok + i] = ds;
        }
        unrotate_store(orow + HID, dk, a.rot, i, 1.0f);
        store32(orow + 2 * HID, dv);
      }
      __syncthreads();
      // pass 2: dq_i = sum_j ds_ij k_j over the frame keys (LDS broadcast), plus the bias gradient of row i
      f32x2 dq[16];
      zero32(dq);
      if (act) {
        for (int j = 0; j < T; ++j) {
          f32x2 r[16];
          lds_row(r, Ks, head * T + j);
          const float ds = DSh[i * NK + ntok + j];
          axpy32(dq, ds, r);
          if (a.bias) Bah[i * T + j] += ds + (tok_bias && j < ntok ? DSh[i * NK + j] : 0.f);
        }
      }
      if (ntok > 0) __syncthreads();
      if (act) {
        if (ntok > 0) {
          f32x2 r[16];
          lds_row(r, Qs, head * T + i);
#pragma unroll
          for (int c = 0; c < 16; ++c) dq[c] += r[c];
        }
        unrotate_store(a.dqkv + (row0 + (long long)i * a.HW) * a.ldqkv + head * DH, dq, a.rot, i, a.q_scale);
      }
    }
  }
  __syncthreads();
  float* part = a.part + (long long)blockIdx.x * a.pstride;
  for (int e = tid; e < HEADS * T * T; e += nth) part[2 * ntok * HID + e] = Bacc[e];
}

// dek / dev [B][ntok][HID] += sum over the sample's workgroups, dbias [HEADS][T][T] += sum over all workgroups.
// Workgroup = 16 consecutive elements x 16 slices of the partials; fixed summation order.
__global__ __launch_bounds__(256) void temporal_attn_bwd_reduce_kernel(const float* __restrict__ part, int pstride, int bps, int B, int ntok, int T,
                                                                      float* __restrict__ dek, float* __restrict__ dev, float* __restrict__ dbias) {
  __shared__ float red[16][17];
  const int e16 = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e16;
  const int ntk = ntok * HID, ntot = B * 2 * ntk;
  const float* src = nullptr;
  float* dst = nullptr;
  int n = 0;
  if (idx < ntot) {
    const int b = idx / (2 * ntk), e = idx - b * 2 * ntk;
    src = part + (long long)b * bps * pstride + e;
    n = bps;
    dst = e < ntk ? dek + (long long)b * ntk + e : dev + (long long)b * ntk + (e - ntk);
  } else if (dbias && idx - ntot < HEADS * T * T) {
    src = part + 2 * ntk + (idx - ntot);
    n = B * bps;
    dst = dbias + (idx - ntot);
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = kg;
  for (; k + 48 < n; k += 64) {
    s0 += src[(long long)k * pstride];
    s1 += src[(long long)(k + 16) * pstride];
    s2 += src[(long long)(k + 32) * pstride];
    s3 += src[(long long)(k + 48) * pstride];
  }
  for (; k < n; k += 16) s0 += src[(long long)k * pstride];
  red[kg][e16] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0 && dst) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e16];
    *dst += t;
  }
}

int tb_blocks_per_sample(int B, int HW) { return (int)max(1LL, min((long long)HW, cdiv(1024, B))); }
int tb_pstride(int ntok, int T) { return 2 * ntok * HID + HEADS * T * T; }

}  // namespace

// floats of scratch vmm_attention_bwd needs in dbuf
extern "C" int64_t vmm_attention_bwd_scratch(int32_t mode, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t ntok) {
  const int64_t base = (int64_t)B * T * HW * heads;
  if (mode != 0 || heads != HEADS || T > 16 || ntok > 16) return base;
  const int64_t fast = (int64_t)B * tb_blocks_per_sample(B, HW) * tb_pstride(ntok, T);
  return fast > base ? fast : base;
}

// Fast path of vmm_attention_bwd for mode 0 (same arguments and results; scratch = dbuf of vmm_attention_bwd_scratch floats).  Returns 1
// (nothing launched) outside its envelope: heads = 8, dim_head = 32, T <= 16, ntok <= 16 shared tokens, ntok <= T when the bias
// also covers the tokens.  dek / dev / dbias are accumulated (+=) without atomics: bit-reproducible.
extern "C" int vmm_temporal_attention_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                          int32_t bias_on_cond, const float* out, const float* dout, int32_t ldo, const float* lse,
                                          const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev, float* dbias,
                                          float* scratch, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 16 || T < 1 || ntok > 16 || (ldqkv & 3) || (ldo & 3) || !scratch) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (ntok > 0 && (!dek || !dev)) return -1;
  if (B <= 0 || HW <= 0) return 0;
  TBArgs a{qkv, ek, ev, bias, out, dout, lse, rot_tab, dqkv, scratch, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, 0, 0, q_scale};
  a.blocks_per_sample = tb_blocks_per_sample(B, HW);
  a.pstride = tb_pstride(ntok, T);
  const size_t shm = sizeof(float) * (size_t)(3 * HEADS * T * DH + HEADS * T * (ntok + T) + 2 * HEADS * T * T + 2 * HEADS * 16);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    attr_set = true;
  }
  hipStream_t s = (hipStream_t)stream;
  static const bool use_valu = getenv("VMM_TEMPORAL_BWD_VALU") != nullptr;  // A/B switch: the VALU / LDS kernel of this file
  if (!use_valu) {
    const int rc = vmm_temporal_attention_bwd_mfma_launch(qkv, ldqkv, ek, ev, ntok, bias, bias_on_cond, out, dout, ldo, lse, rot_tab, q_scale, dqkv, scratch,
                                                          B, T, HW, a.blocks_per_sample, a.pstride, s);
    if (rc != 0) return rc;
  } else {
    hipLaunchKernelGGL(temporal_attn_bwd_kernel, dim3((unsigned)(B * a.blocks_per_sample)), dim3(ntok > 0 ? 2 * NTH : NTH), shm, s, a);
    VMM_LAUNCH_CHECK();
  }
  const int nred = B * 2 * ntok * HID + (bias && dbias ? HEADS * T * T : 0);
  if (nred > 0) {
    hipLaunchKernelGGL(temporal_attn_bwd_reduce_kernel, dim3((unsigned)cdiv(nred, 16)), dim3(256), 0, s, scratch, a.pstride, a.blocks_per_sample, B, ntok,
                       T, dek, dev, bias ? dbias : nullptr);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
