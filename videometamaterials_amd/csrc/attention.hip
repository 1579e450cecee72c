// Attention cores for gfx950.  Inputs are the rows written by the projection
// GEMM (igemm_conv.hip) with q already scaled and q,k already rotated by its epilogue, so the cores are pure
// softmax/weighted-sum arithmetic.  dim_head = 32 for the linear and the mid spatial attention (their constructor default, vddp.py:314, 401:
// Unet3D does not forward attn_dim_head to them, vddp.py:679, 687); the temporal attention core takes any multiple of 4 up to 128 (vddp.py:615).
#include "vmm_common.h"
#include "head_vec.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32;

__device__ __forceinline__ void load32(float (&dst)[DH], const float* src) {
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(src + i * 4);
    dst[i * 4 + 0] = v.x; dst[i * 4 + 1] = v.y; dst[i * 4 + 2] = v.z; dst[i * 4 + 3] = v.w;
  }
}
__device__ __forceinline__ float dot32(const float (&a)[DH], const float* b) {
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(b + i * 4);
    s0 = fmaf(a[i * 4 + 0], v.x, s0); s1 = fmaf(a[i * 4 + 1], v.y, s1);
    s2 = fmaf(a[i * 4 + 2], v.z, s2); s3 = fmaf(a[i * 4 + 3], v.w, s3);
  }
  return (s0 + s1) + (s2 + s3);
}
// online-softmax update with one key/value
__device__ __forceinline__ void online_step(float s, const float* vrow, float& m, float& l, float (&acc)[DH]) {
  const float mn = fmaxf(m, s);
  const float f = __expf(m - mn);
  const float p = __expf(s - mn);
  l = l * f + p;
#pragma unroll
  for (int i = 0; i < DH / 4; ++i) {
    const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + i * 4);
    acc[i * 4 + 0] = fmaf(p, v.x, acc[i * 4 + 0] * f); acc[i * 4 + 1] = fmaf(p, v.y, acc[i * 4 + 1] * f);
    acc[i * 4 + 2] = fmaf(p, v.z, acc[i * 4 + 2] * f); acc[i * 4 + 3] = fmaf(p, v.w, acc[i * 4 + 3] * f);
  }
  m = mn;
}

// ---------------------------------------------------------------- temporal attention: one thread per (b, pixel, head, query frame)
// The one attention family whose head width follows the constructor (attn_dim_head, vddp.py:582, 615): templated on the head slice
// (head_vec.h); dh = 32 is the instance of every shipped configuration.
template <int DM, bool EX>
__device__ __forceinline__ void online_step_hv(float s, const float* vrow, int dh, float& m, float& l, float (&acc)[DM]) {
  const float mn = fmaxf(m, s);
  const float f = __expf(m - mn);
  const float p = __expf(s - mn);
  l = l * f + p;
#pragma unroll
  for (int i = 0; i < DM / 4; ++i) {
    if (HeadVec<DM, EX>::on(i, dh)) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(vrow + i * 4);
      acc[i * 4 + 0] = fmaf(p, v.x, acc[i * 4 + 0] * f); acc[i * 4 + 1] = fmaf(p, v.y, acc[i * 4 + 1] * f);
      acc[i * 4 + 2] = fmaf(p, v.z, acc[i * 4 + 2] * f); acc[i * 4 + 3] = fmaf(p, v.w, acc[i * 4 + 3] * f);
    }
  }
  m = mn;
}

template <int DM, bool EX>
__global__ __launch_bounds__(256) void temporal_attn_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                            const float* __restrict__ ev, int ntok, const float* __restrict__ bias,
                                                            int bias_on_cond, float* __restrict__ out, int ldo, int B, int T, int HW,
                                                            int heads, int dh_, float* __restrict__ lse) {
  using HV = HeadVec<DM, EX>;
  const int dh = EX ? DM : dh_;
  const long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const long long total = (long long)B * HW * heads * T;
  if (gid >= total) return;
  const int i = (int)(gid % T);
  const int head = (int)((gid / T) % heads);
  const long long bp = gid / ((long long)T * heads);
  const int pix = (int)(bp % HW);
  const int b = (int)(bp / HW);
  const int hid = heads * dh;
  const long long row0 = (long long)b * T * HW + pix;  // row of frame 0
  float q[DM], acc[DM];
  HV::ld(q, qkv + (row0 + (long long)i * HW) * ldqkv + head * dh, dh);
  HV::zero(acc);
  float m = -INFINITY, l = 0.f;
  const float* brow = bias ? bias + ((long long)head * T + i) * T : nullptr;
  if (ek) {
    for (int j = 0; j < ntok; ++j) {
      const float* kr = ek + ((long long)b * ntok + j) * hid + head * dh;
      float s = HV::dot(q, kr, dh);
      if (brow && bias_on_cond) s += brow[j];
      online_step_hv<DM, EX>(s, ev + ((long long)b * ntok + j) * hid + head * dh, dh, m, l, acc);
    }
  }
  for (int j = 0; j < T; ++j) {
    const float* r = qkv + (row0 + (long long)j * HW) * ldqkv + head * dh;
    float s = HV::dot(q, r + hid, dh);
    if (brow) s += brow[j];
    online_step_hv<DM, EX>(s, r + 2 * hid, dh, m, l, acc);
  }
  const float inv = 1.0f / l;
  if (lse) lse[(row0 + (long long)i * HW) * heads + head] = m + logf(l);
  float* o = out + (row0 + (long long)i * HW) * ldo + head * dh;
#pragma unroll
  for (int d = 0; d < DM / 4; ++d) {
    if (HV::on(d, dh)) {
      f32x4 v = {acc[d * 4] * inv, acc[d * 4 + 1] * inv, acc[d * 4 + 2] * inv, acc[d * 4 + 3] * inv};
      *reinterpret_cast<f32x4*>(o + d * 4) = v;
    }
  }
}

// ---------------------------------------------------------------- mid spatial attention: block = (query tile, frame, head); K/V tiles via LDS
constexpr int SA_TILE = 128;
__global__ __launch_bounds__(128) void spatial_attn_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ek,
                                                           const float* __restrict__ ev, int ntok, int tok_per_frame,
                                                           float* __restrict__ out, int ldo, int T, int HW, int heads,
                                                           float* __restrict__ lse) {
  __shared__ __attribute__((aligned(16))) float Ks[SA_TILE][DH];
  __shared__ __attribute__((aligned(16))) float Vs[SA_TILE][DH];
  const int tid = threadIdx.x;
  const int head = blockIdx.y % heads;
  const int bt = blockIdx.y / heads;
  const int b = bt / T, t = bt % T;
  const int hid = heads * DH;
  const int qi = blockIdx.x * SA_TILE + tid;
  const bool qvalid = qi < HW;
  const long long row0 = (long long)bt * HW;
  float q[DH], acc[DH];
#pragma unroll
  for (int d = 0; d < DH; ++d) { q[d] = 0.f; acc[d] = 0.f; }
  if (qvalid) load32(q, qkv + (row0 + qi) * ldqkv + head * DH);
  float m = -INFINITY, l = 0.f;
  if (ek) {
    const int j0 = tok_per_frame ? t : 0, j1 = tok_per_frame ? t + 1 : ntok;
    for (int j = j0; j < j1; ++j) {
      const float s = dot32(q, ek + ((long long)b * ntok + j) * hid + head * DH);
      online_step(s, ev + ((long long)b * ntok + j) * hid + head * DH, m, l, acc);
    }
  }
  for (int k0 = 0; k0 < HW; k0 += SA_TILE) {
    const int nkeys = min(SA_TILE, HW - k0);
    __syncthreads();
    for (int e = tid; e < nkeys * (DH / 4); e += SA_TILE) {
      const int kr = e / (DH / 4), c = (e % (DH / 4)) * 4;
      const float* src = qkv + (row0 + k0 + kr) * ldqkv + head * DH + c;
      *reinterpret_cast<f32x4*>(&Ks[kr][c]) = *reinterpret_cast<const f32x4*>(src + hid);
      *reinterpret_cast<f32x4*>(&Vs[kr][c]) = *reinterpret_cast<const f32x4*>(src + 2 * hid);
    }
    __syncthreads();
    for (int j = 0; j < nkeys; ++j) {
      const float s = dot32(q, &Ks[j][0]);
      online_step(s, &Vs[j][0], m, l, acc);
    }
  }
  if (!qvalid) return;
  const float inv = 1.0f / l;
  if (lse) lse[(row0 + qi) * heads + head] = m + logf(l);
  float* o = out + (row0 + qi) * ldo + head * DH;
#pragma unroll
  for (int d = 0; d < DH / 4; ++d) {
    f32x4 v = {acc[d * 4] * inv, acc[d * 4 + 1] * inv, acc[d * 4 + 2] * inv, acc[d * 4 + 3] * inv};
    *reinterpret_cast<f32x4*>(o + d * 4) = v;
  }
}

// ---------------------------------------------------------------- linear attention, pass 1: per (frame, head, split) partial context
// thread (d = tid/8, e0 = (tid%8)*4) owns ctx[d][e0..e0+3]; tiles of LA_TILE rows staged through LDS.
constexpr int LA_TILE = 64;
constexpr int LA_PART = DH * DH + 2 * DH;  // ctx | max | sum
__global__ __launch_bounds__(256) void linattn_partial_kernel(const float* __restrict__ qkv, int ldqkv, int HW, int heads,
                                                              int nsplit, int rows_per_split, float* __restrict__ part) {
  __shared__ __attribute__((aligned(16))) float Ks[LA_TILE][DH + 1];
  __shared__ __attribute__((aligned(16))) float Vs[LA_TILE][DH];
  const int tid = threadIdx.x;
  const int split = blockIdx.x;
  const int fh = blockIdx.y;  // frame*heads + head
  const int head = fh % heads;
  const long long frame = fh / heads;
  const int hid = heads * DH;
  const int d = tid >> 3, e0 = (tid & 7) * 4;
  const int n_begin = split * rows_per_split, n_end = min(n_begin + rows_per_split, HW);
  float m = -INFINITY, ssum = 0.f;
  float c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int n0 = n_begin; n0 < n_end; n0 += LA_TILE) {
    const int nrows = min(LA_TILE, n_end - n0);
    __syncthreads();
    for (int e = tid; e < LA_TILE * (DH / 4); e += 256) {
      const int r = e / (DH / 4), cc = (e % (DH / 4)) * 4;
      f32x4 kv = {-INFINITY, -INFINITY, -INFINITY, -INFINITY}, vv = {0.f, 0.f, 0.f, 0.f};
      if (r < nrows) {
        const float* src = qkv + (frame * HW + n0 + r) * ldqkv + head * DH + cc;
        kv = *reinterpret_cast<const f32x4*>(src + hid);
        vv = *reinterpret_cast<const f32x4*>(src + 2 * hid);
      }
      Ks[r][cc] = kv.x; Ks[r][cc + 1] = kv.y; Ks[r][cc + 2] = kv.z; Ks[r][cc + 3] = kv.w;
      *reinterpret_cast<f32x4*>(&Vs[r][cc]) = vv;
    }
    __syncthreads();
    // tile max of column d (8 lanes share d: each scans 8 rows, then xor-reduce over the 8 lanes)
    float tm = -INFINITY;
#pragma unroll
    for (int r = 0; r < LA_TILE / 8; ++r) tm = fmaxf(tm, Ks[(tid & 7) * (LA_TILE / 8) + r][d]);
    tm = group_max(tm, 8);
    const float mn = fmaxf(m, tm);
    const float f = __expf(m - mn);
    ssum *= f; c[0] *= f; c[1] *= f; c[2] *= f; c[3] *= f;
    m = mn;
    for (int r = 0; r < nrows; ++r) {
      const float p = __expf(Ks[r][d] - mn);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(&Vs[r][e0]);
      ssum += p;
      c[0] = fmaf(p, vv.x, c[0]); c[1] = fmaf(p, vv.y, c[1]); c[2] = fmaf(p, vv.z, c[2]); c[3] = fmaf(p, vv.w, c[3]);
    }
  }
  float* pp = part + ((long long)fh * nsplit + split) * LA_PART;
  *reinterpret_cast<f32x4*>(pp + d * DH + e0) = (f32x4){c[0], c[1], c[2], c[3]};
  if ((tid & 7) == 0) { pp[DH * DH + d] = m; pp[DH * DH + DH + d] = ssum; }
}

// pass 2: merge the splits and the conditioning tokens; block per (frame, head), 256 threads as (d, e0)
__global__ __launch_bounds__(256) void linattn_merge_kernel(const float* __restrict__ part, int nsplit, const float* __restrict__ ek,
                                                            const float* __restrict__ ev, int ntok, int T, int HW, int heads,
                                                            float* __restrict__ ctx, float* __restrict__ kstat) {
  const int tid = threadIdx.x;
  const int fh = blockIdx.x;
  const int head = fh % heads;
  const int b = (fh / heads) / T;
  const int hid = heads * DH;
  const int d = tid >> 3, e0 = (tid & 7) * 4;
  float m = -INFINITY;
  for (int s = 0; s < nsplit; ++s) m = fmaxf(m, part[((long long)fh * nsplit + s) * LA_PART + DH * DH + d]);
  if (ek)
    for (int j = 0; j < ntok; ++j) m = fmaxf(m, ek[((long long)b * ntok + j) * hid + head * DH + d]);
  float ssum = 0.f, c[4] = {0.f, 0.f, 0.f, 0.f};
  for (int s = 0; s < nsplit; ++s) {
    const float* pp = part + ((long long)fh * nsplit + s) * LA_PART;
    const float pm = pp[DH * DH + d];
    const float f = (pm == -INFINITY) ? 0.f : __expf(pm - m);
    ssum += pp[DH * DH + DH + d] * f;
    const f32x4 cv = *reinterpret_cast<const f32x4*>(pp + d * DH + e0);
    c[0] += cv.x * f; c[1] += cv.y * f; c[2] += cv.z * f; c[3] += cv.w * f;
  }
  if (ek) {
    for (int j = 0; j < ntok; ++j) {
      const float p = __expf(ek[((long long)b * ntok + j) * hid + head * DH + d] - m);
      const f32x4 vv = *reinterpret_cast<const f32x4*>(ev + ((long long)b * ntok + j) * hid + head * DH + e0);
      ssum += p;
      c[0] = fmaf(p, vv.x, c[0]); c[1] = fmaf(p, vv.y, c[1]); c[2] = fmaf(p, vv.z, c[2]); c[3] = fmaf(p, vv.w, c[3]);
    }
  }
  if (kstat && (tid & 7) == 0) { kstat[(long long)fh * 2 * DH + d] = m; kstat[(long long)fh * 2 * DH + DH + d] = 1.0f / ssum; }
  const float sc = 1.0f / (ssum * (float)HW);  // softmax normaliser and v / (h*w) (vddp.py:371)
  *reinterpret_cast<f32x4*>(ctx + (long long)fh * DH * DH + d * DH + e0) = (f32x4){c[0] * sc, c[1] * sc, c[2] * sc, c[3] * sc};
}

// pass 3: out[n, head*32 + e] = sum_d ctx[d][e] * softmax_d(q[n, head*32 + :])[d] * scale ; thread per (row, head)
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ctx,
                                                            float* __restrict__ out, int ldo, int HW, int heads, float scale) {
  extern __shared__ __attribute__((aligned(16))) float cs[];  // [heads][DH*DH + 4]
  const int tid = threadIdx.x;
  const long long frame = blockIdx.y;
  const int rows_per_block = 256 / heads;
  const int head = tid % heads;
  const int n = blockIdx.x * rows_per_block + tid / heads;
  constexpr int CSTR = DH * DH + 4;
  for (int e = tid; e < heads * DH * DH / 4; e += 256) {
    const int h = e / (DH * DH / 4), r = e % (DH * DH / 4);
    *reinterpret_cast<f32x4*>(&cs[h * CSTR + r * 4]) = *reinterpret_cast<const f32x4*>(ctx + (frame * heads + h) * DH * DH + r * 4);
  }
  __syncthreads();
  if (n >= HW || tid / heads >= rows_per_block) return;
  float q[DH];
  load32(q, qkv + (frame * HW + n) * ldqkv + head * DH);
  float mx = q[0];
#pragma unroll
  for (int d = 1; d < DH; ++d) mx = fmaxf(mx, q[d]);
  float sum = 0.f;
#pragma unroll
  for (int d = 0; d < DH; ++d) { q[d] = __expf(q[d] - mx); sum += q[d]; }
  const float sc = scale / sum;
  float o[DH];
#pragma unroll
  for (int e = 0; e < DH; ++e) o[e] = 0.f;
  const float* ch = cs + head * CSTR;
#pragma unroll 4
  for (int d = 0; d < DH; ++d) {
    const float qd = q[d] * sc;
#pragma unroll
    for (int e4 = 0; e4 < DH / 4; ++e4) {
      const f32x4 cv = *reinterpret_cast<const f32x4*>(ch + d * DH + e4 * 4);
      o[e4 * 4] = fmaf(qd, cv.x, o[e4 * 4]); o[e4 * 4 + 1] = fmaf(qd, cv.y, o[e4 * 4 + 1]);
      o[e4 * 4 + 2] = fmaf(qd, cv.z, o[e4 * 4 + 2]); o[e4 * 4 + 3] = fmaf(qd, cv.w, o[e4 * 4 + 3]);
    }
  }
  float* op = out + (frame * HW + n) * ldo + head * DH;
#pragma unroll
  for (int e4 = 0; e4 < DH / 4; ++e4) *reinterpret_cast<f32x4*>(op + e4 * 4) = (f32x4){o[e4 * 4], o[e4 * 4 + 1], o[e4 * 4 + 2], o[e4 * 4 + 3]};
}

}  // namespace

extern "C" int vmm_temporal_attention(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                                      const float* bias, int32_t bias_on_cond, float* out, int32_t ldo, int32_t B, int32_t T,
                                      int32_t HW, int32_t heads, int32_t dh, float* lse, vmm_stream_t stream) {
  if (!vmm_head_dim_ok(dh) || heads < 1 || (ldqkv & 3) || (ldo & 3)) return -1;
  if (bias_on_cond && ek && ntok != T) return -2;  // the reference's in-place add needs tokens == frames
  {  // LDS-staged workgroup-per-pixel kernel (temporal_attn_fwd.hip) where it applies (8 heads of 32)
    const int rc = vmm_temporal_attention_staged(qkv, ldqkv, ek, ev, ntok, bias, bias_on_cond, out, ldo, B, T, HW, heads, dh, lse, stream);
    if (rc != 1) return rc;
  }
  const long long total = (long long)B * HW * heads * T;
  if (total <= 0) return 0;
#define VMM_CALL(DM, EX)                                                                                                                    \
  hipLaunchKernelGGL((temporal_attn_kernel<DM, EX>), dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, qkv, ldqkv, ek, ev, ntok, bias, \
                     bias_on_cond, out, ldo, B, T, HW, heads, dh, lse)
  VMM_HEADVEC_DISPATCH(dh, VMM_CALL);
#undef VMM_CALL
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_spatial_attention(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok,
                                     int32_t tok_per_frame, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW,
                                     int32_t heads, int32_t dh, float* lse, vmm_stream_t stream) {
  if (dh != DH || (ldqkv & 3) || (ldo & 3)) return -1;
  if (tok_per_frame && ek && ntok != T) return -2;
  hipLaunchKernelGGL(spatial_attn_kernel, dim3(cdiv(HW, SA_TILE), B * T * heads), dim3(SA_TILE), 0, (hipStream_t)stream, qkv,
                     ldqkv, ek, ev, ntok, tok_per_frame, out, ldo, T, HW, heads, lse);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_linattn_context(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t B,
                                   int32_t T, int32_t HW, int32_t heads, int32_t dh, int32_t nsplit, float* part, float* ctx,
                                   float* kstat, vmm_stream_t stream) {
  if (dh != DH || (ldqkv & 3) || nsplit < 1) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int rows_per_split = cdiv(cdiv(HW, nsplit), LA_TILE) * LA_TILE;
  hipLaunchKernelGGL(linattn_partial_kernel, dim3(nsplit, B * T * heads), dim3(256), 0, s, qkv, ldqkv, HW, heads, nsplit,
                     rows_per_split, part);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * T * heads), dim3(256), 0, s, part, nsplit, ek, ev, ntok, T, HW, heads, ctx, kstat);
  VMM_LAUNCH_CHECK();
  return 0;
}

// cond_attention = 'cross-attention' (vddp.py:354-363): keys / values are the conditioning tokens alone, so the context of a (frame, head) is
// the merge above with no pixel partials -- ctx[d][e] = sum_j softmax_j(ek[j][d]) ev[j][e] / (h w), identical for the frames of a sample
extern "C" int vmm_linattn_cross_context(const float* ek, const float* ev, int32_t ntok, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh,
                                         float* ctx, float* kstat, vmm_stream_t stream) {
  if (dh != DH || !ek || !ev || ntok < 1) return -1;
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * T * heads), dim3(256), 0, (hipStream_t)stream, nullptr, 0, ek, ev, ntok, T, HW, heads, ctx, kstat);
  VMM_LAUNCH_CHECK();
  return 0;
}

// the same with pass 1 on the split-bf16 matrix cores (temporal_core.hip); inference
extern "C" int vmm_linattn_context_bf16x3(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, int32_t B,
                                          int32_t T, int32_t HW, int32_t heads, int32_t dh, int32_t nsplit, float* part, float* ctx,
                                          float* kstat, vmm_stream_t stream) {
  if (dh != DH || (ldqkv & 3) || nsplit < 1) return -1;
  hipStream_t s = (hipStream_t)stream;
  const int rows_per_split = cdiv(cdiv(HW, nsplit), LA_TILE) * LA_TILE;
  if (const int rc = vmm_linattn_partial_bf16x3(qkv, ldqkv, B * T, HW, heads, nsplit, rows_per_split, part, stream)) return rc;
  hipLaunchKernelGGL(linattn_merge_kernel, dim3(B * T * heads), dim3(256), 0, s, part, nsplit, ek, ev, ntok, T, HW, heads, ctx, kstat);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_linattn_apply(const float* qkv, int32_t ldqkv, const float* ctx, float* out, int32_t ldo, int32_t B, int32_t T,
                                 int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (dh != DH || (ldqkv & 3) || (ldo & 3) || heads < 1 || heads > 36) return -1;  // (the fallback keeps every head's 32 x 32 context in LDS: 36 heads = 148 KB)
  {  // fp32 matrix-core row pass (linattn_rows.hip) where it applies (heads a multiple of 4)
    const int rc = vmm_linattn_apply_mfma(qkv, ldqkv, ctx, out, ldo, B * T, HW, heads, 0.17677669529663687f /* 32^-0.5, vddp.py:316 */, stream);
    if (rc != 1) return rc;
  }
  const int rows_per_block = 256 / heads;  // (threads beyond rows_per_block * heads idle: any number of heads)
  const size_t shm = sizeof(float) * heads * (DH * DH + 4);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_apply_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(linattn_apply_kernel, dim3(cdiv(HW, rows_per_block), B * T), dim3(256), shm, (hipStream_t)stream, qkv, ldqkv,
                     ctx, out, ldo, HW, heads, 0.17677669529663687f /* 32^-0.5, vddp.py:316 */);
  VMM_LAUNCH_CHECK();
  return 0;
}
