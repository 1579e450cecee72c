// Shared device helpers for the gfx950 (CDNA4, wave64) kernels of videometamaterials_amd.
// Working layout of every activation inside the library: "frame-major channels-last",
// rows = (b, t, h, w) positions, columns = channels, fp32.  The NCTHW tensors of the
// reference API (vddp.py:730) are converted once at the network edge.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// Single-pass instances of the split-bf16 kernels (the reduced-precision training legs): the translation units listed in build.SINGLE_PASS_SOURCES are
// compiled again with -DVMM_SINGLE_PASS=1 (`train_precision = "bf16"`) and -DVMM_SINGLE_PASS=2 (`train_precision = "fp16"`: the reference's own autocast
// dtype, main.py:34).  In those builds every split product is its hi * hi pass alone (16-bit rounded operands, fp32 accumulation), the lo planes are
// neither formed nor multiplied where the source guards them, and the entry points carry `_bf16` / `_fp16` where the three-pass build says `_bf16x3`
// (VMM_X3(vmm_qkv_bwd_, ) -> vmm_qkv_bwd_bf16x3 / vmm_qkv_bwd_bf16 / vmm_qkv_bwd_fp16); host-only queries (workspace sizes) exist once.
//   mode 2: the 16-bit operand is IEEE half -- split_bf16_pair / vmm_split16 round to fp16 (v_cvt_pk_f16_f32, round-to-nearest-even) and leave a zero
//   lo part, vmm_mfma16 issues v_mfma_f32_32x32x16_f16, weight operands come from vmm_pack_weights' fp16 planes (fmt | 16).  The forward kernels with a
//   single-pass template instance (build.FP16_FORWARD_SOURCES) are compiled in mode 2 as well and export that instance alone, as `_fp16`.
#ifndef VMM_SINGLE_PASS
#define VMM_SINGLE_PASS 0
#endif
#if VMM_SINGLE_PASS == 2
#define VMM_X3(pre, post) pre##fp16##post
#elif VMM_SINGLE_PASS
#define VMM_X3(pre, post) pre##bf16##post
#else
#define VMM_X3(pre, post) pre##bf16x3##post
#endif
// A third operand form (round 6, the fused attention blocks of the sampler): -DVMM_SPLIT_F16=1 keeps the THREE passes of the split product but splits into IEEE-half
// hi | lo (x = hi + lo to 2^-22 for O(1) values, with an absolute floor of 3e-8 from the half denormals -- fine behind a LayerNorm).  gfx950 has no mixed-precision
// subtract for bf16 but it has v_fma_mix_f32 for half: lo = x - hi straight from the packed half, four vector instructions per pair instead of six -- and these
// kernels are bound by vector work in the matrix shadow (tools/ubench/split_f16mix.hip: 31.8 against 39.9 ticks per pair beside an MFMA wave).  Entry points `_f16x3`,
// weight planes = vmm_pack_weights fmt | 32 (fp16 hi | fp16 lo).
#ifndef VMM_SPLIT_F16
#define VMM_SPLIT_F16 0
#endif
#if VMM_SPLIT_F16
#if VMM_SINGLE_PASS
#error "VMM_SPLIT_F16 is a three-pass form"
#endif
#undef VMM_X3
#define VMM_X3(pre, post) pre##f16x3##post
#endif
#define VMM_FP16_OPERANDS (VMM_SINGLE_PASS == 2 || VMM_SPLIT_F16)   // the 16-bit operand TYPE is IEEE half (vmm_mfma16 issues the f16 MFMA)

#define VMM_WAVE 64

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

#define VMM_LAUNCH_CHECK()                         \
  do {                                             \
    hipError_t e__ = hipGetLastError();            \
    if (e__ != hipSuccess) return (int)e__;        \
  } while (0)

// The value lane ^ (1 << bit) holds, without an LDS permute (ds_bpermute: an LDS round trip and an lgkmcnt wait per exchange):
// bits 5 / 4: v_permlane32_swap / v_permlane16_swap (gfx950); bit 3: DPP row_ror:8; bit 2: ds_swizzle SWAP,4 (crossbar only);
// bits 1 / 0: DPP quad_perm.  `bit` must fold to a constant.
__device__ __forceinline__ float lane_xor(float v, int bit) {
  const unsigned u = __builtin_bit_cast(unsigned, v);
  const int i = __builtin_bit_cast(int, v);
  switch (bit) {
    case 5: { const auto r = __builtin_amdgcn_permlane32_swap(u, u, false, false); return __builtin_bit_cast(float, (threadIdx.x & 32) ? r[0] : r[1]); }
    case 4: { const auto r = __builtin_amdgcn_permlane16_swap(u, u, false, false); return __builtin_bit_cast(float, (threadIdx.x & 16) ? r[0] : r[1]); }
    case 3: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x128, 0xf, 0xf, false));
    case 2: return __builtin_bit_cast(float, __builtin_amdgcn_ds_swizzle(i, 0x101f));
    case 1: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0x4E, 0xf, 0xf, false));
    default: return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, i, 0xB1, 0xf, 0xf, false));
  }
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int bit = 5; bit >= 0; --bit) v += lane_xor(v, bit);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int bit = 5; bit >= 0; --bit) v = fmaxf(v, lane_xor(v, bit));
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// reduce inside aligned sub-groups of `width` lanes (width = power of two <= 64)
__device__ __forceinline__ float group_sum(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float group_max(float v, int width) {
  for (int o = width >> 1; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
// the same with the hardware reciprocal (1 ulp) instead of an IEEE division: operand transforms inside GEMM loaders
__device__ __forceinline__ float silu_rcp(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Split two fp32 values into packed bf16 hi | lo parts (x = hi + lo up to 2^-17 relative; both round-to-nearest-even): the operand form of
// every split-bf16 kernel.  Five instructions per pair (v_cvt_pk_bf16_f32, shift, mask, v_pk_add_f32, v_cvt_pk_bf16_f32); gfx950 has no
// v_fma_mix_f32_bf16 that could subtract hi straight from its packed half (the assembler rejects it: "not supported on this GPU").
// (fp16-operand builds, VMM_SINGLE_PASS == 2: hi = the pair rounded to IEEE half, one v_cvt_pk_f16_f32; lo = 0, never multiplied)
__device__ __forceinline__ unsigned split_bf16_pair(float x0, float x1, unsigned& lo) {
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {x0, x1};
#if VMM_SPLIT_F16
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
  float r0, r1;  // x - float(hi half): one v_fma_mix_f32 each (src2 read as f16 from the low / high half of hi, negated)
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(hi));
  asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(hi));
  const f32x2_t r = {r0, r1};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
  return hi;
#elif VMM_FP16_OPERANDS
  typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
  lo = 0u;
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
#else
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  return hi;
#endif
}
// one value: the 16 bits of its hi part, `lo` = the bits of what is left (0 in fp16-operand builds)
__device__ __forceinline__ unsigned short vmm_split16(float v, unsigned short& lo) {
#if VMM_SPLIT_F16
  const _Float16 h = (_Float16)v;
  lo = __builtin_bit_cast(unsigned short, (_Float16)(v - (float)h));
  return __builtin_bit_cast(unsigned short, h);
#elif VMM_FP16_OPERANDS
  lo = 0;
  return __builtin_bit_cast(unsigned short, (_Float16)v);
#else
  const __bf16 h = (__bf16)v;
  lo = __builtin_bit_cast(unsigned short, (__bf16)(v - (float)h));
  return __builtin_bit_cast(unsigned short, h);
#endif
}
// The gradient of the RAW to_qkv rows that the recomputing attention backward (temporal_block_bwd.hip, linattn_block_bwd.hip) hands to the fused to_qkv
// backward (qkv_bwd.hip) is a matrix operand there and nothing else: 1.25 GB per 96 x 96 site as fp32, written once and read once.  The single-pass
// builds store it in the operand's own 16-bit type (fp16 / bf16: the rounding the consumer's split applies anyway, moved to the producer's store --
// the products see the same bits, the gradient is unchanged bit for bit, the round trip is half the bytes).  The split-bf16 build keeps fp32 rows
// (hi | lo would be the same four bytes).  -DVMM_DQKV16=0 on a single-pass object: the fp32 rows again (A/B builds, tools/build_ab.py).
#ifndef VMM_DQKV16
#define VMM_DQKV16 (VMM_SINGLE_PASS != 0)
#endif
#if VMM_DQKV16 && !VMM_SINGLE_PASS
#error "16-bit dqkv rows are the single-pass builds' operand type"
#endif
#if VMM_DQKV16
typedef unsigned short vmm_dqkv_t;
#else
typedef float vmm_dqkv_t;
#endif
__device__ __forceinline__ void st_dqkv4(vmm_dqkv_t* p, float x0, float x1, float x2, float x3) {
#if VMM_DQKV16
  unsigned lo;
  const unsigned h0 = split_bf16_pair(x0, x1, lo), h1 = split_bf16_pair(x2, x3, lo);
  *reinterpret_cast<uint2*>(p) = make_uint2(h0, h1);
#else
  *reinterpret_cast<f32x4*>(p) = f32x4{x0, x1, x2, x3};
#endif
}
__device__ __forceinline__ void st_dqkv1(vmm_dqkv_t* p, float x) {
#if VMM_DQKV16
  unsigned short lo;
  *p = vmm_split16(x, lo);
#else
  *p = x;
#endif
}
// 1.0 as a 16-bit operand (identity fragments of the chained kernels)
#if VMM_FP16_OPERANDS
#define VMM_ONE16 0x3C00u
#else
#define VMM_ONE16 0x3F80u
#endif
// one k16 step on the matrix cores: eight 16-bit operand values per lane and side, given as any 16-byte type (uint4, bf16x8, ...)
typedef __bf16 vmm_bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 vmm_f16x8 __attribute__((ext_vector_type(8)));
template <typename A, typename B>
__device__ __forceinline__ f32x16 vmm_mfma16(const A& a, const B& b, f32x16 c) {
  static_assert(sizeof(A) == 16 && sizeof(B) == 16, "eight 16-bit operand values per lane");
#if VMM_FP16_OPERANDS
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(vmm_f16x8, a), __builtin_bit_cast(vmm_f16x8, b), c, 0, 0, 0);
#else
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(vmm_bf16x8, a), __builtin_bit_cast(vmm_bf16x8, b), c, 0, 0, 0);
#endif
}

// sum over the 16 lanes of a DPP row, every lane gets it: four rotate-and-add steps (v_add_f32_dpp row_ror), no LDS permutes
__device__ __forceinline__ float row_sum16(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x128, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x124, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x122, 0xf, 0xf, false));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x121, 0xf, 0xf, false));
  return v;
}

// ---- activation storage of the "bf16" throughput mode (BASELINE.json configs[3]): feature maps of the two upper levels live in HBM as bf16
// (round-to-nearest-even once, when a kernel stores them; every kernel computes in fp32 / on the matrix cores as before).  Kernels are templated
// on the element type of their activation pointers: float, or bf16s = the 16 stored bits.
typedef unsigned short bf16s;
__device__ __forceinline__ unsigned pack_bf16_pair(float a, float b) {
  typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
  typedef float f32x2_t __attribute__((ext_vector_type(2)));
  const f32x2_t v = {a, b};
  return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
}
__device__ __forceinline__ f32x4 ld4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ f32x4 ld4(const bf16s* p) {
  const uint2 u = *reinterpret_cast<const uint2*>(p);
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
// Raw form of a four-element load, for values that are requested early and consumed later: ld4(const bf16s*) unpacks where it is written, i.e. the
// compiler waits for the load THERE -- a prefetch issued before a barrier then stalls the barrier for an HBM round trip (LABNOTES 9.11).  ldraw4 keeps
// the bits, unpack4 at the point of use.
__device__ __forceinline__ f32x4 ldraw4(const float* p) { return *reinterpret_cast<const f32x4*>(p); }
__device__ __forceinline__ uint2 ldraw4(const bf16s* p) { return *reinterpret_cast<const uint2*>(p); }
__device__ __forceinline__ f32x4 unpack4(const f32x4& v) { return v; }
__device__ __forceinline__ f32x4 unpack4(const uint2& u) {
  return f32x4{__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u)};
}
__device__ __forceinline__ void st4(float* p, const f32x4& v) { *reinterpret_cast<f32x4*>(p) = v; }
__device__ __forceinline__ void st4(bf16s* p, const f32x4& v) {
  *reinterpret_cast<uint2*>(p) = make_uint2(pack_bf16_pair(v.x, v.y), pack_bf16_pair(v.z, v.w));
}
__device__ __forceinline__ float2 ld2(const float* p) { return *reinterpret_cast<const float2*>(p); }
__device__ __forceinline__ float2 ld2(const bf16s* p) {
  const unsigned u = *reinterpret_cast<const unsigned*>(p);
  return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
}
__device__ __forceinline__ void st2(float* p, float a, float b) { *reinterpret_cast<float2*>(p) = make_float2(a, b); }
__device__ __forceinline__ void st2(bf16s* p, float a, float b) { *reinterpret_cast<unsigned*>(p) = pack_bf16_pair(a, b); }
__device__ __forceinline__ float ld1(const float* p) { return *p; }
__device__ __forceinline__ float ld1(const bf16s* p) { return __uint_as_float((unsigned)*p << 16); }
__device__ __forceinline__ void st1(float* p, float v) { *p = v; }
__device__ __forceinline__ void st1(bf16s* p, float v) { *p = (bf16s)(pack_bf16_pair(v, 0.f) & 0xffffu); }

static inline int cdiv(long long a, long long b) { return (int)((a + b - 1) / b); }

// Zero-fill on a stream as a KERNEL.  The library never uses hipMemsetAsync: captured into a hipGraph its memset node was observed
// (ROCm 7.2, MI355X) to run unordered with the kernel nodes around it -- accumulators zeroed after contributions had landed
// (tools/repro_graph_memset.py).  `words` 4-byte words at a 4-byte aligned address.
static __global__ void vmm_zero_words_kernel(unsigned* __restrict__ p, long long words) {
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < words; i += (long long)gridDim.x * blockDim.x) p[i] = 0u;
}
static inline int vmm_zero_async(void* p, size_t bytes, hipStream_t s) {
  const long long words = (long long)(bytes / 4);
  if (words <= 0) return 0;
  const long long want = (words + 255) / 256;
  const int blocks = (int)(want < 2048 ? want : 2048);
  hipLaunchKernelGGL(vmm_zero_words_kernel, dim3(blocks), dim3(256), 0, s, reinterpret_cast<unsigned*>(p), words);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? 0 : (int)e;
}
