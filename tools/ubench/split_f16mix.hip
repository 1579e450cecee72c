// Micro-benchmark: the split of fp32 pairs into TWO packed 16-bit operands (x = hi + lo), three instruction sequences.
//   bf16 : v_cvt_pk_bf16_f32, v_lshlrev_b32, v_and_b32, v_sub_f32 x 2, v_cvt_pk_bf16_f32        (6 per pair: what every split-bf16 kernel runs)
//   f16  : v_cvt_pk_f16_f32, v_cvt_f32_f16 x 2, v_sub_f32 x 2, v_cvt_pk_f16_f32                  (6 per pair: IEEE-half hi | lo, plain code)
//   mix  : v_cvt_pk_f16_f32, v_fma_mix_f32 x 2 (x - hi straight from the packed half), v_cvt_pk_f16_f32   (4 per pair)
// (1) the mix form against the plain fp16 form bit for bit, and how well hi + lo reproduces x for the three (max relative residual over ordinary
// magnitudes); (2) ticks per pair alone and beside an MFMA wave on the same SIMD (waves w and w + 4 of a 512-thread workgroup share a SIMD).
//   hipcc --offload-arch=gfx950 -O3 -fno-slp-vectorize -o split_f16mix tools/ubench/split_f16mix.hip && ./split_f16mix
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int V>
__device__ __forceinline__ unsigned split(float x0, float x1, unsigned& lo) {
  const f32x2_t v = {x0, x1};
  if constexpr (V == 0) {
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
    const f32x2_t r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
    return hi;
  } else if constexpr (V == 1) {
    const f16x2_t h = __builtin_convertvector(v, f16x2_t);
    const f32x2_t r = {x0 - (float)h[0], x1 - (float)h[1]};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    return __builtin_bit_cast(unsigned, h);
  } else {
    const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, f16x2_t));
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel_hi:[0,0,1]" : "=v"(r0) : "v"(x0), "v"(hi));
    asm("v_fma_mix_f32 %0, %1, 1.0, -%2 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "=v"(r1) : "v"(x1), "v"(hi));
    const f32x2_t r = {r0, r1};
    lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, f16x2_t));
    return hi;
  }
}

__global__ void check(const float* x, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned l0, l1, l2;
  const unsigned h0 = split<0>(x[2 * i], x[2 * i + 1], l0);
  const unsigned h1 = split<1>(x[2 * i], x[2 * i + 1], l1);
  const unsigned h2 = split<2>(x[2 * i], x[2 * i + 1], l2);
  o[6 * i] = h0; o[6 * i + 1] = l0; o[6 * i + 2] = h1; o[6 * i + 3] = l1; o[6 * i + 4] = h2; o[6 * i + 5] = l2;
}

template <int V>
__global__ __launch_bounds__(512, 2) void k(float* out, int iters, int do_mfma, int do_valu, unsigned long long* ticks) {
  const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (w < 4) {
    if (do_mfma) {
      f32x16 acc[3];
      for (int c = 0; c < 3; ++c) for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
      bf16x8 a, b;
      for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(lane + i); b[i] = (__bf16)(float)(lane - i); }
      for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 12; ++u) acc[u % 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % 3], 0, 0, 0);
      }
      float s = 0.f;
      for (int c = 0; c < 3; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  } else if (do_valu) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.37f + i * 1.13f;
    unsigned acc = 0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 16; i += 2) {
          unsigned lo;
          const unsigned hi = split<V>(x[i], x[i + 1], lo);
          asm volatile("" : "+v"(x[i]), "+v"(x[i + 1]));  // (keeps the split inside the loop)
          acc ^= hi ^ lo;  // 2 extra v_xor per pair in every variant
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = __uint_as_float(acc);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0) ticks[w] = t1 - t0;
}

template <int V>
void run(const char* name, float* out, unsigned long long* ticks, int mf, int va) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<V>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-28s %7.3f ms | matrix wave: %6.1f ticks/MFMA | vector wave: %6.2f ticks/pair (split + 2 xor)\n", name, ms, mf ? h[0] / (12.0 * iters) : 0.0,
         va ? h[4] / (16.0 * iters) : 0.0);
}

static float half_to_float(unsigned short h) {
  const unsigned s = (h >> 15) & 1, e = (h >> 10) & 31, m = h & 1023;
  float v;
  if (e == 0) v = ldexpf((float)m, -24);
  else if (e == 31) v = m ? NAN : INFINITY;
  else v = ldexpf((float)(m | 1024), (int)e - 25);
  return s ? -v : v;
}
static float bf16_to_float(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }

int main() {
  const int n = 1 << 21;
  float* hx = (float*)malloc(n * 2 * 4);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) {  // the operand range of the attention blocks: LayerNorm rows, q / k / v, probabilities -- |x| in 2^-20 .. 2^6
    unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    b = (b & 0x807FFFFFu) | ((unsigned)(107 + rand() % 27) << 23);
    memcpy(&hx[i], &b, 4);
  }
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 2 * 4); hipMalloc(&dout, (size_t)n * 6 * 4);
  hipMemcpy(dx, hx, n * 2 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  unsigned* ho = (unsigned*)malloc((size_t)n * 6 * 4);
  hipMemcpy(ho, dout, (size_t)n * 6 * 4, hipMemcpyDeviceToHost);
  long long diff = 0;
  double worst[3] = {0, 0, 0}, worst_abs16 = 0;
  for (int i = 0; i < n; ++i) {
    if (ho[6 * i + 2] != ho[6 * i + 4] || ho[6 * i + 3] != ho[6 * i + 5]) ++diff;
    for (int hh = 0; hh < 2; ++hh) {
      const double x = hx[2 * i + hh];
      const double rb = bf16_to_float((ho[6 * i] >> (16 * hh)) & 0xffff) + (double)bf16_to_float((ho[6 * i + 1] >> (16 * hh)) & 0xffff);
      const double rf = half_to_float((ho[6 * i + 4] >> (16 * hh)) & 0xffff) + (double)half_to_float((ho[6 * i + 5] >> (16 * hh)) & 0xffff);
      worst[0] = fmax(worst[0], fabs(rb - x) / fabs(x));
      worst[2] = fmax(worst[2], fabs(rf - x) / fabs(x));
      worst_abs16 = fmax(worst_abs16, fabs(rf - x));
    }
  }
  printf("pairs %d: mix form != plain fp16 form in %lld pairs; max |hi + lo - x| / |x|: bf16 %.3g, fp16 %.3g (max absolute residual of the fp16 form %.3g)\n", n, diff,
         worst[0], worst[2], worst_abs16);

  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  run<0>("bf16 split alone", out, ticks, 0, 1);
  run<1>("fp16 split alone", out, ticks, 0, 1);
  run<2>("fp16 mix split alone", out, ticks, 0, 1);
  run<0>("MFMA alone", out, ticks, 1, 0);
  run<0>("MFMA + bf16 split", out, ticks, 1, 1);
  run<1>("MFMA + fp16 split", out, ticks, 1, 1);
  run<2>("MFMA + fp16 mix split", out, ticks, 1, 1);
  return 0;
}
