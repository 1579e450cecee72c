// Micro-benchmark: the split of fp32 pairs into packed bf16 hi | lo operands, two instruction sequences.
//   old: v_cvt_pk_bf16_f32, v_lshlrev_b32, v_and_b32, v_sub_f32 x 2, v_cvt_pk_bf16_f32                    (6 per pair)
//   new: v_cvt_pk_bf16_f32, v_dot2c_f32_bf16 x 2 (x - hi as a bf16 dot product with (-1, 0) / (0, -1)), v_cvt_pk_bf16_f32   (4 per pair)
// (1) bit-equality of the two on random fp32 patterns (normal, tiny, huge, negative zero), (2) ticks per pair alone and beside an MFMA wave on
// the same SIMD (waves w and w + 4 of a 512-thread workgroup share a SIMD, as in coexec.hip).
//   hipcc --offload-arch=gfx950 -O3 -o split_dot2 tools/ubench/split_dot2.hip && ./split_dot2
// Outcome on MI355X (round 5, LABNOTES 10.8): NOT adopted.  v_dot2c_f32_bf16 is not a full-rate instruction -- the 4-instruction sequence needs 34.3
// ticks per pair against 31.0 for the 6-instruction one (alone), 58.3 against 55.0 beside an MFMA wave on the same SIMD -- and the compiler folds the
// (-1, 0) operand into the inline constant -1.0, which the instruction applies to BOTH halves (x0 - hi0 - hi1: the `lo differs` count below).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ unsigned split_old(float x0, float x1, unsigned& lo) {
  const f32x2_t v = {x0, x1};
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2_t));
  const f32x2_t r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  return hi;
}
__device__ __forceinline__ unsigned split_new(float x0, float x1, unsigned& lo) {
  const f32x2_t v = {x0, x1};
  const bf16x2_t hb = __builtin_convertvector(v, bf16x2_t);
  const float r0 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2_t, 0x0000BF80u), x0, false);
  const float r1 = __builtin_amdgcn_fdot2_f32_bf16(hb, __builtin_bit_cast(bf16x2_t, 0xBF800000u), x1, false);
  const f32x2_t r = {r0, r1};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2_t));
  return __builtin_bit_cast(unsigned, hb);
}

__global__ void check(const float* x, unsigned* o, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  unsigned l0, l1;
  const unsigned h0 = split_old(x[2 * i], x[2 * i + 1], l0);
  const unsigned h1 = split_new(x[2 * i], x[2 * i + 1], l1);
  o[4 * i] = h0; o[4 * i + 1] = l0; o[4 * i + 2] = h1; o[4 * i + 3] = l1;
}

template <int NEW>
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
          const unsigned hi = NEW ? split_new(x[i], x[i + 1], lo) : split_old(x[i], x[i + 1], lo);
          asm volatile("" : "+v"(x[i]), "+v"(x[i + 1]));  // (keeps the split inside the loop)
          acc ^= hi ^ lo;  // 2 extra v_xor per pair in both variants
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = __uint_as_float(acc);
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0) ticks[w] = t1 - t0;
}

template <int NEW>
void run(const char* name, float* out, unsigned long long* ticks, int mf, int va) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<NEW>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<NEW>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  printf("%-40s %7.3f ms | matrix wave: %6.1f ticks/MFMA | vector wave: %6.2f ticks/pair (split + 2 xor)\n", name, ms, mf ? h[0] / (12.0 * iters) : 0.0,
         va ? h[4] / (16.0 * iters) : 0.0);
}

int main() {
  const int n = 1 << 22;
  float* hx = (float*)malloc(n * 2 * 4);
  srand(1);
  for (int i = 0; i < 2 * n; ++i) {
    unsigned b = ((unsigned)rand() << 16) ^ (unsigned)rand() ^ ((unsigned)rand() << 31);
    const int cls = i & 15;
    if (cls == 0) b = (b & 0x807FFFFFu) | ((unsigned)(rand() % 20) << 23);        // tiny exponents (lo goes denormal)
    else if (cls == 1) b = (b & 0x807FFFFFu) | ((unsigned)(230 + rand() % 24) << 23);  // huge but finite
    else if (cls == 2) b = b & 0x80000000u;                                        // +-0
    else b = (b & 0x807FFFFFu) | ((unsigned)(100 + rand() % 56) << 23);            // ordinary magnitudes 2^-27 .. 2^28
    memcpy(&hx[i], &b, 4);
  }
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 2 * 4); hipMalloc(&dout, n * 4 * 4);
  hipMemcpy(dx, hx, n * 2 * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(check, dim3(n / 256), dim3(256), 0, 0, dx, dout, n);
  unsigned* ho = (unsigned*)malloc(n * 4 * 4);
  hipMemcpy(ho, dout, n * 4 * 4, hipMemcpyDeviceToHost);
  long long dh = 0, dl = 0, dl_tiny = 0;
  for (int i = 0; i < n; ++i) {
    if (ho[4 * i] != ho[4 * i + 2]) ++dh;
    if (ho[4 * i + 1] != ho[4 * i + 3]) {
      ++dl;
      // differences confined to results that are denormal in one of the two?
      unsigned a = ho[4 * i + 1], b = ho[4 * i + 3];
      bool tiny = true;
      for (int hh = 0; hh < 2; ++hh) {
        const unsigned ea = (a >> (16 * hh + 7)) & 0xFF, eb = (b >> (16 * hh + 7)) & 0xFF;
        if (((a >> (16 * hh)) & 0xFFFF) != ((b >> (16 * hh)) & 0xFFFF) && ea > 1 && eb > 1) tiny = false;
      }
      if (tiny) ++dl_tiny;
      if (dl <= 5) { float f0, f1; memcpy(&f0, &hx[2 * i], 4); memcpy(&f1, &hx[2 * i + 1], 4); printf("  differ: x = %g %g  lo old %08x new %08x\n", f0, f1, a, b); }
    }
  }
  printf("pairs %d: hi differs %lld, lo differs %lld (of which only in denormal / zero results: %lld)\n", n, dh, dl, dl_tiny);

  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  run<0>("old split alone", out, ticks, 0, 1);
  run<1>("dot2c split alone", out, ticks, 0, 1);
  run<0>("MFMA alone", out, ticks, 1, 0);
  run<0>("MFMA + old split", out, ticks, 1, 1);
  run<1>("MFMA + dot2c split", out, ticks, 1, 1);
  return 0;
}
