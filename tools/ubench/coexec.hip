// Micro-benchmark: how much vector-ALU work fits in the shadow of a matrix-core stream on one CDNA4 SIMD?
// One workgroup of 8 waves per CU (waves w and w + 4 share a SIMD).  Waves 0-3 run `mf` MFMAs per iteration (1 or 3 independent accumulators),
// waves 4-7 run `va` dependent-free v_fma_f32 (or v_pk_fma_f32) per iteration; each role alone, then both together.
//   hipcc --offload-arch=gfx950 -O3 -o coexec tools/ubench/coexec.hip && ./coexec
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int CHAINS, int PK>
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
        for (int u = 0; u < 12; ++u) acc[u % CHAINS] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[u % CHAINS], 0, 0, 0);
      }
      float s = 0.f;
      for (int c = 0; c < CHAINS; ++c) for (int r = 0; r < 16; ++r) s += acc[c][r];
      out[blockIdx.x * 512 + threadIdx.x] = s;
    }
  } else if (do_valu) {
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.001f + i;
    const float m = 1.0001f, c = 0.5f;
    for (int it = 0; it < iters; ++it) {
      if (PK) {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 16; i += 2) {
            f32x2 v = {x[i], x[i + 1]};
            const f32x2 mm = {m, m}, cc = {c, c};
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(v) : "v"(v), "v"(mm), "v"(cc));
            x[i] = v[0]; x[i + 1] = v[1];
          }
      } else {
#pragma unroll
        for (int u = 0; u < 6; ++u)
#pragma unroll
          for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(x[i]) : "v"(x[i]), "v"(m), "v"(c));
      }
    }
    float s = 0.f;
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (blockIdx.x == 0 && lane == 0) ticks[w] = t1 - t0;
}

template <int CHAINS, int PK>
void run(const char* name, float* out, unsigned long long* ticks, int mf, int va) {
  const int iters = 2000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((k<CHAINS, PK>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e0);
  hipLaunchKernelGGL((k<CHAINS, PK>), dim3(256), dim3(512), 0, 0, out, iters, mf, va, ticks);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[8]; hipMemcpy(h, ticks, sizeof(h), hipMemcpyDeviceToHost);
  const double n_mfma = 12.0 * iters, n_valu = (PK ? 48.0 : 96.0) * iters;
  printf("%-44s %7.3f ms | matrix wave: %6.1f ticks/MFMA | vector wave: %5.2f ticks/instr\n", name, ms, mf ? h[0] / n_mfma : 0.0, va ? h[4] / n_valu : 0.0);
}

int main() {
  float* out; unsigned long long* ticks;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&ticks, 64);
  run<3, 0>("MFMA only, 3 chains", out, ticks, 1, 0);
  run<1, 0>("MFMA only, 1 chain", out, ticks, 1, 0);
  run<3, 0>("v_fma_f32 only", out, ticks, 0, 1);
  run<3, 1>("v_pk_fma_f32 only", out, ticks, 0, 1);
  run<3, 0>("MFMA 3 chains + v_fma_f32", out, ticks, 1, 1);
  run<1, 0>("MFMA 1 chain + v_fma_f32", out, ticks, 1, 1);
  run<3, 1>("MFMA 3 chains + v_pk_fma_f32", out, ticks, 1, 1);
  return 0;
}
