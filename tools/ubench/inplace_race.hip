// Microbenchmark / repro: does an IN-PLACE update by the second of two consecutive kernels of one stream always see what the first one wrote,
// when several PROCESSES share the GPU?  (LABNOTES 9.8: the token keys of the temporal attentions -- written by the batched dense launch, rotated in
// place by the next launch -- were the one buffer of the denoiser whose contents depended on timing under multi-process load.)
//   K1: x[i] = f(i, iter)        (grid shape of the dense launch: few small workgroups)
//   K2: x[i] = 2 x[i] + 1        in place (mode 0), or y[i] = 2 x[i] + 1 out of place (mode 1)
//   K3: count elements of the result that are not 2 f(i, iter) + 1
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/inplace_race.hip -o tools/ubench/inplace_race
// run:   for p in 1 2 3 4; do tools/ubench/inplace_race 20000 0 & done; wait       (arguments: iterations, mode)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float f(int i, int it) { return (float)((i * 7 + it * 13) & 1023); }

__global__ void k1(float* x, int n, int it) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) x[i] = f(i, it);
}
__global__ void k2(const float* x, float* y, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) y[i] = 2.f * x[i] + 1.f;
}
__global__ void k3(const float* y, int n, int it, unsigned* err, unsigned* stale) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float want = 2.f * f(i, it) + 1.f;
  if (y[i] != want) {
    atomicAdd(err, 1u);
    if (y[i] == 2.f * (2.f * f(i, it - 1) + 1.f) + 1.f) atomicAdd(stale, 1u);  // K2 read the previous iteration's RESULT instead of K1's output
  }
}

int main(int argc, char** argv) {
  const int iters = argc > 1 ? atoi(argv[1]) : 20000, mode = argc > 2 ? atoi(argv[2]) : 0;
  const int n = 101376;  // the token-key block of the 16-wide test model
  float *x, *y;
  unsigned *cnt, h[2] = {0, 0};
  hipMalloc(&x, n * 4); hipMalloc(&y, n * 4); hipMalloc(&cnt, 8);
  hipMemset(cnt, 0, 8);
  hipStream_t s;
  hipStreamCreate(&s);
  for (int it = 1; it <= iters; ++it) {
    hipLaunchKernelGGL(k1, dim3((n + 63) / 64), dim3(64), 0, s, x, n, it);
    hipLaunchKernelGGL(k2, dim3((n + 255) / 256), dim3(256), 0, s, x, mode ? y : x, n);
    hipLaunchKernelGGL(k3, dim3((n + 255) / 256), dim3(256), 0, s, mode ? y : x, n, it, cnt, cnt + 1);
  }
  hipStreamSynchronize(s);
  hipMemcpy(h, cnt, 8, hipMemcpyDeviceToHost);
  printf("mode %d (%s): %d iterations, %u wrong elements, %u of them = the previous iteration's result transformed again\n", mode,
         mode ? "out of place" : "in place", iters, h[0], h[1]);
  return h[0] ? 1 : 0;
}
