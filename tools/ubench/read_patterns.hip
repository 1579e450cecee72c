// Micro-benchmark: HBM read rate of the access patterns a (rows x 768) fp32 matrix can be streamed with, one 512-thread workgroup per CU, 64-row chunks.
//   0: contiguous -- a chunk (196 608 bytes) read front to back, 16 bytes per lane
//   1: qkv_bwd's pieces -- 64 rows x 96 columns per step (384-byte row segments, 3 072 bytes apart), 8 bytes per lane, 384 of the 512 threads
//   2: the same segments with 16 bytes per lane (4 columns x 4 rows per thread)
//   3: pattern 1 on a piece-blocked layout ([chunk][piece][64][96]: every piece one contiguous 24 576-byte block), 8 bytes per lane
// Two steps in flight per thread (the registers of step s are summed when step s + 2 has been requested), a barrier per step like the kernel.
//   hipcc --offload-arch=gfx950 -O3 -o read_patterns tools/ubench/read_patterns.hip && ./read_patterns
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int PAT>
__global__ __launch_bounds__(512) void k(const float* __restrict__ g, float* out, int nchunks, int cpw) {
  const int tid = threadIdx.x;
  const int c0 = blockIdx.x * cpw, c1 = min(nchunks, c0 + cpw);
  float acc = 0.f;
  if (PAT == 0) {
    f32x4 a[6], b[6];  // a step = 1/8 chunk = 24 576 bytes = 512 threads x 48 bytes
    auto req = [&](long long st, f32x4 (&v)[6]) {
      const float* p = g + st * 6144;
#pragma unroll
      for (int i = 0; i < 3; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + (i * 512 + tid) * 4);
    };
    const long long s0 = (long long)c0 * 8, s1 = (long long)c1 * 8;
    req(s0, a); req(s0 + 1, b);
    for (long long st = s0; st < s1; st += 2) {
      acc += a[0].x + a[1].y + a[2].z; req(st + 2 < s1 ? st + 2 : s0, a); __syncthreads();
      acc += b[0].x + b[1].y + b[2].z; req(st + 3 < s1 ? st + 3 : s0, b); __syncthreads();
    }
  } else if (PAT == 1 || PAT == 3) {
    f32x2 a[8], b[8];
    const bool ld = tid < 384;
    const int go = tid / 48, gp = tid % 48;
    auto req = [&](long long st, f32x2 (&v)[8]) {
      const long long ch = st >> 3; const int pc = (int)(st & 7);
      const float* p = PAT == 1 ? g + (ch * 64 + 8 * go) * 768 + pc * 96 + 2 * gp : g + (ch * 8 + pc) * 6144 + (8 * go) * 96 + 2 * gp;
      const int ld_ = PAT == 1 ? 768 : 96;
#pragma unroll
      for (int i = 0; i < 8; ++i) v[i] = *reinterpret_cast<const f32x2*>(p + i * ld_);
    };
    const long long s0 = (long long)c0 * 8, s1 = (long long)c1 * 8;
    if (ld) { req(s0, a); req(s0 + 1, b); }
    for (long long st = s0; st < s1; st += 2) {
      if (ld) { for (int i = 0; i < 8; ++i) acc += a[i].x; req(st + 2 < s1 ? st + 2 : s0, a); } __syncthreads();
      if (ld) { for (int i = 0; i < 8; ++i) acc += b[i].y; req(st + 3 < s1 ? st + 3 : s0, b); } __syncthreads();
    }
  } else {
    f32x4 a[4], b[4];
    const bool ld = tid < 384;
    const int go = tid / 24, gp = tid % 24;  // 16 groups of 4 rows x 24 column quads
    auto req = [&](long long st, f32x4 (&v)[4]) {
      const long long ch = st >> 3; const int pc = (int)(st & 7);
      const float* p = g + (ch * 64 + 4 * go) * 768 + pc * 96 + 4 * gp;
#pragma unroll
      for (int i = 0; i < 4; ++i) v[i] = *reinterpret_cast<const f32x4*>(p + i * 768);
    };
    const long long s0 = (long long)c0 * 8, s1 = (long long)c1 * 8;
    if (ld) { req(s0, a); req(s0 + 1, b); }
    for (long long st = s0; st < s1; st += 2) {
      if (ld) { for (int i = 0; i < 4; ++i) acc += a[i].x; req(st + 2 < s1 ? st + 2 : s0, a); } __syncthreads();
      if (ld) { for (int i = 0; i < 4; ++i) acc += b[i].y; req(st + 3 < s1 ? st + 3 : s0, b); } __syncthreads();
    }
  }
  out[blockIdx.x * 512 + tid] = acc;
}

template <int PAT>
void run(const char* name, const float* g, float* out, long long rows) {
  const int nchunks = (int)(rows / 64), nwg = 256, cpw = (nchunks + nwg - 1) / nwg;
  const int gx = (nchunks + cpw - 1) / cpw;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 2; ++i) hipLaunchKernelGGL((k<PAT>), dim3(gx), dim3(512), 0, 0, g, out, nchunks, cpw);
  hipEventRecord(e0);
  for (int i = 0; i < 10; ++i) hipLaunchKernelGGL((k<PAT>), dim3(gx), dim3(512), 0, 0, g, out, nchunks, cpw);
  hipEventRecord(e1); hipDeviceSynchronize();
  float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 10;
  printf("%-58s %7.3f ms  %6.0f GB/s\n", name, ms, rows * 768.0 * 4 / ms / 1e6);
}

int main() {
  const long long rows = 4LL * 11 * 96 * 96;
  float *g, *out;
  hipMalloc(&g, rows * 768 * 4); hipMalloc(&out, 256 * 512 * 4);
  hipMemset(g, 0, rows * 768 * 4);
  run<0>("0 contiguous, 16 B / lane", g, out, rows);
  run<1>("1 64 x 96 pieces of the row-major matrix, 8 B / lane", g, out, rows);
  run<2>("2 the same pieces, 16 B / lane", g, out, rows);
  run<3>("3 piece-blocked layout, 8 B / lane", g, out, rows);
  return 0;
}
