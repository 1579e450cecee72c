// Microbenchmark: HBM write bandwidth of the two store shapes an MFMA epilogue can produce for a rows x 768-float output (the to_qkv layer):
//   piece : a lane writes 16 bytes of ITS OWN row (32 rows x 2 adjacent pieces per instruction) -- accumulators with channels in registers
//   run   : 16 lanes cover 256 contiguous bytes of one row (4 rows per instruction)             -- after a transposition
//   dword : 32 lanes write 128 contiguous bytes of one row, 4 bytes each (2 rows per instruction) -- accumulators with rows in registers
// build: hipcc --offload-arch=gfx950 -O3 tools/ubench/store_patterns.hip -o tools/ubench/store_patterns && tools/ubench/store_patterns
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// every wave owns a 64-row x 64-column tile, written as the 3 x 3 / projection epilogues would
template <int MODE>
__global__ __launch_bounds__(256) void store_kernel(float* out, int ld, int ntile_cols) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long tile = (long long)blockIdx.x * 4 + wave;
  const long long row0 = (tile / ntile_cols) * 64;
  const int col0 = (int)(tile % ntile_cols) * 64;
  const f32x4 v = {1.f, 2.f, 3.f, (float)lane};
  if (MODE == 0) {  // piece: lane (row = lane & 31, half lk) writes columns 8 g + 4 lk of column tile j, row tile i
    const int lrow = lane & 31, lk = lane >> 5;
    for (int i = 0; i < 2; ++i)
      for (int j = 0; j < 2; ++j)
        for (int g = 0; g < 4; ++g) *reinterpret_cast<f32x4*>(out + (row0 + i * 32 + lrow) * ld + col0 + j * 32 + 8 * g + 4 * lk) = v;
  } else if (MODE == 1) {  // run: 16 lanes per row
    const int rr = lane >> 4, c4 = (lane & 15) * 4;
    for (int t = 0; t < 16; ++t) *reinterpret_cast<f32x4*>(out + (row0 + rr + 4 * t) * ld + col0 + c4) = v;
  } else {  // dword: lane = column, registers = rows
    const int l31 = lane & 31, lk = lane >> 5;
    for (int j = 0; j < 2; ++j)
      for (int r = 0; r < 32; ++r) out[(row0 + 2 * r + lk) * ld + col0 + j * 32 + l31] = v.w;
  }
}

int main() {
  const long long rows = 202752;  // 8 x 11 x 48 x 48
  const int ld = 768, ntc = ld / 64;
  float* out;
  hipMalloc(&out, rows * ld * sizeof(float));
  const int blocks = (int)(rows / 64 * ntc / 4);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[3] = {"piece (16 B of the lane's own row)", "run (16 lanes x 16 B = one row's 256 B)", "dword (32 lanes x 4 B = 128 B of a row)"};
  for (int mode = 0; mode < 3; ++mode) {
    for (int it = 0; it < 2; ++it) {
      hipEventRecord(e0);
      for (int k = 0; k < 5; ++k) {
        if (mode == 0) hipLaunchKernelGGL(store_kernel<0>, dim3(blocks), dim3(256), 0, 0, out, ld, ntc);
        if (mode == 1) hipLaunchKernelGGL(store_kernel<1>, dim3(blocks), dim3(256), 0, 0, out, ld, ntc);
        if (mode == 2) hipLaunchKernelGGL(store_kernel<2>, dim3(blocks), dim3(256), 0, 0, out, ld, ntc);
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (it) printf("%-45s %7.1f us per launch  %6.2f TB/s\n", names[mode], ms / 5 * 1e3, rows * ld * 4.0 / (ms / 5 * 1e-3) / 1e12);
    }
  }
  return 0;
}
