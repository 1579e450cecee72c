// Second stages of the weight-gradient kernels: the fixed-order totals of the per-row-slice partial blocks (bit-reproducible; fp32 atomics cost 60-70 us
// per launch whatever the shape).  The bodies are shared by the kernels that run behind every weight-gradient launch (wgrad3x3_bf16x3.hip,
// wgrad1x1_bf16x3.hip) and by the batched launch that totals the pending blocks of many layers at once (reduce_batch.hip: vmm_reduce_batch -- the backward
// pass of the training step ran ~130 of these 5-13 us launches, LABNOTES 10.10 / 11.3).  (bx, by) = the workgroup's position in the job's own grid.
#pragma once
#include "vmm_common.h"

namespace vmm_reduce {

constexpr int W9_PART_FLOATS = 9 * 64 * 64;   // wgrad3x3_bf16x3.hip: one workgroup's partial block (nine taps of a 64 x 64 channel block, accumulator order)
constexpr int W1_BLOCK_FLOATS = 128 * 128;    // wgrad1x1_bf16x3.hip: a 128 x 128 channel block

// dbias[co] += the slices' partial rows, fixed order: 32 channels x 8 slice lanes per workgroup (v = index of the bias workgroup)
__device__ __forceinline__ void bias_rows(const float* __restrict__ bias_part, int nz, int Cout, float* __restrict__ dbias, int v, float* redf) {
  const int co = v * 32 + (threadIdx.x & 31), zq = threadIdx.x >> 5;
  float s = 0.f;
  if (co < Cout)
    for (int z = zq; z < nz; z += 8) s += bias_part[(long long)z * Cout + co];
  redf[zq * 32 + (threadIdx.x & 31)] = s;
  __syncthreads();
  if (zq == 0 && co < Cout) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += redf[k * 32 + threadIdx.x];
    dbias[co] += s;
  }
}

// dw[(tap, ci)][co] += sum over row slices z of the 3 x 3 kernel's partial blocks.  A workgroup takes 32 consecutive 16-byte pieces of one block position
// and all slices: thread = (piece, slice lane), eight slice lanes, LDS tree at the end.  Workgroups bx >= n_main of tile 0: the bias rows.
__device__ __forceinline__ void w9_body(const float* __restrict__ part, int nz, int tiles_x, int tiles_y, float* __restrict__ dw, int Cin, int Cout,
                                        const float* __restrict__ bias_part, float* __restrict__ dbias, int n_main, int bx_, int by_, f32x4 (*red)[32]) {
  if (bx_ >= n_main) {
    if (by_ != 0 || !bias_part) return;  // (workgroup-uniform)
    bias_rows(bias_part, nz, Cout, dbias, bx_ - n_main, reinterpret_cast<float*>(&red[0][0]));
    return;
  }
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int tile = by_;                              // = by * tiles_x + bx
  const int piece = bx_ * 32 + e;                    // 16-byte piece inside the block: ((t * 4 + wq) * 4 + j) * 64 + lane
  const long long zstride = (long long)tiles_x * tiles_y * (W9_PART_FLOATS / 4);
  const f32x4* src = reinterpret_cast<const f32x4*>(part) + (long long)tile * (W9_PART_FLOATS / 4) + piece;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = zl; z < nz; z += 8) s += src[z * zstride];
  red[zl][e] = s;
  __syncthreads();
  if (zl == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][e];
    const int lane = piece & 63, j = (piece >> 6) & 3, wq = (piece >> 8) & 3, t = piece >> 10;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    const int ci = bx * 64 + (wq >> 1) * 32 + 8 * j + 4 * (lane >> 5), co = by * 64 + (wq & 1) * 32 + (lane & 31);
    float* o = dw + ((long long)t * Cin + ci) * Cout + co;
    o[0] += s.x;
    o[Cout] += s.y;
    o[2 * Cout] += s.z;
    o[3 * Cout] += s.w;
  }
}

// dw[ci][co] += sum over row slices z of the 1 x 1 kernel's partial blocks (same scheme)
__device__ __forceinline__ void w1_body(const float* __restrict__ part, int nz, int tiles_x, int tiles_y, float* __restrict__ dw, int Cin, int Cout,
                                        const float* __restrict__ bias_part, float* __restrict__ dbias, int n_main, int bx_, int by_, f32x4 (*red)[32]) {
  if (bx_ >= n_main) {
    if (by_ != 0 || !bias_part) return;
    bias_rows(bias_part, nz, Cout, dbias, bx_ - n_main, reinterpret_cast<float*>(&red[0][0]));
    return;
  }
  const int e = threadIdx.x & 31, zl = threadIdx.x >> 5;
  const int tile = by_;
  const int piece = bx_ * 32 + e;                    // ((((wq * 2 + i) * 2 + j) * 4 + q) * 64 + lane
  const long long zstride = (long long)tiles_x * tiles_y * (W1_BLOCK_FLOATS / 4);
  const f32x4* src = reinterpret_cast<const f32x4*>(part) + (long long)tile * (W1_BLOCK_FLOATS / 4) + piece;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  for (int z = zl; z < nz; z += 8) s += src[z * zstride];
  red[zl][e] = s;
  __syncthreads();
  if (zl == 0) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][e];
    const int lane = piece & 63, q = (piece >> 6) & 3, j = (piece >> 8) & 1, i = (piece >> 9) & 1, wq = piece >> 10;
    const int bx = tile % tiles_x, by = tile / tiles_x;
    // accumulator register 4 q + k of lane: row (ci) = k + 8 q + 4 (lane >> 5) of fragment i, column (co) = lane & 31 of fragment j
    const int ci = bx * 128 + (wq >> 1) * 64 + i * 32 + 8 * q + 4 * (lane >> 5), co = by * 128 + (wq & 1) * 64 + j * 32 + (lane & 31);
    if (ci < Cin && co < Cout) {  // (ci is a multiple of 4 and Cin of 64: the four rows are inside or outside together)
      float* o = dw + (long long)ci * Cout + co;
      o[0] += s.x;
      o[Cout] += s.y;
      o[2 * Cout] += s.z;
      o[3 * Cout] += s.w;
    }
  }
}

// out[c] += sum_{k < n} part[k ld + c]: workgroup bx_ = 16 columns x 16 slices of k; fixed summation order (vmm_sum_partials)
__device__ __forceinline__ void rows_body(const float* __restrict__ part, int n, int ld, int C, float* __restrict__ out, int bx_, float (*red)[17]) {
  const int e = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int c = bx_ * 16 + e;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (c < C) {
    const float* src = part + c;
    int k = kg;
    for (; k + 48 < n; k += 64) {
      s0 += src[(long long)k * ld];
      s1 += src[(long long)(k + 16) * ld];
      s2 += src[(long long)(k + 32) * ld];
      s3 += src[(long long)(k + 48) * ld];
    }
    for (; k < n; k += 16) s0 += src[(long long)k * ld];
  }
  red[kg][e] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0 && c < C) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e];
    out[c] += t;
  }
}

}  // namespace vmm_reduce
