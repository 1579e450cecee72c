// Weight gradient of the 3 x 3 "same" convolutions (ResnetBlock projections, vddp.py:268-285), exact fp32 on the matrix cores, gfx950.
//
//   dWp[(tap, ci)][co] += sum over pixels  x[pixel + shift(tap)][ci] * dY[pixel][co]
//
// The generic kernel (wgrad.hip) gives every 64 x 64 tile of dWp -- one tap, 64 input channels, 64 output channels -- its own workgroup:
// nine workgroups gather the same x rows at nine shifts, every row of dY is fetched once per tap, 16 flops per byte that crosses L2 -> LDS;
// it is bound by the latency of those gathered loads (wgrad.hip, note at the kernel).  Here a workgroup owns ALL NINE taps of a
// (64 input channels) x (64 output channels) block: a segment of R image rows x SEGW pixels (24 in all) of dY and its one-pixel neighbourhood of x are
// staged once in LDS (raw fp32, 64 channels = 256 bytes per pixel), and the nine shifted products read that patch at nine offsets --
// 9 x fewer bytes per flop, nine accumulator tiles (144 registers) per wave.  v_mfma_f32_32x32x2_f32 contracts over two pixels per
// instruction and takes both operands exactly as they lie in LDS (lane = channel, lane half = pixel): one ds_read_b32 per operand, the dY
// operand shared by the nine taps, no transposes.  Row segments are spread over blockIdx.z and added with fp32 atomics like in the generic
// kernel; the bias gradient (column sums of dY) is added by the ci-block-0 workgroups the same way.
#include <stdlib.h>
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

struct W3Args {
  vmm_conv_desc p;
  const float* dy; int lddy;
  float* dw;
  float* dbias;           // column sums of dY are ADDED here (or NULL)
  int R, SEGW;            // segment: R image rows x SEGW pixels (R * SEGW pixels per chunk, SEGW even)
  int segs_per_row, chunks_per_img, nchunks;
};

__global__ __launch_bounds__(256, 2) void wgrad3x3_f32_kernel(const W3Args a) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  const vmm_conv_desc& p = a.p;
  const int R = a.R, SEGW = a.SEGW, PW = SEGW + 2;
  const int buf_floats = ((R + 2) * PW + R * SEGW) * 64;   // one stage: x patch [(R + 2)][PW][64], then dY [R][SEGW][64]; two stages
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int l31 = lane & 31, lk = lane >> 5;
  const int wi = wave >> 1, wj = wave & 1;
  const int H = p.Hin, W = p.Win;
  const int Cin = p.C1 + p.C2;
  const int ci0 = blockIdx.x * 64, co0 = blockIdx.y * 64;
  const bool src1 = ci0 < p.C1;
  const float* xsrc = src1 ? p.a1 + ci0 : p.a2 + (ci0 - p.C1);
  const int ldx = src1 ? p.lda1 : p.lda2;
  const bool want_bias = a.dbias && blockIdx.x == 0 && wi == 0;

  f32x16 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  float bsum = 0.f;

  const int n_x = (R + 2) * PW * 16, n_dy = R * SEGW * 16;  // float4 items
  constexpr int NX = 5, ND = 2;  // items per thread: (R, SEGW) = (1, 24): 1248 / 384 float4, (2, 12): 896 / 384 (more would not fit beside 144 accumulators)
  f32x4 xr[NX], dr[ND];
  // the next chunk's rows are requested into registers before the current chunk is multiplied and stored to LDS after it: their latency
  // hides under ~14k cycles of MFMAs
  auto request = [&](int chunk) {
    const int img = chunk / a.chunks_per_img, rem = chunk - img * a.chunks_per_img;
    const int hb = rem / a.segs_per_row, wb = rem - hb * a.segs_per_row;
    const int h0 = hb * R, w0 = wb * SEGW;
#pragma unroll
    for (int n = 0; n < NX; ++n) {
      const int i = tid + 256 * n;
      const int c4 = i & 15, px = i >> 4;
      const int pr = px / PW, pc = px - pr * PW;
      const int h = h0 - 1 + pr, w = w0 - 1 + pc;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < n_x && h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const f32x4*>(xsrc + ((long long)(img * H + h) * W + w) * ldx + c4 * 4);
      xr[n] = v;
    }
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      const int i = tid + 256 * n;
      const int c4 = i & 15, px = i >> 4;
      const int pr = px / SEGW, pc = px - pr * SEGW;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (i < n_dy) v = *reinterpret_cast<const f32x4*>(a.dy + ((long long)(img * H + h0 + pr) * W + w0 + pc) * a.lddy + co0 + c4 * 4);
      dr[n] = v;
    }
  };
  auto stage = [&](int buf) {
    float* xs = sm + buf * buf_floats;
    float* dys = xs + (R + 2) * PW * 64;
#pragma unroll
    for (int n = 0; n < NX; ++n) {
      const int i = tid + 256 * n;
      if (i < n_x) *reinterpret_cast<f32x4*>(xs + i * 4) = xr[n];
    }
#pragma unroll
    for (int n = 0; n < ND; ++n) {
      const int i = tid + 256 * n;
      if (i < n_dy) *reinterpret_cast<f32x4*>(dys + i * 4) = dr[n];
    }
  };
  // lane (channel l31 of this wave's half, pixel parity lk): x at patch position (r + kh, c + lk + kw), dY at (r, c + lk)
  auto operands = [&](const float* xa, const float* db, int c, float (&av)[9], float& bv) {
    bv = db[c * 64];
#pragma unroll
    for (int t = 0; t < 9; ++t) av[t] = xa[((t / 3) * PW + c + (t % 3)) * 64];
  };
  auto multiply = [&](const float (&av)[9], float bv) {
    if (want_bias) bsum += bv;
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[t], bv, acc[t], 0, 0, 0);
  };
  // Two LDS stages, ONE barrier per chunk: chunk i is multiplied out of stage i & 1 while chunk i + 1's rows are in flight (requested
  // after the chunk's first MFMAs have been issued), then those rows go to the other stage and the barrier publishes them.
  int it = 0;
  if ((int)blockIdx.z < a.nchunks) {
    request(blockIdx.z);
    stage(0);
  }
  __syncthreads();
  for (int chunk = blockIdx.z; chunk < a.nchunks; chunk += gridDim.z, ++it) {
    const bool more = chunk + (int)gridDim.z < a.nchunks;
    const float* xs = sm + (it & 1) * buf_floats;
    const float* dys = xs + (R + 2) * PW * 64;
    for (int r = 0; r < R; ++r) {
      const float* xa = xs + ((r * PW + lk) * 64) + wi * 32 + l31;
      const float* db = dys + ((r * SEGW + lk) * 64) + wj * 32 + l31;
      // two pixel pairs per iteration, the operands of the next pair requested before the current nine MFMAs (SEGW / 2 is even)
      float a0[9], a1[9], b0, b1;
      operands(xa, db, 0, a0, b0);
      for (int c = 0; c < SEGW; c += 4) {
        operands(xa, db, c + 2, a1, b1);
        multiply(a0, b0);
        if (r == 0 && c == 0 && more) request(chunk + gridDim.z);
        if (c + 4 < SEGW) operands(xa, db, c + 4, a0, b0);
        multiply(a1, b1);
      }
    }
    if (more) stage((it + 1) & 1);
    __syncthreads();
  }
  // acc[tap][r]: row = input channel (r & 3) + 8 (r >> 2) + 4 lk of this wave's half, column = output channel l31 of its half
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int i = t * Cin + ci0 + wi * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      atomicAdd(&a.dw[(long long)i * p.Cout + co0 + wj * 32 + l31], acc[t][r]);
    }
  if (want_bias) {
    bsum += lane_xor(bsum, 5);
    if (lane < 32) atomicAdd(&a.dbias[co0 + wj * 32 + l31], bsum);
  }
}

}  // namespace

// Same contract as vmm_conv_wgrad_f32 (which forwards the shapes inside this kernel's envelope here): 3 x 3 / stride 1 / pad 1, no fused
// operand transform, C1 and C2 multiples of 64, Cout a multiple of 64, W a multiple of 24, or W = 12 with an even H.  Returns 1 (nothing
// launched) otherwise.
extern "C" int vmm_conv3x3_wgrad_f32(const vmm_conv_desc* dp, const float* dy, int32_t lddy, float* dw_packed, int32_t nsplit, float* dbias,
                                     float* bias_scratch, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  const bool shape_ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.off_h == -1 && d.off_w == -1 && d.sgn_h == 1 && d.sgn_w == 1 && d.Hv == d.Hin &&
                        d.Wv == d.Win && d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && d.a_mode == 0 && !d.wrap_h && !d.wrap_w;
  const bool chan_ok = d.C1 > 0 && d.C1 % 64 == 0 && d.C2 % 64 == 0 && d.Cout % 64 == 0 && (d.lda1 & 3) == 0 && (!d.C2 || (d.lda2 & 3) == 0) && (lddy & 3) == 0;
  if (!shape_ok || !chan_ok || nsplit < 1 || (dbias && !bias_scratch)) return 1;
  W3Args a;
  if (d.Win % 24 == 0) { a.R = 1; a.SEGW = 24; }
  else if (d.Win == 12 && d.Hin % 2 == 0) { a.R = 2; a.SEGW = 12; }
  else return 1;
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  if (M <= 0) return 0;
  a.p = d; a.dy = dy; a.lddy = lddy; a.dw = dw_packed;
  a.dbias = dbias;
  a.segs_per_row = d.Win / a.SEGW;
  a.chunks_per_img = (d.Hin / a.R) * a.segs_per_row;
  a.nchunks = d.nimg * a.chunks_per_img;
  // row slices: one round of workgroups (two per CU) whatever the caller's nsplit (which is sized for the generic kernel's 64 x 64 tiles)
  const int blocks_xy = ((d.C1 + d.C2) / 64) * (d.Cout / 64);
  const int nz = max(1, min(a.nchunks, 512 / blocks_xy));
  (void)nsplit;
  (void)bias_scratch;
  const size_t shm = 2 * sizeof(float) * 64 * ((size_t)(a.R + 2) * (a.SEGW + 2) + (size_t)a.R * a.SEGW);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_f32_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(wgrad3x3_f32_kernel, dim3((d.C1 + d.C2) / 64, d.Cout / 64, nz), dim3(256), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
