// Implicit-GEMM per-frame convolution / projection for gfx950, exact fp32 on the matrix cores
// (v_mfma_f32_32x32x2_f32: bitwise an fp32 fma chain, 157 TFLOP/s peak).  Used by the training plans (forward, data
// gradients) and by inference when Unet3D.precision == "fp32"; the default inference path is igemm_bf16x3.hip.
//
// One kernel covers every dense contraction of the denoiser:
//   3x3 / 7x7 / 4x4-stride-2 convolutions, the transposed 4x4-stride-2 convolution (as 4 output
//   phases of 2x2 taps), 1x1 convolutions and nn.Linear projections.
// Rows (M) are output positions in frame-major channels-last order, K runs over (tap, input channel),
// N over output channels.  A tiles are gathered with zero padding straight from the channels-last
// activation rows (16-byte loads along channels), optionally from two concatenated sources, optionally
// through the producer's fused GroupNorm+FiLM+SiLU; B tiles are 16-byte loads of the k-major weights.
// 256 threads = 4 waves; LDS double-buffered, global loads for chunk k+1 in flight during the MFMAs of k.
// Grid: one dimension, N tiles fastest, so the blocks that share an A row panel run together (L2 reuse).
#include "igemm_common.h"

namespace {

constexpr int BK = 16;

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void igemm_f32_kernel(const vmm_conv_desc p, int n_tiles) {
  constexpr int TM = BM / WM, TN = BN / WN;
  constexpr int MT = TM / 32, NT = TN / 32;
  constexpr int APAD = 4;
  constexpr int A_PASSES = BM / 64;                        // 64 rows x 4 float4 per pass
  constexpr int B_F4 = BK * BN / 4;                        // float4 per B tile
  constexpr int B_PASSES = (B_F4 + 255) / 256;
  __shared__ __attribute__((aligned(16))) float As[2][BK][BM + APAD];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK][BN];

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wm = wave / WN, wn = wave % WN;
  const unsigned M = (unsigned)(p.nimg * p.Hv * p.Wv);
  const unsigned m0 = (blockIdx.x / n_tiles) * BM;
  const int n0 = (blockIdx.x % n_tiles) * BN;
  const int Cin = p.C1 + p.C2;
  const int Ktot = p.KH * p.KW * Cin;
  const int nk = (Ktot + BK - 1) / BK;

  igemm::RowInfo ri[A_PASSES];
#pragma unroll
  for (int ps = 0; ps < A_PASSES; ++ps) ri[ps] = igemm::decode_row(p, m0 + ps * 64 + (tid >> 2), M);
  const int a_k4 = tid & 3;
  igemm::KPos kp;
  kp.init(p, a_k4 * 4, Cin);
  const int b_kk[2] = {tid / (BN / 4), (tid + 256) / (BN / 4)};
  const int b_n4[2] = {tid % (BN / 4), (tid + 256) % (BN / 4)};

  f32x4 areg[A_PASSES];
  f32x4 breg[B_PASSES];

  auto load_chunk = [&](int kc) {  // must be called with kc = 0, 1, 2, ... (kp advances incrementally)
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) areg[ps] = igemm::load_a4(p, ri[ps], kp, Ktot);
    kp.advance(p, BK, Cin);
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int kk = kc * BK + b_kk[ps];
      const int n = n0 + b_n4[ps] * 4;
      if (b_kk[ps] < BK && kk < Ktot && n < p.Cout) v = *reinterpret_cast<const f32x4*>(p.w + (long long)kk * p.Cout + n);
      breg[ps] = v;
    }
  };
  auto store_chunk = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < A_PASSES; ++ps) {
      const int r = ps * 64 + (tid >> 2);
      As[buf][a_k4 * 4 + 0][r] = areg[ps].x;
      As[buf][a_k4 * 4 + 1][r] = areg[ps].y;
      As[buf][a_k4 * 4 + 2][r] = areg[ps].z;
      As[buf][a_k4 * 4 + 3][r] = areg[ps].w;
    }
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      if (b_kk[ps] < BK) *reinterpret_cast<f32x4*>(&Bs[buf][b_kk[ps]][b_n4[ps] * 4]) = breg[ps];
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  load_chunk(0);
  store_chunk(0);
  __syncthreads();
  const int lrow = lane & 31, lk = lane >> 5;
  for (int kc = 0; kc < nk; ++kc) {
    const int buf = kc & 1;
    if (kc + 1 < nk) load_chunk(kc + 1);
#pragma unroll
    for (int kk = 0; kk < BK; kk += 2) {
      float av[MT], bv[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) av[i] = As[buf][kk + lk][wm * TM + i * 32 + lrow];
#pragma unroll
      for (int j = 0; j < NT; ++j) bv[j] = Bs[buf][kk + lk][wn * TN + j * 32 + lrow];
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i], bv[j], acc[i][j], 0, 0, 0);
    }
    if (kc + 1 < nk) store_chunk(buf ^ 1);
    __syncthreads();
  }
  igemm::epilogue<MT, NT>(p, acc, m0 + wm * TM, n0 + wn * TN, M, lane);
}

template <int BM, int BN, int WM, int WN>
int launch(const vmm_conv_desc& d, hipStream_t s) {
  const long long M = (long long)d.nimg * d.Hv * d.Wv;
  const int nt = cdiv(d.Cout, BN);
  hipLaunchKernelGGL((igemm_f32_kernel<BM, BN, WM, WN>), dim3((unsigned)(cdiv(M, BM) * (long long)nt)), dim3(256), 0, s, d, nt);
  VMM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

extern "C" int vmm_conv_igemm_f32(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  hipStream_t s = (hipStream_t)stream;
  if ((d.C1 & 3) || (d.C2 & 3) || (d.lda1 & 3) || (d.C2 && (d.lda2 & 3)) || (d.Cout & 3) || d.KH * d.KW > 1024) return -1;
  if (d.rot_ncols > 0 && (!d.rot_tab || d.rot_dh < 2 || (d.rot_dh & 1))) return -2;
  if (d.a_mode == 1 && (!d.a_coef || d.a_imgs_per_sample <= 0)) return -3;
  if (d.a_img_mod) return -3;  // (shared source frames: the 2-D-tiled 3 x 3 kernel only)
  const long long M = (long long)d.nimg * d.Hv * d.Wv;
  if (M >= (1LL << 31) || (long long)d.nimg * d.Hin * d.Win >= (1LL << 31)) return -4;
  if (M <= 0 || d.Cout <= 0) return 0;
  // tile choice: widest N tile that the layer fills; halve BM when the grid would not cover the 256 CUs twice
  if (d.Cout >= 128) {
    const long long blocks = (long long)cdiv(M, 128) * cdiv(d.Cout, 128);
    if (blocks >= 512) return launch<128, 128, 2, 2>(d, s);
    return launch<64, 128, 1, 4>(d, s);
  }
  if (d.Cout > 32) {
    const long long blocks = (long long)cdiv(M, 128);
    if (blocks >= 512) return launch<128, 64, 2, 2>(d, s);
    return launch<64, 64, 2, 2>(d, s);
  }
  return launch<128, 32, 4, 1>(d, s);
}
