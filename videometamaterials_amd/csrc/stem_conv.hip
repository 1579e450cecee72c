// Stem convolution on the split-bf16 matrix cores, gfx950: Conv3d(C <= 4, 64, (1,k,k)) pad k/2, k <= 8 (vddp.py:600 init_conv, k = 7).
//
// With three input channels the contraction is short (K = 147) and the output wide (64 fp32 per pixel, 207 MB at batch 8): the generic
// implicit-GEMM kernel spends its time decoding (tap, channel) positions for a gather of single pixels and reaches 70 TFLOP/s.  Here
//   * a workgroup owns a 16 x 16 pixel tile; the (16 + k - 1) x 24 pixel neighbourhood is staged ONCE in LDS as raw fp32 rows of 4 channels
//     (16 bytes per pixel);
//   * K is laid out as (kernel row, tap 0..7, channel 0..3): one k16 step = four neighbouring taps of one kernel row, so the eight values a
//     lane contributes are two NEIGHBOURING pixels of the patch = 32 contiguous bytes of LDS, split into bf16 hi | lo in registers
//     (tap 7 and the padding channel carry zero weights: vmm_pack_weights fmt 7);
//   * the 2k k16 steps' weight fragments (56 KB for k = 7) sit in LDS for the workgroup's whole life; workgroups walk a strided list of tiles.
// 12 MFMAs per step and wave against two 32-byte LDS reads and ~40 vector instructions.
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int SP = 24;  // patch pitch in pixels: 16 + 7 taps to the right + 1

struct StemArgs {
  const float* x;    // rows [nimg * H * W][4]
  const uint4* w;    // fmt 7 fragment planes: [2 column tiles][2k steps][hi | lo][64 lanes]
  const float* bias;
  float* out; int ldo;
  int nimg, H, W, k, tiles_x, tiles_y, ntiles;
};

__device__ __forceinline__ void split_pair(float x0, float x1, unsigned& hi, unsigned& lo) { hi = split_bf16_pair(x0, x1, lo); }

template <typename ST>
__global__ __launch_bounds__(256, 2) void stem_conv_kernel(const StemArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  f32x4* patch = reinterpret_cast<f32x4*>(smem);                 // [23][SP] pixels
  uint4* wl = reinterpret_cast<uint4*>(smem + 23 * SP * 16);     // [2][2k][2][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int k = a.k, pad = k >> 1, steps = 2 * k;
  const int H = a.H, W = a.W;

  for (int i = tid; i < 2 * steps * 2 * 64; i += 256) wl[i] = a.w[i];

  // this lane's two pixels of the wave's 4 x 16 strip, as patch positions of kernel element (0, 0)
  int pbase[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) pbase[i] = (4 * wave + 2 * i + (lrow >> 4)) * SP + (lrow & 15) + 2 * lk;

  for (int tile = blockIdx.x; tile < a.ntiles; tile += gridDim.x) {
    const int img = tile / (a.tiles_x * a.tiles_y), t2 = tile - img * (a.tiles_x * a.tiles_y);
    const int ty0 = (t2 / a.tiles_x) * 16, tx0 = (t2 - (t2 / a.tiles_x) * a.tiles_x) * 16;
    __syncthreads();  // the previous tile's reads of the patch are done (first pass: nothing)
    for (int i = tid; i < (15 + k) * SP; i += 256) {
      const int py = i / SP, px = i - py * SP;
      const int h = ty0 - pad + py, w = tx0 - pad + px;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (h >= 0 && h < H && w >= 0 && w < W) v = *reinterpret_cast<const f32x4*>(a.x + ((long long)(img * H + h) * W + w) * 4);
      patch[i] = v;
    }
    __syncthreads();

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
      for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int kh = 0; kh < k; ++kh) {
#pragma unroll
      for (int half = 0; half < 2; ++half) {
        const int ks = kh * 2 + half;
        uint4 wh[2], wlo[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          wh[j] = wl[((j * steps + ks) * 2 + 0) * 64 + lane];
          wlo[j] = wl[((j * steps + ks) * 2 + 1) * 64 + lane];
        }
        uint4 ph[2], pl[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          const f32x4* q = patch + pbase[i] + kh * SP + 4 * half;
          const f32x4 v0 = q[0], v1 = q[1];
          split_pair(v0.x, v0.y, ph[i].x, pl[i].x);
          split_pair(v0.z, v0.w, ph[i].y, pl[i].y);
          split_pair(v1.x, v1.y, ph[i].z, pl[i].z);
          split_pair(v1.z, v1.w, ph[i].w, pl[i].w);
        }
        // weights are the MFMA "A" (rows = output channels), pixels the "B": a lane ends up with 4 x 4 consecutive channels of one pixel
        auto mm = [&](const uint4& wg, const uint4& px, f32x16 c) {
          return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wg), __builtin_bit_cast(bf16x8, px), c, 0, 0, 0);
        };
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mm(wh[j], pl[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mm(wlo[j], ph[i], acc[i][j]);
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
          for (int j = 0; j < 2; ++j) acc[i][j] = mm(wh[j], ph[i], acc[i][j]);
      }
    }

    // acc[i][j]: rows = output channels (r & 3) + 8 (r >> 2) + 4 lk of column tile j, column = pixel i*32 + lrow of this wave's strip
    f32x4 bv[2][4];
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int g = 0; g < 4; ++g) bv[j][g] = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + j * 32 + 8 * g + 4 * lk) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int h = ty0 + 4 * wave + 2 * i + (lrow >> 4), w = tx0 + (lrow & 15);
      if (h < H && w < W) {
        ST* o = reinterpret_cast<ST*>(a.out) + ((long long)(img * H + h) * W + w) * a.ldo + 4 * lk;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
          for (int g = 0; g < 4; ++g) {
            const f32x4 v = {acc[i][j][4 * g] + bv[j][g].x, acc[i][j][4 * g + 1] + bv[j][g].y, acc[i][j][4 * g + 2] + bv[j][g].z,
                             acc[i][j][4 * g + 3] + bv[j][g].w};
            st4(o + j * 32 + 8 * g, v);
          }
      }
    }
  }
}

}  // namespace

// x rows [nimg * H * W][4] (channels >= C are zero or carry zero weights), w = vmm_pack_weights fmt 7 of the (64, C, 1, k, k) tensor, out rows x 64.
// Returns 1 (nothing launched) unless Cout == 64, 1 <= k <= 8, k odd.
template <typename ST>
static int stem_run(const float* x, const float* w_frag, const float* bias, float* out, int32_t ldo, int32_t nimg, int32_t H,
                    int32_t W, int32_t Cout, int32_t k, vmm_stream_t stream) {
  if (Cout != 64 || k < 1 || k > 8 || !(k & 1) || (ldo & 3)) return 1;
  if (nimg <= 0 || H <= 0 || W <= 0) return 0;
  if ((long long)nimg * H * W * 64 >= (1LL << 31)) return 1;
  StemArgs a;
  a.x = x; a.w = reinterpret_cast<const uint4*>(w_frag); a.bias = bias; a.out = out; a.ldo = ldo;
  a.nimg = nimg; a.H = H; a.W = W; a.k = k;
  a.tiles_x = (W + 15) / 16; a.tiles_y = (H + 15) / 16;
  a.ntiles = nimg * a.tiles_x * a.tiles_y;
  const size_t shm = 23 * SP * 16 + (size_t)2 * 2 * k * 2 * 64 * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_kernel<ST>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  const int grid = a.ntiles < 512 ? a.ntiles : 512;
  hipLaunchKernelGGL(stem_conv_kernel<ST>, dim3(grid), dim3(256), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_stem_conv_bf16x3(const float* x, const float* w_frag, const float* bias, float* out, int32_t ldo, int32_t nimg, int32_t H,
                                    int32_t W, int32_t Cout, int32_t k, vmm_stream_t stream) {
  return stem_run<float>(x, w_frag, bias, out, ldo, nimg, H, W, Cout, k, stream);
}
// the same with a bf16-STORED output map (the "bf16" mode; the 4-channel input rows stay fp32)
extern "C" int vmm_stem_conv_bf16x3_a16(const float* x, const float* w_frag, const float* bias, void* out, int32_t ldo, int32_t nimg, int32_t H,
                                        int32_t W, int32_t Cout, int32_t k, vmm_stream_t stream) {
  return stem_run<bf16s>(x, w_frag, bias, static_cast<float*>(out), ldo, nimg, H, W, Cout, k, stream);
}
