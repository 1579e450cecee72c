// Training-sample assembly on the device: what Dataset.__getitem__ of the reference computes per sample on the
// host once its GIFs are decoded (vddp.py:1304-1397), for a whole minibatch gathered by index from a dataset that stays resident in HBM
// as the decoded bytes (5 B per pixel and frame instead of 16).
//
// Per output element, in the reference's operation order with every intermediate rounded to fp32 (no contraction: bit-exact against
// the reference's torch expressions, tests/test_gpu_dataset.py):
//   v = u8 / 255                              ToTensor
//   v = v * un_range + un_lo                  unnorm with the sample's own range        vddp.py:1298
//   v = 0 where the topology byte is 0        vddp.py:1337, 1363, 1385
//   v = (v - g_lo) / g_range                  normalize with the global range           vddp.py:1295
// frames beyond the GIF's length are zero (cast_num_frames pads, vddp.py:1115-1124).  HBM-bound: 2 bytes read, 4 written per element.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/vmm_kernels.h"

namespace {

struct DSArgs {
  const uint8_t* frames;
  const int32_t* index;
  const int32_t* chan_src;
  const float* coef;
  float* out;
  long long HW;
  int n_fields, f, nch, T_out;
};

__device__ __forceinline__ float field_value(unsigned byte, unsigned topo, bool plain, float un_range, float un_lo, float g_lo, float g_range) {
  float v = __fdiv_rn((float)byte, 255.0f);
  if (plain) return v;
  v = __fadd_rn(__fmul_rn(v, un_range), un_lo);
  if (topo == 0) v = 0.0f;
  return __fdiv_rn(__fsub_rn(v, g_lo), g_range);
}

template <bool VEC>
__global__ __launch_bounds__(256) void fields_to_samples_kernel(DSArgs a) {
  const int b = blockIdx.z / a.nch, c = blockIdx.z - b * a.nch;
  const int t = blockIdx.y;
  const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * (VEC ? 4 : 1);
  if (p >= a.HW) return;
  float* o = a.out + (((long long)b * a.nch + c) * a.T_out + t) * a.HW + p;
  if (t >= a.f) {
    if (VEC) *reinterpret_cast<float4*>(o) = make_float4(0.f, 0.f, 0.f, 0.f);
    else *o = 0.0f;
    return;
  }
  const long long n = a.index[b];
  const int src = a.chan_src[c];
  const bool plain = (src & 0x100) != 0;
  const uint8_t* base = a.frames + (n * a.n_fields * a.f + t) * a.HW + p;
  const uint8_t* sp = base + (long long)(src & 0xff) * a.f * a.HW;
  const float* k = a.coef + (n * a.nch + c) * 4;
  const float un_range = k[0], un_lo = k[1], g_lo = k[2], g_range = k[3];
  if (VEC) {
    const uint32_t s4 = *reinterpret_cast<const uint32_t*>(sp);
    const uint32_t t4 = *reinterpret_cast<const uint32_t*>(base);  // field 0 = topology
    float4 r;
    r.x = field_value(s4 & 0xff, t4 & 0xff, plain, un_range, un_lo, g_lo, g_range);
    r.y = field_value((s4 >> 8) & 0xff, (t4 >> 8) & 0xff, plain, un_range, un_lo, g_lo, g_range);
    r.z = field_value((s4 >> 16) & 0xff, (t4 >> 16) & 0xff, plain, un_range, un_lo, g_lo, g_range);
    r.w = field_value(s4 >> 24, t4 >> 24, plain, un_range, un_lo, g_lo, g_range);
    *reinterpret_cast<float4*>(o) = r;
  } else {
    *o = field_value(*sp, *base, plain, un_range, un_lo, g_lo, g_range);
  }
}

}  // namespace

extern "C" int vmm_fields_to_samples(const uint8_t* frames, int32_t n_fields, int32_t f, int64_t HW, const int32_t* index, int32_t B,
                                     const int32_t* chan_src, const float* coef, int32_t nch, int32_t T_out, float* out, vmm_stream_t stream) {
  if (!frames || !index || !chan_src || !coef || !out || n_fields < 1 || n_fields > 255 || f < 1 || HW < 1 || B < 1 || nch < 1 || T_out < 1 ||
      (long long)B * nch > 65535 || T_out > 65535)
    return -1;
  DSArgs a{frames, index, chan_src, coef, out, (long long)HW, n_fields, f, nch, T_out};
  const bool vec = (HW & 3) == 0 && ((uintptr_t)frames & 3) == 0 && ((uintptr_t)out & 15) == 0;
  const long long per = vec ? 1024 : 256;
  dim3 grid((unsigned)((HW + per - 1) / per), (unsigned)T_out, (unsigned)(B * nch));
  if (vec) hipLaunchKernelGGL(fields_to_samples_kernel<true>, grid, dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(fields_to_samples_kernel<false>, grid, dim3(256), 0, (hipStream_t)stream, a);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
