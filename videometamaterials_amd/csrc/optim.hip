// Multi-tensor Adam and EMA for the data-parallel training step (vddp.py:1481,1633 Adam(lr, betas=(0.9,0.999));
// vddp.py:116-129 EMA).  One launch sweeps every parameter tensor through a device-resident job table:
// HBM-bound, 16 B/param read (p, g, m, v) + 12 B written.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

__global__ __launch_bounds__(256) void adam_kernel(const vmm_optim_job* __restrict__ jobs, float lr, float beta1, float beta2, float eps,
                                                   float bc1, float bc2_sqrt, float grad_scale) {
  const vmm_optim_job jb = jobs[blockIdx.y];
  const float step_size = lr / bc1;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (long long)gridDim.x * blockDim.x) {
    const float g = jb.g[i] * grad_scale;
    const float m = beta1 * jb.m[i] + (1.0f - beta1) * g;
    const float v = beta2 * jb.v[i] + (1.0f - beta2) * g * g;
    jb.m[i] = m;
    jb.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;  // torch.optim.Adam: (exp_avg_sq.sqrt() / bias_correction2_sqrt).add_(eps)
    jb.p[i] = jb.p[i] - step_size * (m / denom);
  }
}

__global__ __launch_bounds__(256) void ema_kernel(const vmm_optim_job* __restrict__ jobs, float beta, int copy_only) {
  const vmm_optim_job jb = jobs[blockIdx.y];  // p = online weights, m = EMA weights
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (long long)gridDim.x * blockDim.x)
    jb.m[i] = copy_only ? jb.p[i] : jb.m[i] * beta + (1.0f - beta) * jb.p[i];  // vddp.py:126-129
}

}  // namespace

extern "C" int vmm_adam_step(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float lr, float beta1, float beta2, float eps,
                             int32_t step, float grad_scale, vmm_stream_t stream) {
  if (njobs <= 0) return 0;
  const float bc1 = 1.0f - powf(beta1, (float)step);
  const float bc2s = sqrtf(1.0f - powf(beta2, (float)step));
  const int bx = (int)max(1LL, min((long long)cdiv(max_n, 256 * 4), 64LL));
  hipLaunchKernelGGL(adam_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, lr, beta1, beta2, eps, bc1, bc2s, grad_scale);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_ema_step(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float beta, int32_t copy_only, vmm_stream_t stream) {
  if (njobs <= 0) return 0;
  const int bx = (int)max(1LL, min((long long)cdiv(max_n, 256 * 4), 64LL));
  hipLaunchKernelGGL(ema_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, beta, copy_only);
  VMM_LAUNCH_CHECK();
  return 0;
}
