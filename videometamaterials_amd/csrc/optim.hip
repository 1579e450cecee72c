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

// ---- loss scaling on the device (the `train_precision = "fp16"` leg).  The reference trains under Accelerate(mixed_precision='fp16') (main.py:34):
// accelerator.backward scales the loss, the wrapped optimizer's step() unscales, skips the update when a gradient is inf / nan, and update() moves the
// scale (torch.cuda.amp.GradScaler: init 2^16, x 0.5 on overflow, x 2 after 2000 clean steps; vddp.py:1629-1633).  The same state machine, without a host
// round trip: state[0] scale, [1] growth tracker, [2] found_inf of the step in flight, [3] steps skipped so far, [4] optimiser steps taken.
__global__ void scaler_init_kernel(float* __restrict__ st, float init_scale) {
  st[0] = init_scale; st[1] = 0.f; st[2] = 0.f; st[3] = 0.f; st[4] = 0.f;
}

__global__ __launch_bounds__(256) void grad_nonfinite_kernel(const float* __restrict__ g, long long n, float* __restrict__ st) {
  bool bad = false;
  const long long n4 = n >> 2;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(g)[i];
    // (x - x is 0 for every finite x and nan for inf / nan: one subtract and one compare per element)
    bad |= (v.x - v.x != 0.f) | (v.y - v.y != 0.f) | (v.z - v.z != 0.f) | (v.w - v.w != 0.f);
  }
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) { const float x = g[(n4 << 2) + threadIdx.x]; bad |= x - x != 0.f; }
  if (__any(bad) && (threadIdx.x & 63) == 0) st[2] = 1.0f;  // (benign race: every writer stores the same value)
}

__global__ __launch_bounds__(256) void adam_scaled_kernel(const vmm_optim_job* __restrict__ jobs, float lr, float beta1, float beta2, float eps,
                                                          float extra_scale, const float* __restrict__ st) {
  if (st[2] != 0.f) return;  // an overflowed step: parameters and moments stay as they are (GradScaler.step skips optimizer.step())
  const vmm_optim_job jb = jobs[blockIdx.y];
  const float step = st[4] + 1.0f;
  const float bc1 = 1.0f - powf(beta1, step), bc2_sqrt = sqrtf(1.0f - powf(beta2, step));
  const float step_size = lr / bc1, grad_scale = extra_scale / st[0];
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < jb.n; i += (long long)gridDim.x * blockDim.x) {
    const float g = jb.g[i] * grad_scale;
    const float m = beta1 * jb.m[i] + (1.0f - beta1) * g;
    const float v = beta2 * jb.v[i] + (1.0f - beta2) * g * g;
    jb.m[i] = m;
    jb.v[i] = v;
    const float denom = sqrtf(v) / bc2_sqrt + eps;
    jb.p[i] = jb.p[i] - step_size * (m / denom);
  }
}

__global__ void scaler_update_kernel(float* __restrict__ st, float growth, float backoff, float interval) {
  if (st[2] != 0.f) {  // GradScaler.update(): found_inf -> scale * backoff, tracker reset
    st[0] *= backoff;
    st[1] = 0.f;
    st[3] += 1.0f;
  } else {
    st[4] += 1.0f;
    st[1] += 1.0f;
    if (st[1] >= interval) { st[0] *= growth; st[1] = 0.f; }
  }
  st[2] = 0.f;
}

}  // namespace

extern "C" int vmm_scaler_init(float* state, float init_scale, vmm_stream_t stream) {
  if (!state || !(init_scale > 0.f)) return -1;
  hipLaunchKernelGGL(scaler_init_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, init_scale);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_grad_nonfinite(const float* g, int64_t n, float* state, vmm_stream_t stream) {
  if (!state || (n > 0 && !g) || ((uintptr_t)g & 15)) return -1;
  if (n <= 0) return 0;
  const int blocks = (int)max(1LL, min((long long)cdiv(n, 256 * 16), 2048LL));
  hipLaunchKernelGGL(grad_nonfinite_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, g, (long long)n, state);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_adam_step_scaled(const vmm_optim_job* jobs_dev, int32_t njobs, int64_t max_n, float lr, float beta1, float beta2, float eps,
                                    float extra_scale, const float* state, vmm_stream_t stream) {
  if (!state) return -1;
  if (njobs <= 0) return 0;
  const int bx = (int)max(1LL, min((long long)cdiv(max_n, 256 * 4), 64LL));
  hipLaunchKernelGGL(adam_scaled_kernel, dim3(bx, njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev, lr, beta1, beta2, eps, extra_scale, state);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_scaler_update(float* state, float growth, float backoff, int32_t interval, vmm_stream_t stream) {
  if (!state || !(growth >= 1.f) || !(backoff > 0.f && backoff <= 1.f) || interval < 1) return -1;
  hipLaunchKernelGGL(scaler_update_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, state, growth, backoff, (float)interval);
  VMM_LAUNCH_CHECK();
  return 0;
}

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
