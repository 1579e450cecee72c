// vmm_reduce_batch: the fixed-order second stages of many weight-gradient launches in ONE launch (gfx950).
//
// The weight-gradient kernels (wgrad3x3_bf16x3.hip, wgrad1x1_bf16x3.hip) split the contraction over row slices, one workgroup per slice and channel block,
// and leave partial blocks that a second launch totals in a fixed order.  Alone, those second launches are latency chains of 5-13 us (a few hundred
// workgroups, a handful of dependent loads each); the backward pass of the training step ran 70 of them.  Here every workgroup of one launch looks up its
// job (the jobs' first-workgroup indices are ascending: a scalar scan over <= a few dozen entries) and runs that job's body -- the same code, thread for
// thread, as the per-layer kernels (wgrad_reduce.h), so the sums are the same bits.
#include "wgrad_reduce.h"
#include "../../include/vmm_kernels.h"

namespace {

__global__ __launch_bounds__(256) void reduce_batch_kernel(const vmm_reduce_job* __restrict__ jobs, int njobs) {
  __shared__ f32x4 red[8][32];  // (4 KB: also the 16 x 17 floats of the plain-rows body)
  const int v = (int)blockIdx.x;
  // the job of this workgroup: the last one whose first workgroup is <= v (wave-uniform: scalar loads from the job table)
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].wg0 <= v) lo = mid;
    else hi = mid - 1;
  }
  const vmm_reduce_job jb = jobs[lo];
  const int w = v - jb.wg0;
  if (w >= jb.wgs) return;
  if (jb.kind == 1) vmm_reduce::w9_body(jb.part, jb.nz, jb.tiles_x, jb.tiles_y, jb.out, jb.Cin, jb.Cout, jb.bias_part, jb.dbias, jb.n_main, w % jb.gx, w / jb.gx, red);
  else if (jb.kind == 2) vmm_reduce::w1_body(jb.part, jb.nz, jb.tiles_x, jb.tiles_y, jb.out, jb.Cin, jb.Cout, jb.bias_part, jb.dbias, jb.n_main, w % jb.gx, w / jb.gx, red);
  else vmm_reduce::rows_body(jb.part, jb.nz, jb.ld, jb.Cout, jb.out, w, reinterpret_cast<float (*)[17]>(&red[0][0]));
}

}  // namespace

extern "C" int vmm_reduce_batch(const vmm_reduce_job* jobs_dev, int32_t njobs, int32_t total_wgs, vmm_stream_t stream) {
  if (njobs <= 0 || total_wgs <= 0) return 0;
  if (!jobs_dev) return -1;
  hipLaunchKernelGGL(reduce_batch_kernel, dim3((unsigned)total_wgs), dim3(256), 0, (hipStream_t)stream, jobs_dev, njobs);
  VMM_LAUNCH_CHECK();
  return 0;
}
