// Geometry extraction from sampled videos (integer work, bit-exact against the reference), gfx950.
//
// Trainer.save_preds (vddp.py:1890-1913) turns every sampled video into the binary topology of one quarter of the unit cell
// ("void iff u_2 stays within 0.02 of its zero value in every frame" for Lagrangian videos, "first channel of the first frame > 0.5"
// otherwise), transposes it, and clean_pred (src/utils.py:32-82) removes pixels without neighbours and keeps the largest
// 4-connected component, the first one in networkx's iteration order winning ties.  The reference does this on the host with
// Python loops and a networkx graph per sample; here one workgroup per sample does it out of LDS:
//   1. topology bit per pixel of the quarter (11 coalesced reads per pixel for the Lagrangian rule), transposed on the fly
//   2. isolated-pixel removal (simultaneous rule == the reference's in-place raster scan: an occupied neighbour of an occupied
//      pixel can not have been removed before it is visited)
//   3. connected components by min-label propagation over the pixels that have an occupied neighbour (networkx only sees edges)
//   4. component sizes (LDS atomics) and order keys: a component's place in networkx's iteration is the insertion time of its
//      first node = its earliest axis-0 edge in row-major order, or, without any, after all of those its earliest axis-1 edge
//   5. the largest component, smallest key among equals, written as int32 0 / 1 (row-major [x][y], as geometries.csv)
// A sample without any pair of neighbouring pixels makes the reference raise IndexError (src/utils.py:73); here it yields an
// all-zero geometry.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

constexpr int GEO_MAXP = 96;  // quarter side (frames up to 192 x 192)

__global__ __launch_bounds__(256) void geometry_kernel(const float* __restrict__ v, int C, int T, int P, int mode, float zero_u2,
                                                       int* __restrict__ out) {
  extern __shared__ int lds[];
  const int Q = P / 2, NP = Q * Q;
  int* img = lds;           // [NP] occupancy
  int* lab = img + NP;      // [NP] component label (pixel index) or -1
  int* cnt = lab + NP;      // [NP] size of the component rooted at this pixel
  int* key = cnt + NP;      // [NP] order key of the component rooted at this pixel
  __shared__ int changed, best_cnt, best_key, best_lab;
  const int n = blockIdx.x, tid = threadIdx.x;
  const float* vs = v + (long long)n * C * T * P * P;

  // 1. topology, transposed: geometry pixel (x, y) <- quarter pixel (row y, column x)
  for (int i = tid; i < NP; i += 256) {
    const int x = i / Q, y = i - x * Q;
    int bit;
    if (mode == 0) {  // Lagrangian with several frames: upper-left quarter mirrored along the rows, channel 1 in all frames (vddp.py:1900-1911)
      const int h = Q - 1 - y, w = x;
      const float allowed = __fadd_rn(__fmul_rn(1e-5f, fabsf(zero_u2)), 0.02f);  // torch.isclose: |a - b| <= atol + rtol |b|, two fp32 roundings
      bool all_close = true;
      for (int t = 0; t < T; ++t) {
        const float a = vs[(((long long)1 * T + t) * P + h) * P + w];
        all_close = all_close && (fabsf(a - zero_u2) <= allowed);
      }
      bit = all_close ? 0 : 1;
    } else {          // Eulerian, or a single frame: bottom-left quarter, channel 0, frame 0, binarised at 0.5 (vddp.py:1895-1897, src/utils.py:34-37)
      const int h = Q + y, w = x;
      bit = vs[(long long)h * P + w] > 0.5f ? 1 : 0;
    }
    img[i] = bit;
  }
  __syncthreads();
  // 2. pixels whose four neighbours are all empty go; a neighbour beyond the border counts as present (src/utils.py:46-62)
  for (int i = tid; i < NP; i += 256) {
    const int x = i / Q, y = i - x * Q;
    const int up = x > 0 ? img[i - Q] : 1, dn = x < Q - 1 ? img[i + Q] : 1, lf = y > 0 ? img[i - 1] : 1, rt = y < Q - 1 ? img[i + 1] : 1;
    lab[i] = (img[i] && (up | dn | lf | rt)) ? 1 : 0;  // staged in lab so that every thread reads the original image
  }
  __syncthreads();
  for (int i = tid; i < NP; i += 256) img[i] = lab[i];
  __syncthreads();
  // 3. nodes = occupied pixels with an occupied neighbour inside the image; labels = smallest pixel index of the component
  for (int i = tid; i < NP; i += 256) {
    const int x = i / Q, y = i - x * Q;
    const int nb = (x > 0 ? img[i - Q] : 0) | (x < Q - 1 ? img[i + Q] : 0) | (y > 0 ? img[i - 1] : 0) | (y < Q - 1 ? img[i + 1] : 0);
    lab[i] = (img[i] && nb) ? i : -1;
    cnt[i] = 0;
    key[i] = 0x7fffffff;
  }
  if (tid == 0) { best_cnt = 0; best_key = 0x7fffffff; best_lab = -1; }
  __syncthreads();
  for (;;) {
    if (tid == 0) changed = 0;
    __syncthreads();
    int any = 0;
    for (int i = tid; i < NP; i += 256) {
      int l = lab[i];
      if (l < 0) continue;
      const int x = i / Q, y = i - x * Q;
      int m = l;
      if (x > 0 && lab[i - Q] >= 0) m = min(m, lab[i - Q]);
      if (x < Q - 1 && lab[i + Q] >= 0) m = min(m, lab[i + Q]);
      if (y > 0 && lab[i - 1] >= 0) m = min(m, lab[i - 1]);
      if (y < Q - 1 && lab[i + 1] >= 0) m = min(m, lab[i + 1]);
      if (m < l) { atomicMin(&lab[i], m); any = 1; }  // (labels only decrease: racing readers see a valid, possibly newer label)
    }
    if (any) changed = 1;
    __syncthreads();
    const int c = changed;
    __syncthreads();
    if (!c) break;
  }
  // 4. sizes and networkx order keys per component (root = its smallest pixel index)
  for (int i = tid; i < NP; i += 256) {
    const int l = lab[i];
    if (l < 0) continue;
    const int x = i / Q, y = i - x * Q;
    atomicAdd(&cnt[l], 1);
    if (x < Q - 1 && lab[i + Q] >= 0) atomicMin(&key[l], i);                 // axis-0 edge (x, y)-(x+1, y) starting here
    else if (y < Q - 1 && lab[i + 1] >= 0) atomicMin(&key[l], NP + i);       // axis-1 edge (x, y)-(x, y+1): only counts without axis-0 edges
  }
  __syncthreads();
  // 5. largest component, earliest in networkx's order among equals
  for (int i = tid; i < NP; i += 256)
    if (lab[i] == i) atomicMax(&best_cnt, cnt[i]);
  __syncthreads();
  for (int i = tid; i < NP; i += 256)
    if (lab[i] == i && cnt[i] == best_cnt) atomicMin(&best_key, key[i]);
  __syncthreads();
  for (int i = tid; i < NP; i += 256)
    if (lab[i] == i && cnt[i] == best_cnt && key[i] == best_key) best_lab = i;  // keys are distinct between components
  __syncthreads();
  for (int i = tid; i < NP; i += 256) out[(long long)n * NP + i] = (lab[i] >= 0 && lab[i] == best_lab) ? 1 : 0;
}

}  // namespace

// videos: fp32 (N, C, T, P, P) contiguous (the sampler's NCTHW output); out: int32 [N][(P/2)^2].
// lagrangian != 0 and T > 1: the u_2 rule (needs C >= 2); otherwise the first-frame / first-channel rule.
extern "C" int vmm_extract_geometry(const float* videos, int32_t N, int32_t C, int32_t T, int32_t P, int32_t lagrangian, float zero_u_2,
                                    int32_t* out, vmm_stream_t stream) {
  if (N < 0 || C < 1 || T < 1 || P < 2 || (P & 1) || P / 2 > GEO_MAXP) return -1;
  const int mode = (lagrangian && T > 1) ? 0 : 1;
  if (mode == 0 && C < 2) return -1;
  if (N == 0) return 0;
  const int Q = P / 2;
  const size_t shm = sizeof(int) * 4 * Q * Q;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&geometry_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 152 * 1024);  // + static
    attr_set = true;
  }
  hipLaunchKernelGGL(geometry_kernel, dim3((unsigned)N), dim3(256), shm, (hipStream_t)stream, videos, C, T, P, mode, zero_u_2, out);
  VMM_LAUNCH_CHECK();
  return 0;
}
