// Row passes of linear attention (vddp.py:313-378; forward apply and the backward row pass) on the fp32 matrix cores, gfx950.
// dim_head = 32.  Backward:
//
// Per (frame, head) the three per-row products are [rows x 32] . [32 x 32] GEMMs against the frame's ctx / dctx:
//   G^T   = ctx  . dout^T      dq[n,d] = scale p[n,d] (G[n,d] - sum_d p G),  p = softmax_d(q[n,:])
//   dkt^T = dctx . v^T         dk[n,d] = kt[n,d] (dkt[n,d] / HW - R[d]),     kt = exp(k - max_d) / sum_d (statistics over pixels: kstat)
//   dv^T  = dctx^T . kt^T      dv[n,e] = dv^T[e,n] / HW                       R[d] = sum_e dctx[d,e] ctx[d,e]
// The thread-per-(row, head) kernel in attention_bwd.hip spends 3 x 1024 FMAs and as many LDS operand reads per row on these; here a
// wave owns one head, keeps ctx / dctx / dctx^T as MFMA "A" operands in registers (48 VGPRs) for all its rows and issues
// 48 v_mfma_f32_32x32x2_f32 per tile of 32 rows.  The contraction index of MFMA s is e = 16 * (lane >> 5) + s on both operands, so a
// lane's "B" operand is a contiguous half row of dout / v / kt; the accumulator (lane = row n, registers = d) is exactly the layout the
// per-row softmax algebra wants, the second lane half supplying the other 16 channels (one shuffle per row reduction).
// Rows are staged per wave through padded LDS tiles (coalesced 128-byte head slices in, conflict-free 16-byte operand reads out); the
// next tile's loads are issued before the current tile's MFMAs.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32, RP = 36;                    // LDS row pitch in floats
constexpr int TSZ = 32 * RP;                       // one staged tensor tile
constexpr int WSTRIDE = 4 * TSZ + 96;              // per wave: dout | q | k | v tiles + kmax[32] | kinv[32] | R[32]

struct LBArgs {
  const float *qkv, *dout, *ctx, *dctx, *kstat;
  float* dqkv;
  int ldqkv, lddo, HW, heads, rows_per_block;
  float scale;
};

__global__ __launch_bounds__(256) void linattn_bwd_rows_mfma_kernel(const LBArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lk = lane >> 5;
  const long long frame = blockIdx.y;
  const int head = blockIdx.z * 4 + wave;
  const int hid = a.heads * DH;
  float* Dt = smem + wave * WSTRIDE;
  float* Qt = Dt + TSZ;
  float* Kt = Qt + TSZ;
  float* Vt = Kt + TSZ;
  float* aux = Vt + TSZ;
  const float* C = a.ctx + (frame * a.heads + head) * DH * DH;
  const float* DC = a.dctx + (frame * a.heads + head) * DH * DH;
  // MFMA "A" operands: row = l31, contraction element 16 lk + s
  float ctxA[16], dcA[16], dcT[16];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const f32x4 u = *reinterpret_cast<const f32x4*>(C + l31 * DH + 16 * lk + 4 * c);
    const f32x4 w = *reinterpret_cast<const f32x4*>(DC + l31 * DH + 16 * lk + 4 * c);
    ctxA[4 * c] = u.x; ctxA[4 * c + 1] = u.y; ctxA[4 * c + 2] = u.z; ctxA[4 * c + 3] = u.w;
    dcA[4 * c] = w.x; dcA[4 * c + 1] = w.y; dcA[4 * c + 2] = w.z; dcA[4 * c + 3] = w.w;
  }
#pragma unroll
  for (int s = 0; s < 16; ++s) dcT[s] = DC[(16 * lk + s) * DH + l31];
  {
    float r = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) r = fmaf(ctxA[s], dcA[s], r);
    r += __shfl_xor(r, 32, 64);
    const float* ks = a.kstat + (frame * a.heads + head) * 2 * DH;  // max[32] | 1/sum[32]
    aux[lane] = ks[lane];
    if (lk == 0) aux[64 + l31] = r;
  }
  const float invHW = 1.0f / (float)a.HW;
  const int n_begin = blockIdx.x * a.rows_per_block, n_end = min(n_begin + a.rows_per_block, a.HW);

  // staging: item idx = pass * 64 + lane -> (row idx >> 3, 16-byte chunk idx & 7); tensors dout, q, k, v
  f32x4 pf[4][4];
  auto issue = [&](int n0) {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int idx = ps * 64 + lane, r = idx >> 3, c = (idx & 7) * 4;
      const int n = n0 + r;
      const f32x4 z = {0.f, 0.f, 0.f, 0.f};
      if (n < n_end) {
        const float* src = a.qkv + (frame * a.HW + n) * a.ldqkv + head * DH + c;
        pf[0][ps] = *reinterpret_cast<const f32x4*>(a.dout + (frame * a.HW + n) * a.lddo + head * DH + c);
        pf[1][ps] = *reinterpret_cast<const f32x4*>(src);
        pf[2][ps] = *reinterpret_cast<const f32x4*>(src + hid);
        pf[3][ps] = *reinterpret_cast<const f32x4*>(src + 2 * hid);
      } else {
        pf[0][ps] = z; pf[1][ps] = z; pf[2][ps] = z; pf[3][ps] = z;
      }
    }
  };
  auto stage = [&]() {
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
      const int idx = ps * 64 + lane, r = idx >> 3, c = (idx & 7) * 4;
#pragma unroll
      for (int t = 0; t < 4; ++t) *reinterpret_cast<f32x4*>(Dt + t * TSZ + r * RP + c) = pf[t][ps];
    }
  };
  // a lane's half row 16 lk .. + 15 of a staged tensor (MFMA "B" operand), and its accumulator-layout channels (chunks 2 c + lk)
  auto half_row = [&](float (&dst)[16], const float* tile) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + l31 * RP + 16 * lk + 4 * c);
      dst[4 * c] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
    }
  };
  auto acc_row = [&](float (&dst)[16], const float* tile) {
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(tile + l31 * RP + 4 * (2 * c + lk));
      dst[4 * c] = v.x; dst[4 * c + 1] = v.y; dst[4 * c + 2] = v.z; dst[4 * c + 3] = v.w;
    }
  };
  auto store_acc_layout = [&](float* row, const float (&v)[16]) {  // register r = 4 c + j <-> channel 4 (2 c + lk) + j
#pragma unroll
    for (int c = 0; c < 4; ++c) *reinterpret_cast<f32x4*>(row + 4 * (2 * c + lk)) = (f32x4){v[4 * c], v[4 * c + 1], v[4 * c + 2], v[4 * c + 3]};
  };

  issue(n_begin);
  for (int n0 = n_begin; n0 < n_end; n0 += 32) {
    __syncthreads();  // the previous tile's operand reads are done (first pass: aux is complete)
    stage();
    __syncthreads();
    if (n0 + 32 < n_end) issue(n0 + 32);
    const int n = n0 + l31;
    const bool valid = n < n_end;
    float* orow = a.dqkv + (frame * a.HW + n) * a.ldqkv + head * DH;
    float b[16], x[16];
    f32x16 acc;
    // ---- dq
    half_row(b, Dt);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(ctxA[s], b[s], acc, 0, 0, 0);
    acc_row(x, Qt);
    {
      float mx = x[0];
#pragma unroll
      for (int r = 1; r < 16; ++r) mx = fmaxf(mx, x[r]);
      mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
      float sum = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] = __expf(x[r] - mx); sum += x[r]; }
      sum += __shfl_xor(sum, 32, 64);
      const float inv = 1.0f / sum;
      float pg = 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) { x[r] *= inv; pg = fmaf(x[r], acc[r], pg); }
      pg += __shfl_xor(pg, 32, 64);
#pragma unroll
      for (int r = 0; r < 16; ++r) x[r] = a.scale * x[r] * (acc[r] - pg);
      if (valid) store_acc_layout(orow, x);
    }
    // ---- dk
    half_row(b, Vt);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dcA[s], b[s], acc, 0, 0, 0);
    acc_row(x, Kt);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int d = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const float kt = __expf(x[r] - aux[d]) * aux[32 + d];
      x[r] = kt * (acc[r] * invHW - aux[64 + d]);
    }
    if (valid) store_acc_layout(orow + hid, x);
    // ---- dv
    half_row(b, Kt);
#pragma unroll
    for (int s = 0; s < 16; ++s) b[s] = __expf(b[s] - aux[16 * lk + s]) * aux[32 + 16 * lk + s];
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(dcT[s], b[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) x[r] = acc[r] * invHW;
    if (valid) store_acc_layout(orow + 2 * hid, x);
  }
}

// ---------------------------------------------------------------- forward row pass (vmm_linattn_apply)
// out[n, e] = sum_d ctx[d][e] softmax_d(q[n, :])[d] scale  as  O^T = ctx^T . qt^T: ctx^T resident as the "A" operand, a lane's half row of
// q is its "B" operand after the row softmax (the other half row sits in the partner lane: one shuffle per reduction); no LDS.
__global__ __launch_bounds__(256) void linattn_apply_mfma_kernel(const float* __restrict__ qkv, int ldqkv, const float* __restrict__ ctx,
                                                                 float* __restrict__ out, int ldo, int HW, int heads, int rows_per_block, float scale) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lk = lane >> 5;
  const long long frame = blockIdx.y;
  const int head = blockIdx.z * 4 + wave;
  const float* C = ctx + (frame * heads + head) * DH * DH;
  float cT[16];  // ctx^T: row e = l31, contraction d = 16 lk + s
#pragma unroll
  for (int s = 0; s < 16; ++s) cT[s] = C[(16 * lk + s) * DH + l31];
  const int n_begin = blockIdx.x * rows_per_block, n_end = min(n_begin + rows_per_block, HW);
  for (int n0 = n_begin; n0 < n_end; n0 += 32) {
    const int n = n0 + l31;
    const bool valid = n < n_end;
    float q[16];
    const float* src = qkv + (frame * HW + (valid ? n : n_begin)) * ldqkv + head * DH + 16 * lk;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(src + 4 * c);
      q[4 * c] = v.x; q[4 * c + 1] = v.y; q[4 * c + 2] = v.z; q[4 * c + 3] = v.w;
    }
    float mx = q[0];
#pragma unroll
    for (int s = 1; s < 16; ++s) mx = fmaxf(mx, q[s]);
    mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
    float sum = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) { q[s] = __expf(q[s] - mx); sum += q[s]; }
    sum += __shfl_xor(sum, 32, 64);
    const float sc = scale / sum;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(cT[s], q[s] * sc, acc, 0, 0, 0);
    if (valid) {  // register r = 4 c + j <-> channel e = 4 (2 c + lk) + j of row n
      float* op = out + (frame * HW + n) * ldo + head * DH;
#pragma unroll
      for (int c = 0; c < 4; ++c)
        *reinterpret_cast<f32x4*>(op + 4 * (2 * c + lk)) = (f32x4){acc[4 * c], acc[4 * c + 1], acc[4 * c + 2], acc[4 * c + 3]};
    }
  }
}

}  // namespace

// Row pass of vmm_linattn_apply for heads % 4 == 0 (returns 1 and launches nothing otherwise).
extern "C" int vmm_linattn_apply_mfma(const float* qkv, int32_t ldqkv, const float* ctx, float* out, int32_t ldo, int32_t frames, int32_t HW,
                                      int32_t heads, float scale, vmm_stream_t stream) {
  if (heads % 4 || (ldqkv & 3) || (ldo & 3)) return 1;
  if (frames <= 0 || HW <= 0) return 0;
  const long long groups = (long long)frames * (heads / 4);
  const long long chunks = max(1LL, min((long long)cdiv(HW, 32), cdiv(4096, groups)));
  const int rows_per_block = (int)(cdiv(cdiv(HW, chunks), 32) * 32);
  hipLaunchKernelGGL(linattn_apply_mfma_kernel, dim3((unsigned)cdiv(HW, rows_per_block), (unsigned)frames, (unsigned)(heads / 4)), dim3(256), 0,
                     (hipStream_t)stream, qkv, ldqkv, ctx, out, ldo, HW, heads, rows_per_block, scale);
  VMM_LAUNCH_CHECK();
  return 0;
}

// Row pass of vmm_linattn_bwd for heads % 4 == 0 (returns 1 and launches nothing otherwise): dqkv rows from qkv, dout, ctx, dctx, kstat.
extern "C" int vmm_linattn_bwd_rows_mfma(const float* qkv, int32_t ldqkv, const float* dout, int32_t lddo, const float* ctx, const float* dctx,
                                         const float* kstat, float* dqkv, int32_t frames, int32_t HW, int32_t heads, float scale,
                                         vmm_stream_t stream) {
  if (heads % 4 || (ldqkv & 3) || (lddo & 3)) return 1;
  if (frames <= 0 || HW <= 0) return 0;
  LBArgs a{qkv, dout, ctx, dctx, kstat, dqkv, ldqkv, lddo, HW, heads, 0, scale};
  // rows per workgroup: a multiple of 32, ~2000 workgroups over the launch
  const long long groups = (long long)frames * (heads / 4);
  long long chunks = max(1LL, min((long long)cdiv(HW, 32), cdiv(2048, groups)));
  a.rows_per_block = (int)(cdiv(cdiv(HW, chunks), 32) * 32);
  const size_t shm = sizeof(float) * 4 * WSTRIDE;
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&linattn_bwd_rows_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(linattn_bwd_rows_mfma_kernel, dim3((unsigned)cdiv(HW, a.rows_per_block), (unsigned)frames, (unsigned)(heads / 4)), dim3(256), shm,
                     (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
