// Backward of the temporal attention core on the fp32 matrix cores (v_mfma_f32_16x16x4_f32), gfx950.
// heads = 8, dim_head = 32, T <= 16 frames, <= 16 conditioning tokens (vddp.py:397-466 under autograd).  dqkv = gradient of the raw
// to_qkv output (the projection epilogue's rotation and q-scale are undone on the way out); token-key / value and bias gradients
// leave as one partial per workgroup and are summed in a fixed order by the reduce kernel below (no atomics).
//
// A wave owns one head and walks the workgroup's pixels; a pixel's T x T problem is one 16 x 16 MFMA tile per product:
//   S  = Q K^T, dP = dO V^T                 operands straight from global memory in "row" layout (lane = frame, 8 channels per lane group)
//   p = exp(S + bias - L), dS = p (dP - D)  on the accumulator layout (registers = query i, lanes = key j)
//   dK = dS^T Q, dV = p^T dO                the accumulators ARE the "A" operand (lane = key, contraction = query): no data movement
//   dQ = dS K                               needs dS with lane = query: one 16 x 16 transpose through a wave-private LDS tile
// and the same seven products against the sample's conditioning tokens (their dK / dV accumulate across the pixels in the MFMA
// accumulators themselves).  No workgroup barriers, no LDS operand staging: 80 MFMAs and ~40 global load instructions per pixel
// and head.  History (ms per training step over the 10 sites): thread-per-query / wave-per-key kernels (attention_bwd.hip, still the
// fallback outside the envelope) 18.9; LDS-staged workgroup-per-pixel kernel on packed fp32 FMAs 6.2; this one 3.7.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {
constexpr int DH = 32, HEADS = 8, HID = HEADS * DH;

struct TMArgs {
  const float *qkv, *ek, *ev, *bias, *O, *dO, *lse, *rot;
  float *dqkv, *part;
  int ldqkv, ldo, B, T, HW, ntok, bias_on_cond, blocks_per_sample, pstride;
  float q_scale;
};

__device__ __forceinline__ f32x4 mm(float a, float b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0); }

// row layout of one 32-float head slice: lane (row c, group g) holds channels 8 g .. 8 g + 7
__device__ __forceinline__ void load_row8(float (&dst)[8], const float* p, bool ok) {
  f32x4 u = {0.f, 0.f, 0.f, 0.f}, w = u;
  if (ok) {
    u = *reinterpret_cast<const f32x4*>(p);
    w = *reinterpret_cast<const f32x4*>(p + 4);
  }
  dst[0] = u.x; dst[1] = u.y; dst[2] = u.z; dst[3] = u.w;
  dst[4] = w.x; dst[5] = w.y; dst[6] = w.z; dst[7] = w.w;
}

__global__ __launch_bounds__(512) void temporal_attn_bwd_mfma_kernel(const TMArgs a) {
  // wave-private tiles: dS and dS(tokens) [16][17]; and the row <-> column layout changes of q, k, dO on the way in and of dQ, dK, dV on the way out
  // [16 frames][36]: the column layouts used to be 24 four-byte global loads and 24 four-byte global stores per pixel and head (3.6 TB/s)
  extern __shared__ __attribute__((aligned(16))) float smem_bw[];
  float(*xp)[2][16][17] = reinterpret_cast<float(*)[2][16][17]>(smem_bw);
  constexpr int TPITCH = 36;
  float* tiles = smem_bw + HEADS * 2 * 16 * 17 + (threadIdx.x >> 6) * 3 * 16 * TPITCH;
  const int tid = threadIdx.x, lane = tid & 63, head = tid >> 6, c = lane & 15, g = lane >> 4;
  const int T = a.T, ntok = a.ntok;
  const int b = blockIdx.x / a.blocks_per_sample, blk = blockIdx.x % a.blocks_per_sample;
  const bool tok_bias = a.bias && a.bias_on_cond;
  const bool cT = c < T, cN = c < ntok;

  // bias in accumulator layout: register r <-> query i = 4 g + r, lane <-> key j = c
  float bA[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * g + r;
    bA[r] = (a.bias && i < T && cT) ? a.bias[((long long)head * T + i) * T + c] : 0.f;
  }
  // the sample's conditioning keys / values: row layout (lane = token) and, for dQ, column layout (lane = channel, register = token 4 g + r)
  float ekr[8], evr[8], ekc[2][4];
  {
    const float* er = a.ek + ((long long)b * ntok + c) * HID + head * DH + 8 * g;
    const float* vr = a.ev + ((long long)b * ntok + c) * HID + head * DH + 8 * g;
    load_row8(ekr, er, cN);
    load_row8(evr, vr, cN);
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) ekc[h][r] = (4 * g + r < ntok) ? a.ek[((long long)b * ntok + 4 * g + r) * HID + head * DH + c + 16 * h] : 0.f;
  }
  const f32x4 zero4 = {0.f, 0.f, 0.f, 0.f};
  f32x4 dEK[2] = {zero4, zero4}, dEV[2] = {zero4, zero4}, bacc = zero4;
  float(*tile)[16][17] = xp[head];

  float nq[8], nk[8], nv[8], ng[8], nL;
  {
    const bool ok0 = cT && blk < a.HW;
    const long long r0 = (long long)b * T * a.HW + min(blk, a.HW - 1) + (long long)c * a.HW;
    const float* q0 = a.qkv + r0 * a.ldqkv + head * DH + 8 * g;
    load_row8(nq, q0, ok0);
    load_row8(nk, q0 + HID, ok0);
    load_row8(nv, q0 + 2 * HID, ok0);
    load_row8(ng, a.dO + r0 * a.ldo + head * DH + 8 * g, ok0);
    nL = ok0 ? a.lse[r0 * HEADS + head] : 0.f;
  }
  for (int pix = blk; pix < a.HW; pix += a.blocks_per_sample) {
    const long long row0 = (long long)b * T * a.HW + pix;  // row of frame t = row0 + t * HW
    const long long rc = row0 + (long long)c * a.HW;       // this lane's row in the row layouts
    // this pixel's rows were requested during the previous pixel's arithmetic (nq ..); the next pixel's are requested now: without the prefetch a
    // pixel was load latency + 80 MFMAs + stores in sequence, the same for both waves of a SIMD
    float qr[8], kr[8], vr[8], gr[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) { qr[j] = nq[j]; kr[j] = nk[j]; vr[j] = nv[j]; gr[j] = ng[j]; }
    const float Lq = nL;
    {
      const int pn = pix + a.blocks_per_sample;
      const bool okn = cT && pn < a.HW;
      const long long rn = (long long)b * T * a.HW + min(pn, a.HW - 1) + (long long)c * a.HW;
      const float* qn = a.qkv + rn * a.ldqkv + head * DH + 8 * g;
      load_row8(nq, qn, okn);
      load_row8(nk, qn + HID, okn);
      load_row8(nv, qn + 2 * HID, okn);
      load_row8(ng, a.dO + rn * a.ldo + head * DH + 8 * g, okn);
      nL = okn ? a.lse[rn * HEADS + head] : 0.f;
    }
    // column layouts (lane = channel c + 16 h, register = frame 4 g + r) of k (for dQ), q (for dK) and dO (for dV): through the wave's tiles
    // (rows of frame slots >= T were loaded as zeros)
    float kc[2][4], qc[2][4], gc[2][4];
    {
      float* tq = tiles + (c * TPITCH + 8 * g);
      *reinterpret_cast<f32x4*>(tq) = f32x4{qr[0], qr[1], qr[2], qr[3]};
      *reinterpret_cast<f32x4*>(tq + 4) = f32x4{qr[4], qr[5], qr[6], qr[7]};
      *reinterpret_cast<f32x4*>(tq + 16 * TPITCH) = f32x4{kr[0], kr[1], kr[2], kr[3]};
      *reinterpret_cast<f32x4*>(tq + 16 * TPITCH + 4) = f32x4{kr[4], kr[5], kr[6], kr[7]};
      *reinterpret_cast<f32x4*>(tq + 32 * TPITCH) = f32x4{gr[0], gr[1], gr[2], gr[3]};
      *reinterpret_cast<f32x4*>(tq + 32 * TPITCH + 4) = f32x4{gr[4], gr[5], gr[6], gr[7]};
      // (the tiles are wave-private; the workgroup barrier orders the wave's own writes before its reads AND keeps the eight head-waves on the same pixel:
      // with wave-level ordering only, the heads drift apart and the kernel is 8 % slower -- they read neighbouring 128-byte pieces of the same rows)
      __syncthreads();
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const float* tr = tiles + ((4 * g + r) * TPITCH + c + 16 * h);
          qc[h][r] = tr[0];
          kc[h][r] = tr[16 * TPITCH];
          gc[h][r] = tr[32 * TPITCH];
        }
    }
    float LA[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) LA[r] = __shfl(Lq, 4 * g + r, 64);  // lanes 0..15 hold the values of queries 0..15
    // ---- scores and dP against the frame keys and the token keys (accumulator layout: register = query 4 g + r, lane = key c)
    f32x4 S = zero4, dP = zero4, St = zero4, dPt = zero4;
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      S = mm(qr[s], kr[s], S);
      dP = mm(gr[s], vr[s], dP);
      St = mm(qr[s], ekr[s], St);
      dPt = mm(gr[s], evr[s], dPt);
    }
    // D_i = dO_i . O_i = sum_j p_ij dP_ij over the frame and token keys: the probabilities and dP are in registers anyway (register = query,
    // lane = key: a 16-lane row sum), so the attention output O (1 KB per row) is not read at all
    float p[4], ds[4], pt[4], dst[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const bool qok = 4 * g + r < T;
      p[r] = (qok && cT) ? __expf(S[r] + bA[r] - LA[r]) : 0.f;
      pt[r] = (qok && cN) ? __expf(St[r] + (tok_bias ? bA[r] : 0.f) - LA[r]) : 0.f;
      const float DA = row_sum16(fmaf(p[r], dP[r], pt[r] * dPt[r]));
      ds[r] = p[r] * (dP[r] - DA);
      dst[r] = pt[r] * (dPt[r] - DA);
      bacc[r] += ds[r] + (tok_bias ? dst[r] : 0.f);
      tile[0][4 * g + r][c] = ds[r];
      tile[1][4 * g + r][c] = dst[r];
    }
    __syncthreads();  // (the tiles are wave-private; this orders the wave's own writes before its reads)
    float dsB[4], dstB[4];  // lane = query c, register = key 4 g + r
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      dsB[r] = tile[0][c][4 * g + r];
      dstB[r] = tile[1][c][4 * g + r];
    }
    __syncthreads();  // reads done before the next pixel overwrites
    // ---- gradients: per 16-channel half h
    f32x4 dQ[2] = {zero4, zero4}, dK[2] = {zero4, zero4}, dV[2] = {zero4, zero4};
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dK[h] = mm(ds[r], qc[h][r], dK[h]);     // dK[j][d] = sum_i dS[i][j] Q[i][d]
        dV[h] = mm(p[r], gc[h][r], dV[h]);      // dV[j][d] = sum_i p[i][j] dO[i][d]
        dQ[h] = mm(dsB[r], kc[h][r], dQ[h]);    // dQ[i][d] = sum_j dS[i][j] K[j][d]
        dQ[h] = mm(dstB[r], ekc[h][r], dQ[h]);  //          + sum_t dS[i][t] EK[t][d]
        dEK[h] = mm(dst[r], qc[h][r], dEK[h]);  // accumulated over the workgroup's pixels
        dEV[h] = mm(pt[r], gc[h][r], dEV[h]);
      }
    // ---- store: register r <-> frame 4 g + r, lane <-> channel c + 16 h; undo the interleaved-pair rotation (pairs = adjacent lanes), then back to the
    // row layout through the tiles: lane = frame c, eight channels 8 g .. + 7 = two 16-byte stores per tensor
    __syncthreads();  // (the column reads above are done before the tiles are overwritten)
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int t = 4 * g + r;
        float q = dQ[h][r], k = dK[h][r];
        const float qp = __shfl_xor(q, 1, 64), kp = __shfl_xor(k, 1, 64);
        const int d = c + 16 * h;
        if (a.rot && t < T) {
          const float cs = a.rot[(t * (DH / 2) + (d >> 1)) * 2], sn = a.rot[(t * (DH / 2) + (d >> 1)) * 2 + 1];
          const float sg = (d & 1) ? -sn : sn;  // even: a c + b s, odd: b c - a s
          q = q * cs + qp * sg;
          k = k * cs + kp * sg;
        }
        float* tw = tiles + (t * TPITCH + d);
        tw[0] = q * a.q_scale;
        tw[16 * TPITCH] = k;
        tw[32 * TPITCH] = dV[h][r];
      }
    __syncthreads();
    if (cT) {
      const float* tr = tiles + (c * TPITCH + 8 * g);
      float* o = a.dqkv + rc * a.ldqkv + head * DH + 8 * g;
#pragma unroll
      for (int part = 0; part < 3; ++part) {
        *reinterpret_cast<f32x4*>(o + part * HID) = *reinterpret_cast<const f32x4*>(tr + part * 16 * TPITCH);
        *reinterpret_cast<f32x4*>(o + part * HID + 4) = *reinterpret_cast<const f32x4*>(tr + part * 16 * TPITCH + 4);
      }
    }
    __syncthreads();  // (the row reads are done before the next pixel's writes)
  }
  // ---- per-workgroup partials of the gradients shared by its pixels (summed by temporal_attn_bwd_reduce_kernel)
  float* part = a.part + (long long)blockIdx.x * a.pstride;
#pragma unroll
  for (int h = 0; h < 2; ++h)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int t = 4 * g + r;
      if (t < ntok) {
        part[t * HID + head * DH + c + 16 * h] = dEK[h][r];
        part[ntok * HID + t * HID + head * DH + c + 16 * h] = dEV[h][r];
      }
    }
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = 4 * g + r;
    if (i < T && cT) part[2 * ntok * HID + (head * T + i) * T + c] = bacc[r];
  }
}

// dek / dev [B][ntok][HID] += sum over the sample's workgroups, dbias [HEADS][T][T] += sum over all workgroups.
// Workgroup = 16 consecutive elements x 16 slices of the partials; fixed summation order.
__global__ __launch_bounds__(256) void temporal_attn_bwd_reduce_kernel(const float* __restrict__ part, int pstride, int bps, int B, int ntok, int T,
                                                                      float* __restrict__ dek, float* __restrict__ dev, float* __restrict__ dbias) {
  __shared__ float red[16][17];
  const int e16 = threadIdx.x & 15, kg = threadIdx.x >> 4;
  const int idx = blockIdx.x * 16 + e16;
  const int ntk = ntok * HID, ntot = B * 2 * ntk;
  const float* src = nullptr;
  float* dst = nullptr;
  int n = 0;
  if (idx < ntot) {
    const int b = idx / (2 * ntk), e = idx - b * 2 * ntk;
    src = part + (long long)b * bps * pstride + e;
    n = bps;
    dst = e < ntk ? dek + (long long)b * ntk + e : dev + (long long)b * ntk + (e - ntk);
  } else if (dbias && idx - ntot < HEADS * T * T) {
    src = part + 2 * ntk + (idx - ntot);
    n = B * bps;
    dst = dbias + (idx - ntot);
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int k = kg;
  for (; k + 48 < n; k += 64) {
    s0 += src[(long long)k * pstride];
    s1 += src[(long long)(k + 16) * pstride];
    s2 += src[(long long)(k + 32) * pstride];
    s3 += src[(long long)(k + 48) * pstride];
  }
  for (; k < n; k += 16) s0 += src[(long long)k * pstride];
  red[kg][e16] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (kg == 0 && dst) {
    float t = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) t += red[q][e16];
    *dst += t;
  }
}

int tb_blocks_per_sample(int B, int HW) { return (int)max(1LL, min((long long)HW, cdiv(1024, B))); }
int tb_pstride(int ntok, int T) { return 2 * ntok * HID + HEADS * T * T; }

}  // namespace

// floats of scratch vmm_attention_bwd needs in dbuf
extern "C" int64_t vmm_attention_bwd_scratch(int32_t mode, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t ntok) {
  const int64_t base = (int64_t)B * T * HW * heads;
  if (mode != 0 || heads != HEADS || T > 16 || ntok > 16) return base;
  const int64_t fast = (int64_t)B * tb_blocks_per_sample(B, HW) * tb_pstride(ntok, T);
  return fast > base ? fast : base;
}

// Fast path of vmm_attention_bwd for mode 0 (same arguments and results; scratch = dbuf of vmm_attention_bwd_scratch floats).  Returns 1
// (nothing launched) outside its envelope: heads = 8, dim_head = 32, T <= 16, ntok <= 16 shared tokens, ntok <= T when the bias
// also covers the tokens.  dek / dev / dbias are accumulated (+=) without atomics: bit-reproducible.
extern "C" int vmm_temporal_attention_bwd(const float* qkv, int32_t ldqkv, const float* ek, const float* ev, int32_t ntok, const float* bias,
                                          int32_t bias_on_cond, const float* out, const float* dout, int32_t ldo, const float* lse,
                                          const float* rot_tab, float q_scale, float* dqkv, float* dek, float* dev, float* dbias,
                                          float* scratch, int32_t B, int32_t T, int32_t HW, int32_t heads, int32_t dh, vmm_stream_t stream) {
  if (!ek) ntok = 0;
  if (heads != HEADS || dh != DH || T > 16 || T < 1 || ntok > 16 || (ldqkv & 3) || (ldo & 3) || !scratch) return 1;
  if (bias && bias_on_cond && ntok > T) return 1;
  if (ntok > 0 && (!dek || !dev)) return -1;
  if (B <= 0 || HW <= 0) return 0;
  const int bps = tb_blocks_per_sample(B, HW), pstride = tb_pstride(ntok, T);
  TMArgs a{qkv, ek, ev, bias, out, dout, lse, rot_tab, dqkv, scratch, ldqkv, ldo, B, T, HW, ntok, bias_on_cond, bps, pstride, q_scale};
  hipStream_t s = (hipStream_t)stream;
  constexpr size_t shm = sizeof(float) * (HEADS * 2 * 16 * 17 + HEADS * 3 * 16 * 36);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_attn_bwd_mfma_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_attn_bwd_mfma_kernel, dim3((unsigned)(B * bps)), dim3(512), shm, s, a);
  VMM_LAUNCH_CHECK();
  const int nred = B * 2 * ntok * HID + (bias && dbias ? HEADS * T * T : 0);
  if (nred > 0) {
    hipLaunchKernelGGL(temporal_attn_bwd_reduce_kernel, dim3((unsigned)cdiv(nred, 16)), dim3(256), 0, s, scratch, pstride, bps, B, ntok, T, dek, dev,
                       bias ? dbias : nullptr);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
