// Backward of the small host-of-the-network pieces (training path): tiny dense layers, conditioning tokens,
// embeddings, the NCTHW output projection, the loss.  Sizes are a few thousand elements; everything is
// atomics-into-zeroed-buffers for simplicity, one launch per dependency level.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

__device__ __forceinline__ float act_f(float v, int act) { return act == 1 ? silu_f(v) : (act == 2 ? gelu_erf_f(v) : v); }
__device__ __forceinline__ float act_grad(float v, int act) {
  if (act == 1) { const float s = 1.0f / (1.0f + expf(-v)); return s * (1.0f + v * (1.0f - s)); }
  if (act == 2) return 0.5f * (1.0f + erff(v * 0.70710678118654752440f)) + v * 0.3989422804014327f * expf(-0.5f * v * v);
  return 1.0f;
}

// Stage 1, one wave per (job, output column o): recompute z = act_in(x) W^T + b, g = dy * act_out'(z) (written back over dy),
// dW[o,:] (=|+=) sum_r g_r act_in(x_r,:), db[o] (=|+=) sum_r g_r.      K <= 1024, rows processed in chunks of 64.
// Jobs without an output activation (the FiLM projections and the token key / value layers: the 54 jobs of the last embedding level, whose launch
// closes the backward) take dense_bwd_w_tile_kernel below; the wave-per-column kernel keeps the rest.
__device__ __forceinline__ bool dense_bwd_tiled(const vmm_dense_bwd_job& jb) { return jb.act_out == 0 && jb.K <= 256 && (jb.K & 3) == 0 && jb.rows > 0; }

// Workgroup = 64 output columns of one job: act_in(x) and dy of 32 rows at a time in LDS, thread (column o = tid >> 2, quarter kq = tid & 3) owns
// dW[o][kq + 4 i]: one LDS read of g and K / 4 of x per row.  (The wave-per-column kernel walked the rows through dependent global loads, one in
// flight per wave: 0.2 ms for a level of 44-row jobs.)  g = dy (no output activation), so nothing is written back over dy.
template <int NK>  // K / 4 <= NK: accumulators per thread (compile-time bound, so that a 64-column job does not walk 64 predicated slots)
__device__ __forceinline__ void dense_bwd_w_tile(const vmm_dense_bwd_job& jb, int o0, float* xs, float* gs) {
  constexpr int RC = 32, XP = 260, GP = 65;
  const int tid = threadIdx.x, o = tid >> 2, kq = tid & 3, nk = jb.K >> 2;
  float acc[NK];
#pragma unroll
  for (int i = 0; i < NK; ++i) acc[i] = 0.f;
  float db = 0.f;
  for (int r0 = 0; r0 < jb.rows; r0 += RC) {
    const int nr = min(RC, jb.rows - r0);
    __syncthreads();
    for (int e = tid; e < nr * nk; e += 256) {
      const int r = e / nk, k4 = (e - r * nk) * 4;
      const f32x4 v = *reinterpret_cast<const f32x4*>(jb.x + (long long)(r0 + r) * jb.ldx + k4);
      float* d = xs + r * XP + k4;
      d[0] = act_f(v.x, jb.act_in); d[1] = act_f(v.y, jb.act_in); d[2] = act_f(v.z, jb.act_in); d[3] = act_f(v.w, jb.act_in);
    }
    for (int e = tid; e < nr * 64; e += 256) {
      const int r = e >> 6, c = e & 63;
      gs[r * GP + c] = o0 + c < jb.N ? jb.dy[(long long)(r0 + r) * jb.lddy + o0 + c] : 0.f;
    }
    __syncthreads();
    for (int r = 0; r < nr; ++r) {
      const float g = gs[r * GP + o];
      db += g;
      const float* xr = xs + r * XP + kq;
#pragma unroll
      for (int i = 0; i < NK; ++i)
        if (i < nk) acc[i] = fmaf(g, xr[4 * i], acc[i]);
    }
  }
  if (o0 + o >= jb.N) return;
  if (jb.dw) {
    float* p = jb.dw + (long long)(o0 + o) * jb.K + kq;
#pragma unroll
    for (int i = 0; i < NK; ++i)
      if (i < nk) p[4 * i] = jb.accumulate ? p[4 * i] + acc[i] : acc[i];
  }
  if (kq == 0 && jb.db) jb.db[o0 + o] = jb.accumulate ? jb.db[o0 + o] + db : db;
}

__global__ __launch_bounds__(256) void dense_bwd_w_tile_kernel(const vmm_dense_bwd_job* __restrict__ jobs) {
  const vmm_dense_bwd_job jb = jobs[blockIdx.y];
  const int o0 = blockIdx.x * 64;
  if (!dense_bwd_tiled(jb) || o0 >= jb.N) return;
  __shared__ float xs[32 * 260], gs[32 * 65];
  if (jb.K <= 64) dense_bwd_w_tile<16>(jb, o0, xs, gs);
  else if (jb.K <= 128) dense_bwd_w_tile<32>(jb, o0, xs, gs);
  else dense_bwd_w_tile<64>(jb, o0, xs, gs);
}

__global__ __launch_bounds__(256) void dense_bwd_w_kernel(const vmm_dense_bwd_job* __restrict__ jobs) {
  __shared__ float gsh[4][64];
  const vmm_dense_bwd_job jb = jobs[blockIdx.y];
  if (dense_bwd_tiled(jb)) return;
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int o = blockIdx.x * 4 + wv;
  if (o >= jb.N) return;
  const float* wrow = jb.w + (long long)o * jb.K;
  const float bv = jb.b ? jb.b[o] : 0.f;
  float db = 0.f;
  float dwv[16];
#pragma unroll
  for (int t = 0; t < 16; ++t) dwv[t] = 0.f;
  for (int r0 = 0; r0 < jb.rows; r0 += 64) {
    const int nr = min(64, jb.rows - r0);
    // dy of the block's rows in ONE load (lane = row) and g back in one store after the loop: with a dy load and a dy store inside the row loop
    // every row paid two dependent memory round trips and the stores kept the compiler from overlapping the rows' x loads (44 rows: 0.24 ms)
    const float dyv = lane < nr ? jb.dy[(long long)(r0 + lane) * jb.lddy + o] : 0.f;
    float gmine = 0.f;
#pragma unroll 4
    for (int r = 0; r < nr; ++r) {
      float part = 0.f;
      for (int k = lane; k < jb.K; k += 64) part = fmaf(act_f(jb.x[(long long)(r0 + r) * jb.ldx + k], jb.act_in), wrow[k], part);
      const float z = wave_sum(part) + bv;
      const float g = __shfl(dyv, r, 64) * act_grad(z, jb.act_out);
      db += g;
      if (lane == r) gmine = g;
      if (lane == 0) gsh[wv][r] = g;
    }
    if (lane < nr) jb.dy[(long long)(r0 + lane) * jb.lddy + o] = gmine;
    __builtin_amdgcn_wave_barrier();
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      const int k = lane + 64 * t;
      if (k < jb.K)
        for (int r = 0; r < nr; ++r) dwv[t] = fmaf(gsh[wv][r], act_f(jb.x[(long long)(r0 + r) * jb.ldx + k], jb.act_in), dwv[t]);
    }
    __builtin_amdgcn_wave_barrier();
  }
#pragma unroll
  for (int t = 0; t < 16; ++t) {
    const int k = lane + 64 * t;
    if (k < jb.K && jb.dw) {
      float* p = jb.dw + (long long)o * jb.K + k;
      *p = jb.accumulate ? *p + dwv[t] : dwv[t];
    }
  }
  if (lane == 0 && jb.db) jb.db[o] = jb.accumulate ? jb.db[o] + db : db;
}

// Stage 2, one wave per (job, row r, chunk of 64 input columns): dx[r,k] += act_in'(x[r,k]) * sum_o g[r,o] W[o,k]  (atomic: several
// jobs of one launch may share x, e.g. the conditioning tokens feed every to_k / to_v)
__global__ __launch_bounds__(256) void dense_bwd_x_kernel(const vmm_dense_bwd_job* __restrict__ jobs) {
  const vmm_dense_bwd_job jb = jobs[blockIdx.y];
  if (!jb.dx) return;
  const int lane = threadIdx.x & 63;
  const int unit = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int kchunks = (jb.K + 63) / 64;
  if (unit >= jb.rows * kchunks) return;
  const int r = unit / kchunks, k = (unit % kchunks) * 64 + lane;
  if (k >= jb.K) return;
  // blockIdx.z = slice of 128 output features: these layers have a handful of rows (the batch), so the reduction over N is what there
  // is to parallelise; dx is accumulated with atomics anyway
  const int o0 = blockIdx.z * 128, o1 = min(jb.N, o0 + 128);
  if (o0 >= o1) return;
  // (eight independent products per trip: with two, the 128-feature slice was 64 dependent L2 round trips -- 50 of the launch's 67 us)
  float a8[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  const float* dyr = jb.dy + (long long)r * jb.lddy;
  int o = o0;
  for (; o + 7 < o1; o += 8) {
    float d8[8], w8[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { d8[u] = dyr[o + u]; w8[u] = jb.w[(long long)(o + u) * jb.K + k]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) a8[u] = fmaf(d8[u], w8[u], a8[u]);
  }
  for (; o < o1; ++o) a8[0] = fmaf(dyr[o], jb.w[(long long)o * jb.K + k], a8[0]);
  const float acc = ((a8[0] + a8[1]) + (a8[2] + a8[3])) + ((a8[4] + a8[5]) + (a8[6] + a8[7]));
  atomicAdd(&jb.dx[(long long)r * jb.lddx + k], acc * act_grad(jb.x[(long long)r * jb.ldx + k], jb.act_in));
}

// tokens[b,f,d] = cond[b,f]*w[d] + bias[d] (or null token), pooled = mean_f of the un-dropped tokens; thread per d
__global__ void cond_tokens_bwd_kernel(const float* __restrict__ cond, const uint8_t* __restrict__ mask, const float* __restrict__ dtokens,
                                       const float* __restrict__ dpooled, int B, int F, int D, float* __restrict__ dw, float* __restrict__ dbias,
                                       float* __restrict__ dnull) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float gw = 0.f, gb = 0.f;
  for (int b = 0; b < B; ++b) {
    const bool drop = mask && mask[b];
    const float gp = dpooled ? dpooled[b * D + d] / (float)F : 0.f;
    for (int f = 0; f < F; ++f) {
      const float gt = dtokens[((long long)b * F + f) * D + d];
      const float g = (drop ? 0.f : gt) + gp;
      gw = fmaf(g, cond[b * F + f], gw);
      gb += g;
      if (drop) dnull[f * D + d] += gt;  // exclusive owner of column d: no atomics needed
    }
  }
  dw[d] += gw;
  dbias[d] += gb;
}

// y = LN(x)*w + b over rows; wave per row
__global__ __launch_bounds__(64) void rows_ln_affine_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dw,
                                                                float* __restrict__ db, int D, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (long long)row * D;
  const float* gr = dy + (long long)row * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += xr[k];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int k = lane; k < D; k += 64) { const float d = xr[k] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  float a1 = 0.f, a2 = 0.f;
  for (int k = lane; k < D; k += 64) {
    const float xh = (xr[k] - mean) * rstd, g = gr[k] * w[k];
    a1 += g;
    a2 += g * xh;
    atomicAdd(&dw[k], gr[k] * xh);
    atomicAdd(&db[k], gr[k]);
  }
  a1 = wave_sum(a1) / (float)D;
  a2 = wave_sum(a2) / (float)D;
  for (int k = lane; k < D; k += 64) {
    const float xh = (xr[k] - mean) * rstd;
    atomicAdd(&dx[(long long)row * D + k], rstd * (gr[k] * w[k] - a1 - xh * a2));
  }
}

// out[b,:] = (mask ? null : x[b,:]) + add[b,:]
__global__ void select_add_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                      float* __restrict__ dnull, float* __restrict__ dadd, int B, int D) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float gn = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dout[b * D + d];
    if (mask && mask[b]) gn += g; else if (dx) dx[b * D + d] += g;
    if (dadd) dadd[b * D + d] += g;
  }
  if (dnull) dnull[d] += gn;
}

// out[b, :D] = t[b, :], out[b, D:] = mask ? null : x[b, :]   (cond_to_time = 'concat')
__global__ void select_concat_bwd_kernel(const float* __restrict__ dout, const uint8_t* __restrict__ mask, float* __restrict__ dx,
                                         float* __restrict__ dnull, float* __restrict__ dt, int B, int D) {
  const int d = blockIdx.x * blockDim.x + threadIdx.x;
  if (d >= D) return;
  float gn = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dout[(long long)b * 2 * D + D + d];
    if (mask && mask[b]) gn += g; else if (dx) dx[b * D + d] += g;
    if (dt) dt[b * D + d] += dout[(long long)b * 2 * D + d];
  }
  if (dnull) dnull[d] += gn;
}

// Backward of gru_rec_fwd_kernel through time, one workgroup per sample: with dh = dy_t + carry,
//   dn = dh (1 - z), dz = dh (h_prev - n), carry' = dh z + W_hh^T dgh,   dn_pre = dn (1 - n^2), dz_pre = dz z (1 - z), dr_pre = dn_pre gh_n r (1 - r),
//   dgi = (dr_pre, dz_pre, dn_pre), dgh = (dr_pre, dz_pre, dn_pre r).
// dgi / dgh of every (sample, step) are left for two batched dense backward jobs (weight / bias gradients, and dgi W_ih = the layer below's dy).
__global__ __launch_bounds__(256) void gru_rec_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ gates, const float* __restrict__ hprev,
                                                          const float* __restrict__ whh, float* __restrict__ dgi, float* __restrict__ dgh, int L, int H) {
  extern __shared__ __attribute__((aligned(16))) float gru_sm[];  // dgh of the step [3H] | carry [H]
  float* dghs = gru_sm;
  float* carry = gru_sm + 3 * H;
  const int tid = threadIdx.x;
  const long long b = blockIdx.x;
  for (int j = tid; j < H; j += 256) carry[j] = 0.f;
  __syncthreads();
  for (int t = L - 1; t >= 0; --t) {
    const long long bt = b * L + t;
    float direct[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int j = tid + 256 * i;
      direct[i] = 0.f;
      if (j < H) {
        const float* G = gates + bt * 4 * H;
        const float r = G[j], z = G[H + j], n = G[2 * H + j], an = G[3 * H + j], hp = hprev[bt * H + j];
        const float dh = dy[bt * H + j] + carry[j];
        const float dnp = dh * (1.0f - z) * (1.0f - n * n);
        const float dzp = dh * (hp - n) * z * (1.0f - z);
        const float drp = dnp * an * r * (1.0f - r);
        float* o1 = dgi + bt * 3 * H;
        float* o2 = dgh + bt * 3 * H;
        o1[j] = drp; o1[H + j] = dzp; o1[2 * H + j] = dnp;
        o2[j] = drp; o2[H + j] = dzp; o2[2 * H + j] = dnp * r;
        dghs[j] = drp; dghs[H + j] = dzp; dghs[2 * H + j] = dnp * r;
        direct[i] = dh * z;
      }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int k = tid + 256 * i;
      if (k < H) {
        float acc = direct[i];
        for (int g = 0; g < 3 * H; ++g) acc = fmaf(dghs[g], whh[(long long)g * H + k], acc);
        carry[k] = acc;  // (only its owner read carry[k] above)
      }
    }
    __syncthreads();
  }
}

// tokens[b, n, :] = mask[b] ? null[n, :] : g[b, n, :]: dg is WRITTEN, dnull accumulated; thread per (n, d)
__global__ void tokens_select_bwd_kernel(const float* __restrict__ dtokens, const uint8_t* __restrict__ mask, int B, int N, int D, float* __restrict__ dg,
                                         float* __restrict__ dnull) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  float gn = 0.f;
  for (int b = 0; b < B; ++b) {
    const float g = dtokens[(long long)b * N * D + i];
    const bool drop = mask && mask[b];
    if (drop) gn += g;
    dg[(long long)b * N * D + i] = drop ? 0.f : g;
  }
  if (dnull) dnull[i] += gn;
}

__global__ void relpos_bias_bwd_kernel(const float* __restrict__ dbias, const int32_t* __restrict__ buckets, int n, int heads,
                                       float* __restrict__ demb) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * n * n) return;
  const int h = i / (n * n), ij = i - h * n * n;
  atomicAdd(&demb[buckets[ij] * heads + h], dbias[i]);
}

// tokens[b,n,:] = mask ? null[n,:] : hidden[b,:]
__global__ void tokens_from_hidden_bwd_kernel(const float* __restrict__ dtokens, const uint8_t* __restrict__ mask, int B, int N, int D,
                                              float* __restrict__ dhidden, float* __restrict__ dnull) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= N * D) return;
  const int d = i % D, n = i / D;
  for (int b = 0; b < B; ++b) {
    const float g = dtokens[((long long)b * N + n) * D + d];
    if (mask && mask[b]) dnull[i] += g; else atomicAdd(&dhidden[b * D + d], g);
  }
}

// y = silu(conv1d_k4s2(x)); thread per output element, atomics into dx/dw/db
__global__ void conv1d_k4s2_silu_bwd_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                            const float* __restrict__ dy, float* __restrict__ dx, float* __restrict__ dw, float* __restrict__ db,
                                            int B, int Cin, int Cout, int Lin, int Lout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Cout * Lout) return;
  const int l = i % Lout, co = (i / Lout) % Cout, b = i / (Lout * Cout);
  float z = bias[co];
  for (int ci = 0; ci < Cin; ++ci)
    for (int k = 0; k < 4; ++k) {
      const int p = l * 2 - 1 + k;
      if (p >= 0 && p < Lin) z = fmaf(x[((long long)b * Cin + ci) * Lin + p], w[((long long)co * Cin + ci) * 4 + k], z);
    }
  const float g = dy[i] * act_grad(z, 1);
  atomicAdd(&db[co], g);
  for (int ci = 0; ci < Cin; ++ci)
    for (int k = 0; k < 4; ++k) {
      const int p = l * 2 - 1 + k;
      if (p >= 0 && p < Lin) {
        atomicAdd(&dw[((long long)co * Cin + ci) * 4 + k], g * x[((long long)b * Cin + ci) * Lin + p]);
        if (dx) atomicAdd(&dx[((long long)b * Cin + ci) * Lin + p], g * w[((long long)co * Cin + ci) * 4 + k]);
      }
    }
}

// out[b,co,t,hw] = bias[co] + sum_ci rows[r,ci] w[co,ci]: drows (=), dw/db (atomics, block-reduced)
__global__ __launch_bounds__(256) void pointwise_to_ncthw_bwd_kernel(const float* __restrict__ rows, int ld, int Cin, const float* __restrict__ w,
                                                                     const float* __restrict__ dout, int Cout, int T, int HW,
                                                                     float* __restrict__ drows, int lddr, float* __restrict__ dw,
                                                                     float* __restrict__ db, long long nrows) {
  extern __shared__ float sh[];  // w[Cout*Cin] | dwacc[Cout*Cin] | dbacc[Cout]
  float* ws = sh;
  float* dwa = sh + Cout * Cin;
  float* dba = dwa + Cout * Cin;
  for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) { ws[i] = w[i]; dwa[i] = 0.f; }
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) dba[i] = 0.f;
  __syncthreads();
  for (long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x; r < nrows; r += (long long)gridDim.x * blockDim.x) {
    const int hw = (int)(r % HW);
    const int t = (int)((r / HW) % T);
    const long long b = r / ((long long)HW * T);
    for (int co = 0; co < Cout; ++co) atomicAdd(&dba[co], dout[((b * Cout + co) * T + t) * HW + hw]);
    for (int ci = 0; ci < Cin; ++ci) {
      const float xv = rows[r * ld + ci];
      float acc = 0.f;
      for (int co = 0; co < Cout; ++co) {
        const float g = dout[((b * Cout + co) * T + t) * HW + hw];
        acc = fmaf(g, ws[co * Cin + ci], acc);
        atomicAdd(&dwa[co * Cin + ci], g * xv);
      }
      drows[r * lddr + ci] = acc;
    }
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) atomicAdd(&dw[i], dwa[i]);
  for (int i = threadIdx.x; i < Cout; i += blockDim.x) atomicAdd(&db[i], dba[i]);
}

// The same for Cout <= 4 and Cin / 4 a power of two <= 64 (the final 1x1 conv: 64 -> 3 channels): Cin / 4 lanes per row, each owning four
// input channels -- 16-byte row loads / stores, the dw partial sums in registers; one LDS reduction and one set of atomics per workgroup.
template <int CO>
__global__ __launch_bounds__(256) void pointwise_to_ncthw_bwd_small_kernel(const float* __restrict__ rows, int ld, int Cin, const float* __restrict__ w,
                                                                           const float* __restrict__ dout, int T, int HW,
                                                                           float* __restrict__ drows, int lddr, float* __restrict__ dw,
                                                                           float* __restrict__ db, long long nrows) {
  __shared__ float red[256 * (4 * CO + 1)];
  const int tpr = Cin >> 2, c4 = threadIdx.x % tpr, slot = threadIdx.x / tpr, nslots = 256 / tpr;
  f32x4 wv[CO], dwa[CO];
  float dba[CO];
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    wv[co] = *reinterpret_cast<const f32x4*>(w + co * Cin + c4 * 4);
    dwa[co] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dba[co] = 0.f;
  }
  for (long long r = (long long)blockIdx.x * nslots + slot; r < nrows; r += (long long)gridDim.x * nslots) {
    const int hw = (int)(r % HW);
    const int t = (int)((r / HW) % T);
    const long long b = r / ((long long)HW * T);
    const f32x4 x = *reinterpret_cast<const f32x4*>(rows + r * ld + c4 * 4);
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int co = 0; co < CO; ++co) {
      const float g = dout[((b * CO + co) * T + t) * HW + hw];
      acc.x = fmaf(g, wv[co].x, acc.x); acc.y = fmaf(g, wv[co].y, acc.y); acc.z = fmaf(g, wv[co].z, acc.z); acc.w = fmaf(g, wv[co].w, acc.w);
      dwa[co].x = fmaf(g, x.x, dwa[co].x); dwa[co].y = fmaf(g, x.y, dwa[co].y); dwa[co].z = fmaf(g, x.z, dwa[co].z); dwa[co].w = fmaf(g, x.w, dwa[co].w);
      dba[co] += g;
    }
    *reinterpret_cast<f32x4*>(drows + r * lddr + c4 * 4) = acc;
  }
  float* mine = red + threadIdx.x * (4 * CO + 1);
#pragma unroll
  for (int co = 0; co < CO; ++co) { mine[4 * co] = dwa[co].x; mine[4 * co + 1] = dwa[co].y; mine[4 * co + 2] = dwa[co].z; mine[4 * co + 3] = dwa[co].w; }
  mine[4 * CO] = dba[0];
  __syncthreads();
  // thread e < CO * Cin: dw[co][ci] over the row slots
  for (int e = threadIdx.x; e < CO * Cin; e += 256) {
    const int co = e / Cin, ci = e - co * Cin;
    float sum = 0.f;
    for (int k = 0; k < nslots; ++k) sum += red[(k * tpr + (ci >> 2)) * (4 * CO + 1) + 4 * co + (ci & 3)];
    atomicAdd(&dw[e], sum);
  }
  __syncthreads();
  // db: every lane of a row saw the same g, take lane c4 == 0 of each slot
#pragma unroll
  for (int co = 0; co < CO; ++co) {
    if (c4 == 0) red[slot * CO + co] = dba[co];
  }
  __syncthreads();
  if (threadIdx.x < CO) {
    float sum = 0.f;
    for (int k = 0; k < nslots; ++k) sum += red[k * CO + threadIdx.x];
    atomicAdd(&db[threadIdx.x], sum);
  }
}

// d loss / d pred for mean |noise - pred| or mean (noise - pred)^2, times the upstream scalar gradient
__global__ void loss_grad_kernel(const float* __restrict__ noise, const float* __restrict__ pred, long long n, int squared,
                                 const float* __restrict__ gscale, float* __restrict__ dpred) {
  const float gs = (gscale ? gscale[0] : 1.0f) / (float)n;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
    const float d = pred[i] - noise[i];
    dpred[i] = squared ? 2.0f * d * gs : (d > 0.f ? gs : (d < 0.f ? -gs : 0.f));
  }
}

}  // namespace

extern "C" int vmm_dense_bwd_batched(const vmm_dense_bwd_job* jobs_dev, int32_t njobs, int32_t max_N, int32_t max_xunits,
                                     vmm_stream_t stream) {
  if (njobs <= 0) return 0;
  hipLaunchKernelGGL(dense_bwd_w_kernel, dim3(cdiv(max_N, 4), njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  VMM_LAUNCH_CHECK();
  hipLaunchKernelGGL(dense_bwd_w_tile_kernel, dim3(cdiv(max_N, 64), njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  VMM_LAUNCH_CHECK();
  if (max_xunits > 0) {
    hipLaunchKernelGGL(dense_bwd_x_kernel, dim3(cdiv(max_xunits, 4), njobs, cdiv(max_N, 128)), dim3(256), 0, (hipStream_t)stream, jobs_dev);
    VMM_LAUNCH_CHECK();
  }
  return 0;
}
extern "C" int vmm_cond_tokens_bwd(const float* cond, const uint8_t* mask, const float* dtokens, const float* dpooled, int32_t B, int32_t F,
                                   int32_t D, float* dw, float* dbias, float* dnull_token, vmm_stream_t stream) {
  hipLaunchKernelGGL(cond_tokens_bwd_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, cond, mask, dtokens, dpooled, B, F, D, dw, dbias,
                     dnull_token);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_rows_layernorm_affine_bwd(const float* x, const float* w, const float* dy, float* dx, float* dw, float* db, int32_t rows,
                                             int32_t D, float eps, vmm_stream_t stream) {
  hipLaunchKernelGGL(rows_ln_affine_bwd_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, w, dy, dx, dw, db, D, eps);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_select_add_bwd(const float* dout, const uint8_t* mask, float* dx, float* dnull_row, float* dadd, int32_t B, int32_t D,
                                  vmm_stream_t stream) {
  hipLaunchKernelGGL(select_add_bwd_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, dout, mask, dx, dnull_row, dadd, B, D);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_select_concat_bwd(const float* dout, const uint8_t* mask, float* dx, float* dnull_row, float* dt, int32_t B, int32_t D,
                                     vmm_stream_t stream) {
  hipLaunchKernelGGL(select_concat_bwd_kernel, dim3(cdiv(D, 64)), dim3(64), 0, (hipStream_t)stream, dout, mask, dx, dnull_row, dt, B, D);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_gru_recurrent_bwd(const float* dy, const float* gates, const float* hprev, const float* whh, float* dgi, float* dgh, int32_t B, int32_t L,
                                     int32_t H, vmm_stream_t stream) {
  if (!dy || !gates || !hprev || !whh || !dgi || !dgh || H < 1 || H > 4096 || L < 1) return -1;
  if (B <= 0) return 0;
  hipLaunchKernelGGL(gru_rec_bwd_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * 4 * H, (hipStream_t)stream, dy, gates, hprev, whh, dgi, dgh, L, H);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_tokens_select_bwd(const float* dtokens, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* dg, float* dnull_token,
                                     vmm_stream_t stream) {
  if (!dtokens || !dg) return -1;
  hipLaunchKernelGGL(tokens_select_bwd_kernel, dim3(cdiv(N * D, 256)), dim3(256), 0, (hipStream_t)stream, dtokens, mask, B, N, D, dg, dnull_token);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_relpos_bias_bwd(const float* dbias, const int32_t* buckets, int32_t n, int32_t heads, float* demb, vmm_stream_t stream) {
  hipLaunchKernelGGL(relpos_bias_bwd_kernel, dim3(cdiv(heads * n * n, 256)), dim3(256), 0, (hipStream_t)stream, dbias, buckets, n, heads, demb);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_tokens_from_hidden_bwd(const float* dtokens, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* dhidden,
                                          float* dnull_token, vmm_stream_t stream) {
  hipLaunchKernelGGL(tokens_from_hidden_bwd_kernel, dim3(cdiv(N * D, 256)), dim3(256), 0, (hipStream_t)stream, dtokens, mask, B, N, D, dhidden,
                     dnull_token);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_conv1d_k4s2_silu_bwd(const float* x, const float* w, const float* bias, const float* dy, float* dx, float* dw, float* db,
                                        int32_t B, int32_t Cin, int32_t Cout, int32_t Lin, vmm_stream_t stream) {
  const int Lout = (Lin + 2 - 4) / 2 + 1;
  hipLaunchKernelGGL(conv1d_k4s2_silu_bwd_kernel, dim3(cdiv(B * Cout * Lout, 128)), dim3(128), 0, (hipStream_t)stream, x, w, bias, dy, dx, dw, db, B,
                     Cin, Cout, Lin, Lout);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_pointwise_to_ncthw_bwd(const float* rows, int32_t ld, int32_t Cin, const float* w, const float* dout, int32_t B, int32_t Cout,
                                          int32_t T, int32_t HW, float* drows, int32_t lddr, float* dw, float* db, vmm_stream_t stream) {
  if (Cout * Cin > 4096) return -1;
  const long long nrows = (long long)B * T * HW;
  const int tpr = Cin / 4;
  if (Cout <= 4 && (Cin & 3) == 0 && tpr >= 1 && tpr <= 64 && (tpr & (tpr - 1)) == 0 && (ld & 3) == 0 && (lddr & 3) == 0) {
    const int nslots = 256 / tpr;
    const int nb = (int)max(1LL, min((long long)cdiv(nrows, nslots), 2048LL));
    hipStream_t s = (hipStream_t)stream;
#define VMM_PW_SMALL(CO) hipLaunchKernelGGL(pointwise_to_ncthw_bwd_small_kernel<CO>, dim3(nb), dim3(256), 0, s, rows, ld, Cin, w, dout, T, HW, drows, lddr, dw, db, nrows)
    if (Cout == 1) VMM_PW_SMALL(1); else if (Cout == 2) VMM_PW_SMALL(2); else if (Cout == 3) VMM_PW_SMALL(3); else VMM_PW_SMALL(4);
#undef VMM_PW_SMALL
    VMM_LAUNCH_CHECK();
    return 0;
  }
  const int blocks = (int)max(1LL, min((long long)cdiv(nrows, 256), 512LL));
  const size_t shm = sizeof(float) * (2 * Cout * Cin + Cout);
  hipLaunchKernelGGL(pointwise_to_ncthw_bwd_kernel, dim3(blocks), dim3(256), shm, (hipStream_t)stream, rows, ld, Cin, w, dout, Cout, T, HW, drows,
                     lddr, dw, db, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}
// Data gradient of the stem (init_conv: Conv3d(Cx <= 4, Cout, (1, k, k)) pad k / 2, vddp.py:600) straight into the NCTHW layout of the network
// input: dx[b, c, t, y, x] = sum_{kh, kw, co} g[(b, t, y + pad - kh, x + pad - kw)][co] w[co][c][kh][kw] (zero padding).  Only autograd users that ask
// for the input gradient run it (the training step never does): a plain vector-unit kernel -- a thread per (pixel, c) with the k x k x Cx x Cout weights
// in LDS as [tap][c][co] -- for 3.8 GFLOP at the Lagrangian sizes.
__global__ __launch_bounds__(256) void stem_dgrad_kernel(const float* __restrict__ g, int ldg, const float* __restrict__ w, float* __restrict__ dx, int Cx, int T,
                                                         int H, int W, int Cout, int k, long long nrows, int wrap_h, int wrap_w) {
  extern __shared__ float ws[];  // [k*k][Cx][Cout]
  const int kk = k * k;
  for (int i = threadIdx.x; i < kk * Cx * Cout; i += blockDim.x) {
    const int co = i % Cout, c = (i / Cout) % Cx, tap = i / (Cout * Cx);
    ws[i] = w[((long long)co * Cx + c) * kk + tap];
  }
  __syncthreads();
  const int pad = k / 2, HW = H * W;
  const long long item = (long long)blockIdx.x * blockDim.x + threadIdx.x;  // (row, c), c fastest
  if (item >= nrows * Cx) return;
  const int c = (int)(item % Cx);
  const long long r = item / Cx;
  const int px = (int)(r % W), py = (int)((r / W) % H);
  const long long img = r / HW;
  float acc = 0.f;
  for (int kh = 0; kh < k; ++kh) {
    int yy = py + pad - kh;
    if (wrap_h) yy = ((yy % H) + H) % H;  // periodic padding (vddp.py:163-243): the output row the tap reads from lies across the seam
    if (yy < 0 || yy >= H) continue;
    for (int kw = 0; kw < k; ++kw) {
      int xx = px + pad - kw;
      if (wrap_w) xx = ((xx % W) + W) % W;
      if (xx < 0 || xx >= W) continue;
      const float* gp = g + ((img * H + yy) * W + xx) * ldg;
      const float* wp = ws + ((kh * k + kw) * Cx + c) * Cout;
      for (int co = 0; co < Cout; co += 4) {
        const f32x4 gv = *reinterpret_cast<const f32x4*>(gp + co);
        acc = fmaf(gv.x, wp[co], acc);
        acc = fmaf(gv.y, wp[co + 1], acc);
        acc = fmaf(gv.z, wp[co + 2], acc);
        acc = fmaf(gv.w, wp[co + 3], acc);
      }
    }
  }
  const long long b = img / T;
  const int t = (int)(img % T);
  dx[((b * Cx + c) * T + t) * HW + (long long)py * W + px] = acc;
}

extern "C" int vmm_stem_conv_dgrad(const float* g, int32_t ldg, const float* w, float* dx, int32_t B, int32_t Cx, int32_t T, int32_t H, int32_t W, int32_t Cout,
                                   int32_t k, int32_t wrap_h, int32_t wrap_w, vmm_stream_t stream) {
  if (Cx < 1 || Cout % 4 || (ldg & 3) || k < 1 || !(k & 1) || (size_t)k * k * Cx * Cout * sizeof(float) > 160 * 1024) return -1;
  const long long nrows = (long long)B * T * H * W;
  if (nrows <= 0) return 0;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(stem_dgrad_kernel, dim3((unsigned)cdiv(nrows * Cx, 256)), dim3(256), sizeof(float) * k * k * Cx * Cout, (hipStream_t)stream, g, ldg, w, dx, Cx, T, H, W,
                     Cout, k, nrows, wrap_h, wrap_w);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_loss_grad(const float* noise, const float* pred, int64_t n, int32_t squared, const float* gscale, float* dpred,
                             vmm_stream_t stream) {
  const int blocks = (int)max(1LL, min((long long)cdiv(n, 256), 4096LL));
  hipLaunchKernelGGL(loss_grad_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, noise, pred, (long long)n, squared, gscale, dpred);
  VMM_LAUNCH_CHECK();
  return 0;
}
