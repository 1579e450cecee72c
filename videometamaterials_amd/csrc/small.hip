// Small host-of-the-network pieces: batched tiny dense layers, embeddings,
// conditioning tokens, layout edges.  Negligible FLOPs; the design goal is FEW launches (one grouped
// launch per dependency level) and coalesced access.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

__device__ __forceinline__ float act_apply(float v, int act) {
  if (act == 1) return silu_f(v);
  if (act == 2) return gelu_erf_f(v);
  return v;
}

// One wave per (job, output column o): W[o, :] is read ONCE into registers (16 bytes per lane and 256-wide K chunk) and applied to all rows,
// eight at a time: per row a lane's four products are summed, the eight row sums of the wave then come out of ONE reduce-scatter butterfly
// (4 + 2 + 1 exchanges halving the values a lane carries, three plain steps) instead of eight full wave reductions.  The first version -- a
// wave per (column, 8-row chunk) with dword loads and eight wave_sum calls -- streamed W once per row chunk and spent its time in the
// reductions: 130 us for the 54 jobs of the FiLM / token-key level (20 MB of weights) against ~10 us now.
constexpr int DR = 8;
constexpr int DKC = 4;  // K chunks of 256 kept in registers (K <= 1024 on the vector path; K <= 256 with four columns per wave)

// columns o0 .. o0 + NC - 1 of one job, all rows (see the kernel's comment)
template <int NC>
__device__ __forceinline__ void dense_columns(const vmm_dense_job& jb, int o0, int lane) {
  if (o0 >= jb.N) return;
  constexpr int KC = NC == 1 ? DKC : 1;
  const int nkc = (jb.K + 255) >> 8;
  f32x4 w4[NC][KC];
  float bv[NC];
#pragma unroll
  for (int n = 0; n < NC; ++n) {
    const int o = min(o0 + n, jb.N - 1);  // (columns past N re-read the last one, never stored)
    bv[n] = jb.b ? jb.b[o] : 0.f;
#pragma unroll
    for (int c = 0; c < KC; ++c) {
      const int k = c * 256 + lane * 4;
      w4[n][c] = (c < nkc && k < jb.K) ? *reinterpret_cast<const f32x4*>(jb.w + (long long)o * jb.K + k) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
  }
  for (int r0 = 0; r0 < jb.rows; r0 += DR) {
    float p[NC][DR];
#pragma unroll
    for (int r = 0; r < DR; ++r) {
      const float* xr = jb.x + (long long)min(r0 + r, jb.rows - 1) * jb.ldx;  // (rows past the end re-read the last one, never stored)
      float acc[NC];
#pragma unroll
      for (int n = 0; n < NC; ++n) acc[n] = 0.f;
#pragma unroll
      for (int c = 0; c < KC; ++c) {
        const int k = c * 256 + lane * 4;
        if (c < nkc) {  // wave-uniform
          f32x4 x4 = k < jb.K ? *reinterpret_cast<const f32x4*>(xr + k) : f32x4{0.f, 0.f, 0.f, 0.f};
          if (jb.act_in) { x4.x = act_apply(x4.x, jb.act_in); x4.y = act_apply(x4.y, jb.act_in); x4.z = act_apply(x4.z, jb.act_in); x4.w = act_apply(x4.w, jb.act_in); }
#pragma unroll
          for (int n = 0; n < NC; ++n) {
            acc[n] = fmaf(x4.x, w4[n][c].x, acc[n]);
            acc[n] = fmaf(x4.y, w4[n][c].y, acc[n]);
            acc[n] = fmaf(x4.z, w4[n][c].z, acc[n]);
            acc[n] = fmaf(x4.w, w4[n][c].w, acc[n]);
          }
        }
      }
#pragma unroll
      for (int n = 0; n < NC; ++n) p[n][r] = acc[n];
    }
    const int r = r0 + (((lane >> 5) & 1) * 4 + ((lane >> 4) & 1) * 2 + ((lane >> 3) & 1));
#pragma unroll
    for (int n = 0; n < NC; ++n) {
      // reduce-scatter over lane bits 5, 4, 3: afterwards p[n][0] of lane L is the sum over the 8 lanes {L ^ (b << 3)} of row 4 L5 + 2 L4 + L3
#pragma unroll
      for (int bit = 5, m = 4; bit >= 3; --bit, m >>= 1) {
        const bool hi = (lane >> bit) & 1;
#pragma unroll
        for (int k = 0; k < m; ++k) {
          const float send = hi ? p[n][k] : p[n][k + m], keep = hi ? p[n][k + m] : p[n][k];
          p[n][k] = keep + lane_xor(send, bit);
        }
      }
      p[n][0] += lane_xor(p[n][0], 2);
      p[n][0] += lane_xor(p[n][0], 1);
      p[n][0] += lane_xor(p[n][0], 0);
      if ((lane & 7) == 0 && r < jb.rows && o0 + n < jb.N) {
        float v = act_apply(p[n][0] + bv[n], jb.act_out);
        if (jb.add) v += jb.add[(long long)r * jb.ldadd + o0 + n];
        jb.y[(long long)r * jb.ldy + o0 + n] = v;
      }
    }
  }
}

__global__ __launch_bounds__(256) void dense_batched_kernel(const vmm_dense_job* __restrict__ jobs) {
  const vmm_dense_job jb = jobs[blockIdx.y];
  const int lane = threadIdx.x & 63;
  const bool vec = (jb.K & 3) == 0 && (jb.ldx & 3) == 0 && jb.K <= 256 * DKC && ((((uintptr_t)jb.w) | ((uintptr_t)jb.x)) & 15) == 0;
  // many rows (the token key / value layers: 88 rows against 256 x 256 weights, 36 of the 54 jobs of the last level): four columns per wave, so
  // that the activations -- re-read by every wave of the job, 88 KB each -- cross the L2 -> L1 path a quarter as often (eight columns: fewer, longer waves -- measured slower) (that traffic, 0.77 GB
  // per step with one column per wave, was what the level's time went to)
  if (vec && jb.rows > 16 && jb.K <= 256) {
    dense_columns<4>(jb, (blockIdx.x * 4 + (threadIdx.x >> 6)) * 4, lane);
    return;
  }
  const int o = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (o >= jb.N) return;
  if (vec) {
    dense_columns<1>(jb, o, lane);
    return;
  }
  const float* wrow = jb.w + (long long)o * jb.K;
  const float bv = jb.b ? jb.b[o] : 0.f;
  // any K / alignment: dword loads, one wave reduction per row
  for (int r0 = 0; r0 < jb.rows; r0 += DR) {
    float acc[DR];
#pragma unroll
    for (int r = 0; r < DR; ++r) acc[r] = 0.f;
    for (int k = lane; k < jb.K; k += 64) {
      const float wv = wrow[k];
#pragma unroll
      for (int r = 0; r < DR; ++r) {
        if (r0 + r < jb.rows) acc[r] = fmaf(act_apply(jb.x[(long long)(r0 + r) * jb.ldx + k], jb.act_in), wv, acc[r]);
      }
    }
#pragma unroll
    for (int r = 0; r < DR; ++r) acc[r] = wave_sum(acc[r]);
    if (lane == 0) {
#pragma unroll
      for (int r = 0; r < DR; ++r) {
        if (r0 + r < jb.rows) {
          float v = act_apply(acc[r] + bv, jb.act_out);
          if (jb.add) v += jb.add[(long long)(r0 + r) * jb.ldadd + o];
          jb.y[(long long)(r0 + r) * jb.ldy + o] = v;
        }
      }
    }
  }
}

__global__ void sinusoidal_kernel(const int64_t* __restrict__ t, int B, int dim, float neg_step, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim / 2;
  if (i >= B * half) return;
  const int b = i / half, j = i - b * half;
  const float f = expf((float)j * neg_step);
  const float a = (float)t[b] * f;
  out[b * dim + j] = sinf(a);
  out[b * dim + half + j] = cosf(a);
}

// block per sample b; thread per embedding column
__global__ void cond_tokens_kernel(const float* __restrict__ cond, const float* __restrict__ w, const float* __restrict__ bias,
                                   const float* __restrict__ null_token, const uint8_t* __restrict__ mask, int F, int D,
                                   float* __restrict__ tokens, float* __restrict__ pooled) {
  const int b = blockIdx.x;
  const bool drop = mask && mask[b];
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int f = 0; f < F; ++f) {
      const float v = cond[b * F + f] * w[d] + bias[d];
      s += v;
      tokens[((long long)b * F + f) * D + d] = drop ? null_token[f * D + d] : v;
    }
    pooled[b * D + d] = s / (float)F;
  }
}

// wave per row
__global__ __launch_bounds__(64) void rows_ln_affine_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ b, float* __restrict__ y, int D, float eps) {
  const int row = blockIdx.x, lane = threadIdx.x;
  const float* xr = x + (long long)row * D;
  float s = 0.f;
  for (int k = lane; k < D; k += 64) s += xr[k];
  const float mean = wave_sum(s) / (float)D;
  float q = 0.f;
  for (int k = lane; k < D; k += 64) { const float d = xr[k] - mean; q += d * d; }
  const float rstd = 1.0f / sqrtf(wave_sum(q) / (float)D + eps);
  for (int k = lane; k < D; k += 64) y[(long long)row * D + k] = (xr[k] - mean) * rstd * w[k] + b[k];
}

__global__ void select_add_kernel(const float* __restrict__ x, const float* __restrict__ null_row, const uint8_t* __restrict__ mask,
                                  const float* __restrict__ add, float* __restrict__ out, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  float v = (mask && mask[b]) ? null_row[d] : x[i];
  if (add) v += add[i];
  out[i] = v;
}

// cond_to_time = 'concat' (vddp.py:789): out[b, :D] = t[b, :], out[b, D:] = mask ? null : x[b, :]
__global__ void select_concat_kernel(const float* __restrict__ x, const float* __restrict__ null_row, const uint8_t* __restrict__ mask,
                                     const float* __restrict__ t, float* __restrict__ out, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  out[(long long)b * 2 * D + d] = t[i];
  out[(long long)b * 2 * D + D + d] = (mask && mask[b]) ? null_row[d] : x[i];
}

// focus_present_mask (vddp.py:431, 438-443, 514-524): a sample that "focuses on the present" attends to its own frame only -- softmax over
// one unmasked key is exactly 1, so its attention output IS its value row.  Row-wise patches around the unchanged attention kernels, float4
// per thread, sample of a row = row / rows_per_sample:
//   mode 0 (forward):   b[row] = a[row]      where focus[sample]   (a = the v third of the qkv rows, b = the attention output)
//   mode 1 (backward):  b[row] = focus ? 0 : a[row]                (dO for the core's backward: masked samples contribute nothing through p)
//   mode 2 (backward):  b[row] += a[row]     where focus[sample]   (dv += dO)
__global__ void focus_rows_kernel(int mode, const float* __restrict__ a, int lda, float* __restrict__ b, int ldb, const uint8_t* __restrict__ focus,
                                  long long rows, int rows_per_sample, int ncols4) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * ncols4) return;
  const long long row = i / ncols4;
  const int c = (int)(i - row * ncols4) * 4;
  const bool f = focus[row / rows_per_sample] != 0;
  const f32x4* ap = reinterpret_cast<const f32x4*>(a + row * lda + c);
  f32x4* bp = reinterpret_cast<f32x4*>(b + row * ldb + c);
  if (mode == 0) {
    if (f) *bp = *ap;
  } else if (mode == 1) {
    *bp = f ? f32x4{0.f, 0.f, 0.f, 0.f} : *ap;
  } else if (f) {
    const f32x4 x = *ap, y = *bp;
    *bp = f32x4{x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w};
  }
}

// cond_att_GRU (vddp.py:546-549, 567-571, 769-770): nn.GRU's recurrence for one layer, one workgroup per sample.  The input-side products
// gi = W_ih x_t + b_ih of ALL time steps come from one batched dense launch; this kernel walks the time axis: gh = W_hh h + b_hh (the state in LDS,
// W_hh transposed so that consecutive threads read consecutive addresses), r = sigmoid(gi_r + gh_r), z = sigmoid(gi_z + gh_z),
// n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h.  Latency-bound by construction (L x 3 dependent steps); a few hundred microseconds per call.
__global__ __launch_bounds__(256) void gru_rec_fwd_kernel(const float* __restrict__ gi, const float* __restrict__ whh_t, const float* __restrict__ bhh,
                                                          float* __restrict__ y, float* __restrict__ hprev, float* __restrict__ gates, int L, int H) {
  extern __shared__ __attribute__((aligned(16))) float gru_hs[];  // [2][H]
  const int tid = threadIdx.x;
  const long long b = blockIdx.x;
  for (int j = tid; j < 2 * H; j += 256) gru_hs[j] = 0.f;
  __syncthreads();
  for (int t = 0; t < L; ++t) {
    const float* hc = gru_hs + (t & 1) * H;
    float* hn = gru_hs + ((t + 1) & 1) * H;
    const long long bt = b * L + t;
    for (int j = tid; j < H; j += 256) {
      float ar = bhh[j], az = bhh[H + j], an = bhh[2 * H + j];
      for (int k = 0; k < H; ++k) {
        const float hk = hc[k];
        const float* w = whh_t + (long long)k * 3 * H;
        ar = fmaf(w[j], hk, ar);
        az = fmaf(w[H + j], hk, az);
        an = fmaf(w[2 * H + j], hk, an);
      }
      const float* g = gi + bt * 3 * H;
      const float r = 1.0f / (1.0f + expf(-(g[j] + ar)));
      const float z = 1.0f / (1.0f + expf(-(g[H + j] + az)));
      const float n = tanhf(g[2 * H + j] + r * an);
      const float hp = hc[j];
      const float h = (1.0f - z) * n + z * hp;
      y[bt * H + j] = h;
      hn[j] = h;
      if (gates) {
        float* G = gates + bt * 4 * H;
        G[j] = r; G[H + j] = z; G[2 * H + j] = n; G[3 * H + j] = an;
      }
      if (hprev) hprev[bt * H + j] = hp;
    }
    __syncthreads();
  }
}

// tokens[b, n, :] = mask[b] ? null[n, :] : g[b, n, :]   (vddp.py:773-778 with per-sample, per-token embeddings)
__global__ void tokens_select_kernel(const float* __restrict__ g, const float* __restrict__ null_tok, const uint8_t* __restrict__ mask, int B, int N, int D,
                                     float* __restrict__ tokens) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long long)B * N * D) return;
  const int b = (int)(i / ((long long)N * D));
  tokens[i] = (mask && mask[b]) ? null_tok[i - (long long)b * N * D] : g[i];
}

// in-place interleaved-pair rotation of x[b, n, h*dh + d] by position n; thread per pair
__global__ void rotary_rows_kernel(float* __restrict__ x, const float* __restrict__ tab, int B, int N, int heads, int dh) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dh / 2;
  const long long total = (long long)B * N * heads * half;
  if (i >= total) return;
  const int fi = (int)(i % half);
  const int n = (int)((i / ((long long)half * heads)) % N);
  float* p = x + i * 2;
  const float c = tab[(n * half + fi) * 2], s = tab[(n * half + fi) * 2 + 1];
  const float a = p[0], b = p[1];
  // (scalar arithmetic on purpose, see LABNOTES 9.8: left to the SLP vectoriser this was v_pk_mul_f32 / two v_pk_fma_f32 / s_nop 0 / v_mov_b32 of the second
  // fma's HIGH half, and under multi-process load the v_mov of lanes 48..63 occasionally read the register BEFORE the packed fma's second pass had
  // written it -- p[1] came out as b * c, once in ~30 forwards of four concurrent processes, never in a process running alone)
  float as = a * s;
  asm volatile("" : "+v"(as));
  float bs = b * s;
  asm volatile("" : "+v"(bs));
  float p0 = fmaf(a, c, -bs);
  asm volatile("" : "+v"(p0));
  const float p1 = fmaf(b, c, as);
  p[0] = p0;
  p[1] = p1;
}

__global__ void relpos_bias_kernel(const float* __restrict__ emb, const int32_t* __restrict__ buckets, int n, int heads,
                                   float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= heads * n * n) return;
  const int h = i / (n * n), ij = i - h * n * n;
  out[i] = emb[buckets[ij] * heads + h];
}

__global__ void ncthw_to_rows_kernel(const float* __restrict__ x, int C, int T, int HW, float* __restrict__ rows, int ld,
                                     long long nrows) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int hw = (int)(r % HW);
  const int t = (int)((r / HW) % T);
  const long long b = r / ((long long)HW * T);
  for (int c = 0; c < ld; ++c) rows[r * ld + c] = (c < C) ? x[((b * C + c) * T + t) * HW + hw] : 0.f;
}

__global__ void rows_to_ncthw_kernel(const float* __restrict__ rows, int ld, int C, int T, int HW, float* __restrict__ x,
                                     long long nrows) {
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int hw = (int)(r % HW);
  const int t = (int)((r / HW) % T);
  const long long b = r / ((long long)HW * T);
  for (int c = 0; c < C; ++c) x[((b * C + c) * T + t) * HW + hw] = rows[r * ld + c];
}

__global__ __launch_bounds__(256) void pointwise_to_ncthw_kernel(const float* __restrict__ rows, int ld, int Cin,
                                                                 const float* __restrict__ w, const float* __restrict__ bias,
                                                                 int Cout, int T, int HW, float* __restrict__ out, long long nrows) {
  extern __shared__ float ws[];  // [Cout][Cin]
  for (int i = threadIdx.x; i < Cout * Cin; i += blockDim.x) ws[i] = w[i];
  __syncthreads();
  const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (r >= nrows) return;
  const int hw = (int)(r % HW);
  const int t = (int)((r / HW) % T);
  const long long b = r / ((long long)HW * T);
  const float* xr = rows + r * ld;
  for (int co = 0; co < Cout; ++co) {
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    for (int ci = 0; ci < Cin; ci += 4) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(xr + ci);
      a0 = fmaf(v.x, ws[co * Cin + ci], a0); a1 = fmaf(v.y, ws[co * Cin + ci + 1], a1);
      a2 = fmaf(v.z, ws[co * Cin + ci + 2], a2); a3 = fmaf(v.w, ws[co * Cin + ci + 3], a3);
    }
    out[((b * Cout + co) * T + t) * HW + hw] = (a0 + a1) + (a2 + a3) + (bias ? bias[co] : 0.f);
  }
}

// Conv1d(k=4, s=2, p=1) + SiLU on (B, Cin, Lin) signals (SignalEmbedding 'CNN', vddp.py:553-561); thread per output
__global__ void conv1d_k4s2_silu_kernel(const float* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                                        float* __restrict__ y, int B, int Cin, int Cout, int Lin, int Lout) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * Cout * Lout) return;
  const int l = i % Lout, co = (i / Lout) % Cout, b = i / (Lout * Cout);
  float acc = bias[co];
  for (int ci = 0; ci < Cin; ++ci) {
    const float* xr = x + ((long long)b * Cin + ci) * Lin;
    const float* wr = w + ((long long)co * Cin + ci) * 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const int p = l * 2 - 1 + k;
      if (p >= 0 && p < Lin) acc = fmaf(xr[p], wr[k], acc);
    }
  }
  y[i] = silu_f(acc);
}

// tokens[b, n, :] = mask[b] ? null_token[n, :] : hidden[b, :]   (vddp.py:767,774-777)
__global__ void tokens_from_hidden_kernel(const float* __restrict__ hidden, const float* __restrict__ null_token,
                                          const uint8_t* __restrict__ mask, int B, int N, int D, float* __restrict__ tokens) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * N * D) return;
  const int d = i % D, n = (i / D) % N, b = i / (D * N);
  tokens[i] = (mask && mask[b]) ? null_token[n * D + d] : hidden[b * D + d];
}

}  // namespace

extern "C" int vmm_dense_batched(const vmm_dense_job* jobs_dev, int32_t njobs, int32_t max_n, vmm_stream_t stream) {
  if (njobs <= 0 || max_n <= 0) return 0;
  hipLaunchKernelGGL(dense_batched_kernel, dim3(cdiv(max_n, 4), njobs), dim3(256), 0, (hipStream_t)stream, jobs_dev);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_sinusoidal_embed(const int64_t* t, int32_t B, int32_t dim, float neg_step, float* out, vmm_stream_t stream) {
  hipLaunchKernelGGL(sinusoidal_kernel, dim3(cdiv(B * (dim / 2), 128)), dim3(128), 0, (hipStream_t)stream, t, B, dim, neg_step, out);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_cond_tokens(const float* cond, const float* w, const float* bias, const float* null_token, const uint8_t* mask,
                               int32_t B, int32_t F, int32_t D, float* tokens, float* pooled, vmm_stream_t stream) {
  hipLaunchKernelGGL(cond_tokens_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, cond, w, bias, null_token, mask, F, D, tokens,
                     pooled);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_rows_layernorm_affine(const float* x, const float* w, const float* b, float* y, int32_t rows, int32_t D,
                                         float eps, vmm_stream_t stream) {
  hipLaunchKernelGGL(rows_ln_affine_kernel, dim3(rows), dim3(64), 0, (hipStream_t)stream, x, w, b, y, D, eps);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_select_add(const float* x, const float* null_row, const uint8_t* mask, const float* add, float* out, int32_t B,
                              int32_t D, vmm_stream_t stream) {
  hipLaunchKernelGGL(select_add_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, x, null_row, mask, add, out, B, D);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_select_concat(const float* x, const float* null_row, const uint8_t* mask, const float* t, float* out, int32_t B, int32_t D,
                                 vmm_stream_t stream) {
  if (!x || !t || !out || (mask && !null_row)) return -1;
  hipLaunchKernelGGL(select_concat_kernel, dim3(cdiv(B * D, 256)), dim3(256), 0, (hipStream_t)stream, x, null_row, mask, t, out, B, D);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_focus_rows(int32_t mode, const float* a, int32_t lda, float* b, int32_t ldb, const uint8_t* focus, int32_t B, int32_t rows_per_sample,
                              int32_t ncols, vmm_stream_t stream) {
  if (mode < 0 || mode > 2 || !a || !b || !focus || (lda & 3) || (ldb & 3) || (ncols & 3) || rows_per_sample <= 0) return -1;
  const long long rows = (long long)B * rows_per_sample;
  hipLaunchKernelGGL(focus_rows_kernel, dim3(cdiv(rows * (ncols / 4), 256)), dim3(256), 0, (hipStream_t)stream, mode, a, lda, b, ldb, focus, rows,
                     rows_per_sample, ncols / 4);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_gru_recurrent(const float* gi, const float* whh_t, const float* bhh, float* y, float* hprev, float* gates, int32_t B, int32_t L,
                                 int32_t H, vmm_stream_t stream) {
  if (!gi || !whh_t || !bhh || !y || H < 1 || H > 4096 || L < 1) return -1;
  if (B <= 0) return 0;
  hipLaunchKernelGGL(gru_rec_fwd_kernel, dim3((unsigned)B), dim3(256), sizeof(float) * 2 * H, (hipStream_t)stream, gi, whh_t, bhh, y, hprev, gates, L, H);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_tokens_select(const float* g, const float* null_token, const uint8_t* mask, int32_t B, int32_t N, int32_t D, float* tokens,
                                 vmm_stream_t stream) {
  if (!g || !tokens || (mask && !null_token)) return -1;
  hipLaunchKernelGGL(tokens_select_kernel, dim3(cdiv((long long)B * N * D, 256)), dim3(256), 0, (hipStream_t)stream, g, null_token, mask, B, N, D, tokens);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_rotary_rows(float* x, const float* rot_tab, int32_t B, int32_t N, int32_t heads, int32_t dh,
                               vmm_stream_t stream) {
  const long long total = (long long)B * N * heads * (dh / 2);
  hipLaunchKernelGGL(rotary_rows_kernel, dim3(cdiv(total, 256)), dim3(256), 0, (hipStream_t)stream, x, rot_tab, B, N, heads, dh);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_relpos_bias(const float* emb, const int32_t* buckets, int32_t n, int32_t heads, float* out, vmm_stream_t stream) {
  hipLaunchKernelGGL(relpos_bias_kernel, dim3(cdiv(heads * n * n, 256)), dim3(256), 0, (hipStream_t)stream, emb, buckets, n, heads, out);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_ncthw_to_rows(const float* x, int32_t B, int32_t C, int32_t T, int32_t HW, float* rows, int32_t ld,
                                 vmm_stream_t stream) {
  const long long nrows = (long long)B * T * HW;
  hipLaunchKernelGGL(ncthw_to_rows_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, (hipStream_t)stream, x, C, T, HW, rows, ld, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_rows_to_ncthw(const float* rows, int32_t ld, int32_t B, int32_t C, int32_t T, int32_t HW, float* x,
                                 vmm_stream_t stream) {
  const long long nrows = (long long)B * T * HW;
  hipLaunchKernelGGL(rows_to_ncthw_kernel, dim3(cdiv(nrows, 256)), dim3(256), 0, (hipStream_t)stream, rows, ld, C, T, HW, x, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_pointwise_to_ncthw(const float* rows, int32_t ld, int32_t Cin, const float* w, const float* bias, int32_t B,
                                      int32_t Cout, int32_t T, int32_t HW, float* out, vmm_stream_t stream) {
  if ((Cin & 3) || (ld & 3) || Cout * Cin > 8192) return -1;
  const long long nrows = (long long)B * T * HW;
  hipLaunchKernelGGL(pointwise_to_ncthw_kernel, dim3(cdiv(nrows, 256)), dim3(256), sizeof(float) * Cout * Cin, (hipStream_t)stream,
                     rows, ld, Cin, w, bias, Cout, T, HW, out, nrows);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_conv1d_k4s2_silu(const float* x, const float* w, const float* bias, float* y, int32_t B, int32_t Cin,
                                    int32_t Cout, int32_t Lin, vmm_stream_t stream) {
  const int Lout = (Lin + 2 - 4) / 2 + 1;
  if (Lout < 1) return -1;
  hipLaunchKernelGGL(conv1d_k4s2_silu_kernel, dim3(cdiv(B * Cout * Lout, 128)), dim3(128), 0, (hipStream_t)stream, x, w, bias, y, B,
                     Cin, Cout, Lin, Lout);
  VMM_LAUNCH_CHECK();
  return 0;
}

extern "C" int vmm_tokens_from_hidden(const float* hidden, const float* null_token, const uint8_t* mask, int32_t B, int32_t N,
                                      int32_t D, float* tokens, vmm_stream_t stream) {
  hipLaunchKernelGGL(tokens_from_hidden_kernel, dim3(cdiv(B * N * D, 256)), dim3(256), 0, (hipStream_t)stream, hidden, null_token,
                     mask, B, N, D, tokens);
  VMM_LAUNCH_CHECK();
  return 0;
}
