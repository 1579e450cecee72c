// Fused temporal-attention block for the full-resolution level (C = 64), split-bf16 MFMA + fp32 VALU softmax, gfx950.
//
//   out = x + to_out( softmax_attention( rotary(to_qkv( LayerNorm(x) )) ) )        (vddp.py:615,630,680: Residual(PreNorm(Attention)))
//
// The unfused path writes the 768-wide qkv rows (2.5 GB per site at batch 8) and the 256-wide attention output to HBM and
// reads them back; this kernel keeps everything of a 16-pixel x 11-frame tile on chip: x is read once, out is written once.
// One workgroup = 16 pixels x T frames (rows ordered frame-major: r = t*16 + pixel), 256 threads:
//   phase 0  channel LayerNorm of the T*16 rows, split into bf16 hi/lo, into LDS (A operand of the projections)
//   per head h (8x):
//     phase 1  q,k,v = A (192x64) . Wqkv_h (64x96) on v_mfma_f32_32x32x16_bf16 (3 passes), weights straight from L2 as fragments;
//              epilogue: q *= scale, interleaved-pair rotary on q,k by the frame index; q,k,v rows -> LDS (fp32)
//     phase 2  one thread per (pixel, query frame): 22 keys (conditioning tokens from L1/L2 + 11 frames from LDS), online softmax,
//              relative-position bias; the 32-wide output row is written back over the q slot as bf16 hi/lo
//     phase 3  out_acc += O_h (192x32) . Wout_h (32x64)  (MFMA, accumulators persistent across heads)
//   epilogue   out = out_acc + x  -> HBM
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int TP = 16;            // pixels per workgroup
constexpr int TC = 64;            // channels
constexpr int TRP = 192;          // padded rows (6 MFMA row tiles) >= T*TP
constexpr int XPITCH = TC + 8;    // bf16 per LDS row of the normalised input (144 B)
constexpr int QPITCH = 100;       // floats per LDS row of q|k|v (400 B)
constexpr int DHd = 32;
constexpr int HEADS = 8;
constexpr int HID = HEADS * DHd;

__device__ __forceinline__ unsigned pack_hi(float a, float b, unsigned& lo) {
  const f32x2 v = {a, b};
  const unsigned hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {a - __uint_as_float(hi << 16), b - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
  return hi;
}

struct TBArgs {
  const float* x; int ldx;
  const float* gamma;
  const unsigned short* wqkv;  // [768][64] hi plane | lo plane
  const unsigned short* wout;  // [64][256] hi plane | lo plane
  const float* ek; const float* ev; int ntok;
  const float* bias; int bias_on_cond;
  const float* rot;            // [T][16][2]
  float* out; int ldo;
  int T, HW; float q_scale; float eps; int dbg;  // dbg: ablation bits (1: skip phase 1, 2: skip phase 2, 4: skip phase 3), VMM_TB_DBG
};

__global__ __launch_bounds__(256) void temporal_block_kernel(const TBArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  unsigned short* Xh = reinterpret_cast<unsigned short*>(smem_raw);
  unsigned short* Xl = Xh + TRP * XPITCH;
  float* QKV = reinterpret_cast<float*>(Xl + TRP * XPITCH);
  float* rot_s = QKV + TRP * QPITCH;          // [T][16][2]
  float* bias_s = rot_s + 12 * 32;            // [heads][T][T] (T <= 12)

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int lrow = lane & 31, lk = lane >> 5;
  const int T = a.T, HW = a.HW;
  const int tiles_per_sample = HW / TP;
  const int b = blockIdx.x / tiles_per_sample;
  const int pix0 = (blockIdx.x % tiles_per_sample) * TP;
  const int R = T * TP;
  const long long row_base = (long long)b * T * HW + pix0;  // + t*HW + pl

  // ---- phase 0: LayerNorm + split (thread per row), tables to LDS
  for (int i = tid; i < T * 32; i += 256) rot_s[i] = a.rot[i];
  for (int i = tid; i < HEADS * T * T; i += 256) bias_s[i] = a.bias[i];
  if (tid < TRP) {
    const int r = tid;
    unsigned short* dh = Xh + r * XPITCH;
    unsigned short* dl = Xl + r * XPITCH;
    if (r < R) {
      const int t = r / TP, pl = r % TP;
      const float* xr = a.x + (row_base + (long long)t * HW + pl) * a.ldx;
      f32x4 v[TC / 4];
      float s = 0.f;
#pragma unroll
      for (int c = 0; c < TC / 4; ++c) { v[c] = *reinterpret_cast<const f32x4*>(xr + c * 4); s += v[c].x + v[c].y + v[c].z + v[c].w; }
      const float mean = s * (1.0f / TC);
      float q = 0.f;
#pragma unroll
      for (int c = 0; c < TC / 4; ++c) {
        v[c].x -= mean; v[c].y -= mean; v[c].z -= mean; v[c].w -= mean;
        q += v[c].x * v[c].x + v[c].y * v[c].y + v[c].z * v[c].z + v[c].w * v[c].w;
      }
      const float rstd = 1.0f / sqrtf(q * (1.0f / TC) + a.eps);
#pragma unroll
      for (int c = 0; c < TC / 4; ++c) {
        const f32x4 g = *reinterpret_cast<const f32x4*>(a.gamma + c * 4);
        unsigned l0, l1;
        const unsigned h0 = pack_hi(v[c].x * rstd * g.x, v[c].y * rstd * g.y, l0);
        const unsigned h1 = pack_hi(v[c].z * rstd * g.z, v[c].w * rstd * g.w, l1);
        *reinterpret_cast<uint2*>(dh + c * 4) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(dl + c * 4) = make_uint2(l0, l1);
      }
    } else {
#pragma unroll
      for (int c = 0; c < TC / 4; ++c) {
        *reinterpret_cast<uint2*>(dh + c * 4) = make_uint2(0u, 0u);
        *reinterpret_cast<uint2*>(dl + c * 4) = make_uint2(0u, 0u);
      }
    }
  }
  __syncthreads();

  // row tiles of this wave: waves 0,1 own two of the six 32-row tiles, waves 2,3 one
  const int n_mt = (wave < 2) ? 2 : 1;
  const int mt0 = wave, mt1 = wave + 4;
  f32x16 oacc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) oacc[i][j][r] = 0.f;
  const long long wq_plane = (long long)3 * HID * TC;
  const long long wo_plane = (long long)TC * HID;

  for (int h = 0; h < HEADS; ++h) {
    // ---- phase 1: q,k,v of head h for this wave's row tiles
    {
      f32x16 acc[2][3];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
#pragma unroll
      for (int s = 0; s < TC / 16; ++s) {
        const int ko = s * 16 + lk * 8;
        bf16x8 bh[3], bl[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          const long long n = (long long)(j * HID + h * DHd + lrow);
          bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a.wqkv + n * TC + ko));
          bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a.wqkv + wq_plane + n * TC + ko));
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
          if (i < n_mt) {
            const int row = (i == 0 ? mt0 : mt1) * 32 + lrow;
            const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xh + row * XPITCH + ko));
            const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(Xl + row * XPITCH + ko));
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], acc[i][j], 0, 0, 0);
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], acc[i][j], 0, 0, 0);
            }
          }
        }
      }
      // epilogue: scale, rotary, to LDS.  C layout: col = lrow, row = (r&3) + 8*(r>>2) + 4*lk
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < n_mt) {
          const int mt = (i == 0 ? mt0 : mt1);
#pragma unroll
          for (int r = 0; r < 16; ++r) {
            const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const int t = min(row / TP, T - 1);
            const float2 cs = *reinterpret_cast<const float2*>(rot_s + (t * 16 + (lrow >> 1)) * 2);
#pragma unroll
            for (int j = 0; j < 3; ++j) {
              float v = acc[i][j][r];
              if (j == 0) v *= a.q_scale;
              if (j < 2) {
                const float partner = __shfl_xor(v, 1, 64);
                v = v * cs.x + ((lrow & 1) ? partner : -partner) * cs.y;
              }
              QKV[row * QPITCH + j * DHd + lrow] = v;
            }
          }
        }
      }
    }
    __syncthreads();

    // ---- phase 2: attention, thread per row (pixel pl = r % 16, query frame i = r / 16)
    if (tid < R) {
      const int r = tid, i = r / TP, pl = r % TP;
      float q[DHd], acc[DHd];
#pragma unroll
      for (int d = 0; d < DHd / 4; ++d) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(QKV + r * QPITCH + d * 4);
        q[d * 4] = v.x; q[d * 4 + 1] = v.y; q[d * 4 + 2] = v.z; q[d * 4 + 3] = v.w;
      }
#pragma unroll
      for (int d = 0; d < DHd; ++d) acc[d] = 0.f;
      float m = -INFINITY, l = 0.f;
      const float* brow = bias_s + (h * T + i) * T;
      auto step = [&](const float* kr, const float* vr, float bias_v) {
        float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
#pragma unroll
        for (int d = 0; d < DHd / 4; ++d) {
          const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + d * 4);
          s0 = fmaf(q[d * 4], kv.x, s0); s1 = fmaf(q[d * 4 + 1], kv.y, s1); s2 = fmaf(q[d * 4 + 2], kv.z, s2); s3 = fmaf(q[d * 4 + 3], kv.w, s3);
        }
        const float s = (s0 + s1) + (s2 + s3) + bias_v;
        const float mn = fmaxf(m, s);
        const float f = __expf(m - mn), p = __expf(s - mn);
        l = l * f + p;
#pragma unroll
        for (int d = 0; d < DHd / 4; ++d) {
          const f32x4 vv = *reinterpret_cast<const f32x4*>(vr + d * 4);
          acc[d * 4] = fmaf(p, vv.x, acc[d * 4] * f); acc[d * 4 + 1] = fmaf(p, vv.y, acc[d * 4 + 1] * f);
          acc[d * 4 + 2] = fmaf(p, vv.z, acc[d * 4 + 2] * f); acc[d * 4 + 3] = fmaf(p, vv.w, acc[d * 4 + 3] * f);
        }
        m = mn;
      };
      if (a.ek) {
        for (int j = 0; j < a.ntok; ++j) {
          const long long o = ((long long)b * a.ntok + j) * HID + h * DHd;
          step(a.ek + o, a.ev + o, a.bias_on_cond ? brow[j] : 0.f);
        }
      }
      for (int j = 0; j < T; ++j) {
        const float* kr = QKV + (j * TP + pl) * QPITCH + DHd;
        step(kr, kr + DHd, brow[j]);
      }
      const float inv = 1.0f / l;
      // o row -> bf16 hi | lo over the q slot of this row (only this thread ever read it)
      unsigned short* oh = reinterpret_cast<unsigned short*>(QKV + r * QPITCH);
      unsigned short* ol = oh + DHd;
#pragma unroll
      for (int d = 0; d < DHd / 4; ++d) {
        unsigned l0, l1;
        const unsigned h0 = pack_hi(acc[d * 4] * inv, acc[d * 4 + 1] * inv, l0);
        const unsigned h1 = pack_hi(acc[d * 4 + 2] * inv, acc[d * 4 + 3] * inv, l1);
        *reinterpret_cast<uint2*>(oh + d * 4) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(ol + d * 4) = make_uint2(l0, l1);
      }
    } else if (tid < TRP) {  // padded rows: zero output operand
      unsigned short* oh = reinterpret_cast<unsigned short*>(QKV + tid * QPITCH);
#pragma unroll
      for (int d = 0; d < 2 * DHd / 4; ++d) *reinterpret_cast<uint2*>(oh + d * 4) = make_uint2(0u, 0u);
    }
    __syncthreads();

    // ---- phase 3: out_acc += O_h . Wout_h
#pragma unroll
    for (int s = 0; s < DHd / 16; ++s) {
      const int ko = s * 16 + lk * 8;
      bf16x8 bh[2], bl[2];
#pragma unroll
      for (int j = 0; j < 2; ++j) {
        const long long n = j * 32 + lrow;
        bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a.wout + n * HID + h * DHd + ko));
        bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(a.wout + wo_plane + n * HID + h * DHd + ko));
      }
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        if (i < n_mt) {
          const int row = (i == 0 ? mt0 : mt1) * 32 + lrow;
          const unsigned short* orow = reinterpret_cast<const unsigned short*>(QKV + row * QPITCH);
          const bf16x8 ah = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(orow + ko));
          const bf16x8 al = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(orow + DHd + ko));
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            oacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh[j], oacc[i][j], 0, 0, 0);
            oacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl[j], oacc[i][j], 0, 0, 0);
            oacc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh[j], oacc[i][j], 0, 0, 0);
          }
        }
      }
    }
    __syncthreads();  // the next head overwrites QKV
  }

  // ---- epilogue: residual + store
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (i < n_mt) {
      const int mt = (i == 0 ? mt0 : mt1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = mt * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
        if (row < R) {
          const int t = row / TP, pl = row % TP;
          const long long g = row_base + (long long)t * HW + pl;
#pragma unroll
          for (int j = 0; j < 2; ++j) {
            const int col = j * 32 + lrow;
            a.out[g * a.ldo + col] = oacc[i][j][r] + a.x[g * a.ldx + col];
          }
        }
      }
    }
  }
}

}  // namespace

// Returns 1 (nothing launched) when the shape is outside the kernel's envelope: C == 64, heads == 8, dim_head == 32, T <= 12, HW % 16 == 0.
extern "C" int vmm_temporal_block_bf16x3(const float* x, int32_t ldx, const float* gamma, const float* wqkv_packed, const float* wout_packed,
                                         const float* ek, const float* ev, int32_t ntok, const float* bias, int32_t bias_on_cond,
                                         const float* rot_tab, float* out, int32_t ldo, int32_t B, int32_t T, int32_t HW, int32_t C,
                                         int32_t heads, float q_scale, float eps, vmm_stream_t stream) {
  if (C != TC || heads != HEADS || T * TP > TRP || T > 12 || (HW % TP) || (ldx & 3) || (ldo & 3)) return 1;
  if (bias_on_cond && ek && ntok != T) return -2;
  TBArgs a;
  a.x = x; a.ldx = ldx; a.gamma = gamma;
  a.wqkv = reinterpret_cast<const unsigned short*>(wqkv_packed);
  a.wout = reinterpret_cast<const unsigned short*>(wout_packed);
  a.ek = ek; a.ev = ev; a.ntok = ek ? ntok : 0;
  a.bias = bias; a.bias_on_cond = bias_on_cond; a.rot = rot_tab;
  a.out = out; a.ldo = ldo; a.T = T; a.HW = HW; a.q_scale = q_scale; a.eps = eps;
  { const char* e = getenv("VMM_TB_DBG"); a.dbg = e ? atoi(e) : 0; }
  const size_t shm = sizeof(unsigned short) * 2 * TRP * XPITCH + sizeof(float) * (TRP * QPITCH + 12 * 32 + HEADS * 12 * 12);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&temporal_block_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL(temporal_block_kernel, dim3((unsigned)(B * (HW / TP))), dim3(256), shm, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
