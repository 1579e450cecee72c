// Pieces shared by the implicit-GEMM kernels (igemm_conv.hip: exact fp32 MFMA, igemm_bf16x3.hip: split-bf16 MFMA):
// the gather of one float4 of the A operand (zero padding, two concatenated sources, optional fused
// GroupNorm+FiLM+SiLU), the incremental (tap, channel) decoding of the K index -- no integer division in the K loop --
// and the epilogue (bias, q-scale, rotary, residual, row remapping for the transposed-conv phases).
// All row indices are 32-bit (rows < 2^31); only the final address arithmetic is 64-bit.
#pragma once
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace igemm {

__device__ __forceinline__ float silu_fast(float x) { return x * __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }

struct RowInfo {
  int img;       // frame index, -1 when the row is beyond M
  int ih0, iw0;  // a*stride, b*stride
};

__device__ __forceinline__ RowInfo decode_row(const vmm_conv_desc& p, unsigned m, unsigned M) {
  RowInfo r;
  if (m < M) {
    const unsigned hw = (unsigned)(p.Hv * p.Wv);
    const unsigned img = m / hw;
    const unsigned rem = m - img * hw;
    const unsigned a = rem / (unsigned)p.Wv, b = rem - a * (unsigned)p.Wv;
    r.img = (int)img; r.ih0 = (int)a * p.stride; r.iw0 = (int)b * p.stride;
  } else {
    r.img = -1; r.ih0 = 0; r.iw0 = 0;
  }
  return r;
}

// position of this thread's 4 consecutive K elements: k = tap*Cin + ci, tap = kh*KW + kw; advanced by BK per chunk
struct KPos {
  int ci, kh, kw, k;
  __device__ __forceinline__ void init(const vmm_conv_desc& p, int k0, int Cin) {
    k = k0;
    const int tap = k0 / Cin;
    ci = k0 - tap * Cin;
    kh = tap / p.KW;
    kw = tap - kh * p.KW;
  }
  __device__ __forceinline__ void advance(const vmm_conv_desc& p, int step, int Cin) {
    k += step;
    ci += step;
    while (ci >= Cin) {
      ci -= Cin;
      if (++kw == p.KW) { kw = 0; ++kh; }
    }
  }
};

// one float4 of A for row `ri` at K position `kp` (all-zero when padded / out of range)
__device__ __forceinline__ f32x4 load_a4(const vmm_conv_desc& p, const RowInfo& ri, const KPos& kp, int Ktot) {
  f32x4 v = {0.f, 0.f, 0.f, 0.f};
  int ih = ri.ih0 + p.off_h + p.sgn_h * kp.kh, iw = ri.iw0 + p.off_w + p.sgn_w * kp.kw;
  // periodic padding (vddp.py:163-243): a tap that leaves the frame reads the opposite border (kernels reach at most one frame beyond)
  if (p.wrap_h) ih = ih < 0 ? ih + p.Hin : (ih >= p.Hin ? ih - p.Hin : ih);
  if (p.wrap_w) iw = iw < 0 ? iw + p.Win : (iw >= p.Win ? iw - p.Win : iw);
  if (kp.k < Ktot && ri.img >= 0 && (unsigned)ih < (unsigned)p.Hin && (unsigned)iw < (unsigned)p.Win) {
    const long long pix = (long long)((ri.img * p.Hin + ih) * p.Win + iw);
    if (kp.ci < p.C1) {
      v = *reinterpret_cast<const f32x4*>(p.a1 + pix * p.lda1 + kp.ci);
      if (p.a_mode == 1) {
        const float* cf = p.a_coef + ((long long)(ri.img / p.a_imgs_per_sample) * p.C1 + kp.ci) * 2;
        const f32x4 c0 = *reinterpret_cast<const f32x4*>(cf);
        const f32x4 c1 = *reinterpret_cast<const f32x4*>(cf + 4);
        v.x = silu_fast(v.x * c0.x + c0.y);
        v.y = silu_fast(v.y * c0.z + c0.w);
        v.z = silu_fast(v.z * c1.x + c1.y);
        v.w = silu_fast(v.w * c1.z + c1.w);
      }
    } else {
      v = *reinterpret_cast<const f32x4*>(p.a2 + pix * p.lda2 + (kp.ci - p.C1));
    }
  }
  return v;
}

// Epilogue for one wave: MT x NT accumulator tiles of 32x32 (C/D layout: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)).
template <int MT, int NT>
__device__ __forceinline__ void epilogue(const vmm_conv_desc& p, const f32x16 (&acc)[MT][NT], unsigned m_wave, int n_wave, unsigned M, int lane) {
  const int lrow = lane & 31, lk = lane >> 5;
  const bool identity_rows = (p.oscale == 1 && p.Hout == p.Hv && p.Wout == p.Wv && p.ooh == 0 && p.oow == 0);
  const bool rotary = p.rot_ncols > 0;
  const unsigned hw = (unsigned)(p.Hv * p.Wv);
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const unsigned mb = m_wave + i * 32;
    // frame index / in-frame offset of the tile's first row: one division per tile, rows then wrap at most once when hw >= 32
    const unsigned img_b = mb / hw, rem_b = mb - img_b * hw;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned roff = (r & 3) + 8 * (r >> 2) + 4 * lk;
      const unsigned m = mb + roff;
      const bool mvalid = m < M;
      unsigned img = img_b, rem = rem_b + roff;
      if (hw >= 32u) {
        if (rem >= hw) { rem -= hw; ++img; }
      } else {
        img = m / hw;
        rem = m - img * hw;
      }
      long long orow = m;
      if (!identity_rows) {
        const unsigned a = rem / (unsigned)p.Wv, b = rem - a * (unsigned)p.Wv;
        orow = (long long)((img * p.Hout + a * p.oscale + p.ooh) * p.Wout + b * p.oscale + p.oow);
      }
      int t = 0;
      if (rotary) t = (p.rot_HW == (int)hw) ? (int)(img % (unsigned)p.rot_T) : (int)((m / (unsigned)p.rot_HW) % (unsigned)p.rot_T);
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n_wave + j * 32 + lrow;
        const bool cvalid = col < p.Cout;
        float v = acc[i][j][r];
        if (p.bias && cvalid) v += p.bias[col];
        if (col < p.q_ncols) v *= p.q_scale;
        if (rotary) {  // wave-uniform branch
          const float partner = __shfl_xor(v, 1, 64);
          if (col < p.rot_ncols) {
            // pair index inside the head: rot_dh (the head width, attn_dim_head of the temporal attentions) is even; the table carries identity
            // pairs beyond the rotary span min(32, rot_dh) (hostmath.rotary_table)
            const int fi = ((p.rot_dh & (p.rot_dh - 1)) ? col % p.rot_dh : (col & (p.rot_dh - 1))) >> 1;
            const float2 cs = *reinterpret_cast<const float2*>(p.rot_tab + (t * (p.rot_dh >> 1) + fi) * 2);
            v = v * cs.x + ((col & 1) ? partner : -partner) * cs.y;
          }
        }
        if (mvalid && cvalid) {
          if (p.res) v += p.res[orow * p.ldres + col];
          p.out[orow * p.ldo + col] = v;
        }
      }
    }
  }
}

}  // namespace igemm
