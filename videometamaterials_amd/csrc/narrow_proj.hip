// 1x1 / Linear with a wide contraction and 64 output columns -- the training step's to_out (K = 256 -> 64, vddp.py:325, 421) and the data gradient
// of to_qkv (K = 768 -> 64, autograd of vddp.py:319, 413) at the 96 x 96 level -- on the split-bf16 matrix cores, gfx950.
//
//   out[row][0..64) = sum_k a[row][k] w[k][.] (+ bias) (+ res[row][.])
//
// These are HBM-bound sweeps: 1-3 KB read per row for 256 bytes written, 2 K 64 flops per row = a tenth of the matrix pipes' time at the
// HBM rate.  The generic implicit GEMM stages both operands through LDS behind barriers and reaches 1.6-2.6 TB/s on them; the A-stationary
// projection kernel fits one row tile per CU at K = 256.  Here nothing is staged at all:
//   * activations are the MFMA's B operand (lane = row, eight consecutive k): a lane's fragment is 32 contiguous bytes of its row, loaded
//     straight into registers (the two lane halves read adjacent 32-byte pieces; two k16 steps are requested together, so every 128-byte line
//     of a row is consumed by four back-to-back loads) and split there;
//   * weights are the A operand in the fragment order vmm_pack_weights fmt 2 leaves them in: 16 bytes per lane, plane and step straight from
//     L1 / L2 (K 64 x 4 bytes = 64-192 KB, shared by every wave of the launch);
//   * a wave owns 64 rows x 64 columns (four 32 x 32 accumulators; lane = row, a register quad = four consecutive columns -> 16-byte
//     stores), walks K with two double steps of loads in flight, and never meets another wave: no LDS, no barrier.
#include "vmm_common.h"
#include "../../include/vmm_kernels.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
constexpr int DEPTH = 2;  // double steps (2 x k16 = 128 bytes of every row) of loads in flight per wave

struct NPArgs {
  vmm_conv_desc p;
  long long rows;
  int ksteps;  // K / 16 (even)
  const float* res_coef;  // non-NULL: the residual enters as silu(res * a + b'), (a, b') = res_coef[sample][column][2] (the ResnetBlock tail, vddp.py:311)
  int rows_per_sample;
};

// A16: the rows, the residual and the output are bf16-STORED maps (the "bf16" throughput mode): a lane's fragment is 16 contiguous bytes of its
// row and IS the matrix operand (no split); one pass on the weights' hi planes, like the mode's other kernels.
template <bool A16>
__global__ __launch_bounds__(256, 2) void narrow_proj_x3_kernel(const NPArgs a) {
  const vmm_conv_desc& p = a.p;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int l31 = lane & 31, half = lane >> 5;
  const long long row0 = ((long long)blockIdx.x * 4 + wave) * 64;
  if (row0 >= a.rows) return;
  // this lane's two rows (row fragments 0 and 1); rows past the end re-read the last row and are not stored
  const long long r0 = min(row0 + l31, a.rows - 1), r1 = min(row0 + 32 + l31, a.rows - 1);
  const int KS = a.ksteps, KS1 = p.C1 >> 4;
  auto src = [&](long long r, int ks) -> const float* {  // 8 floats of row r at k = ks * 16 + half * 8 (two sources: concatenated channels)
    return ks < KS1 ? p.a1 + r * p.lda1 + ks * 16 + half * 8 : p.a2 + r * p.lda2 + (ks - KS1) * 16 + half * 8;
  };
  // weights: plane (nt, ks, hi | lo) = 64 lanes x 16 bytes
  const uint4* wq = reinterpret_cast<const uint4*>(p.w) + lane;
  auto wplane = [&](int nt, int ks, int lo) -> const uint4* { return wq + ((long long)(nt * KS + ks) * 2 + lo) * 64; };

  // A request covers TWO k16 steps: the four 16-byte loads of a row fragment then touch one whole 128-byte line of every row back to back
  // (requested one step at a time the second half of a line came ~300 cycles later, after the CU's other waves had pushed it out of the L1)
  auto src16 = [&](long long r, int ks) -> const bf16s* {
    return ks < KS1 ? reinterpret_cast<const bf16s*>(p.a1) + r * p.lda1 + ks * 16 + half * 8
                    : reinterpret_cast<const bf16s*>(p.a2) + r * p.lda2 + (ks - KS1) * 16 + half * 8;
  };
  f32x4 xa[DEPTH][2][2][2];   // [stage][step of the pair][row fragment][first / second four floats]  (A16: [..][0] carries the fragment's 16 bytes as bits)
  uint4 wv[DEPTH][2][2][2];   // [stage][step of the pair][column tile][hi | lo]
  auto request = [&](int st, int kd) {
    const int k = min(2 * kd, KS - 2);  // (the tail re-requests the last pair: unconditional loads, no branch in the step)
#pragma unroll
    for (int u = 0; u < 2; ++u) {
      if constexpr (A16) {
        xa[st][u][0][0] = __builtin_bit_cast(f32x4, *reinterpret_cast<const uint4*>(src16(r0, k + u)));
        xa[st][u][1][0] = __builtin_bit_cast(f32x4, *reinterpret_cast<const uint4*>(src16(r1, k + u)));
        continue;
      }
      const float* s0 = src(r0, k + u);
      const float* s1 = src(r1, k + u);
      xa[st][u][0][0] = *reinterpret_cast<const f32x4*>(s0);
      xa[st][u][0][1] = *reinterpret_cast<const f32x4*>(s0 + 4);
      xa[st][u][1][0] = *reinterpret_cast<const f32x4*>(s1);
      xa[st][u][1][1] = *reinterpret_cast<const f32x4*>(s1 + 4);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
        wv[st][u][nt][0] = *wplane(nt, k + u, 0);
        if constexpr (!A16) wv[st][u][nt][1] = *wplane(nt, k + u, 1);
      }
  };
  f32x16 acc[2][2];  // [column tile][row fragment]
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  const int KD = KS / 2;
#pragma unroll
  for (int s = 0; s < DEPTH; ++s) request(s, s);
  for (int kd0 = 0; kd0 < KD; kd0 += DEPTH) {
#pragma unroll
    for (int s = 0; s < DEPTH; ++s) {
      if (kd0 + s < KD) {  // (wave-uniform)
        bf16x8 bh[2][2], bl[2][2], ah[2][2], al[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
#pragma unroll
          for (int rf = 0; rf < 2; ++rf) {
            if constexpr (A16) {
              bh[u][rf] = __builtin_bit_cast(bf16x8, xa[s][u][rf][0]);
              continue;
            }
            uint4 h, l;
            h.x = split_bf16_pair(xa[s][u][rf][0].x, xa[s][u][rf][0].y, l.x);
            h.y = split_bf16_pair(xa[s][u][rf][0].z, xa[s][u][rf][0].w, l.y);
            h.z = split_bf16_pair(xa[s][u][rf][1].x, xa[s][u][rf][1].y, l.z);
            h.w = split_bf16_pair(xa[s][u][rf][1].z, xa[s][u][rf][1].w, l.w);
            bh[u][rf] = __builtin_bit_cast(bf16x8, h);
            bl[u][rf] = __builtin_bit_cast(bf16x8, l);
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt) {
            ah[u][nt] = __builtin_bit_cast(bf16x8, wv[s][u][nt][0]);
            if constexpr (!A16) al[u][nt] = __builtin_bit_cast(bf16x8, wv[s][u][nt][1]);
          }
        }
        request(s, kd0 + s + DEPTH);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
          // pass-major, lo products first
          if constexpr (!A16) {
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int rf = 0; rf < 2; ++rf) acc[nt][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u][nt], bl[u][rf], acc[nt][rf], 0, 0, 0);
#pragma unroll
            for (int nt = 0; nt < 2; ++nt)
#pragma unroll
              for (int rf = 0; rf < 2; ++rf) acc[nt][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[u][nt], bh[u][rf], acc[nt][rf], 0, 0, 0);
          }
#pragma unroll
          for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int rf = 0; rf < 2; ++rf) acc[nt][rf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[u][nt], bh[u][rf], acc[nt][rf], 0, 0, 0);
        }
      }
    }
  }
  // acc[nt][rf][r]: column nt * 32 + (r & 3) + 8 (r >> 2) + 4 half, row = row fragment rf, lane l31
#pragma unroll
  for (int rf = 0; rf < 2; ++rf) {
    const long long row = row0 + rf * 32 + l31;
    if (row >= a.rows) continue;
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int c = nt * 32 + 8 * q + 4 * half;
        f32x4 v = {acc[nt][rf][4 * q], acc[nt][rf][4 * q + 1], acc[nt][rf][4 * q + 2], acc[nt][rf][4 * q + 3]};
        if (p.bias) v += *reinterpret_cast<const f32x4*>(p.bias + c);
        if (p.res) {
          f32x4 r = A16 ? ld4(reinterpret_cast<const bf16s*>(p.res) + row * p.ldres + c) : *reinterpret_cast<const f32x4*>(p.res + row * p.ldres + c);
          if (a.res_coef) {
            const float* cf = a.res_coef + ((row / a.rows_per_sample) * 64 + c) * 2;
            const f32x4 c0 = *reinterpret_cast<const f32x4*>(cf), c1 = *reinterpret_cast<const f32x4*>(cf + 4);
            r = f32x4{silu_rcp(r.x * c0.x + c0.y), silu_rcp(r.y * c0.z + c0.w), silu_rcp(r.z * c1.x + c1.y), silu_rcp(r.w * c1.z + c1.w)};
          }
          v += r;
        }
        if constexpr (A16) st4(reinterpret_cast<bf16s*>(p.out) + row * p.ldo + c, v);
        else *reinterpret_cast<f32x4*>(p.out + row * p.ldo + c) = v;
      }
  }
}

}  // namespace

// d->w = vmm_pack_weights fmt 2 of the (K, 64) operand.  Envelope: KH = KW = 1, stride 1, identity row mapping, Cout == 64, C1 (and C2)
// multiples of 16 with K = C1 + C2 >= 64, no fused operand transform, no rotary / q-scale epilogue; bias / residual as in vmm_conv_igemm_*
// (res may alias out).  Returns 1 (nothing launched) outside it.
static int np_launch(const vmm_conv_desc& d, const float* res_coef, int rows_per_sample, vmm_stream_t stream) {
  const bool shape_ok = d.KH == 1 && d.KW == 1 && d.stride == 1 && d.off_h == 0 && d.off_w == 0 && d.Hv == d.Hin && d.Wv == d.Win && d.oscale == 1 &&
                        d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 && d.a_mode == 0 && !d.a_img_mod && !d.rot_ncols && !d.q_ncols;
  const int K = d.C1 + d.C2;
  const bool chan_ok = d.Cout == 64 && d.C1 > 0 && d.C1 % 16 == 0 && d.C2 % 16 == 0 && K >= 64 && K % 32 == 0 && (d.lda1 & 3) == 0 && (!d.C2 || (d.lda2 & 3) == 0) &&
                       (d.ldo & 3) == 0 && (!d.res || (d.ldres & 3) == 0);
  if (!shape_ok || !chan_ok) return 1;
  if (d.act_bf16 && (d.act_bf16 != 3 || (d.lda1 & 7) || (d.C2 && (d.lda2 & 7)))) return -1;  // bf16-stored rows, residual and output alike
  NPArgs a;
  a.p = d;
  a.rows = (long long)d.nimg * d.Hv * d.Wv;
  a.ksteps = K / 16;
  a.res_coef = res_coef;
  a.rows_per_sample = rows_per_sample;
  if (a.rows <= 0) return 0;
  if (d.act_bf16) hipLaunchKernelGGL(narrow_proj_x3_kernel<true>, dim3((unsigned)cdiv(a.rows, 256)), dim3(256), 0, (hipStream_t)stream, a);
  else hipLaunchKernelGGL(narrow_proj_x3_kernel<false>, dim3((unsigned)cdiv(a.rows, 256)), dim3(256), 0, (hipStream_t)stream, a);
  VMM_LAUNCH_CHECK();
  return 0;
}
extern "C" int vmm_proj_narrow_bf16x3(const vmm_conv_desc* dp, vmm_stream_t stream) { return np_launch(*dp, nullptr, 1, stream); }
// ResnetBlock tail (vddp.py:311) on the same kernel: out = silu(res * a + b') + proj(x), (a, b') = res_coef [B][64][2] from vmm_groupnorm_coef
// (vmm_proj_bf16x3_res_silu's contract; res may alias out)
extern "C" int vmm_proj_narrow_bf16x3_res_silu(const vmm_conv_desc* dp, const float* res_coef, int32_t rows_per_sample, vmm_stream_t stream) {
  if (!dp->res || !res_coef || rows_per_sample <= 0) return -1;
  return np_launch(*dp, res_coef, rows_per_sample, stream);
}
