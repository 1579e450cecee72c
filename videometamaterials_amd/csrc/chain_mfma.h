// Pieces shared by the "chained" split-bf16 attention-block kernels (temporal_block_bwd.hip, linattn_block_bwd.hip), gfx950.
//
// An MFMA 32x32 accumulator X{R, C} holds, per lane, ONE column (lane & 31) and 16 rows (register r <-> row (r & 3) + 8 (r >> 2) + 4 (lane >> 5)).
// Split into bf16 hi | lo fragments in register order (F2: two k16 steps), it is an operand of the next MFMA whose contraction runs over X's rows:
//   as the A operand it is X^T (A[i][k] = X[k][i]), as the B operand it is X (B[k][n] = X[k][n]),
// provided both operands enumerate the contraction index in the same order -- which any two accumulators do.  So
//   mmT(X, Y) = X^T . Y        (contraction over the rows of X and Y)
// and a product with the identity as the B operand transposes an accumulator on the matrix pipe (four single-pass MFMAs, exact for hi + lo):
//   transp(X) = X^T . I.
// Operands that come from memory and meet an accumulator are laid out in that register order ("slot" order) by whoever writes them.
#pragma once
#include "vmm_common.h"

namespace chain {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

struct F2 { uint4 h[2], l[2]; };  // hi | lo fragments of the two k16 steps of a 32-row contraction

// contraction / row index that element j of lane half lk holds in k16 step s (accumulator register order)
__device__ __forceinline__ int slot(int s, int lk, int j) { return (j & 3) + 8 * (2 * s + (j >> 2)) + 4 * lk; }
// row of accumulator register r
__device__ __forceinline__ int row_of(int r, int lk) { return (r & 3) + 8 * (r >> 2) + 4 * lk; }

__device__ __forceinline__ f32x16 zero16() {
  f32x16 c;
#pragma unroll
  for (int r = 0; r < 16; ++r) c[r] = 0.f;
  return c;
}

__device__ __forceinline__ void split8(const f32x16& c, int r0, uint4& hi, uint4& lo) {
  hi.x = split_bf16_pair(c[r0 + 0], c[r0 + 1], lo.x);
  hi.y = split_bf16_pair(c[r0 + 2], c[r0 + 3], lo.y);
  hi.z = split_bf16_pair(c[r0 + 4], c[r0 + 5], lo.z);
  hi.w = split_bf16_pair(c[r0 + 6], c[r0 + 7], lo.w);
}
__device__ __forceinline__ void split8v(const float (&v)[8], uint4& hi, uint4& lo) {
  hi.x = split_bf16_pair(v[0], v[1], lo.x);
  hi.y = split_bf16_pair(v[2], v[3], lo.y);
  hi.z = split_bf16_pair(v[4], v[5], lo.z);
  hi.w = split_bf16_pair(v[6], v[7], lo.w);
}
__device__ __forceinline__ F2 tofrag(const f32x16& c) {
  F2 f;
  split8(c, 0, f.h[0], f.l[0]);
  split8(c, 8, f.h[1], f.l[1]);
  return f;
}

__device__ __forceinline__ f32x16 mfma1(const uint4& a, const uint4& b, f32x16 c) {
  return vmm_mfma16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c);
}
// three passes of the split product: lo.hi + hi.lo + hi.hi
__device__ __forceinline__ f32x16 mfma3(const uint4& ah, const uint4& al, const uint4& bh, const uint4& bl, f32x16 c) {
  if constexpr (!VMM_SINGLE_PASS) {
    c = mfma1(al, bh, c);
    c = mfma1(ah, bl, c);
  }
  c = mfma1(ah, bh, c);
  return c;
}
// c += X^T . Y
__device__ __forceinline__ f32x16 mmT(const F2& X, const F2& Y, f32x16 c) {
#pragma unroll
  for (int s = 0; s < 2; ++s) c = mfma3(X.h[s], X.l[s], Y.h[s], Y.l[s], c);
  return c;
}
// identity as a B operand in slot order: I[s] element j = (slot(s, lk, j) == lane & 31) as bf16
__device__ __forceinline__ void identity_frags(int lane, uint4 (&I)[2]) {
  const int lrow = lane & 31, lk = lane >> 5;
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    unsigned w[4];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
      const unsigned e0 = slot(s, lk, 2 * p) == lrow ? VMM_ONE16 : 0u, e1 = slot(s, lk, 2 * p + 1) == lrow ? VMM_ONE16 : 0u;
      w[p] = e0 | (e1 << 16);
    }
    I[s] = uint4{w[0], w[1], w[2], w[3]};
  }
}
// X^T as an accumulator (= hi + lo of X, exactly)
__device__ __forceinline__ f32x16 transp(const F2& X, const uint4 (&I)[2]) {
  f32x16 c = zero16();
#pragma unroll
  for (int s = 0; s < 2; ++s) {
    c = mfma1(X.h[s], I[s], c);
    c = mfma1(X.l[s], I[s], c);
  }
  return c;
}
// the same for a 16-row accumulator piece given as one k16 step (rows 0 .. 15 <-> registers 0 .. 7): columns 0 .. 15 of the result are its transpose
__device__ __forceinline__ f32x16 transp16(const uint4& xh, const uint4& xl, const uint4 (&I)[2]) {
  f32x16 c = zero16();
  c = mfma1(xh, I[0], c);
  c = mfma1(xl, I[0], c);
  return c;
}

}  // namespace chain
