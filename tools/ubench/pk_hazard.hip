// Standalone reproducer attempt for LABNOTES 9.8: the in-place rotation of the temporal attentions' token keys once came out as p1 = b c instead of
// b c + a s in lanes 48..63 of a wave -- about once in 20 000 executions of the wave-instruction, only while other PROCESSES had waves on the SIMD.  The
// ISA of that build (round 4) was
//     v_pk_mul_f32 v[2:3], ...                        ; v3 = b c
//     v_pk_fma_f32 v[8:9], ...                        ; v8 = p0 = a c - b s
//     v_pk_fma_f32 v[2:3], ... op_sel_hi:[1,0,1]      ; v3 = s a + v3 = p1
//     s_nop 0
//     v_mov_b32 v9, v3                                ; <- read the OLD v3 (= b c)
//     global_store_dwordx2 ..., v[8:9]
// i.e. a non-packed read of a packed-fp32 instruction's HIGH result one wait state after its issue (the s_nop 0 is hipcc's own: LLVM's hazard recogniser
// gives a VOP3P op_sel destination one wait state before a VALU read).  This program replays that sequence, with the registers of the original, in inline
// asm (`seq` 0), the sequence today's hipcc emits for the same source (`seq` 1: v_pk_mul / s_nop 0 / in-place cross-half v_pk_fma, result read at once),
// `seq` 2 = seq 0 with NO wait state at all (the control: does the hardware interlock the read by itself?) and `seq` 3 = seq 0 with two wait states,
// eight waves per SIMD, fresh operands every trip, and counts results that differ from the scalar fmaf chain -- and how many of those equal b c exactly.
//
// build:  hipcc --offload-arch=gfx950 -O3 tools/ubench/pk_hazard.hip -o tools/ubench/pk_hazard
// run:    for p in 1 2 3 4; do tools/ubench/pk_hazard 0 400 & done; wait     (arguments: seq, launches; tools/ubench/run_pk_hazard.sh does all three)
// Outcome of the run committed with this file: LABNOTES 10.3.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ float val(unsigned h) {  // a float in [-2, 2) with a full mantissa
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return (float)(int)(h >> 8) * (1.f / 4194304.f) - 2.f;
}

// X = v[6:7] = (a, b), Y = v[4:5] = (c, s);  T = v[2:3], R = v[8:9] -- the registers of the original; WAIT = what stands between the packed fma and the read
#define PK_SEQ_ORIGINAL(WAIT)                                                                                                     \
  asm volatile("v_mov_b32 v6, %2\n\t"                                                                                            \
               "v_mov_b32 v7, %3\n\t"                                                                                            \
               "v_mov_b32 v4, %4\n\t"                                                                                            \
               "v_mov_b32 v5, %5\n\t"                                                                                            \
               "s_nop 4\n\t"                                                                                                     \
               "v_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,1] op_sel_hi:[1,0]\n\t"             /* v2 = b s, v3 = b c */         \
               "v_pk_fma_f32 v[8:9], v[6:7], v[4:5], v[2:3] op_sel_hi:[1,1,1] neg_lo:[0,0,1]\n\t" /* v8 = a c - b s = p0 */        \
               "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[1,0,1]\n\t"                /* v3 = s a + v3 = p1 */         \
               WAIT "v_mov_b32 v9, v3\n\t"                                                                                       \
               "s_nop 4\n\t"                                                                                                     \
               "v_mov_b32 %0, v8\n\t"                                                                                            \
               "v_mov_b32 %1, v9\n\t"                                                                                            \
               : "=v"(p0), "=v"(p1)                                                                                              \
               : "v"(a), "v"(b), "v"(c), "v"(s)                                                                                  \
               : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9")

// SEQ 0: the original with hipcc's own `s_nop 0`; 2: no wait state at all (does the hardware interlock the read by itself?); 3: `s_nop 1`;
// 4: the STORE of v[8:9] right behind the v_mov, result read back from memory (hypothesis B, see there);
// 1: what today's hipcc emits for `p[0] = a c - b s; p[1] = b c + a s` -- v_pk_mul / s_nop 0 / an in-place cross-half v_pk_fma, read at once
template <int SEQ>
__global__ __launch_bounds__(256) void pk_kernel(int trips, unsigned seed, unsigned* counts, float* out) {
  const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
  unsigned bad = 0, bad_bc = 0, bad_p0 = 0;
  for (int it = 0; it < trips; ++it) {
    const unsigned k = (tid * 0x9e3779b9u) ^ (seed + it * 0x85ebca6bu);
    const float a = val(k), b = val(k + 1), c = val(k + 2), s = val(k + 3);
    float p0, p1;
    if constexpr (SEQ == 0) PK_SEQ_ORIGINAL("s_nop 0\n\t");
    if constexpr (SEQ == 2) PK_SEQ_ORIGINAL("");
    if constexpr (SEQ == 3) PK_SEQ_ORIGINAL("s_nop 1\n\t");
    if constexpr (SEQ == 1)
      asm volatile("v_mov_b32 v6, %2\n\t"
                   "v_mov_b32 v7, %3\n\t"
                   "v_mov_b32 v4, %4\n\t"
                   "v_mov_b32 v5, %5\n\t"
                   "s_nop 4\n\t"
                   "v_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[0,1] neg_hi:[0,1]\n\t"                 // v2 = a s, v3 = -(b s)
                   "s_nop 0\n\t"
                   "v_pk_fma_f32 v[2:3], v[6:7], v[4:5], v[2:3] op_sel:[0,0,1] op_sel_hi:[1,0,0]\n\t"  // v2 = a c + v3 = p0, v3 = b c + v2 (old) = p1
                   "v_mov_b32 %0, v2\n\t"
                   "v_mov_b32 %1, v3\n\t"
                   : "=v"(p0), "=v"(p1)
                   : "v"(a), "v"(b), "v"(c), "v"(s)
                   : "v2", "v3", "v4", "v5", "v6", "v7");
    if constexpr (SEQ == 4) {
      // hypothesis B: not the v_mov's read of v3 but the STORE's read of v9 is early -- the original stored v[8:9] right behind the v_mov, and what the
      // p0 instruction leaves in v9 (its unused high half) is b c + 0 in the original too.  Store from inside the sequence, read back from memory.
      float* q = out + (size_t)tid * 2;
      asm volatile("v_mov_b32 v6, %1\n\t"
                   "v_mov_b32 v7, %2\n\t"
                   "v_mov_b32 v4, %3\n\t"
                   "v_mov_b32 v5, %4\n\t"
                   "v_mov_b32 v10, 0\n\t"
                   "v_mov_b32 v11, 0\n\t"
                   "s_nop 4\n\t"
                   "v_pk_mul_f32 v[2:3], v[6:7], v[4:5] op_sel:[1,1] op_sel_hi:[1,0]\n\t"               // v2 = b s, v3 = b c
                   "v_pk_fma_f32 v[8:9], v[6:7], v[4:5], v[10:11] op_sel_hi:[1,0,1]\n\t"                // v9 = b c + 0 (the unused half of the p0 instruction)
                   "v_pk_fma_f32 v[8:9], v[6:7], v[4:5], v[2:3] op_sel_hi:[0,0,0] neg_lo:[0,0,1] neg_hi:[0,0,1]\n\t"  // v8 = a c - b s = p0 (v9 = a c - b s too)
                   "v_pk_fma_f32 v[10:11], v[6:7], v[4:5], v[10:11] op_sel_hi:[1,0,1]\n\t"              // v11 = b c
                   "v_mov_b32 v9, v11\n\t"                                                             // v9 = b c: what a too-early store would carry
                   "v_pk_fma_f32 v[2:3], v[4:5], v[6:7], v[2:3] op_sel_hi:[1,0,1]\n\t"                  // v3 = s a + v3 = p1
                   "s_nop 0\n\t"
                   "v_mov_b32 v9, v3\n\t"
                   "global_store_dwordx2 %0, v[8:9], off\n\t"
                   "s_waitcnt vmcnt(0)\n\t"
                   :
                   : "v"(q), "v"(a), "v"(b), "v"(c), "v"(s)
                   : "v2", "v3", "v4", "v5", "v6", "v7", "v8", "v9", "v10", "v11", "memory");
      p0 = __builtin_nontemporal_load(q);
      p1 = __builtin_nontemporal_load(q + 1);
    }
    // the same roundings as the instruction sequence: one product rounded, then a fused multiply-add
    const float want0 = fmaf(a, c, -(b * s));
    const float want1 = SEQ == 1 ? fmaf(b, c, a * s) : fmaf(s, a, b * c);
    if (p1 != want1) {
      ++bad;
      if (p1 == b * c) ++bad_bc;
    }
    if (p0 != want0) ++bad_p0;
  }
  if (bad) atomicAdd(counts, bad);
  if (bad_bc) atomicAdd(counts + 1, bad_bc);
  if (bad_p0) atomicAdd(counts + 2, bad_p0);
  if (bad && (threadIdx.x & 63) >= 48) atomicAdd(counts + 3, bad);
}

int main(int argc, char** argv) {
  const int seq = argc > 1 ? atoi(argv[1]) : 0, launches = argc > 2 ? atoi(argv[2]) : 400;
  unsigned *cnt, h[4] = {0, 0, 0, 0};
  float* out;
  hipMalloc(&out, (size_t)256 * 8 * 256 * 8);
  hipMalloc(&cnt, 16);
  hipMemset(cnt, 0, 16);
  hipStream_t st;
  hipStreamCreate(&st);
  const int blocks = 256 * 8, trips = 2000;  // 8 blocks of 4 waves per CU = 8 waves per SIMD; 2048 * 4 * 2000 = 16.4 M wave-executions per launch
  for (int l = 0; l < launches; ++l) {
    const unsigned seed = 0x1234567u * (l + 1);
    if (seq == 0) hipLaunchKernelGGL(pk_kernel<0>, dim3(blocks), dim3(256), 0, st, trips, seed, cnt, out);
    else if (seq == 1) hipLaunchKernelGGL(pk_kernel<1>, dim3(blocks), dim3(256), 0, st, trips, seed, cnt, out);
    else if (seq == 2) hipLaunchKernelGGL(pk_kernel<2>, dim3(blocks), dim3(256), 0, st, trips, seed, cnt, out);
    else if (seq == 3) hipLaunchKernelGGL(pk_kernel<3>, dim3(blocks), dim3(256), 0, st, trips, seed, cnt, out);
    else hipLaunchKernelGGL(pk_kernel<4>, dim3(blocks), dim3(256), 0, st, trips / 4, seed, cnt, out);
  }
  hipStreamSynchronize(st);
  hipMemcpy(h, cnt, 16, hipMemcpyDeviceToHost);
  printf("seq %d: %.1f M wave-executions; p1 wrong %u (of them = b c exactly: %u; in lanes 48..63: %u), p0 wrong %u\n", seq,
         (double)blocks * 4 * trips * launches / 1e6, h[0], h[1], h[3], h[2]);
  return h[0] || h[2] ? 1 : 0;
}
