// Layout of the workspace the fused linear-attention forward (linattn_block.hip) leaves behind, shared with its recomputing backward
// (linattn_block_bwd.hip):  part [frame][split][head][LA_PART floats: max[32] | sum[32] | ctx^T[e][d]]  then  ctxfrag [frame][head][256 uint4].
#pragma once

constexpr int LA_PART = 64 + 32 * 32;

// position slices per frame of the forward's context pass: (number of slices, 32-pixel tiles per slice)
static inline int vmm_linattn_block_split(int frames, int HW, int* sps) {
  const int tiles = HW / 32;
  int ns = tiles / 4 < 768 / (frames > 0 ? frames : 1) ? tiles / 4 : 768 / (frames > 0 ? frames : 1);
  if (ns < 1) ns = 1;
  *sps = (tiles + ns - 1) / ns;
  return (tiles + *sps - 1) / *sps;
}
