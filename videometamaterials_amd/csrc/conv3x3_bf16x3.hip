// 3x3 (stride 1, zero padding 1) per-frame convolution with an LDS-resident halo patch, split-bf16 MFMA ("bf16x3"), gfx950.
//
// The generic implicit GEMM (igemm_bf16x3.hip) gathers the A operand once per tap: 9 loads, 9 fp32->bf16 splits and 9
// GroupNorm+SiLU evaluations per activation element, which leaves the matrix cores waiting on L2 latency.  Here a workgroup
// owns 128 consecutive rows (pixels in frame-major order; a tile may straddle frames); for each chunk of 32 input channels it
// stages the rows [m0-(W+1), m0+128+(W+1)) ONCE (transform + hi/lo split fused into the staging), and the nine taps are nine
// row-shifted views of that patch: A fragments are ds_read_b128 at patch row (r + (W+1) + dh*W + dw).  Image borders -- and
// therefore also every read that would cross into a neighbouring frame -- are applied as per-lane tap masks.
// Weights (fmt-1 pre-split bf16, [Cout][Kpad]) stream through a double-buffered LDS tile, one (tap, channel chunk) at a time,
// prefetched two taps ahead in registers.  The next patch chunk is prefetched into registers while the 9 x 12..24 MFMAs of the
// current one run.  Layers with few rows (12x12 level) split the channel chunks over blockIdx.y and combine with fp32 atomics.
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CK = 32;    // channels per chunk
constexpr int CROW = 40;  // LDS row pitch in bf16 (80 bytes)
constexpr int CM = 128;   // rows per workgroup

__device__ __forceinline__ void split2c(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int BN, int MAXP>
__global__ __launch_bounds__(256) void conv3x3_x3_kernel(const vmm_conv_desc p, int Kpad, int n_tiles, int PR, int ksplit) {
  constexpr int TN = BN / 2, NT = TN / 32, MT = 2;  // 2 x 2 waves, wave tile 64 x TN
  constexpr int B_ITEMS = BN * 8, B_PASSES = (B_ITEMS + 255) / 256;
  extern __shared__ __attribute__((aligned(16))) unsigned short smem[];
  unsigned short* Ph = smem;
  unsigned short* Pl = Ph + PR * CROW;
  unsigned short* Bh = Pl + PR * CROW;      // [2][BN][CROW]
  unsigned short* Bl = Bh + 2 * BN * CROW;  // [2][BN][CROW]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int lrow = lane & 31, lk = lane >> 5;
  const int W = p.Win, H = p.Hin, HW = H * W;
  const int M = p.nimg * HW;
  const int m0 = (blockIdx.x / n_tiles) * CM;
  const int n0 = (blockIdx.x % n_tiles) * BN;
  const int Cin = p.C1 + p.C2;
  const int nch_all = Cin / CK;
  const int c_begin = (int)((long long)blockIdx.y * nch_all / ksplit), c_end = (int)((long long)(blockIdx.y + 1) * nch_all / ksplit);
  const int halo = W + 1;

  // tap masks of this lane's output pixels (bit t = kh*3+kw set when the tap reads inside the image of the pixel's own frame)
  unsigned tapmask[MT];
  int prow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int local = wm * 64 + i * 32 + lrow;
    const int m = m0 + local;
    prow[i] = local + halo;
    unsigned msk = 0;
    if (m < M) {
      const int pix = m % HW;
      const int h = pix / W, w = pix - h * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) msk |= 1u << t;
      }
    }
    tapmask[i] = msk;
  }

  // patch staging roles: item e = tid + ps*256 -> patch row e>>3 (global row m0 - halo + row), float4 e&7 of the channel chunk
  const int n_items = PR * 8;
  const int k4 = tid & 7;
  f32x4 preg[MAXP];
  auto load_patch = [&](int cc) {
    const int c0 = cc * CK;
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = (src1 ? c0 : c0 - p.C1) + k4 * 4;
    const bool xform = src1 && p.a_mode == 1;
    const int rows_per_sample = HW * p.a_imgs_per_sample;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int e = tid + ps * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      const int g = m0 - halo + (e >> 3);
      if (e < n_items && g >= 0 && g < M) {
        v = *reinterpret_cast<const f32x4*>(src + (long long)g * ld + cb);
        if (xform) {
          const float* cf = p.a_coef + ((long long)(g / rows_per_sample) * p.C1 + cb) * 2;
          const f32x4 a = *reinterpret_cast<const f32x4*>(cf);
          const f32x4 b = *reinterpret_cast<const f32x4*>(cf + 4);
          v.x = igemm::silu_fast(v.x * a.x + a.y);
          v.y = igemm::silu_fast(v.y * a.z + a.w);
          v.z = igemm::silu_fast(v.z * b.x + b.y);
          v.w = igemm::silu_fast(v.w * b.z + b.w);
        }
      }
      preg[ps] = v;
    }
  };
  auto store_patch = [&]() {
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int e = tid + ps * 256;
      if (e < n_items) {
        unsigned h0, l0, h1, l1;
        split2c(preg[ps].x, preg[ps].y, h0, l0);
        split2c(preg[ps].z, preg[ps].w, h1, l1);
        *reinterpret_cast<uint2*>(&Ph[(e >> 3) * CROW + k4 * 4]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(&Pl[(e >> 3) * CROW + k4 * 4]) = make_uint2(l0, l1);
      }
    }
  };
  // weight tile roles; tiles are numbered q = (cc - c_begin)*9 + tap
  const unsigned short* wbase = reinterpret_cast<const unsigned short*>(p.w);
  const long long plane = (long long)p.Cout * Kpad;
  const int n_tiles_b = (c_end - c_begin) * 9;
  auto load_b = [&](int q, uint4 (&reg)[B_PASSES]) {
    const int cc = c_begin + q / 9, tap = q % 9;
    const int koff = tap * Cin + cc * CK;
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int e = tid + ps * 256;
      uint4 v = {0u, 0u, 0u, 0u};
      if (e < B_ITEMS && q < n_tiles_b) {
        const int seg = e & 3, n = (e >> 2) % BN, pl = e / (4 * BN);
        if (n0 + n < p.Cout) v = *reinterpret_cast<const uint4*>(wbase + pl * plane + (long long)(n0 + n) * Kpad + koff + seg * 8);
      }
      reg[ps] = v;
    }
  };
  auto store_b = [&](int buf, const uint4 (&reg)[B_PASSES]) {
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int e = tid + ps * 256;
      if (e < B_ITEMS) {
        const int seg = e & 3, n = (e >> 2) % BN, pl = e / (4 * BN);
        unsigned short* dst = (pl ? Bl : Bh) + (buf * BN + n) * CROW + seg * 8;
        *reinterpret_cast<uint4*>(dst) = reg[ps];
      }
    }
  };

  f32x16 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  // one tap: MFMAs of B tile `q` (LDS buffer q&1) against the patch, while B tile q+2 is fetched and q+1 moves regs -> LDS
  auto tap_step = [&](int q, int tap, uint4 (&reg_next)[B_PASSES], uint4 (&reg_far)[B_PASSES], bool patch_swap) {
    load_b(q + 2, reg_far);
    const int toff = (tap / 3 - 1) * W + (tap % 3 - 1);
    const int bbuf = q & 1;
    const unsigned short* bh_t = Bh + bbuf * BN * CROW;
    const unsigned short* bl_t = Bl + bbuf * BN * CROW;
#pragma unroll
    for (int s = 0; s < CK / 16; ++s) {
      const int ko = s * 16 + lk * 8;
      bf16x8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        uint4 vh = *reinterpret_cast<const uint4*>(&Ph[(prow[i] + toff) * CROW + ko]);
        uint4 vl = *reinterpret_cast<const uint4*>(&Pl[(prow[i] + toff) * CROW + ko]);
        if (!((tapmask[i] >> tap) & 1u)) { vh = make_uint4(0u, 0u, 0u, 0u); vl = vh; }
        ah[i] = __builtin_bit_cast(bf16x8, vh);
        al[i] = __builtin_bit_cast(bf16x8, vl);
      }
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        bh[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&bh_t[(wn * TN + j * 32 + lrow) * CROW + ko]));
        bl[j] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4*>(&bl_t[(wn * TN + j * 32 + lrow) * CROW + ko]));
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
    }
    store_b(bbuf ^ 1, reg_next);  // tile q+1 (fetched one tap ago); the buffer was last read before the previous barrier
    if (patch_swap) {
      __syncthreads();  // every wave is done reading the patch of this channel chunk
      store_patch();
    }
    __syncthreads();
  };

  if (c_begin < c_end) {
    uint4 regA[B_PASSES], regB[B_PASSES];
    load_patch(c_begin);
    load_b(0, regA);
    load_b(1, regB);
    store_patch();
    store_b(0, regA);
    __syncthreads();
    // tile q is consumed from LDS[q&1]; tile q+1 sits in regB (even q) / regA (odd q); tile q+2 is fetched into the other set
    int q = 0;
    for (int cc = c_begin; cc < c_end; ++cc) {
      const bool more = cc + 1 < c_end;
      if (more) load_patch(cc + 1);  // in flight during the nine taps
#pragma unroll
      for (int tap = 0; tap < 9; ++tap) {
        const bool swap = (tap == 8) && more;
        if ((q & 1) == 0) tap_step(q, tap, regB, regA, swap);
        else tap_step(q, tap, regA, regB, swap);
        ++q;
      }
    }
  }

  // epilogue
  const bool first = blockIdx.y == 0;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = m0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (m >= M) continue;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + lrow;
        if (col < p.Cout) {
          float v = acc[i][j][r];
          if (first) {
            if (p.bias) v += p.bias[col];
            if (p.res) v += p.res[(long long)m * p.ldres + col];
          }
          float* o = p.out + (long long)m * p.ldo + col;
          if (ksplit > 1) atomicAdd(o, v); else *o = v;
        }
      }
    }
  }
}

template <int BN, int MAXP>
int launch_c3(const vmm_conv_desc& d, int Kpad, int PR, int ksplit, hipStream_t s) {
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  const int nt = cdiv(d.Cout, BN);
  const size_t shm = sizeof(unsigned short) * ((size_t)2 * PR * CROW + (size_t)4 * BN * CROW);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<BN, MAXP>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_x3_kernel<BN, MAXP>), dim3((unsigned)(cdiv(M, CM) * nt), ksplit), dim3(256), shm, s, d, Kpad, nt, PR, ksplit);
  VMM_LAUNCH_CHECK();
  return 0;
}

}  // namespace

// Returns 1 when the descriptor is outside this kernel's envelope (caller falls back to vmm_conv_igemm_bf16x3).
extern "C" int vmm_conv3x3_bf16x3(const vmm_conv_desc* dp, vmm_stream_t stream) {
  const vmm_conv_desc& d = *dp;
  const bool shape_ok = d.KH == 3 && d.KW == 3 && d.stride == 1 && d.off_h == -1 && d.off_w == -1 && d.sgn_h == 1 && d.sgn_w == 1 &&
                        d.Hv == d.Hin && d.Wv == d.Win && d.oscale == 1 && d.Hout == d.Hv && d.Wout == d.Wv && d.ooh == 0 && d.oow == 0 &&
                        d.rot_ncols == 0 && d.q_ncols == 0;
  const bool chan_ok = (d.C1 % CK == 0) && (d.C2 % CK == 0) && (d.Cout % 4 == 0) && d.Cout >= 64 && (d.lda1 & 3) == 0 && (!d.C2 || (d.lda2 & 3) == 0);
  const int PR = CM + 2 * (d.Win + 1);
  if (!shape_ok || !chan_ok || PR > 384) return 1;
  if (d.a_mode == 1 && (!d.a_coef || d.a_imgs_per_sample <= 0)) return -3;
  const long long M = (long long)d.nimg * d.Hin * d.Win;
  if (M >= (1LL << 31)) return -4;
  const int Ktot = 9 * (d.C1 + d.C2);
  const int Kpad = (Ktot + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  const int bn = d.Cout >= 128 ? 128 : 64;
  // few-row layers (12x12 level): split the channel chunks so that >= ~1000 workgroups are in flight; partial sums meet in fp32 atomics
  const long long blocks = (long long)cdiv(M, CM) * cdiv(d.Cout, bn);
  const int nch = (d.C1 + d.C2) / CK;
  int ksplit = 1;
  if (blocks < 768) ksplit = (int)max(1LL, min((long long)min(nch, 8), 2048 / max(blocks, 1LL)));
  if (ksplit > 1) {
    if (d.ldo != d.Cout) return 1;
    hipError_t e = hipMemsetAsync(d.out, 0, sizeof(float) * (size_t)M * d.Cout, s);
    if (e != hipSuccess) return (int)e;
  }
  const bool small = PR <= 8 * 32;  // W <= 63: 8 patch items per thread instead of 12 (fewer VGPRs)
  if (bn == 128) return small ? launch_c3<128, 8>(d, Kpad, PR, ksplit, s) : launch_c3<128, 12>(d, Kpad, PR, ksplit, s);
  return small ? launch_c3<64, 8>(d, Kpad, PR, ksplit, s) : launch_c3<64, 12>(d, Kpad, PR, ksplit, s);
}
