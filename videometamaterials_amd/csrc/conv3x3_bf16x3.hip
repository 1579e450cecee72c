// 3x3 (stride 1, zero padding 1) per-frame convolution with an LDS-resident halo patch, split-bf16 MFMA ("bf16x3"), gfx950.
//
// The generic implicit GEMM (igemm_bf16x3.hip) gathers the A operand once per tap: 9 loads, 9 fp32->bf16 splits and 9
// GroupNorm+SiLU evaluations per activation element, which leaves the matrix cores waiting on L2 latency.  Here a workgroup
// owns 128 consecutive pixels of one frame; for each chunk of 32 input channels it stages the pixels [p0-(W+1), p0+128+(W+1))
// ONCE (transform + hi/lo split fused into the staging), and the nine taps are nine row-shifted views of that patch:
// A fragments are ds_read_b128 at patch row (r + (W+1) + dh*W + dw), image borders are applied as per-lane tap masks.
// Weights (fmt-1 pre-split bf16, [Cout][Kpad]) stream through a double-buffered LDS tile, one (tap, channel chunk) at a time.
// The next patch chunk is prefetched into registers while the 9 x 12..24 MFMAs of the current one run.
// Traffic per output tile: each input element is read ~1.0x (+ halo) instead of 9x; LDS rows keep the 80-byte pitch of
// igemm_bf16x3.hip (conflict-free 16-byte fragment reads).
#include "igemm_common.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

constexpr int CK = 32;    // channels per chunk
constexpr int CROW = 40;  // LDS row pitch in bf16 (80 bytes)
constexpr int CM = 128;   // pixels per workgroup
constexpr int MAXP = 12;  // patch float4 items per thread (patch rows <= 384)

__device__ __forceinline__ void split2c(float x0, float x1, unsigned& hi, unsigned& lo) {
  const f32x2 v = {x0, x1};
  hi = __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
  const f32x2 r = {x0 - __uint_as_float(hi << 16), x1 - __uint_as_float(hi & 0xFFFF0000u)};
  lo = __builtin_bit_cast(unsigned, __builtin_convertvector(r, bf16x2));
}

template <int BN>
__global__ __launch_bounds__(256) void conv3x3_x3_kernel(const vmm_conv_desc p, int Kpad, int n_tiles, int tiles_per_frame, int PR) {
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
  const int mtile = blockIdx.x / n_tiles;
  const int n0 = (blockIdx.x % n_tiles) * BN;
  const int img = mtile / tiles_per_frame;
  const int pix0 = (mtile % tiles_per_frame) * CM;
  const int Cin = p.C1 + p.C2;
  const int nchunks = Cin / CK;
  const int halo = W + 1;

  // tap masks of this lane's output pixels (bit t = kh*3+kw set when the tap reads inside the image)
  unsigned tapmask[MT];
  int prow[MT];
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int local = wm * 64 + i * 32 + lrow;
    const int pix = pix0 + local;
    prow[i] = local + halo;
    unsigned msk = 0;
    if (pix < HW) {
      const int h = pix / W, w = pix - h * W;
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int hh = h + t / 3 - 1, ww = w + t % 3 - 1;
        if (hh >= 0 && hh < H && ww >= 0 && ww < W) msk |= 1u << t;
      }
    }
    tapmask[i] = msk;
  }

  // patch staging roles
  const int n_items = PR * 8;
  f32x4 preg[MAXP];
  auto load_patch = [&](int cc) {
    const int c0 = cc * CK;
    const bool src1 = c0 < p.C1;
    const float* src = src1 ? p.a1 : p.a2;
    const int ld = src1 ? p.lda1 : p.lda2;
    const int cb = src1 ? c0 : c0 - p.C1;
    const bool xform = src1 && p.a_mode == 1;
    const float* cf0 = xform ? p.a_coef + ((long long)(img / p.a_imgs_per_sample) * p.C1 + cb) * 2 : nullptr;
#pragma unroll
    for (int ps = 0; ps < MAXP; ++ps) {
      const int e = tid + ps * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (e < n_items) {
        const int r = e >> 3, k4 = e & 7;
        const int pix = pix0 - halo + r;
        if (pix >= 0 && pix < HW) {
          v = *reinterpret_cast<const f32x4*>(src + ((long long)img * HW + pix) * ld + cb + k4 * 4);
          if (xform) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(cf0 + k4 * 8);
            const f32x4 b = *reinterpret_cast<const f32x4*>(cf0 + k4 * 8 + 4);
            v.x = igemm::silu_fast(v.x * a.x + a.y);
            v.y = igemm::silu_fast(v.y * a.z + a.w);
            v.z = igemm::silu_fast(v.z * b.x + b.y);
            v.w = igemm::silu_fast(v.w * b.z + b.w);
          }
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
        const int r = e >> 3, k4 = e & 7;
        unsigned h0, l0, h1, l1;
        split2c(preg[ps].x, preg[ps].y, h0, l0);
        split2c(preg[ps].z, preg[ps].w, h1, l1);
        *reinterpret_cast<uint2*>(&Ph[r * CROW + k4 * 4]) = make_uint2(h0, h1);
        *reinterpret_cast<uint2*>(&Pl[r * CROW + k4 * 4]) = make_uint2(l0, l1);
      }
    }
  };
  // weight tile roles
  const unsigned short* wbase = reinterpret_cast<const unsigned short*>(p.w);
  const long long plane = (long long)p.Cout * Kpad;
  uint4 breg[B_PASSES];
  auto load_b = [&](int cc, int tap) {
    const int koff = tap * Cin + cc * CK;
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int e = tid + ps * 256;
      uint4 v = {0u, 0u, 0u, 0u};
      if (e < B_ITEMS) {
        const int seg = e & 3, n = (e >> 2) % BN, pl = e / (4 * BN);
        if (n0 + n < p.Cout) v = *reinterpret_cast<const uint4*>(wbase + pl * plane + (long long)(n0 + n) * Kpad + koff + seg * 8);
      }
      breg[ps] = v;
    }
  };
  auto store_b = [&](int buf) {
#pragma unroll
    for (int ps = 0; ps < B_PASSES; ++ps) {
      const int e = tid + ps * 256;
      if (e < B_ITEMS) {
        const int seg = e & 3, n = (e >> 2) % BN, pl = e / (4 * BN);
        unsigned short* dst = (pl ? Bl : Bh) + (buf * BN + n) * CROW + seg * 8;
        *reinterpret_cast<uint4*>(dst) = breg[ps];
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

  load_patch(0);
  load_b(0, 0);
  store_patch();
  store_b(0);
  __syncthreads();
  int bbuf = 0;
  for (int cc = 0; cc < nchunks; ++cc) {
    const bool more = cc + 1 < nchunks;
    if (more) load_patch(cc + 1);  // in flight during the nine taps
#pragma unroll 1
    for (int tap = 0; tap < 9; ++tap) {
      const bool last = tap == 8;
      if (!last) load_b(cc, tap + 1);
      else if (more) load_b(cc + 1, 0);
      const int toff = (tap / 3 - 1) * W + (tap % 3 - 1);
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
          for (int j = 0; j < NT; ++j) {
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[i], bh[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[i], bh[j], acc[i][j], 0, 0, 0);
          }
      }
      if (!last || more) store_b(bbuf ^ 1);
      if (last && more) {
        __syncthreads();  // every wave is done reading the patch of chunk cc
        store_patch();
      }
      __syncthreads();
      bbuf ^= 1;
    }
  }

  // epilogue: rows of this tile are pixels pix0 .. pix0+127 of frame img
#pragma unroll
  for (int i = 0; i < MT; ++i) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int pix = pix0 + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lk;
      if (pix >= HW) continue;
      const long long orow = (long long)img * HW + pix;
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = n0 + wn * TN + j * 32 + lrow;
        if (col < p.Cout) {
          float v = acc[i][j][r];
          if (p.bias) v += p.bias[col];
          if (p.res) v += p.res[orow * p.ldres + col];
          p.out[orow * p.ldo + col] = v;
        }
      }
    }
  }
}

template <int BN>
int launch_c3(const vmm_conv_desc& d, int Kpad, hipStream_t s) {
  const int HW = d.Hin * d.Win;
  const int tpf = cdiv(HW, CM);
  const int nt = cdiv(d.Cout, BN);
  const int PR = CM + 2 * (d.Win + 1);
  const size_t shm = sizeof(unsigned short) * ((size_t)2 * PR * CROW + (size_t)4 * BN * CROW);
  static bool attr_set = false;
  if (!attr_set) {
    hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_x3_kernel<BN>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    attr_set = true;
  }
  hipLaunchKernelGGL((conv3x3_x3_kernel<BN>), dim3((unsigned)((long long)d.nimg * tpf * nt)), dim3(256), shm, s, d, Kpad, nt, tpf, PR);
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
  if (!shape_ok || !chan_ok || PR * 8 > MAXP * 256) return 1;
  if (d.a_mode == 1 && (!d.a_coef || d.a_imgs_per_sample <= 0)) return -3;
  if ((long long)d.nimg * d.Hin * d.Win >= (1LL << 31)) return -4;
  const int Ktot = 9 * (d.C1 + d.C2);
  const int Kpad = (Ktot + 31) / 32 * 32;
  hipStream_t s = (hipStream_t)stream;
  if (d.Cout >= 128) return launch_c3<128>(d, Kpad, s);
  return launch_c3<64>(d, Kpad, s);
}
