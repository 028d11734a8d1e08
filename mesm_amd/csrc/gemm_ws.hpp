// Device pieces shared by the k-split LDS-DMA GEMM kernels (gemm.hip: wstage / wstage64 and their grouped launches;
// gemm_pk.hip: the persistent split-bf16 kernel): wave-private LDS-DMA staging of 32 x 32 operand slabs, the exact
// three-term bf16 split of f32 fragments, the argument block of a grouped launch and the carried slope reductions.
#pragma once
#include "gemm_common.hpp"

namespace {

// The reduction of the partials does not get a launch of its own (14 per step, ~2.3 us each in the graph): it is
// CARRIED by the next GEMM launch on the stream -- wave 0 of that launch's first workgroup sums the partials of up to four
// pending reductions before its own work (stream order: the producing kernel has finished).  Host-side queue; what no
// launch has picked up is reduced by mesm_gemm_flush_side (called at the end of every backward block).
struct SideRed {
  const float* ws[4];
  float* dst[4];
  int n[4];
  int count;
};

__device__ __forceinline__ void side_reduce(const SideRed& sr) {
  if (sr.count == 0 || blockIdx.x != 0 || blockIdx.y != 0 || blockIdx.z != 0 || (threadIdx.x >> 6) != 0) return;
  const int lane = threadIdx.x & 63;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    if (i < sr.count) {
      float a = 0.0f;
      for (int k = lane; k < sr.n[i]; k += 64) a += sr.ws[i][k];
      a = wave_sum(a);
      if (lane == 0) sr.dst[i][0] += a;
    }
  }
}


constexpr int WS_SLAB = 32 * 32;  // floats

template <int LAYOUT>
__device__ __forceinline__ void ws_issue(const float* __restrict__ base, int64_t ld, int o0, int extent,
                                         int kb, int k1, float* slab, int lane) {
  const int sr = lane >> 3, pos = lane & 7;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const float* g;
    if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
      const int r = 8 * q + sr;
      int row = o0 + r;
      row = row < extent ? row : extent - 1;
      const int c = pos ^ ((r >> 1) & 7);
      int k = kb + 4 * c;
      k = k < k1 ? k : k1 - 4;  // tail: clamped garbage, zeroed at fragment read
      g = base + (int64_t)row * ld + k;
    } else {
      int k = kb + 8 * q + (((sr & 1) << 2) | (sr >> 1));
      k = k < k1 ? k : k1 - 1;
      int o = o0 + 4 * pos;
      const int e4 = (extent + 3) & ~3;  // a chunk may straddle the extent (into the next reduce row: gemm_kmain)
      o = o + 4 <= e4 ? o : e4 - 4;
      g = base + (int64_t)k * ld + o;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                     (__attribute__((address_space(3))) void*)(slab + q * 256), 16, 0, 0);
  }
}

// fragment of one stage: v[s][j] = operand[outer = lane & 31][k = kb + 8s + 4h + j]
template <int LAYOUT>
__device__ __forceinline__ void ws_read(const float* slab, int li, int h, float (&v)[4][4]) {
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
      const int pos = (2 * s + h) ^ ((li >> 1) & 7);
      const float4 x = *reinterpret_cast<const float4*>(slab + li * 32 + pos * 4);
      v[s][0] = x.x; v[s][1] = x.y; v[s][2] = x.z; v[s][3] = x.w;
    } else {
#pragma unroll
      for (int j = 0; j < 4; ++j) v[s][j] = slab[(8 * s + 2 * j + h) * 32 + li];
    }
  }
}


#ifndef MESM_GROUP_MAX
#define MESM_GROUP_MAX 8  // (12: the larger kernel-argument segment costs every grouped launch more than the 4 merged launches save, 4.923 vs 4.896 ms)
#endif
constexpr int GROUP_MAX = MESM_GROUP_MAX;
struct GroupArgs {
  MesmGemmArgs p[GROUP_MAX];
  int start[GROUP_MAX + 1];  // first workgroup of every problem
  int n;
};


// Split-precision products (BF = 6: the default for the large products since round 4; BF = 3 experimental): a stage
// on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate) with every f32 operand value split exactly into bf16 terms
// x = hi + mid + lo (8 mantissa bits each; hi and mid by truncation, so x - hi and x - hi - mid are exact f32
// subtractions) and the significant cross products accumulated in f32:
//   BF = 6: hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi   (error ~2^-24 of |a||b|, the f32 product's own)
//   BF = 3: hi*hi + hi*mid + mid*hi                              (error ~2^-16)
// Selected at run time by MESM_GEMM_BF16X=6|3 (bench.py reports both under roofline.experimental with the parity
// suite's verdict at unchanged tolerances); the split is done on the fragment registers (~6 VALU per value).
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

template <int BF>
struct SplitFrag {
  u32x4 hi[2], mid[2], lo[2];  // [bf16 k-step of 16][4 dwords = 8 bf16]
  // v[s][j] = operand[outer][kb + 8 s + 4 h + j]: k-step t takes s = 2t, 2t + 1 (the same slot map on both operands)
  __device__ __forceinline__ void make(const float (&v)[4][4]) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * t + (i >> 1)][2 * (i & 1)], x1 = v[2 * t + (i >> 1)][2 * (i & 1) + 1];
        const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
        hi[t][i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
        const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
        const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
        mid[t][i] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
        if (BF == 6) {
          const float q0 = r0 - __uint_as_float(m0 & 0xFFFF0000u), q1 = r1 - __uint_as_float(m1 & 0xFFFF0000u);
          lo[t][i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
        }
      }
  }
};

template <int BF>
__device__ __forceinline__ f32x16 split_mma(const SplitFrag<BF>& a, const SplitFrag<BF>& b, f32x16 acc) {
#define MESM_BF(x) __builtin_bit_cast(bf16x8, x)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (BF == 6) {  // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.lo[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.lo[t]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid[t]), MESM_BF(b.mid[t]), acc, 0, 0, 0);
    }
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.mid[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
  }
#undef MESM_BF
  return acc;
}


}  // namespace
