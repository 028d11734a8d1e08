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


// Split-precision products: a stage on the 16-deep half-precision matrix instructions (16x the f32 MFMA rate) with every
// f32 operand value split in the wave's fragment registers and the significant cross products accumulated in f32.
//   BF = 6 (round 4): x = hi + mid + lo, three bf16 terms by truncation (both residual subtractions exact), six products
//           hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi on v_mfma_f32_32x32x16_bf16 (error ~2^-24 of |a||b|);
//           ~5.5 VALU per value.
//   BF = 2 (round 6): x' = 2^e x, hi = f16_rne(x'), lo = f16_rne(x' - hi) (22-24 significand bits), THREE products
//           lo*hi + hi*lo + hi*hi on v_mfma_f32_32x32x16_f16; 2 VALU per value (v_fma_mix{lo,hi}_f16 scales, subtracts,
//           rounds and packs in one instruction) + 0.5 for the range watch.  fp16 has 5 exponent bits, so the power-of-two
//           scale 2^e is DYNAMIC and owned by the wave: it keeps each operand's fragment maximum inside [2^3, 2^15) (target
//           [2^10, 2^11): every value down to 2^-12 of the fragment maximum keeps its full 22 bits, smaller ones an absolute
//           error of 2^-35 of it); when a stage leaves the band the wave picks a new exponent and rescales its accumulators
//           (v_ldexp_f32, wave-uniform branch: never taken on homogeneous data after the first stage).  The accumulators
//           are brought back to unit scale before the cross-wave reduction.  No scale crosses a kernel boundary.
// Selected at run time by MESM_GEMM_BF16X = 6 | 2 (0 = exact f32 MFMA); bench.py names the arithmetic in `dtype`.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
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
        const float q0 = r0 - __uint_as_float(m0 & 0xFFFF0000u), q1 = r1 - __uint_as_float(m1 & 0xFFFF0000u);
        lo[t][i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
      }
  }
};

template <int BF>
__device__ __forceinline__ f32x16 split_mma(const SplitFrag<BF>& a, const SplitFrag<BF>& b, f32x16 acc) {
#define MESM_BF(x) __builtin_bit_cast(bf16x8, x)
#pragma unroll
  for (int t = 0; t < 2; ++t) {  // smallest terms first
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.lo[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.lo[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid[t]), MESM_BF(b.mid[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.mid[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi[t]), MESM_BF(b.hi[t]), acc, 0, 0, 0);
  }
#undef MESM_BF
  return acc;
}

// ---- BF = 2: two fp16 terms -----------------------------------------------------------------------------------------
struct HalfFrag {
  u32x4 hi[2], lo[2];  // [f16 k-step of 16][4 dwords = 8 f16], the slot map of SplitFrag
  // s = 2^e (wave-uniform).  x * s and x * s - hi are exact in f32 (a power-of-two product; a residual of <= 13 bits), so
  // each v_fma_mix rounds ONCE, to fp16, to nearest even.
  __device__ __forceinline__ void make(const float (&v)[4][4], float s) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float x0 = v[2 * t + (i >> 1)][2 * (i & 1)], x1 = v[2 * t + (i >> 1)][2 * (i & 1) + 1];
        unsigned h, l;
        asm("v_fma_mixlo_f16 %0, %1, %2, 0" : "=v"(h) : "v"(x0), "v"(s));
        asm("v_fma_mixhi_f16 %0, %1, %2, 0" : "+v"(h) : "v"(x1), "v"(s));
        asm("v_fma_mixlo_f16 %0, %1, %2, -%3 op_sel:[0,0,0] op_sel_hi:[0,0,1]" : "=v"(l) : "v"(x0), "v"(s), "v"(h));
        asm("v_fma_mixhi_f16 %0, %1, %2, -%3 op_sel:[0,0,1] op_sel_hi:[0,0,1]" : "+v"(l) : "v"(x1), "v"(s), "v"(h));
        hi[t][i] = h;
        lo[t][i] = l;
      }
  }
};

__device__ __forceinline__ f32x16 half_mma(const HalfFrag& a, const HalfFrag& b, f32x16 acc) {
#define MESM_HF(x) __builtin_bit_cast(f16x8, x)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(MESM_HF(a.lo[t]), MESM_HF(b.hi[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(MESM_HF(a.hi[t]), MESM_HF(b.lo[t]), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(MESM_HF(a.hi[t]), MESM_HF(b.hi[t]), acc, 0, 0, 0);
  }
#undef MESM_HF
  return acc;
}

// |x| maximum of a lane's 32 fragment values (NaNs drop out of v_max; an infinity stays).  v_max3_f32 with |.| source
// modifiers: one instruction per two values (the compiler's fmaxf canonicalises every operand first: 7 per four).
__device__ __forceinline__ float frag_amax(const float (&u)[4][4], const float (&w)[4][4]) {
  float m0 = 0.0f, m1 = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m0) : "v"(u[s][j]), "v"(u[s][j + 1]));
      asm("v_max3_f32 %0, |%1|, |%2|, %0" : "+v"(m1) : "v"(w[s][j]), "v"(w[s][j + 1]));
    }
  asm("v_max_f32 %0, %0, %1" : "+v"(m0) : "v"(m1));
  return m0;
}
__device__ __forceinline__ float frag_amax_finite(const float (&u)[4][4], const float (&w)[4][4]) {
  float m = 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float a = __builtin_fabsf(u[s][j]), b = __builtin_fabsf(w[s][j]);
      m = __builtin_fmaxf(m, a <= 3.4028234e38f ? a : 0.0f);
      m = __builtin_fmaxf(m, b <= 3.4028234e38f ? b : 0.0f);
    }
  return m;
}

// The wave's scale exponent of one operand and the band its fragment maxima must stay in (bit patterns of |x|).
struct HalfScale {
  int e;                 // values are multiplied by 2^e
  unsigned over, under;  // |x| bits >= over: 2^e |x| >= 2^15 (fp16 would overflow); no lane >= under: fragment maximum < 2^3
  __device__ __forceinline__ void set(int e_) {
    e = e_;
    over = (unsigned)(127 + 15 - e_) << 23;
    under = (unsigned)(127 + 3 - e_) << 23;
  }
  __device__ __forceinline__ float scale() const { return __uint_as_float((unsigned)(127 + e) << 23); }
  // wave-uniform: does this stage's fragment leave the band?  (an all-zero fragment does not)
  __device__ __forceinline__ bool leaves(float lane_amax) const {
    const unsigned u = __float_as_uint(lane_amax);
    const bool up = __builtin_amdgcn_ballot_w64(u >= over) != 0;
    const bool in = __builtin_amdgcn_ballot_w64(u >= under) != 0;
    const bool any = __builtin_amdgcn_ballot_w64(u != 0u) != 0;
    return up || (!in && any);
  }
  // new exponent from the wave's largest FINITE |x|: 2^e |x|max in [2^10, 2^11)
  __device__ __forceinline__ int pick(float lane_amax_finite) const {
    const float wm = wave_max(lane_amax_finite);
    if (wm == 0.0f) return e;
    int eb = (int)((__float_as_uint(wm) >> 23) & 0xFFu);
    eb = eb < 1 ? 1 : eb;
    int ne = 137 - eb;
    ne = ne > 110 ? 110 : (ne < -110 ? -110 : ne);
    return ne;
  }
};

// Both operands' scales of a wave and a bound on what its accumulators hold (in units of 2^-(ea + eb)): a stage adds at
// most 32 x 2^15 x 2^15 = 2^35 per element, so <= 2^43 between two re-picks of a <= 256-stage range.  Scaling DOWN (a stage
// with larger values arrived) is always safe; scaling UP (smaller values: keeps THEIR significand bits) is capped so that
// the accumulators stay below 2^100.  `repick` returns the exponent the accumulators have to be shifted by.
struct HalfScales {
  HalfScale a, b;
  int bexp;
  __device__ __forceinline__ void init() {
    a.set(0);
    b.set(0);
    bexp = -1000;
  }
  __device__ __forceinline__ int repick(float fa, float fb, bool empty) {
    int ea = a.pick(fa), eb = b.pick(fb);
    int delta = 0;
    if (!empty) {
      bexp = (bexp > 43 ? bexp : 43) + 1;
      const int room = 100 - bexp;
      int excess = (ea - a.e) + (eb - b.e) - room;
      if (excess > 0) {
        int ua = ea - a.e;
        ua = ua > 0 ? (ua < excess ? ua : excess) : 0;
        ea -= ua;
        excess -= ua;
        int ub = eb - b.e;
        ub = ub > 0 ? (ub < excess ? ub : excess) : 0;
        eb -= ub;
      }
      delta = (ea - a.e) + (eb - b.e);
      bexp += delta;
    }
    a.set(ea);
    b.set(eb);
    return delta;
  }
};


}  // namespace
