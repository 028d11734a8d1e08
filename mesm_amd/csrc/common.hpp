// Shared device helpers for the gfx950 kernels (wave64, CDNA4).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "mesm_gfx950.h"

#define MESM_WAVE 64

static inline int mesm_launch_status() {
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? MESM_OK : (MESM_ELAUNCH - (int)e);
}

// Counter-based keep/drop decision shared by every dropout site in the library:
// lowbias32 finaliser over (index * golden + seed).  keep iff hash >= p * 2^32.
__device__ __forceinline__ uint32_t mesm_hash32(uint32_t idx, uint32_t seed) {
  uint32_t x = idx * 0x9E3779B9u + seed;
  x ^= x >> 16;
  x *= 0x7feb352du;
  x ^= x >> 15;
  x *= 0x846ca68bu;
  x ^= x >> 16;
  return x;
}

__host__ __device__ __forceinline__ uint32_t mesm_drop_threshold(float p) {
  double t = (double)p * 4294967296.0;
  if (t < 0.0) t = 0.0;
  if (t > 4294967295.0) t = 4294967295.0;
  return (uint32_t)t;
}

__device__ __forceinline__ float mesm_dropout_apply(float x, uint32_t idx, uint32_t seed,
                                                    uint32_t thresh, float inv_keep) {
  return mesm_hash32(idx, seed) >= thresh ? x * inv_keep : 0.0f;
}

__device__ __forceinline__ float mesm_act(float x, int act, float slope) {
  if (act == MESM_ACT_RELU) return x > 0.0f ? x : 0.0f;
  if (act == MESM_ACT_PRELU) return x > 0.0f ? x : slope * x;
  return x;
}

// WRITE-THROUGH output stores (round 6).  A plain store leaves its line dirty in the XCD's L2 until the kernel boundary's
// release writes every dirty line back -- time at the END of the launch, behind the last tile, that nothing overlaps; a
// store with sc1 leaves L2 when it is issued, while other workgroups still compute (tools/probe/store_tail.hip: a chain of
// launches that each write 5 / 20 / 39 MB: 3.82 / 6.48 / 9.11 -> 3.47 / 5.35 / 7.98 us per launch; 4-byte sc1 stores of a
// half wave's 128 contiguous bytes cost what 16-byte ones do).  The price: the line is dropped from that L2, so a consumer
// workgroup that lands on the SAME XCD reads it from the Infinity Cache like the other seven XCDs' do (+0.2-0.9 us in the
// probe's reader pass below 20 MB).  In the step the GEMM epilogues alone: 3.370 -> 3.332 ms, with every size of output
// (thresholds at 1 M / 4 M elements gave less / nothing: profiles/r6q/ab_sc1.txt).  -DMESM_WT_STORES=0: plain stores.
#ifndef MESM_WT_STORES
#define MESM_WT_STORES 1
#endif
typedef float mesm_f32x4 __attribute__((ext_vector_type(4)));
typedef float mesm_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ void mesm_store_wt(float* p, float v) {
#if MESM_WT_STORES
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // global_store_dword ... sc1
#else
  *p = v;
#endif
}
__device__ __forceinline__ void mesm_store_wt2(float* p, float a, float b) {
#if MESM_WT_STORES
  const mesm_f32x2 v = {a, b};
  asm volatile("global_store_dwordx2 %0, %1, off sc1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<float2*>(p) = make_float2(a, b);
#endif
}
__device__ __forceinline__ void mesm_store_wt4(float* p, float a, float b, float c, float d) {
#if MESM_WT_STORES
  const mesm_f32x4 v = {a, b, c, d};
  // (s_nop: a vector-memory store of more than 64 bits reads its data registers late; a VALU write to them within the next
  // wait states corrupts the store -- the compiler pads its own stores, not inline assembly: LayerNorm twin outputs differed)
  asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
#else
  *reinterpret_cast<float4*>(p) = make_float4(a, b, c, d);
#endif
}

__device__ __forceinline__ void mesm_store_wt4(float* p, const float4& v) { mesm_store_wt4(p, v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void mesm_store_wt16(void* p, const uint4& v) {  // 16 bytes of any type
  mesm_store_wt4(reinterpret_cast<float*>(p), __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z),
                 __uint_as_float(v.w));
}

// Wave-wide reductions on the DPP path (gfx9 row operations): four in-row butterfly steps
// (quad_perm xor 1, xor 2, row_half_mirror, row_mirror), two row broadcasts (row_bcast:15 into rows
// 1 and 3, row_bcast:31 into rows 2 and 3) and one v_readlane of lane 63 -- 7 VALU-rate instructions.
// `__shfl_xor` compiles to ds_bpermute_b32 on gfx950 (an LDS-crossbar round trip per step, 6 steps per
// reduction): the per-row softmax / LayerNorm statistics were latency chains of those.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float mesm_dpp(float old, float v) {
  return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, 0xF, false));
}

__device__ __forceinline__ float wave_sum(float v) {
  v += mesm_dpp<0xB1, 0xF>(0.0f, v);   // quad_perm [1,0,3,2]
  v += mesm_dpp<0x4E, 0xF>(0.0f, v);   // quad_perm [2,3,0,1]
  v += mesm_dpp<0x141, 0xF>(0.0f, v);  // row_half_mirror
  v += mesm_dpp<0x140, 0xF>(0.0f, v);  // row_mirror: every lane holds its row-of-16 sum
  v += mesm_dpp<0x142, 0xA>(0.0f, v);  // row_bcast:15 -> rows 1, 3
  v += mesm_dpp<0x143, 0xC>(0.0f, v);  // row_bcast:31 -> rows 2, 3: row 3 holds the total
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

__device__ __forceinline__ float wave_max(float v) {
  v = fmaxf(v, mesm_dpp<0xB1, 0xF>(v, v));
  v = fmaxf(v, mesm_dpp<0x4E, 0xF>(v, v));
  v = fmaxf(v, mesm_dpp<0x141, 0xF>(v, v));
  v = fmaxf(v, mesm_dpp<0x140, 0xF>(v, v));
  v = fmaxf(v, mesm_dpp<0x142, 0xA>(v, v));
  v = fmaxf(v, mesm_dpp<0x143, 0xC>(v, v));
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), 63));
}

// v + the value held by lane (lane ^ 32) / (lane ^ 16): gfx950's v_permlane{32,16}_swap exchanges the
// odd half (odd rows) of one register with the even half (even rows) of another in one VALU op.
__device__ __forceinline__ float add_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ float add_xor16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// Sum over the 64 / W lane groups {lane % W} (every lane of a group ends with the group's total):
// the lanes that differ only in bits >= log2(W).  In-row steps are DPP row rotations.
template <int W>
__device__ __forceinline__ float sum_across_groups(float v) {
  static_assert(W == 4 || W == 8 || W == 16 || W == 32 || W == 64, "group width");
  if (W <= 4) v += mesm_dpp<0x124, 0xF>(0.0f, v);  // row_ror:4
  if (W <= 8) v += mesm_dpp<0x128, 0xF>(0.0f, v);  // row_ror:8
  if (W <= 16) v = add_xor16(v);
  if (W <= 32) v = add_xor32(v);
  return v;
}

// Sum within aligned groups of N consecutive lanes (N in {2, 4, 8, 16}); all lanes get the total.
template <int N>
__device__ __forceinline__ float sum_within(float v) {
  static_assert(N == 1 || N == 2 || N == 4 || N == 8 || N == 16, "group size");
  if (N >= 2) v += mesm_dpp<0xB1, 0xF>(0.0f, v);
  if (N >= 4) v += mesm_dpp<0x4E, 0xF>(0.0f, v);
  if (N >= 8) v += mesm_dpp<0x141, 0xF>(0.0f, v);
  if (N >= 16) v += mesm_dpp<0x140, 0xF>(0.0f, v);
  return v;
}

// Row of the key-padding / query-padding masks that batch row b, head hd sees under the T2V mask quirk (SURVEY Q1:
// the reference repeats its (N, Lq, Lk) mask head-major but nn.MultiheadAttention indexes it batch-major):
//   b' = (b * H + hd) mod N   inside the group of `stride` stacked rows that b belongs to.
// stride = mask_group (0: the whole batch); N = *mask_mod when given (a device scalar: the VALID pairs of a batch padded
// to a captured capacity, graphed.py), else the stride.  Rows at or beyond N (padding pairs) see their own masks.
__device__ __forceinline__ int mesm_quirk_row(const MesmAttnArgs& p, int b, int hd) {
  const int stride = p.mask_group > 0 ? p.mask_group : p.B;
  const int mod = p.mask_mod ? *p.mask_mod : stride;
  const int bl = b % stride;
  return (b - bl) + (bl < mod ? (bl * p.H + hd) % mod : bl);
}
