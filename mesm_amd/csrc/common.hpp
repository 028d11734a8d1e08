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

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
