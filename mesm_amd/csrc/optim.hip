// Optimizer tail of the reference train loop (train.py:68-72, runner.py:348-352) on the FLAT
// parameter / gradient buffers: global-norm gradient clipping (nn.utils.clip_grad_norm_) and the
// AdamW update (torch.optim.AdamW, amsgrad off) in two launches for all 273 tensors --
// SURVEY.md §8f row 1.  HBM-bound: the update reads p, g, m, v and writes p, m, v once
// (28 B per element, 13.7 M elements = 383 MB per step).
#include "common.hpp"

namespace {

constexpr int OPT_THREADS = 256;

// partial sums of squares of the flat gradient buffer, one per workgroup (plain stores);
// workgroup 0 also advances the device-side step counter
__global__ __launch_bounds__(OPT_THREADS) void sumsq_kernel(const float* __restrict__ g, int64_t n4,
                                                           float* __restrict__ partials,
                                                           int32_t* __restrict__ step) {
  __shared__ float sh[OPT_THREADS / 64];
  float a = 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    a += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) {
    partials[blockIdx.x] = sh[0] + sh[1] + sh[2] + sh[3];
    if (blockIdx.x == 0 && step) step[0] += 1;
  }
}

// total gradient norm from the partials (every workgroup sums them itself: <= 1024 floats)
__device__ __forceinline__ float total_norm(const float* __restrict__ partials, int np, float* sh) {
  float a = 0.0f;
  for (int i = threadIdx.x; i < np; i += OPT_THREADS) a += partials[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  const float t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return sqrtf(t);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_scale_kernel(float* __restrict__ g, int64_t n4,
                                                                const float* __restrict__ partials, int np,
                                                                float max_norm, float* __restrict__ norm_out) {
  __shared__ float sh[OPT_THREADS / 64];
  const float norm = total_norm(partials, np, sh);
  if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
  const float coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
  if (coef >= 1.0f) return;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
    float4 v = reinterpret_cast<float4*>(g)[i];
    v.x *= coef; v.y *= coef; v.z *= coef; v.w *= coef;
    reinterpret_cast<float4*>(g)[i] = v;
  }
}

// AdamW on groups of 4 elements; active4[i] == 0 skips group i (parameters that have no gradient this
// step are left untouched, exactly like torch.optim skips p.grad is None: no decay, no state update)
__global__ __launch_bounds__(OPT_THREADS) void adamw_kernel(
    float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
    const uint8_t* __restrict__ active4, int64_t n4, const float* __restrict__ partials, int np,
    float max_norm, const float* __restrict__ lr_p, float beta1, float beta2, float eps, float wd,
    const int32_t* __restrict__ step, float* __restrict__ norm_out) {
  __shared__ float sh[OPT_THREADS / 64];
  float coef = 1.0f;
  if (max_norm > 0.0f) {
    const float norm = total_norm(partials, np, sh);
    if (blockIdx.x == 0 && threadIdx.x == 0 && norm_out) norm_out[0] = norm;
    coef = fminf(max_norm / (norm + 1e-6f), 1.0f);
  }
  const float lr = lr_p[0];
  const float t = (float)step[0];
  const float bc1 = 1.0f - powf(beta1, t);
  const float bc2 = 1.0f - powf(beta2, t);
  const float step_size = lr / bc1;
  const float inv_sqrt_bc2 = 1.0f / sqrtf(bc2);
  const float decay = 1.0f - lr * wd;
  for (int64_t i = (int64_t)blockIdx.x * OPT_THREADS + threadIdx.x; i < n4; i += (int64_t)gridDim.x * OPT_THREADS) {
    if (active4 && !active4[i]) continue;
    float4 pv = reinterpret_cast<float4*>(p)[i];
    const float4 gv = reinterpret_cast<const float4*>(g)[i];
    float4 mv = reinterpret_cast<float4*>(m)[i];
    float4 vv = reinterpret_cast<float4*>(v)[i];
    float pe[4] = {pv.x, pv.y, pv.z, pv.w}, ge[4] = {gv.x, gv.y, gv.z, gv.w};
    float me[4] = {mv.x, mv.y, mv.z, mv.w}, ve[4] = {vv.x, vv.y, vv.z, vv.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float gg = ge[e] * coef;
      pe[e] *= decay;
      me[e] = beta1 * me[e] + (1.0f - beta1) * gg;
      ve[e] = beta2 * ve[e] + (1.0f - beta2) * gg * gg;
      const float denom = sqrtf(ve[e]) * inv_sqrt_bc2 + eps;
      pe[e] -= step_size * (me[e] / denom);
    }
    reinterpret_cast<float4*>(p)[i] = make_float4(pe[0], pe[1], pe[2], pe[3]);
    reinterpret_cast<float4*>(m)[i] = make_float4(me[0], me[1], me[2], me[3]);
    reinterpret_cast<float4*>(v)[i] = make_float4(ve[0], ve[1], ve[2], ve[3]);
  }
}

inline int opt_blocks(int64_t n4) {
  int64_t b = (n4 + OPT_THREADS - 1) / OPT_THREADS;
  if (b > 1024) b = 1024;
  if (b < 1) b = 1;
  return (int)b;
}

}  // namespace

extern "C" int mesm_grad_sumsq(const float* g, int64_t n, float* partials, int32_t* np_out, int32_t* step,
                               void* stream) {
  if (!g || !partials || !np_out || n <= 0 || (n & 3) || ((uintptr_t)g & 15)) return MESM_EINVAL;
  const int nb = opt_blocks(n / 4);
  *np_out = nb;
  hipLaunchKernelGGL(sumsq_kernel, dim3(nb), dim3(OPT_THREADS), 0, (hipStream_t)stream, g, n / 4, partials, step);
  return mesm_launch_status();
}

extern "C" int mesm_clip_grad(float* g, int64_t n, const float* partials, int32_t np, float max_norm,
                              float* norm_out, void* stream) {
  if (!g || !partials || n <= 0 || (n & 3) || np <= 0 || np > 1024 || max_norm <= 0.f) return MESM_EINVAL;
  hipLaunchKernelGGL(clip_scale_kernel, dim3(opt_blocks(n / 4)), dim3(OPT_THREADS), 0, (hipStream_t)stream, g,
                     n / 4, partials, np, max_norm, norm_out);
  return mesm_launch_status();
}

extern "C" int mesm_adamw_step(float* p, const float* g, float* m, float* v, const uint8_t* active4, int64_t n,
                               const float* partials, int32_t np, float max_norm, const float* lr,
                               float beta1, float beta2, float eps, float weight_decay, const int32_t* step,
                               float* norm_out, void* stream) {
  if (!p || !g || !m || !v || !lr || !step || n <= 0 || (n & 3)) return MESM_EINVAL;
  if (max_norm > 0.f && (!partials || np <= 0 || np > 1024)) return MESM_EINVAL;
  if ((((uintptr_t)p | (uintptr_t)g | (uintptr_t)m | (uintptr_t)v) & 15) != 0) return MESM_EALIGN;
  hipLaunchKernelGGL(adamw_kernel, dim3(opt_blocks(n / 4)), dim3(OPT_THREADS), 0, (hipStream_t)stream, p, g, m, v,
                     active4, n / 4, partials, np, max_norm, lr, beta1, beta2, eps, weight_decay, step, norm_out);
  return mesm_launch_status();
}
