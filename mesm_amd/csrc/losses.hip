// Per-pair loss reductions of model/criterion.py and the matching cost + assignment of
// model/matcher.py as gfx950 kernels.  All are HBM/latency-bound and tiny except the
// masked-LM NLL over the vocabulary (N*Lw rows x C ~ 5k classes).
#include "common.hpp"
#include "loss_bodies.hpp"

namespace {

__device__ __forceinline__ float block_sum_256(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return t;
}

// ---------------------------------------------------------------- label-smoothed NLL
// one workgroup per row (criterion.py:291-306).  NPT > 0: the row (C <= 256 NPT classes) is read ONCE, NPT independent
// loads per thread in flight together, and both passes run on registers (the two-pass form below walked the row twice
// with one load in flight per thread: 19.5 -> 9.3 us at 1024 x 5003); NPT = 0: any C, two passes over the L2-resident row.
template <int NPT>
__global__ __launch_bounds__(256) void nll_fwd_kernel(
    const float* __restrict__ logit, const int64_t* __restrict__ label,
    const uint8_t* __restrict__ mask, float* __restrict__ row_loss,
    float* __restrict__ row_lse, uint8_t* __restrict__ correct, int C, float eps) {
  __shared__ float sh[8];
  __shared__ float shm[4];
  __shared__ int shi[4];
  const int64_t r = blockIdx.x;
  const float* x = logit + r * C;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  // pass 1: max + first argmax + plain sum
  float m = -INFINITY, s = 0.0f;
  int am = 0x7fffffff;
  float v[NPT > 0 ? NPT : 1];
  if (NPT > 0) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int c = threadIdx.x + k * 256;
      v[k] = c < C ? x[c] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int c = threadIdx.x + k * 256;
      if (c < C) {
        s += v[k];
        if (v[k] > m) { m = v[k]; am = c; }
      }
    }
  } else {
    for (int c = threadIdx.x; c < C; c += 256) {
      float t = x[c];
      s += t;
      if (t > m) { m = t; am = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o, 64);
    int a2 = __shfl_xor(am, o, 64);
    if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
  }
  if (lane == 0) { shm[wave] = m; shi[wave] = am; }
  __syncthreads();
  float M = shm[0];
  int AM = shi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (shm[w] > M || (shm[w] == M && shi[w] < AM)) { M = shm[w]; AM = shi[w]; }
  __syncthreads();
  const float total = block_sum_256(s, sh);
  // pass 2: sum exp
  float e = 0.0f;
  if (NPT > 0) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) e += __expf(v[k] - M);  // (exp(-inf) = 0 beyond C)
  } else {
    for (int c = threadIdx.x; c < C; c += 256) e += __expf(x[c] - M);
  }
  const float E = block_sum_256(e, sh);
  if (threadIdx.x == 0) {
    const float lse = M + __logf(E);
    const int64_t lab = label[r];
    const float nll = -(x[lab] - lse);
    const float smooth = -(total - (float)C * lse);
    float loss = (1.0f - eps) * nll + eps / (float)C * smooth;
    if (mask && mask[r] == 0) loss = 0.0f;
    row_loss[r] = loss;
    row_lse[r] = lse;
    if (correct) correct[r] = (AM == (int)lab) ? 1 : 0;
  }
}

__global__ __launch_bounds__(256) void nll_bwd_kernel(
    const float* __restrict__ logit, const int64_t* __restrict__ label,
    const float* __restrict__ row_lse, const float* __restrict__ row_grad,
    float* __restrict__ dlogit, int C, float eps) {
  nll_bwd_body(logit, label, row_lse, row_grad[blockIdx.x], dlogit, C, eps, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---------------------------------------------------------------- saliency losses
// one workgroup (deterministic sum) of NW waves, a wave per pair (16 waves; 8 with the 20-element arrays, whose
// registers do not fit the 128-VGPR budget of a 1,024-thread workgroup)
template <int NE, int NW>
__global__ __launch_bounds__(64 * NW) void saliency_fwd_kernel(
    const float* __restrict__ s_pos, const float* __restrict__ s_neg,
    const double* __restrict__ label, const uint8_t* __restrict__ vmask,
    const int64_t* __restrict__ pos_idx, const int64_t* __restrict__ neg_idx, int N, int L, int P,
    float rank_coef, float margin, float* __restrict__ out_loss, const int32_t* __restrict__ n_valid) {
  if (n_valid) N = *n_valid;  // pairs [n_valid, N) are padding of a captured capacity: not in the mean
  __shared__ float part[NW];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float acc = 0.0f;  // lane 0 of each wave accumulates its rows
  for (int n = wave; n < N; n += NW) {
    const float* sp = s_pos + (int64_t)n * L;
    const float* sn = s_neg + (int64_t)n * L;
    const uint8_t* vm = vmask + (int64_t)n * L;
    SalRow<NE> R;
    sal_row_stats(sp, sn, label + (int64_t)n * L, vm, L, lane, R);
    // neg-pair term: sum_l -log(1 - sigmoid(s_neg)) * vmask
    float np = 0.0f;
    for (int l = lane; l < L; l += 64) {
      float sg = 1.0f / (1.0f + __expf(-sn[l]));
      np += -__logf(1.0f - sg) * (vm[l] ? 1.0f : 0.0f);
    }
    np = wave_sum(np);
    float trip = 0.0f;
    if (pos_idx && lane < P) {
      float ps = sp[pos_idx[(int64_t)n * P + lane]];
      float ns = sp[neg_idx[(int64_t)n * P + lane]];
      float t = margin + ns - ps;
      trip = t > 0.0f ? t : 0.0f;
    }
    trip = wave_sum(trip);
    acc += R.rank / ((float)N * rank_coef) + np / (float)N;
    if (pos_idx) acc += trip / (float)(N * P) * 2.0f;
  }
  if (lane == 0) part[wave] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < NW; ++w) t += part[w];
    *out_loss = t;
  }
}

template <int NE>
__global__ __launch_bounds__(256) void saliency_bwd_kernel(
    const float* __restrict__ s_pos, const float* __restrict__ s_neg,
    const double* __restrict__ label, const uint8_t* __restrict__ vmask,
    const int64_t* __restrict__ pos_idx, const int64_t* __restrict__ neg_idx, int N, int L, int P,
    float rank_coef, float margin, const float* __restrict__ gscale, float* __restrict__ ds_pos,
    float* __restrict__ ds_neg, const int32_t* __restrict__ n_valid) {
  saliency_bwd_body<NE>(s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, *gscale, ds_pos, ds_neg, n_valid,
                        blockIdx.x);
}

}  // namespace

extern "C" int mesm_nll_smooth_fwd(const float* logit, const int64_t* label, const uint8_t* mask,
                                   float* row_loss, float* row_lse, uint8_t* correct, int64_t R,
                                   int32_t C, float eps, void* stream) {
  if (!logit || !label || !row_loss || !row_lse || R <= 0 || C <= 0) return MESM_EINVAL;
  const dim3 grid((unsigned)R), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (C <= 256 * 8) hipLaunchKernelGGL(nll_fwd_kernel<8>, grid, block, 0, st, logit, label, mask, row_loss, row_lse, correct, C, eps);
  else if (C <= 256 * 20) hipLaunchKernelGGL(nll_fwd_kernel<20>, grid, block, 0, st, logit, label, mask, row_loss, row_lse, correct, C, eps);
  else hipLaunchKernelGGL(nll_fwd_kernel<0>, grid, block, 0, st, logit, label, mask, row_loss, row_lse, correct, C, eps);
  return mesm_launch_status();
}

extern "C" int mesm_nll_smooth_bwd(const float* logit, const int64_t* label, const float* row_lse,
                                   const float* row_grad, float* dlogit, int64_t R, int32_t C,
                                   float eps, void* stream) {
  if (!logit || !label || !row_lse || !row_grad || !dlogit || R <= 0 || C <= 0) return MESM_EINVAL;
  int gy = (C + 1023) / 1024;
  if (gy < 1) gy = 1;
  hipLaunchKernelGGL(nll_bwd_kernel, dim3((unsigned)R, gy), dim3(256), 0, (hipStream_t)stream,
                     logit, label, row_lse, row_grad, dlogit, C, eps);
  return mesm_launch_status();
}

extern "C" int mesm_saliency_loss_fwd_nv(const float* s_pos, const float* s_neg, const double* label,
                                         const uint8_t* vmask, const int64_t* pos_idx,
                                         const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                                         float rank_coef, float margin, float* out_loss, const int32_t* n_valid,
                                         void* stream) {
  if (!s_pos || !s_neg || !label || !vmask || !out_loss || N <= 0 || L <= 0) return MESM_EINVAL;
  if (2 * L > 64 * SAL_MAXE) return MESM_EINVAL;
  if ((pos_idx == nullptr) != (neg_idx == nullptr)) return MESM_EINVAL;
  if (pos_idx && (P <= 0 || P > 64)) return MESM_EINVAL;
  if (2 * L <= 64 * 4)
    hipLaunchKernelGGL((saliency_fwd_kernel<4, 16>), dim3(1), dim3(64 * 16), 0, (hipStream_t)stream, s_pos, s_neg,
                       label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, out_loss, n_valid);
  else
    hipLaunchKernelGGL((saliency_fwd_kernel<SAL_MAXE, 8>), dim3(1), dim3(64 * 8), 0, (hipStream_t)stream, s_pos,
                       s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, out_loss, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_saliency_loss_fwd(const float* s_pos, const float* s_neg, const double* label,
                                      const uint8_t* vmask, const int64_t* pos_idx,
                                      const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                                      float rank_coef, float margin, float* out_loss,
                                      void* stream) {
  return mesm_saliency_loss_fwd_nv(s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, out_loss,
                                   nullptr, stream);
}

extern "C" int mesm_saliency_loss_bwd_nv(const float* s_pos, const float* s_neg, const double* label,
                                         const uint8_t* vmask, const int64_t* pos_idx,
                                         const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                                         float rank_coef, float margin, const float* gscale,
                                         float* ds_pos, float* ds_neg, const int32_t* n_valid, void* stream) {
  if (!s_pos || !s_neg || !label || !vmask || !gscale || !ds_pos || !ds_neg || N <= 0 || L <= 0)
    return MESM_EINVAL;
  if (2 * L > 64 * SAL_MAXE) return MESM_EINVAL;
  if ((pos_idx == nullptr) != (neg_idx == nullptr)) return MESM_EINVAL;
  if (pos_idx && (P <= 0 || P > 64)) return MESM_EINVAL;
  if (2 * L <= 64 * 4)
    hipLaunchKernelGGL(saliency_bwd_kernel<4>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin,
                       gscale, ds_pos, ds_neg, n_valid);
  else
    hipLaunchKernelGGL(saliency_bwd_kernel<SAL_MAXE>, dim3((N + 3) / 4), dim3(256), 0, (hipStream_t)stream,
                       s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin,
                       gscale, ds_pos, ds_neg, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_saliency_loss_bwd(const float* s_pos, const float* s_neg, const double* label,
                                      const uint8_t* vmask, const int64_t* pos_idx,
                                      const int64_t* neg_idx, int32_t N, int32_t L, int32_t P,
                                      float rank_coef, float margin, const float* gscale,
                                      float* ds_pos, float* ds_neg, void* stream) {
  return mesm_saliency_loss_bwd_nv(s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, gscale,
                                   ds_pos, ds_neg, nullptr, stream);
}
