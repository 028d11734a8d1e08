// Per-pair loss reductions of model/criterion.py and the matching cost + assignment of
// model/matcher.py as gfx950 kernels.  All are HBM/latency-bound and tiny except the
// masked-LM NLL over the vocabulary (N*Lw rows x C ~ 5k classes).
#include "common.hpp"
#include "loss_bodies.hpp"

namespace {

// ---------------------------------------------------------------- label-smoothed NLL
// one workgroup per row (criterion.py:291-306); the body is in loss_bodies.hpp (nll_fwd_body).
template <int NPT>
__global__ __launch_bounds__(256) void nll_fwd_kernel(
    const float* __restrict__ logit, const int64_t* __restrict__ label,
    const uint8_t* __restrict__ mask, float* __restrict__ row_loss,
    float* __restrict__ row_lse, uint8_t* __restrict__ correct, int C, float eps) {
  __shared__ NllShared sh;
  nll_fwd_body<NPT>(logit, label, mask, row_loss, row_lse, correct, C, eps, blockIdx.x, true, threadIdx.x, sh);
}

__global__ __launch_bounds__(256) void nll_bwd_kernel(
    const float* __restrict__ logit, const int64_t* __restrict__ label,
    const float* __restrict__ row_lse, const float* __restrict__ row_grad,
    float* __restrict__ dlogit, int C, float eps) {
  nll_bwd_body(logit, label, row_lse, row_grad[blockIdx.x], dlogit, C, eps, blockIdx.x, blockIdx.y, gridDim.y);
}

// ---------------------------------------------------------------- saliency losses
// one workgroup (deterministic sum) of NW waves, a wave per pair (16 waves; 8 with the 20-element arrays, whose
// registers do not fit the 128-VGPR budget of a 1,024-thread workgroup); body: loss_bodies.hpp
template <int NE, int NW>
__global__ __launch_bounds__(64 * NW) void saliency_fwd_kernel(
    const float* __restrict__ s_pos, const float* __restrict__ s_neg,
    const double* __restrict__ label, const uint8_t* __restrict__ vmask,
    const int64_t* __restrict__ pos_idx, const int64_t* __restrict__ neg_idx, int N, int L, int P,
    float rank_coef, float margin, float* __restrict__ out_loss, const int32_t* __restrict__ n_valid) {
  saliency_fwd_body<NE, NW>(s_pos, s_neg, label, vmask, pos_idx, neg_idx, N, L, P, rank_coef, margin, out_loss, n_valid);
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
