// Device bodies of loss kernels that run both as launches of their own (losses.hip) and as workgroups of the merged
// criterion launches (criterion.hip: mesm_criterion_fwd, mesm_criterion_bwd).
#pragma once
#include "common.hpp"

namespace {

// criterion.py:139-221.  One workgroup, one wave per pair (rows strided over 4 waves).
constexpr int SAL_STAGES = 11;   // rand_idx in range(1, 12)
constexpr int SAL_MAXE = 20;     // elements of [pos || neg] per lane: 2L <= 1280
// The per-lane arrays are sized by the template parameter NE (4: 2L <= 256, every QVHighlights / Charades batch;
// 20: up to TACoS' 512 clips).  With the 20-element arrays the one-workgroup forward (1,024 threads, 128 VGPRs per
// lane) spilled 336 bytes per lane to scratch memory and took 40 us for 32 rows of 150 elements.
template <int NE>
struct SalRow {
  float x[NE];    // scores / tau after the -1e3 padding fill
  float w[NE];    // sum_r [label >= r] * vmask / (cnt_r + 1e-6)
  float mx, T;          // row max, sum exp + 1e-6
  int amax;
  float rank;           // sum_r -(S_r / (cnt_r + 1e-6)) over stages with positives
  float wsum;
};

template <int NE>
__device__ __forceinline__ void sal_row_stats(const float* sp, const float* sn, const double* lab,
                                              const uint8_t* vm, int L, int lane, SalRow<NE>& R) {
  const int n2 = 2 * L;
  float cnt[SAL_STAGES];
#pragma unroll
  for (int r = 0; r < SAL_STAGES; ++r) cnt[r] = 0.0f;
  float mx = -INFINITY;
  int amax = 0x7fffffff;
  int stage_of[NE];  // number of stages r (1..11) with label >= r, for this element
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int j = e * 64 + lane;
    float x = -INFINITY;
    int st = 0;
    if (j < n2) {
      const int l = j < L ? j : j - L;
      const float v = vm[l] ? 1.0f : 0.0f;
      const float s = j < L ? sp[l] : sn[l];
      x = (v * s + (1.0f - v) * -1e3f) / 0.5f;
      const double lb = j < L ? lab[l] : 0.0;
#pragma unroll
      for (int r = 0; r < SAL_STAGES; ++r)
        if (lb >= (double)(r + 1)) { cnt[r] += 1.0f; st = r + 1; }
      if (x > mx) { mx = x; amax = j; }
    }
    R.x[e] = x;
    stage_of[e] = st;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(mx, o, 64);
    int a2 = __shfl_xor(amax, o, 64);
    if (m2 > mx || (m2 == mx && a2 < amax)) { mx = m2; amax = a2; }
  }
  float inv[SAL_STAGES];
#pragma unroll
  for (int r = 0; r < SAL_STAGES; ++r) {
    cnt[r] = wave_sum(cnt[r]);
    inv[r] = cnt[r] > 0.0f ? 1.0f / (cnt[r] + 1e-6f) : 0.0f;
  }
  float se = 0.0f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int j = e * 64 + lane;
    if (j < n2) se += __expf(R.x[e] - mx);
  }
  const float T = wave_sum(se) + 1e-6f;
  const float logT = __logf(T);
  float rank = 0.0f, wsum = 0.0f;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int j = e * 64 + lane;
    float w = 0.0f;
    if (j < n2) {
      const int l = j < L ? j : j - L;
      const float v = vm[l] ? 1.0f : 0.0f;
      // labels are monotone in r: element is positive for stages 1..stage_of
#pragma unroll
      for (int r = 0; r < SAL_STAGES; ++r)
        if (r < stage_of[e]) w += inv[r];
      w *= v;
      const float lp = (R.x[e] - mx) - logT;
      rank -= w * lp;
      wsum += w;
    }
    R.w[e] = w;
  }
  R.mx = mx; R.T = T; R.amax = amax;
  R.rank = wave_sum(rank);
  R.wsum = wave_sum(wsum);
}

// d logit of the label-smoothed NLL for row r (criterion.py:291-306), g = the row's weight; columns strided over `ny` workgroups
__device__ __forceinline__ void nll_bwd_body(const float* __restrict__ logit, const int64_t* __restrict__ label,
                                             const float* __restrict__ row_lse, float g, float* __restrict__ dlogit, int C,
                                             float eps, int64_t r, int by, int ny) {
  const float lse = row_lse[r];
  const int lab = (int)label[r];
  const float* x = logit + r * C;
  float* d = dlogit + r * C;
  const float u = eps / (float)C;
  for (int c = by * 256 + threadIdx.x; c < C; c += ny * 256) {
    float v = 0.0f;
    if (g != 0.0f) {
      float p = __expf(x[c] - lse);
      v = g * (p - (c == lab ? (1.0f - eps) : 0.0f) - u);
    }
    mesm_store_wt(d + c, v);  // (20 MB at C = 5003: write-through, common.hpp)
  }
}

// saliency backward for the 4 pairs of workgroup `bid` (a wave per pair); gs = upstream gradient of loss_saliency
template <int NE>
__device__ __forceinline__ void saliency_bwd_body(
    const float* __restrict__ s_pos, const float* __restrict__ s_neg,
    const double* __restrict__ label, const uint8_t* __restrict__ vmask,
    const int64_t* __restrict__ pos_idx, const int64_t* __restrict__ neg_idx, int N, int L, int P,
    float rank_coef, float margin, float gs, float* __restrict__ ds_pos,
    float* __restrict__ ds_neg, const int32_t* __restrict__ n_valid, int bid) {
  const int lane = threadIdx.x & 63;
  const int n = bid * 4 + (threadIdx.x >> 6);
  if (n >= N) return;
  if (n_valid) {
    N = *n_valid;
    if (n >= N) {  // padding pair: zero gradient rows
      for (int l = lane; l < L; l += 64) ds_pos[(int64_t)n * L + l] = ds_neg[(int64_t)n * L + l] = 0.0f;
      return;
    }
  }
  const float* sp = s_pos + (int64_t)n * L;
  const float* sn = s_neg + (int64_t)n * L;
  const uint8_t* vm = vmask + (int64_t)n * L;
  SalRow<NE> R;
  sal_row_stats(sp, sn, label + (int64_t)n * L, vm, L, lane, R);
  const float k_rank = 1.0f / ((float)N * rank_coef);
  const float eps_over_T = 1e-6f / R.T;
#pragma unroll
  for (int e = 0; e < NE; ++e) {
    const int j = e * 64 + lane;
    if (j < 2 * L) {
      const int l = j < L ? j : j - L;
      const float v = vm[l] ? 1.0f : 0.0f;
      const float q = __expf(R.x[e] - R.mx) / R.T;
      // d rank_row / d x_j = -(w_j - wsum * (q_j + [j == argmax] * eps/T))
      float dx = -(R.w[e] - R.wsum * (q + (j == R.amax ? eps_over_T : 0.0f)));
      float g = k_rank * dx * (1.0f / 0.5f) * v;  // x = (v*s + (1-v)*-1e3) / tau
      if (j >= L) {
        float sg = 1.0f / (1.0f + __expf(-sn[l]));
        g += v * sg / (float)N;  // d/ds [-log(1 - sigmoid(s))] = sigmoid(s)
        ds_neg[(int64_t)n * L + l] = gs * g;
      } else {
        if (pos_idx) {
          const float kt = 2.0f / (float)(N * P);
          for (int c = 0; c < P; ++c) {
            const int64_t pi = pos_idx[(int64_t)n * P + c], ni = neg_idx[(int64_t)n * P + c];
            const float t = margin + sp[ni] - sp[pi];
            if (t >= 0.0f) {
              if (ni == l) g += kt;
              if (pi == l) g -= kt;
            }
          }
        }
        ds_pos[(int64_t)n * L + l] = gs * g;
      }
    }
  }
}

// ---------------------------------------------------------------- label-smoothed NLL, forward
// One 256-thread group per row (criterion.py:291-306): a workgroup of its own (nll_fwd_kernel) or a quarter of a
// 1,024-thread workgroup of the merged criterion forward; `t` = the thread's index in its group, `sh` = the group's LDS.
// NPT > 0: the row (C <= 256 NPT classes) is read ONCE, NPT independent loads per thread in flight together, and both
// passes run on registers (the two-pass form walked the row twice with one load in flight per thread: 19.5 -> 9.3 us at
// 1024 x 5003); NPT = 0: any C, two passes over the L2-resident row.  live = false: a group past the last row (it takes
// part in the barriers and writes nothing).
struct NllShared {
  float sh[4];
  float shm[4];
  int shi[4];
};

__device__ __forceinline__ float nll_group_sum(float v, float* sh, int t) {
  v = wave_sum(v);
  if ((t & 63) == 0) sh[t >> 6] = v;
  __syncthreads();
  const float r = sh[0] + sh[1] + sh[2] + sh[3];
  __syncthreads();
  return r;
}

template <int NPT>
__device__ __forceinline__ void nll_fwd_body(const float* __restrict__ logit, const int64_t* __restrict__ label,
                                             const uint8_t* __restrict__ mask, float* __restrict__ row_loss,
                                             float* __restrict__ row_lse, uint8_t* __restrict__ correct, int C, float eps,
                                             int64_t r, bool live, int t, NllShared& S) {
  const float* x = logit + r * C;
  const int wave = t >> 6;
  // pass 1: max + first argmax + plain sum
  float m = -INFINITY, s = 0.0f;
  int am = 0x7fffffff;
  float v[NPT > 0 ? NPT : 1];
  if (NPT > 0) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int c = t + k * 256;
      v[k] = c < C ? x[c] : -INFINITY;
    }
#pragma unroll
    for (int k = 0; k < NPT; ++k) {
      const int c = t + k * 256;
      if (c < C) {
        s += v[k];
        if (v[k] > m) { m = v[k]; am = c; }
      }
    }
  } else {
    for (int c = t; c < C; c += 256) {
      float u = x[c];
      s += u;
      if (u > m) { m = u; am = c; }
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    float m2 = __shfl_xor(m, o, 64);
    int a2 = __shfl_xor(am, o, 64);
    if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
  }
  if ((t & 63) == 0) { S.shm[wave] = m; S.shi[wave] = am; }
  __syncthreads();
  float M = S.shm[0];
  int AM = S.shi[0];
#pragma unroll
  for (int w = 1; w < 4; ++w)
    if (S.shm[w] > M || (S.shm[w] == M && S.shi[w] < AM)) { M = S.shm[w]; AM = S.shi[w]; }
  __syncthreads();
  const float total = nll_group_sum(s, S.sh, t);
  // pass 2: sum exp
  float e = 0.0f;
  if (NPT > 0) {
#pragma unroll
    for (int k = 0; k < NPT; ++k) e += __expf(v[k] - M);  // (exp(-inf) = 0 beyond C)
  } else {
    for (int c = t; c < C; c += 256) e += __expf(x[c] - M);
  }
  const float E = nll_group_sum(e, S.sh, t);
  if (t == 0 && live) {
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

// ---------------------------------------------------------------- saliency losses, forward
// one workgroup (deterministic sum) of NW waves, a wave per pair
template <int NE, int NW>
__device__ __forceinline__ void saliency_fwd_body(
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

}  // namespace
