// LayerNorm forward / backward, one wave64 per row, the row cached in registers
// (NCH chunks of 64*VEC floats per lane), wave-shuffle reductions.
// HBM-bound: fwd reads x once and writes y once (+2 floats/row); bwd reads dy, x once,
// writes dx once; dgamma/dbeta are reduced per workgroup through LDS and added
// atomically (few adders per address: the grid is capped).
#include <cstddef>
#include "common.hpp"

namespace {

constexpr int LN_THREADS = 256;
constexpr int LN_WAVES = LN_THREADS / 64;
// backward: workgroup size and cap are tuning knobs (tools/ln_bench.py): every workgroup ends with 2 D float
// atomics onto the same dgamma / dbeta addresses, so fewer, fatter workgroups trade tail contention
// against rows in flight
// against rows in flight.  Measured, 4800 / 2400 / 1024 / 320 rows of 256: 256 threads x 128 workgroups
// 11.4 / 8.0 / 5.6 / 4.1 us; 1024 threads, 128 workgroups from 4000 rows and 64 below: 8.1 / 5.9 / 4.8 / 3.8 us.
#ifndef MESM_LNB_THREADS
#define MESM_LNB_THREADS 1024
#endif
#ifndef MESM_LNB_CAP
#define MESM_LNB_CAP 128
#endif
constexpr int LNB_THREADS = MESM_LNB_THREADS;
constexpr int LNB_WAVES = LNB_THREADS / 64;

// optional dropout fused behind the normalisation (LinearLayer: LN -> Dropout -> Linear,
// model.py:421-431): the forward writes dropout(LN(x)), the backward masks dy while loading it.
struct LnDrop {
  uint32_t thresh;  // 0 = off
  uint32_t seed;
  float inv_keep;
  const uint32_t* seed_offset;
};

inline LnDrop make_drop(float p, uint32_t seed, const uint32_t* seed_offset) {
  LnDrop d;
  d.thresh = p > 0.f ? mesm_drop_threshold(p) : 0u;
  d.seed = seed;
  d.inv_keep = 1.0f / (1.0f - p);
  d.seed_offset = seed_offset;
  return d;
}

template <int VEC>
__device__ __forceinline__ void ld_vec(const float* __restrict__ p, float* d) {
  if (VEC == 4) {
    float4 t = *reinterpret_cast<const float4*>(p);
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
  } else if (VEC == 2) {
    float2 t = *reinterpret_cast<const float2*>(p);
    d[0] = t.x; d[1] = t.y;
  } else {
    d[0] = *p;
  }
}

template <int VEC>
__device__ __forceinline__ void st_lds(float* __restrict__ p, const float* s) {  // (LDS scratch: plain stores)
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(s[0], s[1], s[2], s[3]);
  else if (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(s[0], s[1]);
  else *p = s[0];
}

// (output stores: write-through, common.hpp, unless built with -DMESM_LN_WT=0)
#ifndef MESM_LN_WT
#define MESM_LN_WT 1
#endif
template <int VEC>
__device__ __forceinline__ void st_vec(float* __restrict__ p, const float* s) {
#if MESM_LN_WT
  if (VEC == 4) mesm_store_wt4(p, s[0], s[1], s[2], s[3]);
  else if (VEC == 2) mesm_store_wt2(p, s[0], s[1]);
  else mesm_store_wt(p, s[0]);
#else
  if (VEC == 4) *reinterpret_cast<float4*>(p) = make_float4(s[0], s[1], s[2], s[3]);
  else if (VEC == 2) *reinterpret_cast<float2*>(p) = make_float2(s[0], s[1]);
  else *p = s[0];
#endif
}

// TAIL (VEC = 4 only): D is not a multiple of 4, so rows start 8- or 4-byte aligned.  The lane chunks are SHIFTED per
// row by s = (row * D) & 3 elements so that every interior chunk is a 16-byte ALIGNED access (misaligned 16-byte accesses
// run at about half rate here: 2400 x 2818 20.8 -> 29.7 us); the first and the last chunk of a row are partial and go
// element by element.  The 2-wide / 1-wide paths this replaces ran 36 / 72 narrow accesses per lane and row:
// 8192 x 4098 (TACoS) 192 us = 1.4 TB/s against 69 us at 4096 columns.
// A TWIN output (tw.y != NULL): the same input normalised once, written twice under two dropout masks -- the reference
// projects the raw video features twice (model.py:166, 201: the main path and the SS-MESM copy of the batch), each through
// its own Dropout; when the copy IS the batch the two LayerNorms read the same 27 MB.
struct LnTwin {
  float* y;
  float* mean;
  float* rstd;
  LnDrop dr;
};

template <int VEC, int NCH, bool TAIL = false>
__device__ __forceinline__ void ln_fwd_body(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t rows, int D, float eps, LnDrop dr,
    const float* __restrict__ add, float* __restrict__ y2, int bid, int nblk, const LnTwin tw = LnTwin{}) {
  const int lane = threadIdx.x & 63;
  const int64_t wave_global = (int64_t)bid * LN_WAVES + (threadIdx.x >> 6);
  const int64_t nwaves = (int64_t)nblk * LN_WAVES;
  const float invD = 1.0f / (float)D;
  const uint32_t dseed = dr.seed + (dr.seed_offset ? *dr.seed_offset : 0u);
  const uint32_t dseed_b = tw.dr.seed + (tw.dr.seed_offset ? *tw.dr.seed_offset : 0u);
  for (int64_t row = wave_global; row < rows; row += nwaves) {
    const float* xr = x + row * D;
    const int sh = TAIL ? (int)((row * D) & 3) : 0;  // chunk c of lane l starts at column (c * 64 + l) * VEC - sh
    float v[NCH][VEC];
    float s = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * VEC - sh;
      if (col >= 0 && col + VEC <= D) {  // (without TAIL: D % VEC == 0 guaranteed by the host)
        ld_vec<VEC>(xr + col, v[c]);
#pragma unroll
        for (int e = 0; e < VEC; ++e) s += v[c][e];
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          v[c][e] = (TAIL && col + e >= 0 && col + e < D) ? xr[col + e] : 0.0f;
          s += v[c][e];
        }
      }
    }
    const float mu = wave_sum(s) * invD;
    float q = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * VEC - sh;
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        const float d = v[c][e] - mu;
        if (col + e >= 0 && col + e < D) q += d * d;
      }
    }
    const float var = wave_sum(q) * invD;
    const float rs = rsqrtf(var + eps);
    float* yr = y + row * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      const int col = (c * 64 + lane) * VEC - sh;
      const bool full = col >= 0 && col + VEC <= D;
      if (TAIL && !full) {  // a row's partial first / last chunk, element by element
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (col + e >= 0 && col + e < D) {
            float o = (v[c][e] - mu) * rs * gamma[col + e] + beta[col + e];
            if (tw.y) {
              float ob = o;
              if (tw.dr.thresh) ob = mesm_dropout_apply(ob, (uint32_t)(row * D + col + e), dseed_b, tw.dr.thresh, tw.dr.inv_keep);
              tw.y[row * D + col + e] = ob;
            }
            if (dr.thresh) o = mesm_dropout_apply(o, (uint32_t)(row * D + col + e), dseed, dr.thresh, dr.inv_keep);
            yr[col + e] = o;
            if (y2) y2[row * D + col + e] = o + add[row * D + col + e];
          }
        }
      } else if (full) {
        float g[VEC], b[VEC], o[VEC];
        if (TAIL) {  // (gamma + col is only 8-byte aligned when the row is shifted)
#pragma unroll
          for (int e = 0; e < VEC; e += 2) {
            ld_vec<2>(gamma + col + e, g + e);
            ld_vec<2>(beta + col + e, b + e);
          }
        } else {
          ld_vec<VEC>(gamma + col, g);
          ld_vec<VEC>(beta + col, b);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = (v[c][e] - mu) * rs * g[e] + b[e];
        if (tw.y) {
          float ob[VEC];
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            ob[e] = o[e];
            if (tw.dr.thresh) ob[e] = mesm_dropout_apply(ob[e], (uint32_t)(row * D + col + e), dseed_b, tw.dr.thresh, tw.dr.inv_keep);
          }
          st_vec<VEC>(tw.y + row * D + col, ob);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (dr.thresh) o[e] = mesm_dropout_apply(o[e], (uint32_t)(row * D + col + e), dseed, dr.thresh, dr.inv_keep);
        st_vec<VEC>(yr + col, o);
        if (y2) {  // y + add: the `with_pos_embed` query of the attention block that consumes y
          float a[VEC];
          ld_vec<VEC>(add + row * D + col, a);
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] += a[e];
          st_vec<VEC>(y2 + row * D + col, o);
        }
      }
    }
    if (lane == 0) {
      mean[row] = mu;
      rstd[row] = rs;
      if (tw.y) { tw.mean[row] = mu; tw.rstd[row] = rs; }
    }
  }
}

// one LayerNorm problem of a grouped launch (mesm_layernorm_{fwd,bwd}_group): the argument lists of the plain kernels
struct LnProb {
  const float *x, *gamma, *beta;
  float *y, *mean, *rstd;
  int64_t rows;
  int D;
  float eps;
  LnDrop dr;
  const float* add;
  float* y2;
  // backward
  const float* dy;
  float *dx, *dgamma, *dbeta;
  int accumulate_dx;
  float* dx2;
  LnDrop dr2;
  const float *dyb, *addend;
  int relu_in;
};
constexpr int LN_GROUP_MAX = 8;
struct LnGroup {
  LnProb p[LN_GROUP_MAX];
  int start[LN_GROUP_MAX + 1];  // first workgroup of every problem
  int n;
};

// the problem a workgroup of a grouped launch belongs to, read from the kernarg segment with a wave-uniform
// dynamic offset (see gemm_wstage_group_kernel for why not g.p[gi])
__device__ __forceinline__ LnProb ln_group_pick(const LnGroup& g, int& local, int& nblk) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < LN_GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const int first = *reinterpret_cast<const int*>(ka + offsetof(LnGroup, start) + (size_t)gi * sizeof(int));
  const int next = *reinterpret_cast<const int*>(ka + offsetof(LnGroup, start) + (size_t)(gi + 1) * sizeof(int));
  local = bid - first;
  nblk = next - first;
  return *reinterpret_cast<const LnProb*>(ka + offsetof(LnGroup, p) + (size_t)gi * sizeof(LnProb));
}

template <int VEC, int NCH, bool TAIL = false>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma,
    const float* __restrict__ beta, float* __restrict__ y, float* __restrict__ mean,
    float* __restrict__ rstd, int64_t rows, int D, float eps, LnDrop dr,
    const float* __restrict__ add, float* __restrict__ y2) {
  ln_fwd_body<VEC, NCH, TAIL>(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, blockIdx.x, gridDim.x);
}

template <int NCH>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_twin_kernel(
    const float* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta, float* __restrict__ y,
    float* __restrict__ mean, float* __restrict__ rstd, int64_t rows, int D, float eps, LnDrop dr, const LnTwin tw) {
  ln_fwd_body<4, NCH, true>(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, nullptr, nullptr, blockIdx.x, gridDim.x, tw);
}

// up to LN_GROUP_MAX independent LayerNorms (same VEC / NCH class) in ONE launch
template <int VEC, int NCH>
__global__ __launch_bounds__(LN_THREADS) void ln_fwd_group_kernel(const LnGroup g) {
  int local, nblk;
  const LnProb q = ln_group_pick(g, local, nblk);
  ln_fwd_body<VEC, NCH>(q.x, q.gamma, q.beta, q.y, q.mean, q.rstd, q.rows, q.D, q.eps, q.dr, q.add, q.y2, local, nblk);
}

template <int VEC, int NCH, bool LDS_REDUCE>
__device__ __forceinline__ void ln_bwd_body(
    const float* __restrict__ dy, const float* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int64_t rows, int D, int accumulate_dx, LnDrop dr,
    float* __restrict__ dx2, LnDrop dr2, const float* __restrict__ dyb, const float* __restrict__ addend,
    int bid, int nblk, int relu_in = 0) {
  extern __shared__ __attribute__((aligned(16))) float red[];  // LNB_WAVES * D when LDS_REDUCE
  const uint32_t dseed = dr.seed + (dr.seed_offset ? *dr.seed_offset : 0u);
  // second output: dx under the dropout mask of the block that PRODUCED the LayerNorm input
  // (y = LN(res + dropout(block(.)))): that block's backward wants mask * dx, which used to be a separate
  // element-wise launch (30 per step); the mask index is the block output's dense index row * D + col
  const uint32_t dseed2 = dr2.seed + (dr2.seed_offset ? *dr2.seed_offset : 0u);
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int64_t wave_global = (int64_t)bid * LNB_WAVES + wave;
  const int64_t nwaves = (int64_t)nblk * LNB_WAVES;
  const float invD = 1.0f / (float)D;

  float g[NCH][VEC];
  float dg[NCH][VEC], db[NCH][VEC];
#pragma unroll
  for (int c = 0; c < NCH; ++c) {
    int col = (c * 64 + lane) * VEC;
#pragma unroll
    for (int e = 0; e < VEC; ++e) { dg[c][e] = 0.0f; db[c][e] = 0.0f; g[c][e] = 0.0f; }
    if (col < D) ld_vec<VEC>(gamma + col, g[c]);
  }

  for (int64_t row = wave_global; row < rows; row += nwaves) {
    const float mu = mean[row];
    const float rs = rstd[row];
    const float* xr = x + row * D;
    const float* dyr = dy + row * D;
    float xh[NCH][VEC], dv[NCH][VEC];
    bool xpos[NCH][VEC];  // x > 0: the ReLU that produced x (relu_in), applied to dx on store
    float s1 = 0.0f, s2 = 0.0f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int col = (c * 64 + lane) * VEC;
#pragma unroll
      for (int e = 0; e < VEC; ++e) xpos[c][e] = true;
      if (col < D) {
        float xv[VEC];
        ld_vec<VEC>(xr + col, xv);
        if (relu_in) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) xpos[c][e] = xv[e] > 0.0f;
        }
        ld_vec<VEC>(dyr + col, dv[c]);
        if (dyb) {  // y had a second consumer (y + pos went to an attention block): its gradient joins here
          float t2[VEC];
          ld_vec<VEC>(dyb + row * D + col, t2);
#pragma unroll
          for (int e = 0; e < VEC; ++e) dv[c][e] += t2[e];
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          if (dr.thresh) dv[c][e] = mesm_dropout_apply(dv[c][e], (uint32_t)(row * D + col + e), dseed, dr.thresh, dr.inv_keep);
          xh[c][e] = (xv[e] - mu) * rs;
          float dyg = dv[c][e] * g[c][e];
          s1 += dyg * xh[c][e];
          s2 += dyg;
          dg[c][e] += dv[c][e] * xh[c][e];
          db[c][e] += dv[c][e];
        }
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) { xh[c][e] = 0.0f; dv[c][e] = 0.0f; }
      }
    }
    if (dx == nullptr) continue;  // parameter gradients only (a grouped launch's input LayerNorm over raw features)
    const float c1 = wave_sum(s1) * invD;
    const float c2 = wave_sum(s2) * invD;
    float* dxr = dx + row * D;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int col = (c * 64 + lane) * VEC;
      if (col < D) {
        float o[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) o[e] = rs * (dv[c][e] * g[c][e] - c2 - xh[c][e] * c1);
        if (accumulate_dx || addend) {  // + the gradient that reaches x on another route (a residual branch)
          float old[VEC];
          ld_vec<VEC>((addend ? addend + row * D : dxr) + col, old);
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] += old[e];
        }
        if (relu_in) {  // x = relu(z): hand d z to the producing block (its own mask launch goes away)
#pragma unroll
          for (int e = 0; e < VEC; ++e) o[e] = xpos[c][e] ? o[e] : 0.0f;
        }
        st_vec<VEC>(dxr + col, o);
        if (dx2) {
#pragma unroll
          for (int e = 0; e < VEC; ++e)
            o[e] = mesm_dropout_apply(o[e], (uint32_t)(row * D + col + e), dseed2, dr2.thresh, dr2.inv_keep);
          st_vec<VEC>(dx2 + row * D + col, o);
        }
      }
    }
  }

  if (LDS_REDUCE) {
    // two rounds (dgamma, dbeta) through one LNB_WAVES x D buffer
#pragma unroll
    for (int round = 0; round < 2; ++round) {
#pragma unroll
      for (int c = 0; c < NCH; ++c) {
        int col = (c * 64 + lane) * VEC;
        if (col < D) st_lds<VEC>(red + wave * D + col, round == 0 ? dg[c] : db[c]);
      }
      __syncthreads();
      float* dst = round == 0 ? dgamma : dbeta;
      for (int col = threadIdx.x; col < D; col += LNB_THREADS) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < LNB_WAVES; ++w) t += red[w * D + col];
        atomicAdd(dst + col, t);
      }
      __syncthreads();
    }
  } else {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
      int col = (c * 64 + lane) * VEC;
      if (col < D) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          atomicAdd(dgamma + col + e, dg[c][e]);
          atomicAdd(dbeta + col + e, db[c][e]);
        }
      }
    }
  }
}

template <int VEC, int NCH, bool LDS_REDUCE>
__global__ __launch_bounds__(LNB_THREADS) void ln_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ x,
    const float* __restrict__ gamma, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dx, float* __restrict__ dgamma,
    float* __restrict__ dbeta, int64_t rows, int D, int accumulate_dx, LnDrop dr,
    float* __restrict__ dx2, LnDrop dr2, const float* __restrict__ dyb, const float* __restrict__ addend) {
  ln_bwd_body<VEC, NCH, LDS_REDUCE>(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, dr, dx2, dr2, dyb,
                                    addend, blockIdx.x, gridDim.x);
}

template <int VEC, int NCH>
__global__ __launch_bounds__(LNB_THREADS) void ln_bwd_group_kernel(const LnGroup g) {
  int local, nblk;
  const LnProb q = ln_group_pick(g, local, nblk);
  ln_bwd_body<VEC, NCH, true>(q.dy, q.x, q.gamma, q.mean, q.rstd, q.dx, q.dgamma, q.dbeta, q.rows, q.D, q.accumulate_dx,
                              q.dr, q.dx2, q.dr2, q.dyb, q.addend, local, nblk, q.relu_in);
}

// Parameter gradients only (dx == NULL: the input needs no gradient, e.g. the LayerNorm over the raw
// 2818-d video features): a column-parallel reduction.  A workgroup owns 64 columns x one row chunk; its 4 waves
// take every 4th row of the chunk, 4 rows in flight each (the row loop is a chain of dependent global loads
// otherwise), meet in LDS, and wave 0 issues the one atomic per column and workgroup.
// dy_b / dr_b (optional): the gradient of a TWIN output (LnTwin: the same x normalised once, two dropout masks) -- x is read
// once for both.
__global__ __launch_bounds__(256) void ln_bwd_params_kernel(
    const float* __restrict__ dy, const float* __restrict__ x, const float* __restrict__ mean,
    const float* __restrict__ rstd, float* __restrict__ dgamma, float* __restrict__ dbeta, int64_t rows,
    int D, int rows_per_block, LnDrop dr, const float* __restrict__ dy_b, LnDrop dr_b) {
  __shared__ float red[2][3][64];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int col = blockIdx.x * 64 + lane;
  const bool live = col < D;
  const int c = live ? col : D - 1;
  const uint32_t dseed = dr.seed + (dr.seed_offset ? *dr.seed_offset : 0u);
  const uint32_t dseed_b = dr_b.seed + (dr_b.seed_offset ? *dr_b.seed_offset : 0u);
  const int64_t r0 = (int64_t)blockIdx.y * rows_per_block;
  const int64_t r1 = r0 + rows_per_block < rows ? r0 + rows_per_block : rows;
  float dg = 0.0f, db = 0.0f;
  for (int64_t r = r0 + w; r < r1; r += 16) {
    float d[4], d2[4], xv[4], mu[4], rs[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = r + 4 * u < r1 ? r + 4 * u : r1 - 1;
      d[u] = dy[rr * D + c];
      d2[u] = dy_b ? dy_b[rr * D + c] : 0.0f;
      xv[u] = x[rr * D + c];
      mu[u] = mean[rr];
      rs[u] = rstd[rr];
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int64_t rr = r + 4 * u;
      if (rr < r1) {
        float dv = d[u];
        if (dr.thresh) dv = mesm_dropout_apply(dv, (uint32_t)(rr * D + c), dseed, dr.thresh, dr.inv_keep);
        if (dy_b) {
          float dvb = d2[u];
          if (dr_b.thresh) dvb = mesm_dropout_apply(dvb, (uint32_t)(rr * D + c), dseed_b, dr_b.thresh, dr_b.inv_keep);
          dv += dvb;
        }
        dg += dv * (xv[u] - mu[u]) * rs[u];
        db += dv;
      }
    }
  }
  if (w) {
    red[0][w - 1][lane] = dg;
    red[1][w - 1][lane] = db;
  }
  __syncthreads();
  if (w == 0 && live) {
    dg += red[0][0][lane] + red[0][1][lane] + red[0][2][lane];
    db += red[1][0][lane] + red[1][1][lane] + red[1][2][lane];
    atomicAdd(dgamma + col, dg);
    atomicAdd(dbeta + col, db);
  }
}

inline int pick_vec(int D, const void* a, const void* b, const void* c, const void* d) {
  int vec = 4;
  while (vec > 1) {
    bool ok = (D % vec == 0);
    const void* ps[4] = {a, b, c, d};
    for (const void* p : ps) ok = ok && (p == nullptr || ((uintptr_t)p % (4 * vec)) == 0);
    if (ok) break;
    vec >>= 1;
  }
  return vec;
}

template <int VEC, int NCH, bool TAIL = false>
int fwd_launch(const float* x, const float* gamma, const float* beta, float* y, float* mean,
               float* rstd, int64_t rows, int D, float eps, LnDrop dr, const float* add, float* y2,
               hipStream_t s) {
  int64_t blocks = (rows + LN_WAVES - 1) / LN_WAVES;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL((ln_fwd_kernel<VEC, NCH, TAIL>), dim3((unsigned)blocks), dim3(LN_THREADS), 0, s, x,
                     gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2);
  return mesm_launch_status();
}

// wide rows whose length is not a multiple of 4 (2818, 4098 feature columns): 16-byte accesses with a partial last chunk
int fwd_launch_tail(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd,
                    int64_t rows, int D, float eps, LnDrop dr, const float* add, float* y2, hipStream_t s) {
  const int need = (D + 3 + 255) / 256;
  if (need <= 8) return fwd_launch<4, 8, true>(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  if (need <= 12) return fwd_launch<4, 12, true>(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  if (need <= 17) return fwd_launch<4, 17, true>(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  return MESM_EINVAL;
}

template <int VEC, int NCH>
int bwd_launch(const float* dy, const float* x, const float* gamma, const float* mean,
               const float* rstd, float* dx, float* dgamma, float* dbeta, int64_t rows, int D,
               int acc, LnDrop dr, float* dx2, LnDrop dr2, const float* dyb, const float* addend, hipStream_t s) {
  int64_t blocks = (rows + LNB_WAVES - 1) / LNB_WAVES;
  if (D <= 1024) {
    // workgroup cap: see the note at LNB_THREADS
    const int64_t cap = rows >= 4000 ? MESM_LNB_CAP : MESM_LNB_CAP / 2;
    if (blocks > cap) blocks = cap;
    size_t lds = (size_t)LNB_WAVES * D * sizeof(float);
    hipLaunchKernelGGL((ln_bwd_kernel<VEC, NCH, true>), dim3((unsigned)blocks), dim3(LNB_THREADS),
                       lds, s, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, acc, dr, dx2, dr2, dyb, addend);
  } else {
    // workgroup cap: see the note at LNB_THREADS
    const int64_t cap = rows >= 4000 ? MESM_LNB_CAP : MESM_LNB_CAP / 2;
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((ln_bwd_kernel<VEC, NCH, false>), dim3((unsigned)blocks),
                       dim3(LNB_THREADS), 0, s, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows,
                       D, acc, dr, dx2, dr2, dyb, addend);
  }
  return mesm_launch_status();
}

#define LN_DISPATCH(FN, VEC, ...)                                  \
  do {                                                             \
    int need = (D + 64 * VEC - 1) / (64 * VEC);                    \
    if (need <= 1) return FN<VEC, 1>(__VA_ARGS__);                 \
    if (need <= 2) return FN<VEC, 2>(__VA_ARGS__);                 \
    if (need <= 4) return FN<VEC, 4>(__VA_ARGS__);                 \
    if (need <= 8) return FN<VEC, 8>(__VA_ARGS__);                 \
    if (need <= 16) return FN<VEC, 16>(__VA_ARGS__);               \
    if (need <= 24) return FN<VEC, 24>(__VA_ARGS__);               \
    if (need <= 36) return FN<VEC, 36>(__VA_ARGS__);               \
    return MESM_EINVAL;                                            \
  } while (0)

}  // namespace

extern "C" int mesm_layernorm_fwd2(const float* x, const float* gamma, const float* beta,
                                   float* y, float* mean, float* rstd, int64_t rows, int32_t D,
                                   float eps, float drop_p, uint32_t drop_seed,
                                   const uint32_t* seed_offset, const float* add, float* y2, void* stream) {
  if (!x || !gamma || !beta || !y || !mean || !rstd || rows < 0 || D <= 0) return MESM_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f) return MESM_EINVAL;
  if ((add == nullptr) != (y2 == nullptr)) return MESM_EINVAL;
  if (rows == 0) return MESM_OK;
  hipStream_t s = (hipStream_t)stream;
  const LnDrop dr = make_drop(drop_p, drop_seed, seed_offset);
  int vec = pick_vec(D, x, y, gamma, beta);
  if (add && vec > pick_vec(D, add, y2, nullptr, nullptr)) vec = pick_vec(D, add, y2, nullptr, nullptr);
  static const bool no_tail = getenv("MESM_LN_NO_TAIL") != nullptr;  // A/B: the narrow-vector paths
  if (vec < 4 && !no_tail && D % 4 == 2 && D >= 1024 && D + 3 <= 17 * 256 && pick_vec(4, x, y, gamma, beta) == 4 &&
      (!add || pick_vec(4, add, y2, nullptr, nullptr) == 4))
    return fwd_launch_tail(x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  if (vec == 4) LN_DISPATCH(fwd_launch, 4, x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  if (vec == 2) LN_DISPATCH(fwd_launch, 2, x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
  LN_DISPATCH(fwd_launch, 1, x, gamma, beta, y, mean, rstd, rows, D, eps, dr, add, y2, s);
}

extern "C" int mesm_layernorm_fwd(const float* x, const float* gamma, const float* beta,
                                  float* y, float* mean, float* rstd, int64_t rows, int32_t D,
                                  float eps, float drop_p, uint32_t drop_seed,
                                  const uint32_t* seed_offset, void* stream) {
  return mesm_layernorm_fwd2(x, gamma, beta, y, mean, rstd, rows, D, eps, drop_p, drop_seed, seed_offset, nullptr,
                             nullptr, stream);
}

extern "C" int mesm_layernorm_bwd3(const float* dy, const float* x, const float* gamma,
                                   const float* mean, const float* rstd, float* dx,
                                   float* dgamma, float* dbeta, int64_t rows, int32_t D,
                                   int32_t accumulate_dx, float drop_p, uint32_t drop_seed,
                                   const uint32_t* seed_offset, float* dx2, float drop2_p,
                                   uint32_t drop2_seed, const float* dyb, const float* addend, void* stream) {
  if (!dy || !x || !gamma || !mean || !rstd || !dgamma || !dbeta || rows < 0 || D <= 0)
    return MESM_EINVAL;
  if (drop_p < 0.f || drop_p >= 1.f || drop2_p < 0.f || drop2_p >= 1.f) return MESM_EINVAL;
  if (dx2 && !dx) return MESM_EINVAL;
  const LnDrop dr2 = make_drop(drop2_p, drop2_seed, seed_offset);
  if (rows == 0) return MESM_OK;
  hipStream_t s = (hipStream_t)stream;
  const LnDrop dr = make_drop(drop_p, drop_seed, seed_offset);
  if (!dx) {
    if (accumulate_dx || dyb || addend) return MESM_EINVAL;
    const int cb = (D + 63) / 64;
    int rb = (int)((1024 + cb - 1) / cb);  // about 1024 workgroups, of 32 rows or more
    if ((int64_t)rb * 32 > rows) rb = (int)((rows + 31) / 32);
    const int rpb = (int)((rows + rb - 1) / rb);
    rb = (int)((rows + rpb - 1) / rpb);
    hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(cb, rb), dim3(256), 0, s, dy, x, mean, rstd, dgamma,
                       dbeta, rows, D, rpb, dr, (const float*)nullptr, LnDrop{});
    return mesm_launch_status();
  }
  int vec = pick_vec(D, x, dy, dx, gamma);
  const int vec2 = pick_vec(D, dyb, addend, dx2, nullptr);
  if (vec2 < vec) vec = vec2;
  if (vec == 4)
    LN_DISPATCH(bwd_launch, 4, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, dr, dx2, dr2, dyb, addend, s);
  if (vec == 2)
    LN_DISPATCH(bwd_launch, 2, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, dr, dx2, dr2, dyb, addend, s);
  LN_DISPATCH(bwd_launch, 1, dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, dr, dx2, dr2, dyb, addend, s);
}

extern "C" int mesm_layernorm_bwd2(const float* dy, const float* x, const float* gamma,
                                   const float* mean, const float* rstd, float* dx,
                                   float* dgamma, float* dbeta, int64_t rows, int32_t D,
                                   int32_t accumulate_dx, float drop_p, uint32_t drop_seed,
                                   const uint32_t* seed_offset, float* dx2, float drop2_p,
                                   uint32_t drop2_seed, void* stream) {
  return mesm_layernorm_bwd3(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, drop_p, drop_seed,
                             seed_offset, dx2, drop2_p, drop2_seed, nullptr, nullptr, stream);
}

extern "C" int mesm_layernorm_bwd(const float* dy, const float* x, const float* gamma,
                                  const float* mean, const float* rstd, float* dx,
                                  float* dgamma, float* dbeta, int64_t rows, int32_t D,
                                  int32_t accumulate_dx, float drop_p, uint32_t drop_seed,
                                  const uint32_t* seed_offset, void* stream) {
  return mesm_layernorm_bwd2(dy, x, gamma, mean, rstd, dx, dgamma, dbeta, rows, D, accumulate_dx, drop_p,
                             drop_seed, seed_offset, nullptr, 0.0f, 0u, stream);
}

// ------------------------------------------------------------------------------------------------
// Grouped launches: independent LayerNorms of one launch phase (the lockstep chains of ops.py) share ONE kernel.
// Problems of the two common classes (D <= 256 or D <= 512, D % 4 == 0, 16-byte aligned: one / two float4 chunks per
// lane) go into groups of up to LN_GROUP_MAX per class; anything else runs through the plain entry points.
namespace {

// 0: runs alone; 1: D <= 256 (one float4 chunk per lane); 2: D <= 512 (two chunks: the 512-d text features' input
// LayerNorms -- words, sentences, the two learned tokens -- used to be four launches forward and four backward)
int ln_group_class(const MesmLnArgs& a, bool bwd) {
  if (a.D > 512 || a.D % 4 != 0 || a.rows <= 0) return 0;
  const void* ps[] = {a.x, a.gamma, a.beta, a.y, a.add, a.y2, a.dy, a.dx, a.dx2, a.dyb, a.addend};
  for (const void* q : ps)
    if (q != nullptr && ((uintptr_t)q % 16) != 0) return 0;
  // parameter gradients only: the column-parallel kernel, unless the problem is small enough to ride in a group
  if (bwd && a.dx == nullptr && (a.rows > 2048 || a.accumulate_dx || a.dx2 || a.addend || a.relu_in)) return 0;
  return a.D <= 256 ? 1 : 2;
}

// twins: two LayerNorm problems of one phase over the SAME x / gamma / beta with different dropout masks (the raw video
// features projected for the main path and for the SS-MESM copy of the batch when the copy is the batch itself)
bool ln_fwd_twins(const MesmLnArgs& a, const MesmLnArgs& b) {
  return a.x == b.x && a.gamma == b.gamma && a.beta == b.beta && a.rows == b.rows && a.D == b.D && a.eps == b.eps && !a.add &&
         !b.add && !a.y2 && !b.y2 && a.y != b.y && a.D % 4 == 2 && a.D >= 1024 && a.D + 3 <= 17 * 256 &&
         pick_vec(4, a.x, a.y, a.gamma, a.beta) == 4 && pick_vec(4, b.y, nullptr, nullptr, nullptr) == 4;
}

int ln_fwd_twin_launch(const MesmLnArgs& a, const MesmLnArgs& b, hipStream_t s) {
  const LnDrop dr = make_drop(a.drop_p, a.drop_seed, a.seed_offset);
  LnTwin tw;
  tw.y = b.y; tw.mean = b.mean; tw.rstd = b.rstd;
  tw.dr = make_drop(b.drop_p, b.drop_seed, b.seed_offset);
  int64_t blocks = (a.rows + LN_WAVES - 1) / LN_WAVES;
  if (blocks > 4096) blocks = 4096;
  const int need = (a.D + 3 + 255) / 256;
  if (need <= 8)
    hipLaunchKernelGGL(ln_fwd_twin_kernel<8>, dim3((unsigned)blocks), dim3(LN_THREADS), 0, s, a.x, a.gamma, a.beta, a.y, a.mean,
                       a.rstd, a.rows, a.D, a.eps, dr, tw);
  else if (need <= 12)
    hipLaunchKernelGGL(ln_fwd_twin_kernel<12>, dim3((unsigned)blocks), dim3(LN_THREADS), 0, s, a.x, a.gamma, a.beta, a.y,
                       a.mean, a.rstd, a.rows, a.D, a.eps, dr, tw);
  else
    hipLaunchKernelGGL(ln_fwd_twin_kernel<17>, dim3((unsigned)blocks), dim3(LN_THREADS), 0, s, a.x, a.gamma, a.beta, a.y,
                       a.mean, a.rstd, a.rows, a.D, a.eps, dr, tw);
  return mesm_launch_status();
}

bool ln_bwd_param_twins(const MesmLnArgs& a, const MesmLnArgs& b) {
  return !a.dx && !b.dx && a.x == b.x && a.dgamma == b.dgamma && a.dbeta == b.dbeta && a.rows == b.rows && a.D == b.D &&
         a.dy != b.dy && !a.dyb && !b.dyb && !a.addend && !b.addend && !a.dx2 && !b.dx2 && a.D > 512;
}

int ln_bwd_param_twin_launch(const MesmLnArgs& a, const MesmLnArgs& b, hipStream_t s) {
  const int cb = (a.D + 63) / 64;
  int rb = (int)((1024 + cb - 1) / cb);  // (as mesm_layernorm_bwd3: about 1024 workgroups, of 32 rows or more)
  if ((int64_t)rb * 32 > a.rows) rb = (int)((a.rows + 31) / 32);
  const int rpb = (int)((a.rows + rb - 1) / rb);
  rb = (int)((a.rows + rpb - 1) / rpb);
  hipLaunchKernelGGL(ln_bwd_params_kernel, dim3(cb, rb), dim3(256), 0, s, a.dy, a.x, a.mean, a.rstd, a.dgamma, a.dbeta, a.rows,
                     a.D, rpb, make_drop(a.drop_p, a.drop_seed, a.seed_offset), b.dy,
                     make_drop(b.drop_p, b.drop_seed, b.seed_offset));
  return mesm_launch_status();
}

LnProb ln_prob(const MesmLnArgs& a) {
  LnProb q;
  q.x = a.x; q.gamma = a.gamma; q.beta = a.beta; q.y = a.y; q.mean = a.mean; q.rstd = a.rstd;
  q.rows = a.rows; q.D = a.D; q.eps = a.eps;
  q.dr = make_drop(a.drop_p, a.drop_seed, a.seed_offset);
  q.add = a.add; q.y2 = a.y2;
  q.dy = a.dy; q.dx = a.dx; q.dgamma = a.dgamma; q.dbeta = a.dbeta; q.accumulate_dx = a.accumulate_dx;
  q.dx2 = a.dx2; q.dr2 = make_drop(a.drop2_p, a.drop2_seed, a.seed_offset);
  q.dyb = a.dyb; q.addend = a.addend;
  q.relu_in = a.relu_in;
  return q;
}

}  // namespace

extern "C" int mesm_layernorm_fwd_group(const MesmLnArgs* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 64) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  LnGroup g[2];  // one per class
  g[0].n = g[1].n = 0;
  g[0].start[0] = g[1].start[0] = 0;
  int rc = MESM_OK;
  auto flush = [&](int c) {
    if (g[c].n == 0) return;
    if (c == 0)
      hipLaunchKernelGGL((ln_fwd_group_kernel<4, 1>), dim3((unsigned)g[c].start[g[c].n]), dim3(LN_THREADS), 0, s, g[c]);
    else
      hipLaunchKernelGGL((ln_fwd_group_kernel<4, 2>), dim3((unsigned)g[c].start[g[c].n]), dim3(LN_THREADS), 0, s, g[c]);
    rc = mesm_launch_status();
    g[c].n = 0;
  };
  int ngroupable[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) ngroupable[ln_group_class(list[i], false)]++;
  bool done[64] = {};
  for (int i = 0; i < n && rc == MESM_OK; ++i) {
    if (done[i]) continue;
    const MesmLnArgs& a = list[i];
    if ((a.add == nullptr) != (a.y2 == nullptr) || a.drop_p < 0.f || a.drop_p >= 1.f) return MESM_EINVAL;
    const int cls = ln_group_class(a, false);
    if (cls == 0 && a.x && a.gamma && a.beta && a.y && a.mean && a.rstd) {
      int j = i + 1;
      while (j < n && !(ln_group_class(list[j], false) == 0 && list[j].y && list[j].mean && list[j].rstd &&
                        list[j].drop_p >= 0.f && list[j].drop_p < 1.f && ln_fwd_twins(a, list[j])))
        ++j;
      if (j < n) {
        done[j] = true;
        rc = ln_fwd_twin_launch(a, list[j], s);
        continue;
      }
    }
    if (cls && ngroupable[cls] >= 2) {
      if (!a.x || !a.gamma || !a.beta || !a.y || !a.mean || !a.rstd) return MESM_EINVAL;
      int64_t blocks = (a.rows + LN_WAVES - 1) / LN_WAVES;
      if (blocks > 4096) blocks = 4096;
      LnGroup& gg = g[cls - 1];
      gg.p[gg.n] = ln_prob(a);
      gg.start[gg.n + 1] = gg.start[gg.n] + (int)blocks;
      if (++gg.n == LN_GROUP_MAX) flush(cls - 1);
    } else {
      rc = mesm_layernorm_fwd2(a.x, a.gamma, a.beta, a.y, a.mean, a.rstd, a.rows, a.D, a.eps, a.drop_p, a.drop_seed,
                               a.seed_offset, a.add, a.y2, stream);
    }
  }
  for (int c = 0; c < 2 && rc == MESM_OK; ++c) flush(c);
  return rc;
}

extern "C" int mesm_layernorm_bwd_group(const MesmLnArgs* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 64) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  LnGroup g[2];
  g[0].n = g[1].n = 0;
  g[0].start[0] = g[1].start[0] = 0;
  int rc = MESM_OK;
  auto flush = [&](int c) {
    if (g[c].n == 0) return;
    if (c == 0)
      hipLaunchKernelGGL((ln_bwd_group_kernel<4, 1>), dim3((unsigned)g[c].start[g[c].n]), dim3(LNB_THREADS),
                         (size_t)LNB_WAVES * 256 * sizeof(float), s, g[c]);
    else
      hipLaunchKernelGGL((ln_bwd_group_kernel<4, 2>), dim3((unsigned)g[c].start[g[c].n]), dim3(LNB_THREADS),
                         (size_t)LNB_WAVES * 512 * sizeof(float), s, g[c]);
    rc = mesm_launch_status();
    g[c].n = 0;
  };
  int ngroupable[3] = {0, 0, 0};
  for (int i = 0; i < n; ++i) ngroupable[ln_group_class(list[i], true)]++;
  bool done[64] = {};
  for (int i = 0; i < n && rc == MESM_OK; ++i) {
    if (done[i]) continue;
    const MesmLnArgs& a = list[i];
    if (a.drop_p < 0.f || a.drop_p >= 1.f || a.drop2_p < 0.f || a.drop2_p >= 1.f) return MESM_EINVAL;
    const int cls = ln_group_class(a, true);
    if (cls == 0 && !a.dx && a.dy && a.x && a.mean && a.rstd && a.dgamma && a.dbeta && !a.accumulate_dx) {
      int j = i + 1;
      while (j < n && !(ln_group_class(list[j], true) == 0 && list[j].dy && list[j].drop_p >= 0.f && list[j].drop_p < 1.f &&
                        !list[j].accumulate_dx && ln_bwd_param_twins(a, list[j])))
        ++j;
      if (j < n) {
        done[j] = true;
        rc = ln_bwd_param_twin_launch(a, list[j], s);
        continue;
      }
    }
    if (a.relu_in && cls == 0) return MESM_EINVAL;  // the ReLU mask exists on the grouped kernels only
    if (cls && (ngroupable[cls] >= 2 || a.relu_in)) {
      if (!a.dy || !a.x || !a.gamma || !a.mean || !a.rstd || !a.dgamma || !a.dbeta) return MESM_EINVAL;
      int64_t blocks = (a.rows + LNB_WAVES - 1) / LNB_WAVES;
      const int64_t cap = a.rows >= 4000 ? MESM_LNB_CAP : MESM_LNB_CAP / 2;
      if (blocks > cap) blocks = cap;
      LnGroup& gg = g[cls - 1];
      gg.p[gg.n] = ln_prob(a);
      gg.start[gg.n + 1] = gg.start[gg.n] + (int)blocks;
      if (++gg.n == LN_GROUP_MAX) flush(cls - 1);
    } else {
      rc = mesm_layernorm_bwd3(a.dy, a.x, a.gamma, a.mean, a.rstd, a.dx, a.dgamma, a.dbeta, a.rows, a.D, a.accumulate_dx,
                               a.drop_p, a.drop_seed, a.seed_offset, a.dx2, a.drop2_p, a.drop2_seed, a.dyb, a.addend,
                               stream);
    }
  }
  for (int c = 0; c < 2 && rc == MESM_OK; ++c) flush(c);
  return rc;
}
