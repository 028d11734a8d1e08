// Device code shared by the GEMM translation units (gemm.hip: exact-f32 kernels; gemm_px.hip: split-bf16 plane kernels):
// the tile-order map, the staged epilogue over MesmGemmArgs, the K-tail helpers of the LDS-DMA kernels.
#pragma once
#include <cstddef>
#include "common.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int NTHREADS = 256;
constexpr int BK_MAX = 64;  // split-K chunks and the tail granularity are multiples of this

struct XForm {
  int act;
  float slope;
  uint32_t thresh;  // 0 => no dropout
  uint32_t seed;
  float inv_keep;
  int64_t lld;  // logical row length for the dropout index
};

// PReLU slope gradient: every workgroup stores ONE partial sum (plain store) into the workspace
// slot of its linear block id and a 1-workgroup kernel adds them into the parameter gradient.
// (One float atomic per wave on the single dslope address cost +45 us per GEMM: same-address
// atomics serialise at the memory side.)
__device__ __forceinline__ int64_t linear_block() {
  return ((int64_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
}

// XCD-aware tile order.  Workgroups are dealt round-robin to the 8 XCDs (private 4 MB L2 each) by
// their linear id.  With the natural (x = M-tile fastest) order the N-tiles of one M-tile land on 8
// different XCDs whenever gridDim.x % 8 != 0, so every A tile was fetched into up to 8 L2s (PMC: 4 x the
// algorithmic HBM-side traffic on the GEMMs).  Remap: XCD c owns a CONTIGUOUS range of logical tiles,
// and logical tiles run N-fastest, so all column tiles of a row tile share one L2.
__device__ __forceinline__ void xcd_tile(int lid, int mt, int nt, int& bx, int& by) {
  const int total = mt * nt;
  const int xcd = lid & 7, idx = lid >> 3;
  const int q = total >> 3, r = total & 7;
  const int t = xcd * q + (xcd < r ? xcd : r) + idx;  // bijection on [0, total)
  bx = t / nt;
  by = t - bx * nt;
}

// Split-K launches (grid = tiles x 1 x S, or a grouped launch's run of tiles x S workgroups): `lid` = tile + tiles * slice
// as launched.  With xcd_tile() alone every XCD walks ALL k-slices of its tile range, so the operand that is shared
// between its row tiles is fetched into all 8 L2s (PMC, round 3: a 1024 x 256 x 4800 weight gradient at S = 4 pulled
// 72 MB for 25 MB of operands).  Here an XCD works on ONE slice: S in {2, 4, 8}: 8 / S XCDs share a slice and split its
// tiles in contiguous ranges; S a multiple of 8: XCD c takes the slices = c (mod 8).  `MESM_XCD_Z` = 0 keeps the old map
// (A/B builds).  The map needs the tile count to divide evenly; otherwise the old one applies.
#ifndef MESM_XCD_Z
#define MESM_XCD_Z 1
#endif
__device__ __forceinline__ void xcd_tile_z(int lid, int mt, int nt, int S, int& bx, int& by, int& z) {
  const int T = mt * nt;
  if (MESM_XCD_Z && S > 1) {
    const int xcd = lid & 7, idx = lid >> 3;
    if (S <= 8 && (8 % S) == 0 && T % (8 / S) == 0) {
      const int G = 8 / S;
      z = xcd / G;
      const int t = (xcd - z * G) * (T / G) + idx;
      if (nt > mt) {  // the XCDs of a slice split the LONGER tile axis: each then reads all of the smaller operand only
        by = t / mt;
        bx = t - by * mt;
      } else {
        bx = t / nt;
        by = t - bx * nt;
      }
      return;
    }
    if ((S & 7) == 0) {
      const int zi = idx / T, t = idx - zi * T;
      z = zi * 8 + xcd;
      bx = t / nt;
      by = t - bx * nt;
      return;
    }
  }
  z = S > 1 ? lid / T : 0;
  xcd_tile(lid - z * T, mt, nt, bx, by);
}

__device__ __forceinline__ void dslope_store(const MesmGemmArgs& p, float part, float* sh4, int64_t slot) {
  part = wave_sum(part);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh4[wave] = part;
  __syncthreads();
  if (threadIdx.x == 0) {
    float t = 0.0f;
    for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh4[w];
    p.dslope_ws[slot] = t;
  }
}

inline bool aligned_to(const void* p, size_t b) { return p == nullptr || ((uintptr_t)p % b) == 0; }

// Reduce indices [gemm_kmain, K) are NOT staged by the LDS-DMA kernels (wstage / wstage64 / lds64):
//  * a reduce-contiguous operand is staged in 16-byte chunks along k: the main loop ends at K & ~3;
//  * an outer-contiguous operand whose outer extent is not a multiple of 4 has one chunk per reduce row
//    that runs 1-3 floats into the NEXT reduce row (harmless: those outer positions are never stored);
//    for the last reduce row that would be past the end of the matrix, so the main loop stops before it.
// The remaining 1-4 indices are added to the accumulators by scalar loads (tail_accumulate).
__device__ __forceinline__ int gemm_kmain(const MesmGemmArgs& p) {
  const bool a_red = p.a_layout == MESM_LAYOUT_REDUCE_CONTIG, b_red = p.b_layout == MESM_LAYOUT_REDUCE_CONTIG;
  int km = p.K;
  if (a_red || b_red) km &= ~3;
  if ((!a_red && (p.M & 3)) || (!b_red && (p.N & 3))) {
    const int lim = (p.K - 1) & ~3;
    km = km < lim ? km : lim;
  }
  return km;
}

// operand element (outer index o, reduce index k) with the operand transform of the main loop
template <int LAYOUT, bool XF>
__device__ __forceinline__ float tail_elem(const float* __restrict__ base, int64_t ld, int o, int k, const XForm& xf) {
  float x = LAYOUT == MESM_LAYOUT_REDUCE_CONTIG ? base[(int64_t)o * ld + k] : base[(int64_t)k * ld + o];
  if (XF) {
    x = mesm_act(x, xf.act, xf.slope);
    if (xf.thresh)
      x = mesm_dropout_apply(x, (uint32_t)(LAYOUT == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)o * xf.lld + k : (int64_t)k * xf.lld + o),
                             xf.seed, xf.thresh, xf.inv_keep);
  }
  return x;
}

// t[i] += sum over k in [km, K) of A(rbase + RO(i), k) * B(col, k)   (rows / columns clamped into the matrix)
template <int NV, int LA, int LB, bool XF, typename RowOff>
__device__ __forceinline__ void tail_accumulate(const MesmGemmArgs& p, float (&t)[NV], int rbase, int col, int km,
                                                const XForm& xa, const XForm& xb, RowOff RO) {
  const int colc = col < p.N ? col : p.N - 1;
  for (int k = km; k < p.K; ++k) {
    const float y = tail_elem<LB, XF>(p.B, p.ldb, colc, k, xb);
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row = rbase + RO(i);
      row = row < p.M ? row : p.M - 1;
      t[i] += tail_elem<LA, XF>(p.A, p.lda, row, k, xa) * y;
    }
  }
}

// Epilogue of NV accumulator values per lane: value i belongs to row rbase + RO(i) (RO = row offset
// table of the MFMA 32x32 accumulator layout) and column col.  Written in STAGES over the whole
// register array -- scale+bias, activation, dropout, activation-gradient, residual / read-modify-write,
// store -- so that every run-time flag of MesmGemmArgs is tested once per wave, not once per element, and
// the side loads of a stage are all in flight together; row addresses are one per-lane 64-bit base plus
// wave-uniform multiples of the leading dimension.  (In-kernel stamps, tools/l64_trace.py: the
// per-element form took ~8,000 cycles per 32x32 tile, as long as 3.5 k-tiles of MFMA work.)
// FULL: the wave's 32 x 32 tile lies inside C, no bounds handling at all.
template <int NV, bool FULL, typename RowOff>
__device__ __forceinline__ float staged_epilogue(const MesmGemmArgs& p, float (&t)[NV], int rbase, int col,
                                                 float slope, uint32_t seed_off, bool first_split, RowOff RO) {
  const bool colok = FULL || col < p.N;
  const int colc = colok ? col : p.N - 1;
  bool ok[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) ok[i] = FULL || (colok && rbase + RO(i) < p.M);
  // element offset of value i in a row-major side matrix with leading dimension ld, clamped into the matrix
  auto off = [&](int i, int64_t lane_base, int64_t ld) -> int64_t {
    const int64_t o = lane_base + (int64_t)RO(i) * ld;
    if (FULL) return o;
    const int64_t last = (int64_t)(p.M - 1) * ld + colc;
    return o < last ? o : last;
  };

#ifdef MESM_EPI_NO_SIDE  // probe build: side operands replaced by constants (how much of a launch is their latency?)
  const float bias_v = (p.bias != nullptr && first_split) ? 0.5f : 0.0f;
#else
  const float bias_v = (p.bias != nullptr && first_split) ? p.bias[colc] : 0.0f;
#endif
  const float sc = p.out_scale;
#ifdef MESM_LN_PROBE
  // Probe build (tools/probe/ln_stats.py): LayerNorm WITHOUT a launch of its own, the decomposition of the
  // round-5 review -- the producing GEMM's epilogue accumulates the row statistics of its output (reserved0 == 1, below the
  // residual add), the CONSUMING GEMM runs on the raw tensor with weights pre-scaled by gamma and normalises in its epilogue:
  //   LN(x) W^T + b = rstd_r (x W'^T - mu_r c) + d,   W' = W diag(gamma), c = W gamma, d = W beta + b     (reserved0 == 2)
  // stats = dslope_ws: (sum, sum of squares) per row of a D = ldaux wide tensor; c = aux (a vector here); d = bias.
  if (p.reserved0 == 2) {
    const float invD = 1.0f / (float)p.ldaux;
    const float cn = p.aux[colc];
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      int row = rbase + RO(i);
      row = row < p.M ? row : p.M - 1;
      const float2 st = reinterpret_cast<const float2*>(p.dslope_ws)[row];
      const float mu = st.x * invD;
      const float rs = rsqrtf(fmaxf(st.y * invD - mu * mu, 0.0f) + 1e-5f);
      t[i] = rs * (t[i] - mu * cn) + bias_v;
    }
  } else
#endif
#pragma unroll
  for (int i = 0; i < NV; ++i) t[i] = t[i] * sc + bias_v;
  if (p.pre_out != nullptr) {  // second output: the pre-activation (what the backward's e_actgrad reads as aux)
    float* pp = p.pre_out + ((int64_t)rbase * p.ldpre + col);
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) mesm_store_wt(pp + (int64_t)RO(i) * p.ldpre, t[i]);
  }
  if (p.e_act != MESM_ACT_NONE) {
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = mesm_act(t[i], p.e_act, slope);
  }
  if (p.e_drop_p > 0.f) {
    const uint32_t thresh = mesm_drop_threshold(p.e_drop_p);
    const float inv_keep = 1.0f / (1.0f - p.e_drop_p);
    const uint32_t seed = p.e_drop_seed + seed_off;
    // dense index of C, modulo 2^32 (e_drop_row0: this launch computes rows [row0, row0 + M) of a taller C)
    const uint32_t idx0 = (uint32_t)(rbase + p.e_drop_row0) * (uint32_t)p.N + (uint32_t)col;
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] = mesm_dropout_apply(t[i], idx0 + (uint32_t)RO(i) * (uint32_t)p.N, seed, thresh, inv_keep);
  }
  float dslope_part = 0.0f;
  if (p.e_actgrad != MESM_ACT_NONE) {
    const int64_t lb = (int64_t)rbase * p.ldaux + colc;
    float z[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) z[i] = p.aux[off(i, lb, p.ldaux)];
    if (p.e_actgrad == MESM_ACT_RELU) {
#pragma unroll
      for (int i = 0; i < NV; ++i) t[i] = z[i] > 0.0f ? t[i] : 0.0f;
    } else {
#pragma unroll
      for (int i = 0; i < NV; ++i)
        if (z[i] <= 0.0f) {
          if (ok[i]) dslope_part += t[i] * z[i];
          t[i] *= slope;
        }
    }
  }
  const bool use_res = p.residual != nullptr && first_split;
  const bool rmw = p.accumulate == 1;
  if (use_res || rmw) {
    float add[NV];
#pragma unroll
    for (int i = 0; i < NV; ++i) add[i] = 0.0f;
    if (use_res) {
      const int64_t lb = (int64_t)rbase * p.ldr + colc;
#pragma unroll
#ifdef MESM_EPI_NO_SIDE
      for (int i = 0; i < NV; ++i) add[i] = 0.25f;
#else
      for (int i = 0; i < NV; ++i) add[i] = p.residual[off(i, lb, p.ldr)];
#endif
    }
    if (rmw) {
      const int64_t lb = (int64_t)rbase * p.ldc + colc;
#pragma unroll
      for (int i = 0; i < NV; ++i) add[i] += p.C[off(i, lb, p.ldc)];
    }
#pragma unroll
    for (int i = 0; i < NV; ++i) t[i] += add[i];
  }
#ifdef MESM_LN_PROBE
  if (p.reserved0 == 1) {  // row statistics of the finished output: 32 columns of a row sit in the 32 lanes of a half wave
    const int lane_ = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
      const float v = ok[i] ? t[i] : 0.0f;
      float s1 = add_xor16(sum_within<16>(v));
      float s2 = add_xor16(sum_within<16>(v * v));
      const int row = rbase + RO(i);
      if ((lane_ & 31) == 0 && row < p.M) {
        atomicAdd(p.dslope_ws + 2 * (int64_t)row, s1);
        atomicAdd(p.dslope_ws + 2 * (int64_t)row + 1, s2);
      }
    }
  }
#endif
  float* cp = p.C + ((int64_t)rbase * p.ldc + col);
  if (p.accumulate == 2) {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) atomicAdd(cp + (int64_t)RO(i) * p.ldc, t[i]);
  } else {
#pragma unroll
    for (int i = 0; i < NV; ++i)
      if (ok[i]) mesm_store_wt(cp + (int64_t)RO(i) * p.ldc, t[i]);  // (write-through: common.hpp)
  }
  return dslope_part;
}

// Epilogue shared by the k-split kernels: the four waves hold partial sums of the same 32x32 tile;
// they meet in LDS (Red: 4 x 16 x 64 floats) and wave w takes accumulator registers [4w, 4w+4)
// (rows 4h + rr + 8w) through the staged epilogue.
template <int LA, int LB, bool XF, int NW = 4>
__device__ __forceinline__ void ksplit_epilogue(const MesmGemmArgs& p, const f32x16& acc, float* Red, int m0,
                                                int n0, float slope, uint32_t seed_off, int bz, int64_t slot,
                                                int km, const XForm& xa, const XForm& xb) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int li = lane & 31, h = lane >> 5;
  constexpr int NV = 16 / NW;  // accumulator registers per wave after the cross-wave sum
  float vals[NV];
  if (NW > 1) {
#pragma unroll
    for (int r = 0; r < 16; ++r) Red[(wave * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    const int r0 = wave * NV;
#pragma unroll
    for (int rr = 0; rr < NV; ++rr) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < NW; ++w) t += Red[(w * 16 + r0 + rr) * 64 + lane];
      vals[rr] = t;
    }
  } else {
#pragma unroll
    for (int rr = 0; rr < NV; ++rr) vals[rr] = acc[rr];
  }
  const bool first_split = (p.split_k <= 1) || (bz == 0);
  // register r <-> row 4h + (r & 3) + 8 (r >> 2); this wave owns registers [wave * NV, wave * NV + NV)
  const int r0 = wave * NV;
  const int rbase = m0 + 4 * h + (r0 & 3) + 8 * (r0 >> 2);
  auto RO = [](int i) { return NV <= 4 ? i : (i & 3) + 8 * (i >> 2); };
  if (first_split && km < p.K) tail_accumulate<NV, LA, LB, XF>(p, vals, rbase, n0 + li, km, xa, xb, RO);
  float dslope_part;
  if (m0 + 32 <= p.M && n0 + 32 <= p.N)
    dslope_part = staged_epilogue<NV, true>(p, vals, rbase, n0 + li, slope, seed_off, first_split, RO);
  else
    dslope_part = staged_epilogue<NV, false>(p, vals, rbase, n0 + li, slope, seed_off, first_split, RO);
  if (p.e_actgrad == MESM_ACT_PRELU && p.dslope) dslope_store(p, dslope_part, Red, slot);
}

// Epilogue of ONE wave-owned 32 x 32 tile held in 16 accumulator registers (register r <-> row
// 4h + (r & 3) + 8 (r >> 2), column lane & 31).
template <int LA, int LB, bool XF>
__device__ __forceinline__ void tile16_epilogue(const MesmGemmArgs& p, const f32x16& acc, int row0, int col0,
                                                float slope, uint32_t seed_off, int bz, float* sh4,
                                                int64_t slot, int km, const XForm& xa, const XForm& xb) {
  const int lane = threadIdx.x & 63;
  const int li = lane & 31, h = lane >> 5;
  const bool first_split = (p.split_k <= 1) || (bz == 0);
  float t[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) t[i] = acc[i];
  auto RO = [](int i) { return (i & 3) + 8 * (i >> 2); };
  if (first_split && km < p.K) tail_accumulate<16, LA, LB, XF>(p, t, row0 + 4 * h, col0 + li, km, xa, xb, RO);
  float dslope_part;
  if (row0 + 32 <= p.M && col0 + 32 <= p.N)
    dslope_part = staged_epilogue<16, true>(p, t, row0 + 4 * h, col0 + li, slope, seed_off, first_split, RO);
  else
    dslope_part = staged_epilogue<16, false>(p, t, row0 + 4 * h, col0 + li, slope, seed_off, first_split, RO);
  if (p.e_actgrad == MESM_ACT_PRELU && p.dslope) dslope_store(p, dslope_part, sh4, slot);
}

// column-sum share of the tail reduce indices for output row gm (added by the lanes that own the atomics)
template <int LA, bool XF>
__device__ __forceinline__ float tail_colsum(const MesmGemmArgs& p, int gm, int km, const XForm& xa) {
  float c = 0.0f;
  if (gm < p.M)
    for (int k = km; k < p.K; ++k) c += tail_elem<LA, XF>(p.A, p.lda, gm, k, xa);
  return c;
}

struct Blk {
  int x, y, z;    // tile coordinates of this workgroup inside ITS problem
  int64_t slot;   // linear id inside its problem (dslope workspace slot)
};


}  // namespace

// host-side hooks implemented in gemm.hip: the PReLU slope-gradient partials of a launch are queued and carried by the
// next GEMM launch of the same stream (or reduced by mesm_gemm_flush_side); nblocks = workspace slots the launch wrote
int mesm_gemm_dslope_finish(const MesmGemmArgs& a, int64_t nblocks, hipStream_t s);
