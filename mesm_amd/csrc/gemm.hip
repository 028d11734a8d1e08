// Exact-f32 MFMA GEMM for gfx950 (v_mfma_f32_32x32x2_f32), LDS-tiled, with the
// prologue / epilogue fusions the MESM hot path needs (see include/mesm_gfx950.h).
//
// Tiling: 256 threads = 4 waves in a 2x2 grid; block tile BM x BN in {128x128, 64x64},
// K step 16.  Each wave owns a (BM/2)x(BN/2) sub-tile = TM x TN MFMA tiles of 32x32.
// LDS image of both operands is [k][outer] so that lane l of the 32x32x2 MFMA reads
// A[i = l&31][k = l>>5] / B[k = l>>5][j = l&31] with a conflict-free ds_read_b32
// (the two 32-lane halves read different k rows).  Operands are staged
// global -> registers -> LDS (a transpose is needed when the reduce index is the
// contiguous one, which LDS-DMA cannot do) and double-buffered: the global loads of
// tile t+1 are in flight while tile t is multiplied.
#include <cstdio>
#include <cstdlib>
#include <utility>
#include <vector>

#include <cstddef>
#include <algorithm>
#include "gemm_ws.hpp"


namespace {

int mesm_gemm_force_tile();  // MESM_GEMM_TILE / MESM_GEMM_BF16X, read once at load (definitions next to dispatch)
int mesm_gemm_bf16x();

__global__ __launch_bounds__(256) void dslope_reduce_kernel(const float* __restrict__ ws, int64_t n,
                                                            float* __restrict__ dst) {
  __shared__ float sh[4];
  float a = 0.0f;
  for (int64_t i = threadIdx.x; i < n; i += 256) a += ws[i];
  a = wave_sum(a);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = a;
  __syncthreads();
  if (threadIdx.x == 0) dst[0] += sh[0] + sh[1] + sh[2] + sh[3];
}

struct SidePending {
  const float* ws;
  float* dst;
  int n;
  hipStream_t stream;  // the stream the producing launch went to: only a launch on the SAME stream may carry it
};
std::vector<SidePending> g_side;
const bool g_side_on = getenv("MESM_DSLOPE_LAUNCH") == nullptr;  // MESM_DSLOPE_LAUNCH=1: one launch per reduction (A/B)

inline int dslope_finish_n(const MesmGemmArgs& a, int64_t n, hipStream_t s) {
  if (a.e_actgrad != MESM_ACT_PRELU || !a.dslope) return MESM_OK;
  if (g_side_on && n < (1 << 30)) {
    g_side.push_back({a.dslope_ws, a.dslope, (int)n, s});
    return MESM_OK;
  }
  hipLaunchKernelGGL(dslope_reduce_kernel, dim3(1), dim3(256), 0, s, a.dslope_ws, n, a.dslope);
  return mesm_launch_status();
}

inline int dslope_finish(const MesmGemmArgs& a, dim3 grid, hipStream_t s) {
  return dslope_finish_n(a, (int64_t)grid.x * grid.y * grid.z, s);
}

// up to four pending reductions for the launch that is about to be issued on stream s (stream order is what makes
// the partials visible to it: entries produced on another stream stay queued for mesm_gemm_flush_side)
inline SideRed take_side(hipStream_t s) {
  SideRed sr = {};
  for (size_t i = 0; i < g_side.size() && sr.count < 4;) {
    if (g_side[i].stream != s) { ++i; continue; }
    const SidePending e = g_side[i];
    g_side.erase(g_side.begin() + i);
    sr.ws[sr.count] = e.ws; sr.dst[sr.count] = e.dst; sr.n[sr.count] = e.n;
    ++sr.count;
  }
  return sr;
}

// every pending reduction with a launch of its own, each on the stream that produced its partials
inline int flush_side(hipStream_t) {
  for (const SidePending& e : g_side)
    hipLaunchKernelGGL(dslope_reduce_kernel, dim3(1), dim3(256), 0, e.stream, e.ws, (int64_t)e.n, e.dst);
  const bool any = !g_side.empty();
  g_side.clear();
  return any ? mesm_launch_status() : MESM_OK;
}

// One operand tile: ROWS (outer index) x BK (reduce index), staged through registers.
// Loads are written branch-free so that every global load of a k-step is in flight before
// the first wait: full tiles use unguarded vector loads (rows clamped with a select), the
// single K-tail tile uses clamped scalar loads + selects.
template <int ROWS, int BK, int LAYOUT, int VEC, bool ADD>
struct Tile {
  static constexpr int NVEC = ROWS * BK / (VEC * NTHREADS);
  static_assert(NVEC >= 1, "tile too small for the block");
  float r[NVEC][VEC];
  float t[ADD ? NVEC : 1][VEC];

  __device__ __forceinline__ static void coords(int tid, int v, int& o, int& k) {
    int vid = tid + v * NTHREADS;
    if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
      constexpr int VPR = BK / VEC;
      o = vid / VPR;
      k = (vid % VPR) * VEC;
    } else {
      constexpr int VPR = ROWS / VEC;
      k = vid / VPR;
      o = (vid % VPR) * VEC;
    }
  }

  __device__ __forceinline__ static void load_vec(const float* __restrict__ p, float* dst) {
    if (VEC == 4) {
      float4 x = *reinterpret_cast<const float4*>(p);
      dst[0] = x.x; dst[1] = x.y; dst[2] = x.z; dst[3] = x.w;
    } else if (VEC == 2) {
      float2 x = *reinterpret_cast<const float2*>(p);
      dst[0] = x.x; dst[1] = x.y;
    } else {
      dst[0] = *p;
    }
  }

  // FULLK: every reduce index of the tile is < kend.  For OUTER_CONTIG operands the host
  // guarantees extent % VEC == 0, so a vector is either fully inside or fully outside.
  template <bool FULLK>
  __device__ __forceinline__ void load(const float* __restrict__ base,
                                       const float* __restrict__ add, int64_t ld, int o0,
                                       int extent, int k0, int kend, int tid) {
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
      int o, k;
      coords(tid, v, o, k);
      if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
        int go = o0 + o;
        go = go < extent ? go : extent - 1;
        const int gk = k0 + k;
        const int64_t rowoff = (int64_t)go * ld;
        if (FULLK) {
          load_vec(base + rowoff + gk, r[v]);
          if (ADD) load_vec(add + rowoff + gk, t[v]);
        } else {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            const int kk = gk + e;
            const int kc = kk < kend ? kk : kend - 1;
            float x = base[rowoff + kc];
            r[v][e] = kk < kend ? x : 0.0f;
            if (ADD) {
              float y = add[rowoff + kc];
              t[v][e] = kk < kend ? y : 0.0f;
            }
          }
        }
      } else {
        const int gk = k0 + k;
        const int kc = (FULLK || gk < kend) ? gk : kend - 1;
        int go = o0 + o;
        go = (go + VEC <= extent) ? go : extent - VEC;
        const int64_t off = (int64_t)kc * ld + go;
        load_vec(base + off, r[v]);
        if (ADD) load_vec(add + off, t[v]);
        if (!FULLK) {
#pragma unroll
          for (int e = 0; e < VEC; ++e) {
            r[v][e] = gk < kend ? r[v][e] : 0.0f;
            if (ADD) t[v][e] = gk < kend ? t[v][e] : 0.0f;
          }
        }
      }
    }
  }

  // addend + activation / dropout on the staged registers.  OUTER_IS_ROW: logical index is
  // outer*lld + reduce (operand A), otherwise reduce*lld + outer (operand B).
  template <bool OUTER_IS_ROW>
  __device__ __forceinline__ void finish(const XForm& xf, int o0, int k0, int tid) {
    if (ADD) {
#pragma unroll
      for (int v = 0; v < NVEC; ++v)
#pragma unroll
        for (int e = 0; e < VEC; ++e) r[v][e] += t[v][e];
    }
    if (xf.act == MESM_ACT_NONE && xf.thresh == 0) return;
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
      int o, k;
      coords(tid, v, o, k);
#pragma unroll
      for (int e = 0; e < VEC; ++e) {
        int go = o0 + o + (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG ? 0 : e);
        int gk = k0 + k + (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG ? e : 0);
        float x = mesm_act(r[v][e], xf.act, xf.slope);
        if (xf.thresh) {
          int64_t idx = OUTER_IS_ROW ? (int64_t)go * xf.lld + gk : (int64_t)gk * xf.lld + go;
          x = mesm_dropout_apply(x, (uint32_t)idx, xf.seed, xf.thresh, xf.inv_keep);
        }
        r[v][e] = x;
      }
    }
  }

  // LDS image: S[k][outer], row stride `stride` floats.
  __device__ __forceinline__ void store(float* __restrict__ S, int stride, int tid) const {
#pragma unroll
    for (int v = 0; v < NVEC; ++v) {
      int o, k;
      coords(tid, v, o, k);
      if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) S[(k + e) * stride + o] = r[v][e];
      } else {
        float* d = S + k * stride + o;
        if (VEC == 4) {
          *reinterpret_cast<float4*>(d) = make_float4(r[v][0], r[v][1], r[v][2], r[v][3]);
        } else if (VEC == 2) {
          *reinterpret_cast<float2*>(d) = make_float2(r[v][0], r[v][1]);
        } else {
          d[0] = r[v][0];
        }
      }
    }
  }
};

// ADD: 0 = no addend, 1 = A2 present, 2 = B2 present
template <int BM, int BN, int BK, int LA, int LB, int VEC, int ADD>
__global__ __launch_bounds__(NTHREADS) void gemm_f32_kernel(const MesmGemmArgs p) {
  // LDS row padding: +2 keeps the transposing ds_write_b32 of reduce-contiguous operands
  // conflict-free; +4 for outer-contiguous operands keeps rows 16-byte aligned for ds_write_b128
  // and avoids an exact power-of-two row stride (with stride 32 the 32x32 k-split configuration
  // read zeros for k >= 56 at columns 27/31 on gfx950 / ROCm 7.2 -- not understood, reproducible
  // gone with any padded stride).
  constexpr int PA = (LA == MESM_LAYOUT_REDUCE_CONTIG) ? 2 : 4;
  constexpr int PB = (LB == MESM_LAYOUT_REDUCE_CONTIG) ? 2 : 4;
  constexpr int SA = BM + PA;
  constexpr int SB = BN + PB;
  // wave grid: 2x2 over the (M, N) tile, or 1x1x4 over the k-pairs of a 32x32 tile (WGK = 4):
  // the four waves then accumulate partial products of the SAME 32x32 tile and are summed
  // through LDS before the epilogue -- 4x more, 4x shorter workgroups for the small GEMMs.
  constexpr int WGK = (BM == 32) ? 4 : 1;
  constexpr int WGM = (WGK == 1) ? 2 : 1, WGN = (WGK == 1) ? 2 : 1;
  constexpr int WM = BM / WGM, WN = BN / WGN;
  constexpr int TM = WM / 32, TN = WN / 32;
  constexpr int RN = 16 / WGK;  // accumulator registers each wave takes through the epilogue
  static_assert(WGK == 1 || (TM == 1 && TN == 1), "k-split waves own one 32x32 tile");

  __shared__ __attribute__((aligned(16))) float As[2][BK * SA];
  __shared__ __attribute__((aligned(16))) float Bs[2][BK * SB];
  __shared__ float Red[WGK > 1 ? WGK * 16 * 64 : 4];  // partial tiles of the k-split waves

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = tid >> 6;
  const int wn = wave % WGN, wm = (wave / WGN) % WGM, wk = wave / (WGN * WGM);
  const int m0 = blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  const int KM = p.K;  // this kernel stages every reduce index itself
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = blockIdx.z * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (blockIdx.z > 0) return;
      kbeg = kend = KM;
    }
  }

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;  // dropout index = dense index of the operand AS STORED
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  Tile<BM, BK, LA, VEC, ADD == 1> ta;
  Tile<BN, BK, LB, VEC, ADD == 2> tb;

  f32x16 acc[TM][TN];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;

  // bias-gradient side product: column sums of the A tile over the reduce index.
  const bool do_colsum = (p.colsum != nullptr) && (blockIdx.y == 0);
  float csum = 0.0f;
  constexpr int CS_KSTEP = NTHREADS / BM;  // threads per A column
  const int cs_i = tid % BM;
  const int cs_k = tid / BM;

  // One k-tile: all LDS fragment reads are issued first (counted lgkmcnt waits let MFMA kk start
  // as soon as its own operands have landed), then the dependent MFMA chain runs back to back.
  auto compute = [&](int buf) {
    const float* __restrict__ a_s = As[buf];
    const float* __restrict__ b_s = Bs[buf];
    constexpr int NKK = BK / 2 / WGK;  // k-pairs per wave per tile
    float a[NKK][TM], b[NKK][TN];
#pragma unroll
    for (int q = 0; q < NKK; ++q) {
      const int krow = 2 * (q * WGK + wk) + (lane >> 5);
#pragma unroll
      for (int i = 0; i < TM; ++i) a[q][i] = a_s[krow * SA + wm * WM + i * 32 + (lane & 31)];
#pragma unroll
      for (int j = 0; j < TN; ++j) b[q][j] = b_s[krow * SB + wn * WN + j * 32 + (lane & 31)];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int q = 0; q < NKK; ++q) {
#pragma unroll
      for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[q][i], b[q][j], acc[i][j], 0, 0, 0);
    }
    if (do_colsum) {
#pragma unroll
      for (int k = 0; k < BK; k += CS_KSTEP) csum += a_s[(k + cs_k) * SA + cs_i];
    }
    __builtin_amdgcn_sched_barrier(0);
  };

  const int nfull = (kend - kbeg) / BK;
  const bool has_tail = ((kend - kbeg) % BK) != 0;
  const int ntiles = nfull + (has_tail ? 1 : 0);

  if (nfull > 0) {
    ta.template load<true>(p.A, p.A2, p.lda, m0, p.M, kbeg, kend, tid);
    tb.template load<true>(p.B, p.B2, p.ldb, n0, p.N, kbeg, kend, tid);
  } else {
    ta.template load<false>(p.A, p.A2, p.lda, m0, p.M, kbeg, kend, tid);
    tb.template load<false>(p.B, p.B2, p.ldb, n0, p.N, kbeg, kend, tid);
  }
  ta.template finish<LA == MESM_LAYOUT_REDUCE_CONTIG>(xa, m0, kbeg, tid);
  tb.template finish<LB == MESM_LAYOUT_REDUCE_CONTIG>(xb, n0, kbeg, tid);
  ta.store(As[0], SA, tid);
  tb.store(Bs[0], SB, tid);
  __syncthreads();

  int buf = 0;
  int kt = 0;
  // steady state: the next tile is a full one, its loads fly while this tile is multiplied
  for (; kt + 1 < nfull; ++kt) {
    const int knext = kbeg + (kt + 1) * BK;
    ta.template load<true>(p.A, p.A2, p.lda, m0, p.M, knext, kend, tid);
    tb.template load<true>(p.B, p.B2, p.ldb, n0, p.N, knext, kend, tid);
    __builtin_amdgcn_sched_barrier(0);  // keep the global loads ahead of the MFMA block
    compute(buf);
    ta.template finish<LA == MESM_LAYOUT_REDUCE_CONTIG>(xa, m0, knext, tid);
    tb.template finish<LB == MESM_LAYOUT_REDUCE_CONTIG>(xb, n0, knext, tid);
    ta.store(As[buf ^ 1], SA, tid);
    tb.store(Bs[buf ^ 1], SB, tid);
    __syncthreads();
    buf ^= 1;
  }
  if (kt + 1 < ntiles) {  // one guarded K-tail tile follows the current one
    const int knext = kbeg + (kt + 1) * BK;
    ta.template load<false>(p.A, p.A2, p.lda, m0, p.M, knext, kend, tid);
    tb.template load<false>(p.B, p.B2, p.ldb, n0, p.N, knext, kend, tid);
    __builtin_amdgcn_sched_barrier(0);
    compute(buf);
    ta.template finish<LA == MESM_LAYOUT_REDUCE_CONTIG>(xa, m0, knext, tid);
    tb.template finish<LB == MESM_LAYOUT_REDUCE_CONTIG>(xb, n0, knext, tid);
    ta.store(As[buf ^ 1], SA, tid);
    tb.store(Bs[buf ^ 1], SB, tid);
    __syncthreads();
    buf ^= 1;
  }
  compute(buf);
  // The last MFMA must have written ALL its accumulator registers before they are read below:
  // hipcc (ROCm 7.2) was seen hoisting v_accvgpr_read of a[14:15] above its own hazard s_nop
  // next to a sched_barrier, so the wait states are spelled out (16-pass MFMA: >= 18).
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if (do_colsum) {
    int gi = m0 + cs_i;
    if (gi < p.M) atomicAdd(p.colsum + gi, csum);
  }

  // ---- k-split waves: sum the four partial 32x32 tiles through LDS; wave w keeps registers
  // [RN*w, RN*w + RN) of the sum and takes them through the epilogue ----
  float vals[TM][TN][RN];
  int r0 = 0;
  if (WGK > 1) {
    float* red = Red;
#pragma unroll
    for (int r = 0; r < 16; ++r) red[(wk * 16 + r) * 64 + lane] = acc[0][0][r];
    __syncthreads();
    r0 = wk * RN;
#pragma unroll
    for (int rr = 0; rr < RN; ++rr) {
      float t = 0.0f;
#pragma unroll
      for (int w = 0; w < WGK; ++w) t += red[(w * 16 + r0 + rr) * 64 + lane];
      vals[0][0][rr] = t;
    }
  } else {
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
      for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) vals[i][j][rr] = acc[i][j][rr];
  }

  // ---- epilogue: all side loads of a 32x32 tile are issued before the first use ----
  const bool first_split = (p.split_k <= 1) || (blockIdx.z == 0);
  const uint32_t e_thresh = p.e_drop_p > 0.f ? mesm_drop_threshold(p.e_drop_p) : 0u;
  const float e_inv_keep = 1.0f / (1.0f - p.e_drop_p);
  const bool use_bias = p.bias != nullptr && first_split;
  const bool use_res = p.residual != nullptr && first_split;
  const bool use_aux = p.e_actgrad != MESM_ACT_NONE;
  const bool rmw = p.accumulate == 1;
  float dslope_part = 0.0f;
#pragma unroll
  for (int i = 0; i < TM; ++i) {
#pragma unroll
    for (int j = 0; j < TN; ++j) {
      const int col = n0 + wn * WN + j * 32 + (lane & 31);
      const bool colok = col < p.N;
      const int colc = colok ? col : p.N - 1;
      const int rbase = m0 + wm * WM + i * 32 + 4 * (lane >> 5);
      float resv[RN], auxv[RN], oldv[RN];
      float bias_v = 0.0f;
      if (use_bias) bias_v = p.bias[colc];
      if (use_res) {
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) {
          const int r = r0 + rr;
          int row = rbase + (r & 3) + 8 * (r >> 2);
          row = row < p.M ? row : p.M - 1;
          resv[rr] = p.residual[(int64_t)row * p.ldr + colc];
        }
      }
      if (use_aux) {
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) {
          const int r = r0 + rr;
          int row = rbase + (r & 3) + 8 * (r >> 2);
          row = row < p.M ? row : p.M - 1;
          auxv[rr] = p.aux[(int64_t)row * p.ldaux + colc];
        }
      }
      if (rmw) {
#pragma unroll
        for (int rr = 0; rr < RN; ++rr) {
          const int r = r0 + rr;
          int row = rbase + (r & 3) + 8 * (r >> 2);
          row = row < p.M ? row : p.M - 1;
          oldv[rr] = p.C[(int64_t)row * p.ldc + colc];
        }
      }
#pragma unroll
      for (int rr = 0; rr < RN; ++rr) {
        const int r = r0 + rr;
        const int row = rbase + (r & 3) + 8 * (r >> 2);
        float t = vals[i][j][rr] * p.out_scale + bias_v;
        t = mesm_act(t, p.e_act, slope);
        if (e_thresh)
          t = mesm_dropout_apply(t, (uint32_t)((int64_t)(row + p.e_drop_row0) * p.N + col), p.e_drop_seed + seed_off, e_thresh,
                                 e_inv_keep);
        if (use_aux) {
          const float z = auxv[rr];
          if (p.e_actgrad == MESM_ACT_RELU) {
            t = z > 0.0f ? t : 0.0f;
          } else if (z <= 0.0f) {
            if (row < p.M && colok) dslope_part += t * z;
            t *= slope;
          }
        }
        if (use_res) t += resv[rr];
        if (rmw) t += oldv[rr];
        if (row < p.M && colok) {
          float* c = p.C + (int64_t)row * p.ldc + col;
          if (p.accumulate == 2) atomicAdd(c, t);
          else *c = t;
        }
      }
    }
  }
  if (p.e_actgrad == MESM_ACT_PRELU && p.dslope) dslope_store(p, dslope_part, Red, linear_block());
}

template <int BM, int BN, int LA, int LB, int VEC, int ADD>
int launch(const MesmGemmArgs& a, hipStream_t s) {
  // 64x64 tiles run few MFMAs per k-step: a deeper k-step (64) keeps the MFMA block longer than
  // the global-load latency it has to hide; 128x128 tiles already do with 32.
  constexpr int BK = (BM == 128) ? 32 : 64;
  dim3 grid((a.M + BM - 1) / BM, (a.N + BN - 1) / BN, a.split_k > 1 ? a.split_k : 1);
  hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, BK, LA, LB, VEC, ADD>), grid, dim3(NTHREADS), 0, s, a);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

template <int BM, int BN, int VEC, int ADD>
int launch_layout(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch<BM, BN, R, R, VEC, ADD>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch<BM, BN, R, O, VEC, ADD>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch<BM, BN, O, O, VEC, ADD>(a, s);
  if (a.a_layout == O && a.b_layout == R) return launch<BM, BN, O, R, VEC, ADD>(a, s);
  return MESM_EINVAL;
}

template <int VEC>
int launch_tile(const MesmGemmArgs& a, hipStream_t s) {
  const int add = a.A2 ? 1 : (a.B2 ? 2 : 0);
  const long z = a.split_k > 1 ? a.split_k : 1;
  auto blocks = [&](int t) { return (long)((a.M + t - 1) / t) * ((a.N + t - 1) / t) * z; };
  // tuning knob (tools/gemm_bench4.py): MESM_GEMM_TILE=32|64|128 pins the configuration
  const int force_tile = mesm_gemm_force_tile();
  if (force_tile == 128 && add == 0) return launch_layout<128, 128, VEC, 0>(a, s);
  if (force_tile == 64) {
    if (add == 0) return launch_layout<64, 64, VEC, 0>(a, s);
    if (add == 1) return launch_layout<64, 64, VEC, 1>(a, s);
    return launch_layout<64, 64, VEC, 2>(a, s);
  }
  if (force_tile == 32) {
    if (add == 0) return launch_layout<32, 32, VEC, 0>(a, s);
    if (add == 1) return launch_layout<32, 32, VEC, 1>(a, s);
    return launch_layout<32, 32, VEC, 2>(a, s);
  }
  // largest tile that still gives every CU about two workgroups; the small 32x32 k-split tile
  // otherwise (most d x d GEMMs of the step: 2400 x 256 x 256 -> 600 workgroups)
  if (blocks(128) >= 512 && add == 0) return launch_layout<128, 128, VEC, 0>(a, s);
  if (blocks(64) >= 512) {
    if (add == 0) return launch_layout<64, 64, VEC, 0>(a, s);
    if (add == 1) return launch_layout<64, 64, VEC, 1>(a, s);
    return launch_layout<64, 64, VEC, 2>(a, s);
  }
  if (add == 0) return launch_layout<32, 32, VEC, 0>(a, s);
  if (add == 1) return launch_layout<32, 32, VEC, 1>(a, s);
  return launch_layout<32, 32, VEC, 2>(a, s);
}

// ------------------------------------------------------------------------------------------------
// Small-problem kernel ("frag"): one 32x32 output tile per workgroup, the reduce range split over
// the four waves, MFMA operand fragments loaded STRAIGHT from global memory into registers -- no
// LDS staging, no barrier in the k loop.  Most GEMMs of the step are (2400 | 1024 | 320) x 256 x 256:
// far too small to amortise a staged pipeline (their k loop is 4 tiles), so what counts is how
// much latency sits on the critical path of a workgroup.  Here it is ONE round trip: a wave issues
// the loads of its whole reduce range slice up front (16 B per lane for a reduce-contiguous operand
// = 4 MFMAs worth, one coalesced dword per MFMA for an outer-contiguous operand), then runs its
// MFMA chain; the four partial tiles meet in LDS once, for the epilogue.
// Lane l of a wave holds row/column (l & 31); in MFMA j of a k-step of 8 the lane half h = l >> 5
// supplies k = kb + 4h + j (both operands use the same map, so any k permutation is harmless).
constexpr int FRAG_STEPS = 4;  // k-steps of 8 per loop iteration (32 reduce indices)

template <int LAYOUT, bool ADD>
struct Frag {
  float v[FRAG_STEPS][4];

  // Unguarded: every k in [kb, kb + 32) is inside the wave's range.  No arithmetic touches the
  // loaded registers here, so the loads of iteration t+1 stay in flight behind the MFMAs of t.
  __device__ __forceinline__ void load_full(const float* __restrict__ base, const float* __restrict__ add,
                                            int64_t ld, int row, int kb, int h) {
#pragma unroll
    for (int s = 0; s < FRAG_STEPS; ++s) {
      const int k = kb + 8 * s + 4 * h;
      if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
        const float4 x = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + k);
        v[s][0] = x.x; v[s][1] = x.y; v[s][2] = x.z; v[s][3] = x.w;
        if (ADD) {
          const float4 y = *reinterpret_cast<const float4*>(add + (int64_t)row * ld + k);
          v[s][0] += y.x; v[s][1] += y.y; v[s][2] += y.z; v[s][3] += y.w;
        }
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          v[s][j] = base[(int64_t)(k + j) * ld + row];
          if (ADD) v[s][j] += add[(int64_t)(k + j) * ld + row];
        }
      }
    }
  }

  // Guarded tail (at most one per wave): k >= k1 contributes zeros.
  __device__ __forceinline__ void load_tail(const float* __restrict__ base, const float* __restrict__ add,
                                            int64_t ld, int row, int kb, int k1, int h) {
#pragma unroll
    for (int s = 0; s < FRAG_STEPS; ++s) {
      const int k = kb + 8 * s + 4 * h;
      if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
        const bool ok = k < k1;  // ranges are multiples of 4: a vector is all in or all out
        const int kc = ok ? k : 0;
        float4 x = *reinterpret_cast<const float4*>(base + (int64_t)row * ld + kc);
        if (ADD) {
          const float4 y = *reinterpret_cast<const float4*>(add + (int64_t)row * ld + kc);
          x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
        v[s][0] = ok ? x.x : 0.0f; v[s][1] = ok ? x.y : 0.0f;
        v[s][2] = ok ? x.z : 0.0f; v[s][3] = ok ? x.w : 0.0f;
      } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = k + j < k1;
          const int kc = ok ? k + j : 0;
          float x = base[(int64_t)kc * ld + row];
          if (ADD) x += add[(int64_t)kc * ld + row];
          v[s][j] = ok ? x : 0.0f;
        }
      }
    }
  }

  // activation / dropout on the fragment; `o` = unclamped outer index of this lane
  template <bool OUTER_IS_ROW>
  __device__ __forceinline__ void finish(const XForm& xf, int o, int kb, int h) {
#pragma unroll
    for (int s = 0; s < FRAG_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int gk = kb + 8 * s + 4 * h + j;
        float x = mesm_act(v[s][j], xf.act, xf.slope);
        if (xf.thresh) {
          const int64_t idx = OUTER_IS_ROW ? (int64_t)o * xf.lld + gk : (int64_t)gk * xf.lld + o;
          x = mesm_dropout_apply(x, (uint32_t)idx, xf.seed, xf.thresh, xf.inv_keep);
        }
        v[s][j] = x;
      }
  }
};

// XF: operand transforms (activation / dropout on A or B) compiled in; the plain instantiation
// carries none of that code.
template <int LA, int LB, int ADD, bool XF>
__global__ __launch_bounds__(NTHREADS) void gemm_frag_kernel(const MesmGemmArgs p) {
  __shared__ float Red[4 * 16 * 64];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blockIdx.x * 32, n0 = blockIdx.y * 32;

  const int KM = p.K;  // this kernel stages every reduce index itself
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = blockIdx.z * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (blockIdx.z > 0) return;
      kbeg = kend = KM;
    }
  }
  // this wave's slice of [kbeg, kend): a multiple of 32 long except for the last wave's tail
  const int kw = (((kend - kbeg + 3) >> 2) + 31) & ~31;
  const int k0 = kbeg + wave * kw;
  const int k1 = k0 + kw < kend ? k0 + kw : kend;
  const int nfull = k1 > k0 ? (k1 - k0) >> 5 : 0;
  const bool tail = k1 > k0 && ((k1 - k0) & 31) != 0;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;  // dropout index = dense index of the operand AS STORED
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  const int ra = m0 + li < p.M ? m0 + li : p.M - 1;
  const int rb = n0 + li < p.N ? n0 + li : p.N - 1;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float csum = 0.0f;
  const bool do_colsum = (p.colsum != nullptr) && (blockIdx.y == 0);

  Frag<LA, ADD == 1> fa, na;
  Frag<LB, ADD == 2> fb, nb;

  auto mma = [&](Frag<LA, ADD == 1>& a, Frag<LB, ADD == 2>& b, int kb) {
    if (XF) {
      a.template finish<LA == MESM_LAYOUT_REDUCE_CONTIG>(xa, m0 + li, kb, h);
      b.template finish<LB == MESM_LAYOUT_REDUCE_CONTIG>(xb, n0 + li, kb, h);
    }
#pragma unroll
    for (int s = 0; s < FRAG_STEPS; ++s)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.v[s][j], b.v[s][j], acc, 0, 0, 0);
    if (do_colsum) {
#pragma unroll
      for (int s = 0; s < FRAG_STEPS; ++s)
#pragma unroll
        for (int j = 0; j < 4; ++j) csum += a.v[s][j];
    }
  };

  // Register ping-pong, branch-free in the steady state (a conditional prefetch makes hipcc merge
  // its vmcnt bookkeeping pessimistically and wait for the NEW loads before the MFMAs of the old
  // ones): the loads of iteration t+1 are always in flight behind the MFMAs of iteration t.
  // sched_barrier: keep every load group ahead of the MFMA block it overlaps (hipcc otherwise sinks
  // the loads between the MFMAs to recycle fragment registers, which halves the prefetch distance)
#define MESM_SB() __builtin_amdgcn_sched_barrier(0)
  if (nfull > 0) {
    fa.load_full(p.A, p.A2, p.lda, ra, k0, h);
    fb.load_full(p.B, p.B2, p.ldb, rb, k0, h);
  }
  int it = 0;
  for (; it + 2 < nfull; it += 2) {
    const int kb = k0 + 32 * it;
    na.load_full(p.A, p.A2, p.lda, ra, kb + 32, h);
    nb.load_full(p.B, p.B2, p.ldb, rb, kb + 32, h);
    MESM_SB();
    mma(fa, fb, kb);
    MESM_SB();
    fa.load_full(p.A, p.A2, p.lda, ra, kb + 64, h);
    fb.load_full(p.B, p.B2, p.ldb, rb, kb + 64, h);
    MESM_SB();
    mma(na, nb, kb + 32);
    MESM_SB();
  }
  if (nfull - it == 2) {
    const int kb = k0 + 32 * it;
    na.load_full(p.A, p.A2, p.lda, ra, kb + 32, h);
    nb.load_full(p.B, p.B2, p.ldb, rb, kb + 32, h);
    MESM_SB();
    mma(fa, fb, kb);
    mma(na, nb, kb + 32);
  } else if (nfull - it == 1) {
    mma(fa, fb, k0 + 32 * it);
  }
#undef MESM_SB
  if (tail) {
    const int kb = k0 + 32 * nfull;
    fa.load_tail(p.A, p.A2, p.lda, ra, kb, k1, h);
    fb.load_tail(p.B, p.B2, p.ldb, rb, kb, k1, h);
    mma(fa, fb, kb);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if (do_colsum) {
    csum = add_xor32(csum);
    if (h == 0 && m0 + li < p.M && csum != 0.0f) atomicAdd(p.colsum + m0 + li, csum);
  }

  ksplit_epilogue<LA, LB, XF>(p, acc, Red, m0, n0, slope, seed_off, blockIdx.z, linear_block(), p.K, XForm{}, XForm{});
}

template <int LA, int LB>
int launch_frag_add(const MesmGemmArgs& a, hipStream_t s) {
  dim3 grid((a.M + 31) / 32, (a.N + 31) / 32, a.split_k > 1 ? a.split_k : 1);
  const int add = a.A2 ? 1 : (a.B2 ? 2 : 0);
  const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
  if (xf) {
    if (add == 0) hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 0, true>), grid, dim3(NTHREADS), 0, s, a);
    else if (add == 1) hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 1, true>), grid, dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 2, true>), grid, dim3(NTHREADS), 0, s, a);
  } else {
    if (add == 0) hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 0, false>), grid, dim3(NTHREADS), 0, s, a);
    else if (add == 1) hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 1, false>), grid, dim3(NTHREADS), 0, s, a);
    else hipLaunchKernelGGL((gemm_frag_kernel<LA, LB, 2, false>), grid, dim3(NTHREADS), 0, s, a);
  }
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

int launch_frag(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_frag_add<R, R>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch_frag_add<R, O>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch_frag_add<O, O>(a, s);
  return launch_frag_add<O, R>(a, s);
}

// the frag kernel reads reduce-contiguous operands as 16-byte vectors
bool frag_ok(const MesmGemmArgs& a) {
  auto vec_ok = [&](const float* p, const float* p2, int64_t ld) {
    return (ld % 4 == 0) && aligned_to(p, 16) && aligned_to(p2, 16);
  };
  if (a.a_layout == MESM_LAYOUT_REDUCE_CONTIG && !(vec_ok(a.A, a.A2, a.lda) && a.K % 4 == 0)) return false;
  if (a.b_layout == MESM_LAYOUT_REDUCE_CONTIG && !(vec_ok(a.B, a.B2, a.ldb) && a.K % 4 == 0)) return false;
  return true;
}

// ------------------------------------------------------------------------------------------------
// "wstage" kernel: same decomposition as the frag kernel (32x32 tile per workgroup, reduce range
// split over the four waves) but each wave brings ITS OWN k-slice in through LDS in full 128-byte
// lines with LDS-DMA (global_load_lds_dwordx4: one instruction = 8 rows x 128 B, no VGPRs), then
// reads MFMA fragments back with ds_read_b128.  tools/probe/loads.hip: the line-shaped loads move the
// 39 MB a 2400x256x256 GEMM pulls out of L2 in 3.2-3.7 us, the fragment-shaped ones (32 B per row
// per instruction) in 5.9 us -- and that traffic, not the MFMA work (2.0 us), bounds these GEMMs.
// The slabs are wave-private, so there is no barrier in the k loop: a wave waits on its own vmcnt.
//
// Stage = 32 reduce indices; slab = 32 x 32 floats (4 KB) per operand; 2 stages per wave (64 KB per
// workgroup, 2 workgroups per CU).  LDS-DMA writes lane-linear, so the bank-conflict-free image is made
// by permuting the SOURCE addresses:
//   reduce-contiguous operand: slot (row r, 16-B position p) holds chunk p ^ ((r >> 1) & 7) of row r;
//     the reader (row i, chunk c = 2s + h) finds it at position c ^ ((i >> 1) & 7) with one ds_read_b128;
//   outer-contiguous operand: LDS row 8q + sr holds reduce index 8q + (((sr & 1) << 2) | (sr >> 1)), so
//     that k and k + 4 (the two lane halves) sit in opposite bank halves; read with ds_read_b32.
#ifdef MESM_L64_TRACE
// in-kernel time stamps (s_memtime, shader cycles) of wave 0 of the first 1024 workgroups: 32 slots each
__device__ unsigned long long l64_trace[1024 * 32];
__device__ __forceinline__ void l64_stamp(int slot) {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
  __builtin_amdgcn_sched_barrier(0);
  if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.z == 0 && slot < 32) l64_trace[blockIdx.x * 32 + slot] = t;
}
#define L64_STAMP(i) l64_stamp(i)
#else
#define L64_STAMP(i)
#endif

#ifndef MESM_WS_NW
#define MESM_WS_NW 4
#endif
constexpr int WS_NW = MESM_WS_NW;  // waves per wstage workgroup = k-split factor inside the workgroup (1, 2, 4, 8).
// Measured (tools/gemm_sweep.py, 2400 x 256 x 256 / 4800 x 256 x 256 / 256 x 256 x 2400 split 4):
// 4 waves 8.6 / 13.3 / 7.6 us; 2 waves 9.3 / 15.8 / 10.6; 8 waves 9.3 / 14.4 / 8.5; 1 wave 9.6 / 14.4 / 17.1.
static_assert(WS_NW == 1 || WS_NW == 2 || WS_NW == 4 || WS_NW == 8, "wstage waves");
constexpr int WS_THREADS = 64 * WS_NW;
#ifndef MESM_WS_WAVES
#define MESM_WS_WAVES 0
#endif
#if MESM_WS_WAVES > 0
#define WS_BOUNDS __launch_bounds__(WS_THREADS, MESM_WS_WAVES)
#else
#define WS_BOUNDS __launch_bounds__(WS_THREADS)
#endif

// WS_STAGES: wave-private k-tiles resident in LDS per operand pair.  2 = double buffer (64 KB per workgroup,
// 2 workgroups per CU), 1 = load / read / refill in place (32 KB, 5 per CU).  Measured (tools/gemm_sweep.py):
// the single stage wins wherever a wave has at most 2 k-tiles or the launch is a single round
// (2400 x 256 x 256 9.6 -> 8.5 us, 4800 x 256 x 256 14.2 -> 13.0, 256 x 256 x 2400 split 4 8.1 -> 7.6),
// the double buffer on long k loops over many rounds (4800 x 256 x 1024: 35.7 vs 38.9 us).
template <int LA, int LB, bool XF, int WS_STAGES>
__device__ __forceinline__ void wstage_body(const MesmGemmArgs& p, const Blk blk, float* L) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blk.x * 32, n0 = blk.y * 32;
  L64_STAMP(0);

  const int KM = gemm_kmain(p);  // reduce indices [KM, K) are added in the epilogue of the first k-slice
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = blk.z * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (blk.z > 0) return;
      kbeg = kend = KM;
    }
  }
  const int kw = (((kend - kbeg + WS_NW - 1) / WS_NW) + 31) & ~31;
  const int k0 = kbeg + wave * kw;
  const int k1 = k0 + kw < kend ? k0 + kw : kend;
  const int nst = k1 > k0 ? (k1 - k0 + 31) >> 5 : 0;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;  // dropout index = dense index of the operand AS STORED
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  float* mine = L + wave * (WS_STAGES * 2 * WS_SLAB);
  auto issue = [&](int st) {
    float* buf = mine + (st % WS_STAGES) * (2 * WS_SLAB);
    const int kb = k0 + 32 * st;
    ws_issue<LA>(p.A, p.lda, m0, p.M, kb, k1, buf, lane);
    ws_issue<LB>(p.B, p.ldb, n0, p.N, kb, k1, buf + WS_SLAB, lane);
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float csum = 0.0f;
  const bool do_colsum = (p.colsum != nullptr) && (blk.y == 0);

  L64_STAMP(1);
  if (nst > 0) issue(0);
  if (WS_STAGES > 1 && nst > 1) issue(1);
  L64_STAMP(2);
  for (int st = 0; st < nst; ++st) {
    // each stage is 8 LDS-DMA instructions; leave the next stage in flight
    if (WS_STAGES > 1 && st + 1 < nst) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    L64_STAMP(3 + 3 * st);
    const float* buf = mine + (st % WS_STAGES) * (2 * WS_SLAB);
    float a[4][4], b[4][4];
    ws_read<LA>(buf, li, h, a);
    ws_read<LB>(buf + WS_SLAB, li, h, b);
    const int kb = k0 + 32 * st;
    if (st + WS_STAGES < nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: slab is free
      issue(st + WS_STAGES);
    }
    if (kb + 32 > k1) {  // partial last stage: reduce indices >= k1 contribute zeros
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = kb + 8 * s_ + 4 * h + j < k1;
          a[s_][j] = ok ? a[s_][j] : 0.0f;
          b[s_][j] = ok ? b[s_][j] : 0.0f;
        }
    }
    if (XF) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gk = kb + 8 * s_ + 4 * h + j;
          float x = mesm_act(a[s_][j], xa.act, xa.slope);
          if (xa.thresh)
            x = mesm_dropout_apply(x, (uint32_t)(LA == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)(m0 + li) * xa.lld + gk : (int64_t)gk * xa.lld + m0 + li),
                                   xa.seed, xa.thresh, xa.inv_keep);
          a[s_][j] = x;
          float y = mesm_act(b[s_][j], xb.act, xb.slope);
          if (xb.thresh)
            y = mesm_dropout_apply(y, (uint32_t)(LB == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)(n0 + li) * xb.lld + gk : (int64_t)gk * xb.lld + n0 + li),
                                   xb.seed, xb.thresh, xb.inv_keep);
          b[s_][j] = y;
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][j], b[s_][j], acc, 0, 0, 0);
    if (do_colsum) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) csum += a[s_][j];
    }
    L64_STAMP(5 + 3 * st);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  L64_STAMP(28);

  if (do_colsum) {
    csum = add_xor32(csum);
    if (wave == 0 && blk.z == 0 && KM < p.K) csum += tail_colsum<LA, XF>(p, m0 + li, KM, xa);
    if (h == 0 && m0 + li < p.M && csum != 0.0f) atomicAdd(p.colsum + m0 + li, csum);
  }
  __syncthreads();  // every wave is done with its slabs: the reduction buffer aliases them
  ksplit_epilogue<LA, LB, XF, WS_NW>(p, acc, L, m0, n0, slope, seed_off, blk.z, blk.slot, KM, xa, xb);
  L64_STAMP(29);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  L64_STAMP(30);
}

constexpr int ws_lds_floats(int stages) {  // [wave][stage][operand]; at least the cross-wave reduction buffer
  return WS_NW * stages * 2 * WS_SLAB > WS_NW * 16 * 64 ? WS_NW * stages * 2 * WS_SLAB : WS_NW * 16 * 64;
}  // [wave][stage][operand]: 64 KB at 2 stages, 32 KB at 1

// which staging depth a problem gets (host side): k-tiles per wave and workgroups of the launch
inline int ws_stages_for(const MesmGemmArgs& a) {
  const long z = a.split_k > 1 ? a.split_k : 1;
  const long kper_wave = ((a.K + z - 1) / z + WS_NW - 1) / WS_NW;
  const long wgs = (long)((a.M + 31) / 32) * ((a.N + 31) / 32) * z;
  return (kper_wave <= 64 || wgs <= 512) ? 1 : 2;
}

template <int LA, int LB, bool XF, int STAGES>
__global__ WS_BOUNDS void gemm_wstage_kernel(const MesmGemmArgs p, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[ws_lds_floats(STAGES)];
  Blk blk;
  blk.slot = linear_block();
  xcd_tile_z((int)blk.slot, (p.M + 31) / 32, (p.N + 31) / 32, p.split_k, blk.x, blk.y, blk.z);
  wstage_body<LA, LB, XF, STAGES>(p, blk, L);
}

// Grouped launch: up to GROUP_MAX independent small problems in ONE kernel (a launch costs 1.66 us of
// dispatch + ~3 us of exposed latency chain whatever its size, and the backward of every block as well
// as the decoder are made of independent 5 us GEMMs).  The workgroups of all problems are laid out
// back to back on blockIdx.x; layouts / transforms are selected per problem at run time (wave-uniform).
// MASK / XFM: the operand-layout pairs (bit = 2 * (A outer-contiguous) + (B outer-contiguous)) and whether operand transforms
// occur among the launch's members; only those bodies are in the kernel (the kernel with all eight is 109 KB of code: see
// gemm_wstage64_group_kernel).
template <int STAGES, int MASK = 15, bool XFM = true>
__global__ WS_BOUNDS void gemm_wstage_group_kernel(const GroupArgs g, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[ws_lds_floats(STAGES)];
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  // The selected problem is read from the kernarg segment with a wave-uniform DYNAMIC offset (scalar
  // loads).  Indexing the by-value parameter itself (g.p[gi]) made hipcc copy all of GroupArgs into
  // per-lane scratch first: 1,840 bytes of private memory per lane, written and re-read by every workgroup.
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const MesmGemmArgs p = *reinterpret_cast<const MesmGemmArgs*>(ka + offsetof(GroupArgs, p) + (size_t)gi * sizeof(MesmGemmArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(GroupArgs, start) + (size_t)gi * sizeof(int));
  const int local = bid - first;
  const int mt = (p.M + 31) / 32, nt = (p.N + 31) / 32;
  Blk blk;
  xcd_tile_z(local, mt, nt, p.split_k, blk.x, blk.y, blk.z);  // (local & 7 names an XCD up to a rotation by the problem's first workgroup)
  blk.slot = local;
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  const bool xf = p.a_act != MESM_ACT_NONE || p.b_act != MESM_ACT_NONE || p.a_drop_p > 0.f || p.b_drop_p > 0.f;
  const int sel = (p.a_layout == O ? 2 : 0) + (p.b_layout == O ? 1 : 0);
  if (!xf) {
    if ((MASK & 1) && sel == 0) wstage_body<R, R, false, STAGES>(p, blk, L);
    else if ((MASK & 2) && sel == 1) wstage_body<R, O, false, STAGES>(p, blk, L);
    else if ((MASK & 4) && sel == 2) wstage_body<O, R, false, STAGES>(p, blk, L);
    else if ((MASK & 8) && sel == 3) wstage_body<O, O, false, STAGES>(p, blk, L);
  } else if (XFM) {
    if ((MASK & 1) && sel == 0) wstage_body<R, R, true, STAGES>(p, blk, L);
    else if ((MASK & 2) && sel == 1) wstage_body<R, O, true, STAGES>(p, blk, L);
    else if ((MASK & 4) && sel == 2) wstage_body<O, R, true, STAGES>(p, blk, L);
    else if ((MASK & 8) && sel == 3) wstage_body<O, O, true, STAGES>(p, blk, L);
  }
}

template <int LA, int LB>
int launch_wstage_l(const MesmGemmArgs& a, hipStream_t s) {
  dim3 grid(((a.M + 31) / 32) * ((a.N + 31) / 32), 1, a.split_k > 1 ? a.split_k : 1);  // 1-D: xcd_tile() maps it
  const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
  const bool one = ws_stages_for(a) == 1;
  const SideRed sr = take_side(s);
  if (xf && one) hipLaunchKernelGGL((gemm_wstage_kernel<LA, LB, true, 1>), grid, dim3(WS_THREADS), 0, s, a, sr);
  else if (xf) hipLaunchKernelGGL((gemm_wstage_kernel<LA, LB, true, 2>), grid, dim3(WS_THREADS), 0, s, a, sr);
  else if (one) hipLaunchKernelGGL((gemm_wstage_kernel<LA, LB, false, 1>), grid, dim3(WS_THREADS), 0, s, a, sr);
  else hipLaunchKernelGGL((gemm_wstage_kernel<LA, LB, false, 2>), grid, dim3(WS_THREADS), 0, s, a, sr);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

int launch_wstage(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_wstage_l<R, R>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch_wstage_l<R, O>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch_wstage_l<O, O>(a, s);
  return launch_wstage_l<O, R>(a, s);
}

// LDS-DMA moves 16-byte chunks, from ANY 4-byte aligned address (measured: rows of 2818 or 5003 floats,
// 8- and 4-byte aligned, give exact results and 51 us instead of 78 for 2400 x 256 x 2818): no alignment
// rule on pointers or leading dimensions.  What 16-byte granularity cannot express is handled outside
// the MFMA loop (gemm_kmain / tail_accumulate): the last K % 4 reduce indices of reduce-contiguous
// operands, and the last reduce row when an outer-contiguous operand's chunk would straddle past the
// end of the matrix.  No addend operands (A2 / B2) on this path.
bool wstage_ok(const MesmGemmArgs& a) {
  if (a.A2 || a.B2) return false;
  auto ok = [&](int layout, int extent) {
    if (layout == MESM_LAYOUT_REDUCE_CONTIG) return a.K >= 4;
    return extent >= 4;
  };
  return ok(a.a_layout, a.M) && ok(a.b_layout, a.N);
}

// ------------------------------------------------------------------------------------------------
// "wstage64": the k-split idea on a 64 x 64 tile.  Each of the four waves computes the WHOLE tile
// (2 x 2 accumulators) over its quarter of the reduce range, staged wave-privately by LDS-DMA like
// wstage.  Per 32-deep stage a wave loads 16 KB and issues 64 MFMAs (wstage: 8 KB for 16), so the
// L2 traffic per flop halves and one stage of MFMAs (4096 cycles) covers the latency of the next
// stage's loads: a single LDS buffer per wave suffices, because all fragments of a stage are in
// registers before its refill is issued.  Used when the problem has enough 64 x 64 tiles to occupy
// the chip (the 2400- and 4800-row d x d GEMMs, the split-K weight gradients).
template <int LA, int LB, bool XF, int BF = 0>
__device__ __forceinline__ void wstage64_body(const MesmGemmArgs& p, const Blk blk, float* L) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blk.x * 64, n0 = blk.y * 64;
  L64_STAMP(0);

  const int KM = gemm_kmain(p);  // reduce indices [KM, K) are added in the epilogue of the first k-slice
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = blk.z * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (blk.z > 0) return;
      kbeg = kend = KM;
    }
  }
  const int kw = (((kend - kbeg + 3) >> 2) + 31) & ~31;
  const int k0 = kbeg + wave * kw;
  const int k1 = k0 + kw < kend ? k0 + kw : kend;
  const int nst = k1 > k0 ? (k1 - k0 + 31) >> 5 : 0;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  float* mine = L + wave * (4 * WS_SLAB);  // A rows 0-31, A rows 32-63, B rows 0-31, B rows 32-63
  auto issue = [&](int st) {
    const int kb = k0 + 32 * st;
    ws_issue<LA>(p.A, p.lda, m0, p.M, kb, k1, mine, lane);
    ws_issue<LA>(p.A, p.lda, m0 + 32, p.M, kb, k1, mine + WS_SLAB, lane);
    ws_issue<LB>(p.B, p.ldb, n0, p.N, kb, k1, mine + 2 * WS_SLAB, lane);
    ws_issue<LB>(p.B, p.ldb, n0 + 32, p.N, kb, k1, mine + 3 * WS_SLAB, lane);
  };

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float csum[2] = {0.0f, 0.0f};
  const bool do_colsum = (p.colsum != nullptr) && (blk.y == 0);
  HalfScales hs;  // BF == 2: this wave's operand scales (gemm_ws.hpp)
  hs.init();

  L64_STAMP(1);
  if (nst > 0) issue(0);
  L64_STAMP(2);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    L64_STAMP(3 + 3 * st);
    float a[2][4][4], b[2][4][4];
    ws_read<LA>(mine, li, h, a[0]);
    ws_read<LA>(mine + WS_SLAB, li, h, a[1]);
    ws_read<LB>(mine + 2 * WS_SLAB, li, h, b[0]);
    ws_read<LB>(mine + 3 * WS_SLAB, li, h, b[1]);
    const int kb = k0 + 32 * st;
    if (st + 1 < nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: refill the slabs
      issue(st + 1);
    }
    if (kb + 32 > k1) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const bool ok = kb + 8 * s_ + 4 * h + j < k1;
            a[t][s_][j] = ok ? a[t][s_][j] : 0.0f;
            b[t][s_][j] = ok ? b[t][s_][j] : 0.0f;
          }
    }
    if (XF) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const int gk = kb + 8 * s_ + 4 * h + j;
            const int gm = m0 + 32 * t + li, gn = n0 + 32 * t + li;
            float x = mesm_act(a[t][s_][j], xa.act, xa.slope);
            if (xa.thresh)
              x = mesm_dropout_apply(x, (uint32_t)(LA == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gm * xa.lld + gk : (int64_t)gk * xa.lld + gm),
                                     xa.seed, xa.thresh, xa.inv_keep);
            a[t][s_][j] = x;
            float y = mesm_act(b[t][s_][j], xb.act, xb.slope);
            if (xb.thresh)
              y = mesm_dropout_apply(y, (uint32_t)(LB == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gn * xb.lld + gk : (int64_t)gk * xb.lld + gn),
                                     xb.seed, xb.thresh, xb.inv_keep);
            b[t][s_][j] = y;
          }
    }
    if (BF == 2) {
      const float ma = frag_amax(a[0], a[1]), mb = frag_amax(b[0], b[1]);
      if (hs.a.leaves(ma) || hs.b.leaves(mb)) {  // wave-uniform; homogeneous operands: taken at the wave's first stage only
        const int delta = hs.repick(frag_amax_finite(a[0], a[1]), frag_amax_finite(b[0], b[1]), st == 0);
        if (delta != 0) {
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[ti][tj][r] = __builtin_ldexpf(acc[ti][tj][r], delta);
        }
      }
      HalfFrag fa[2], fb[2];
      const float sca = hs.a.scale(), scb = hs.b.scale();
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        fa[t].make(a[t], sca);
        fb[t].make(b[t], scb);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = half_mma(fa[ti], fb[tj], acc[ti][tj]);
    } else if (BF > 0) {
      SplitFrag<BF> sa[2], sb[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        sa[t].make(a[t]);
        sb[t].make(b[t]);
      }
#pragma unroll
      for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = split_mma<BF>(sa[ti], sb[tj], acc[ti][tj]);
    } else {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int ti = 0; ti < 2; ++ti)
#pragma unroll
            for (int tj = 0; tj < 2; ++tj)
              acc[ti][tj] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[ti][s_][j], b[tj][s_][j], acc[ti][tj], 0, 0, 0);
    }
    if (do_colsum) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int j = 0; j < 4; ++j) csum[t] += a[t][s_][j];
    }
    L64_STAMP(5 + 3 * st);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  L64_STAMP(28);
  if (BF == 2 && hs.a.e + hs.b.e != 0) {  // back to unit scale before the waves' partial tiles meet
    const int back = -(hs.a.e + hs.b.e);
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
      for (int tj = 0; tj < 2; ++tj)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ti][tj][r] = __builtin_ldexpf(acc[ti][tj][r], back);
  }

  if (do_colsum) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float c = add_xor32(csum[t]);
      const int gm = m0 + 32 * t + li;
      if (wave == 0 && blk.z == 0 && KM < p.K) c += tail_colsum<LA, XF>(p, gm, KM, xa);
      if (h == 0 && gm < p.M && c != 0.0f) atomicAdd(p.colsum + gm, c);
    }
  }
  __syncthreads();  // every wave is done with its slabs: the reduction buffer aliases them
  // the four partial 64 x 64 tiles meet in LDS; wave w then owns sub-tile (w >> 1, w & 1)
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)  // 16-byte LDS accesses: 16 writes + 16 reads per wave instead of 64 + 64
        reinterpret_cast<float4*>(L)[((wave * 4 + ti * 2 + tj) * 4 + r4) * 64 + lane] =
            make_float4(acc[ti][tj][4 * r4], acc[ti][tj][4 * r4 + 1], acc[ti][tj][4 * r4 + 2], acc[ti][tj][4 * r4 + 3]);
  __syncthreads();
  f32x16 sum;
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) {
    float4 t = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
      const float4 u = reinterpret_cast<const float4*>(L)[((w * 4 + wave) * 4 + r4) * 64 + lane];
      t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
    }
    sum[4 * r4] = t.x; sum[4 * r4 + 1] = t.y; sum[4 * r4 + 2] = t.z; sum[4 * r4 + 3] = t.w;
  }
  __syncthreads();  // dslope_store reuses the head of L
  tile16_epilogue<LA, LB, XF>(p, sum, m0 + 32 * (wave >> 1), n0 + 32 * (wave & 1), slope, seed_off, blk.z, L, blk.slot,
                              KM, xa, xb);
  L64_STAMP(29);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  L64_STAMP(30);
}

template <int LA, int LB, bool XF, int BF = 0>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_wstage64_kernel(const MesmGemmArgs p, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[4 * 4 * WS_SLAB];  // 4 waves x 4 slabs = 64 KB
  Blk blk;
  blk.slot = linear_block();
  xcd_tile_z((int)blk.slot, (p.M + 63) / 64, (p.N + 63) / 64, p.split_k, blk.x, blk.y, blk.z);
  wstage64_body<LA, LB, XF, BF>(p, blk, L);
}

// Grouped launch of 64 x 64 k-split tiles (same GroupArgs as the 32 x 32 one): the problems of one call that are large
// enough for this tile share ONE launch whatever their shapes, layouts and fusions -- a kernel costs ~9 us of ramp
// whatever it computes, and in split-bf16 mode a 64 x 64-per-wave tile is the only one that amortises the split.
// MASK: the operand-layout pairs the launch's members use (bit = 2 * (A outer-contiguous) + (B outer-contiguous)); only
// those bodies are in the kernel.  With all four side by side the kernel is ~200 KB of code of which a workgroup runs one
// variant; the forward's calls (one layout pair) and the backward's (dW + dX: two) get kernels a quarter / half that size:
// 3.295 -> 3.277 ms per step (round 6; MASK = 15 stays for BF = 0 / 6 and anything else).
template <int BF, int MASK = 15>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_wstage64_group_kernel(const GroupArgs g, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[4 * 4 * WS_SLAB];
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();  // (see gemm_wstage_group_kernel)
  const MesmGemmArgs p = *reinterpret_cast<const MesmGemmArgs*>(ka + offsetof(GroupArgs, p) + (size_t)gi * sizeof(MesmGemmArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(GroupArgs, start) + (size_t)gi * sizeof(int));
  const int local = bid - first;
  Blk blk;
  xcd_tile_z(local, (p.M + 63) / 64, (p.N + 63) / 64, p.split_k, blk.x, blk.y, blk.z);
  blk.slot = local;
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  const bool xf = p.a_act != MESM_ACT_NONE || p.b_act != MESM_ACT_NONE || p.a_drop_p > 0.f || p.b_drop_p > 0.f;
  const int sel = (p.a_layout == O ? 2 : 0) + (p.b_layout == O ? 1 : 0);
  if (!xf) {
    if ((MASK & 1) && sel == 0) wstage64_body<R, R, false, BF>(p, blk, L);
    else if ((MASK & 2) && sel == 1) wstage64_body<R, O, false, BF>(p, blk, L);
    else if ((MASK & 4) && sel == 2) wstage64_body<O, R, false, BF>(p, blk, L);
    else if ((MASK & 8) && sel == 3) wstage64_body<O, O, false, BF>(p, blk, L);
  } else if (BF == 0) {  // (operand transforms in split mode stay out of the group: register budget)
    if (sel == 0) wstage64_body<R, R, true, 0>(p, blk, L);
    else if (sel == 1) wstage64_body<R, O, true, 0>(p, blk, L);
    else if (sel == 2) wstage64_body<O, R, true, 0>(p, blk, L);
    else wstage64_body<O, O, true, 0>(p, blk, L);
  }
}


// MESM_GEMM_BF16X: 6 = split-bf16 products, three exact terms (see SplitFrag); 2 = two fp16 terms, three products (HalfFrag);
// anything else = exact f32
inline bool layout_masks() {  // MESM_G64_LAYOUT_MASKS=0: the four-variant kernel for every grouped 64 x 64 launch (A/B)
  static const bool v = []() { const char* e = getenv("MESM_G64_LAYOUT_MASKS"); return !(e && atoi(e) == 0); }();
  return v;
}
inline int bf16x_mode() { return mesm_gemm_bf16x(); }
int mesm_gemm_group64();


template <int LA, int LB>
int launch_wstage64_l(const MesmGemmArgs& a, hipStream_t s) {
  dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), 1, a.split_k > 1 ? a.split_k : 1);  // 1-D: xcd_tile() maps it
  const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
  const int bf = xf ? 0 : bf16x_mode();  // (operand transforms + split: 260-288 VGPRs, one workgroup per CU or spills)
  const SideRed sr = take_side(s);
  if (bf == 6) hipLaunchKernelGGL((gemm_wstage64_kernel<LA, LB, false, 6>), grid, dim3(NTHREADS), 0, s, a, sr);
  else if (bf == 2) hipLaunchKernelGGL((gemm_wstage64_kernel<LA, LB, false, 2>), grid, dim3(NTHREADS), 0, s, a, sr);
  else if (xf) hipLaunchKernelGGL((gemm_wstage64_kernel<LA, LB, true>), grid, dim3(NTHREADS), 0, s, a, sr);
  else hipLaunchKernelGGL((gemm_wstage64_kernel<LA, LB, false>), grid, dim3(NTHREADS), 0, s, a, sr);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

int launch_wstage64(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_wstage64_l<R, R>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch_wstage64_l<R, O>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch_wstage64_l<O, O>(a, s);
  return launch_wstage64_l<O, R>(a, s);
}

// ------------------------------------------------------------------------------------------------
// "wtall": the k-split idea on a TALL tile of RB x 1 blocks of 32 x 32 (RB = 3: 96 x 32).
// The step's 2400 = 75 x 32 row activations with d = 256 columns make 152 workgroups of 64 x 64 (59 % of
// one round); 96 x 32 makes 25 x 8 = 200, every workgroup the same size.  Each of
// the four waves computes the whole tile over its quarter of the reduce range (RB accumulators), staged
// wave-privately by LDS-DMA exactly like wstage64; the partial tiles meet in LDS and wave w takes
// accumulator registers [4w, 4w + 4) of every block through the staged epilogue.
template <int LA, int LB, bool XF, int RB>
__device__ __forceinline__ void wtall_body(const MesmGemmArgs& p, const Blk blk, float* L) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int m0 = blk.x * (32 * RB), n0 = blk.y * 32;

  const int KM = gemm_kmain(p);
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = blk.z * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (blk.z > 0) return;
      kbeg = kend = KM;
    }
  }
  const int kw = (((kend - kbeg + 3) >> 2) + 31) & ~31;
  const int k0 = kbeg + wave * kw;
  const int k1 = k0 + kw < kend ? k0 + kw : kend;
  const int nst = k1 > k0 ? (k1 - k0 + 31) >> 5 : 0;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  float* mine = L + wave * ((RB + 1) * WS_SLAB);  // RB slabs of A rows, one slab of B rows
  auto issue = [&](int st) {
    const int kb = k0 + 32 * st;
#pragma unroll
    for (int t = 0; t < RB; ++t) ws_issue<LA>(p.A, p.lda, m0 + 32 * t, p.M, kb, k1, mine + t * WS_SLAB, lane);
    ws_issue<LB>(p.B, p.ldb, n0, p.N, kb, k1, mine + RB * WS_SLAB, lane);
  };

  f32x16 acc[RB];
#pragma unroll
  for (int i = 0; i < RB; ++i)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;

  if (nst > 0) issue(0);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float a[RB][4][4], b[4][4];
#pragma unroll
    for (int t = 0; t < RB; ++t) ws_read<LA>(mine + t * WS_SLAB, li, h, a[t]);
    ws_read<LB>(mine + RB * WS_SLAB, li, h, b);
    const int kb = k0 + 32 * st;
    if (st + 1 < nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: refill the slabs
      issue(st + 1);
    }
    if (kb + 32 > k1) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = kb + 8 * s_ + 4 * h + j < k1;
#pragma unroll
          for (int t = 0; t < RB; ++t) a[t][s_][j] = ok ? a[t][s_][j] : 0.0f;
          b[s_][j] = ok ? b[s_][j] : 0.0f;
        }
    }
    if (XF) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gk = kb + 8 * s_ + 4 * h + j;
#pragma unroll
          for (int t = 0; t < RB; ++t) {
            const int gm = m0 + 32 * t + li;
            float x = mesm_act(a[t][s_][j], xa.act, xa.slope);
            if (xa.thresh)
              x = mesm_dropout_apply(x, (uint32_t)(LA == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gm * xa.lld + gk : (int64_t)gk * xa.lld + gm),
                                     xa.seed, xa.thresh, xa.inv_keep);
            a[t][s_][j] = x;
          }
          const int gn = n0 + li;
          float y = mesm_act(b[s_][j], xb.act, xb.slope);
          if (xb.thresh)
            y = mesm_dropout_apply(y, (uint32_t)(LB == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gn * xb.lld + gk : (int64_t)gk * xb.lld + gn),
                                   xb.seed, xb.thresh, xb.inv_keep);
          b[s_][j] = y;
        }
    }
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int t = 0; t < RB; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[t][s_][j], b[s_][j], acc[t], 0, 0, 0);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  __syncthreads();  // every wave is done with its slabs: the reduction buffer aliases them
  // Red[w][t][r][lane]; wave w afterwards owns registers [4w, 4w + 4) of every block: rows 4h + 8w + i
  // NOTE: a wave's own partial stays in registers (three quarters of the LDS traffic)
#pragma unroll
  for (int t = 0; t < RB; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r)
      if ((r >> 2) != wave) L[((wave * RB + t) * 16 + r) * 64 + lane] = acc[t][r];
  __syncthreads();
  constexpr int NV = 4 * RB;
  float vals[NV];
#pragma unroll
  for (int t = 0; t < RB; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float own = 0.0f;
      // acc[t][4 * wave + i] with a wave-uniform dynamic index: select instead of scratch
#pragma unroll
      for (int w = 0; w < 4; ++w) own = (w == wave) ? acc[t][4 * w + i] : own;
      float sum = own;
#pragma unroll
      for (int w = 0; w < 4; ++w)
        if (w != wave) sum += L[((w * RB + t) * 16 + 4 * wave + i) * 64 + lane];
      vals[4 * t + i] = sum;
    }
  __syncthreads();  // dslope_store reuses the head of L
  const bool first_split = (p.split_k <= 1) || (blk.z == 0);
  const int rbase = m0 + 4 * h + 8 * wave;
  auto RO = [](int i) { return (i & 3) + 32 * (i >> 2); };
  if (first_split && KM < p.K) tail_accumulate<NV, LA, LB, XF>(p, vals, rbase, n0 + li, KM, xa, xb, RO);
  float dslope_part;
  if (m0 + 32 * RB <= p.M && n0 + 32 <= p.N)
    dslope_part = staged_epilogue<NV, true>(p, vals, rbase, n0 + li, slope, seed_off, first_split, RO);
  else
    dslope_part = staged_epilogue<NV, false>(p, vals, rbase, n0 + li, slope, seed_off, first_split, RO);
  if (p.e_actgrad == MESM_ACT_PRELU && p.dslope) dslope_store(p, dslope_part, L, blk.slot);
}

template <int LA, int LB, bool XF, int RB>
__global__ __launch_bounds__(NTHREADS) void gemm_wtall_kernel(const MesmGemmArgs p) {
  __shared__ __attribute__((aligned(16))) float L[4 * (RB + 1) * WS_SLAB];  // RB = 5: 96 KB, RB = 3: 64 KB
  Blk blk;
  blk.slot = linear_block();
  xcd_tile_z((int)blk.slot, (p.M + 32 * RB - 1) / (32 * RB), (p.N + 31) / 32, p.split_k, blk.x, blk.y, blk.z);
  wtall_body<LA, LB, XF, RB>(p, blk, L);
}

template <int LA, int LB, int RB>
int launch_wtall_l(const MesmGemmArgs& a, hipStream_t s) {
  dim3 grid(((a.M + 32 * RB - 1) / (32 * RB)) * ((a.N + 31) / 32), 1, a.split_k > 1 ? a.split_k : 1);
  const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
  if (xf) hipLaunchKernelGGL((gemm_wtall_kernel<LA, LB, true, RB>), grid, dim3(NTHREADS), 0, s, a);
  else hipLaunchKernelGGL((gemm_wtall_kernel<LA, LB, false, RB>), grid, dim3(NTHREADS), 0, s, a);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

template <int RB>
int launch_wtall(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_wtall_l<R, R, RB>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch_wtall_l<R, O, RB>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch_wtall_l<O, O, RB>(a, s);
  return launch_wtall_l<O, R, RB>(a, s);
}

// wtall has no column-sum output (the bias-gradient side product of the weight-gradient GEMMs)
bool wtall_ok(const MesmGemmArgs& a) { return wstage_ok(a) && a.colsum == nullptr; }

// ------------------------------------------------------------------------------------------------
// "lds64" kernel for the large GEMMs of the step (FFN 4800 x 1024 x 256 and transposes, the 2818-wide
// input projections, the vocabulary head): 64 x 64 tile, 2 x 2 waves of 32 x 32, k-tiles of 32 staged
// by LDS-DMA in full 128-byte lines into a 3-deep ring (48 KB: three workgroups per CU), ONE raw
// s_barrier per k-tile, fragments by ds_read_b128.  The loads are inline asm so that hipcc's vmcnt
// bookkeeping does not drain the ring before every fragment read (it waits vmcnt(0) on any LDS access
// that may alias an LDS-DMA it knows about): the only waits on the ring are the counted ones below.
// Slab layouts / source-address swizzles are the wstage kernel's (ws_issue_asm mirrors ws_issue).
#ifndef MESM_L64_STAGES
#define MESM_L64_STAGES 3
#endif
#ifndef MESM_L64_WAVES
#define MESM_L64_WAVES 0
#endif
#if MESM_L64_WAVES > 0
#define L64_BOUNDS __launch_bounds__(NTHREADS, MESM_L64_WAVES)  // second argument = waves per SIMD the register budget must admit
#else
#define L64_BOUNDS __launch_bounds__(NTHREADS)
#endif
constexpr int L64_STAGES = MESM_L64_STAGES;  // ring depth: STAGES - 1 k-tiles in flight behind the one being multiplied (16 KB each; at most 8)
static_assert(L64_STAGES >= 2 && L64_STAGES <= 8, "ring depth");
constexpr int L64_STAGE_FLOATS = 4 * WS_SLAB;  // A rows 0-31, A rows 32-63, B rows 0-31, B rows 32-63

__device__ __forceinline__ void glds16(const float* g, unsigned lds_byte_addr) {
  // LDS-DMA: 16 bytes per lane, destination = wave-uniform LDS byte address (M0) + lane * 16
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(g), "s"(lds_byte_addr)
               : "memory");
}

// Per-lane source addresses of the 4 LDS-DMA instructions a wave issues per k-tile for its 32-row slab,
// computed ONCE (row clamp, swizzle, 64-bit row * ld): inside the k loop a full k-tile only adds the
// wave-uniform k advance.  Address rules are ws_issue's.
template <int LAYOUT>
struct L64Src {
  const float* ptr[4];  // k-tile 0 of the workgroup's reduce range
  int64_t kstep;        // elements per k-tile: 32 (reduce-contiguous) or 32 * ld (outer-contiguous)
  __device__ __forceinline__ void init(const float* __restrict__ base, int64_t ld, int o0, int extent, int kbeg,
                                       int lane) {
    const int sr = lane >> 3, pos = lane & 7;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
        const int r = 8 * q + sr;
        int row = o0 + r;
        row = row < extent ? row : extent - 1;
        const int c = pos ^ ((r >> 1) & 7);
        ptr[q] = base + (int64_t)row * ld + (kbeg + 4 * c);
      } else {
        const int k = kbeg + 8 * q + (((sr & 1) << 2) | (sr >> 1));
        int o = o0 + 4 * pos;
        const int e4 = (extent + 3) & ~3;
        o = o + 4 <= e4 ? o : e4 - 4;
        ptr[q] = base + (int64_t)k * ld + o;
      }
    }
    kstep = LAYOUT == MESM_LAYOUT_REDUCE_CONTIG ? 32 : 32 * ld;
  }
  // k-tile st lies entirely inside the reduce range
  __device__ __forceinline__ void issue_full(int st, unsigned slab_byte_addr) const {
    const int64_t adv = (int64_t)st * kstep;
#pragma unroll
    for (int q = 0; q < 4; ++q) glds16(ptr[q] + adv, slab_byte_addr + q * 1024);
  }
};

// one 32-row slab of a k-tile: 4 LDS-DMA instructions of this wave (same address rules as ws_issue)
template <int LAYOUT>
__device__ __forceinline__ void l64_issue(const float* __restrict__ base, int64_t ld, int o0, int extent,
                                          int kb, int kend, unsigned slab_byte_addr, int lane) {
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
      k = k < kend ? k : kend - 4;
      g = base + (int64_t)row * ld + k;
    } else {
      int k = kb + 8 * q + (((sr & 1) << 2) | (sr >> 1));
      k = k < kend ? k : kend - 1;
      int o = o0 + 4 * pos;
      const int e4 = (extent + 3) & ~3;
      o = o + 4 <= e4 ? o : e4 - 4;
      g = base + (int64_t)k * ld + o;
    }
    glds16(g, slab_byte_addr + q * 1024);
  }
}


// Tuning builds (tools/build_variant.sh): -DMESM_L64_STAGES / -DMESM_L64_WAVES change ring depth and the
// occupancy target; -DMESM_L64_NO_LOAD / _NO_MFMA / _NO_STORE are kill switches that remove one phase
// (wrong results, timing only) -- the decomposition quoted in DESIGN.md section 8 comes from them.
template <int LA, int LB, bool XF>
__global__ L64_BOUNDS void gemm_lds64_kernel(const MesmGemmArgs p, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[L64_STAGES * L64_STAGE_FLOATS];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = tid >> 6;
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  L64_STAMP(0);
  int tbx, tby, tbz;
  xcd_tile_z((int)linear_block(), (p.M + 63) / 64, (p.N + 63) / 64, p.split_k, tbx, tby, tbz);
  const int m0 = tbx * 64, n0 = tby * 64;

  const int KM = gemm_kmain(p);  // reduce indices [KM, K) are added in the epilogue of the first k-slice
  int kbeg = 0, kend = KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = tbz * chunk;
    kend = kbeg + chunk < KM ? kbeg + chunk : KM;
    if (kbeg >= KM) {
      if (tbz > 0) return;
      kbeg = kend = KM;
    }
  }
  const int nst = (kend - kbeg + 31) >> 5;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  XForm xa, xb;
  xa.act = p.a_act; xa.slope = slope; xa.thresh = p.a_drop_p > 0.f ? mesm_drop_threshold(p.a_drop_p) : 0u;
  xa.seed = p.a_drop_seed + seed_off; xa.inv_keep = 1.0f / (1.0f - p.a_drop_p);
  xa.lld = LA == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.M;
  xb.act = p.b_act; xb.slope = slope; xb.thresh = p.b_drop_p > 0.f ? mesm_drop_threshold(p.b_drop_p) : 0u;
  xb.seed = p.b_drop_seed + seed_off; xb.inv_keep = 1.0f / (1.0f - p.b_drop_p);
  xb.lld = LB == MESM_LAYOUT_REDUCE_CONTIG ? p.K : p.N;

  // wave w stages slab w of every k-tile: 0/1 = A rows [0,32) / [32,64), 2/3 = B rows [0,32) / [32,64).
  // wave_u / L_base are wave-uniform scalars, so every LDS-DMA destination is SALU arithmetic.
  const int wave_u = __builtin_amdgcn_readfirstlane(wave);
  const unsigned L_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) float*)L;
  L64Src<LA> srcA;
  L64Src<LB> srcB;
  if (wave_u < 2) srcA.init(p.A, p.lda, m0 + 32 * wave_u, p.M, kbeg, lane);
  else srcB.init(p.B, p.ldb, n0 + 32 * (wave_u - 2), p.N, kbeg, lane);
  int i_slot = 0;  // ring slot of the next k-tile to issue
  auto issue = [&](int st) {
#ifdef MESM_L64_NO_LOAD
    return;
#endif
    const unsigned slab = L_base + (unsigned)(i_slot * L64_STAGE_FLOATS + wave_u * WS_SLAB) * 4u;
    i_slot = i_slot + 1 == L64_STAGES ? 0 : i_slot + 1;
    const int kb = kbeg + 32 * st;
    if (kb + 32 <= kend) {
      if (wave_u < 2) srcA.issue_full(st, slab);
      else srcB.issue_full(st, slab);
    } else if (wave_u < 2) {
      l64_issue<LA>(p.A, p.lda, m0 + 32 * wave_u, p.M, kb, kend, slab, lane);
    } else {
      l64_issue<LB>(p.B, p.ldb, n0 + 32 * (wave_u - 2), p.N, kb, kend, slab, lane);
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float csum = 0.0f;
  const bool do_colsum = (p.colsum != nullptr) && (tby == 0) && (wn == 0);

  constexpr int AHEAD = L64_STAGES - 1;  // k-tiles in flight behind the one being multiplied
  int c_slot = 0;                        // ring slot of the k-tile being multiplied
  L64_STAMP(1);
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < nst) issue(i);
  L64_STAMP(2);
  for (int st = 0; st < nst; ++st) {
    // this wave's 4 LDS-DMA instructions of k-tile st have landed; the min(AHEAD - 1, nst - 1 - st)
    // k-tiles issued after it may still fly (4 instructions each, returned in issue order) ...
    {
      const int later = nst - 1 - st < AHEAD - 1 ? nst - 1 - st : AHEAD - 1;
      switch (later) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 1: asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); break;
        case 2: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 3: asm volatile("s_waitcnt vmcnt(12)" ::: "memory"); break;
        case 4: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        case 5: asm volatile("s_waitcnt vmcnt(20)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(24)" ::: "memory"); break;
      }
    }
    L64_STAMP(3 + 3 * st);  // this wave's loads of k-tile st landed
    // ... and so have the other waves': one barrier per k-tile.  It also orders the fragment reads of
    // k-tile st-1 (finished by every wave before it arrived here) ahead of the refill issued below.
    __builtin_amdgcn_s_barrier();
    L64_STAMP(4 + 3 * st);  // every wave's landed
    if (st + AHEAD < nst) issue(st + AHEAD);
    const float* buf = L + c_slot * L64_STAGE_FLOATS;
    c_slot = c_slot + 1 == L64_STAGES ? 0 : c_slot + 1;
    float a[4][4], b[4][4];
    ws_read<LA>(buf + wm * WS_SLAB, li, h, a);
    ws_read<LB>(buf + (2 + wn) * WS_SLAB, li, h, b);
    const int kb = kbeg + 32 * st;
    if (kb + 32 > kend) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const bool ok = kb + 8 * s_ + 4 * h + j < kend;
          a[s_][j] = ok ? a[s_][j] : 0.0f;
          b[s_][j] = ok ? b[s_][j] : 0.0f;
        }
    }
    if (XF) {
      const int gm = m0 + 32 * wm + li, gn = n0 + 32 * wn + li;
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          const int gk = kb + 8 * s_ + 4 * h + j;
          float x = mesm_act(a[s_][j], xa.act, xa.slope);
          if (xa.thresh)
            x = mesm_dropout_apply(x, (uint32_t)(LA == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gm * xa.lld + gk : (int64_t)gk * xa.lld + gm),
                                   xa.seed, xa.thresh, xa.inv_keep);
          a[s_][j] = x;
          float y = mesm_act(b[s_][j], xb.act, xb.slope);
          if (xb.thresh)
            y = mesm_dropout_apply(y, (uint32_t)(LB == MESM_LAYOUT_REDUCE_CONTIG ? (int64_t)gn * xb.lld + gk : (int64_t)gk * xb.lld + gn),
                                   xb.seed, xb.thresh, xb.inv_keep);
          b[s_][j] = y;
        }
    }
#ifdef MESM_L64_NO_MFMA
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[4 * s_ + j] += a[s_][j] * b[s_][j];
#else
#pragma unroll
    for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s_][j], b[s_][j], acc, 0, 0, 0);
#endif
    if (do_colsum) {
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) csum += a[s_][j];
    }
    L64_STAMP(5 + 3 * st);  // MFMAs of k-tile st issued
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  L64_STAMP(28);
#ifdef MESM_L64_NO_STORE
  if (acc[0] != 12345.678f) return;
#endif

  if (do_colsum) {
    csum = add_xor32(csum);
    const int gm = m0 + 32 * wm + li;
    if (tbz == 0 && KM < p.K) csum += tail_colsum<LA, XF>(p, gm, KM, xa);
    if (h == 0 && gm < p.M && csum != 0.0f) atomicAdd(p.colsum + gm, csum);
  }

  tile16_epilogue<LA, LB, XF>(p, acc, m0 + 32 * wm, n0 + 32 * wn, slope, seed_off, tbz, L, linear_block(), KM, xa, xb);
  L64_STAMP(29);  // stores issued
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  L64_STAMP(30);  // stores acknowledged
#ifdef MESM_L64_TRACE
  {
    unsigned hw;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
    unsigned xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    if (threadIdx.x == 0 && blockIdx.x < 1024 && blockIdx.z == 0) l64_trace[blockIdx.x * 32 + 31] = ((unsigned long long)xcc << 32) | hw;
  }
#endif
}

#ifdef MESM_L64_TRACE
extern "C" int mesm_l64_trace_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(l64_trace), sizeof(unsigned long long) * 1024 * 32);
}
#endif

template <int LA, int LB>
int launch_lds64_l(const MesmGemmArgs& a, hipStream_t s) {
  dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), 1, a.split_k > 1 ? a.split_k : 1);  // 1-D: xcd_tile() maps it
  const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
  const SideRed sr = take_side(s);
  if (xf) hipLaunchKernelGGL((gemm_lds64_kernel<LA, LB, true>), grid, dim3(NTHREADS), 0, s, a, sr);
  else hipLaunchKernelGGL((gemm_lds64_kernel<LA, LB, false>), grid, dim3(NTHREADS), 0, s, a, sr);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : dslope_finish(a, grid, s);
}

int launch_lds64(const MesmGemmArgs& a, hipStream_t s) {
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_lds64_l<R, R>(a, s);
  if (a.a_layout == R && a.b_layout == O) return launch_lds64_l<R, O>(a, s);
  if (a.a_layout == O && a.b_layout == O) return launch_lds64_l<O, O>(a, s);
  return launch_lds64_l<O, R>(a, s);
}

// Tape of the GEMM launches of one step (argument structs as launched), for bench.py's
// roofline measurement: recorded while a step is captured into a HIP graph (whose private
// memory pool keeps every pointer valid), replayed back-to-back from C++ with an event pair
// around every launch.
struct TapeEntry {
  std::vector<MesmGemmArgs> args;  // one problem (mesm_gemm_f32) or a group (mesm_gemm_group)
  std::vector<int> vecs;
  int n = 1;
  int group = 0;
};
struct Tape {
  bool recording = false;
  std::vector<TapeEntry> launches;
};
Tape g_tape;

// tuning switches, read ONCE when the library is loaded (they used to be getenv calls per dispatch)
int g_force_tile = []() { const char* e = getenv("MESM_GEMM_TILE"); return e ? atoi(e) : 0; }();
// MESM_GEMM_BF16X: 2 (default since round 6) = the large products on v_mfma_f32_32x32x16_f16 over operands split into two
// fp16 terms under a wave-owned power-of-two scale, THREE cross products, f32 accumulate (gemm_ws.hpp; error at or below the
// f32 MFMA kernels' own, tools/bf16x_check.py, tests/test_gemm_f16x3_gpu.py); 6 = three bf16 terms by truncation, six
// products on v_mfma_f32_32x32x16_bf16 (round 4's default);
// 0 = every product on v_mfma_f32_32x32x2_f32.  (The two-term bf16 form, 16 significand bits, failed 13 parity tests: deleted.)
int g_bf16x = []() { const char* e = getenv("MESM_GEMM_BF16X"); const int m = e ? atoi(e) : 2; return (m == 2 || m == 6) ? m : 0; }();

int mesm_gemm_force_tile() { return g_force_tile; }
int mesm_gemm_bf16x() { return g_bf16x; }
// MESM_GEMM_GROUP64: 1 = the problems of a grouped call that the 64 x 64 k-split kernel takes share ONE
// launch, and the call's other products ride along (default in the three-term split-bf16 mode: 4.40 -> 4.01
// ms/step; in exact-f32 mode it loses, 4.609 -> 4.642, and stays off; off in the experimental two-term mode as well, whose
// grouped kernel is not instantiated: every product of that mode then runs the two-term split); 0 = launched one by one
int g_group64 = []() { const char* e = getenv("MESM_GEMM_GROUP64"); return e ? atoi(e) : -1; }();
int mesm_gemm_group64() { return g_group64 >= 0 ? g_group64 : (g_bf16x != 0 ? 1 : 0); }

int dispatch(const MesmGemmArgs& a, int vec, hipStream_t s) {
  {
    // small problems (fewer than ~2 workgroups of 64x64 per CU): register-fragment kernel
    const long z = a.split_k > 1 ? a.split_k : 1;
    const long b64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * z;
    const int force = g_force_tile;  // 1 = frag kernel, 32 | 64 | 128 = staged tile, 0 = auto
    // 1 = frag, 2 = wstage (k-split 32x32, wave-private LDS-DMA), 3 = lds64 (64x64, LDS-DMA ring),
    // 4 = wstage64 (k-split 64x64),
    // 32 | 64 | 128 = register-staged tile, 0 = auto
    // tall 96 x 32 k-split tiles: the 2400-row, 256-wide GEMMs with a long reduce range (2400 x 256 x 2818 input
    // projections 52 -> 43 us, x 1024 22.3 -> 19.2, x 512 13.4 -> 12.7): 25 x 8 = 200 equal workgroups in one
    // round instead of 152.  (160 x 32 on the 4800-row GEMMs measured no gain: 14.1 vs 13.2 us at K = 256 --
    // those are bound by launch + first-load + epilogue latency, not by the second round of tiles.)
    {
      const long t96 = (long)((a.M + 95) / 96) * ((a.N + 31) / 32);
      if ((force == 6 || (force == 0 && z == 1 && a.K >= 512 && t96 >= 160 && t96 <= 256)) && wtall_ok(a))
        return launch_wtall<3>(a, s);
    }
    // experimental split-bf16 mode: the 64 x 64-per-wave kernel is the one that carries it (the split costs VALU per
    // fragment value, amortised over four products there) -- every problem of >= 2400 output rows or reduce indices
    // with enough tiles goes there, whatever its round count
    // (... or >= 512 tiles whatever the extents: the 1024 x 5003 x 256 vocabulary head used to fall through to the
    // exact-f32 64 x 64 ring, 37 us; the split kernel takes its 1,264 tiles in 28)
    if (force == 0 && bf16x_mode() != 0 && wstage_ok(a) && b64 >= 128 && (a.M >= 2400 || a.K >= 2400 || b64 >= 512) &&
        a.a_act == MESM_ACT_NONE && a.b_act == MESM_ACT_NONE && a.a_drop_p == 0.f && a.b_drop_p == 0.f)
      return launch_wstage64(a, s);
    if ((force == 3 || (force == 0 && b64 >= 512)) && wstage_ok(a)) return launch_lds64(a, s);
    // k-split 64 x 64 (half the L2 traffic per flop): wins only when its workgroups fit ONE round on the
    // 256 CUs (300 workgroups = two rounds: 4800 x 256 x 1024 59 us vs 43 us with 32 x 32 tiles) and a
    // wave has >= 4 stages to pipeline: the split-K FFN weight gradients and the 2400-row K = 1024 GEMMs
    const long kper = ((a.K + z - 1) / z + 3) / 4;
    if ((force == 4 || (force == 0 && b64 >= 128 && b64 <= 256 && kper >= 128)) && wstage_ok(a))
      return launch_wstage64(a, s);
    if ((force == 2 || (force == 0 && b64 < 512)) && wstage_ok(a)) return launch_wstage(a, s);
    if ((force == 1 || (force == 0 && b64 < 512)) && frag_ok(a)) return launch_frag(a, s);
  }
  if (vec == 4) return launch_tile<4>(a, s);
  if (vec == 2) return launch_tile<2>(a, s);
  return launch_tile<1>(a, s);
}

// argument validation / normalisation shared by the single and the grouped entry
int prepare(MesmGemmArgs& a, int& vec) {
  if (!a.A || !a.B || !a.C) return MESM_EINVAL;
  if (a.M <= 0 || a.N <= 0 || a.K <= 0) return MESM_EINVAL;
  if (a.a_layout < 0 || a.a_layout > 1 || a.b_layout < 0 || a.b_layout > 1) return MESM_EINVAL;
  if (a.e_actgrad != MESM_ACT_NONE && !a.aux) return MESM_EINVAL;
  if (a.e_actgrad == MESM_ACT_PRELU && a.dslope && !a.dslope_ws) return MESM_EINVAL;
  if (a.pre_out && (a.split_k > 1 || a.accumulate != 0 || !aligned_to(a.pre_out, 4))) return MESM_EINVAL;
  if ((a.a_act == MESM_ACT_PRELU || a.b_act == MESM_ACT_PRELU || a.e_act == MESM_ACT_PRELU ||
       a.e_actgrad == MESM_ACT_PRELU) && !a.slope)
    return MESM_EINVAL;
  if (a.a_drop_p < 0.f || a.a_drop_p >= 1.f || a.b_drop_p < 0.f || a.b_drop_p >= 1.f ||
      a.e_drop_p < 0.f || a.e_drop_p >= 1.f)
    return MESM_EINVAL;
  if (a.split_k < 1) a.split_k = 1;
  if (a.split_k > 1) {
    // (the epilogue dropout is linear in the partial sums: every k-slice applies the same mask, bias and residual join
    // the first slice; so is the ReLU gradient -- a 0 / 1 factor per element read from aux; activations and the PReLU
    // gradient with its slope reduction are not taken)
    if (a.e_act != MESM_ACT_NONE || (a.e_actgrad != MESM_ACT_NONE && a.e_actgrad != MESM_ACT_RELU)) return MESM_EINVAL;
    a.accumulate = 2;
    int max_split = (a.K + BK_MAX - 1) / BK_MAX;
    if (a.split_k > max_split) a.split_k = max_split;
  }
  if (a.accumulate < 0 || a.accumulate > 2) return MESM_EINVAL;
  if (a.out_scale == 0.0f) a.out_scale = 1.0f;
  // widest vector width every operand supports
  if (a.A2 && a.B2) return MESM_EINVAL;
  vec = 4;
  while (vec > 1) {
    bool ok = (a.lda % vec == 0) && (a.ldb % vec == 0) && aligned_to(a.A, 4 * vec) &&
              aligned_to(a.A2, 4 * vec) && aligned_to(a.B, 4 * vec) && aligned_to(a.B2, 4 * vec);
    if (a.a_layout == MESM_LAYOUT_OUTER_CONTIG) ok = ok && (a.M % vec == 0);
    if (a.b_layout == MESM_LAYOUT_OUTER_CONTIG) ok = ok && (a.N % vec == 0);
    if (ok) break;
    vec >>= 1;
  }
  if (!aligned_to(a.A, 4) || !aligned_to(a.B, 4) || !aligned_to(a.C, 4)) return MESM_EALIGN;
  return MESM_OK;
}

// does the auto dispatch send this problem to the wstage kernel (the one that can be grouped)?
bool groupable(const MesmGemmArgs& a) {
  const int force = g_force_tile;
  if (force != 0 && force != 2) return false;
  {
    const long b64g = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * (a.split_k > 1 ? a.split_k : 1);
    if (force == 0 && bf16x_mode() != 0 && b64g >= 128 && (a.M >= 2400 || a.K >= 2400 || b64g >= 512))
      return false;  // split mode: goes to the split-bf16 kernel on its own
  }
  const long z = a.split_k > 1 ? a.split_k : 1;
  const long b64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * z;
  const long kper = ((a.K + z - 1) / z + 3) / 4;
  if (force == 0 && b64 >= 128 && b64 <= 256 && kper >= 128) return false;  // goes to wstage64
  return (force == 2 || b64 < 512) && wstage_ok(a);
}

}  // namespace

extern "C" int mesm_gemm_f32(const MesmGemmArgs* args, void* stream) {
  if (!args) return MESM_EINVAL;
  MesmGemmArgs a = *args;
  int vec = 1;
  const int rc = prepare(a, vec);
  if (rc != MESM_OK) return rc;
  if (g_tape.recording) { TapeEntry e; e.args = {a}; e.vecs = {vec}; g_tape.launches.push_back(e); }
  return dispatch(a, vec, (hipStream_t)stream);
}

namespace {
int launch_group(const MesmGemmArgs* list, const int* vecs, int n, hipStream_t s) {
  // problems the wstage kernel takes go into grouped launches of up to GROUP_MAX; the rest go alone
  GroupArgs g;
  g.n = 0;
  g.start[0] = 0;
  int rc = MESM_OK;
  auto flush = [&]() {
    if (g.n == 0) return;
    if (g.n == 1) {
      rc = launch_wstage(g.p[0], s);
    } else {
      bool one = true;  // single-stage staging only if every problem of the group wants it
      for (int k = 0; k < g.n; ++k) one = one && ws_stages_for(g.p[k]) == 1;
      const SideRed sr = take_side(s);
      constexpr int O_ = MESM_LAYOUT_OUTER_CONTIG;
      int mask = 0;
      bool any_xf = false;
      for (int k = 0; k < g.n; ++k) {
        const MesmGemmArgs& a = g.p[k];
        mask |= 1 << ((a.a_layout == O_ ? 2 : 0) + (a.b_layout == O_ ? 1 : 0));
        any_xf = any_xf || a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
      }
      if (!layout_masks() || any_xf || !one) mask = 15;
      const dim3 gr(g.start[g.n]), bl(WS_THREADS);
      if (one && mask == 1) hipLaunchKernelGGL((gemm_wstage_group_kernel<1, 1, false>), gr, bl, 0, s, g, sr);
      else if (one && mask == 2) hipLaunchKernelGGL((gemm_wstage_group_kernel<1, 2, false>), gr, bl, 0, s, g, sr);
      else if (one && mask == 8) hipLaunchKernelGGL((gemm_wstage_group_kernel<1, 8, false>), gr, bl, 0, s, g, sr);
      else if (one && mask == 10) hipLaunchKernelGGL((gemm_wstage_group_kernel<1, 10, false>), gr, bl, 0, s, g, sr);
      else if (one) hipLaunchKernelGGL(gemm_wstage_group_kernel<1>, gr, bl, 0, s, g, sr);
      else hipLaunchKernelGGL(gemm_wstage_group_kernel<2>, gr, bl, 0, s, g, sr);
      rc = mesm_launch_status();
      for (int k = 0; k < g.n && rc == MESM_OK; ++k) {
        const MesmGemmArgs& a = g.p[k];
        dim3 grid((a.M + 31) / 32, (a.N + 31) / 32, a.split_k > 1 ? a.split_k : 1);
        rc = dslope_finish(a, grid, s);
      }
    }
    g.n = 0;
  };
  // A kernel costs ~9 us of ramp whatever it computes (profiles/r3c/tile_compare.txt), more than a specialised tile
  // saves on a mid-size product (the tall 96 x 32 tile on 2400 x 256 x 512: 13.4 -> 12.7 us alone): in a call that has a
  // grouped launch anyway, the products below MID_GF that the 32 x 32 kernel can take join it instead of getting a
  // kernel of their own.  MESM_GEMM_GROUP_MID=0: the standalone rule only (A/B).
  static const bool group_mid = []() { const char* e = getenv("MESM_GEMM_GROUP_MID"); return !(e && atoi(e) == 0); }();
  constexpr double MID_GF = 1.5e9;
  int n_plain = 0;
  for (int i = 0; i < n; ++i) n_plain += groupable(list[i]) ? 1 : 0;
  auto joins = [&](const MesmGemmArgs& a) {
    if (groupable(a)) return true;
    if (!group_mid || n_plain == 0 || bf16x_mode() != 0) return false;
    if (g_force_tile != 0) return false;
    return wstage_ok(a) && 2.0 * a.M * a.N * a.K < MID_GF;
  };
  // 64 x 64 class: what the single dispatch would hand to the k-split 64 x 64 kernel
  GroupArgs g64;
  g64.n = 0;
  g64.start[0] = 0;
  auto flush64 = [&]() {
    if (g64.n == 0) return;
    if (g64.n == 1) {
      rc = launch_wstage64(g64.p[0], s);
    } else {
      const SideRed sr = take_side(s);
      const int bf = bf16x_mode();
      constexpr int O_ = MESM_LAYOUT_OUTER_CONTIG;
      int mask = 0;  // the operand-layout pairs of this launch's members (see the kernel)
      for (int k = 0; k < g64.n; ++k) mask |= 1 << ((g64.p[k].a_layout == O_ ? 2 : 0) + (g64.p[k].b_layout == O_ ? 1 : 0));
      if (!layout_masks()) mask = 15;
      const dim3 gr(g64.start[g64.n]);
      if (bf == 6) hipLaunchKernelGGL(gemm_wstage64_group_kernel<6>, gr, dim3(NTHREADS), 0, s, g64, sr);
      else if (bf == 2 && mask == 1) hipLaunchKernelGGL((gemm_wstage64_group_kernel<2, 1>), gr, dim3(NTHREADS), 0, s, g64, sr);
      else if (bf == 2 && mask == 2) hipLaunchKernelGGL((gemm_wstage64_group_kernel<2, 2>), gr, dim3(NTHREADS), 0, s, g64, sr);
      else if (bf == 2 && mask == 8) hipLaunchKernelGGL((gemm_wstage64_group_kernel<2, 8>), gr, dim3(NTHREADS), 0, s, g64, sr);
      else if (bf == 2 && mask == 10) hipLaunchKernelGGL((gemm_wstage64_group_kernel<2, 10>), gr, dim3(NTHREADS), 0, s, g64, sr);
      else if (bf == 2) hipLaunchKernelGGL(gemm_wstage64_group_kernel<2>, dim3(g64.start[g64.n]), dim3(NTHREADS), 0, s, g64, sr);
      else hipLaunchKernelGGL(gemm_wstage64_group_kernel<0>, dim3(g64.start[g64.n]), dim3(NTHREADS), 0, s, g64, sr);
      rc = mesm_launch_status();
      for (int k = 0; k < g64.n && rc == MESM_OK; ++k) {
        const MesmGemmArgs& a = g64.p[k];
        rc = dslope_finish_n(a, (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64) * (a.split_k > 1 ? a.split_k : 1), s);
      }
    }
    g64.n = 0;
  };
  auto joins64 = [&](const MesmGemmArgs& a) {
    if (mesm_gemm_group64() == 0 || g_force_tile != 0 || !wstage_ok(a)) return false;
    const long z = a.split_k > 1 ? a.split_k : 1;
    const long b64 = (long)((a.M + 63) / 64) * ((a.N + 63) / 64) * z;
    const long kper = ((a.K + z - 1) / z + 3) / 4;
    const long t96 = (long)((a.M + 95) / 96) * ((a.N + 31) / 32);
    static const int min_dim = []() { const char* e = getenv("MESM_G64_MINDIM"); return e ? atoi(e) : 1024; }();
    static const int min_b64 = []() { const char* e = getenv("MESM_G64_MINB64"); return e ? atoi(e) : 128; }();
    static const int tall = []() { const char* e = getenv("MESM_G64_TALL"); return e ? atoi(e) : 1; }();
    if (!tall && z == 1 && a.K >= 512 && t96 >= 160 && t96 <= 256 && wtall_ok(a)) return false;  // the tall tile's shapes
    const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
    if (bf16x_mode() != 0) return !xf && b64 >= min_b64 && (a.M >= min_dim || a.K >= min_dim);
    return b64 >= 128 && b64 <= 256 && kper >= 128;
  };
  // a call that has a 64 x 64 launch anyway: products of >= join_mf MFLOP that the 32 x 32 group would take ride along
  static const double join_mf = []() { const char* e = getenv("MESM_G64_JOIN_MF"); return e ? atof(e) : 0.0; }();
  int n64 = 0;
  for (int i = 0; i < n; ++i) n64 += joins64(list[i]) ? 1 : 0;
  auto rides64 = [&](const MesmGemmArgs& a) {
    if (n64 == 0 || join_mf < 0.0 || mesm_gemm_group64() == 0 || !joins(a)) return false;
    const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
    if (xf && bf16x_mode() != 0) return false;
    return 2.0 * a.M * a.N * a.K >= join_mf * 1e6;
  };
  for (int i = 0; i < n && rc == MESM_OK; ++i) {
    const MesmGemmArgs& a = list[i];
    if (joins64(a) || rides64(a)) {
      const int wgs = ((a.M + 63) / 64) * ((a.N + 63) / 64) * (a.split_k > 1 ? a.split_k : 1);
      g64.p[g64.n] = a;
      g64.start[g64.n + 1] = g64.start[g64.n] + wgs;
      if (++g64.n == GROUP_MAX) flush64();
    } else if (joins(a)) {
      const int wgs = ((a.M + 31) / 32) * ((a.N + 31) / 32) * (a.split_k > 1 ? a.split_k : 1);
      g.p[g.n] = a;
      g.start[g.n + 1] = g.start[g.n] + wgs;
      if (++g.n == GROUP_MAX) flush();
    } else {
      rc = dispatch(a, vecs[i], s);
    }
  }
  if (rc == MESM_OK) flush();
  if (rc == MESM_OK) flush64();
  static const bool plan_log = getenv("MESM_GEMM_PLAN_LOG") != nullptr;  // tools/gemm_plan.py: what a call turned into
  if (plan_log) {
    fprintf(stderr, "[gemm plan] call of %d:", n);
    for (int i = 0; i < n; ++i) {
      const MesmGemmArgs& a = list[i];
      const bool xf = a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f;
      const char* where = (joins64(a) || rides64(a)) ? "g64" : (joins(a) ? "g32" : "alone");
      fprintf(stderr, " %dx%dx%d%s%s/s%d%s%s->%s", a.M, a.N, a.K, a.a_layout == MESM_LAYOUT_OUTER_CONTIG ? "T" : "N",
              a.b_layout == MESM_LAYOUT_REDUCE_CONTIG ? "T" : "N", a.split_k, xf ? "(xf)" : "", (a.A2 || a.B2) ? "(2nd)" : "", where);
    }
    fprintf(stderr, "\n");
  }
  return rc;
}
}  // namespace

extern "C" int mesm_gemm_group(const MesmGemmArgs* args, int32_t n, void* stream) {
  if (!args || n <= 0 || n > 64) return MESM_EINVAL;
  MesmGemmArgs list[64];
  int vecs[64];
  for (int i = 0; i < n; ++i) {
    list[i] = args[i];
    const int rc = prepare(list[i], vecs[i]);
    if (rc != MESM_OK) return rc;
  }
  // longest tiles first: the members of a call are independent, a workgroup's time is its slice of the reduce range, and
  // the hardware hands out workgroups in index order -- a long-K member (a split-K weight gradient: 19 stages per wave)
  // behind short ones (dX at K = 256: 1-2 stages) would start its tiles last and end the launch alone
  static const bool lpt = [] { const char* e = getenv("MESM_GEMM_LPT"); return !e || atoi(e) != 0; }();
  if (lpt && n > 1) {
    int order[64];
    for (int i = 0; i < n; ++i) order[i] = i;
    auto depth = [&](int i) { return (list[i].K + list[i].split_k - 1) / list[i].split_k; };
    std::stable_sort(order, order + n, [&](int a, int b) { return depth(a) > depth(b); });
    MesmGemmArgs l2[64];
    int v2[64];
    for (int i = 0; i < n; ++i) { l2[i] = list[order[i]]; v2[i] = vecs[order[i]]; }
    for (int i = 0; i < n; ++i) { list[i] = l2[i]; vecs[i] = v2[i]; }
  }
  if (g_tape.recording) {
    TapeEntry e;
    e.args.assign(list, list + n);
    e.vecs.assign(vecs, vecs + n);
    e.group = 1;
    e.n = n;
    g_tape.launches.push_back(e);
  }
  return launch_group(list, vecs, n, (hipStream_t)stream);
}

int mesm_gemm_dslope_finish(const MesmGemmArgs& a, int64_t nblocks, hipStream_t s) { return dslope_finish_n(a, nblocks, s); }

// tuning tools flip the two dispatch switches between calls of one process (< 0: keep)
extern "C" int mesm_gemm_get_bf16x(void) { return g_bf16x; }

extern "C" int mesm_gemm_set_switches(int32_t force_tile, int32_t bf16x) {
  if (force_tile >= 0) g_force_tile = force_tile;
  if (bf16x >= 0) g_bf16x = (bf16x == 2 || bf16x == 6) ? bf16x : 0;
  return MESM_OK;
}

extern "C" int mesm_gemm_flush_side(void* stream) { return flush_side((hipStream_t)stream); }
// forget what is pending without reducing it (error paths: the workspaces may be gone)
extern "C" int mesm_gemm_drop_side(void) {
  g_side.clear();
  return MESM_OK;
}

extern "C" int mesm_gemm_tape(int32_t record) {
  if (record) g_tape.launches.clear();
  g_tape.recording = record != 0;
  return MESM_OK;
}

extern "C" int mesm_gemm_tape_replay(void* stream, int32_t reps, double* total_ms, int64_t* launches,
                                     double* total_flops, double* total_bytes) {
  if (!total_ms || !launches || !total_flops || !total_bytes || reps < 1) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const size_t n = g_tape.launches.size();
  // ONE event pair per repetition around the back-to-back launches of the whole tape (an event pair
  // per launch adds ~5 us of its own to these 5-50 us kernels): average launch duration = total / n.
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MESM_ELAUNCH;
  double ms = 0.0, flops = 0.0, bytes = 0.0;
  int rc = MESM_OK;
  for (int r = 0; r < reps && rc == MESM_OK; ++r) {
    hipEventRecord(e0, s);
    for (size_t i = 0; i < n && rc == MESM_OK; ++i) {
      const TapeEntry& e = g_tape.launches[i];
      rc = e.group ? launch_group(e.args.data(), e.vecs.data(), e.n, s) : dispatch(e.args[0], e.vecs[0], s);
    }
    if (rc == MESM_OK) rc = flush_side(s);
    hipEventRecord(e1, s);
    hipStreamSynchronize(s);
    float t = 0.0f;
    hipEventElapsedTime(&t, e0, e1);
    ms += t;
    for (size_t i = 0; i < n; ++i)
      for (const MesmGemmArgs& a : g_tape.launches[i].args) {
        flops += 2.0 * (double)a.M * (double)a.N * (double)a.K;
        bytes += 4.0 * ((double)a.M * a.K + (double)a.K * a.N + (double)a.M * a.N);
      }
  }
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *total_ms = ms;
  *launches = (int64_t)n * reps;
  *total_flops = flops;
  *total_bytes = bytes;
  return rc;
}

// Algorithmic bytes of ONE pass over the tape, two ways: `operands` = A + B read once, C written once (the figure
// mesm_gemm_tape_replay reports); `with_sides` = that plus every side matrix the fused epilogue / prologue must touch
// once: second operands (A2, B2), the residual, the activation-gradient aux, a read of C when the launch accumulates,
// the second output.  Bias / colsum / slope vectors and split-K partial sums are not counted.
extern "C" int mesm_gemm_tape_bytes(double* operands, double* with_sides) {
  if (!operands || !with_sides) return MESM_EINVAL;
  double o = 0.0, w = 0.0;
  for (const TapeEntry& e : g_tape.launches)
    for (const MesmGemmArgs& a : e.args) {
      const double mk = (double)a.M * a.K, kn = (double)a.K * a.N, mn = (double)a.M * a.N;
      o += 4.0 * (mk + kn + mn);
      w += 4.0 * (mk * (a.A2 ? 2 : 1) + kn * (a.B2 ? 2 : 1) +
                  mn * (1 + (a.accumulate ? 1 : 0) + (a.residual ? 1 : 0) +
                        ((a.aux && a.e_actgrad != MESM_ACT_NONE) ? 1 : 0) + (a.pre_out ? 1 : 0)));
    }
  *operands = o;
  *with_sides = w;
  return MESM_OK;
}

// One entry of the tape timed alone: `reps` back-to-back launches of launch `idx` under one event pair, and what
// it carries (tools/tape_profile.py: which launches of the step lose the most time against the big-GEMM rate).
// shapes: up to 64 x (M, N, K, split_k) int32; returns the number of problems in *n_problems.
extern "C" int mesm_gemm_tape_entry(void* stream, int32_t idx, int32_t reps, double* ms, int32_t* n_problems,
                                    int32_t* shapes) {
  if (!ms || !n_problems || !shapes || reps < 1) return MESM_EINVAL;
  if (idx < 0 || (size_t)idx >= g_tape.launches.size()) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  const TapeEntry& e = g_tape.launches[idx];
  hipEvent_t e0, e1;
  if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess) return MESM_ELAUNCH;
  int rc = MESM_OK;
  hipEventRecord(e0, s);
  for (int r = 0; r < reps && rc == MESM_OK; ++r)
    rc = e.group ? launch_group(e.args.data(), e.vecs.data(), e.n, s) : dispatch(e.args[0], e.vecs[0], s);
  if (rc == MESM_OK) rc = flush_side(s);
  hipEventRecord(e1, s);
  hipStreamSynchronize(s);
  float t = 0.0f;
  hipEventElapsedTime(&t, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  *ms = (double)t / reps;
  *n_problems = (int32_t)e.args.size();
  for (size_t k = 0; k < e.args.size() && k < 64; ++k) {
    const MesmGemmArgs& a = e.args[k];
    shapes[4 * k] = a.M; shapes[4 * k + 1] = a.N; shapes[4 * k + 2] = a.K;
    shapes[4 * k + 3] = (a.split_k > 1 ? a.split_k : 1) | (a.a_layout << 8) | (a.b_layout << 9) | (groupable(a) ? 1 << 10 : 0);
  }
  return rc;
}

extern "C" int mesm_gemm_tape_size(void) { return (int)g_tape.launches.size(); }

