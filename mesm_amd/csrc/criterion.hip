// Fused criterion blocks (model/criterion.py of the reference): each loss is ONE forward and
// ONE backward launch instead of a chain of ~50 element-wise ATen kernels.  A serial HIP-graph
// chain costs ~6 us per launch on MI355X whatever the kernel does, so for these latency-bound
// reductions the launch count is the cost.
//
//   set loss   : Hungarian match (matcher.py:39-117) + loss_spans (criterion.py:71-110)
//                + loss_labels (:112-137) for one decoder layer
//   rec_ss     : loss_rec_ss (:223-274)
//   rec_fw     : reductions after cal_nll_loss (:276-306)
//   sal score  : saliency dot product (model.py:301-302)
//   text prep  : post_process_text (model.py:145-152)
//   wsum       : total = sum_k w_k * loss_k (criterion.py:361-365)
#include "common.hpp"
#include "loss_bodies.hpp"

namespace {

// ------------------------------------------------------------------------------------------
// 1-D generalized IoU of a predicted span (x1,x2) with a target (g1,g2): span_utils.py:92-121
struct Giou {
  float inter, uni, enc, raw_i, raw_e;
};
__device__ __forceinline__ float giou_1d(float x1, float x2, float g1, float g2, Giou& s) {
  s.raw_i = fminf(x2, g2) - fmaxf(x1, g1);
  s.inter = fmaxf(s.raw_i, 0.0f);
  s.uni = (x2 - x1) + (g2 - g1) - s.inter;
  s.raw_e = fmaxf(x2, g2) - fminf(x1, g1);
  s.enc = fmaxf(s.raw_e, 0.0f);
  return s.inter / s.uni - (s.enc - s.uni) / s.enc;
}

__device__ __forceinline__ float tie(float a, float b) {  // d max(a,b)/da with torch's tie split
  return a > b ? 1.0f : (a == b ? 0.5f : 0.0f);
}

// ------------------------------------------------------------------------------------------
// Shortest-augmenting-path assignment in fp64 on the fp32 costs -- the algorithm of
// scipy.optimize.linear_sum_assignment (matcher.py:108-117 calls it on a (Q x T) cost block of any shape).
// Like scipy the solver walks the SHORTER side as rows: targets when T < Q (every target gets a query),
// queries when T >= Q (every query gets a target, T - Q targets stay unmatched: match_q = -1).
// Work arrays live in LDS, element i of thread `tid` at [i * nthr + tid] (conflict-free).
struct Sap {
  double *cost, *u, *v, *minv;
  int *p, *way;
  uint8_t* used;
  int nthr, tid, Q;
  __device__ __forceinline__ double& C(int t, int q) { return cost[(t * Q + q) * nthr + tid]; }
  __device__ __forceinline__ double& U(int i) { return u[i * nthr + tid]; }
  __device__ __forceinline__ double& V(int j) { return v[j * nthr + tid]; }
  __device__ __forceinline__ double& MV(int j) { return minv[j * nthr + tid]; }
  __device__ __forceinline__ int& P(int j) { return p[j * nthr + tid]; }
  __device__ __forceinline__ int& W(int j) { return way[j * nthr + tid]; }
  __device__ __forceinline__ uint8_t& US(int j) { return used[j * nthr + tid]; }
  // carve the arrays for `nthr` problems of at most Tmax x Q out of a dynamic LDS block
  __device__ void carve(unsigned char* smem, int Tmax, int Q_, int nthr_, int tid_) {
    nthr = nthr_; tid = tid_; Q = Q_;
    const int M1 = (Tmax > Q_ ? Tmax : Q_) + 1;
    double* dp = reinterpret_cast<double*>(smem);
    cost = dp; dp += (size_t)Tmax * Q_ * nthr;
    u = dp; dp += (size_t)M1 * nthr;
    v = dp; dp += (size_t)M1 * nthr;
    minv = dp; dp += (size_t)M1 * nthr;
    int* ip = reinterpret_cast<int*>(dp);
    p = ip; ip += (size_t)M1 * nthr;
    way = ip; ip += (size_t)M1 * nthr;
    used = reinterpret_cast<uint8_t*>(ip);
  }
};

// QROWS = false: rows = targets 1..T, columns = queries 1..Q (T < Q); true: rows = queries, columns = targets.
// On return P(j) = the row assigned to column j (0 = none).
template <bool QROWS>
__device__ void sap_solve(Sap& s, int T, int Q) {
  const int nR = QROWS ? Q : T, nC = QROWS ? T : Q;
  for (int i = 0; i <= nR; ++i) s.U(i) = 0.0;
  for (int j = 0; j <= nC; ++j) { s.V(j) = 0.0; s.P(j) = 0; s.W(j) = 0; }
  for (int i = 1; i <= nR; ++i) {
    s.P(0) = i;
    int j0 = 0;
    for (int j = 0; j <= nC; ++j) { s.MV(j) = 1e300; s.US(j) = 0; }
    do {
      s.US(j0) = 1;
      const int i0 = s.P(j0);
      double delta = 1e300;
      int j1 = 0;
      const double ui0 = s.U(i0);
      for (int j = 1; j <= nC; ++j) {
        if (!s.US(j)) {
          const double cur = (QROWS ? s.C(j - 1, i0 - 1) : s.C(i0 - 1, j - 1)) - ui0 - s.V(j);
          double mv = s.MV(j);
          if (cur < mv) { mv = cur; s.MV(j) = cur; s.W(j) = j0; }
          if (mv < delta) { delta = mv; j1 = j; }
        }
      }
      for (int j = 0; j <= nC; ++j) {
        if (s.US(j)) { s.U(s.P(j)) += delta; s.V(j) -= delta; }
        else s.MV(j) -= delta;
      }
      j0 = j1;
    } while (s.P(j0) != 0);
    do {
      const int j1 = s.W(j0);
      s.P(j0) = s.P(j1);
      j0 = j1;
    } while (j0);
  }
}

__host__ __device__ inline size_t sap_bytes_per_thread(int Tmax, int Q) {
  // doubles: cost T*Q, u / v / minv max(T,Q)+1 each;  ints: p, way;  bytes: used (padded to 8)
  const size_t M1 = (size_t)(Tmax > Q ? Tmax : Q) + 1;
  return ((size_t)Tmax * Q + 3 * M1) * 8 + 2 * M1 * 4 + (M1 + 7) / 8 * 8;
}
// pairs solved side by side by one workgroup (one thread each) within 60 KB of LDS; 0 = does not fit
inline int sap_pairs_per_pass(int Tmax, int Q, int N) {
  const size_t per = sap_bytes_per_thread(Tmax, Q);
  int P = 64;
  while (P > 1 && per * P > 60 * 1024) P >>= 1;
  if (per * P > 60 * 1024) return 0;
  while (P > 1 && P / 2 >= N) P >>= 1;
  return P;
}

// the wave-per-pair form (set losses): waves of a 1,024-thread workgroup that work side by side, a (Tmax x Q) block of
// doubles in LDS each, within 60 KB
constexpr int SAPW_THREADS = 1024;
inline int sapw_waves_per_pass(int Tmax, int Q, int N) {
  const size_t per = (size_t)Tmax * Q * 8;
  int P = SAPW_THREADS / 64;
  while (P > 1 && (per * P > 60 * 1024 || P / 2 >= N)) P >>= 1;
  return per * P > 60 * 1024 ? 0 : P;
}

// fp32 cost block of pair b exactly as matcher.py:70-105 builds it, widened to fp64 for the solver
__device__ __forceinline__ void sap_fill_cost(Sap& s, const float* __restrict__ logits, const float* __restrict__ spans,
                                              const float* __restrict__ tgt_cxw, const float* __restrict__ tgt_xx,
                                              int b, int t0, int T, int Q, float w_span, float w_giou, float w_class,
                                              float* __restrict__ cost_out, int Tmax) {
  for (int q = 0; q < Q; ++q) {
    const float l0 = logits[((int64_t)b * Q + q) * 2], l1 = logits[((int64_t)b * Q + q) * 2 + 1];
    const float mx = fmaxf(l0, l1);
    const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
    const float prob0 = e0 / (e0 + e1);
    const float cx = spans[((int64_t)b * Q + q) * 2], w = spans[((int64_t)b * Q + q) * 2 + 1];
    const float x1 = cx - 0.5f * w, x2 = cx + 0.5f * w;
    for (int t = 0; t < T; ++t) {
      const float tc = tgt_cxw[(int64_t)(t0 + t) * 2], tw = tgt_cxw[(int64_t)(t0 + t) * 2 + 1];
      const float g1 = tgt_xx[(int64_t)(t0 + t) * 2], g2 = tgt_xx[(int64_t)(t0 + t) * 2 + 1];
      const float c_span = fabsf(cx - tc) + fabsf(w - tw);
      Giou gi;
      const float giou = giou_1d(x1, x2, g1, g2, gi);
      const float c = w_span * c_span + w_giou * (-giou) + w_class * (-prob0);
      s.C(t, q) = (double)c;
      if (cost_out) cost_out[((int64_t)b * Q + q) * Tmax + t] = c;
    }
  }
}

// the assignment of pair b: match_q[t0 + t] = query of target t, -1 where the target stays unmatched (T > Q)
__device__ __forceinline__ void sap_assign(Sap& s, int t0, int T, int Q, int32_t* __restrict__ match_q) {
  if (T < Q) {
    sap_solve<false>(s, T, Q);
    for (int j = 1; j <= Q; ++j)
      if (s.P(j) != 0) match_q[t0 + s.P(j) - 1] = j - 1;
  } else {
    sap_solve<true>(s, T, Q);
    for (int j = 1; j <= T; ++j) match_q[t0 + j - 1] = s.P(j) - 1;
  }
}

// matcher.py:39-117 alone (one thread per pair, blockDim.x pairs per workgroup)
__global__ __launch_bounds__(64) void match_kernel(
    const float* __restrict__ logits, const float* __restrict__ spans,
    const float* __restrict__ tgt_cxw, const float* __restrict__ tgt_xx,
    const int32_t* __restrict__ tgt_off, int N, int Q, int Tmax, float w_span, float w_giou,
    float w_class, float* __restrict__ cost_out, int32_t* __restrict__ match_q) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  Sap s;
  s.carve(smem, Tmax, Q, blockDim.x, threadIdx.x);
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= N) return;
  const int t0 = tgt_off[b];
  const int T = tgt_off[b + 1] - t0;
  sap_fill_cost(s, logits, spans, tgt_cxw, tgt_xx, b, t0, T, Q, w_span, w_giou, w_class, cost_out, Tmax);
  sap_assign(s, t0, T, Q, match_q);
}

// ------------------------------------------------------------------------------------------
// The same assignment with one WAVE per pair (round 5): lane = column.  The thread-per-pair form above is a chain of
// dependent LDS round trips and fp64 operations -- 33 us for 32 pairs of 10 queries x <= 5 targets, with 31 of a wave's
// lanes waiting on the longest pair -- and sat on the critical path of the criterion forward.  Here a column's state
// (v, minv, way, used, its row p and THAT ROW's potential u) lives in its lane's registers; a step of the search is one
// LDS read of the cost column, the elementwise fp64 updates of the sequential algorithm (the same operations on the same
// operands: the potentials are bit-identical), a DPP minimum and a ballot for the first column that attains it (the
// sequential scan's strict `<` keeps the lowest index too).  u travels with its row when the augmenting path reassigns it.
__device__ __forceinline__ double dpp_min_f64_row(double v) {  // every lane: the minimum over its row of 16 lanes
#define MESM_DMIN_STEP(CTRL)                                                                                   \
  {                                                                                                            \
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, 0xF, 0xF, false);   \
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, 0xF, 0xF, false);   \
    const double o = __hiloint2double(hi, lo);                                                                 \
    v = o < v ? o : v;                                                                                         \
  }
  MESM_DMIN_STEP(0xB1)   // quad_perm [1,0,3,2]
  MESM_DMIN_STEP(0x4E)   // quad_perm [2,3,0,1]
  MESM_DMIN_STEP(0x141)  // row_half_mirror
  MESM_DMIN_STEP(0x140)  // row_mirror
#undef MESM_DMIN_STEP
  return v;
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), l), __builtin_amdgcn_readlane(__double2loint(v), l));
}

// cost: the pair's (T x Q) block in LDS as doubles, [t * Q + q].  QROWS as in sap_solve.  Returns this lane's column's row
// (1-based, 0 = none): lane = query (QROWS false) or target (QROWS true).
template <bool QROWS>
__device__ __forceinline__ int sapw_solve(const double* __restrict__ cost, int T, int Q, int lane) {
  const int nR = QROWS ? Q : T, nC = QROWS ? T : Q;
  const bool col = lane < nC;
  double v = 0.0, ucol = 0.0;
  int p = 0;
  for (int i = 1; i <= nR; ++i) {
    double u_i = 0.0;  // the potential of row i, which sits in the virtual column 0 during the search
    double minv = 1e300;
    int way = 0, j0 = 0;
    bool used = false;
    for (int guard = 0; guard <= nC + 1; ++guard) {  // (a column joins the tree per turn: nC turns at most; NaN costs end here)
      if (lane == j0 - 1) used = true;
      const int i0 = j0 ? __builtin_amdgcn_readlane(p, j0 - 1) : i;
      const double ui0 = j0 ? readlane_f64(ucol, j0 - 1) : u_i;
      double key = 1e300;
      if (col && !used) {
        const double c = QROWS ? cost[lane * Q + (i0 - 1)] : cost[(i0 - 1) * Q + lane];
        const double cur = c - ui0 - v;
        if (cur < minv) { minv = cur; way = j0; }
        key = minv;
      }
      double m = dpp_min_f64_row(key);
      if (nC > 16) {
        const double m1 = readlane_f64(m, 16), m2 = readlane_f64(m, 32), m3 = readlane_f64(m, 48);
        m = readlane_f64(m, 0);
        m = m1 < m ? m1 : m; m = m2 < m ? m2 : m; m = m3 < m ? m3 : m;
      } else {
        m = readlane_f64(m, 0);
      }
      const double delta = m;
      int j1 = 0;
      if (delta < 1e300) j1 = __ffsll((unsigned long long)__ballot(key == delta));  // first column at the minimum, 1-based
      if (used) { ucol += delta; v -= delta; }
      else if (col) minv -= delta;
      u_i += delta;
      j0 = j1;
      if (j0 == 0 || __builtin_amdgcn_readlane(p, j0 - 1) == 0) break;
    }
    for (int guard = 0; j0 && guard <= nC + 1; ++guard) {  // flip the path: column j0 takes the row of the column before it
      const int j1 = __builtin_amdgcn_readlane(way, j0 - 1);
      const int pj1 = j1 ? __builtin_amdgcn_readlane(p, j1 - 1) : i;
      const double uj1 = j1 ? readlane_f64(ucol, j1 - 1) : u_i;
      if (lane == j0 - 1) { p = pj1; ucol = uj1; }
      j0 = j1;
    }
  }
  return col ? p : 0;
}

// One wave per pair, `P` pairs per pass (the cost blocks' LDS budget); one workgroup, so the loss sums are reduced without
// atomics (deterministic).  out[0..3] = loss_span, loss_giou, loss_label, class_error.  The span terms are means over the
// MATCHED (query, target) pairs: sum_b min(T_b, Q) of them (criterion.py:104-107, :133).
__device__ __forceinline__ void set_loss_fwd_body(const float* __restrict__ logits, const float* __restrict__ spans,
                                                  const float* __restrict__ tgt_cxw, const float* __restrict__ tgt_xx,
                                                  const int32_t* __restrict__ tgt_off, int N, int Q, int Tmax,
                                                  float w_span, float w_giou, float w_class, float eos_coef, int P,
                                                  int32_t* __restrict__ match_q, float* __restrict__ out,
                                                  const int32_t* __restrict__ n_valid) {
  // n_valid (device scalar, NULL = N): the pairs [n_valid, N) are padding of a captured capacity (graphed.py):
  // they take no part in the matching, the sums or the denominators
  if (n_valid) N = *n_valid;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nw = (int)(blockDim.x >> 6);
  double* cost = reinterpret_cast<double*>(smem) + (size_t)wave * Tmax * Q;
  __shared__ float red[5][16];

  float a_l1 = 0.0f, a_giou = 0.0f, a_ce = 0.0f, a_ok = 0.0f, a_cnt = 0.0f;
  if (wave < P) {
    for (int b = wave; b < N; b += P) {
      const int t0 = tgt_off[b];
      const int T = tgt_off[b + 1] - t0;
      const float* lg = logits + (int64_t)b * Q * 2;
      const float* sp = spans + (int64_t)b * Q * 2;
      // fp32 cost block exactly as matcher.py:70-105 builds it, widened to fp64 for the solver (sap_fill_cost)
      for (int e = lane; e < T * Q; e += 64) {
        const int t = e / Q, q = e - t * Q;
        const float l0 = lg[q * 2], l1 = lg[q * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
        const float prob0 = e0 / (e0 + e1);
        const float cx = sp[q * 2], w = sp[q * 2 + 1];
        const float tc = tgt_cxw[(int64_t)(t0 + t) * 2], tw = tgt_cxw[(int64_t)(t0 + t) * 2 + 1];
        const float g1 = tgt_xx[(int64_t)(t0 + t) * 2], g2 = tgt_xx[(int64_t)(t0 + t) * 2 + 1];
        const float c_span = fabsf(cx - tc) + fabsf(w - tw);
        Giou gi;
        const float giou = giou_1d(cx - 0.5f * w, cx + 0.5f * w, g1, g2, gi);
        cost[e] = (double)(w_span * c_span + w_giou * (-giou) + w_class * (-prob0));
      }
      // (a wave's LDS writes are visible to its own later reads: no barrier between the waves' independent problems)
      int q = -1, t = -1;  // this lane's matched (query, target), if it has one
      bool fg = false;     // lane = query: is it matched
      if (T < Q) {
        const int r = sapw_solve<false>(cost, T, Q, lane);  // lane = query, r = its target (1-based)
        if (r) { q = lane; t = t0 + r - 1; match_q[t] = lane; }
        fg = r != 0;
      } else {
        const int r = sapw_solve<true>(cost, T, Q, lane);  // lane = target, r = its query (0: the target stays unmatched)
        if (lane < T) match_q[t0 + lane] = r - 1;
        if (r) { q = r - 1; t = t0 + lane; }
        fg = true;  // every query is a row of the assignment: all matched
      }
      if (q >= 0) {
        a_cnt += 1.0f;
        const float cx = sp[q * 2], w = sp[q * 2 + 1];
        a_l1 += fabsf(cx - tgt_cxw[(int64_t)t * 2]) + fabsf(w - tgt_cxw[(int64_t)t * 2 + 1]);
        Giou gi;
        a_giou += 1.0f - giou_1d(cx - 0.5f * w, cx + 0.5f * w, tgt_xx[(int64_t)t * 2], tgt_xx[(int64_t)t * 2 + 1], gi);
        a_ok += (lg[q * 2] >= lg[q * 2 + 1]) ? 1.0f : 0.0f;  // argmax == foreground (first max wins)
      }
      if (lane < Q) {
        const float l0 = lg[lane * 2], l1 = lg[lane * 2 + 1];
        const float mx = fmaxf(l0, l1);
        const float lse = mx + logf(expf(l0 - mx) + expf(l1 - mx));
        a_ce += fg ? -(l0 - lse) : -(l1 - lse) * eos_coef;
      }
    }
  }
  float vals[5] = {a_l1, a_giou, a_ce, a_ok, a_cnt};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    float v = wave_sum(vals[k]);
    if (lane == 0) red[k][wave] = v;
  }
  __syncthreads();
  if (tid == 0) {
    float r[5] = {0, 0, 0, 0, 0};
    for (int k = 0; k < 5; ++k)
      for (int w = 0; w < nw; ++w) r[k] += red[k][w];
    const float nm = r[4];
    out[0] = r[0] / (2.0f * nm);
    out[1] = r[1] / nm;
    out[2] = r[2] / (float)(N * Q);
    out[3] = 100.0f - r[3] * (100.0f / nm);
  }
}

__global__ __launch_bounds__(SAPW_THREADS) void set_loss_fwd_kernel(const float* __restrict__ logits, const float* __restrict__ spans,
                                    const float* __restrict__ tgt_cxw, const float* __restrict__ tgt_xx,
                                    const int32_t* __restrict__ tgt_off, int N, int Q, int Tmax,
                                    float w_span, float w_giou, float w_class, float eos_coef, int P,
                                    int32_t* __restrict__ match_q, float* __restrict__ out,
                                    const int32_t* __restrict__ n_valid) {
  set_loss_fwd_body(logits, spans, tgt_cxw, tgt_xx, tgt_off, N, Q, Tmax, w_span, w_giou, w_class, eos_coef, P, match_q, out,
                    n_valid);
}

// The decoder layers' set losses (main + auxiliary, criterion.py:338-357) are independent one-workgroup latency chains
// (a wave per pair, the pairs' loss terms summed inside the workgroup): one launch, a workgroup per layer.
constexpr int SET_LAYERS_MAX = 8;
struct SetLossLayers {
  const float* logits[SET_LAYERS_MAX];
  const float* spans[SET_LAYERS_MAX];
  int32_t* match_q[SET_LAYERS_MAX];
  float* out[SET_LAYERS_MAX];
};
__global__ __launch_bounds__(SAPW_THREADS) void set_loss_fwd_layers_kernel(const SetLossLayers L, const float* __restrict__ tgt_cxw,
                                           const float* __restrict__ tgt_xx, const int32_t* __restrict__ tgt_off, int N,
                                           int Q, int Tmax, float w_span, float w_giou, float w_class, float eos_coef, int P,
                                           const int32_t* __restrict__ n_valid) {
  const float *lg = L.logits[0], *sp = L.spans[0];
  int32_t* mq = L.match_q[0];
  float* out = L.out[0];
#pragma unroll
  for (int k = 1; k < SET_LAYERS_MAX; ++k)
    if ((int)blockIdx.x == k) { lg = L.logits[k]; sp = L.spans[k]; mq = L.match_q[k]; out = L.out[k]; }
  set_loss_fwd_body(lg, sp, tgt_cxw, tgt_xx, tgt_off, N, Q, Tmax, w_span, w_giou, w_class, eos_coef, P, mq, out, n_valid);
}

// one thread per (pair, query): g_span / g_giou / g_label = upstream gradients of loss_span, loss_giou, loss_label
__device__ __forceinline__ void set_loss_bwd_body(
    const float* __restrict__ logits, const float* __restrict__ spans, const float* __restrict__ tgt_cxw,
    const float* __restrict__ tgt_xx, const int32_t* __restrict__ tgt_off,
    const int32_t* __restrict__ match_q, int N, int Q, float eos_coef, float g_span, float g_giou, float g_label,
    float* __restrict__ dlogits, float* __restrict__ dspans, const int32_t* __restrict__ n_valid, int bid) {
  const int i = bid * blockDim.x + threadIdx.x;
  const int Ncap = N;
  if (n_valid) N = *n_valid;  // padding pairs: zero gradients, and out of every denominator
  // matched (query, target) pairs of the batch = sum_b min(T_b, Q): the denominators of the span terms
  __shared__ float sh_cnt[4];
  float cnt = 0.0f;
  for (int b2 = threadIdx.x; b2 < N; b2 += blockDim.x) cnt += (float)min(tgt_off[b2 + 1] - tgt_off[b2], Q);
  cnt = wave_sum(cnt);
  if ((threadIdx.x & 63) == 0) sh_cnt[threadIdx.x >> 6] = cnt;
  __syncthreads();
  const float sumT = sh_cnt[0] + sh_cnt[1] + sh_cnt[2] + sh_cnt[3];
  if (i >= Ncap * Q) return;
  const int b = i / Q, q = i % Q;
  if (b >= N) {
    dlogits[(int64_t)i * 2] = dlogits[(int64_t)i * 2 + 1] = 0.0f;
    dspans[(int64_t)i * 2] = dspans[(int64_t)i * 2 + 1] = 0.0f;
    return;
  }
  const int t0 = tgt_off[b], T = tgt_off[b + 1] - t0;
  int t = -1;
  for (int k = 0; k < T; ++k)
    if (match_q[t0 + k] == q) t = t0 + k;
  const float l0 = logits[(int64_t)i * 2], l1 = logits[(int64_t)i * 2 + 1];
  const float mx = fmaxf(l0, l1);
  const float e0 = expf(l0 - mx), e1 = expf(l1 - mx);
  const float p0 = e0 / (e0 + e1), p1 = e1 / (e0 + e1);
  const float gl = g_label / (float)(N * Q) * (t >= 0 ? 1.0f : eos_coef);
  // d(-logp[cls])/dl_c = p_c - [c == cls]
  dlogits[(int64_t)i * 2] = gl * (p0 - (t >= 0 ? 1.0f : 0.0f));
  dlogits[(int64_t)i * 2 + 1] = gl * (p1 - (t >= 0 ? 0.0f : 1.0f));
  float dcx = 0.0f, dw = 0.0f;
  if (t >= 0) {
    const float cx = spans[(int64_t)i * 2], w = spans[(int64_t)i * 2 + 1];
    const float tc = tgt_cxw[(int64_t)t * 2], tw = tgt_cxw[(int64_t)t * 2 + 1];
    const float ks = g_span / (2.0f * sumT);
    const float a = cx - tc, c = w - tw;
    dcx += ks * (a > 0.0f ? 1.0f : (a < 0.0f ? -1.0f : 0.0f));
    dw += ks * (c > 0.0f ? 1.0f : (c < 0.0f ? -1.0f : 0.0f));
    const float x1 = cx - 0.5f * w, x2 = cx + 0.5f * w;
    const float g1 = tgt_xx[(int64_t)t * 2], g2 = tgt_xx[(int64_t)t * 2 + 1];
    Giou s;
    giou_1d(x1, x2, g1, g2, s);
    const float ci = s.raw_i >= 0.0f ? 1.0f : 0.0f, ce = s.raw_e >= 0.0f ? 1.0f : 0.0f;
    const float dI2 = ci * tie(g2, x2);   // d min(x2,g2)/dx2 = [x2 < g2]
    const float dI1 = -ci * tie(x1, g1);  // -d max(x1,g1)/dx1
    const float dU2 = 1.0f - dI2, dU1 = -1.0f - dI1;
    const float dE2 = ce * tie(x2, g2);
    const float dE1 = -ce * tie(g1, x1);  // -d min(x1,g1)/dx1 = -[x1 < g1]
    // L = 1 - giou = 2 - I/U - U/E
    const float iu2 = 1.0f / (s.uni * s.uni), ie2 = 1.0f / (s.enc * s.enc);
    const float dL2 = -(dI2 * s.uni - s.inter * dU2) * iu2 - (dU2 * s.enc - s.uni * dE2) * ie2;
    const float dL1 = -(dI1 * s.uni - s.inter * dU1) * iu2 - (dU1 * s.enc - s.uni * dE1) * ie2;
    const float kg = g_giou / sumT;
    dcx += kg * (dL1 + dL2);
    dw += kg * 0.5f * (dL2 - dL1);
  }
  dspans[(int64_t)i * 2] = dcx;
  dspans[(int64_t)i * 2 + 1] = dw;
}

__global__ __launch_bounds__(256) void set_loss_bwd_kernel(
    const float* __restrict__ logits, const float* __restrict__ spans, const float* __restrict__ tgt_cxw,
    const float* __restrict__ tgt_xx, const int32_t* __restrict__ tgt_off,
    const int32_t* __restrict__ match_q, int N, int Q, float eos_coef, const float* __restrict__ g,
    float* __restrict__ dlogits, float* __restrict__ dspans, const int32_t* __restrict__ n_valid) {
  set_loss_bwd_body(logits, spans, tgt_cxw, tgt_xx, tgt_off, match_q, N, Q, eos_coef, g[0], g[1], g[2], dlogits, dspans, n_valid,
                    blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// rec_ss (criterion.py:223-274).  Stage A, one workgroup per pair: masked means over the GT clips /
// the valid (expanded) words, L2-normalised.
__device__ __forceinline__ float block_sum(float v, float* sh) {
  v = wave_sum(v);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  __syncthreads();
  if (lane == 0) sh[wave] = v;
  __syncthreads();
  float t = 0.0f;
  for (int w = 0; w < (int)(blockDim.x >> 6); ++w) t += sh[w];
  return t;
}

constexpr int SSM_G = 4;  // row groups per workgroup (independent load chains)
__device__ __forceinline__ void ss_mean_fwd_body(
    const float* __restrict__ pv, const uint8_t* __restrict__ cmask, int Lv,
    const float* __restrict__ ew, const uint8_t* __restrict__ wmask, int Le, int D,
    float* __restrict__ cn, float* __restrict__ wn, float* __restrict__ stats, int n) {
  __shared__ float sh[16];
  __shared__ float part[2][SSM_G][1024];
  const int t = threadIdx.x & 255, g = threadIdx.x >> 8;
  // valid clips / words of the pair, counted by the whole workgroup (a per-thread loop over the mask bytes was a
  // chain of Lv + Le dependent loads in front of everything else)
  // (the mask bytes also go to LDS: read inside the row loops they made every iteration wait for a global load)
  __shared__ uint8_t cmS[1024], wmS[256];
  const bool inl = Lv <= 1024 && Le <= 256;
  float ccnt = 0.0f, wcnt = 0.0f;
  for (int l = threadIdx.x; l < Lv; l += 256 * SSM_G) {
    const uint8_t b = cmask[(int64_t)n * Lv + l];
    if (inl) cmS[l] = b;
    ccnt += b ? 1.0f : 0.0f;
  }
  for (int l = threadIdx.x; l < Le; l += 256 * SSM_G) {
    const uint8_t b = wmask[(int64_t)n * Le + l];
    if (inl) wmS[l] = b;
    wcnt += b ? 1.0f : 0.0f;
  }
  ccnt = block_sum(ccnt, sh);
  wcnt = block_sum(wcnt, sh);
  // a row takes `tpr` threads (a float4 each when D % 4 == 0), so the workgroup walks G = 1024 / tpr rows side by
  // side: 16 at D = 256, i.e. 5 rounds for 75 clips, and the rounds of a thread are unrolled so that their loads are in
  // flight together (one workgroup per pair: nothing else hides their latency)
  const int vw = (D & 3) ? 1 : 4;
  const int tpr = (((D + vw - 1) / vw + 63) / 64) * 64;
  const int G = (256 * SSM_G) / tpr;
  const int rg = threadIdx.x / tpr, col = (threadIdx.x % tpr) * vw;
  float cs[4] = {0, 0, 0, 0}, ws[4] = {0, 0, 0, 0};
  if (rg < G && col < D) {
#pragma unroll 5
    for (int l = rg; l < Lv; l += G) {
      const float m = (inl ? cmS[l] : cmask[(int64_t)n * Lv + l]) ? 1.0f : 0.0f;
      const float* r = pv + ((int64_t)n * Lv + l) * D + col;
      if (vw == 4) {
        const float4 v = *reinterpret_cast<const float4*>(r);
        cs[0] += m * v.x; cs[1] += m * v.y; cs[2] += m * v.z; cs[3] += m * v.w;
      } else {
        cs[0] += m * r[0];
      }
    }
#pragma unroll 5
    for (int l = rg; l < Le; l += G) {
      const float m = (inl ? wmS[l] : wmask[(int64_t)n * Le + l]) ? 1.0f : 0.0f;
      const float* r = ew + ((int64_t)n * Le + l) * D + col;
      if (vw == 4) {
        const float4 v = *reinterpret_cast<const float4*>(r);
        ws[0] += m * v.x; ws[1] += m * v.y; ws[2] += m * v.z; ws[3] += m * v.w;
      } else {
        ws[0] += m * r[0];
      }
    }
    float* p0 = &part[0][0][0] + rg * D + col;  // [G][D] <= 4096 floats each
    float* p1 = &part[1][0][0] + rg * D + col;
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (e < vw) { p0[e] = cs[e]; p1[e] = ws[e]; }
  }
  __syncthreads();
  float c2 = 0.0f, w2 = 0.0f, cm = 0.0f, wm = 0.0f;
  const int c = threadIdx.x;  // thread = column for the rest (D <= 1024)
  if (c < D) {
    float a = 0.0f, b2 = 0.0f;
    for (int gg = 0; gg < G; ++gg) { a += (&part[0][0][0])[gg * D + c]; b2 += (&part[1][0][0])[gg * D + c]; }
    cm = a / ccnt; wm = b2 / wcnt;
    c2 = cm * cm; w2 = wm * wm;
  }
  const float cnorm = fmaxf(sqrtf(block_sum(c2, sh)), 1e-12f);
  const float wnorm = fmaxf(sqrtf(block_sum(w2, sh)), 1e-12f);
  if (c < D) {
    cn[(int64_t)n * D + c] = cm / cnorm;
    wn[(int64_t)n * D + c] = wm / wnorm;
  }
  if (threadIdx.x == 0) {
    stats[n * 4 + 0] = ccnt; stats[n * 4 + 1] = wcnt;
    stats[n * 4 + 2] = cnorm; stats[n * 4 + 3] = wnorm;
  }
}

__global__ __launch_bounds__(256 * SSM_G) void ss_mean_fwd_kernel(
    const float* __restrict__ pv, const uint8_t* __restrict__ cmask, int Lv,
    const float* __restrict__ ew, const uint8_t* __restrict__ wmask, int Le, int D,
    float* __restrict__ cn, float* __restrict__ wn, float* __restrict__ stats) {
  ss_mean_fwd_body(pv, cmask, Lv, ew, wmask, Le, D, cn, wn, stats, blockIdx.x);
}

// Stage B, one workgroup per row n: sim[n,:] = cn[n] wn^T / tau (saved) and the row's loss term
// (SupCon-style, +1e-6 inside the log); stage C sums the N row terms (deterministic).
__global__ __launch_bounds__(1024) void ss_loss_fwd_kernel(
    const float* __restrict__ cn, const float* __restrict__ wn, const uint8_t* __restrict__ pos, int N,
    int D, float inv_tau, float* __restrict__ sim, float* __restrict__ rowloss, const int32_t* __restrict__ n_valid) {
  extern __shared__ float srow[];  // N
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ld = N;  // row stride of sim / pos: the allocated extent
  if (n_valid) {     // padding pairs are neither rows nor columns of the contrast
    N = *n_valid;
    if (n >= N) {
      if (threadIdx.x == 0) rowloss[n] = 0.0f;
      return;
    }
  }
  float q[16];  // D <= 1024: this lane's slice of cn[n]
#pragma unroll
  for (int i = 0; i < 16; ++i) { const int c = lane + 64 * i; q[i] = c < D ? cn[(int64_t)n * D + c] : 0.0f; }
  // two columns per turn of a wave: both columns' loads are in flight before the first reduction starts
  const int nw = (int)(blockDim.x >> 6);  // (16 waves: one turn covers 32 columns)
  for (int k = wave; k < N; k += 2 * nw) {
    const int k2 = k + nw < N ? k + nw : k;
    float a = 0.0f, b = 0.0f;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int c = lane + 64 * i;
      if (c < D) { a += q[i] * wn[(int64_t)k * D + c]; b += q[i] * wn[(int64_t)k2 * D + c]; }
    }
    a = wave_sum(a);
    b = wave_sum(b);
    if (lane == 0) {
      srow[k] = a * inv_tau; sim[(int64_t)n * ld + k] = a * inv_tau;
      if (k + nw < N) { srow[k2] = b * inv_tau; sim[(int64_t)n * ld + k2] = b * inv_tau; }
    }
  }
  __syncthreads();
  if (wave == 0) {  // the row's log-sum-exp and positive terms across the lanes of one wave
    float m = -INFINITY;
    for (int k = lane; k < N; k += 64) m = fmaxf(m, srow[k]);
    m = wave_max(m);
    float S = 0.0f, num = 0.0f, cnt = 0.0f;
    for (int k = lane; k < N; k += 64) S += expf(srow[k] - m);
    S = wave_sum(S);
    const float logS = logf(S + 1e-6f);
    for (int k = lane; k < N; k += 64)
      if (pos[(int64_t)n * ld + k]) { num += (srow[k] - m) - logS; cnt += 1.0f; }
    num = wave_sum(num);
    cnt = wave_sum(cnt);
    if (lane == 0) rowloss[n] = -num / (cnt + 1e-6f);
  }
}

__global__ __launch_bounds__(64) void ss_reduce_kernel(const float* __restrict__ rowloss, int N,
                                                       float* __restrict__ out, const int32_t* __restrict__ n_valid) {
  if (n_valid) N = *n_valid;
  float a = 0.0f;
  for (int n = threadIdx.x; n < N; n += 64) a += rowloss[n];
  a = wave_sum(a);
  if (threadIdx.x == 0) out[0] = a / (float)N;
}

// Backward, one workgroup per pair n: rebuild dsim (N x N) in LDS from the saved sim, then
// d cn[n] = dsim[n,:] wn / tau, d wn[n] = dsim[:,n]^T cn / tau, through the L2 normalisation and the
// masked means into d projed_video_feat[n] (Lv, D) and d expanded_words_feat[n] (Le, D).
__device__ __forceinline__ void ss_bwd_body(
    const float* __restrict__ cn, const float* __restrict__ wn, const uint8_t* __restrict__ pos,
    const float* __restrict__ sim, const float* __restrict__ stats, const uint8_t* __restrict__ cmask,
    const uint8_t* __restrict__ wmask, int N, int D, int Lv, int Le, float inv_tau,
    float g0, float* __restrict__ dpv, float* __restrict__ dew, const int32_t* __restrict__ n_valid, int bid) {
  // dynamic LDS: dsim (ld rows of ld + 1 floats: lane = row accesses stay conflict-free), staged with sim and turned into
  // d sim in place | the positive mask (bytes) | the pair's clip / word masks.  Everything a thread then loops over comes
  // from LDS: the loops below used to be chains of dependent global loads (one row of sim per thread, one mask byte per
  // output row): 38 us for 2.4 MB of output.
  extern __shared__ float dsim[];
  __shared__ float sh[8];
  const int n = bid;
  const int ld = N;  // row stride of sim / pos: the allocated extent
  const int lp = ld + 1;
  float* cmL = dsim + ld * lp;  // Lv
  float* wmL = cmL + Lv;        // Le
  uint8_t* posL = reinterpret_cast<uint8_t*>(wmL + Le);  // ld * ld
  // everything this workgroup reads from memory besides wn / cn of the other pairs is requested up front, four
  // elements of sim / pos per thread at a time: taken one by one inside the loops these were a dozen round trips
  const int nv = n_valid ? *n_valid : N;
  const float ccnt = stats[n * 4 + 0], wcnt = stats[n * 4 + 1];
  const float cnorm = stats[n * 4 + 2], wnorm = stats[n * 4 + 3];
  float yc[4], yw[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = threadIdx.x + j * 256;
    yc[j] = c < D ? cn[(int64_t)n * D + c] : 0.0f;
    yw[j] = c < D ? wn[(int64_t)n * D + c] : 0.0f;
  }
  for (int base = 0; base < ld * ld; base += 1024) {
    float sv[4];
    uint8_t pb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + threadIdx.x + 256 * u;
      sv[u] = idx < ld * ld ? sim[idx] : 0.0f;
      pb[u] = idx < ld * ld ? pos[idx] : 0;
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      const int idx = base + threadIdx.x + 256 * u;
      if (idx < ld * ld) {
        const int r = idx / ld, k = idx - r * ld;
        dsim[r * lp + k] = sv[u];
        posL[idx] = pb[u];
      }
    }
  }
  for (int l = threadIdx.x; l < Lv; l += 256) cmL[l] = cmask[(int64_t)n * Lv + l] ? 1.0f : 0.0f;
  for (int l = threadIdx.x; l < Le; l += 256) wmL[l] = wmask[(int64_t)n * Le + l] ? 1.0f : 0.0f;
  __syncthreads();
  N = nv;  // padding pairs: no row, no column; their own gradients come out zero below
  const float gs = g0 / (float)N;
  // d sim row by row: 8 lanes share a row (a thread per row left one 32-iteration chain of LDS reads and expf per
  // loop to a single wave)
  const int seg = threadIdx.x & 7;
  for (int rbase = 0; rbase < N; rbase += 32) {
    const int r = rbase + (threadIdx.x >> 3);
    const bool live = r < N;
    float* row = dsim + (live ? r : 0) * lp;
    const uint8_t* prow = posL + (live ? r : 0) * ld;
    float m = -INFINITY;
    int am = 0x7fffffff;
    if (live)
      for (int k = seg; k < N; k += 8)
        if (row[k] > m) { m = row[k]; am = k; }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) {  // the first index of the maximum, as a serial scan finds it
      const float m2 = __shfl_xor(m, o);
      const int a2 = __shfl_xor(am, o);
      if (m2 > m || (m2 == m && a2 < am)) { m = m2; am = a2; }
    }
    float S = 0.0f, cnt = 0.0f;
    if (live)
      for (int k = seg; k < N; k += 8) {
        S += expf(row[k] - m);
        cnt += prow[k] ? 1.0f : 0.0f;
      }
#pragma unroll
    for (int o = 1; o < 8; o <<= 1) { S += __shfl_xor(S, o); cnt += __shfl_xor(cnt, o); }
    const float a = gs / (cnt + 1e-6f);
    const float iS = 1.0f / (S + 1e-6f);
    if (live)
      for (int k = seg; k < N; k += 8) {
        float d = -a * (prow[k] ? 1.0f : 0.0f) + a * cnt * expf(row[k] - m) * iS;
        if (k == am) d += a * cnt * 1e-6f * iS;
        row[k] = d * inv_tau;
      }
  }
  __syncthreads();
  float dc[4] = {0, 0, 0, 0}, dw[4] = {0, 0, 0, 0};
  const int kmax = n < N ? N : 0;  // a padding pair takes part in nothing
#pragma unroll 16
  for (int k = 0; k < kmax; ++k) {
    const float a = dsim[n * lp + k], b = dsim[k * lp + n];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < D) {
        dc[j] += a * wn[(int64_t)k * D + c];
        dw[j] += b * cn[(int64_t)k * D + c];
      }
    }
  }
  float pc = 0.0f, pw = 0.0f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    pc += yc[j] * dc[j];
    pw += yw[j] * dw[j];
  }
  pc = block_sum(pc, sh);
  pw = block_sum(pw, sh);
  // y = x / max(|x|, eps): dx = (dy - y (y.dy)) / |x|  (|x| > eps; F.normalize's clamp branch has
  // zero gradient through the norm, dx = dy / eps)
  const bool cok = cnorm > 1e-12f, wok = wnorm > 1e-12f;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    dc[j] = (cok ? (dc[j] - yc[j] * pc) : dc[j]) / (cnorm * ccnt);
    dw[j] = (wok ? (dw[j] - yw[j] * pw) : dw[j]) / (wnorm * wcnt);
  }
  if ((D & 3) == 0) {
    // rows out as float4 lanes, 4 rows per turn of the workgroup (a thread per column walked all Lv + Le rows alone)
    __shared__ __attribute__((aligned(16))) float dcw[2][1024];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < D) { dcw[0][c] = dc[j]; dcw[1][c] = dw[j]; }
    }
    __syncthreads();
    const int rg = threadIdx.x >> 6;
    for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
      const float4 a = *reinterpret_cast<const float4*>(&dcw[0][c]);
      const float4 b = *reinterpret_cast<const float4*>(&dcw[1][c]);
      for (int l = rg; l < Lv; l += 4) {
        const float m = cmL[l];
        *reinterpret_cast<float4*>(dpv + ((int64_t)n * Lv + l) * D + c) = make_float4(m * a.x, m * a.y, m * a.z, m * a.w);
      }
      for (int l = rg; l < Le; l += 4) {
        const float m = wmL[l];
        *reinterpret_cast<float4*>(dew + ((int64_t)n * Le + l) * D + c) = make_float4(m * b.x, m * b.y, m * b.z, m * b.w);
      }
    }
    return;
  }
  for (int l = 0; l < Lv; ++l) {
    const float m = cmL[l];
    float* r = dpv + ((int64_t)n * Lv + l) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < D) r[c] = m * dc[j];
    }
  }
  for (int l = 0; l < Le; ++l) {
    const float m = wmL[l];
    float* r = dew + ((int64_t)n * Le + l) * D;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = threadIdx.x + j * 256;
      if (c < D) r[c] = m * dw[j];
    }
  }
}

__global__ __launch_bounds__(256) void ss_bwd_kernel(
    const float* __restrict__ cn, const float* __restrict__ wn, const uint8_t* __restrict__ pos,
    const float* __restrict__ sim, const float* __restrict__ stats, const uint8_t* __restrict__ cmask,
    const uint8_t* __restrict__ wmask, int N, int D, int Lv, int Le, float inv_tau,
    const float* __restrict__ g, float* __restrict__ dpv, float* __restrict__ dew, const int32_t* __restrict__ n_valid) {
  ss_bwd_body(cn, wn, pos, sim, stats, cmask, wmask, N, D, Lv, Le, inv_tau, g[0], dpv, dew, n_valid, blockIdx.x);
}

// ------------------------------------------------------------------------------------------
// The whole criterion backward as ONE launch (round 5): its kernels are independent of each other -- each reads d total,
// its loss weight and what the forward saved, and writes its own gradient tensors -- and every one of them is a 256-thread
// kernel, so their workgroups share a grid: [set-loss layers | saliency | masked-LM NLL | rec_ss], a role per range of
// block indices.  What used to be separate launches in front of them is folded in: the scale vector d total x weight
// (each role multiplies its own) and the per-row weights of the NLL (mask / (N x words of the pair), a 32-byte count per
// workgroup).  7 launches -> 1; the launch lasts as long as its longest role (rec_ss, ~17 us).
template <int NE>
__global__ __launch_bounds__(256) void crit_bwd_kernel(const MesmCritBwd a, const int4 r0, const int4 r1) {
  extern __shared__ float dyn_[];  // (ss_bwd_body's dsim: the same dynamic segment)
  // the rec_ss workgroups are the long ones (one per pair, ~19 us): they take the first block indices so that they start
  // first, the thousands of short NLL row groups fill in behind them
  const int nss = r0.w - r0.z;
  const int bid = (int)blockIdx.x < nss ? r0.z + (int)blockIdx.x : (int)blockIdx.x - nss;
  const float gt = a.g_total[0];
  // role ranges: r0 = (set layers end, saliency end, nll end, ss end); r1.x = workgroups per set layer, r1.y = nll column groups
  if (bid < r0.x) {
    const int l = bid / r1.x, b = bid - l * r1.x;
    const float* w = a.weights + a.set_slot[l];
    set_loss_bwd_body(a.set_logits[l], a.set_spans[l], a.tgt_cxw, a.tgt_xx, a.tgt_off, a.set_match[l], a.N, a.Q, a.eos_coef,
                      gt * w[0], gt * w[1], gt * w[2], a.set_dlogits[l], a.set_dspans[l], a.n_valid, b);
  } else if (bid < r0.y) {
    saliency_bwd_body<NE>(a.s_pos, a.s_neg, a.sal_label, a.vmask, a.pos_idx, a.neg_idx, a.N, a.sal_L, a.sal_P, a.rank_coef,
                          a.margin, gt * a.weights[a.sal_slot], a.ds_pos, a.ds_neg, a.n_valid, bid - r0.x);
  } else if (bid < r0.z) {
    const int b = bid - r0.y;
    const int64_t r = b / r1.y;
    const int by = b - (int)r * r1.y;
    // row weight g x mask / (N x valid words of the row's pair) (criterion.py:293-299; recfw_rowgrad_kernel)
    __shared__ float sh_cnt;
    const int n = (int)(r / a.fw_Lw);
    int Nv = a.N;
    if (a.n_valid) Nv = *a.n_valid;
    if (threadIdx.x < 64) {
      float c = 0.0f;
      for (int w = threadIdx.x; w < a.fw_Lw; w += 64) c += a.words_mask[(int64_t)n * a.fw_Lw + w] ? 1.0f : 0.0f;
      c = wave_sum(c);
      if (threadIdx.x == 0) sh_cnt = c;
    }
    __syncthreads();
    float g = 0.0f;
    if (n < Nv && a.words_mask[r]) g = gt * a.weights[a.fw_slot] / ((float)Nv * sh_cnt);
    nll_bwd_body(a.logit, a.label, a.row_lse, g, a.dlogit, a.fw_C, a.fw_eps, r, by, r1.y);
  } else {
    ss_bwd_body(a.cn, a.wn, a.ss_pos, a.sim, a.stats, a.cmask, a.wmask, a.N, a.ss_D, a.ss_Lv, a.ss_Le, 1.0f / a.ss_tau,
                gt * a.weights[a.ss_slot], a.dpv, a.dew, a.n_valid, bid - r0.z);
  }
}

// ------------------------------------------------------------------------------------------
// rec_fw reductions (criterion.py:293-304): out[0] = mean_n( sum_w row_loss / cnt_n ),
// out[1] = sum(correct * mask) / sum(mask).  One workgroup; one wave per pair.
__device__ __forceinline__ void recfw_reduce_body(const float* __restrict__ row_loss,
                                                  const uint8_t* __restrict__ correct,
                                                  const uint8_t* __restrict__ mask, int N,
                                                  int Lw, float* __restrict__ out,
                                                  const int32_t* __restrict__ n_valid) {
  if (n_valid) N = *n_valid;
  __shared__ float sh[3][16];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = (int)(blockDim.x >> 6);  // (16 waves: two pairs each at N = 32)
  float loss = 0.0f, ok = 0.0f, tot = 0.0f;
  for (int n = wave; n < N; n += nw) {
    float s = 0.0f, c = 0.0f, k = 0.0f;
    for (int w = lane; w < Lw; w += 64) {
      const float m = mask[(int64_t)n * Lw + w] ? 1.0f : 0.0f;
      s += row_loss[(int64_t)n * Lw + w];
      c += m;
      k += m * (correct[(int64_t)n * Lw + w] ? 1.0f : 0.0f);
    }
    s = wave_sum(s); c = wave_sum(c); k = wave_sum(k);
    loss += s / c; ok += k; tot += c;
  }
  if (lane == 0) { sh[0][wave] = loss; sh[1][wave] = ok; sh[2][wave] = tot; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float a = 0, b = 0, c = 0;
    for (int w = 0; w < nw; ++w) { a += sh[0][w]; b += sh[1][w]; c += sh[2][w]; }
    out[0] = a / (float)N;
    out[1] = b / c;
  }
}

__global__ __launch_bounds__(1024) void recfw_reduce_kernel(const float* __restrict__ row_loss,
                                                           const uint8_t* __restrict__ correct,
                                                           const uint8_t* __restrict__ mask, int N,
                                                           int Lw, float* __restrict__ out,
                                                           const int32_t* __restrict__ n_valid) {
  recfw_reduce_body(row_loss, correct, mask, N, Lw, out, n_valid);
}

// ------------------------------------------------------------------------------------------
// The criterion forward in three launches instead of eight (round 5).  Its blocks are independent until the weighted sum,
// and the longest of them -- the set losses' assignment, a one-wave latency chain of ~30 us per decoder layer -- used to
// run with the rest of the chip idle.  Launch A holds every block's first stage as workgroup ranges of one grid of
// 1,024-thread workgroups: [set-loss layers | saliency | rec_ss masked means | NLL rows, a row per 256-thread quarter];
// launch B is rec_ss' similarity rows (they need every pair's mean: a grid-wide dependency, hence a launch); launch C, one
// workgroup, finishes rec_fw (row losses -> loss, accuracy), rec_ss (row terms -> loss) and the weighted total.  Same
// device functions as the separate launches: the results are bit-identical.
template <int NPT>
__global__ __launch_bounds__(1024) void crit_fwd_kernel(const MesmCritFwd a, const int4 r0, const int P) {
  const int bid = blockIdx.x;
  if (bid < r0.x) {
    set_loss_fwd_body(a.set_logits[bid], a.set_spans[bid], a.tgt_cxw, a.tgt_xx, a.tgt_off, a.N, a.Q, a.Tmax, a.w_span, a.w_giou,
                      a.w_class, a.eos_coef, P, a.set_match[bid], a.lv + a.set_slot[bid], a.n_valid);
  } else if (bid < r0.y) {
    saliency_fwd_body<4, 16>(a.s_pos, a.s_neg, a.sal_label, a.vmask, a.pos_idx, a.neg_idx, a.N, a.sal_L, a.sal_P, a.rank_coef,
                             a.margin, a.lv + a.sal_slot, a.n_valid);
  } else if (bid < r0.z) {
    ss_mean_fwd_body(a.pv, a.cmask, a.ss_Lv, a.ew, a.wmask, a.ss_Le, a.ss_D, a.cn, a.wn, a.stats, bid - r0.y);
  } else {
    __shared__ NllShared nsh[4];
    const int q = threadIdx.x >> 8;
    const int64_t R = (int64_t)a.N * a.fw_Lw;
    int64_t r = (int64_t)(bid - r0.z) * 4 + q;
    const bool live = r < R;
    if (!live) r = R - 1;
    nll_fwd_body<NPT>(a.logit, a.label, a.words_mask, a.row_loss, a.row_lse, a.correct, a.fw_C, a.fw_eps, r, live,
                      threadIdx.x & 255, nsh[q]);
  }
}

__global__ __launch_bounds__(1024) void crit_tail_kernel(const MesmCritFwd a) {
  if (a.fw_on) recfw_reduce_body(a.row_loss, a.correct, a.words_mask, a.N, a.fw_Lw, a.lv + a.fw_slot, a.n_valid);
  if (threadIdx.x < 64 && a.ss_on) {  // (ss_reduce_kernel)
    int N = a.N;
    if (a.n_valid) N = *a.n_valid;
    const float* rowloss = a.stats + (size_t)a.N * 4;
    float t = 0.0f;
    for (int n = threadIdx.x; n < N; n += 64) t += rowloss[n];
    t = wave_sum(t);
    if (threadIdx.x == 0) a.lv[a.ss_slot] = t / (float)N;
  }
  if (threadIdx.x == 0) {  // (wsum_kernel; this thread wrote the two values above itself)
    float t = 0.0f;
    for (int k = 0; k < a.n_slots; ++k)
      if (a.weights[k] != 0.0f) t += a.weights[k] * a.lv[k];
    a.total[0] = t;
  }
}

// row_grad[n,w] = g * mask / (N * cnt_n): the per-row weights mesm_nll_smooth_bwd consumes
__global__ __launch_bounds__(64) void recfw_rowgrad_kernel(const uint8_t* __restrict__ mask, int N, int Lw,
                                                          const float* __restrict__ g,
                                                          float* __restrict__ row_grad,
                                                          const int32_t* __restrict__ n_valid) {
  const int n = blockIdx.x, lane = threadIdx.x;
  if (n_valid) {
    N = *n_valid;
    if (n >= N) {  // padding pair
      for (int w = lane; w < Lw; w += 64) row_grad[(int64_t)n * Lw + w] = 0.0f;
      return;
    }
  }
  float c = 0.0f;
  for (int w = lane; w < Lw; w += 64) c += mask[(int64_t)n * Lw + w] ? 1.0f : 0.0f;
  c = wave_sum(c);
  const float k = g[0] / ((float)N * c);
  for (int w = lane; w < Lw; w += 64) row_grad[(int64_t)n * Lw + w] = mask[(int64_t)n * Lw + w] ? k : 0.0f;
}

// ------------------------------------------------------------------------------------------
// saliency score (model.py:301-302): s[n,l] = <a[n,l,:], b[n,:]> * scale.  One wave per row.
__global__ __launch_bounds__(256) void rowdot_fwd_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b, int64_t rows, int L,
                                                         int D, float scale, float* __restrict__ s) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* x = a + r * D;
  const float* y = b + (r / L) * D;
  float acc = 0.0f;
  for (int c = lane * 4; c < D; c += 256) {
    const float4 u = *reinterpret_cast<const float4*>(x + c);
    const float4 v = *reinterpret_cast<const float4*>(y + c);
    acc += u.x * v.x + u.y * v.y + u.z * v.z + u.w * v.w;
  }
  acc = wave_sum(acc);
  if (lane == 0) s[r] = acc * scale;
}

// da[n,l,:] = ds[n,l] * b[n,:] * scale;  db[n,:] = sum_l ds[n,l] * a[n,l,:] * scale.
// One workgroup per (pair, 256-column slab); thread = column.
__global__ __launch_bounds__(256) void rowdot_bwd_kernel(const float* __restrict__ a,
                                                         const float* __restrict__ b,
                                                         const float* __restrict__ ds, int L, int D,
                                                         float scale, float* __restrict__ da,
                                                         float* __restrict__ db) {
  const int n = blockIdx.x;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (c >= D) return;
  const float bv = b[(int64_t)n * D + c] * scale;
  float acc = 0.0f;
#pragma unroll 5
  for (int l = 0; l < L; ++l) {
    const float d = ds[(int64_t)n * L + l];
    const int64_t o = ((int64_t)n * L + l) * D + c;
    acc += d * a[o];
    da[o] = d * bv;
  }
  db[(int64_t)n * D + c] = acc * scale;
}

// ------------------------------------------------------------------------------------------
// post_process_text (model.py:145-152): per-word L2 normalisation (eps 1e-5), word mask =
// (sum of normalised features != 0), sentence feature = normalised mean over valid words.
// One workgroup per pair; a wave per word.
constexpr int TP_THREADS = 1024;
__global__ __launch_bounds__(TP_THREADS) void text_prep_kernel(const float* __restrict__ x, int Lw, int D,
                                                        int normalize, float* __restrict__ words,
                                                        uint8_t* __restrict__ wmask,
                                                        float* __restrict__ sent) {
  extern __shared__ float sm[];  // D accumulators + Lw flags
  float* accv = sm;
  float* flags = sm + D;
  __shared__ float sh[16];
  const int n = blockIdx.x;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int c = threadIdx.x; c < D; c += TP_THREADS) accv[c] = 0.0f;
  __syncthreads();
  for (int w = wave; w < Lw; w += TP_THREADS / 64) {
    const float* r = x + ((int64_t)n * Lw + w) * D;
    float* o = words + ((int64_t)n * Lw + w) * D;
    float s2 = 0.0f, s = 0.0f;
    if (D <= 512) {  // the row in registers, its 8 loads in flight together (same summation order as the loops below)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = lane + 64 * u < D ? r[lane + 64 * u] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) s2 += v[u] * v[u];
      s2 = wave_sum(s2);
      const float inv = normalize ? 1.0f / fmaxf(sqrtf(s2), 1e-5f) : 1.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (lane + 64 * u < D) { const float t = v[u] * inv; o[lane + 64 * u] = t; s += t; }
    } else {
      for (int c = lane; c < D; c += 64) { const float v = r[c]; s2 += v * v; }
      s2 = wave_sum(s2);
      const float inv = normalize ? 1.0f / fmaxf(sqrtf(s2), 1e-5f) : 1.0f;
      for (int c = lane; c < D; c += 64) { const float v = r[c] * inv; o[c] = v; s += v; }
    }
    s = wave_sum(s);
    if (lane == 0) {
      flags[w] = s != 0.0f ? 1.0f : 0.0f;
      wmask[(int64_t)n * Lw + w] = s != 0.0f ? 1 : 0;
    }
  }
  __threadfence_block();
  __syncthreads();
  float cnt = 0.0f;
  for (int w = 0; w < Lw; ++w) cnt += flags[w];
  float part = 0.0f;
  // the reference sums ALL words (pads are zero vectors) and divides by the number of valid ones
  for (int c = threadIdx.x; c < D; c += TP_THREADS) {
    float a = 0.0f;
    const float* wc = words + (int64_t)n * Lw * D + c;
    int w = 0;
    for (; w + 8 <= Lw; w += 8) {  // 8 loads in flight, added in word order
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = wc[(int64_t)(w + u) * D];
#pragma unroll
      for (int u = 0; u < 8; ++u) a += t[u];
    }
    for (; w < Lw; ++w) a += wc[(int64_t)w * D];
    a /= cnt;
    accv[c] = a;
    part += a * a;
  }
  const float nrm = sqrtf(block_sum(part, sh));
  const float inv = normalize ? 1.0f / fmaxf(nrm, 1e-5f) : 1.0f;
  for (int c = threadIdx.x; c < D; c += TP_THREADS) sent[(int64_t)n * D + c] = accv[c] * inv;
}

// ------------------------------------------------------------------------------------------
// total = sum_k w[k] * vals[k];  and its backward gv[k] = g * w[k]  (criterion.py:361-365)
__global__ void wsum_kernel(const float* __restrict__ vals, const float* __restrict__ w, int n,
                            float* __restrict__ out) {
  float t = 0.0f;
  for (int k = 0; k < n; ++k)
    if (w[k] != 0.0f) t += w[k] * vals[k];
  out[0] = t;
}
__global__ void scale_vec_kernel(const float* __restrict__ g, const float* __restrict__ w, int n,
                                 float* __restrict__ out) {
  const int k = threadIdx.x;
  if (k < n) out[k] = g[0] * w[k];
}

}  // namespace

extern "C" int mesm_set_loss_fwd_nv(const float* logits, const float* spans, const float* tgt_cxw,
                                    const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
                                    int32_t Tmax, float w_span, float w_giou, float w_class,
                                    float eos_coef, int32_t* match_q, float* out4, const int32_t* n_valid, void* stream) {
  if (!logits || !spans || !tgt_cxw || !tgt_xx || !tgt_off || !match_q || !out4) return MESM_EINVAL;
  if (N <= 0 || Q <= 0 || Q > 64 || Tmax <= 0 || Tmax > 64) return MESM_EINVAL;
  const size_t per = (size_t)Tmax * Q * 8;
  const int P = sapw_waves_per_pass(Tmax, Q, N);
  if (P == 0) return MESM_EINVAL;
  hipLaunchKernelGGL(set_loss_fwd_kernel, dim3(1), dim3(SAPW_THREADS), per * P, (hipStream_t)stream, logits, spans,
                     tgt_cxw, tgt_xx, tgt_off, N, Q, Tmax, w_span, w_giou, w_class, eos_coef, P,
                     match_q, out4, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_match(const float* logits, const float* spans, const float* tgt_cxw,
                          const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
                          int32_t Tmax, float w_span, float w_giou, float w_class, float* cost,
                          int32_t* match_q, void* stream) {
  if (!logits || !spans || !tgt_cxw || !tgt_xx || !tgt_off || !match_q) return MESM_EINVAL;
  if (N <= 0 || Q <= 0 || Q > 64 || Tmax <= 0 || Tmax > 64) return MESM_EINVAL;
  const int P = sap_pairs_per_pass(Tmax, Q, N);
  if (P == 0) return MESM_EINVAL;
  hipLaunchKernelGGL(match_kernel, dim3((N + P - 1) / P), dim3(P), sap_bytes_per_thread(Tmax, Q) * P,
                     (hipStream_t)stream, logits, spans, tgt_cxw, tgt_xx, tgt_off, N, Q, Tmax, w_span, w_giou,
                     w_class, cost, match_q);
  return mesm_launch_status();
}

extern "C" int mesm_set_loss_fwd_layers(const float* const* logits, const float* const* spans, int32_t n_layers,
                                        const float* tgt_cxw, const float* tgt_xx, const int32_t* tgt_off, int32_t N,
                                        int32_t Q, int32_t Tmax, float w_span, float w_giou, float w_class, float eos_coef,
                                        int32_t* const* match_q, float* const* out4, const int32_t* n_valid, void* stream) {
  if (!logits || !spans || !tgt_cxw || !tgt_xx || !tgt_off || !match_q || !out4) return MESM_EINVAL;
  if (n_layers <= 0 || n_layers > SET_LAYERS_MAX) return MESM_EINVAL;
  if (N <= 0 || Q <= 0 || Q > 64 || Tmax <= 0 || Tmax > 64) return MESM_EINVAL;
  SetLossLayers L = {};
  for (int k = 0; k < n_layers; ++k) {
    if (!logits[k] || !spans[k] || !match_q[k] || !out4[k]) return MESM_EINVAL;
    L.logits[k] = logits[k]; L.spans[k] = spans[k]; L.match_q[k] = match_q[k]; L.out[k] = out4[k];
  }
  const size_t per = (size_t)Tmax * Q * 8;
  const int P = sapw_waves_per_pass(Tmax, Q, N);
  if (P == 0) return MESM_EINVAL;
  hipLaunchKernelGGL(set_loss_fwd_layers_kernel, dim3(n_layers), dim3(SAPW_THREADS), per * P, (hipStream_t)stream, L, tgt_cxw, tgt_xx,
                     tgt_off, N, Q, Tmax, w_span, w_giou, w_class, eos_coef, P, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_set_loss_fwd(const float* logits, const float* spans, const float* tgt_cxw,
                                 const float* tgt_xx, const int32_t* tgt_off, int32_t N, int32_t Q,
                                 int32_t Tmax, float w_span, float w_giou, float w_class,
                                 float eos_coef, int32_t* match_q, float* out4, void* stream) {
  return mesm_set_loss_fwd_nv(logits, spans, tgt_cxw, tgt_xx, tgt_off, N, Q, Tmax, w_span, w_giou, w_class, eos_coef,
                              match_q, out4, nullptr, stream);
}

extern "C" int mesm_set_loss_bwd_nv(const float* logits, const float* spans, const float* tgt_cxw,
                                    const float* tgt_xx, const int32_t* tgt_off, const int32_t* match_q,
                                    int32_t N, int32_t Q, float eos_coef, const float* g4, float* dlogits,
                                    float* dspans, const int32_t* n_valid, void* stream) {
  if (!logits || !spans || !tgt_cxw || !tgt_xx || !tgt_off || !match_q || !g4 || !dlogits || !dspans)
    return MESM_EINVAL;
  if (N <= 0 || Q <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(set_loss_bwd_kernel, dim3((N * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                     logits, spans, tgt_cxw, tgt_xx, tgt_off, match_q, N, Q, eos_coef, g4, dlogits,
                     dspans, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_set_loss_bwd(const float* logits, const float* spans, const float* tgt_cxw,
                                 const float* tgt_xx, const int32_t* tgt_off, const int32_t* match_q,
                                 int32_t N, int32_t Q, float eos_coef, const float* g4, float* dlogits,
                                 float* dspans, void* stream) {
  return mesm_set_loss_bwd_nv(logits, spans, tgt_cxw, tgt_xx, tgt_off, match_q, N, Q, eos_coef, g4, dlogits, dspans,
                              nullptr, stream);
}

extern "C" int mesm_rec_ss_fwd_nv(const float* pv, const uint8_t* cmask, int32_t Lv, const float* ew,
                                  const uint8_t* wmask, int32_t Le, const uint8_t* pos, int32_t N,
                                  int32_t D, float tau, float* cn, float* wn, float* stats, float* sim,
                                  float* out, const int32_t* n_valid, void* stream) {
  if (!pv || !cmask || !ew || !wmask || !pos || !cn || !wn || !stats || !sim || !out) return MESM_EINVAL;
  if (N <= 0 || D <= 0 || D > 1024 || Lv <= 0 || Le <= 0 || tau <= 0.f) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  hipLaunchKernelGGL(ss_mean_fwd_kernel, dim3(N), dim3(256 * SSM_G), 0, s, pv, cmask, Lv, ew, wmask, Le,
                     D, cn, wn, stats);
  // stats (N, 4) has one spare use: the row losses are staged in column 0 of a second block
  float* rowloss = stats + (size_t)N * 4;
  hipLaunchKernelGGL(ss_loss_fwd_kernel, dim3(N), dim3(1024), (size_t)N * 4, s, cn, wn, pos, N, D,
                     1.0f / tau, sim, rowloss, n_valid);
  hipLaunchKernelGGL(ss_reduce_kernel, dim3(1), dim3(64), 0, s, rowloss, N, out, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_rec_ss_fwd(const float* pv, const uint8_t* cmask, int32_t Lv, const float* ew,
                               const uint8_t* wmask, int32_t Le, const uint8_t* pos, int32_t N,
                               int32_t D, float tau, float* cn, float* wn, float* stats, float* sim,
                               float* out, void* stream) {
  return mesm_rec_ss_fwd_nv(pv, cmask, Lv, ew, wmask, Le, pos, N, D, tau, cn, wn, stats, sim, out, nullptr, stream);
}

extern "C" int mesm_rec_ss_bwd_nv(const float* cn, const float* wn, const uint8_t* pos, const float* sim,
                                  const float* stats, const uint8_t* cmask, const uint8_t* wmask,
                                  int32_t N, int32_t D, int32_t Lv, int32_t Le, float tau, const float* g,
                                  float* dpv, float* dew, const int32_t* n_valid, void* stream) {
  if (!cn || !wn || !pos || !sim || !stats || !cmask || !wmask || !g || !dpv || !dew) return MESM_EINVAL;
  const size_t lds = ((size_t)N * (N + 1) + Lv + Le) * 4 + (size_t)N * N;
  if (N <= 0 || lds > 64 * 1024 || D <= 0 || D > 1024 || tau <= 0.f) return MESM_EINVAL;  // (N <= ~100 pairs)
  hipLaunchKernelGGL(ss_bwd_kernel, dim3(N), dim3(256), lds, (hipStream_t)stream, cn, wn,
                     pos, sim, stats, cmask, wmask, N, D, Lv, Le, 1.0f / tau, g, dpv, dew, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_rec_ss_bwd(const float* cn, const float* wn, const uint8_t* pos, const float* sim,
                               const float* stats, const uint8_t* cmask, const uint8_t* wmask,
                               int32_t N, int32_t D, int32_t Lv, int32_t Le, float tau, const float* g,
                               float* dpv, float* dew, void* stream) {
  return mesm_rec_ss_bwd_nv(cn, wn, pos, sim, stats, cmask, wmask, N, D, Lv, Le, tau, g, dpv, dew, nullptr, stream);
}

extern "C" int mesm_criterion_fwd(const MesmCritFwd* args, void* stream) {
  if (!args) return MESM_EINVAL;
  const MesmCritFwd& a = *args;
  hipStream_t s = (hipStream_t)stream;
  if (!a.lv || !a.weights || !a.total || a.N <= 0 || a.n_slots <= 0 || a.n_slots > 256) return MESM_EINVAL;
  if (a.n_set < 0 || a.n_set > SET_LAYERS_MAX) return MESM_EINVAL;
  int P = 0;
  size_t lds = 0;
  if (a.n_set > 0) {
    if (!a.tgt_cxw || !a.tgt_xx || !a.tgt_off || a.Q <= 0 || a.Q > 64 || a.Tmax <= 0 || a.Tmax > 64) return MESM_EINVAL;
    for (int l = 0; l < a.n_set; ++l)
      if (!a.set_logits[l] || !a.set_spans[l] || !a.set_match[l] || a.set_slot[l] < 0 || a.set_slot[l] + 4 > a.n_slots)
        return MESM_EINVAL;
    P = sapw_waves_per_pass(a.Tmax, a.Q, a.N);
    if (P == 0) return MESM_EINVAL;
    lds = (size_t)a.Tmax * a.Q * 8 * P;
  }
  bool sal_inside = false;
  if (a.sal_on) {
    if (!a.s_pos || !a.s_neg || !a.sal_label || !a.vmask || a.sal_L <= 0 || a.sal_slot < 0 || a.sal_slot >= a.n_slots ||
        (a.sal_P > 0 && (!a.pos_idx || !a.neg_idx)))
      return MESM_EINVAL;
    sal_inside = 2 * a.sal_L <= 256;  // (the 20-element rows need more registers than a 1,024-thread workgroup has)
    if (!sal_inside) {
      const int rc = mesm_saliency_loss_fwd_nv(a.s_pos, a.s_neg, a.sal_label, a.vmask, a.pos_idx, a.neg_idx, a.N, a.sal_L, a.sal_P,
                                               a.rank_coef, a.margin, a.lv + a.sal_slot, a.n_valid, stream);
      if (rc != MESM_OK) return rc;
    }
  }
  if (a.fw_on) {
    if (!a.logit || !a.label || !a.words_mask || !a.row_loss || !a.row_lse || !a.correct || a.fw_Lw <= 0 || a.fw_C <= 0 ||
        a.fw_slot < 0 || a.fw_slot + 2 > a.n_slots)
      return MESM_EINVAL;
  }
  if (a.ss_on) {
    if (!a.pv || !a.cmask || !a.ew || !a.wmask || !a.ss_pos || !a.cn || !a.wn || !a.stats || !a.sim) return MESM_EINVAL;
    if (a.ss_D <= 0 || a.ss_D > 1024 || a.ss_Lv <= 0 || a.ss_Le <= 0 || a.ss_tau <= 0.f || a.ss_slot < 0 || a.ss_slot >= a.n_slots)
      return MESM_EINVAL;
  }
  int4 r0;
  r0.x = a.n_set;
  r0.y = r0.x + (sal_inside ? 1 : 0);
  r0.z = r0.y + (a.ss_on ? a.N : 0);
  const int64_t nll_blocks = a.fw_on ? ((int64_t)a.N * a.fw_Lw + 3) / 4 : 0;
  if (nll_blocks > (1 << 30)) return MESM_EINVAL;
  r0.w = r0.z + (int)nll_blocks;
  if (r0.w > 0) {
    if (!a.fw_on || a.fw_C <= 2048)
      hipLaunchKernelGGL(crit_fwd_kernel<8>, dim3(r0.w), dim3(1024), lds, s, a, r0, P);
    else if (a.fw_C <= 5120)
      hipLaunchKernelGGL(crit_fwd_kernel<20>, dim3(r0.w), dim3(1024), lds, s, a, r0, P);
    else
      hipLaunchKernelGGL(crit_fwd_kernel<0>, dim3(r0.w), dim3(1024), lds, s, a, r0, P);
  }
  if (a.ss_on)
    hipLaunchKernelGGL(ss_loss_fwd_kernel, dim3(a.N), dim3(1024), (size_t)a.N * 4, s, a.cn, a.wn, a.ss_pos, a.N, a.ss_D,
                       1.0f / a.ss_tau, a.sim, a.stats + (size_t)a.N * 4, a.n_valid);
  hipLaunchKernelGGL(crit_tail_kernel, dim3(1), dim3(1024), 0, s, a);
  return mesm_launch_status();
}

extern "C" int mesm_criterion_bwd(const MesmCritBwd* args, void* stream) {
  if (!args) return MESM_EINVAL;
  const MesmCritBwd& a = *args;
  if (!a.g_total || !a.weights || a.N <= 0) return MESM_EINVAL;
  if (a.n_set < 0 || a.n_set > 8) return MESM_EINVAL;
  int per_layer = 0;
  if (a.n_set > 0) {
    if (a.Q <= 0 || !a.tgt_cxw || !a.tgt_xx || !a.tgt_off) return MESM_EINVAL;
    for (int l = 0; l < a.n_set; ++l)
      if (!a.set_logits[l] || !a.set_spans[l] || !a.set_match[l] || !a.set_dlogits[l] || !a.set_dspans[l]) return MESM_EINVAL;
    per_layer = (a.N * a.Q + 255) / 256;
  }
  int4 r0, r1;
  r0.x = a.n_set * per_layer;
  r1.x = per_layer > 0 ? per_layer : 1;
  int sal_blocks = 0;
  if (a.sal_on) {
    if (!a.s_pos || !a.s_neg || !a.sal_label || !a.vmask || !a.ds_pos || !a.ds_neg || a.sal_L <= 0 || 2 * a.sal_L > 64 * SAL_MAXE ||
        (a.sal_P > 0 && (!a.pos_idx || !a.neg_idx)))
      return MESM_EINVAL;
    sal_blocks = (a.N + 3) / 4;
  }
  r0.y = r0.x + sal_blocks;
  int gy = 1;
  int64_t nll_blocks = 0;
  if (a.fw_on) {
    if (!a.logit || !a.label || !a.row_lse || !a.words_mask || !a.dlogit || a.fw_Lw <= 0 || a.fw_C <= 0) return MESM_EINVAL;
    gy = (a.fw_C + 1023) / 1024;
    nll_blocks = (int64_t)a.N * a.fw_Lw * gy;
    if (nll_blocks > (1 << 30)) return MESM_EINVAL;
  }
  r1.y = gy;
  r0.z = r0.y + (int)nll_blocks;
  size_t lds = 0;
  if (a.ss_on) {
    if (!a.cn || !a.wn || !a.ss_pos || !a.sim || !a.stats || !a.cmask || !a.wmask || !a.dpv || !a.dew) return MESM_EINVAL;
    lds = ((size_t)a.N * (a.N + 1) + a.ss_Lv + a.ss_Le) * 4 + (size_t)a.N * a.N;
    if (lds > 64 * 1024 || a.ss_D <= 0 || a.ss_D > 1024 || a.ss_tau <= 0.f) return MESM_EINVAL;
  }
  r0.w = r0.z + (a.ss_on ? a.N : 0);
  r1.z = r1.w = 0;
  if (r0.w <= 0) return MESM_OK;
  if (a.sal_on && 2 * a.sal_L > 256)
    hipLaunchKernelGGL(crit_bwd_kernel<SAL_MAXE>, dim3(r0.w), dim3(256), lds, (hipStream_t)stream, a, r0, r1);
  else
    hipLaunchKernelGGL(crit_bwd_kernel<4>, dim3(r0.w), dim3(256), lds, (hipStream_t)stream, a, r0, r1);
  return mesm_launch_status();
}

extern "C" int mesm_rec_fw_reduce_nv(const float* row_loss, const uint8_t* correct, const uint8_t* mask,
                                     int32_t N, int32_t Lw, float* out2, const int32_t* n_valid, void* stream) {
  if (!row_loss || !correct || !mask || !out2 || N <= 0 || Lw <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(recfw_reduce_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, row_loss, correct,
                     mask, N, Lw, out2, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_rec_fw_reduce(const float* row_loss, const uint8_t* correct, const uint8_t* mask,
                                  int32_t N, int32_t Lw, float* out2, void* stream) {
  return mesm_rec_fw_reduce_nv(row_loss, correct, mask, N, Lw, out2, nullptr, stream);
}

extern "C" int mesm_rec_fw_rowgrad_nv(const uint8_t* mask, int32_t N, int32_t Lw, const float* g,
                                      float* row_grad, const int32_t* n_valid, void* stream) {
  if (!mask || !g || !row_grad || N <= 0 || Lw <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(recfw_rowgrad_kernel, dim3(N), dim3(64), 0, (hipStream_t)stream, mask, N, Lw, g,
                     row_grad, n_valid);
  return mesm_launch_status();
}

extern "C" int mesm_rec_fw_rowgrad(const uint8_t* mask, int32_t N, int32_t Lw, const float* g,
                                   float* row_grad, void* stream) {
  return mesm_rec_fw_rowgrad_nv(mask, N, Lw, g, row_grad, nullptr, stream);
}

extern "C" int mesm_rowdot_fwd(const float* a, const float* b, int32_t N, int32_t L, int32_t D,
                               float scale, float* s, void* stream) {
  if (!a || !b || !s || N <= 0 || L <= 0 || D <= 0 || (D % 4) != 0) return MESM_EINVAL;
  const int64_t rows = (int64_t)N * L;
  hipLaunchKernelGGL(rowdot_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, a, b, rows, L, D, scale, s);
  return mesm_launch_status();
}

extern "C" int mesm_rowdot_bwd(const float* a, const float* b, const float* ds, int32_t N, int32_t L,
                               int32_t D, float scale, float* da, float* db, void* stream) {
  if (!a || !b || !ds || !da || !db || N <= 0 || L <= 0 || D <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(rowdot_bwd_kernel, dim3(N, (D + 255) / 256), dim3(256), 0, (hipStream_t)stream, a,
                     b, ds, L, D, scale, da, db);
  return mesm_launch_status();
}

extern "C" int mesm_text_prep(const float* x, int32_t N, int32_t Lw, int32_t D, int32_t normalize,
                              float* words, uint8_t* wmask, float* sent, void* stream) {
  if (!x || !words || !wmask || !sent || N <= 0 || Lw <= 0 || D <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(text_prep_kernel, dim3(N), dim3(TP_THREADS), (size_t)(D + Lw) * 4, (hipStream_t)stream, x,
                     Lw, D, normalize, words, wmask, sent);
  return mesm_launch_status();
}

extern "C" int mesm_weighted_sum(const float* vals, const float* weights, int32_t n, float* out,
                                 void* stream) {
  if (!vals || !weights || !out || n <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(wsum_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, vals, weights, n, out);
  return mesm_launch_status();
}

extern "C" int mesm_scale_vec(const float* g, const float* weights, int32_t n, float* out, void* stream) {
  if (!g || !weights || !out || n <= 0 || n > 256) return MESM_EINVAL;
  hipLaunchKernelGGL(scale_vec_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, g, weights, n, out);
  return mesm_launch_status();
}
