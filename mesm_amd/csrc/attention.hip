// Multi-head attention core for the MESM shapes (B*H ~ 256 heads, Lq,Lk <= ~700,
// head dims 8..64), fp32 VALU math (the north star reserves MFMA for the dense
// projections).  One workgroup (4 waves) per (batch, head[, query chunk / key tile]).
//
// Forward: keys are processed in tiles of 64 (one key per lane).  K/V tiles are staged
// coalesced into LDS; each wave walks its query rows: lane j computes the score of key
// j against the row (q broadcast from LDS), row max / row sum by wave shuffles, online
// softmax across key tiles with the running (m, l, o) row state kept in LDS, P V with
// lanes re-mapped to (feature d, key phase g) and the probabilities exchanged through a
// per-wave LDS row.
//
// Backward: one workgroup per (batch, head, 64-key tile); lane j owns key j: K_j, V_j
// and the dK_j, dV_j accumulators live in registers while the workgroup sweeps all
// query rows (P is recomputed from the saved log-sum-exp).  dQ rows are reduced over
// lanes through LDS and written (single key tile) or added atomically (several tiles).
#include "common.hpp"
#include "attention_blk.hpp"
#include "attention_mfma.hpp"
#include <cstdlib>
#include <cstddef>

namespace {

// Both kernels are latency-bound per wave (every query row is a chain of LDS reads, ~17 cross-lane
// exchanges and a P V / dQ sweep, ~1.5-2k cycles), so the lever is rows per wave: the forward
// splits the queries into chunks of 16 (4 rows per wave; K/V tiles are L2-resident, re-staging
// them per chunk is cheap), the backward sweeps all queries with 8 waves per workgroup.
#ifndef MESM_AT_WAVES
#define MESM_AT_WAVES 4
#endif
constexpr int AT_WAVES = MESM_AT_WAVES;  // forward workgroup: AT_WAVES waves x 4 query rows each
constexpr int AT_THREADS = 64 * AT_WAVES;
constexpr int KT = 64;   // keys per tile (one per lane)
constexpr int QCH = 4 * AT_WAVES;  // query rows per forward workgroup (4 per wave: the P V sweep handles 4 rows at once)
static_assert(QCH / AT_WAVES == 4, "the batched sweeps read one float4 of row slots per key");

struct MaskCtx {
  bool kp;   // kpad[b, j]
  bool kp2;  // kpad[b2, j] (quirk mode)
};

template <int DK, int DV>
__global__ __launch_bounds__(AT_THREADS) void attn_fwd_kernel(const MesmAttnArgs p) {
  constexpr int SK = DK + 4;  // padded K row (conflict-free ds_read_b128 per lane)
  constexpr int G = 64 / DV;  // key phases in the P V step
  __shared__ __attribute__((aligned(16))) float Ks[KT * SK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * DV];
  __shared__ __attribute__((aligned(16))) float Qs[QCH * DK];
  __shared__ __attribute__((aligned(16))) float Os[QCH * DV];
  __shared__ float Ms[QCH], Ls[QCH];
  __shared__ __attribute__((aligned(16))) float Ps[AT_WAVES * KT * (QCH / AT_WAVES)];  // [wave][key][row slot]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int bh = blockIdx.x;
  const int b = bh / p.H, h = bh % p.H;
  const int q0 = blockIdx.y * QCH;
  const int nq = (p.Lq - q0) < QCH ? (p.Lq - q0) : QCH;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const bool causal = p.mask_mode == MESM_MASK_CAUSAL;

  // split heads (q2 / k2 given): the head's DK features are [ q[h*DK/2 ...] || q2[h*DK/2 ...] ], two tensors of
  // H * DK/2 columns with the same strides -- the decoder's per-head [content || position] queries and keys
  // (transformer.py:778-784) read in place instead of from an interleaved copy
  constexpr int DKH = DK / 2;
  const bool split = p.q2 != nullptr;
  const float* qb = p.q + (int64_t)b * p.q_bs + (int64_t)h * (split ? DKH : DK);
  const float* kb = p.k + (int64_t)b * p.k_bs + (int64_t)h * (split ? DKH : DK);
  const float* qb2 = split ? p.q2 + (int64_t)b * p.q_bs + (int64_t)h * DKH - DKH : qb;  // indexed with c >= DKH
  const float* kb2 = split ? p.k2 + (int64_t)b * p.k_bs + (int64_t)h * DKH - DKH : kb;
  const float* kadd = (split && p.k_add) ? p.k_add + (int64_t)b * p.k_bs + (int64_t)h * DKH : nullptr;
  const float* vb = p.v + (int64_t)b * p.v_bs + (int64_t)h * DV;

  // stage the query chunk and reset the row state
  for (int idx = tid; idx < QCH * (DK / 4); idx += AT_THREADS) {
    int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < nq) t = *reinterpret_cast<const float4*>((c >= DKH ? qb2 : qb) + (int64_t)(q0 + r) * p.q_ls + c);
    *reinterpret_cast<float4*>(Qs + r * DK + c) = t;
  }
  for (int idx = tid; idx < QCH * DV; idx += AT_THREADS) Os[idx] = 0.0f;
  if (tid < QCH) { Ms[tid] = -INFINITY; Ls[tid] = 0.0f; }

  const uint32_t thresh = p.drop_p > 0.f ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);

  const int ntiles = (p.Lk + KT - 1) / KT;
  for (int t = 0; t < ntiles; ++t) {
    const int k0 = t * KT;
    __syncthreads();  // previous tile fully consumed (and Qs/Os initialised)
    for (int idx = tid; idx < KT * (DK / 4); idx += AT_THREADS) {
      int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + r < p.Lk) {
        x = *reinterpret_cast<const float4*>((c >= DKH ? kb2 : kb) + (int64_t)(k0 + r) * p.k_ls + c);
        if (kadd && c < DKH) {  // content half = kcontent + kpos (decoder layer 0, transformer.py:773-776)
          const float4 y = *reinterpret_cast<const float4*>(kadd + (int64_t)(k0 + r) * p.k_ls + c);
          x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
        }
      }
      *reinterpret_cast<float4*>(Ks + r * SK + c) = x;
    }
    for (int idx = tid; idx < KT * (DV / 4); idx += AT_THREADS) {
      int r = idx / (DV / 4), c = (idx % (DV / 4)) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (k0 + r < p.Lk) x = *reinterpret_cast<const float4*>(vb + (int64_t)(k0 + r) * p.v_ls + c);
      *reinterpret_cast<float4*>(Vs + r * DV + c) = x;
    }
    const int j = k0 + lane;
    const bool jvalid = j < p.Lk;
    bool kp = !jvalid, kp2 = false;
    if (jvalid && p.kpad) kp = p.kpad[(int64_t)b * p.Lk + j] != 0;
    if (jvalid && quirk) kp2 = p.kpad[(int64_t)b2 * p.Lk + j] != 0;
    __syncthreads();

    float kreg[DK];
#pragma unroll
    for (int c = 0; c < DK; c += 4) {
      float4 x = *reinterpret_cast<const float4*>(Ks + lane * SK + c);
      kreg[c] = x.x; kreg[c + 1] = x.y; kreg[c + 2] = x.z; kreg[c + 3] = x.w;
    }

    // A wave owns rows wave, wave + 4, wave + 8, wave + 12 of the chunk.  Phase 1, per row: scores,
    // online-softmax state, (dropped) probabilities into Ps[key][row slot].  Phase 2, ONE sweep over the
    // keys for all four rows: a V element is read once and multiplied into four accumulators (the
    // per-row sweep was half of the row's instructions; the kernel is VALU-issue bound).
    constexpr int RW = QCH / AT_WAVES;
    float alpha4[RW];
#pragma unroll
    for (int rr = 0; rr < RW; ++rr) {
      const int r = wave + rr * AT_WAVES;
      float pd = 0.0f;
      alpha4[rr] = 0.0f;
      if (r < nq) {
        const int i = q0 + r;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < DK; c += 4) {
          float4 qv = *reinterpret_cast<const float4*>(Qs + r * DK + c);
          s += qv.x * kreg[c] + qv.y * kreg[c + 1] + qv.z * kreg[c + 2] + qv.w * kreg[c + 3];
        }
        s *= p.scale;
        bool masked = kp;
        if (quirk) {
          bool qp = p.qpad[(int64_t)b2 * p.Lq + i] != 0;
          masked = masked || (qp && kp2);
        }
        if (causal && j > i) masked = true;
        if (masked) s = -INFINITY;
        const float m_old = Ms[r];
        const float m_new = fmaxf(m_old, wave_max(s));
        const float pj = (m_new == -INFINITY) ? 0.0f : __expf(s - m_new);
        const float alpha = (m_old == -INFINITY) ? 0.0f : __expf(m_old - m_new);
        const float l_new = Ls[r] * alpha + wave_sum(pj);
        pd = pj;
        // the fp16 SDPA of the frozen CLIP tower hands the probabilities to P V in fp16 while the row sum keeps
        // the fp32 values (ATen's CPU flash kernel for reduced types): same rounding point here
        if (causal) pd = (float)(_Float16)pj;
        if (thresh) {
          uint32_t idx = (uint32_t)(((int64_t)bh * p.Lq + i) * p.Lk + j);
          pd = mesm_dropout_apply(pj, idx, drop_seed, thresh, inv_keep);
        }
        alpha4[rr] = alpha;
        if (lane == 0) { Ms[r] = m_new; Ls[r] = l_new; }
      }
      Ps[(wave * KT + lane) * RW + rr] = pd;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const int d = lane % DV, g = lane / DV;
      float acc[RW];
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) acc[rr] = 0.0f;
      const int kend = (p.Lk - k0) < KT ? (p.Lk - k0) : KT;  // keys beyond Lk carry p = 0: skip them (Lk = 33: 17 steps, not 32)
#pragma unroll 4
      for (int jj = g; jj < kend; jj += G) {
        const float4 pr = *reinterpret_cast<const float4*>(Ps + (wave * KT + jj) * RW);
        const float v = Vs[jj * DV + d];
        acc[0] += pr.x * v; acc[1] += pr.y * v; acc[2] += pr.z * v; acc[3] += pr.w * v;
      }
#pragma unroll
      for (int rr = 0; rr < RW; ++rr) {
        const int r = wave + rr * AT_WAVES;
        const float t = sum_across_groups<DV>(acc[rr]);
        if (g == 0 && r < nq) Os[r * DV + d] = Os[r * DV + d] * alpha4[rr] + t;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // finalise this wave's rows
  float* ob = p.o + (int64_t)b * p.o_bs + (int64_t)h * DV;
  for (int r = wave; r < nq; r += AT_WAVES) {
    const int i = q0 + r;
    const float l = Ls[r];
    if (lane < DV) mesm_store_wt(ob + ((int64_t)i * p.o_ls + lane), Os[r * DV + lane] / l);  // (write-through: common.hpp)
    if (lane == 0 && p.lse) p.lse[(int64_t)bh * p.Lq + i] = Ms[r] + __logf(l);
  }
}

// BW_WAVES waves per workgroup, 4 query rows per wave per staged chunk.  Measured (tools/attn_bench.py, 64 x 8
// heads): 8 waves 42 / 79 / 57 us for (Lq 75, Lk 33) / (76, 76) / (33, 75); 4 waves 32 / 58 / 36; 2 waves
// 50 / 53 / 29 -- 4 waves when there is one key tile, 2 when there are several (twice the workgroups).
template <int DK, int DV, int BW_WAVES>
__device__ __forceinline__ void attn_bwd_body(const MesmAttnArgs& p, const int bh, const int ktile, const int ntiles) {
  constexpr int BW_THREADS = 64 * BW_WAVES;
  constexpr int QCB = 4 * BW_WAVES;  // query rows staged per chunk
  constexpr int SK = DK + 4;
  constexpr int SV = DV + 4;
  constexpr int GQ = 64 / DK >= 1 ? 64 / DK : 1;  // key phases in the dQ step (DK <= 64)
  constexpr int DR = (DK > DV ? DK : DV) + 1;  // +1: lane-per-row accesses stay conflict-free
  __shared__ __attribute__((aligned(16))) float Ks[KT * SK];
  __shared__ __attribute__((aligned(16))) float Vs[KT * SV];
  __shared__ __attribute__((aligned(16))) float Qs[QCB * DK];
  __shared__ __attribute__((aligned(16))) float dOs[QCB * DV];
  __shared__ float Dl[QCB], Lse[QCB];
  __shared__ __attribute__((aligned(16))) float Ps[BW_WAVES * KT * (QCB / BW_WAVES)];  // [wave][key][row slot]
  __shared__ __attribute__((aligned(16))) float Red[KT * DR];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = bh / p.H, h = bh % p.H;
  const int k0 = ktile * KT;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const bool dq_atomic = ntiles > 1;

  constexpr int DKH = DK / 2;  // split heads: see attn_fwd_kernel
  const bool split = p.q2 != nullptr;
  const int hq = h * (split ? DKH : DK);
  const float* qb = p.q + (int64_t)b * p.q_bs + hq;
  const float* kb = p.k + (int64_t)b * p.k_bs + hq;
  const float* qb2 = split ? p.q2 + (int64_t)b * p.q_bs + hq - DKH : qb;
  const float* kb2 = split ? p.k2 + (int64_t)b * p.k_bs + hq - DKH : kb;
  const float* kadd = (split && p.k_add) ? p.k_add + (int64_t)b * p.k_bs + hq : nullptr;
  const float* vb = p.v + (int64_t)b * p.v_bs + (int64_t)h * DV;
  const float* ob = p.o + (int64_t)b * p.o_bs + (int64_t)h * DV;
  const float* dob = p.d_o + (int64_t)b * p.o_bs + (int64_t)h * DV;
  float* dqb = p.dq + (int64_t)b * p.q_bs + hq;
  float* dkb = p.dk_ + (int64_t)b * p.k_bs + hq;
  float* dqb2 = split ? p.dq2 + (int64_t)b * p.q_bs + hq - DKH : dqb;
  float* dkb2 = split ? p.dk2 + (int64_t)b * p.k_bs + hq - DKH : dkb;
  float* dvb = p.dv_ + (int64_t)b * p.v_bs + (int64_t)h * DV;

  for (int idx = tid; idx < KT * (DK / 4); idx += BW_THREADS) {
    int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k0 + r < p.Lk) {
      x = *reinterpret_cast<const float4*>((c >= DKH ? kb2 : kb) + (int64_t)(k0 + r) * p.k_ls + c);
      if (kadd && c < DKH) {
        const float4 y = *reinterpret_cast<const float4*>(kadd + (int64_t)(k0 + r) * p.k_ls + c);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
    }
    *reinterpret_cast<float4*>(Ks + r * SK + c) = x;
  }
  for (int idx = tid; idx < KT * (DV / 4); idx += BW_THREADS) {
    int r = idx / (DV / 4), c = (idx % (DV / 4)) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (k0 + r < p.Lk) x = *reinterpret_cast<const float4*>(vb + (int64_t)(k0 + r) * p.v_ls + c);
    *reinterpret_cast<float4*>(Vs + r * SV + c) = x;
  }
  const int j = k0 + lane;
  const bool jvalid = j < p.Lk;
  bool kp = !jvalid, kp2 = false;
  if (jvalid && p.kpad) kp = p.kpad[(int64_t)b * p.Lk + j] != 0;
  if (jvalid && quirk) kp2 = p.kpad[(int64_t)b2 * p.Lk + j] != 0;
  __syncthreads();

  float kreg[DK], vreg[DV], dkacc[DK], dvacc[DV];
#pragma unroll
  for (int c = 0; c < DK; c += 4) {
    float4 x = *reinterpret_cast<const float4*>(Ks + lane * SK + c);
    kreg[c] = x.x; kreg[c + 1] = x.y; kreg[c + 2] = x.z; kreg[c + 3] = x.w;
    dkacc[c] = dkacc[c + 1] = dkacc[c + 2] = dkacc[c + 3] = 0.0f;
  }
#pragma unroll
  for (int c = 0; c < DV; c += 4) {
    float4 x = *reinterpret_cast<const float4*>(Vs + lane * SV + c);
    vreg[c] = x.x; vreg[c + 1] = x.y; vreg[c + 2] = x.z; vreg[c + 3] = x.w;
    dvacc[c] = dvacc[c + 1] = dvacc[c + 2] = dvacc[c + 3] = 0.0f;
  }

  const uint32_t thresh = p.drop_p > 0.f ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);

  for (int qc = 0; qc < p.Lq; qc += QCB) {
    const int nq = (p.Lq - qc) < QCB ? (p.Lq - qc) : QCB;
    __syncthreads();  // previous chunk consumed
    for (int idx = tid; idx < QCB * (DK / 4); idx += BW_THREADS) {
      int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
      float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
      if (r < nq) x = *reinterpret_cast<const float4*>((c >= DKH ? qb2 : qb) + (int64_t)(qc + r) * p.q_ls + c);
      *reinterpret_cast<float4*>(Qs + r * DK + c) = x;
    }
    // dO chunk + delta_i = sum_d dO[i,d] * O[i,d]; DV/4 consecutive threads share a row
    for (int idx = tid; idx < QCB * (DV / 4); idx += BW_THREADS) {
      int r = idx / (DV / 4), c = (idx % (DV / 4)) * 4;
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
      if (r < nq) {
        g = *reinterpret_cast<const float4*>(dob + (int64_t)(qc + r) * p.o_ls + c);
        o = *reinterpret_cast<const float4*>(ob + (int64_t)(qc + r) * p.o_ls + c);
      }
      *reinterpret_cast<float4*>(dOs + r * DV + c) = g;
      float part = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
      part = sum_within<DV / 4>(part);
      if ((idx % (DV / 4)) == 0) Dl[r] = part;
    }
    if (tid < QCB) Lse[tid] = (tid < nq) ? p.lse[(int64_t)bh * p.Lq + qc + tid] : 0.0f;
    __syncthreads();

    // rows wave, wave + 8, wave + 16, wave + 24 of the chunk: per-row phase (P recomputed, dV, dS, dK
    // in registers; dS into Ps[key][row slot]), then ONE dQ sweep over the keys for the four rows
    constexpr int RB = QCB / BW_WAVES;
#pragma unroll
    for (int rr = 0; rr < RB; ++rr) {
      const int r = wave + rr * BW_WAVES;
      float ds = 0.0f;
      if (r < nq) {
        const int i = qc + r;
        float s = 0.0f;
#pragma unroll
        for (int c = 0; c < DK; c += 4) {
          float4 qv = *reinterpret_cast<const float4*>(Qs + r * DK + c);
          s += qv.x * kreg[c] + qv.y * kreg[c + 1] + qv.z * kreg[c + 2] + qv.w * kreg[c + 3];
        }
        s *= p.scale;
        bool masked = kp;
        if (quirk) {
          bool qp = p.qpad[(int64_t)b2 * p.Lq + i] != 0;
          masked = masked || (qp && kp2);
        }
        float pj = masked ? 0.0f : __expf(s - Lse[r]);
        float km = 1.0f;
        if (thresh) {
          uint32_t idx = (uint32_t)(((int64_t)bh * p.Lq + i) * p.Lk + j);
          km = mesm_hash32(idx, drop_seed) >= thresh ? inv_keep : 0.0f;
        }
        const float pd = pj * km;
        float dp = 0.0f;
#pragma unroll
        for (int c = 0; c < DV; c += 4) {
          float4 g = *reinterpret_cast<const float4*>(dOs + r * DV + c);
          dp += g.x * vreg[c] + g.y * vreg[c + 1] + g.z * vreg[c + 2] + g.w * vreg[c + 3];
          dvacc[c] += pd * g.x; dvacc[c + 1] += pd * g.y;
          dvacc[c + 2] += pd * g.z; dvacc[c + 3] += pd * g.w;
        }
        ds = pj * (dp * km - Dl[r]) * p.scale;
#pragma unroll
        for (int c = 0; c < DK; c += 4) {
          float4 qv = *reinterpret_cast<const float4*>(Qs + r * DK + c);
          dkacc[c] += ds * qv.x; dkacc[c + 1] += ds * qv.y;
          dkacc[c + 2] += ds * qv.z; dkacc[c + 3] += ds * qv.w;
        }
      }
      Ps[(wave * KT + lane) * RB + rr] = ds;
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    {
      const int d = lane % DK, g = lane / DK;
      float acc[RB];
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) acc[rr] = 0.0f;
      const int kend = (p.Lk - k0) < KT ? (p.Lk - k0) : KT;  // dS of the keys beyond Lk is 0
#pragma unroll 4
      for (int jj = g; jj < kend; jj += GQ) {
        const float4 pr = *reinterpret_cast<const float4*>(Ps + (wave * KT + jj) * RB);
        const float kv = Ks[jj * SK + d];
        acc[0] += pr.x * kv; acc[1] += pr.y * kv; acc[2] += pr.z * kv; acc[3] += pr.w * kv;
      }
#pragma unroll
      for (int rr = 0; rr < RB; ++rr) {
        const int r = wave + rr * BW_WAVES;
        const float t = sum_across_groups<DK>(acc[rr]);
        if (g == 0 && r < nq) {
          float* dst = (d >= DKH ? dqb2 : dqb) + (int64_t)(qc + r) * p.q_ls + d;
          if (dq_atomic) atomicAdd(dst, t);
          else *dst = t;
        }
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }

  // deterministic cross-wave reduction of dK then dV through LDS, wave by wave
  for (int w = 0; w < BW_WAVES; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < DK; ++c) {
        float t = dkacc[c];
        if (w > 0) t += Red[lane * DR + c];
        Red[lane * DR + c] = t;
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < KT * DK; idx += BW_THREADS) {
    int r = idx / DK, c = idx % DK;
    if (k0 + r < p.Lk) mesm_store_wt((c >= DKH ? dkb2 : dkb) + ((int64_t)(k0 + r) * p.k_ls + c), Red[r * DR + c]);
  }
  for (int w = 0; w < BW_WAVES; ++w) {
    __syncthreads();
    if (wave == w) {
#pragma unroll
      for (int c = 0; c < DV; ++c) {
        float t = dvacc[c];
        if (w > 0) t += Red[lane * DR + c];
        Red[lane * DR + c] = t;
      }
    }
  }
  __syncthreads();
  for (int idx = tid; idx < KT * DV; idx += BW_THREADS) {
    int r = idx / DV, c = idx % DV;
    if (k0 + r < p.Lk) mesm_store_wt(dvb + ((int64_t)(k0 + r) * p.v_ls + c), Red[r * DR + c]);
  }
}

template <int DK, int DV, int BW_WAVES>
__global__ __launch_bounds__(64 * BW_WAVES) void attn_bwd_kernel(const MesmAttnArgs p) {
  attn_bwd_body<DK, DV, BW_WAVES>(p, blockIdx.x, blockIdx.y, gridDim.y);
}

// Grouped launch (see attn_mfma_fwd_group_kernel): up to 8 independent backward problems with dk = dv = 32 and
// packed heads in ONE launch of 4-wave workgroups; workgroup -> (problem, batch * head, key tile).
constexpr int ATTNB_GROUP_MAX = 8;
struct AttnBGroup {
  MesmAttnArgs p[ATTNB_GROUP_MAX];
  int start[ATTNB_GROUP_MAX + 1];
  int n;
};

__global__ __launch_bounds__(256) void attn_bwd_group_kernel(const AttnBGroup g) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < ATTNB_GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const MesmAttnArgs p = *reinterpret_cast<const MesmAttnArgs*>(ka + offsetof(AttnBGroup, p) + (size_t)gi * sizeof(MesmAttnArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(AttnBGroup, start) + (size_t)gi * sizeof(int));
  const int local = bid - first;
  const int ntiles = (p.Lk + KT - 1) / KT;
  const int bh = local / ntiles;
  attn_bwd_body<32, 32, 4>(p, bh, local - bh * ntiles, ntiles);
}

// the 16 x 16-block matrix-core backward (attention_blk.hip) takes the short ranges; MESM_ATTN_BLK=0 (or
// MESM_ATTN_LEGACY) keeps the lane-per-key kernels above, for A/B
bool blk_bwd_enabled() {
  static const bool on = [] {
    const char* e = getenv("MESM_ATTN_BLK");
    return getenv("MESM_ATTN_LEGACY") == nullptr && !(e && atoi(e) == 0);
  }();
  return on;
}

bool blk_fwd_enabled() {
  static const bool on = [] {
    const char* e = getenv("MESM_ATTN_BLK_FWD");
    return getenv("MESM_ATTN_LEGACY") == nullptr && !(e && atoi(e) == 0);
  }();
  return on;
}

int check_common(const MesmAttnArgs& a) {
  if (!a.q || !a.k || !a.v || !a.o) return MESM_EINVAL;
  if (a.B <= 0 || a.H <= 0 || a.Lq <= 0 || a.Lk <= 0) return MESM_EINVAL;
  if (a.drop_p < 0.f || a.drop_p >= 1.f) return MESM_EINVAL;
  if (a.mask_mode == MESM_MASK_T2V_QUIRK && (!a.qpad || !a.kpad)) return MESM_EINVAL;
  // float4 staging: every row start must be 16-byte aligned
  const int64_t strides[8] = {a.q_bs, a.q_ls, a.k_bs, a.k_ls, a.v_bs, a.v_ls, a.o_bs, a.o_ls};
  for (int64_t s : strides)
    if (s % 4 != 0) return MESM_EALIGN;
  const void* ptrs[4] = {a.q, a.k, a.v, a.o};
  for (const void* ptr : ptrs)
    if (((uintptr_t)ptr % 16) != 0) return MESM_EALIGN;
  if (a.dk % 4 != 0 || a.dv % 4 != 0) return MESM_EINVAL;
  if ((a.q2 == nullptr) != (a.k2 == nullptr)) return MESM_EINVAL;
  if (a.k_add && (!a.k2 || ((uintptr_t)a.k_add % 16) != 0)) return MESM_EINVAL;
  if (a.q2 && (a.dk % 8 != 0 || ((uintptr_t)a.q2 % 16) != 0 || ((uintptr_t)a.k2 % 16) != 0)) return MESM_EALIGN;
  return MESM_OK;
}

#define ATTN_DISPATCH(KERNEL, GRID, AT_THREADS, ...)                                          \
  do {                                                                                     \
    if (a.dk == 32 && a.dv == 32) hipLaunchKernelGGL((KERNEL<32, 32 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 64 && a.dv == 32) hipLaunchKernelGGL((KERNEL<64, 32 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 8 && a.dv == 8) hipLaunchKernelGGL((KERNEL<8, 8 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 16 && a.dv == 8) hipLaunchKernelGGL((KERNEL<16, 8 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 16 && a.dv == 16) hipLaunchKernelGGL((KERNEL<16, 16 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 32 && a.dv == 16) hipLaunchKernelGGL((KERNEL<32, 16 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else if (a.dk == 64 && a.dv == 64) hipLaunchKernelGGL((KERNEL<64, 64 __VA_OPT__(,) __VA_ARGS__>), GRID, dim3(AT_THREADS), 0, s, a); \
    else return MESM_EINVAL;                                                               \
  } while (0)

}  // namespace

extern "C" int mesm_attn_fwd(const MesmAttnArgs* args, void* stream) {
  if (!args) return MESM_EINVAL;
  MesmAttnArgs a = *args;
  int rc = check_common(a);
  if (rc != MESM_OK) return rc;
  hipStream_t s = (hipStream_t)stream;
  // the step's hot shapes run on the matrix cores (MESM_ATTN_LEGACY=1: this file's lane-per-key kernels, for A/B)
  static const bool legacy = getenv("MESM_ATTN_LEGACY") != nullptr;
  if (blk_fwd_enabled() && mesm_attn_blk_fwd_ok(a)) return mesm_attn_blk_fwd(a, s);
  // long key ranges with enough query rows to fill the chip (the rule of the 32 x 32 two-pass kernel it replaces)
  if (blk_fwd_enabled() && a.Lk > 128 && a.Lq >= 128 && mesm_attn_blk_fwd_long_ok(a)) return mesm_attn_blk_fwd_long(a, s);
  if (!legacy && mesm_attn_mfma_ok(a)) return mesm_attn_mfma_fwd(a, s);
  dim3 grid(a.B * a.H, (a.Lq + QCH - 1) / QCH);
  ATTN_DISPATCH(attn_fwd_kernel, grid, AT_THREADS);
  return mesm_launch_status();
}

// Does mesm_attn_bwd ADD into dq for this shape (the lane-per-key kernel with several 64-key tiles: the caller must
// hand in zeros), or write it (every matrix-core path, one key tile)?  Same predicates as the dispatch below, on a
// shape-only argument block (packed layouts), so callers can skip the zero fill.
extern "C" int mesm_attn_bwd_accumulates_dq(int32_t B, int32_t H, int32_t Lq, int32_t Lk, int32_t dk, int32_t dv,
                                            int32_t split) {
  MesmAttnArgs a = {};
  static float dummy;
  a.B = B; a.H = H; a.Lq = Lq; a.Lk = Lk; a.dk = dk; a.dv = dv;
  a.q_ls = a.k_ls = (int64_t)H * (split ? dk / 2 : dk);
  a.v_ls = a.o_ls = (int64_t)H * dv;
  a.q_bs = a.q_ls * Lq; a.k_bs = a.k_ls * Lk; a.v_bs = a.v_ls * Lk; a.o_bs = a.o_ls * Lq;
  if (split) { a.q2 = a.k2 = &dummy; a.dq2 = a.dk2 = &dummy; }
  static const bool legacy = getenv("MESM_ATTN_LEGACY") != nullptr;
  if (!legacy && mesm_attn_mfma_bwd_ok(a)) return 0;
  if (blk_bwd_enabled() && (mesm_attn_blk_bwd_ok(a) || (a.Lk > 128 && mesm_attn_blk_bwd_long_ok(a)))) return 0;
  return Lk > KT ? 1 : 0;
}

extern "C" int mesm_attn_bwd(const MesmAttnArgs* args, void* stream) {
  if (!args) return MESM_EINVAL;
  MesmAttnArgs a = *args;
  int rc = check_common(a);
  if (rc != MESM_OK) return rc;
  if (!a.lse || !a.d_o || !a.dq || !a.dk_ || !a.dv_) return MESM_EINVAL;
  if (a.mask_mode == MESM_MASK_CAUSAL) return MESM_EINVAL;  // the causal text encoder is frozen: forward only
  if (a.q2 && (!a.dq2 || !a.dk2 || ((uintptr_t)a.dq2 % 16) != 0 || ((uintptr_t)a.dk2 % 16) != 0)) return MESM_EINVAL;
  const void* ptrs[4] = {a.d_o, a.dq, a.dk_, a.dv_};
  for (const void* ptr : ptrs)
    if (((uintptr_t)ptr % 16) != 0) return MESM_EALIGN;
  hipStream_t s = (hipStream_t)stream;
  static const bool legacy = getenv("MESM_ATTN_LEGACY") != nullptr;
  if (!legacy && mesm_attn_mfma_bwd_ok(a)) return mesm_attn_mfma_bwd(a, s);
  if (blk_bwd_enabled() && mesm_attn_blk_bwd_ok(a)) return mesm_attn_blk_bwd(a, s);
  if (blk_bwd_enabled() && a.Lk > 128 && mesm_attn_blk_bwd_long_ok(a)) return mesm_attn_blk_bwd_long(a, s);
  dim3 grid(a.B * a.H, (a.Lk + KT - 1) / KT);
  if (grid.y > 1) ATTN_DISPATCH(attn_bwd_kernel, grid, 128, 2);  // several key tiles: 2 waves per workgroup
  else ATTN_DISPATCH(attn_bwd_kernel, grid, 256, 4);
  return mesm_launch_status();
}

namespace {
int check_bwd(const MesmAttnArgs& a) {
  int rc = check_common(a);
  if (rc != MESM_OK) return rc;
  if (!a.lse || !a.d_o || !a.dq || !a.dk_ || !a.dv_) return MESM_EINVAL;
  if (a.mask_mode == MESM_MASK_CAUSAL) return MESM_EINVAL;
  if (a.q2 && (!a.dq2 || !a.dk2 || ((uintptr_t)a.dq2 % 16) != 0 || ((uintptr_t)a.dk2 % 16) != 0)) return MESM_EINVAL;
  const void* ptrs[4] = {a.d_o, a.dq, a.dk_, a.dv_};
  for (const void* ptr : ptrs)
    if (((uintptr_t)ptr % 16) != 0) return MESM_EALIGN;
  return MESM_OK;
}
}  // namespace

extern "C" int mesm_attn_fwd_group(const MesmAttnArgs* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 64) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static const bool legacy = getenv("MESM_ATTN_LEGACY") != nullptr;
  MesmAttnArgs grp[8], blk[8];
  int ng = 0, nblk = 0, rc = MESM_OK;
  auto flush = [&]() {
    if (nblk > 0) {
      rc = nblk == 1 ? mesm_attn_blk_fwd(blk[0], s) : mesm_attn_blk_fwd_group(blk, nblk, s);
      nblk = 0;
      if (rc != MESM_OK) return;
    }
    if (ng == 0) return;
    rc = ng == 1 ? mesm_attn_mfma_fwd(grp[0], s) : mesm_attn_mfma_fwd_group(grp, ng, s);
    ng = 0;
  };
  for (int i = 0; i < n && rc == MESM_OK; ++i) {
    rc = check_common(list[i]);
    if (rc != MESM_OK) return rc;
    if (blk_fwd_enabled() && mesm_attn_blk_fwd_groupable(list[i])) {
      blk[nblk++] = list[i];
      if (nblk == 8) flush();
    } else if (!legacy && mesm_attn_mfma_groupable(list[i])) {
      grp[ng++] = list[i];
      if (ng == 8) flush();
    } else {
      rc = mesm_attn_fwd(&list[i], stream);
    }
  }
  if (rc == MESM_OK) flush();
  return rc;
}

extern "C" int mesm_attn_bwd_group(const MesmAttnArgs* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 64) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  static const bool legacy = getenv("MESM_ATTN_LEGACY") != nullptr;
  AttnBGroup g;
  g.n = 0;
  g.start[0] = 0;
  MesmAttnArgs blk[8];
  int nblk = 0;
  int rc = MESM_OK;
  auto flush = [&]() {
    if (nblk > 0) {
      rc = nblk == 1 ? mesm_attn_blk_bwd(blk[0], s) : mesm_attn_blk_bwd_group(blk, nblk, s);
      nblk = 0;
      if (rc != MESM_OK) return;
    }
    if (g.n == 0) return;
    if (g.n == 1) {
      rc = mesm_attn_bwd(&g.p[0], stream);
    } else {
      hipLaunchKernelGGL(attn_bwd_group_kernel, dim3((unsigned)g.start[g.n]), dim3(256), 0, s, g);
      rc = mesm_launch_status();
    }
    g.n = 0;
  };
  for (int i = 0; i < n && rc == MESM_OK; ++i) {
    const MesmAttnArgs& a = list[i];
    rc = check_bwd(a);
    if (rc != MESM_OK) return rc;
    const bool lane_per_key = legacy || !mesm_attn_mfma_bwd_ok(a);
    if (lane_per_key && blk_bwd_enabled() && mesm_attn_blk_bwd_groupable(a)) {
      blk[nblk++] = a;
      if (nblk == 8) flush();
    } else if (lane_per_key && blk_bwd_enabled() &&
               (mesm_attn_blk_bwd_ok(a) || (a.Lk > 128 && mesm_attn_blk_bwd_long_ok(a)))) {
      // a block kernel takes it when issued alone and WRITES dq (mesm_attn_bwd_accumulates_dq says 0 for it, so the
      // caller hands in uninitialised memory): it must never reach the lane-per-key group, which ADDS into dq
      rc = mesm_attn_bwd(&a, stream);
    } else if (lane_per_key && a.dk == 32 && a.dv == 32 && !a.q2) {
      g.p[g.n] = a;
      g.start[g.n + 1] = g.start[g.n] + a.B * a.H * ((a.Lk + KT - 1) / KT);
      if (++g.n == ATTNB_GROUP_MAX) flush();
    } else {
      rc = mesm_attn_bwd(&a, stream);
    }
  }
  if (rc == MESM_OK) flush();
  return rc;
}
