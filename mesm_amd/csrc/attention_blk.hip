// Attention backward on 16 x 16 score blocks (v_mfma_f32_16x16x4_f32) for the step's short ranges: dk = dv = 32, packed
// heads, Lq and Lk up to ~100 (transformer.py:528-533, 592-598, 640-644 backward; attention.py:329-386).
//
// One workgroup (4 waves) per (batch, head).  Q, dO, K, V of the head are staged ONCE into LDS (36-float rows: the
// fragment reads below are conflict-free), with delta_i = dO_i . O_i, the saved log-sum-exp and the mask bytes beside
// them.  The work is cut into UNITS that need no exchange between waves and no atomics:
//
//   J unit (one 16-key block, all query blocks):  S = Q K^T and dP = dO V^T with the KEY on the lane
//       (C/D layout: column = lane & 15 = key, rows = 4 (lane >> 4) + r = queries), so the four accumulator registers
//       of P_drop and dS ARE the A operands of dV += P_drop^T dO and dK += dS^T Q (sum over the block's queries);
//       dV / dK of the key block stay in registers over the whole sweep and are stored once.
//   I unit (one 16-query block, all key blocks):  the TRANSPOSED blocks S^T = K Q^T, dP^T = V dO^T (operands swapped:
//       the QUERY on the lane), so dS^T's registers are the A operands of dQ += dS K.
//
// The score blocks are computed twice (once per orientation): 56 instead of 40 matrix instructions per block pair, in
// exchange for no LDS transposition, no cross-wave reduction of dQ and 45 KB of LDS at 76 x 76 (three workgroups per
// CU).  Waves draw units from an LDS counter (J units first: they are the longer ones).  The lane-per-key kernel of
// attention.hip used half its lanes at Lk = 33 / 76 and walked the queries four at a time:
// 75 x 33 / 76 x 76 / 33 x 75 (64 x 8 heads) 28 / 51 / 27 us there.
#include <hip/hip_runtime.h>

#include "attention_blk.hpp"
#include "common.hpp"

// Output stores: write-through (common.hpp mesm_store_wt) unless built with -DMESM_ATTN_WT=0
#ifndef MESM_ATTN_WT
#define MESM_ATTN_WT 1
#endif
#if MESM_ATTN_WT
#define ATTN_ST(ptr, val) mesm_store_wt((ptr), (val))
#else
#define ATTN_ST(ptr, val) (*(ptr) = (val))
#endif

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int BS = 36;  // LDS row stride in floats
#ifndef MESM_BLK_THREADS
#define MESM_BLK_THREADS 512
#endif
constexpr int BT = MESM_BLK_THREADS;  // 8 waves: a 76 x 76 head has 5 + 5 units

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

template <int DK>
struct BlkShape {
  static constexpr int QS = DK + 4;  // Q / K row stride (dO / V rows: BS)
  int LqP, LkP;                      // padded to 16
  __host__ __device__ BlkShape(int Lq, int Lk) : LqP((Lq + 15) & ~15), LkP((Lk + 15) & ~15) {}
  __host__ __device__ size_t floats() const { return (size_t)(LqP + LkP) * (QS + BS) + 3 * LqP + 2 * LkP + 4; }
};

// N consecutive floats of an LDS row (the lane's share of a 16 x DK operand: reduce indices N kq .. N kq + N - 1)
template <int N>
__device__ __forceinline__ void loadN(const float* p, float* f) {
#pragma unroll
  for (int t = 0; t < N; t += 4) {
    const float4 a = *reinterpret_cast<const float4*>(p + t);
    f[t] = a.x; f[t + 1] = a.y; f[t + 2] = a.z; f[t + 3] = a.w;
  }
}

// DK = 64: the decoder's split heads ([content || position] halves of q / q2, k (+ k_add) / k2; the gradients go
// back into dq / dq2, dk_ / dk2), or a packed 64-wide head.
template <int DK, bool DROP>
__device__ __forceinline__ void attn_blk_bwd_body(const MesmAttnArgs& p, const int bh) {
  constexpr int QS = DK + 4, DQ = DK / 4, ND = DK / 16;
  constexpr int DKH = DK / 2;
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const BlkShape<DK> sh(p.Lq, p.Lk);
  float* Qs = smem;
  float* Gs = Qs + sh.LqP * QS;   // dO
  float* Ks = Gs + sh.LqP * BS;
  float* Vs = Ks + sh.LkP * QS;
  float* Lse = Vs + sh.LkP * BS;  // LqP
  float* Dl = Lse + sh.LqP;       // LqP
  float* Qp2 = Dl + sh.LqP;       // LqP: query padded in the quirk row
  float* Kp = Qp2 + sh.LqP;       // LkP: key masked for this row (own padding, or beyond Lk)
  float* Kp2 = Kp + sh.LkP;       // LkP: key padded in the quirk row
  int* next = reinterpret_cast<int*>(Kp2 + sh.LkP);

  const int tid = threadIdx.x, lane = tid & 63;
  const int b = bh / p.H, h = bh % p.H;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const int Lq = p.Lq, Lk = p.Lk;

  const bool split = DK == 64 && p.q2 != nullptr;
  const int hq = h * (split ? DKH : DK);
  const float* qb = p.q + (int64_t)b * p.q_bs + hq;
  const float* kb = p.k + (int64_t)b * p.k_bs + hq;
  const float* qb2 = split ? p.q2 + (int64_t)b * p.q_bs + hq - DKH : qb;
  const float* kb2 = split ? p.k2 + (int64_t)b * p.k_bs + hq - DKH : kb;
  const float* kadd = (split && p.k_add) ? p.k_add + (int64_t)b * p.k_bs + hq : nullptr;
  const float* vb = p.v + (int64_t)b * p.v_bs + h * 32;
  const float* ob = p.o + (int64_t)b * p.o_bs + h * 32;
  const float* gb = p.d_o + (int64_t)b * p.o_bs + h * 32;

  // ---- stage the head
  for (int idx = tid; idx < sh.LqP * (DK / 4); idx += BT) {
    const int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
    float4 q = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Lq) q = *reinterpret_cast<const float4*>((split && c >= DKH ? qb2 : qb) + (int64_t)r * p.q_ls + c);
    *reinterpret_cast<float4*>(Qs + r * QS + c) = q;
  }
  for (int idx = tid; idx < sh.LqP * 8; idx += BT) {  // 8 consecutive threads per 32-float row
    const int r = idx >> 3, c = (idx & 7) * 4;
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), o = g;
    if (r < Lq) {
      g = *reinterpret_cast<const float4*>(gb + (int64_t)r * p.o_ls + c);
      o = *reinterpret_cast<const float4*>(ob + (int64_t)r * p.o_ls + c);
    }
    *reinterpret_cast<float4*>(Gs + r * BS + c) = g;
    float part = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
    part = sum_within<8>(part);
    if ((idx & 7) == 0) Dl[r] = part;
  }
  for (int idx = tid; idx < sh.LkP * (DK / 4); idx += BT) {
    const int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
    float4 k = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Lk) {
      k = *reinterpret_cast<const float4*>((split && c >= DKH ? kb2 : kb) + (int64_t)r * p.k_ls + c);
      if (kadd && c < DKH) {  // content half = kcontent + kpos (decoder layer 0, transformer.py:773-776)
        const float4 y = *reinterpret_cast<const float4*>(kadd + (int64_t)r * p.k_ls + c);
        k.x += y.x; k.y += y.y; k.z += y.z; k.w += y.w;
      }
    }
    *reinterpret_cast<float4*>(Ks + r * QS + c) = k;
  }
  for (int idx = tid; idx < sh.LkP * 8; idx += BT) {
    const int r = idx >> 3, c = (idx & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Lk) v = *reinterpret_cast<const float4*>(vb + (int64_t)r * p.v_ls + c);
    *reinterpret_cast<float4*>(Vs + r * BS + c) = v;
  }
  for (int r = tid; r < sh.LqP; r += BT) {
    // rows beyond Lq: exp(s - 1e30) = 0, so they add nothing anywhere
    Lse[r] = r < Lq ? p.lse[(int64_t)bh * Lq + r] : 1e30f;
    Qp2[r] = (quirk && r < Lq && p.qpad[(int64_t)b2 * Lq + r] != 0) ? 1.0f : 0.0f;
  }
  for (int j = tid; j < sh.LkP; j += BT) {
    bool m = j >= Lk;
    if (!m && p.kpad) m = p.kpad[(int64_t)b * Lk + j] != 0;
    Kp[j] = m ? 1.0f : 0.0f;
    Kp2[j] = (quirk && j < Lk && p.kpad[(int64_t)b2 * Lk + j] != 0) ? 1.0f : 0.0f;
  }
  if (tid == 0) *next = 0;
  __syncthreads();

  const uint32_t thresh = DROP ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const float scale = p.scale;
  const uint32_t row0 = (uint32_t)bh * (uint32_t)Lq;  // dropout index = (row0 + i) * Lk + j

  const int nI = sh.LqP >> 4, nJ = sh.LkP >> 4;
  const int jl = lane & 15, kq = lane >> 4;

  for (;;) {
    int u = 0;
    if (lane == 0) u = atomicAdd(next, 1);
    u = __builtin_amdgcn_readfirstlane(u);
    if (u >= nI + nJ) break;
    if (u < nJ) {
      // ------------------------------------------------------------------ J unit: keys j0 .. j0 + 15
      const int j0 = u << 4;
      float kf[DQ], vf[8];
      loadN<DQ>(Ks + (j0 + jl) * QS + DQ * kq, kf);
      loadN<8>(Vs + (j0 + jl) * BS + 8 * kq, vf);
      const float kpj = Kp[j0 + jl], kp2j = Kp2[j0 + jl];
      const uint32_t j = (uint32_t)(j0 + jl);
      f32x4 dVa[2], dKa[ND];
#pragma unroll
      for (int c = 0; c < 2; ++c) dVa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int c = 0; c < ND; ++c) dKa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int ib = 0; ib < nI; ++ib) {
        const int i0 = ib << 4;
        float qf[DQ], gf[8];
        loadN<DQ>(Qs + (i0 + jl) * QS + DQ * kq, qf);
        loadN<8>(Gs + (i0 + jl) * BS + 8 * kq, gf);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          s = mfma16(qf[t], kf[t], s);     // S[i0 + 4 kq + r][j0 + jl]
          dp = mfma16(gf[t], vf[t], dp);   // dP, same layout
        }
#pragma unroll
        for (int t = 8; t < DQ; ++t) s = mfma16(qf[t], kf[t], s);
        const int ir = i0 + 4 * kq;
        // B operands of the products below (rows ir + r of dO and Q, 16-column blocks): issued ahead of the
        // element-wise work so their LDS latency hides behind it
        float gbv[2][4], qbv[ND][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int c = 0; c < 2; ++c) gbv[c][r] = Gs[(ir + r) * BS + 16 * c + jl];
#pragma unroll
          for (int c = 0; c < ND; ++c) qbv[c][r] = Qs[(ir + r) * QS + 16 * c + jl];
        }
        const float4 lse4 = *reinterpret_cast<const float4*>(Lse + ir);
        const float4 dl4 = *reinterpret_cast<const float4*>(Dl + ir);
        const float4 qp4 = *reinterpret_cast<const float4*>(Qp2 + ir);
        const float lse_[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
        const float dl_[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
        const float qp_[4] = {qp4.x, qp4.y, qp4.z, qp4.w};
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mk = fmaf(qp_[r], kp2j, kpj);  // != 0: masked (own key padding, or the quirk row's pair)
          const float e = __expf(s[r] * scale - lse_[r]);
          const float pj = mk != 0.0f ? 0.0f : e;
          float km = 1.0f;
          if (DROP) {
            const uint32_t idx = (row0 + (uint32_t)(ir + r)) * (uint32_t)Lk + j;
            km = mesm_hash32(idx, drop_seed) >= thresh ? inv_keep : 0.0f;
          }
          pd[r] = pj * km;
          ds[r] = pj * (dp[r] * km - dl_[r]) * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
#pragma unroll
          for (int c = 0; c < 2; ++c) dVa[c] = mfma16(pd[r], gbv[c][r], dVa[c]);
#pragma unroll
          for (int c = 0; c < ND; ++c) dKa[c] = mfma16(ds[r], qbv[c][r], dKa[c]);
        }
      }
      float* dkb = p.dk_ + (int64_t)b * p.k_bs + hq;
      float* dkb2 = split ? p.dk2 + (int64_t)b * p.k_bs + hq - DKH : dkb;
      float* dvb = p.dv_ + (int64_t)b * p.v_bs + h * 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = j0 + 4 * kq + r;
        if (jj < Lk) {
#pragma unroll
          for (int c = 0; c < ND; ++c) ATTN_ST((16 * c >= DKH ? dkb2 : dkb) + ((int64_t)jj * p.k_ls + 16 * c + jl), dKa[c][r]);
#pragma unroll
          for (int c = 0; c < 2; ++c) ATTN_ST(dvb + ((int64_t)jj * p.v_ls + 16 * c + jl), dVa[c][r]);
        }
      }
    } else {
      // ------------------------------------------------------------------ I unit: queries i0 .. i0 + 15
      const int i0 = (u - nJ) << 4;
      float qf[DQ], gf[8];
      loadN<DQ>(Qs + (i0 + jl) * QS + DQ * kq, qf);
      loadN<8>(Gs + (i0 + jl) * BS + 8 * kq, gf);
      const float lse_i = Lse[i0 + jl], dl_i = Dl[i0 + jl];
      const float qp_i = Qp2[i0 + jl];
      const uint32_t rowi = (row0 + (uint32_t)(i0 + jl)) * (uint32_t)Lk;
      f32x4 dQa[ND];
#pragma unroll
      for (int c = 0; c < ND; ++c) dQa[c] = (f32x4){0.f, 0.f, 0.f, 0.f};
      for (int jb = 0; jb < nJ; ++jb) {
        const int j0 = jb << 4;
        float kf[DQ], vf[8];
        loadN<DQ>(Ks + (j0 + jl) * QS + DQ * kq, kf);
        loadN<8>(Vs + (j0 + jl) * BS + 8 * kq, vf);
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          st = mfma16(kf[t], qf[t], st);    // S^T[j0 + 4 kq + r][i0 + jl]
          dpt = mfma16(vf[t], gf[t], dpt);
        }
#pragma unroll
        for (int t = 8; t < DQ; ++t) st = mfma16(kf[t], qf[t], st);
        const int jr = j0 + 4 * kq;
        float kbv[ND][4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < ND; ++c) kbv[c][r] = Ks[(jr + r) * QS + 16 * c + jl];
        const float4 kp4 = *reinterpret_cast<const float4*>(Kp + jr);
        const float4 kq4 = *reinterpret_cast<const float4*>(Kp2 + jr);
        const float kp_[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
        const float kp2_[4] = {kq4.x, kq4.y, kq4.z, kq4.w};
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mk = fmaf(qp_i, kp2_[r], kp_[r]);
          const float e = __expf(st[r] * scale - lse_i);
          const float pj = mk != 0.0f ? 0.0f : e;
          float km = 1.0f;
          if (DROP) km = mesm_hash32(rowi + (uint32_t)(jr + r), drop_seed) >= thresh ? inv_keep : 0.0f;
          ds[r] = pj * (dpt[r] * km - dl_i) * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int c = 0; c < ND; ++c) dQa[c] = mfma16(ds[r], kbv[c][r], dQa[c]);
      }
      float* dqb = p.dq + (int64_t)b * p.q_bs + hq;
      float* dqb2 = split ? p.dq2 + (int64_t)b * p.q_bs + hq - DKH : dqb;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = i0 + 4 * kq + r;
        if (ii < Lq) {
#pragma unroll
          for (int c = 0; c < ND; ++c) ATTN_ST((16 * c >= DKH ? dqb2 : dqb) + ((int64_t)ii * p.q_ls + 16 * c + jl), dQa[c][r]);
        }
      }
    }
  }
}

template <int DK, bool DROP>
__global__ __launch_bounds__(BT) void attn_blk_bwd_kernel(const MesmAttnArgs p) { attn_blk_bwd_body<DK, DROP>(p, blockIdx.x); }

// ------------------------------------------------------------------------------------------------------------
// Long ranges (TACoS' 513 x 513 encoder, transformer.py:640-644 backward): a head no longer fits in LDS, so the two
// kinds of units become two kinds of WORKGROUPS -- J workgroups own 128 keys (a wave: 16 keys, K / V fragments and the
// dK / dV accumulators in registers) and stream the queries through LDS in chunks of 64 (Q, dO, delta, log-sum-exp,
// mask bytes); I workgroups own 128 queries and stream the keys.  Same block arithmetic as above, no atomics.
constexpr int LC = 64;   // rows of the streamed side per LDS chunk (128: no change, 393 / 204 vs 395 / 209 us)
constexpr int LW = 128;  // rows a workgroup owns (8 waves x 16)

template <bool DROP>
__global__ __launch_bounds__(BT) void attn_blk_bwd_long_kernel(const MesmAttnArgs p) {
  __shared__ __attribute__((aligned(16))) float Xs[LC * BS];  // Q (J workgroups) / K (I workgroups) chunk
  __shared__ __attribute__((aligned(16))) float Ys[LC * BS];  // dO / V chunk
  __shared__ __attribute__((aligned(16))) float Aux[3][LC];   // lse, delta, quirk-row query padding / key masks
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 15, kq = lane >> 4;
  const int Lq = p.Lq, Lk = p.Lk;
  const int nJc = (Lk + LW - 1) / LW, nIc = (Lq + LW - 1) / LW;
  const int bh = blockIdx.x / (nJc + nIc), part = blockIdx.x % (nJc + nIc);
  const int b = bh / p.H, h = bh % p.H;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const float* qb = p.q + (int64_t)b * p.q_bs + h * 32;
  const float* kb = p.k + (int64_t)b * p.k_bs + h * 32;
  const float* vb = p.v + (int64_t)b * p.v_bs + h * 32;
  const float* ob = p.o + (int64_t)b * p.o_bs + h * 32;
  const float* gb = p.d_o + (int64_t)b * p.o_bs + h * 32;
  const uint32_t thresh = DROP ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const float scale = p.scale;
  const uint32_t row0 = (uint32_t)bh * (uint32_t)Lq;

  auto load8g = [&](const float* base, int64_t ls, int row, int nrows, float* f) {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (row < nrows) {
      a = *reinterpret_cast<const float4*>(base + (int64_t)row * ls + 8 * kq);
      c = *reinterpret_cast<const float4*>(base + (int64_t)row * ls + 8 * kq + 4);
    }
    f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w;
    f[4] = c.x; f[5] = c.y; f[6] = c.z; f[7] = c.w;
  };

  // TAIL workgroups (round 6).  TACoS' encoder runs 513 = 4 x 128 + 1 rows a side: the fifth workgroup of a side owned ONE
  // block of 16 rows -- one busy wave walking all 33 blocks of the other side while seven waves only helped to stage, for as
  // long as a full workgroup takes (2 of 10 workgroups per head in the backward).  A workgroup whose share is a single block
  // now SPREADS that block over its eight waves by (chunk, block) unit -- unit u goes to wave u % 8 -- and adds the waves'
  // partial gradients up through LDS at the end: it is done in about a third of the time.  (Nine-wave workgroups of 144 rows
  // were 45 % slower: 3 + 2 + 2 + 2 waves on the four SIMDs.  Handing the block to the side's last FULL workgroup, every wave
  // taking its turns at it, kept a second set of fragments and accumulators alive: 89 -> 156 registers, one workgroup per CU,
  // 393 -> 473 us.)
  if (part < nJc) {
    // ---------------------------------------------------------------- J workgroup: keys [part * 128, + 128)
    const bool spread = Lk - part * LW <= 16;  // this workgroup's share is one block: every wave works on it
    const int j0 = part * LW + (spread ? 0 : 16 * wave);
    const bool active = j0 < Lk;
    float kf[8], vf[8];
    load8g(kb, p.k_ls, j0 + jl, active ? Lk : 0, kf);
    load8g(vb, p.v_ls, j0 + jl, active ? Lk : 0, vf);
    const int j = j0 + jl;
    float kpj = 1.0f, kp2j = 0.0f;
    if (j < Lk) {
      kpj = (p.kpad && p.kpad[(int64_t)b * Lk + j] != 0) ? 1.0f : 0.0f;
      kp2j = (quirk && p.kpad[(int64_t)b2 * Lk + j] != 0) ? 1.0f : 0.0f;
    }
    f32x4 dVa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    f32x4 dKa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int unit = 0;
    for (int ic = 0; ic < Lq; ic += LC) {
      __syncthreads();  // the previous chunk has been consumed
      for (int idx = tid; idx < LC * 8; idx += BT) {
        const int r = idx >> 3, c = (idx & 7) * 4, i = ic + r;
        float4 q = make_float4(0.f, 0.f, 0.f, 0.f), g = q, o = q;
        if (i < Lq) {
          q = *reinterpret_cast<const float4*>(qb + (int64_t)i * p.q_ls + c);
          g = *reinterpret_cast<const float4*>(gb + (int64_t)i * p.o_ls + c);
          o = *reinterpret_cast<const float4*>(ob + (int64_t)i * p.o_ls + c);
        }
        *reinterpret_cast<float4*>(Xs + r * BS + c) = q;
        *reinterpret_cast<float4*>(Ys + r * BS + c) = g;
        float part_ = g.x * o.x + g.y * o.y + g.z * o.z + g.w * o.w;
        part_ = sum_within<8>(part_);
        if ((idx & 7) == 0) Aux[1][r] = part_;
      }
      if (tid < LC) {
        const int i = ic + tid;
        Aux[0][tid] = i < Lq ? p.lse[(int64_t)bh * Lq + i] : 1e30f;
        Aux[2][tid] = (quirk && i < Lq && p.qpad[(int64_t)b2 * Lq + i] != 0) ? 1.0f : 0.0f;
      }
      __syncthreads();
      if (!active) continue;
      const int nb = (min(LC, Lq - ic) + 15) >> 4;
      for (int ib = 0; ib < nb; ++ib, ++unit) {
        if (spread && (unit & 7) != wave) continue;
        const int i0 = ib << 4;  // inside the chunk
        float qf[8], gf[8];
        loadN<8>(Xs + (i0 + jl) * BS + 8 * kq, qf);
        loadN<8>(Ys + (i0 + jl) * BS + 8 * kq, gf);
        f32x4 s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          s = mfma16(qf[t], kf[t], s);
          dp = mfma16(gf[t], vf[t], dp);
        }
        const int ir = i0 + 4 * kq;
        float gb0[4], gb1[4], qb0[4], qb1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          gb0[r] = Ys[(ir + r) * BS + jl]; gb1[r] = Ys[(ir + r) * BS + 16 + jl];
          qb0[r] = Xs[(ir + r) * BS + jl]; qb1[r] = Xs[(ir + r) * BS + 16 + jl];
        }
        const float4 lse4 = *reinterpret_cast<const float4*>(&Aux[0][ir]);
        const float4 dl4 = *reinterpret_cast<const float4*>(&Aux[1][ir]);
        const float4 qp4 = *reinterpret_cast<const float4*>(&Aux[2][ir]);
        const float lse_[4] = {lse4.x, lse4.y, lse4.z, lse4.w};
        const float dl_[4] = {dl4.x, dl4.y, dl4.z, dl4.w};
        const float qp_[4] = {qp4.x, qp4.y, qp4.z, qp4.w};
        float pd[4], ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mk = fmaf(qp_[r], kp2j, kpj);
          const float e = __expf(s[r] * scale - lse_[r]);
          const float pj = mk != 0.0f ? 0.0f : e;
          float km = 1.0f;
          if (DROP) {
            const uint32_t idx = (row0 + (uint32_t)(ic + ir + r)) * (uint32_t)Lk + (uint32_t)j;
            km = mesm_hash32(idx, drop_seed) >= thresh ? inv_keep : 0.0f;
          }
          pd[r] = pj * km;
          ds[r] = pj * (dp[r] * km - dl_[r]) * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dVa[0] = mfma16(pd[r], gb0[r], dVa[0]);
          dKa[0] = mfma16(ds[r], qb0[r], dKa[0]);
          dVa[1] = mfma16(pd[r], gb1[r], dVa[1]);
          dKa[1] = mfma16(ds[r], qb1[r], dKa[1]);
        }
      }
    }
    if (spread) {
      // the waves' shares of the block's gradients meet in LDS (the chunk buffers are free), wave after wave; wave 0 stores
      for (int w = 0; w < BT / 64; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
          for (int a = 0; a < 4; ++a) {
            f32x4& v = a == 0 ? dVa[0] : (a == 1 ? dVa[1] : (a == 2 ? dKa[0] : dKa[1]));
            float4* slot = reinterpret_cast<float4*>(Xs) + a * 64 + lane;  // 4 x 64 x 4 = 1024 of Xs' 2304 floats
            if (w > 0) {
              const float4 t = *slot;
              v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
            if (w + 1 < BT / 64) *slot = make_float4(v[0], v[1], v[2], v[3]);
          }
        }
      }
    }
    if (active && (!spread || wave == BT / 64 - 1)) {
      float* dkb = p.dk_ + (int64_t)b * p.k_bs + h * 32;
      float* dvb = p.dv_ + (int64_t)b * p.v_bs + h * 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int jj = j0 + 4 * kq + r;
        if (jj < Lk) {
          ATTN_ST(dkb + ((int64_t)jj * p.k_ls + jl), dKa[0][r]);
          ATTN_ST(dkb + ((int64_t)jj * p.k_ls + 16 + jl), dKa[1][r]);
          ATTN_ST(dvb + ((int64_t)jj * p.v_ls + jl), dVa[0][r]);
          ATTN_ST(dvb + ((int64_t)jj * p.v_ls + 16 + jl), dVa[1][r]);
        }
      }
    }
  } else {
    // ---------------------------------------------------------------- I workgroup: queries [ic0, ic0 + 128)
    const bool spread = Lq - (part - nJc) * LW <= 16;  // (see the J side)
    const int i0 = (part - nJc) * LW + (spread ? 0 : 16 * wave);
    const bool active = i0 < Lq;
    const int i = i0 + jl;
    float qf[8], gf[8], of[8];
    load8g(qb, p.q_ls, i, active ? Lq : 0, qf);
    load8g(gb, p.o_ls, i, active ? Lq : 0, gf);
    load8g(ob, p.o_ls, i, active ? Lq : 0, of);
    float dl_i = 0.0f;
#pragma unroll
    for (int t = 0; t < 8; ++t) dl_i += gf[t] * of[t];
    dl_i = add_xor32(add_xor16(dl_i));  // the row's other 24 features sit in the lanes 16 / 32 / 48 apart
    const float lse_i = (active && i < Lq) ? p.lse[(int64_t)bh * Lq + i] : 1e30f;
    const float qp_i = (quirk && active && i < Lq && p.qpad[(int64_t)b2 * Lq + i] != 0) ? 1.0f : 0.0f;
    const uint32_t rowi = (row0 + (uint32_t)i) * (uint32_t)Lk;
    f32x4 dQa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    int unit = 0;
    for (int jc = 0; jc < Lk; jc += LC) {
      __syncthreads();
      for (int idx = tid; idx < LC * 8; idx += BT) {
        const int r = idx >> 3, c = (idx & 7) * 4, jj = jc + r;
        float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
        if (jj < Lk) {
          k = *reinterpret_cast<const float4*>(kb + (int64_t)jj * p.k_ls + c);
          v = *reinterpret_cast<const float4*>(vb + (int64_t)jj * p.v_ls + c);
        }
        *reinterpret_cast<float4*>(Xs + r * BS + c) = k;
        *reinterpret_cast<float4*>(Ys + r * BS + c) = v;
      }
      if (tid < LC) {
        const int jj = jc + tid;
        bool m = jj >= Lk;
        if (!m && p.kpad) m = p.kpad[(int64_t)b * Lk + jj] != 0;
        Aux[0][tid] = m ? 1.0f : 0.0f;
        Aux[1][tid] = (quirk && jj < Lk && p.kpad[(int64_t)b2 * Lk + jj] != 0) ? 1.0f : 0.0f;
      }
      __syncthreads();
      if (!active) continue;
      const int nb = (min(LC, Lk - jc) + 15) >> 4;
      for (int jb = 0; jb < nb; ++jb, ++unit) {
        if (spread && (unit & 7) != wave) continue;
        const int j0 = jb << 4;
        float kf[8], vf[8];
        loadN<8>(Xs + (j0 + jl) * BS + 8 * kq, kf);
        loadN<8>(Ys + (j0 + jl) * BS + 8 * kq, vf);
        f32x4 st = {0.f, 0.f, 0.f, 0.f}, dpt = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) {
          st = mfma16(kf[t], qf[t], st);
          dpt = mfma16(vf[t], gf[t], dpt);
        }
        const int jr = j0 + 4 * kq;
        float kb0[4], kb1[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) { kb0[r] = Xs[(jr + r) * BS + jl]; kb1[r] = Xs[(jr + r) * BS + 16 + jl]; }
        const float4 kp4 = *reinterpret_cast<const float4*>(&Aux[0][jr]);
        const float4 kq4 = *reinterpret_cast<const float4*>(&Aux[1][jr]);
        const float kp_[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
        const float kp2_[4] = {kq4.x, kq4.y, kq4.z, kq4.w};
        float ds[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mk = fmaf(qp_i, kp2_[r], kp_[r]);
          const float e = __expf(st[r] * scale - lse_i);
          const float pj = mk != 0.0f ? 0.0f : e;
          float km = 1.0f;
          if (DROP) km = mesm_hash32(rowi + (uint32_t)(jc + jr + r), drop_seed) >= thresh ? inv_keep : 0.0f;
          ds[r] = pj * (dpt[r] * km - dl_i) * scale;
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dQa[0] = mfma16(ds[r], kb0[r], dQa[0]);
          dQa[1] = mfma16(ds[r], kb1[r], dQa[1]);
        }
      }
    }
    if (spread) {
      for (int w = 0; w < BT / 64; ++w) {
        __syncthreads();
        if (wave == w) {
#pragma unroll
          for (int a = 0; a < 2; ++a) {
            float4* slot = reinterpret_cast<float4*>(Xs) + a * 64 + lane;
            if (w > 0) {
              const float4 t = *slot;
              dQa[a][0] += t.x; dQa[a][1] += t.y; dQa[a][2] += t.z; dQa[a][3] += t.w;
            }
            if (w + 1 < BT / 64) *slot = make_float4(dQa[a][0], dQa[a][1], dQa[a][2], dQa[a][3]);
          }
        }
      }
    }
    if (active && (!spread || wave == BT / 64 - 1)) {
      float* dqb = p.dq + (int64_t)b * p.q_bs + h * 32;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int ii = i0 + 4 * kq + r;
        if (ii < Lq) {
          ATTN_ST(dqb + ((int64_t)ii * p.q_ls + jl), dQa[0][r]);
          ATTN_ST(dqb + ((int64_t)ii * p.q_ls + 16 + jl), dQa[1][r]);
        }
      }
    }
  }
}

constexpr int BLK_GROUP_MAX = 8;
struct BlkGroup {
  MesmAttnArgs p[BLK_GROUP_MAX];
  int start[BLK_GROUP_MAX + 1];
  int n;
};

template <bool DROP>
__global__ __launch_bounds__(BT) void attn_blk_bwd_group_kernel(const BlkGroup g) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < BLK_GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const MesmAttnArgs p = *reinterpret_cast<const MesmAttnArgs*>(ka + offsetof(BlkGroup, p) + (size_t)gi * sizeof(MesmAttnArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(BlkGroup, start) + (size_t)gi * sizeof(int));
  if (DROP && p.drop_p > 0.f) attn_blk_bwd_body<32, true>(p, bid - first);
  else attn_blk_bwd_body<32, false>(p, bid - first);
}

// ------------------------------------------------------------------------------------------------------------
// Forward on the same blocks.  One workgroup per (batch, head): K and V staged once; a wave owns 16 queries (the
// QUERY on the lane: S^T = K Q^T), keeps the scores of ALL key blocks in registers (NJ x 4), so the row maximum and
// sum are in-lane loops plus two lane exchanges (lanes 16 / 32 apart hold the other keys of the row), and the
// normalised, dropped probabilities are the A operands of O = P V.  DK = 64: the decoder's split heads
// ([content || position] halves from q / q2 and k (+ k_add) / k2, transformer.py:778-784).
__device__ __forceinline__ float max_xor16(float v) {
  auto r = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float max_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}

template <int DK>
struct FwdShape {
  static constexpr int KS = DK + 4;  // K row stride
  int LkP;
  __host__ __device__ explicit FwdShape(int Lk) : LkP((Lk + 15) & ~15) {}
  __host__ __device__ size_t floats() const { return (size_t)LkP * (KS + BS) + 2 * LkP + 4; }
};

template <int DK, int NJ, bool DROP>
__device__ __forceinline__ void attn_blk_fwd_body(const MesmAttnArgs& p, const int bh) {
  constexpr int KS = DK + 4, DQ = DK / 4;  // DQ reduce indices per lane group
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const FwdShape<DK> sh(p.Lk);
  float* Ks = smem;
  float* Vs = Ks + sh.LkP * KS;
  float* Kp = Vs + sh.LkP * BS;
  float* Kp2 = Kp + sh.LkP;
  int* next = reinterpret_cast<int*>(Kp2 + sh.LkP);

  const int tid = threadIdx.x, lane = tid & 63, nthr = blockDim.x;
  const int b = bh / p.H, h = bh % p.H;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const int Lq = p.Lq, Lk = p.Lk;
  constexpr int DKH = DK / 2;
  const bool split = DK == 64 && p.q2 != nullptr;
  const int hq = h * (split ? DKH : DK);
  const float* kb = p.k + (int64_t)b * p.k_bs + hq;
  const float* kb2 = split ? p.k2 + (int64_t)b * p.k_bs + hq - DKH : kb;
  const float* kadd = (split && p.k_add) ? p.k_add + (int64_t)b * p.k_bs + hq : nullptr;
  const float* vb = p.v + (int64_t)b * p.v_bs + h * 32;

  for (int idx = tid; idx < sh.LkP * (DK / 4); idx += nthr) {
    const int r = idx / (DK / 4), c = (idx % (DK / 4)) * 4;
    float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Lk) {
      x = *reinterpret_cast<const float4*>((split && c >= DKH ? kb2 : kb) + (int64_t)r * p.k_ls + c);
      if (kadd && c < DKH) {
        const float4 y = *reinterpret_cast<const float4*>(kadd + (int64_t)r * p.k_ls + c);
        x.x += y.x; x.y += y.y; x.z += y.z; x.w += y.w;
      }
    }
    *reinterpret_cast<float4*>(Ks + r * KS + c) = x;
  }
  for (int idx = tid; idx < sh.LkP * 8; idx += nthr) {
    const int r = idx >> 3, c = (idx & 7) * 4;
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (r < Lk) v = *reinterpret_cast<const float4*>(vb + (int64_t)r * p.v_ls + c);
    *reinterpret_cast<float4*>(Vs + r * BS + c) = v;
  }
  for (int j = tid; j < sh.LkP; j += nthr) {
    bool m = j >= Lk;
    if (!m && p.kpad) m = p.kpad[(int64_t)b * Lk + j] != 0;
    Kp[j] = m ? 1.0f : 0.0f;
    Kp2[j] = (quirk && j < Lk && p.kpad[(int64_t)b2 * Lk + j] != 0) ? 1.0f : 0.0f;
  }
  if (tid == 0) *next = 0;
  __syncthreads();

  const uint32_t thresh = DROP ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const float scale = p.scale;
  const int nI = (Lq + 15) >> 4, nJ = sh.LkP >> 4;
  const int jl = lane & 15, kq = lane >> 4;
  const float* qb = p.q + (int64_t)b * p.q_bs + hq;
  const float* qb2 = split ? p.q2 + (int64_t)b * p.q_bs + hq - DKH : qb;
  float* ob = p.o + (int64_t)b * p.o_bs + h * 32;

  for (;;) {
    int u = 0;
    if (lane == 0) u = atomicAdd(next, 1);
    u = __builtin_amdgcn_readfirstlane(u);
    if (u >= nI) break;
    const int i0 = u << 4, i = i0 + jl;
    // the lane's share of its query row: reduce indices DQ kq .. DQ kq + DQ - 1 (split heads: lane groups 2, 3 = q2)
    float qf[DQ];
    {
      const float* src = (split && kq >= 2 ? qb2 : qb) + (int64_t)(i < Lq ? i : 0) * p.q_ls + DQ * kq;
#pragma unroll
      for (int t = 0; t < DQ; t += 4) {
        float4 x = make_float4(0.f, 0.f, 0.f, 0.f);
        if (i < Lq) x = *reinterpret_cast<const float4*>(src + t);
        qf[t] = x.x; qf[t + 1] = x.y; qf[t + 2] = x.z; qf[t + 3] = x.w;
      }
    }
    const float qp_i = (quirk && i < Lq && p.qpad[(int64_t)b2 * Lq + i] != 0) ? 1.0f : 0.0f;
    f32x4 st[NJ];
    float m = -INFINITY;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      st[jb] = (f32x4){0.f, 0.f, 0.f, 0.f};
      if (jb < nJ) {
        const float* krow = Ks + ((jb << 4) + jl) * KS + DQ * kq;
#pragma unroll
        for (int t = 0; t < DQ; t += 4) {
          const float4 kf = *reinterpret_cast<const float4*>(krow + t);
          st[jb] = mfma16(kf.x, qf[t], st[jb]);      // S^T[j0 + 4 kq + r][i0 + jl]
          st[jb] = mfma16(kf.y, qf[t + 1], st[jb]);
          st[jb] = mfma16(kf.z, qf[t + 2], st[jb]);
          st[jb] = mfma16(kf.w, qf[t + 3], st[jb]);
        }
        const int jr = (jb << 4) + 4 * kq;
        const float4 kp4 = *reinterpret_cast<const float4*>(Kp + jr);
        const float4 kq4 = *reinterpret_cast<const float4*>(Kp2 + jr);
        const float kp_[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
        const float kp2_[4] = {kq4.x, kq4.y, kq4.z, kq4.w};
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float mk = fmaf(qp_i, kp2_[r], kp_[r]);
          const float sv = mk != 0.0f ? -INFINITY : st[jb][r] * scale;
          st[jb][r] = sv;
          m = fmaxf(m, sv);
        }
      }
    }
    m = max_xor32(max_xor16(m));
    float l = 0.0f;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      if (jb < nJ) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float pj = (m == -INFINITY) ? 0.0f : __expf(st[jb][r] - m);
          st[jb][r] = pj;
          l += pj;
        }
      }
    }
    l = add_xor32(add_xor16(l));
    const float inv_l = 1.0f / l;
    f32x4 oa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    const uint32_t rowi = ((uint32_t)bh * (uint32_t)Lq + (uint32_t)i) * (uint32_t)Lk;
#pragma unroll
    for (int jb = 0; jb < NJ; ++jb) {
      if (jb < nJ) {
        const int jr = (jb << 4) + 4 * kq;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float pd = st[jb][r] * inv_l;
          if (DROP) pd = mesm_hash32(rowi + (uint32_t)(jr + r), drop_seed) >= thresh ? pd * inv_keep : 0.0f;
          const float* vrow = Vs + (jr + r) * BS + jl;
          oa[0] = mfma16(pd, vrow[0], oa[0]);    // O[i0 + 4 kq + r'][jl], [16 + jl]
          oa[1] = mfma16(pd, vrow[16], oa[1]);
        }
      }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = i0 + 4 * kq + r;
      if (ii < Lq) {
        ATTN_ST(ob + ((int64_t)ii * p.o_ls + jl), oa[0][r]);
        ATTN_ST(ob + ((int64_t)ii * p.o_ls + 16 + jl), oa[1][r]);
      }
    }
    if (kq == 0 && i < Lq && p.lse) p.lse[(int64_t)bh * Lq + i] = m + __logf(l);
  }
}

template <int DK, int NJ, bool DROP>
__global__ __launch_bounds__(512) void attn_blk_fwd_kernel(const MesmAttnArgs p) {
  attn_blk_fwd_body<DK, NJ, DROP>(p, blockIdx.x);
}

template <int NJ, bool DROP>
__global__ __launch_bounds__(512) void attn_blk_fwd_group_kernel(const BlkGroup g) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < BLK_GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const MesmAttnArgs p = *reinterpret_cast<const MesmAttnArgs*>(ka + offsetof(BlkGroup, p) + (size_t)gi * sizeof(MesmAttnArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(BlkGroup, start) + (size_t)gi * sizeof(int));
  if (DROP && p.drop_p > 0.f) attn_blk_fwd_body<32, NJ, true>(p, bid - first);
  else attn_blk_fwd_body<32, NJ, false>(p, bid - first);
}

// Forward for long key ranges (TACoS' 513 x 513 encoder): a workgroup owns 128 queries (a wave: 16, the query on the
// lane) and streams the keys through LDS in chunks of 64, TWICE -- pass 1 scores only (running row maximum and sum per
// lane, merged over the four lane groups at the end), pass 2 scores again, normalised dropped probabilities straight
// into O = P V.  24 matrix instructions per block pair, no rescaling of accumulators (their rows are not the lane's row).
template <bool DROP>
__global__ __launch_bounds__(BT) void attn_blk_fwd_long_kernel(const MesmAttnArgs p) {
  __shared__ __attribute__((aligned(16))) float Ks[LC * BS];
  __shared__ __attribute__((aligned(16))) float Vs[LC * BS];
  __shared__ __attribute__((aligned(16))) float Aux[2][LC];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int jl = lane & 15, kq = lane >> 4;
  const int Lq = p.Lq, Lk = p.Lk;
  const int nIc = (Lq + LW - 1) / LW;
  const int bh = blockIdx.x / nIc, part = blockIdx.x % nIc;
  const int b = bh / p.H, h = bh % p.H;
  const int b2 = mesm_quirk_row(p, b, h);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const float* qb = p.q + (int64_t)b * p.q_bs + h * 32;
  const float* kb = p.k + (int64_t)b * p.k_bs + h * 32;
  const float* vb = p.v + (int64_t)b * p.v_bs + h * 32;
  const uint32_t thresh = DROP ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const float scale = p.scale;
  const int i0 = part * LW + 16 * wave, i = i0 + jl;
  const bool active = i0 < Lq;
  float qf[8];
  {
    float4 a = make_float4(0.f, 0.f, 0.f, 0.f), c = a;
    if (active && i < Lq) {
      a = *reinterpret_cast<const float4*>(qb + (int64_t)i * p.q_ls + 8 * kq);
      c = *reinterpret_cast<const float4*>(qb + (int64_t)i * p.q_ls + 8 * kq + 4);
    }
    qf[0] = a.x; qf[1] = a.y; qf[2] = a.z; qf[3] = a.w; qf[4] = c.x; qf[5] = c.y; qf[6] = c.z; qf[7] = c.w;
  }
  const float qp_i = (quirk && active && i < Lq && p.qpad[(int64_t)b2 * Lq + i] != 0) ? 1.0f : 0.0f;
  float m_run = -INFINITY, l_run = 0.0f, inv_l = 0.0f;
  f32x4 oa[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
  const uint32_t rowi = ((uint32_t)bh * (uint32_t)Lq + (uint32_t)i) * (uint32_t)Lk;
  for (int pass = 0; pass < 2; ++pass) {
    for (int jc = 0; jc < Lk; jc += LC) {
      __syncthreads();
      for (int idx = tid; idx < LC * 8; idx += BT) {
        const int r = idx >> 3, c = (idx & 7) * 4, jj = jc + r;
        float4 k = make_float4(0.f, 0.f, 0.f, 0.f), v = k;
        if (jj < Lk) {
          k = *reinterpret_cast<const float4*>(kb + (int64_t)jj * p.k_ls + c);
          if (pass) v = *reinterpret_cast<const float4*>(vb + (int64_t)jj * p.v_ls + c);
        }
        *reinterpret_cast<float4*>(Ks + r * BS + c) = k;
        if (pass) *reinterpret_cast<float4*>(Vs + r * BS + c) = v;
      }
      if (tid < LC) {
        const int jj = jc + tid;
        bool m = jj >= Lk;
        if (!m && p.kpad) m = p.kpad[(int64_t)b * Lk + jj] != 0;
        Aux[0][tid] = m ? 1.0f : 0.0f;
        Aux[1][tid] = (quirk && jj < Lk && p.kpad[(int64_t)b2 * Lk + jj] != 0) ? 1.0f : 0.0f;
      }
      __syncthreads();
      if (!active) continue;
      const int nb = (min(LC, Lk - jc) + 15) >> 4;
      for (int jb = 0; jb < nb; ++jb) {
        const int j0 = jb << 4, jr = j0 + 4 * kq;
        float kf[8];
        loadN<8>(Ks + (j0 + jl) * BS + 8 * kq, kf);
        f32x4 st = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int t = 0; t < 8; ++t) st = mfma16(kf[t], qf[t], st);  // S^T[jc + j0 + 4 kq + r][i0 + jl]
        const float4 kp4 = *reinterpret_cast<const float4*>(&Aux[0][jr]);
        const float4 kq4 = *reinterpret_cast<const float4*>(&Aux[1][jr]);
        const float kp_[4] = {kp4.x, kp4.y, kp4.z, kp4.w};
        const float kp2_[4] = {kq4.x, kq4.y, kq4.z, kq4.w};
        float sv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) sv[r] = fmaf(qp_i, kp2_[r], kp_[r]) != 0.0f ? -INFINITY : st[r] * scale;
        if (pass == 0) {
          const float m_new = fmaxf(fmaxf(m_run, fmaxf(sv[0], sv[1])), fmaxf(sv[2], sv[3]));
          if (m_new != -INFINITY) {
            float acc = l_run * __expf(m_run - m_new);  // (m_run = -inf: l_run is 0, exp(-inf) = 0)
#pragma unroll
            for (int r = 0; r < 4; ++r) acc += __expf(sv[r] - m_new);
            l_run = acc;
            m_run = m_new;
          }
        } else {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            float pd = (m_run == -INFINITY) ? 0.0f : __expf(sv[r] - m_run) * inv_l;
            if (DROP) pd = mesm_hash32(rowi + (uint32_t)(jc + jr + r), drop_seed) >= thresh ? pd * inv_keep : 0.0f;
            const float* vrow = Vs + (jr + r) * BS + jl;
            oa[0] = mfma16(pd, vrow[0], oa[0]);
            oa[1] = mfma16(pd, vrow[16], oa[1]);
          }
        }
      }
    }
    if (pass == 0) {
      // merge the four lane groups' (max, sum) of the row
      const float M = max_xor32(max_xor16(m_run));
      const float part_l = (m_run == -INFINITY) ? 0.0f : l_run * __expf(m_run - M);
      const float L = add_xor32(add_xor16(part_l));
      m_run = M;
      l_run = L;
      inv_l = 1.0f / L;  // (a fully masked row: 0 * inf = NaN, like the reference's softmax over -inf)
    }
  }
  if (active) {
    float* ob = p.o + (int64_t)b * p.o_bs + h * 32;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int ii = i0 + 4 * kq + r;
      if (ii < Lq) {
        ATTN_ST(ob + ((int64_t)ii * p.o_ls + jl), oa[0][r]);
        ATTN_ST(ob + ((int64_t)ii * p.o_ls + 16 + jl), oa[1][r]);
      }
    }
    if (kq == 0 && i < Lq && p.lse) p.lse[(int64_t)bh * Lq + i] = m_run + __logf(l_run);
  }
}

template <int DK>
size_t fwd_lds_bytes(const MesmAttnArgs& a) { return FwdShape<DK>(a.Lk).floats() * sizeof(float); }
int fwd_waves(const MesmAttnArgs& a) {
  const int nI = (a.Lq + 15) / 16;
  return nI < 8 ? nI : 8;
}

size_t lds_bytes(const MesmAttnArgs& a) {
  return (a.dk == 64 ? BlkShape<64>(a.Lq, a.Lk).floats() : BlkShape<32>(a.Lq, a.Lk).floats()) * sizeof(float);
}

}  // namespace

// dk = dv = 32 packed heads (groupable), or dk = 64 / dv = 32 split or packed heads; the staged head within 64 KB of
// LDS (Lq + Lk up to ~220 at dk = 32)
bool mesm_attn_blk_bwd_ok(const MesmAttnArgs& a) {
  if (a.dv != 32 || a.mask_mode == MESM_MASK_CAUSAL || lds_bytes(a) > 64 * 1024) return false;
  if (a.dk == 32) return !a.q2 && !a.k2 && !a.k_add;
  return a.dk == 64 && ((a.q2 && a.k2 && a.dq2 && a.dk2) || (!a.q2 && !a.k2 && !a.k_add));
}
bool mesm_attn_blk_bwd_groupable(const MesmAttnArgs& a) { return mesm_attn_blk_bwd_ok(a) && a.dk == 32; }

int mesm_attn_blk_bwd(const MesmAttnArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.H)), block(BT);
  const size_t lds = lds_bytes(a);
  const bool drop = a.drop_p > 0.f;
  if (a.dk == 32) {
    if (drop) hipLaunchKernelGGL((attn_blk_bwd_kernel<32, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((attn_blk_bwd_kernel<32, false>), grid, block, lds, s, a);
  } else {
    if (drop) hipLaunchKernelGGL((attn_blk_bwd_kernel<64, true>), grid, block, lds, s, a);
    else hipLaunchKernelGGL((attn_blk_bwd_kernel<64, false>), grid, block, lds, s, a);
  }
  return mesm_launch_status();
}

int mesm_attn_blk_bwd_group(const MesmAttnArgs* list, int n, hipStream_t s) {
  if (n <= 0 || n > BLK_GROUP_MAX) return MESM_EINVAL;
  BlkGroup g;
  g.n = n;
  g.start[0] = 0;
  size_t lds = 0;
  bool any_drop = false;
  for (int i = 0; i < n; ++i) {
    any_drop = any_drop || list[i].drop_p > 0.f;
    g.p[i] = list[i];
    g.start[i + 1] = g.start[i] + list[i].B * list[i].H;
    const size_t need = lds_bytes(list[i]);
    lds = need > lds ? need : lds;
  }
  if (any_drop) hipLaunchKernelGGL(attn_blk_bwd_group_kernel<true>, dim3((unsigned)g.start[n]), dim3(BT), lds, s, g);
  else hipLaunchKernelGGL(attn_blk_bwd_group_kernel<false>, dim3((unsigned)g.start[n]), dim3(BT), lds, s, g);
  return mesm_launch_status();
}

// forward: dk = 32 packed heads or dk = 64 split heads, dv = 32, at most 128 keys (their scores stay in registers)
bool mesm_attn_blk_fwd_ok(const MesmAttnArgs& a) {
  if (a.dv != 32 || a.mask_mode == MESM_MASK_CAUSAL || a.Lk > 128) return false;
  if (a.dk == 32) return !a.q2 && !a.k2 && !a.k_add;
  // (one query block per head = one busy wave per workgroup: the lane-per-key kernel is quicker there,
  // 10 x 75 split heads 10.3 vs 11.6 us)
  return a.dk == 64 && a.Lq > 16 && ((a.q2 && a.k2) || (!a.q2 && !a.k2 && !a.k_add));
}
bool mesm_attn_blk_fwd_groupable(const MesmAttnArgs& a) { return mesm_attn_blk_fwd_ok(a) && a.dk == 32; }

#define BLK_FWD_NJ(KERNEL, NJV, ...)                                                                  \
  do {                                                                                                \
    switch (NJV) {                                                                                    \
      case 1: KERNEL(1, __VA_ARGS__); break;                                                          \
      case 2: KERNEL(2, __VA_ARGS__); break;                                                          \
      case 3: KERNEL(3, __VA_ARGS__); break;                                                          \
      case 4: KERNEL(4, __VA_ARGS__); break;                                                          \
      case 5: KERNEL(5, __VA_ARGS__); break;                                                          \
      case 6: KERNEL(6, __VA_ARGS__); break;                                                          \
      default: KERNEL(8, __VA_ARGS__); break;                                                         \
    }                                                                                                 \
  } while (0)

int mesm_attn_blk_fwd(const MesmAttnArgs& a, hipStream_t s) {
  const int nj = (a.Lk + 15) / 16;
  const dim3 grid((unsigned)(a.B * a.H)), block(64 * fwd_waves(a));
  const bool drop = a.drop_p > 0.f;
#define LAUNCH1(NJ, DK)                                                                                        \
  if (drop) hipLaunchKernelGGL((attn_blk_fwd_kernel<DK, NJ, true>), grid, block, fwd_lds_bytes<DK>(a), s, a); \
  else hipLaunchKernelGGL((attn_blk_fwd_kernel<DK, NJ, false>), grid, block, fwd_lds_bytes<DK>(a), s, a)
  if (a.dk == 32) BLK_FWD_NJ(LAUNCH1, nj, 32);
  else BLK_FWD_NJ(LAUNCH1, nj, 64);
#undef LAUNCH1
  return mesm_launch_status();
}

int mesm_attn_blk_fwd_group(const MesmAttnArgs* list, int n, hipStream_t s) {
  if (n <= 0 || n > BLK_GROUP_MAX) return MESM_EINVAL;
  BlkGroup g;
  g.n = n;
  g.start[0] = 0;
  size_t lds = 0;
  int nj = 1, waves = 1;
  bool drop = false;
  for (int i = 0; i < n; ++i) {
    g.p[i] = list[i];
    g.start[i + 1] = g.start[i] + list[i].B * list[i].H;
    const size_t need = fwd_lds_bytes<32>(list[i]);
    lds = need > lds ? need : lds;
    const int j = (list[i].Lk + 15) / 16;
    nj = j > nj ? j : nj;
    const int w = fwd_waves(list[i]);
    waves = w > waves ? w : waves;
    drop = drop || list[i].drop_p > 0.f;
  }
  const dim3 grid((unsigned)g.start[n]), block(64 * waves);
#define LAUNCHG(NJ, UNUSED)                                                                       \
  if (drop) hipLaunchKernelGGL((attn_blk_fwd_group_kernel<NJ, true>), grid, block, lds, s, g);     \
  else hipLaunchKernelGGL((attn_blk_fwd_group_kernel<NJ, false>), grid, block, lds, s, g)
  BLK_FWD_NJ(LAUNCHG, nj, 0);
#undef LAUNCHG
  return mesm_launch_status();
}

// dk = dv = 32 packed heads of any length (what mesm_attn_blk_bwd_ok() leaves because the head does not fit in LDS)
bool mesm_attn_blk_bwd_long_ok(const MesmAttnArgs& a) {
  return a.dk == 32 && a.dv == 32 && !a.q2 && !a.k2 && !a.k_add && a.mask_mode != MESM_MASK_CAUSAL;
}

int mesm_attn_blk_bwd_long(const MesmAttnArgs& a, hipStream_t s) {
  const int per = (a.Lk + LW - 1) / LW + (a.Lq + LW - 1) / LW;
  const dim3 grid((unsigned)(a.B * a.H * per)), block(BT);
  if (a.drop_p > 0.f) hipLaunchKernelGGL(attn_blk_bwd_long_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_blk_bwd_long_kernel<false>, grid, block, 0, s, a);
  return mesm_launch_status();
}

bool mesm_attn_blk_fwd_long_ok(const MesmAttnArgs& a) {
  return a.dk == 32 && a.dv == 32 && !a.q2 && !a.k2 && !a.k_add && a.mask_mode != MESM_MASK_CAUSAL;
}

int mesm_attn_blk_fwd_long(const MesmAttnArgs& a, hipStream_t s) {
  const dim3 grid((unsigned)(a.B * a.H * ((a.Lq + LW - 1) / LW))), block(BT);
  if (a.drop_p > 0.f) hipLaunchKernelGGL(attn_blk_fwd_long_kernel<true>, grid, block, 0, s, a);
  else hipLaunchKernelGGL(attn_blk_fwd_long_kernel<false>, grid, block, 0, s, a);
  return mesm_launch_status();
}

