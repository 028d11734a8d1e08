// Attention core on the matrix cores for the step's hot shapes: head dims dk = dv = 32 (T2V 75 x 33, encoder
// 76 x 76, MLM 33 x 75, SS 1..8 x 75 with all scores in registers; longer key ranges, e.g. TACoS' 513 x 513
// encoder, in a two-pass loop over key blocks, below), every mask rule and the dropout hash of
// attention.hip.  mesm_attn_fwd / mesm_attn_bwd hand these shapes over (attention.hip: dispatch); split heads, the
// causal CLIP mode, other head dims and the backward of short query ranges stay on the lane-per-key kernels.  (fp32 MFMA
// and fp32 VALU have the same peak on this chip: the matrix cores buy fewer instructions and no cross-lane
// traffic, not flops.  A matrix-core backward -- (query block, key block) pairs per wave, dQ / dK / dV summed in LDS
// -- measured 27 / 37 us at 75 x 33 / 76 x 76 with plain LDS stores (wrong sums), 44 / 78 us with ds_add_f32, 30 / 43 us
// with per-key-block barriers, 32 / 41 us with a wave per key block and private LDS slices (no atomics, no barriers
// in the loop), against 30 / 52 us for the lane-per-key backward.  The ISA shows why: ~2,000 VALU instructions per
// (query block, key block) pair -- 64-bit addresses of 68 loads, mask bits, exp, the three-multiply dropout hash,
// LDS transposition -- next to 80 MFMAs.  Dropped; profiles/r2n.)
//
// One WAVE per (batch, head, 32-query block), no LDS, no cross-wave traffic.  The trick is to compute the
// TRANSPOSED score block S^T = K Q^T with v_mfma_f32_32x32x2_f32 (A = K rows, B = Q rows): the accumulator then
// holds, in lane (i = lane % 32, half = lane / 32), the scores of query i against the 16 keys
// j(r, half) = 4 half + (r & 3) + 8 (r >> 2), r = 0..15 -- i.e. a query row lives in TWO lanes.  Hence
//   * masks, softmax max / sum, dropout, normalisation are per-lane register loops plus ONE exchange with lane ^ 32;
//   * the probabilities are ALREADY the A operand of O = P V: MFMA step r takes A[i][k = half] = P[i][j(r, half)]
//     = accumulator register r, and B[k = half][d = lane % 32] = V[j(r, half)][d], a coalesced 128-byte row read.
// (The lane-per-key kernel spent its time on ~17 cross-lane exchanges per query row and a P V sweep through LDS.)
// Reduce-index order inside a dot product is free, so the Q / K fragments take features c = 16 half + step:
// every lane reads 64 contiguous bytes of its row.
#include <cstddef>
#include "attention_mfma.hpp"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

__device__ __forceinline__ float max_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float sum_xor32(float v) {
  auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// key (or query) index inside a 32-block held by accumulator register r of lane half hf
__device__ __forceinline__ constexpr int acc_row(int r, int hf) { return 4 * hf + (r & 3) + 8 * (r >> 2); }

// 16 features [16 hf, 16 hf + 16) of one 32-wide head row
__device__ __forceinline__ void load_frag(const float* row, float (&f)[16]) {
#pragma unroll
  for (int c = 0; c < 16; c += 4) {
    const float4 x = *reinterpret_cast<const float4*>(row + c);
    f[c] = x.x; f[c + 1] = x.y; f[c + 2] = x.z; f[c + 3] = x.w;
  }
}

struct RowMasks {
  uint32_t k[4];   // bit j: key 32 kb + j is padded (or past Lk)
  uint32_t k2[4];  // the same for batch b2 (T2V quirk)
  bool qp;         // qpad[b2, i]
};

template <int NKB>
__device__ __forceinline__ void load_masks(const MesmAttnArgs& p, int b, int b2, int i, bool ivalid, int li, bool quirk,
                                           RowMasks& m) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int j = 32 * kb + li;
    bool kp = j >= p.Lk, kp2 = false;
    if (!kp && p.kpad) kp = p.kpad[(int64_t)b * p.Lk + j] != 0;
    if (j < p.Lk && quirk) kp2 = p.kpad[(int64_t)b2 * p.Lk + j] != 0;
    m.k[kb] = (uint32_t)__ballot(kp);
    m.k2[kb] = (uint32_t)__ballot(kp2);
  }
  m.qp = quirk && ivalid && p.qpad[(int64_t)b2 * p.Lq + i] != 0;
}

// S^T block kb: scores of this lane's query row against keys 32 kb + acc_row(r, hf), scaled and masked.
// A last block with at most 4 keys inside the sequence (Lk = 33: the sentence token + 32 words) is not worth 16
// MFMAs: its keys are registers 0..3 of the LOWER lane half, so each is one 16-feature partial dot product per
// lane plus the exchange with lane ^ 32; the upper half holds keys 4.. of the block, all past Lk.
template <int NKB>
__device__ __forceinline__ void score_blocks(const MesmAttnArgs& p, int b, int hd, int li, int hf, const float (&qf)[16],
                                             const RowMasks& m, f32x16 (&st)[NKB]) {
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int kvalid = p.Lk - 32 * kb;
    if (kb == NKB - 1 && kvalid <= 4) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (t >= kvalid) break;
        const float* krow = p.k + (int64_t)b * p.k_bs + (int64_t)(32 * kb + t) * p.k_ls + hd * 32 + 16 * hf;
        float kf[16];
        load_frag(krow, kf);
        float d = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) d += kf[c] * qf[c];
        acc[t] = sum_xor32(d);
      }
    } else {
      const int j = 32 * kb + li;
      const float* krow = p.k + (int64_t)b * p.k_bs + (int64_t)(j < p.Lk ? j : p.Lk - 1) * p.k_ls + hd * 32 + 16 * hf;
      float kf[16];
      load_frag(krow, kf);
#pragma unroll
      for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], acc, 0, 0, 0);
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = acc_row(r, hf);
      const bool masked = ((m.k[kb] >> jl) & 1u) || (m.qp && ((m.k2[kb] >> jl) & 1u));
      acc[r] = masked ? -INFINITY : acc[r] * p.scale;
    }
    st[kb] = acc;
  }
}

template <int NKB>
__device__ __forceinline__ void attn_mfma_fwd_body(const MesmAttnArgs& p, int blk) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, hf = lane >> 5;
  const int nqb = (p.Lq + 31) >> 5;
  const int item = blk * 4 + wave;
  if (item >= p.B * p.H * nqb) return;
  const int bh = item / nqb, qb = item - bh * nqb;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q0 = qb * 32;
  const int i = q0 + li;
  const bool ivalid = i < p.Lq;
  const int b2 = mesm_quirk_row(p, b, hd);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;

  float qf[16];
  load_frag(p.q + (int64_t)b * p.q_bs + (int64_t)(ivalid ? i : p.Lq - 1) * p.q_ls + hd * 32 + 16 * hf, qf);
  RowMasks m;
  load_masks<NKB>(p, b, b2, i, ivalid, li, quirk, m);

  // V rows of the P V steps, requested before the score MFMAs so that their latency is hidden behind them
  const float* vb = p.v + (int64_t)b * p.v_bs + hd * 32 + li;
  float vf[NKB][16];
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * kb + acc_row(r, hf);
      vf[kb][r] = j < p.Lk ? vb[(int64_t)j * p.v_ls] : 0.0f;
    }

  f32x16 st[NKB];
  score_blocks<NKB>(p, b, hd, li, hf, qf, m, st);

  // softmax over the row's keys: 16 NKB registers here, the other half of the row in lane ^ 32
  float mx = -INFINITY;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[kb][r]);
  mx = max_xor32(mx);
  float l = 0.0f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pj = (mx == -INFINITY) ? 0.0f : __expf(st[kb][r] - mx);
      st[kb][r] = pj;
      l += pj;
    }
  l = sum_xor32(l);
  // a row whose keys are all masked: l = 0, 0 * inf = NaN in every output feature, like the reference
  const float inv_l = 1.0f / l;
  const uint32_t thresh = p.drop_p > 0.f ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const uint32_t row_idx = (uint32_t)(((int64_t)bh * p.Lq + i) * p.Lk);
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float pd = st[kb][r];
      if (thresh) pd = mesm_dropout_apply(pd, row_idx + (uint32_t)(32 * kb + acc_row(r, hf)), drop_seed, thresh, inv_keep);
      st[kb][r] = pd * inv_l;
    }

  // O = P V: step r multiplies P[:, j(r, half)] (register r) with V rows j(r, 0), j(r, 1)
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.0f;
#pragma unroll
  for (int kb = 0; kb < NKB; ++kb) {
    const int kvalid = p.Lk - 32 * kb;  // keys of this block inside the sequence (wave-uniform)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (acc_row(r, 0) >= kvalid) continue;  // both halves' keys are past Lk (their p is 0)
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[kb][r], vf[kb][r], o, 0, 0, 0);
    }
  }

  float* ob = p.o + (int64_t)b * p.o_bs + hd * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int io = q0 + acc_row(r, hf);
    if (io < p.Lq) ob[(int64_t)io * p.o_ls] = o[r];
  }
  if (hf == 0 && ivalid && p.lse) p.lse[(int64_t)bh * p.Lq + i] = mx + __logf(l);
}

template <int NKB>
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const MesmAttnArgs p) {
  attn_mfma_fwd_body<NKB>(p, blockIdx.x);
}

// Grouped launch: up to ATTN_GROUP_MAX independent problems of this kernel's class in ONE launch (the lockstep
// chains of ops.py: enhance / SS-MESM / MLM attention of the same phase).  MAXNKB = the largest key-block count of
// the group (registers are sized for it); the problem is picked per workgroup, its key-block count per wave.
constexpr int ATTN_GROUP_MAX = 8;
struct AttnGroup {
  MesmAttnArgs p[ATTN_GROUP_MAX];
  int start[ATTN_GROUP_MAX + 1];
  int n;
};

template <int MAXNKB>
__global__ __launch_bounds__(256) void attn_mfma_fwd_group_kernel(const AttnGroup g) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < ATTN_GROUP_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  // wave-uniform dynamic offset into the kernarg segment (indexing the by-value struct would copy it to scratch)
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const MesmAttnArgs p = *reinterpret_cast<const MesmAttnArgs*>(ka + offsetof(AttnGroup, p) + (size_t)gi * sizeof(MesmAttnArgs));
  const int first = *reinterpret_cast<const int*>(ka + offsetof(AttnGroup, start) + (size_t)gi * sizeof(int));
  const int nkb = (p.Lk + 31) >> 5;
  if (nkb == 1) attn_mfma_fwd_body<1>(p, bid - first);
  else if (MAXNKB >= 2 && nkb == 2) attn_mfma_fwd_body<2>(p, bid - first);
  else if (MAXNKB >= 3 && nkb == 3) attn_mfma_fwd_body<3>(p, bid - first);
  else if (MAXNKB >= 4) attn_mfma_fwd_body<4>(p, bid - first);
}

// ------------------------------------------------------------------------------------------------
// Forward for LONG key ranges (Lk > 128: TACoS' 513-clip encoder, group videos of several segments): the key
// blocks are a run-time loop, one 32 x 32 score block in registers at a time, in TWO passes -- pass 1 finds the
// row maxima (scores only), pass 2 recomputes the scores, exponentiates against the final maximum, sums the row
// and accumulates the UN-normalised P V; the output is divided by the row sum at the end.  No running rescale of
// the output accumulator (its rows live in registers of other lanes than the row's statistics): the second
// score product is 16 MFMAs per block, cheaper than the exchanges.  The one exchange is the final 1 / l, through
// a 64-float LDS line per wave.
__global__ __launch_bounds__(256) void attn_mfma_fwd_long_kernel(const MesmAttnArgs p) {
  __shared__ float Inv[4][32];
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, hf = lane >> 5;
  const int nqb = (p.Lq + 31) >> 5, nkb = (p.Lk + 31) >> 5;
  const int item = blockIdx.x * 4 + wave;
  if (item >= p.B * p.H * nqb) return;
  const int bh = item / nqb, qb = item - bh * nqb;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q0 = qb * 32;
  const int i = q0 + li;
  const bool ivalid = i < p.Lq;
  const int b2 = mesm_quirk_row(p, b, hd);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const bool qp = quirk && ivalid && p.qpad[(int64_t)b2 * p.Lq + i] != 0;

  float qf[16];
  load_frag(p.q + (int64_t)b * p.q_bs + (int64_t)(ivalid ? i : p.Lq - 1) * p.q_ls + hd * 32 + 16 * hf, qf);
  const float* kbase = p.k + (int64_t)b * p.k_bs + hd * 32 + 16 * hf;
  const float* vb = p.v + (int64_t)b * p.v_bs + hd * 32 + li;

  // masked, scaled scores of key block kb (registers = keys acc_row(r, hf))
  auto scores = [&](int kb, f32x16& acc) {
    const int j = 32 * kb + li;
    float kf[16];
    load_frag(kbase + (int64_t)(j < p.Lk ? j : p.Lk - 1) * p.k_ls, kf);
    bool kp = j >= p.Lk, kp2 = false;
    if (!kp && p.kpad) kp = p.kpad[(int64_t)b * p.Lk + j] != 0;
    if (j < p.Lk && quirk) kp2 = p.kpad[(int64_t)b2 * p.Lk + j] != 0;
    const uint32_t mk = (uint32_t)__ballot(kp), mk2 = (uint32_t)__ballot(kp2);
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
#pragma unroll
    for (int s = 0; s < 16; ++s) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s], qf[s], acc, 0, 0, 0);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int jl = acc_row(r, hf);
      const bool masked = ((mk >> jl) & 1u) || (qp && ((mk2 >> jl) & 1u));
      acc[r] = masked ? -INFINITY : acc[r] * p.scale;
    }
  };

  float mx = -INFINITY;
  for (int kb = 0; kb < nkb; ++kb) {
    f32x16 st;
    scores(kb, st);
#pragma unroll
    for (int r = 0; r < 16; ++r) mx = fmaxf(mx, st[r]);
  }
  mx = max_xor32(mx);

  const uint32_t thresh = p.drop_p > 0.f ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const uint32_t row_idx = (uint32_t)(((int64_t)bh * p.Lq + i) * p.Lk);
  float l = 0.0f;
  f32x16 o;
#pragma unroll
  for (int r = 0; r < 16; ++r) o[r] = 0.0f;
  for (int kb = 0; kb < nkb; ++kb) {
    float vf[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int j = 32 * kb + acc_row(r, hf);
      vf[r] = j < p.Lk ? vb[(int64_t)j * p.v_ls] : 0.0f;
    }
    f32x16 st;
    scores(kb, st);
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float pj = (mx == -INFINITY) ? 0.0f : __expf(st[r] - mx);
      l += pj;
      st[r] = thresh ? mesm_dropout_apply(pj, row_idx + (uint32_t)(32 * kb + acc_row(r, hf)), drop_seed, thresh, inv_keep) : pj;
    }
    const int kvalid = p.Lk - 32 * kb;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      if (acc_row(r, 0) >= kvalid) continue;
      o = __builtin_amdgcn_mfma_f32_32x32x2f32(st[r], vf[r], o, 0, 0, 0);
    }
  }
  l = sum_xor32(l);
  // 1 / l of row i sits in lane i; the output accumulator holds rows acc_row(r, hf) in its registers
  if (hf == 0) Inv[wave][li] = 1.0f / l;  // (all keys masked: 1 / 0 = inf, 0 * inf = NaN like the reference)
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  float* ob = p.o + (int64_t)b * p.o_bs + hd * 32 + li;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int io = q0 + acc_row(r, hf);
    if (io < p.Lq) ob[(int64_t)io * p.o_ls] = o[r] * Inv[wave][acc_row(r, hf)];
  }
  if (hf == 0 && ivalid && p.lse) p.lse[(int64_t)bh * p.Lq + i] = mx + __logf(l);
}

// ------------------------------------------------------------------------------------------------
// Backward for MANY query rows against few keys (TACoS: 512 clips x 17 words, where the lane-per-key backward
// walks all 512 rows in one workgroup per head: 131 us).  One workgroup (4 waves) per (batch, head).  Wave w owns
// key block kb = w % nkb for the whole kernel -- its K / V fragments, mask bits and the dK / dV accumulators stay
// in registers -- and walks the 32-query blocks part, part + P, ... (part = w / nkb, P = 4 / nkb waves share a key
// block).  Nothing is accumulated in shared memory: every partial block is written ONCE into an LDS slice private
// to its producer (dQ blocks indexed by key block, dK / dV blocks by part when P > 1) and the workgroup adds the
// slices while it writes the three matrices out in full 128-byte rows -- deterministic, no atomics, no
// zero-initialised outputs.  Per (query block, key block):
//   S^T = K Q^T, dP^T = V dO^T                (transposed layout: lane = query i, registers = keys j(r, half))
//   p = exp(s - lse_i), keep-mask, ds = p (dp keep - delta_i) scale          (per-lane loops, like the forward)
//   dQ  = dS K          A = dS^T registers, B = K rows j(r, half) read as coalesced 128-byte rows
//   dV += P^T dO, dK += dS^T Q   need lane = key: the register block goes through a wave-private 32 x 33 LDS
//                                tile (conflict-free both ways) and comes back as the A operand; B = dO / Q rows.
// At the step's own shapes (75 x 33, 76 x 76, 33 x 75) this form measured 32 / 41 / 29 us against 30 / 52 / 29 us
// for the lane-per-key kernel (VALU-issue bound either way), so it is dispatched for long query ranges only.
__global__ __launch_bounds__(256) void attn_mfma_bwd_kernel(const MesmAttnArgs p) {
  __shared__ float Tr[4][32 * 33];  // per wave: P, then dS, on their way to the lane = key layout
  extern __shared__ float Sl[];     // slices of 32 x 32: dQ [kb][qb], then (P > 1) dK [part][kb], dV [part][kb]
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, hf = lane >> 5;
  const int bh = blockIdx.x;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int Lq = p.Lq, Lk = p.Lk;
  const int nqb = (Lq + 31) >> 5, nkb = (Lk + 31) >> 5;
  const int P = 4 / nkb;  // waves per key block (nkb <= 4)
  const int kb = wave % nkb, part = wave / nkb;
  const bool active = part < P;
  const int b2 = mesm_quirk_row(p, b, hd);
  const bool quirk = (p.mask_mode == MESM_MASK_T2V_QUIRK) && p.qpad && p.kpad;
  const uint32_t thresh = p.drop_p > 0.f ? mesm_drop_threshold(p.drop_p) : 0u;
  const float inv_keep = 1.0f / (1.0f - p.drop_p);
  const uint32_t drop_seed = p.drop_seed + (p.seed_offset ? *p.seed_offset : 0u);
  const float scale = p.scale;

  float* SlQ = Sl;
  float* SlK = Sl + nkb * nqb * 1024;
  float* SlV = SlK + P * nkb * 1024;

  // Addressing: ONE 64-bit base per tensor (batch and head folded in, wave-uniform) and 32-bit element offsets
  // built with 24-bit multiplies and adds -- row * stride products in 64 bits (v_mad_u64_u32, v_mul_lo_u32,
  // quarter rate) were ~350 of the ~1,800 vector instructions of a block pair.  mesm_attn_mfma_bwd_ok() checks
  // that a batch slice fits 2^24 elements per row stride and 2^31 in total.
  const int hcol = hd * 32;
  const float* qb_ = p.q + (int64_t)b * p.q_bs + hcol;
  const float* kb_ = p.k + (int64_t)b * p.k_bs + hcol;
  const float* vb_ = p.v + (int64_t)b * p.v_bs + hcol;
  const float* ob_ = p.o + (int64_t)b * p.o_bs + hcol;
  const float* dob_ = p.d_o + (int64_t)b * p.o_bs + hcol;
  const uint32_t qls = (uint32_t)p.q_ls, kls = (uint32_t)p.k_ls, vls = (uint32_t)p.v_ls, ols = (uint32_t)p.o_ls;
  float* tr = &Tr[wave][0];
  // LDS offsets of the transposition tile: constant per lane
  const int tw = (4 * hf) * 33 + li;   // + (r & 3) * 33 + (r >> 2) * 264 : write [key row][query col]
  const int trd = li * 33 + 4 * hf;    // + (r & 3) + 8 (r >> 2)          : read  [key = lane][query row]

  if (active) {
    const int k0 = 32 * kb;
    const int kvalid = Lk - k0;  // keys of the block inside the sequence
    const int j = k0 + li;
    const uint32_t jc = (uint32_t)(j < Lk ? j : Lk - 1);
    bool kp = j >= Lk, kp2 = false;
    if (!kp && p.kpad) kp = p.kpad[(uint32_t)(b * Lk + j)] != 0;
    if (j < Lk && quirk) kp2 = p.kpad[(uint32_t)(b2 * Lk + j)] != 0;
    const uint32_t mk = (uint32_t)__ballot(kp), mk2 = (uint32_t)__ballot(kp2);
    float kf[16], vf[16];
    load_frag(kb_ + (__umul24(jc, kls) + 16 * hf), kf);
    load_frag(vb_ + (__umul24(jc, vls) + 16 * hf), vf);
    // rows k0 + j(r, half) of K read as columns: B operand of dQ, the same for every query block
    float kc[16];
    {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // rows past Lk (a partial last key block) read row Lk - 1 and contribute zeros: no load leaves K
        const int jo = k0 + acc_row(r, hf);
        const float x = kb_[__umul24((uint32_t)(jo < Lk ? jo : Lk - 1), kls) + li];
        kc[r] = jo < Lk ? x : 0.0f;
      }
    }
    f32x16 dKa, dVa;
#pragma unroll
    for (int r = 0; r < 16; ++r) dKa[r] = dVa[r] = 0.0f;
    const bool any_mask = (mk | mk2) != 0u;  // wave-uniform

    for (int qb = part; qb < nqb; qb += P) {
      const int q0 = 32 * qb;
      const int i = q0 + li;
      const bool ivalid = i < Lq;
      const uint32_t ic = (uint32_t)(ivalid ? i : Lq - 1);
      const int qvalid = Lq - q0;
      f32x16 st, dpt;
      float delta;
      {
        float qf[16], gf[16], of[16];
        const uint32_t oo = __umul24(ic, ols) + 16 * hf;
        load_frag(qb_ + (__umul24(ic, qls) + 16 * hf), qf);
        load_frag(dob_ + oo, gf);
        load_frag(ob_ + oo, of);
#pragma unroll
        for (int r = 0; r < 16; ++r) st[r] = dpt[r] = 0.0f;
#pragma unroll
        for (int s_ = 0; s_ < 16; ++s_) {
          st = __builtin_amdgcn_mfma_f32_32x32x2f32(kf[s_], qf[s_], st, 0, 0, 0);
          dpt = __builtin_amdgcn_mfma_f32_32x32x2f32(vf[s_], gf[s_], dpt, 0, 0, 0);
        }
        delta = 0.0f;
#pragma unroll
        for (int c = 0; c < 16; ++c) delta += gf[c] * of[c];
        delta = sum_xor32(delta);
      }
      // rows q0 + j(r, half) of dO and Q read as columns (the B operands of dV, dK), requested now that the row
      // fragments are dead: their latency hides behind the exp / hash loop.  Rows past Lq read row Lq - 1 (their
      // A operands are zero).
      float gc[16], qc[16];
      {
        const int rb = q0 + 4 * hf;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int io = rb + (r & 3) + 8 * (r >> 2);
          io = io < Lq ? io : Lq - 1;
          gc[r] = dob_[__umul24((uint32_t)io, ols) + li];
          qc[r] = qb_[__umul24((uint32_t)io, qls) + li];
        }
      }
      {
        const float lse_i = ivalid ? p.lse[(uint32_t)(bh * Lq + i)] : 0.0f;
        const bool qp = quirk && ivalid && p.qpad[(uint32_t)(b2 * Lq + i)] != 0;
        // dropout index ((bh Lq + i) Lk + j) * golden, formed by additions from one product per block pair
        const uint32_t h0 = ((uint32_t)(bh * Lq + i) * (uint32_t)Lk + (uint32_t)(k0 + 4 * hf)) * 0x9E3779B9u + drop_seed;
        const bool row_off = !ivalid || false;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int jl = acc_row(r, hf);
          bool masked = row_off;
          if (any_mask) masked = masked || ((mk >> jl) & 1u) || (qp && ((mk2 >> jl) & 1u));
          const float pj = masked ? 0.0f : __expf(st[r] * scale - lse_i);
          float km = 1.0f;
          if (thresh) {
            uint32_t x = h0 + (uint32_t)((r & 3) + 8 * (r >> 2)) * 0x9E3779B9u;  // = mesm_hash32(idx, seed)
            x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
            km = x >= thresh ? inv_keep : 0.0f;
          }
          st[r] = pj * km;                              // what multiplied V in the forward
          dpt[r] = pj * (dpt[r] * km - delta) * scale;  // dS (scale folded in)
        }
      }
      // P on its way to the lane = key layout (the tile is free: the previous block's reads are complete)
#pragma unroll
      for (int r = 0; r < 16; ++r) tr[tw + (r & 3) * 33 + (r >> 2) * 264] = st[r];
      // dQ block of this key block: A = dS^T registers, B = K rows; stored once into slice [kb][qb]
      {
        f32x16 dQa;
#pragma unroll
        for (int r = 0; r < 16; ++r) dQa[r] = 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc_row(r, 0) >= kvalid) continue;  // both halves' keys are past Lk
          dQa = __builtin_amdgcn_mfma_f32_32x32x2f32(dpt[r], kc[r], dQa, 0, 0, 0);
        }
        float* sq = SlQ + (kb * nqb + qb) * 1024 + (4 * hf) * 32 + li;
#pragma unroll
        for (int r = 0; r < 16; ++r) sq[((r & 3) + 8 * (r >> 2)) * 32] = dQa[r];
      }
      {
        float pn[16];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) pn[r] = tr[trd + (r & 3) + 8 * (r >> 2)];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) tr[tw + (r & 3) * 33 + (r >> 2) * 264] = dpt[r];  // dS follows
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc_row(r, 0) >= qvalid) continue;  // both halves' query rows are past Lq
          dVa = __builtin_amdgcn_mfma_f32_32x32x2f32(pn[r], gc[r], dVa, 0, 0, 0);
        }
      }
      {
        float sn[16];
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) sn[r] = tr[trd + (r & 3) + 8 * (r >> 2)];
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          if (acc_row(r, 0) >= qvalid) continue;
          dKa = __builtin_amdgcn_mfma_f32_32x32x2f32(sn[r], qc[r], dKa, 0, 0, 0);
        }
      }
    }
    if (P == 1) {  // this wave alone owns the key block: straight to memory
      float* dkb = p.dk_ + (int64_t)b * p.k_bs + hcol + li;
      float* dvb = p.dv_ + (int64_t)b * p.v_bs + hcol + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int jo = k0 + acc_row(r, hf);
        if (jo < Lk) {
          dkb[__umul24((uint32_t)jo, kls)] = dKa[r];
          dvb[__umul24((uint32_t)jo, vls)] = dVa[r];
        }
      }
    } else {
      float* sk = SlK + (part * nkb + kb) * 1024 + li;
      float* sv = SlV + (part * nkb + kb) * 1024 + li;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        sk[acc_row(r, hf) * 32] = dKa[r];
        sv[acc_row(r, hf) * 32] = dVa[r];
      }
    }
  }
  __syncthreads();
  // write-out, one 128-byte row per 32 threads, adding the producers' slices
  {
    const int c = threadIdx.x & 31, r0 = threadIdx.x >> 5;
    float* dqb = p.dq + (int64_t)b * p.q_bs + hcol + c;
    for (int r = r0; r < Lq; r += 8) {
      float t = 0.0f;
      for (int kk = 0; kk < nkb; ++kk) t += SlQ[(kk * nqb) * 1024 + r * 32 + c];
      dqb[__umul24((uint32_t)r, qls)] = t;
    }
    if (P > 1) {
      float* dkb = p.dk_ + (int64_t)b * p.k_bs + hcol + c;
      float* dvb = p.dv_ + (int64_t)b * p.v_bs + hcol + c;
      for (int r = r0; r < Lk; r += 8) {
        float tk = 0.0f, tv = 0.0f;
        for (int pp = 0; pp < P; ++pp) {
          tk += SlK[pp * nkb * 1024 + r * 32 + c];
          tv += SlV[pp * nkb * 1024 + r * 32 + c];
        }
        dkb[__umul24((uint32_t)r, kls)] = tk;
        dvb[__umul24((uint32_t)r, vls)] = tv;
      }
    }
  }
}

}  // namespace

bool mesm_attn_mfma_ok(const MesmAttnArgs& a) {
  if (!(a.dk == 32 && a.dv == 32 && !a.q2 && !a.k2 && !a.k_add && a.mask_mode != MESM_MASK_CAUSAL)) return false;
  // long key ranges: the two-pass kernel walks the key blocks serially per wave, which pays off only with enough
  // query blocks to fill the chip (513 x 513: 398 -> 286 us; but 8 x 512: 25 -> 66 us, 16 x 200: 17 -> 31 us)
  return a.Lk <= 128 || a.Lq >= 128;
}

int mesm_attn_mfma_fwd(const MesmAttnArgs& a, hipStream_t s) {
  const int nqb = (a.Lq + 31) / 32;
  const long items = (long)a.B * a.H * nqb;
  dim3 grid((unsigned)((items + 3) / 4));
  const int nkb = (a.Lk + 31) / 32;
  if (nkb > 4) {
    hipLaunchKernelGGL(attn_mfma_fwd_long_kernel, grid, dim3(256), 0, s, a);
    return mesm_launch_status();
  }
  if (nkb == 1) hipLaunchKernelGGL(attn_mfma_fwd_kernel<1>, grid, dim3(256), 0, s, a);
  else if (nkb == 2) hipLaunchKernelGGL(attn_mfma_fwd_kernel<2>, grid, dim3(256), 0, s, a);
  else if (nkb == 3) hipLaunchKernelGGL(attn_mfma_fwd_kernel<3>, grid, dim3(256), 0, s, a);
  else hipLaunchKernelGGL(attn_mfma_fwd_kernel<4>, grid, dim3(256), 0, s, a);
  return mesm_launch_status();
}

// problems of the register-resident forward (at most 4 key blocks) may share a launch
bool mesm_attn_mfma_groupable(const MesmAttnArgs& a) { return mesm_attn_mfma_ok(a) && a.Lk <= 128; }

int mesm_attn_mfma_fwd_group(const MesmAttnArgs* list, int n, hipStream_t s) {
  AttnGroup g;
  g.n = n;
  g.start[0] = 0;
  int maxnkb = 1;
  for (int i = 0; i < n; ++i) {
    const MesmAttnArgs& a = list[i];
    const int nqb = (a.Lq + 31) / 32;
    const long items = (long)a.B * a.H * nqb;
    g.p[i] = a;
    g.start[i + 1] = g.start[i] + (int)((items + 3) / 4);
    const int nkb = (a.Lk + 31) / 32;
    maxnkb = nkb > maxnkb ? nkb : maxnkb;
  }
  dim3 grid((unsigned)g.start[n]);
  if (maxnkb == 1) hipLaunchKernelGGL(attn_mfma_fwd_group_kernel<1>, grid, dim3(256), 0, s, g);
  else if (maxnkb == 2) hipLaunchKernelGGL(attn_mfma_fwd_group_kernel<2>, grid, dim3(256), 0, s, g);
  else if (maxnkb == 3) hipLaunchKernelGGL(attn_mfma_fwd_group_kernel<3>, grid, dim3(256), 0, s, g);
  else hipLaunchKernelGGL(attn_mfma_fwd_group_kernel<4>, grid, dim3(256), 0, s, g);
  return mesm_launch_status();
}

#ifndef MESM_ATTN_BWD_MIN_LQ
#define MESM_ATTN_BWD_MIN_LQ 256
#endif
static size_t bwd_slices_bytes(const MesmAttnArgs& a) {
  const int nqb = (a.Lq + 31) / 32, nkb = (a.Lk + 31) / 32;
  const int P = 4 / nkb;
  return (size_t)(nkb * nqb + (P > 1 ? 2 * P * nkb : 0)) * 1024 * sizeof(float);
}

// long query ranges against at most 128 keys, as far as the partial blocks fit in LDS (160 KB - 17 KB static)
bool mesm_attn_mfma_bwd_ok(const MesmAttnArgs& a) {
  // 32-bit element offsets inside a batch slice, 24-bit row * stride products
  const int64_t lim24 = 1 << 24;
  const bool small = a.q_ls < lim24 && a.k_ls < lim24 && a.v_ls < lim24 && a.o_ls < lim24 && a.Lq < lim24 &&
                     (int64_t)a.Lq * a.q_ls < (1ll << 31) && (int64_t)a.Lq * a.o_ls < (1ll << 31) &&
                     (int64_t)a.Lk * a.k_ls < (1ll << 31) && (int64_t)a.Lk * a.v_ls < (1ll << 31) &&
                     (int64_t)a.B * a.H * a.Lq < (1ll << 31) && (int64_t)a.B * a.Lk < (1ll << 31) &&
                     (int64_t)a.B * a.Lq < (1ll << 31);
  return small && a.dk == 32 && a.dv == 32 && !a.q2 && !a.k2 && !a.k_add && a.mask_mode != MESM_MASK_CAUSAL &&
         a.Lk <= 128 && a.Lq >= MESM_ATTN_BWD_MIN_LQ && bwd_slices_bytes(a) <= 140 * 1024;
}

int mesm_attn_mfma_bwd(const MesmAttnArgs& a, hipStream_t s) {
  const size_t lds = bwd_slices_bytes(a);
  // dynamic LDS beyond 64 KB has to be allowed once per function AND per device (the attribute lives with the
  // device's copy of the code object); the flags are only ever set, so a race between threads repeats the call
  static bool raised[64] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return MESM_ELAUNCH;
  if (dev < 0 || dev >= 64 || !raised[dev]) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(attn_mfma_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            140 * 1024) != hipSuccess)
      return MESM_ELAUNCH;
    if (dev >= 0 && dev < 64) raised[dev] = true;
  }
  hipLaunchKernelGGL(attn_mfma_bwd_kernel, dim3((unsigned)(a.B * a.H)), dim3(256), lds, s, a);
  return mesm_launch_status();
}
