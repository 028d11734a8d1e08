// Attention core on the matrix cores for the step's hot shapes: head dims dk = dv = 32 (T2V 75 x 33, encoder
// 76 x 76, MLM 33 x 75, SS 1..8 x 75 with all scores in registers; longer key ranges, e.g. TACoS' 513 x 513
// encoder, in a two-pass loop over key blocks, below), every mask rule and the dropout hash of
// attention.hip.  mesm_attn_fwd hands these shapes over (attention.hip: dispatch); split heads, the causal CLIP
// mode, other head dims, longer key ranges and the whole backward stay on the lane-per-key kernels.  (fp32 MFMA
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
__global__ __launch_bounds__(256) void attn_mfma_fwd_kernel(const MesmAttnArgs p) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 31, hf = lane >> 5;
  const int nqb = (p.Lq + 31) >> 5;
  const int item = blockIdx.x * 4 + wave;
  if (item >= p.B * p.H * nqb) return;
  const int bh = item / nqb, qb = item - bh * nqb;
  const int b = bh / p.H, hd = bh - b * p.H;
  const int q0 = qb * 32;
  const int i = q0 + li;
  const bool ivalid = i < p.Lq;
  const int mg = p.mask_group > 0 ? p.mask_group : p.B;
  const int b2 = (b / mg) * mg + ((b % mg) * p.H + hd) % mg;
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
  const int mg = p.mask_group > 0 ? p.mask_group : p.B;
  const int b2 = (b / mg) * mg + ((b % mg) * p.H + hd) % mg;
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
