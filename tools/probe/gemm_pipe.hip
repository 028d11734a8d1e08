// Software-pipelined form of the split-bf16 k-split 64 x 64 kernel (round 5).
//
// gemm_wstage64_kernel<.., 6> (gemm.hip) runs a 32-deep stage of a wave as PHASES that add up: wait for the LDS-DMA loads,
// read the fragments, split them into three bf16 terms (352 VALU instructions), then 48 matrix instructions -- ~4,000 cycles
// of which the matrix pipe is busy 1,536 (profiles/r4c/w64_probe.txt), and a second wave on the SIMD does not fill the gaps
// (profiles/r4e/w64_anti_probe.txt).  A matrix instruction leaves the SIMD's vector issue free for 24 of its 32 cycles
// (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"): the split of the NEXT operands can ride in those gaps if it is
// there in program order.  Here every stage is two phases of 24 matrix instructions:
//     phase A: products of k-step 0 of stage s   ||  split of k-step 1 of stage s, LDS reads of the fragments of stage s + 1
//     phase B: products of k-step 1 of stage s   ||  split of k-step 0 of stage s + 1, LDS-DMA issue of stage s + 3
// with the interleaving requested from the scheduler (sched_group_barrier: one matrix instruction, then its share of the
// vector / LDS / LDS-DMA instructions).  That needs the raw fragments of two stages and the split terms of two k-steps in
// registers (~320) and two slab sets per wave in LDS (128 KB per workgroup): ONE workgroup per CU, one wave per SIMD, with
// the 512-register budget that leaves -- the occupancy the step's 270-300-tile launches have on most CUs anyway.
// Same tile, same k-split over the four waves, same cross-wave reduction and staged epilogue as gemm_wstage64_kernel;
// three-term split only, no operand transforms.  MESM_GEMM_PIPE=1 selects it for the single-problem launches.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gemm_ws.hpp"

namespace {

struct SplitK {  // one 16-deep k-step of a 32-row fragment: three exact bf16 terms, 8 values each per lane
  u32x4 hi, mid, lo;
};

// v[s][j] = operand[outer][kb + 8 s + 4 h + j]; k-step t takes s = 2t, 2t + 1 (SplitFrag's slot map)
__device__ __forceinline__ SplitK split_k(const float (&v)[4][4], int t) {
  SplitK o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const float x0 = v[2 * t + (i >> 1)][2 * (i & 1)], x1 = v[2 * t + (i >> 1)][2 * (i & 1) + 1];
    const unsigned u0 = __float_as_uint(x0), u1 = __float_as_uint(x1);
    o.hi[i] = __builtin_amdgcn_perm(u1, u0, 0x07060302u);
    const float r0 = x0 - __uint_as_float(u0 & 0xFFFF0000u), r1 = x1 - __uint_as_float(u1 & 0xFFFF0000u);
    const unsigned m0 = __float_as_uint(r0), m1 = __float_as_uint(r1);
    o.mid[i] = __builtin_amdgcn_perm(m1, m0, 0x07060302u);
    const float q0 = r0 - __uint_as_float(m0 & 0xFFFF0000u), q1 = r1 - __uint_as_float(m1 & 0xFFFF0000u);
    o.lo[i] = __builtin_amdgcn_perm(__float_as_uint(q1), __float_as_uint(q0), 0x07060302u);
  }
  return o;
}

__device__ __forceinline__ f32x16 mma6(const SplitK& a, const SplitK& b, f32x16 acc) {
#define MESM_BF(x) __builtin_bit_cast(bf16x8, x)
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.lo), MESM_BF(b.hi), acc, 0, 0, 0);  // smallest terms first
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi), MESM_BF(b.lo), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid), MESM_BF(b.mid), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.mid), MESM_BF(b.hi), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi), MESM_BF(b.mid), acc, 0, 0, 0);
  acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(MESM_BF(a.hi), MESM_BF(b.hi), acc, 0, 0, 0);
#undef MESM_BF
  return acc;
}

// ---- hand-placed interleaving.  hipcc does not keep a source-level interleaving of pure vector arithmetic with matrix
// builtins: instruction selection linearises each block with the whole split in front of a run of 24 bare matrix
// instructions, sched_group_barrier requests were followed in some phases only and sched_barrier fences do not move what
// is already placed (ISA of the first two versions).  So the order is pinned the only way the compiler cannot undo: every
// matrix instruction and every half of a pair's split is an `asm volatile` statement (volatile asm statements keep their
// program order); gap g of a phase = matrix instruction g, then its share of the 32 split halves (5 and 6 vector
// instructions).  Accumulators live in AGPRs ("+a"), leaving the 256 architectural VGPRs to fragments and split terms.
// Hazards inside the asm are respected by construction: a split term is consumed by matrix instructions one phase (> 700
// cycles) after it is written and overwritten one phase after its last reader has issued.
template <int HALF>
__device__ __forceinline__ void split_half(const float (&v)[4][4], int t, int i, SplitK& o, float& r0, float& r1,
                                           unsigned sel) {
  if (HALF == 0) {
    const float x0 = v[2 * t + (i >> 1)][2 * (i & 1)], x1 = v[2 * t + (i >> 1)][2 * (i & 1) + 1];
    unsigned hi;
    asm volatile(
        "v_and_b32 %[r0], 0xffff0000, %[x0]\n\t"
        "v_and_b32 %[r1], 0xffff0000, %[x1]\n\t"
        "v_perm_b32 %[hi], %[x1], %[x0], %[sel]\n\t"
        "v_sub_f32 %[r0], %[x0], %[r0]\n\t"
        "v_sub_f32 %[r1], %[x1], %[r1]"
        : [r0] "=&v"(r0), [r1] "=&v"(r1), [hi] "=&v"(hi)
        : [x0] "v"(x0), [x1] "v"(x1), [sel] "s"(sel));
    o.hi[i] = hi;
  } else {
    unsigned mid, lo;
    float t0, t1;
    asm volatile(
        "v_perm_b32 %[mid], %[r1], %[r0], %[sel]\n\t"
        "v_and_b32 %[t0], 0xffff0000, %[r0]\n\t"
        "v_and_b32 %[t1], 0xffff0000, %[r1]\n\t"
        "v_sub_f32 %[t0], %[r0], %[t0]\n\t"
        "v_sub_f32 %[t1], %[r1], %[t1]\n\t"
        "v_perm_b32 %[lo], %[t1], %[t0], %[sel]"
        : [mid] "=&v"(mid), [lo] "=&v"(lo), [t0] "=&v"(t0), [t1] "=&v"(t1)
        : [r0] "v"(r0), [r1] "v"(r1), [sel] "s"(sel));
    o.mid[i] = mid;
    o.lo[i] = lo;
  }
}

// matrix instruction q (0..5) of the six products of (A block i, B block j), smallest terms first.  The four accumulators
// are a[0:15], a[16:31], a[32:47], a[48:63] BY NAME, outside the register allocator's view: as asm operands it kept them in
// different AGPR tuples on different paths of the stage loop and its v_accvgpr copies ran right behind matrix instructions
// it cannot see inside the asm -- no wait states, one stale register per tile.  Every statement declares all 64 clobbered,
// so nothing of the compiler's lives there across any of them; pipe_acc_zero / pipe_acc_read are the only way in and out.
#define PIPE_ACC_CLOBBER "a0", "a1", "a2", "a3", "a4", "a5", "a6", "a7", "a8", "a9", "a10", "a11", "a12", "a13", "a14", "a15", "a16", "a17", "a18", "a19", "a20", "a21", "a22", "a23", "a24", "a25", "a26", "a27", "a28", "a29", "a30", "a31", "a32", "a33", "a34", "a35", "a36", "a37", "a38", "a39", "a40", "a41", "a42", "a43", "a44", "a45", "a46", "a47", "a48", "a49", "a50", "a51", "a52", "a53", "a54", "a55", "a56", "a57", "a58", "a59", "a60", "a61", "a62", "a63"
__device__ __forceinline__ void mma_one(const SplitK& a, const SplitK& b, int c, int q) {
  const u32x4& x = q == 0 ? a.lo : ((q == 2 || q == 3) ? a.mid : a.hi);
  const u32x4& y = q == 1 ? b.lo : ((q == 2 || q == 4) ? b.mid : b.hi);
  if (c == 0) asm volatile("v_mfma_f32_32x32x16_bf16 a[0:15], %0, %1, a[0:15]" : : "v"(x), "v"(y) : PIPE_ACC_CLOBBER);
  else if (c == 1) asm volatile("v_mfma_f32_32x32x16_bf16 a[16:31], %0, %1, a[16:31]" : : "v"(x), "v"(y) : PIPE_ACC_CLOBBER);
  else if (c == 2) asm volatile("v_mfma_f32_32x32x16_bf16 a[32:47], %0, %1, a[32:47]" : : "v"(x), "v"(y) : PIPE_ACC_CLOBBER);
  else asm volatile("v_mfma_f32_32x32x16_bf16 a[48:63], %0, %1, a[48:63]" : : "v"(x), "v"(y) : PIPE_ACC_CLOBBER);
}

__device__ __forceinline__ void pipe_acc_zero() {
  const u32x4 z = {0u, 0u, 0u, 0u};
  asm volatile(
      "v_mfma_f32_32x32x16_bf16 a[0:15], %0, %0, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 a[16:31], %0, %0, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 a[32:47], %0, %0, 0\n\t"
      "v_mfma_f32_32x32x16_bf16 a[48:63], %0, %0, 0"
      : : "v"(z) : PIPE_ACC_CLOBBER);
}

// accumulator block c (16 registers) into VGPRs; the caller has let the last matrix instruction finish
template <int C>
__device__ __forceinline__ f32x16 pipe_acc_read() {
  f32x16 o;
#define PIPE_RD(r) asm volatile("v_accvgpr_read_b32 %0, a%1" : "=v"(o[r]) : "n"(16 * C + r) : PIPE_ACC_CLOBBER)
  PIPE_RD(0); PIPE_RD(1); PIPE_RD(2); PIPE_RD(3); PIPE_RD(4); PIPE_RD(5); PIPE_RD(6); PIPE_RD(7);
  PIPE_RD(8); PIPE_RD(9); PIPE_RD(10); PIPE_RD(11); PIPE_RD(12); PIPE_RD(13); PIPE_RD(14); PIPE_RD(15);
#undef PIPE_RD
  return o;
}

// slot s (4 values of the 32-deep stage) of one fragment
template <int LAYOUT>
__device__ __forceinline__ void ws_read_slot(const float* slab, int li, int h, float (&v)[4][4], int s) {
  if (LAYOUT == MESM_LAYOUT_REDUCE_CONTIG) {
    const int pos = (2 * s + h) ^ ((li >> 1) & 7);
    const float4 x = *reinterpret_cast<const float4*>(slab + li * 32 + pos * 4);
    v[s][0] = x.x; v[s][1] = x.y; v[s][2] = x.z; v[s][3] = x.w;
  } else {
#pragma unroll
    for (int j = 0; j < 4; ++j) v[s][j] = slab[(8 * s + 2 * j + h) * 32 + li];
  }
}

// One phase: the 24 matrix instructions of k-step T of (ma, mb) with, in their gaps, the split of k-step ST of the raw
// fragments (xa, xb) into (oa, ob) when SPLIT, and the reads of the next stage's fragments from `nbuf` into (na, nb) when READ.
template <int LA, int LB, bool SPLIT, bool READ>
__device__ __forceinline__ void pipe_phase(const SplitK (&ma)[2], const SplitK (&mb)[2],
                                           const float (&xa)[2][4][4], const float (&xb)[2][4][4], int st_k, SplitK (&oa)[2],
                                           SplitK (&ob)[2], const float* nbuf, float (&na)[2][4][4], float (&nb)[2][4][4],
                                           int li, int h) {
  float r0[16], r1[16];  // residuals of the 16 pairs between their two halves
  const unsigned sel = 0x07060302u;
#pragma unroll
  for (int g = 0; g < 24; ++g) {
    const int c = g / 6, q = g - 6 * c;
    mma_one(ma[c >> 1], mb[c & 1], c, q);
    if (SPLIT) {
#pragma unroll
      for (int ch = (g * 32) / 24; ch < ((g + 1) * 32) / 24; ++ch) {
        const int pr = ch >> 1, f = pr >> 2, i = pr & 3;  // pair, fragment (0: A rows 0-31, 1: B 0-31, 2: A 32-63, 3: B 32-63)
        if ((ch & 1) == 0) {
          if (f == 0) split_half<0>(xa[0], st_k, i, oa[0], r0[pr], r1[pr], sel);
          else if (f == 1) split_half<0>(xb[0], st_k, i, ob[0], r0[pr], r1[pr], sel);
          else if (f == 2) split_half<0>(xa[1], st_k, i, oa[1], r0[pr], r1[pr], sel);
          else split_half<0>(xb[1], st_k, i, ob[1], r0[pr], r1[pr], sel);
        } else {
          if (f == 0) split_half<1>(xa[0], st_k, i, oa[0], r0[pr], r1[pr], sel);
          else if (f == 1) split_half<1>(xb[0], st_k, i, ob[0], r0[pr], r1[pr], sel);
          else if (f == 2) split_half<1>(xa[1], st_k, i, oa[1], r0[pr], r1[pr], sel);
          else split_half<1>(xb[1], st_k, i, ob[1], r0[pr], r1[pr], sel);
        }
      }
    }
    if (READ && g < 16) {
      const int f = g >> 2, sl = g & 3;
      if (f == 0) ws_read_slot<LA>(nbuf, li, h, na[0], sl);
      else if (f == 1) ws_read_slot<LA>(nbuf + WS_SLAB, li, h, na[1], sl);
      else if (f == 2) ws_read_slot<LB>(nbuf + 2 * WS_SLAB, li, h, nb[0], sl);
      else ws_read_slot<LB>(nbuf + 3 * WS_SLAB, li, h, nb[1], sl);
    }
  }
}

struct PipeGeom {
  int m0, n0, k0, k1, nst;
};

template <int LA, int LB>
__device__ __forceinline__ void pipe_issue(const MesmGemmArgs& p, const PipeGeom& g, int st, float* mine, int lane) {
  float* buf = mine + (st & 1) * (4 * WS_SLAB);
  const int kb = g.k0 + 32 * st;
  ws_issue<LA>(p.A, p.lda, g.m0, p.M, kb, g.k1, buf, lane);
  ws_issue<LA>(p.A, p.lda, g.m0 + 32, p.M, kb, g.k1, buf + WS_SLAB, lane);
  ws_issue<LB>(p.B, p.ldb, g.n0, p.N, kb, g.k1, buf + 2 * WS_SLAB, lane);
  ws_issue<LB>(p.B, p.ldb, g.n0 + 32, p.N, kb, g.k1, buf + 3 * WS_SLAB, lane);
}

template <int LA, int LB>
__device__ __forceinline__ void pipe_read(const float* buf, int li, int h, float (&a)[2][4][4], float (&b)[2][4][4]) {
  ws_read<LA>(buf, li, h, a[0]);
  ws_read<LA>(buf + WS_SLAB, li, h, a[1]);
  ws_read<LB>(buf + 2 * WS_SLAB, li, h, b[0]);
  ws_read<LB>(buf + 3 * WS_SLAB, li, h, b[1]);
}

__device__ __forceinline__ void pipe_tail_mask(float (&a)[2][4][4], float (&b)[2][4][4], int kb, int k1, int h) {
  if (kb + 32 > k1) {  // partial last stage: reduce indices >= k1 contribute zeros
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
}

// One stage.  In: the raw fragments of the stage (ca, cb) and the split terms of its k-step 0 (s0a, s0b).  NEXT: the
// following stage's fragments are read into (na, nb) during phase A (its loads have landed: the caller waited) and the split
// terms of ITS k-step 0 replace s0a / s0b during phase B; the slabs that held it are refilled (stage st + 3) between the
// phases.
template <int LA, int LB, bool NEXT>
__device__ __forceinline__ void pipe_stage(const MesmGemmArgs& p, const PipeGeom& g, int st,
                                           const float (&ca)[2][4][4], const float (&cb)[2][4][4], float (&na)[2][4][4],
                                           float (&nb)[2][4][4], SplitK (&s0a)[2], SplitK (&s0b)[2], float* mine, int lane,
                                           float (&csum)[2], bool do_colsum) {
  const int li = lane & 31, h = lane >> 5;
  SplitK s1a[2], s1b[2];
  // ---- phase A: products of k-step 0 || split of k-step 1 || reads of the next stage's fragments
  pipe_phase<LA, LB, true, NEXT>(s0a, s0b, ca, cb, 1, s1a, s1b, mine + ((st + 1) & 1) * (4 * WS_SLAB), na, nb, li, h);
  if (do_colsum) {
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
        for (int j = 0; j < 4; ++j) csum[t] += ca[t][s_][j];
  }
  if (NEXT) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the next stage's fragments are in registers: its slabs are free
#ifndef MESM_PIPE_NO_DMA  // (probe build: wrong results -- what would the loop cost if another wave issued the LDS-DMA?)
    if (st + 3 < g.nst) pipe_issue<LA, LB>(p, g, st + 3, mine, lane);
#endif
    pipe_tail_mask(na, nb, g.k0 + 32 * (st + 1), g.k1, h);
  }
  // ---- phase B: products of k-step 1 || split of the next stage's k-step 0
  pipe_phase<LA, LB, NEXT, false>(s1a, s1b, na, nb, 0, s0a, s0b, mine, na, nb, li, h);
}

template <int LA, int LB>
__device__ __forceinline__ void wpipe_body(const MesmGemmArgs& p, const Blk blk, float* L) {
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  PipeGeom g;
  g.m0 = blk.x * 64;
  g.n0 = blk.y * 64;
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
  g.k0 = kbeg + wave * kw;
  g.k1 = g.k0 + kw < kend ? g.k0 + kw : kend;
  g.nst = g.k1 > g.k0 ? (g.k1 - g.k0 + 31) >> 5 : 0;
  const int nst = g.nst;
  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;
  float* mine = L + wave * (8 * WS_SLAB);  // two sets of A rows 0-31, A rows 32-63, B rows 0-31, B rows 32-63

  pipe_acc_zero();
  float csum[2] = {0.0f, 0.0f};
  const bool do_colsum = (p.colsum != nullptr) && (blk.y == 0);

  if (nst > 0) {
    pipe_issue<LA, LB>(p, g, 0, mine, lane);
    if (nst > 1) pipe_issue<LA, LB>(p, g, 1, mine, lane);
    if (nst > 1) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float ra[2][4][4], rb[2][4][4], qa[2][4][4], qb[2][4][4];
    pipe_read<LA, LB>(mine, li, h, ra, rb);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (nst > 2) pipe_issue<LA, LB>(p, g, 2, mine, lane);
    pipe_tail_mask(ra, rb, g.k0, g.k1, h);
    SplitK s0a[2], s0b[2];
    s0a[0] = split_k(ra[0], 0);
    s0a[1] = split_k(ra[1], 0);
    s0b[0] = split_k(rb[0], 0);
    s0b[1] = split_k(rb[1], 0);
    // stages in pairs (the raw fragments alternate between two register sets); before a stage reads the next one's
    // slabs, everything but the youngest 16 LDS-DMA instructions (the stage after that) has landed
    int st = 0;
    for (; st + 2 < nst; st += 2) {
      if (st + 2 < nst) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      pipe_stage<LA, LB, true>(p, g, st, ra, rb, qa, qb, s0a, s0b, mine, lane, csum, do_colsum);
      if (st + 3 < nst) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pipe_stage<LA, LB, true>(p, g, st + 1, qa, qb, ra, rb, s0a, s0b, mine, lane, csum, do_colsum);
    }
    // one or two stages left, the current one's fragments in (ra, rb)
    if (st + 1 < nst) {
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      pipe_stage<LA, LB, true>(p, g, st, ra, rb, qa, qb, s0a, s0b, mine, lane, csum, do_colsum);
      pipe_stage<LA, LB, false>(p, g, st + 1, qa, qb, ra, rb, s0a, s0b, mine, lane, csum, do_colsum);
    } else {
      pipe_stage<LA, LB, false>(p, g, st, ra, rb, qa, qb, s0a, s0b, mine, lane, csum, do_colsum);
    }
  }
  // (the compiler does not know what the asm statements were: the wait states between the last matrix instruction, 16
  // passes, and the first read of an accumulator are ours to provide)
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");

  f32x16 acc[2][2];
  acc[0][0] = pipe_acc_read<0>();
  acc[0][1] = pipe_acc_read<1>();
  acc[1][0] = pipe_acc_read<2>();
  acc[1][1] = pipe_acc_read<3>();
  XForm xa = {}, xb = {};
  if (do_colsum) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float c = add_xor32(csum[t]);
      const int gm = g.m0 + 32 * t + li;
      if (wave == 0 && blk.z == 0 && KM < p.K) c += tail_colsum<LA, false>(p, gm, KM, xa);
      if (h == 0 && gm < p.M && c != 0.0f) atomicAdd(p.colsum + gm, c);
    }
  }
  __syncthreads();  // every wave is done with its slabs: the reduction buffer aliases them
#pragma unroll
  for (int ti = 0; ti < 2; ++ti)
#pragma unroll
    for (int tj = 0; tj < 2; ++tj)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
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
  tile16_epilogue<LA, LB, false>(p, sum, g.m0 + 32 * (wave >> 1), g.n0 + 32 * (wave & 1), slope, seed_off, blk.z, L, blk.slot,
                                 KM, xa, xb);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

constexpr int PIPE_LDS = 4 * 8 * WS_SLAB * 4;  // bytes: 4 waves x 2 sets x 4 slabs

template <int LA, int LB>
__global__ __launch_bounds__(NTHREADS, 1) void gemm_wpipe_kernel(const MesmGemmArgs p, const SideRed sr) {
  side_reduce(sr);
  extern __shared__ __attribute__((aligned(16))) float L[];
  Blk blk;
  blk.slot = linear_block();
  xcd_tile_z((int)blk.slot, (p.M + 63) / 64, (p.N + 63) / 64, p.split_k, blk.x, blk.y, blk.z);
  wpipe_body<LA, LB>(p, blk, L);
}

template <int LA, int LB>
int launch_pipe_l(const MesmGemmArgs& a, const SideRed& sr, hipStream_t s) {
  static bool attr = false;
  if (!attr) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wpipe_kernel<LA, LB>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            PIPE_LDS) != hipSuccess)
      return MESM_ELAUNCH;
    attr = true;
  }
  dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), 1, a.split_k > 1 ? a.split_k : 1);
  hipLaunchKernelGGL((gemm_wpipe_kernel<LA, LB>), grid, dim3(NTHREADS), PIPE_LDS, s, a, sr);
  return mesm_launch_status();
}

}  // namespace

// single-problem launch (split-bf16, no operand transforms); the caller queues the slope-gradient reduction
int mesm_gemm_pipe_launch(const MesmGemmArgs& a, const void* side_red, hipStream_t s) {
  SideRed sr;
  memcpy(&sr, side_red, sizeof(sr));
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  if (a.a_layout == R && a.b_layout == R) return launch_pipe_l<R, R>(a, sr, s);
  if (a.a_layout == R && a.b_layout == O) return launch_pipe_l<R, O>(a, sr, s);
  if (a.a_layout == O && a.b_layout == O) return launch_pipe_l<O, O>(a, sr, s);
  return launch_pipe_l<O, R>(a, sr, s);
}
