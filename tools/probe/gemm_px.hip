// Split-bf16 "plane" GEMMs for gfx950: C = epi(A B) with every f32 operand value held as THREE bf16 planes
//     x = hi + mid + lo        (8 significant bits each, by truncation: x - hi and x - hi - mid are exact f32 subtractions)
// written ONCE by whoever produced the operand (mesm_split_planes, a GEMM / LayerNorm epilogue, the once-per-step weight
// split) and the six significant cross products
//     hi*hi + hi*mid + mid*hi + mid*mid + hi*lo + lo*hi       (dropped: mid*lo, lo*mid, lo*lo ~ 2^-24 |a||b|)
// accumulated in f32 on v_mfma_f32_32x32x16_bf16 (16x the f32 MFMA rate, so 6 products cost 3/8 of the exact-f32
// instruction time).  Measured accuracy (tools/px_check.py): max relative error against fp64 <= the exact-f32 kernel's on
// every shape of the step's census -- the f32 product's own rounding is 2^-24 too.
//
// Replaces the >= 1 GFLOP nn.Linear products of the step and their two backward GEMMs
// (/root/reference/model/transformer.py:537, 603-608, 647, 794; model.py:427-434); everything smaller stays on the exact-f32
// kernels of gemm.hip.  Epilogues are gemm.hip's (staged_epilogue over MesmGemmArgs).
//
// Plane layout (MesmPlanes): bf16, COLUMN-BLOCK major -- [cols_pad / 16][rows_pad][16]: element (r, c) of the logical
// tensor at ((c / 16) * rows_pad + r) * 16 + c % 16; both extents padded to a multiple of 32 with ZEROS (so no reduce-index
// tail exists in the kernels).  A stage of the kernels is 16 reduce indices (one v_mfma_f32_32x32x16_bf16 step); in this
// layout both uses of a tensor fetch it in whole 128-byte lines:
//   "R" (reduce index = column: forward x W^T, dX = dY W): 16 columns of 64 rows are 2 KB contiguous;
//   "O" (reduce index = row: the weight gradients dW = dY^T x, both operands activations stored [reduce][outer]): 16 rows of
//       64 columns are four contiguous 512-byte pieces; the MFMA fragment (8 consecutive reduce indices of one outer index
//       per lane) comes out of LDS through ds_read_b64_tr_b16, gfx950's transposing read.
// (Row-major planes were measured first: a 16- or 32-deep stage then takes 32 or 64 bytes out of every 128-byte line, the
// 4 waves x 2 operands x 3 planes of a CU touch 6x the L1's size per stage, and every line came out of L2 four / two times.)
//
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#include "gemm_common.hpp"

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int R_ = MESM_LAYOUT_REDUCE_CONTIG, O_ = MESM_LAYOUT_OUTER_CONTIG;

// ---- staging -------------------------------------------------------------------------------------------------------
// A STAGE is 16 reduce indices (one v_mfma_f32_32x32x16_bf16 step) of the wave's operands, three planes each:
//   R operand of NB outer blocks: image [32 NB outer rows][16 k] per plane = 32-byte rows (2 chunks of 16 bytes), NB KB;
//   O operand of 2 outer blocks:  image [16 k rows][64 outer] per plane = 128-byte rows (8 chunks), 2 KB.
// LDS-DMA writes lane-linear (wave-uniform LDS base + lane * 16 bytes), so the bank-conflict-free images are made by
// permuting the per-lane SOURCE address:
//   R: chunk c of row r sits at position c ^ ((r >> 3) & 1)   (the 16 rows a ds_read_b128 lane group reads at one chunk
//      then fall on 16 different 16-byte bank slots);
//   O: chunk ch of k row k sits at position ch ^ (((k >> 1) & 1) << 2)   (the 4 k rows x 64 bytes that a 32-lane half of
//      ds_read_b64_tr_b16 reads then cover the 64 banks exactly once).
// Loads are inline asm in the SGPR-base + 32-bit-VGPR-offset form: the per-lane offsets are computed once, a stage only
// advances three scalar plane bases; and hipcc's own vmcnt bookkeeping (it drains every LDS-DMA it knows about before any
// LDS read) stays out of the ring -- the only waits on it are the counted ones in the main loop.
__device__ __forceinline__ void px_glds(const void* sbase, unsigned voff, unsigned lds_byte_addr) {
#ifdef MESM_PX_NO_LOAD  // probe build: no LDS-DMA at all (the ring holds whatever LDS held)
  return;
#endif
  unsigned keep;
  asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
               : "=&s"(keep)
               : "v"(voff), "s"(sbase), "s"(lds_byte_addr)
               : "memory");
}

template <int LAYOUT, int NB>
struct PxSrc {
  static constexpr int NI = LAYOUT == R_ ? NB : 2;        // LDS-DMA instructions per plane and stage
  static constexpr int PLANE = LAYOUT == R_ ? NB * 1024 : 2048;  // bytes per plane and stage
  static_assert(LAYOUT == R_ || NB == 2, "outer-contiguous operands are staged 64 outer indices wide");
  unsigned voff[NI];       // per-lane byte offset from the plane's stage base
  const char* base[3];     // wave-uniform: plane + this wave's first reduce index
  int64_t kstep;           // bytes per stage
  __device__ __forceinline__ void init(const MesmPlanes& P, int o0, int k0, int lane) {
    // P.ld = elements between column blocks (rows_pad * 16)
#pragma unroll
    for (int q = 0; q < NI; ++q) {
      if (LAYOUT == R_) {
        const int row = 32 * q + (lane >> 1), pos = lane & 1;
        const int c = pos ^ ((row >> 3) & 1);
        int ro = o0 + row;
        ro = ro < P.rows ? ro : P.rows - 1;
        voff[q] = (unsigned)(ro * 16 + 8 * c) * 2u;
      } else {
        const int krow = 8 * q + (lane >> 3), pos = lane & 7;
        const int ch = pos ^ (((krow >> 1) & 1) << 2);
        int cb = (o0 >> 4) + (ch >> 1);
        const int ncb = P.cols >> 4;
        cb = cb < ncb ? cb : ncb - 1;
        voff[q] = (unsigned)(cb * (int)P.ld + krow * 16 + (ch & 1) * 8) * 2u;
      }
    }
    const int64_t first = LAYOUT == R_ ? (int64_t)(k0 >> 4) * P.ld : (int64_t)k0 * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) base[pl] = reinterpret_cast<const char*>(P.p[pl] + first);
    kstep = LAYOUT == R_ ? 2 * P.ld : 512;
  }
  // stage `st` of this wave -> LDS byte address `lds` (wave-uniform)
  __device__ __forceinline__ void issue(int st, unsigned lds) const {
    const int64_t adv = (int64_t)st * kstep;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const char* b = base[pl] + adv;
#pragma unroll
      for (int q = 0; q < NI; ++q) px_glds(b, voff[q], lds + pl * PLANE + q * 1024);
    }
  }
  // instruction `idx` (compile-time, 0 .. 3 NI - 1) of stage byte advance `adv`
  template <int IDX>
  __device__ __forceinline__ void issue_one(int64_t adv, unsigned lds) const {
    constexpr int pl = IDX / NI, q = IDX % NI;
    px_glds(base[pl] + adv, voff[q], lds + pl * PLANE + q * 1024);
  }
};

// ---- fragments -----------------------------------------------------------------------------------------------------
// lane l = (li = l & 31, h = l >> 5) of v_mfma_f32_32x32x16_bf16 holds operand[outer = li][k = 8 h + j], j = 0..7.
// f[t][pl]: outer block t, plane pl, of one stage.
template <int LAYOUT, int NB>
struct PxFrag {
  static constexpr int PLANE = PxSrc<LAYOUT, NB>::PLANE;
  u32x4 f[NB][3];
  static constexpr int NF = 3 * NB;  // fragments per stage
  // per-lane byte offset inside a plane's stage image (outer block 0)
  __device__ __forceinline__ static int lane_off(int lane) {
    if (LAYOUT == R_) {
      const int li = lane & 31, h = lane >> 5;
      return li * 32 + ((h ^ ((li >> 3) & 1)) << 4);
    }
    const int g4 = lane >> 4, i = lane & 15, qrow = i >> 2, pp = i & 3, h = g4 >> 1;
    const int b = (qrow >> 1) & 1;
    const int c2 = 2 * (g4 & 1) + (pp >> 1);
    return 128 * (8 * h + qrow) + 16 * (4 * b + c2) + 8 * (pp & 1);
  }
  // fragment IDX (compile-time: plane IDX / NB, outer block IDX % NB) of the stage image at `lds`
  template <int IDX>
  __device__ __forceinline__ void read_one(const char* lds, int a0) {
    constexpr int pl = IDX / NB, t = IDX % NB;
    if (LAYOUT == R_) {
      f[t][pl] = *reinterpret_cast<const u32x4*>(lds + pl * PLANE + t * 1024 + a0);
    } else {
      const char* a = lds + pl * PLANE + (t ? (a0 ^ 64) : a0);
      const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
      const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 512));
      const u32x2 l2 = __builtin_bit_cast(u32x2, lo4), h2 = __builtin_bit_cast(u32x2, hi4);
      f[t][pl] = u32x4{l2[0], l2[1], h2[0], h2[1]};
    }
  }
  __device__ __forceinline__ void read(const char* lds, int lane) {
    if (LAYOUT == R_) {
      const int li = lane & 31, h = lane >> 5;
      const int a0 = li * 32 + ((h ^ ((li >> 3) & 1)) << 4);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < NB; ++t) f[t][pl] = *reinterpret_cast<const u32x4*>(lds + pl * PLANE + t * 1024 + a0);
    } else {
      // a 16-lane group reads a block of 4 k rows x 16 outer columns: lane 4 qrow + pp of the group supplies the
      // address of row qrow, columns 4 pp .. 4 pp + 3, and receives column (lane & 15) of the 4 rows
      const int g4 = lane >> 4, i = lane & 15, qrow = i >> 2, pp = i & 3, h = g4 >> 1;
      const int b = (qrow >> 1) & 1;
      const int c2 = 2 * (g4 & 1) + (pp >> 1);
      const int a0 = 128 * (8 * h + qrow) + 16 * (4 * b + c2) + 8 * (pp & 1);  // outer block 0; block 1 = address ^ 64
#pragma unroll
      for (int pl = 0; pl < 3; ++pl)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
          const char* a = lds + pl * PLANE + (t ? (a0 ^ 64) : a0);
          const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
          const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 512));
          const u32x2 l2 = __builtin_bit_cast(u32x2, lo4), h2 = __builtin_bit_cast(u32x2, hi4);
          f[t][pl] = u32x4{l2[0], l2[1], h2[0], h2[1]};
        }
    }
  }
  // hi + mid + lo summed over the 8 reduce indices this lane holds of outer block t
  __device__ __forceinline__ float ksum(int t) const {
    float s_ = 0.0f;
#pragma unroll
    for (int pl = 2; pl >= 0; --pl)
#pragma unroll
      for (int w = 0; w < 4; ++w) {
        const unsigned u = f[t][pl][w];
        s_ += __uint_as_float(u << 16) + __uint_as_float(u & 0xFFFF0000u);
      }
    return s_;
  }
};

#define PX_BF(x) __builtin_bit_cast(bf16x8, x)
// the six products of one accumulator block
template <int LA, int LB, int TM>
__device__ __forceinline__ void px_mma_block(const PxFrag<LA, TM>& a, const PxFrag<LB, 2>& b, f32x16& c, int ti, int tj) {
#ifdef MESM_PX_NO_MMA  // probe build (tools/build_variant.sh): keeps the fragments alive, issues no matrix instruction
  asm volatile("" ::"v"(a.f[ti][0]), "v"(a.f[ti][1]), "v"(a.f[ti][2]), "v"(b.f[tj][0]), "v"(b.f[tj][1]), "v"(b.f[tj][2]));
  return;
#endif
  // smallest terms first
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][2]), PX_BF(b.f[tj][0]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][0]), PX_BF(b.f[tj][2]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][1]), PX_BF(b.f[tj][1]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][1]), PX_BF(b.f[tj][0]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][0]), PX_BF(b.f[tj][1]), c, 0, 0, 0);
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(a.f[ti][0]), PX_BF(b.f[tj][0]), c, 0, 0, 0);
}
#undef PX_BF

template <int N>
__device__ __forceinline__ void px_wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// LDS-DMA instructions [LO, HI) of a stage's flat list (A's 3 NI_A, then B's 3 NI_B), unrolled at compile time
template <int LO, int HI, typename SrcA, typename SrcB>
__device__ __forceinline__ void px_issue_range(const SrcA& sa, const SrcB& sb, int64_t adv_a, int64_t adv_b, unsigned slot,
                                               unsigned a_bytes) {
  if constexpr (LO < HI) {
    constexpr int NA = 3 * SrcA::NI;
    if constexpr (LO < NA) sa.template issue_one<LO>(adv_a, slot);
    else sb.template issue_one<LO - NA>(adv_b, slot + a_bytes);
    px_issue_range<LO + 1, HI>(sa, sb, adv_a, adv_b, slot, a_bytes);
  }
}

// accumulator blocks B0 .. NBLK - 1 of one stage; after block b the LDS-DMA instructions [b G / NBLK, (b + 1) G / NBLK) of
// the refill are issued (when `refill`): the ~5 scalar + 1 vector-memory instruction of an LDS-DMA issue ride in the
// shadow of the matrix instructions instead of standing in front of them
template <int B0, int NBLK, int G, bool REFILL, int LA, int LB, int TM, typename SrcA, typename SrcB>
__device__ __forceinline__ void px_stage(const PxFrag<LA, TM>& fa, const PxFrag<LB, 2>& fb, f32x16 (&acc)[TM][2],
                                         const SrcA& sa, const SrcB& sb, int64_t adv_a, int64_t adv_b, unsigned slot,
                                         unsigned a_bytes) {
  if constexpr (B0 < NBLK) {
    px_mma_block<LA, LB, TM>(fa, fb, acc[B0 / 2][B0 % 2], B0 / 2, B0 % 2);
    if constexpr (REFILL) px_issue_range<B0 * G / NBLK, (B0 + 1) * G / NBLK>(sa, sb, adv_a, adv_b, slot, a_bytes);
    px_stage<B0 + 1, NBLK, G, REFILL, LA, LB, TM>(fa, fb, acc, sa, sb, adv_a, adv_b, slot, a_bytes);
  }
}

// fragments [LO, HI) of a stage's flat list (A's 3 TM, then B's 6) into the NEXT register set
template <int LO, int HI, typename FA, typename FB>
__device__ __forceinline__ void px_read_range(FA& na, FB& nb, const char* slot_a, const char* slot_b, int a0a, int a0b) {
  if constexpr (LO < HI) {
    if constexpr (LO < FA::NF) na.template read_one<LO>(slot_a, a0a);
    else nb.template read_one<LO - FA::NF>(slot_b, a0b);
    px_read_range<LO + 1, HI>(na, nb, slot_a, slot_b, a0a, a0b);
  }
}

// One iteration of the register-double-buffered loop (RING = 2): the matrix instructions of the CURRENT stage (fragments
// already in registers) with, in their shadow, first the fragment reads of the NEXT stage into the other register set
// (blocks 0 .. H - 1), then -- those reads complete, the slot is free -- the LDS-DMA refill of that slot with the stage
// after the two in flight (blocks H .. NBLK - 1).  sched_barrier(0) after every block pins the interleave.
template <int B0, int NBLK, int G, bool READ, bool REFILL, int LA, int LB, int TM, typename SrcA, typename SrcB>
__device__ __forceinline__ void px_stage2(const PxFrag<LA, TM>& ca, const PxFrag<LB, 2>& cb, PxFrag<LA, TM>& na, PxFrag<LB, 2>& nb,
                                          f32x16 (&acc)[TM][2], const char* rslot_a, const char* rslot_b, int a0a, int a0b,
                                          const SrcA& sa, const SrcB& sb, int64_t adv_a, int64_t adv_b, unsigned wslot,
                                          unsigned a_bytes) {
  if constexpr (B0 < NBLK) {
    constexpr int H = NBLK / 2, NR = 3 * TM + 6;
    px_mma_block<LA, LB, TM>(ca, cb, acc[B0 / 2][B0 % 2], B0 / 2, B0 % 2);
    if constexpr (READ && B0 < H) px_read_range<B0 * NR / H, (B0 + 1) * NR / H>(na, nb, rslot_a, rslot_b, a0a, a0b);
    if constexpr (REFILL && B0 == H - 1) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if constexpr (REFILL && B0 >= H)
      px_issue_range<(B0 - H) * G / (NBLK - H), (B0 - H + 1) * G / (NBLK - H)>(sa, sb, adv_a, adv_b, wslot, a_bytes);
    __builtin_amdgcn_sched_barrier(0);
    px_stage2<B0 + 1, NBLK, G, READ, REFILL, LA, LB, TM>(ca, cb, na, nb, acc, rslot_a, rslot_b, a0a, a0b, sa, sb, adv_a, adv_b,
                                                          wslot, a_bytes);
  }
}

// Kernel "px": (32 TM) x 64 output tile per workgroup (TM = 2: 64 x 64; TM = 3: 96 x 64, reduce-contiguous A only), 4 waves,
// each wave the WHOLE tile over its quarter of the reduce range (TM x 2 accumulator blocks), operands staged
// WAVE-PRIVATELY (no barrier in the loop) in a ring of two 16-deep stages that is refilled in place: when a stage's
// fragments are in registers its slot takes the stage after next, so two stages (24-30 KB per wave, 96-120 KB per CU) are
// in flight under the 24 TM / 2 matrix instructions of the current one.  tools/probe/l2lds.hip: a CU pulls ~48 B/clk out
// of its L2 once >= 32 KB are in flight; a 64 x 64 stage is 12 KB per 24 instructions = 512 B per matrix instruction, the
// pipe wants one per 8 clk per CU -- the L2 -> LDS path bounds the kernel at ~75 % of the six-product rate.
// RING = 2: two stages in flight under the current one's matrix instructions (ring refilled in place), one workgroup per CU
// (96-120 KB of LDS) -- for the launches that are one round of tiles anyway (N = 256 outputs: 200-300 tiles).
// RING = 1: one stage buffer per wave (48 KB per workgroup, registers capped for three workgroups per CU): a wave's own
// load is exposed, the other waves of its SIMD cover it, and one workgroup's prologue / epilogue / store drain hides under
// its neighbours' main loops -- for launches of many rounds (the 1024-wide FFN products: 1200 tiles).
// NW = 8 (with RING = 1): the reduce range split over EIGHT waves, two per SIMD.  A wave spends ~90 cycles issuing each
// 1 KB LDS-DMA instruction (15 per 36 matrix instructions at 96 x 64), during which its own matrix instructions cannot
// issue: with one wave per SIMD the pipe idles under the load issue (measured: 2,900 cycles per stage against 1,150 of
// matrix work, unchanged by any reordering inside the wave); with two, one wave's load issue runs beside the other's
// matrix instructions.
template <int LA, int LB, int TM, bool CS, int RING, int NW>
__global__ __launch_bounds__(64 * NW, NW == 8 ? 2 : (RING == 1 ? 3 : 1)) void gemm_px_kernel(const MesmGemmArgs p, const MesmPlanes PA, const MesmPlanes PB) {
  static_assert(NW == 4 || (NW == 8 && RING == 1), "wave count");
  extern __shared__ __attribute__((aligned(16))) char px_lds[];
  using SrcA = PxSrc<LA, TM>;
  using SrcB = PxSrc<LB, 2>;
  constexpr int A_BYTES = 3 * SrcA::PLANE, B_BYTES = 3 * SrcB::PLANE, SLOT = A_BYTES + B_BYTES;
  constexpr int G = 3 * (SrcA::NI + SrcB::NI);  // LDS-DMA instructions per stage and wave
  constexpr int BM = 32 * TM;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  Blk blk;
  xcd_tile(blockIdx.x, (p.M + BM - 1) / BM, (p.N + 63) / 64, blk.x, blk.y);
  blk.z = blockIdx.z;
  blk.slot = linear_block();
  const int m0 = blk.x * BM, n0 = blk.y * 64;

  // reduce range of this workgroup (split-K over blockIdx.z) and of this wave: multiples of 16 -- the planes are zero-padded
  const int KP = LA == R_ ? PA.cols : PA.rows;
  int kbeg = 0, kend = KP;
  if (p.split_k > 1) {
    int chunk = (KP + p.split_k - 1) / p.split_k;
    chunk = (chunk + 63) & ~63;
    kbeg = blk.z * chunk;
    kend = kbeg + chunk < KP ? kbeg + chunk : KP;
    if (kbeg >= KP) {
      if (blk.z > 0) return;
      kbeg = kend = KP;
    }
  }
  const int kw = (((kend - kbeg + NW - 1) / NW) + 15) & ~15;
  const int k0 = kbeg + wave * kw;
  const int k1 = k0 + kw < kend ? k0 + kw : kend;
  const int nst = (p.reserved0 & 8) ? 0 : (k1 > k0 ? (k1 - k0) >> 4 : 0);  // (8: tuning, no main loop)

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;

  char* mine = px_lds + wave * (RING * SLOT);
  const unsigned mine_addr = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)px_lds + (unsigned)wave * (RING * SLOT);
  SrcA sa;
  SrcB sb;
  sa.init(PA, m0, k0, lane);
  sb.init(PB, n0, k0, lane);
  auto issue = [&](int st, int slot_i) {
    const unsigned slot = mine_addr + (unsigned)slot_i * SLOT;
    sa.issue(st, slot);
    sb.issue(st, slot + A_BYTES);
  };

  f32x16 acc[TM][2];
#pragma unroll
  for (int i = 0; i < TM; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float csum[TM];
#pragma unroll
  for (int i = 0; i < TM; ++i) csum[i] = 0.0f;
  const bool do_colsum = (p.colsum != nullptr) && (blk.y == 0);

  // Stage order: the wave's stages are walked from a workgroup-dependent start and wrap around (any order sums the same
  // terms): workgroups that share an operand panel -- every row tile reads the same B, 75 of them in lockstep -- would
  // otherwise ask the same L2 channel for the same lines at the same time (measured: 17 -> 480 us on 4800 x 256 x 256
  // depending on where the buffers happened to lie).
  const int rot = nst > 0 ? (int)((unsigned)(blk.x * 5 + blk.y * 3 + blk.z) % (unsigned)nst) : 0;
  auto stage_of = [&](int i) { const int s_ = i + rot; return s_ < nst ? s_ : s_ - nst; };
  using T_ = std::true_type;
  using F_ = std::false_type;
  if constexpr (RING == 1) {
    // one stage buffer per wave: wait, fragments to registers, refill in place under the matrix instructions; the wave's
    // own load latency is covered by the other waves of its SIMD (three workgroups per CU)
    if (nst > 0) issue(stage_of(0), 0);
    auto iteration = [&](int it, auto refill_tag) {
      constexpr bool REFILL = decltype(refill_tag)::value;
      px_wait_vm<0>();
      PxFrag<LA, TM> fa;
      PxFrag<LB, 2> fb;
      fa.read(mine, lane);
      fb.read(mine + A_BYTES, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // the stage's fragments are in registers: refill its slot
      const int nxt = REFILL ? stage_of(it + 1) : 0;
      px_stage<0, 2 * TM, G, REFILL, LA, LB, TM>(fa, fb, acc, sa, sb, (int64_t)nxt * sa.kstep, (int64_t)nxt * sb.kstep, mine_addr,
                                                  (unsigned)A_BYTES);
      if (CS && do_colsum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) csum[i] += fa.ksum(i);
      }
    };
    int it = 0;
    for (; it + 1 < nst; ++it) iteration(it, T_{});
    if (it < nst) iteration(it, F_{});
  } else {
    // Ring of two slots, two fragment register sets.  Iteration `it` multiplies the fragments of stage it (in registers
    // since the previous iteration) while it reads the fragments of stage it + 1 (landed: issued two iterations ago) into
    // the other set and then refills that slot with stage it + 3.  Nothing waits in front of the matrix instructions but
    // the first fragment read of the prologue.
    PxFrag<LA, TM> fa0, fa1;
    PxFrag<LB, 2> fb0, fb1;
    const int a0a = PxFrag<LA, TM>::lane_off(lane), a0b = PxFrag<LB, 2>::lane_off(lane);
    if (nst > 0) issue(stage_of(0), 0);
    if (nst > 1) issue(stage_of(1), 1);
    if (nst > 1) px_wait_vm<G>();
    else px_wait_vm<0>();
    if (nst > 0) {
      fa0.read(mine, lane);
      fb0.read(mine + A_BYTES, lane);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      if (nst > 2) issue(stage_of(2), 0);
    }
    // PAR: parity of `it` (which register set is current); READ: stage it + 1 exists; REFILL: stage it + 3 exists;
    // MORE: stage it + 2 exists (it may still fly while stage it + 1 is waited for)
    auto iteration = [&](int it, auto par_tag, auto read_tag, auto refill_tag, auto more_tag) {
      constexpr int PAR = decltype(par_tag)::value;
      constexpr bool READ = decltype(read_tag)::value, REFILL = decltype(refill_tag)::value, MORE = decltype(more_tag)::value;
      if (READ) {
        if (MORE) px_wait_vm<G>();
        else px_wait_vm<0>();
      }
      const int si = PAR ^ 1;  // slot of stage it + 1
      const int nxt = REFILL ? stage_of(it + 3) : 0;
      if (PAR == 0)
        px_stage2<0, 2 * TM, G, READ, REFILL, LA, LB, TM>(fa0, fb0, fa1, fb1, acc, mine + si * SLOT, mine + si * SLOT + A_BYTES, a0a,
                                                          a0b, sa, sb, (int64_t)nxt * sa.kstep, (int64_t)nxt * sb.kstep,
                                                          mine_addr + (unsigned)si * SLOT, (unsigned)A_BYTES);
      else
        px_stage2<0, 2 * TM, G, READ, REFILL, LA, LB, TM>(fa1, fb1, fa0, fb0, acc, mine + si * SLOT, mine + si * SLOT + A_BYTES, a0a,
                                                          a0b, sa, sb, (int64_t)nxt * sa.kstep, (int64_t)nxt * sb.kstep,
                                                          mine_addr + (unsigned)si * SLOT, (unsigned)A_BYTES);
      if (CS && do_colsum) {
#pragma unroll
        for (int i = 0; i < TM; ++i) csum[i] += PAR == 0 ? fa0.ksum(i) : fa1.ksum(i);
      }
    };
    using P0 = std::integral_constant<int, 0>;
    using P1 = std::integral_constant<int, 1>;
    int it = 0;
    for (; it + 4 < nst; it += 2) {  // both iterations of the pair read, refill and have a stage in flight behind them
      iteration(it, P0{}, T_{}, T_{}, T_{});
      iteration(it + 1, P1{}, T_{}, T_{}, T_{});
    }
    // tail: at most four iterations, flags by position
    for (; it < nst; ++it) {
      const bool rd = it + 1 < nst, more = it + 2 < nst, rf = it + 3 < nst;
      if ((it & 1) == 0) {
        if (rf) iteration(it, P0{}, T_{}, T_{}, T_{});
        else if (more) iteration(it, P0{}, T_{}, F_{}, T_{});
        else if (rd) iteration(it, P0{}, T_{}, F_{}, F_{});
        else iteration(it, P0{}, F_{}, F_{}, F_{});
      } else {
        if (rf) iteration(it, P1{}, T_{}, T_{}, T_{});
        else if (more) iteration(it, P1{}, T_{}, F_{}, T_{});
        else if (rd) iteration(it, P1{}, T_{}, F_{}, F_{});
        else iteration(it, P1{}, F_{}, F_{}, F_{});
      }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");

  if (do_colsum) {
#pragma unroll
    for (int t = 0; t < TM; ++t) {
      const float c = add_xor32(csum[t]);
      const int gm = m0 + 32 * t + li;
      if (h == 0 && gm < p.M && c != 0.0f) atomicAdd(p.colsum + gm, c);
    }
  }
  if ((p.reserved0 & 4) && acc[0][0][0] != 12345.678f) return;  // (4: tuning, no epilogue)
  __syncthreads();  // every wave is done with its ring: the reduction buffer aliases it
  // The NW partial tiles meet in LDS in log2(NW) exchange rounds (a butterfly: 32 KB at NW = 4 instead of the 64 KB of an
  // all-to-all, which is what lets three workgroups share a CU).  In round d (partner w ^ (NW >> (d + 1))) a wave keeps
  // half of the accumulator registers it still owns and hands the other half over; after the last round wave w owns
  // registers [w RPW, (w + 1) RPW) of every block, RPW = 16 / NW (rows 4 h + (r & 3) + 8 (r >> 2) of each 32-row block).
  float* L = reinterpret_cast<float*>(px_lds);
  constexpr int NBLK = TM * 2;
  constexpr int RPW = 16 / NW;
  {
    int lo = 0;  // first register of the range this wave still owns (wave-uniform), span halves every round
#pragma unroll
    for (int span = 16; span > RPW; span >>= 1) {
      const int hs = span >> 1;
      const int bit = NW * hs / 16;          // partner distance: NW / 2, NW / 4, ..
      const bool upper = (wave & bit) != 0;  // keeps the upper half of its range
      const int keep_lo = upper ? lo + hs : lo, give_lo = upper ? lo : lo + hs;
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r >= give_lo && r < give_lo + hs) L[((wave * NBLK + ti * 2 + tj) * 8 + (r - give_lo)) * 64 + lane] = acc[ti][tj][r];
      __syncthreads();
#pragma unroll
      for (int ti = 0; ti < TM; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
          for (int r = 0; r < 16; ++r)
            if (r >= keep_lo && r < keep_lo + hs)
              acc[ti][tj][r] += L[(((wave ^ bit) * NBLK + ti * 2 + tj) * 8 + (r - keep_lo)) * 64 + lane];
      __syncthreads();
      lo = keep_lo;
    }
  }
  const bool first_split = (p.split_k <= 1) || (blk.z == 0);
  const int r0 = wave * RPW;  // this wave's registers [r0, r0 + RPW) of every block
  const int rbase = m0 + 4 * h;
  auto RO = [r0](int i) { const int r = r0 + (i % RPW); return (r & 3) + 8 * (r >> 2) + 32 * (i / RPW); };
  float dslope_part = 0.0f;
  float vals[2][RPW * TM];
#pragma unroll
  for (int tj = 0; tj < 2; ++tj)
#pragma unroll
    for (int ti = 0; ti < TM; ++ti)
#pragma unroll
      for (int i = 0; i < RPW; ++i) {
        float own = 0.0f;  // acc[ti][tj][r0 + i] with a wave-uniform dynamic index: selects instead of scratch
#pragma unroll
        for (int w = 0; w < NW; ++w) own = (w == wave) ? acc[ti][tj][w * RPW + i] : own;
        vals[tj][RPW * ti + i] = own;
      }
  __syncthreads();  // dslope_store reuses the head of L
#pragma unroll
  for (int tj = 0; tj < 2; ++tj) {
    const int col = n0 + 32 * tj + li;
    if (m0 + BM <= p.M && n0 + 32 * tj + 32 <= p.N)
      dslope_part += staged_epilogue<RPW * TM, true>(p, vals[tj], rbase, col, slope, seed_off, first_split, RO);
    else
      dslope_part += staged_epilogue<RPW * TM, false>(p, vals[tj], rbase, col, slope, seed_off, first_split, RO);
  }
  if (p.e_actgrad == MESM_ACT_PRELU && p.dslope) dslope_store(p, dslope_part, L, blk.slot);
}

// ---- shared-stage kernel ("pxs") --------------------------------------------------------------------------------------
// The exact-f32 ring kernel's structure (gemm.hip: gemm_lds64_kernel) on plane operands: 64 x 64 tile, 2 x 2 waves of one
// 32 x 32 accumulator block each, stages of 32 reduce indices (two 16-deep substeps x three planes x two operands = 24 KB)
// staged ONCE per workgroup by LDS-DMA into a ring of NS slots -- each wave issues 6 of the stage's 24 instructions, a
// quarter of what a k-split wave issues for the same matrix work (the per-wave LDS-DMA issue cost, ~90 cycles per 1 KB
// instruction, is what bounded the k-split kernels) -- one raw s_barrier per stage, 48-72 KB of LDS and ~100 registers: two
// or three workgroups per CU, so one workgroup's prologue, barrier waits, epilogue and store drain run under its neighbours'
// matrix instructions.
constexpr int PXS_OPER = 12288, PXS_STAGE = 2 * PXS_OPER;  // bytes: [A | B] x [hi | mid | lo] x [substep 0 | 1] x 2 KB

template <int LAYOUT>
struct PxsSrc {
  unsigned voff[2];     // per-lane byte offsets of the two 1 KB halves of a 2 KB plane-substep image
  int64_t sub_step;     // bytes between the two substeps of a stage
  int64_t stage_step;   // bytes between stages
  __device__ __forceinline__ void init(const MesmPlanes& P, int o0, int lane) {
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      if (LAYOUT == R_) {
        const int row = 32 * q + (lane >> 1), pos = lane & 1;
        const int c = pos ^ ((row >> 3) & 1);
        int ro = o0 + row;
        ro = ro < P.rows ? ro : P.rows - 1;
        voff[q] = (unsigned)(ro * 16 + 8 * c) * 2u;
      } else {
        const int krow = 8 * q + (lane >> 3), pos = lane & 7;
        const int ch = pos ^ (((krow >> 1) & 1) << 2);
        int cb = (o0 >> 4) + (ch >> 1);
        const int ncb = P.cols >> 4;
        cb = cb < ncb ? cb : ncb - 1;
        voff[q] = (unsigned)(cb * (int)P.ld + krow * 16 + (ch & 1) * 8) * 2u;
      }
    }
    sub_step = LAYOUT == R_ ? 2 * P.ld : 512;
    stage_step = 2 * sub_step;
  }
};

// one fragment: plane image of one substep at `img` (2 KB), outer block t, per-lane offset a0 (PxFrag::lane_off)
template <int LAYOUT>
__device__ __forceinline__ u32x4 pxs_frag(const char* img, int t, int a0) {
  if (LAYOUT == R_) return *reinterpret_cast<const u32x4*>(img + t * 1024 + a0);
  const char* a = img + (t ? (a0 ^ 64) : a0);
  const s16x4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a));
  const s16x4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(a + 512));
  const u32x2 l2 = __builtin_bit_cast(u32x2, lo4), h2 = __builtin_bit_cast(u32x2, hi4);
  return u32x4{l2[0], l2[1], h2[0], h2[1]};
}

template <int LA, int LB, bool CS, int NS>
__global__ __launch_bounds__(NTHREADS, NS == 2 ? 3 : 2) void gemm_pxs_kernel(const MesmGemmArgs p, const MesmPlanes PA, const MesmPlanes PB) {
  extern __shared__ __attribute__((aligned(16))) char px_lds[];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  Blk blk;
  xcd_tile(blockIdx.x, (p.M + 63) / 64, (p.N + 63) / 64, blk.x, blk.y);
  blk.z = blockIdx.z;
  blk.slot = linear_block();
  const int m0 = blk.x * 64, n0 = blk.y * 64;

  const int KP = LA == R_ ? PA.cols : PA.rows;  // padded reduce extent: a multiple of 32, zeros beyond K
  int kbeg = 0, kend = KP;
  if (p.split_k > 1) {
    int chunk = (KP + p.split_k - 1) / p.split_k;
    chunk = (chunk + 31) & ~31;
    kbeg = blk.z * chunk;
    kend = kbeg + chunk < KP ? kbeg + chunk : KP;
    if (kbeg >= KP) {
      if (blk.z > 0) return;
      kbeg = kend = KP;
    }
  }
  const int nst = (kend - kbeg) >> 5;

  const float slope = p.slope ? *p.slope : 0.0f;
  const uint32_t seed_off = p.seed_offset ? *p.seed_offset : 0u;

  // waves 0 / 1 stage operand A, waves 2 / 3 operand B; the operand's 12 instructions per stage (plane x substep x half)
  // are split 6 / 6 between its two waves
  const bool mine_a = wave < 2;
  const int part = wave & 1;
  PxsSrc<LA> sa;
  PxsSrc<LB> sb;
  sa.init(PA, m0, lane);
  sb.init(PB, n0, lane);
  const char* base[3];
  {
    const MesmPlanes& P = mine_a ? PA : PB;
    const bool red = mine_a ? (LA == R_) : (LB == R_);
    const int64_t first = red ? (int64_t)(kbeg >> 4) * P.ld : (int64_t)kbeg * 16;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) base[pl] = reinterpret_cast<const char*>(P.p[pl] + first);
  }
  const unsigned v0 = mine_a ? sa.voff[0] : sb.voff[0], v1 = mine_a ? sa.voff[1] : sb.voff[1];
  const int64_t sub_step = mine_a ? sa.sub_step : sb.sub_step, stage_step = mine_a ? sa.stage_step : sb.stage_step;
  const unsigned lds0 = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)px_lds;
  // the stages are walked from a workgroup-dependent start (see the k-split kernel: L2 channel hot spots otherwise)
  const int rot = nst > 0 ? (int)((unsigned)(blk.x * 5 + blk.y * 3 + blk.z) % (unsigned)nst) : 0;
  auto stage_of = [&](int i) { const int s_ = i + rot; return s_ < nst ? s_ : s_ - nst; };
  int i_slot = 0;
  auto issue = [&](int i) {
    const int64_t adv = (int64_t)stage_of(i) * stage_step;
    const unsigned slot = lds0 + (unsigned)i_slot * PXS_STAGE + (mine_a ? 0u : (unsigned)PXS_OPER);
    i_slot = i_slot + 1 == NS ? 0 : i_slot + 1;
#pragma unroll
    for (int j = 0; j < 6; ++j) {
      const int idx = part * 6 + j;          // 0 .. 11: plane = idx / 4, substep = (idx / 2) & 1, half = idx & 1
      const int pl = idx >> 2, sub = (idx >> 1) & 1, q = idx & 1;
      const uint64_t bv = (uint64_t)((pl == 0 ? base[0] : (pl == 1 ? base[1] : base[2])) + adv + (sub ? sub_step : 0));
      // (wave-uniform by construction; the asm wants it in SGPRs)
      const uint64_t bs = ((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(bv >> 32)) << 32) |
                          (uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)bv);
      px_glds(reinterpret_cast<const void*>(bs), q ? v1 : v0,
              __builtin_amdgcn_readfirstlane(slot + (unsigned)(pl * 4096 + sub * 2048 + q * 1024)));
    }
  };

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
  float csum = 0.0f;
  const bool do_colsum = CS && (p.colsum != nullptr) && (blk.y == 0) && (wn == 0);
  const int a0a = PxFrag<LA, 2>::lane_off(lane), a0b = PxFrag<LB, 2>::lane_off(lane);

  constexpr int AHEAD = NS - 1;
#pragma unroll
  for (int i = 0; i < AHEAD; ++i)
    if (i < nst) issue(i);
  int c_slot = 0;
  for (int st = 0; st < nst; ++st) {
    // this wave's 6 instructions of stage st have landed; the min(AHEAD - 1, nst - 1 - st) stages issued after it may fly
    const int later = nst - 1 - st < AHEAD - 1 ? nst - 1 - st : AHEAD - 1;
    if (later == 0) px_wait_vm<0>();
    else if (later == 1) px_wait_vm<6>();
    else px_wait_vm<12>();
    // ... and so have the other waves'; the barrier also orders every wave's fragment reads of stage st - 1 ahead of the
    // refill of its slot issued below
    __builtin_amdgcn_s_barrier();
    if (st + AHEAD < nst) issue(st + AHEAD);
    const char* sl = px_lds + c_slot * PXS_STAGE;
    c_slot = c_slot + 1 == NS ? 0 : c_slot + 1;
    u32x4 fa[2][3], fb[2][3];
#pragma unroll
    for (int sub = 0; sub < 2; ++sub)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        fa[sub][pl] = pxs_frag<LA>(sl + pl * 4096 + sub * 2048, wm, a0a);
        fb[sub][pl] = pxs_frag<LB>(sl + PXS_OPER + pl * 4096 + sub * 2048, wn, a0b);
      }
#define PX_BF(x) __builtin_bit_cast(bf16x8, x)
#pragma unroll
    for (int sub = 0; sub < 2; ++sub) {  // smallest terms first
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][2]), PX_BF(fb[sub][0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][0]), PX_BF(fb[sub][2]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][1]), PX_BF(fb[sub][1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][1]), PX_BF(fb[sub][0]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][0]), PX_BF(fb[sub][1]), acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(PX_BF(fa[sub][0]), PX_BF(fb[sub][0]), acc, 0, 0, 0);
    }
#undef PX_BF
    if (do_colsum) {
#pragma unroll
      for (int sub = 0; sub < 2; ++sub)
#pragma unroll
        for (int pl = 2; pl >= 0; --pl)
#pragma unroll
          for (int w = 0; w < 4; ++w) {
            const unsigned u = fa[sub][pl][w];
            csum += __uint_as_float(u << 16) + __uint_as_float(u & 0xFFFF0000u);
          }
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  if (do_colsum) {
    const float c = add_xor32(csum);
    const int gm = m0 + 32 * wm + li;
    if (h == 0 && gm < p.M && c != 0.0f) atomicAdd(p.colsum + gm, c);
  }
  __syncthreads();  // every wave is done with the ring: dslope_store uses its head
  tile16_epilogue<R_, R_, false>(p, acc, m0 + 32 * wm, n0 + 32 * wn, slope, seed_off, blk.z, reinterpret_cast<float*>(px_lds),
                                 blk.slot, p.K, XForm{}, XForm{});
}

// ---- operand split ---------------------------------------------------------------------------------------------------
// x (rows x cols f32, leading dimension ld) -> hi / mid / lo planes [rows_pad][ldp], zeros in the padding.
// One thread = 8 consecutive columns of one row: 32 bytes in, 3 x 16 bytes out.
__device__ __forceinline__ void px_split3(float x, unsigned& hi, unsigned& mid, unsigned& lo) {
  const unsigned u = __float_as_uint(x);
  const bool fin = (u & 0x7F800000u) != 0x7F800000u;  // inf / nan: hi carries it, the residual planes stay zero
  hi = u >> 16;
  const float r1 = fin ? x - __uint_as_float(u & 0xFFFF0000u) : 0.0f;
  const unsigned m = __float_as_uint(r1);
  mid = m >> 16;
  const float r2 = r1 - __uint_as_float(m & 0xFFFF0000u);
  lo = __float_as_uint(r2) >> 16;
}

struct SplitDesc {
  const float* x;
  int64_t ld;
  int32_t rows, cols;
  MesmPlanes out;
  int32_t start;  // first workgroup of this tensor inside a grouped launch
  int32_t pad_;
};

// A wave takes 8 rows x 64 columns: lane = 8 * (row in the group) + piece, piece = 2 * (column block of the four) + half --
// 8 lanes read 256 contiguous bytes of a row, and the 8 rows of one column block are written as 256 contiguous bytes per plane.
__device__ __forceinline__ void px_split_body(const SplitDesc& d, int local_block) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int ncg = (d.out.cols + 63) >> 6;            // 64-column groups
  const int64_t wid = (int64_t)local_block * 4 + wave;
  const int64_t nw = (int64_t)(d.out.rows >> 3) * ncg;
  if (wid >= nw) return;
  const int rg = (int)(wid / ncg), cg = (int)(wid - (int64_t)rg * ncg);
  const int row = rg * 8 + (lane >> 3), piece = lane & 7;
  const int cb = cg * 4 + (piece >> 1), c0 = cb * 16 + (piece & 1) * 8;
  if (cb >= (d.out.cols >> 4)) return;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.0f;
  if (row < d.rows) {
    const float* src = d.x + (int64_t)row * d.ld + c0;
    if (c0 + 8 <= d.cols && ((((uintptr_t)src) & 15) == 0)) {
      const float4 a = *reinterpret_cast<const float4*>(src), b = *reinterpret_cast<const float4*>(src + 4);
      v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w; v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
    } else {
#pragma unroll
      for (int e = 0; e < 8; ++e)
        if (c0 + e < d.cols) v[e] = src[e];
    }
  }
  unsigned hi[8], mid[8], lo[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) px_split3(v[e], hi[e], mid[e], lo[e]);
  const int64_t o = (int64_t)cb * d.out.ld + (int64_t)row * 16 + (piece & 1) * 8;
  u32x4 ph, pm, pl;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    ph[w] = hi[2 * w] | (hi[2 * w + 1] << 16);
    pm[w] = mid[2 * w] | (mid[2 * w + 1] << 16);
    pl[w] = lo[2 * w] | (lo[2 * w + 1] << 16);
  }
  *reinterpret_cast<u32x4*>(d.out.p[0] + o) = ph;
  *reinterpret_cast<u32x4*>(d.out.p[1] + o) = pm;
  *reinterpret_cast<u32x4*>(d.out.p[2] + o) = pl;
}

__global__ __launch_bounds__(256) void px_split_kernel(const SplitDesc d) { px_split_body(d, blockIdx.x); }

// many tensors in one launch (the once-per-step split of every weight matrix): descriptors in device memory, sorted by
// `start`; a workgroup finds its tensor by bisection
__global__ __launch_bounds__(256) void px_split_table_kernel(const SplitDesc* __restrict__ tab, int n) {
  const int bid = blockIdx.x;
  int lo = 0, hi = n - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (tab[mid].start <= bid) lo = mid;
    else hi = mid - 1;
  }
  const SplitDesc d = tab[lo];
  px_split_body(d, bid - d.start);
}

int px_check_planes(const MesmPlanes& P) {
  for (int i = 0; i < 3; ++i)
    if (!P.p[i] || (((uintptr_t)P.p[i]) & 15)) return MESM_EALIGN;
  if (P.rows <= 0 || P.cols <= 0 || (P.rows & 31) || (P.cols & 31) || P.ld < (int64_t)P.rows * 16 || (P.ld & 7)) return MESM_EINVAL;
  if ((int64_t)(P.cols >> 4) * P.ld >= (1ll << 31)) return MESM_EINVAL;  // 32-bit byte offsets inside a plane
  return MESM_OK;
}

template <int LA, int LB, int TM, bool CS, int RING, int NW>
int px_launch_ring(const MesmGemmArgs& a, const MesmPlanes& PA, const MesmPlanes& PB, hipStream_t s) {
  constexpr int SLOT = 3 * PxSrc<LA, TM>::PLANE + 3 * PxSrc<LB, 2>::PLANE;
  // the cross-wave reduction buffer aliases the ring
  constexpr int RED = TM * 2 * NW * 8 * 64 * 4;  // round 1 of the butterfly: 8 of 16 registers per block and wave
  constexpr int LDS = NW * RING * SLOT > RED ? NW * RING * SLOT : RED;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_px_kernel<LA, LB, TM, CS, RING, NW>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS) != hipSuccess)
      return MESM_ELAUNCH;
    attr_done = true;
  }
  const int z = a.split_k > 1 ? a.split_k : 1;
  dim3 grid(((a.M + 32 * TM - 1) / (32 * TM)) * ((a.N + 63) / 64), 1, z);
  hipLaunchKernelGGL((gemm_px_kernel<LA, LB, TM, CS, RING, NW>), grid, dim3(64 * NW), LDS, s, a, PA, PB);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : mesm_gemm_dslope_finish(a, (int64_t)grid.x * z, s);
}

int g_px_ring = []() { const char* e = getenv("MESM_PX_RING"); return e ? atoi(e) : 0; }();

template <int LA, int LB, int TM, bool CS>
int px_launch_cs(const MesmGemmArgs& a, const MesmPlanes& PA, const MesmPlanes& PB, hipStream_t s) {
  // many rounds of tiles -> three small workgroups per CU (RING = 1); a single round -> the deep ring (MESM_PX_RING pins)
  const long z = a.split_k > 1 ? a.split_k : 1;
  const long tiles = (long)((a.M + 32 * TM - 1) / (32 * TM)) * ((a.N + 63) / 64) * z;
  const int mode = g_px_ring ? g_px_ring : (tiles > 512 ? 1 : 8);  // 1: RING 1 x 4 waves, 2: RING 2 x 4 waves, 8: 8 waves
  const long kper8 = ((LA == R_ ? PA.cols : PA.rows) / z) / 8;
  if (mode == 8 && kper8 >= 64) return px_launch_ring<LA, LB, TM, CS, 1, 8>(a, PA, PB, s);
  if (TM == 2 && mode == 1) return px_launch_ring<LA, LB, 2, CS, 1, 4>(a, PA, PB, s);
  return px_launch_ring<LA, LB, TM, CS, 2, 4>(a, PA, PB, s);
}

template <int LA, int LB, bool CS, int NS>
int pxs_launch_ns(const MesmGemmArgs& a, const MesmPlanes& PA, const MesmPlanes& PB, hipStream_t s) {
  constexpr int LDS = NS * PXS_STAGE;
  static bool attr_done = false;
  if (!attr_done) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(gemm_pxs_kernel<LA, LB, CS, NS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS) != hipSuccess)
      return MESM_ELAUNCH;
    attr_done = true;
  }
  const int z = a.split_k > 1 ? a.split_k : 1;
  dim3 grid(((a.M + 63) / 64) * ((a.N + 63) / 64), 1, z);
  hipLaunchKernelGGL((gemm_pxs_kernel<LA, LB, CS, NS>), grid, dim3(NTHREADS), LDS, s, a, PA, PB);
  const int rc = mesm_launch_status();
  return rc != MESM_OK ? rc : mesm_gemm_dslope_finish(a, (int64_t)grid.x * z, s);
}

template <int LA, int LB>
int pxs_launch(const MesmGemmArgs& a, const MesmPlanes& PA, const MesmPlanes& PB, int ns, hipStream_t s) {
  if (a.colsum) return ns == 2 ? pxs_launch_ns<LA, LB, true, 2>(a, PA, PB, s) : pxs_launch_ns<LA, LB, true, 3>(a, PA, PB, s);
  return ns == 2 ? pxs_launch_ns<LA, LB, false, 2>(a, PA, PB, s) : pxs_launch_ns<LA, LB, false, 3>(a, PA, PB, s);
}

template <int LA, int LB, int TM>
int px_launch(const MesmGemmArgs& a, const MesmPlanes& PA, const MesmPlanes& PB, hipStream_t s) {
  if (g_px_ring == 32 || g_px_ring == 33 || g_px_ring == 0) return pxs_launch<LA, LB>(a, PA, PB, g_px_ring == 32 ? 2 : 3, s);
  return a.colsum ? px_launch_cs<LA, LB, TM, true>(a, PA, PB, s) : px_launch_cs<LA, LB, TM, false>(a, PA, PB, s);
}

// 96 x 64 tiles when they fill the 256 CUs in fewer / fuller rounds than 64 x 64 (4800 x 256: 200 workgroups in one round
// against 300 in two), and move fewer bytes per matrix instruction; MESM_PX_TILE=64|96 pins (tuning)
int g_px_debug = []() { const char* e = getenv("MESM_PX_DEBUG"); return e ? atoi(e) : 0; }();
int g_px_tile = []() { const char* e = getenv("MESM_PX_TILE"); return e ? atoi(e) : 0; }();
bool px_use96(const MesmGemmArgs& a) {
  if (a.a_layout != R_) return false;
  if (g_px_tile == 64) return false;
  if (g_px_tile == 96) return true;
  const long z = a.split_k > 1 ? a.split_k : 1;
  const long nt = (a.N + 63) / 64;
  const long t64 = (long)((a.M + 63) / 64) * nt * z, t96 = (long)((a.M + 95) / 96) * nt * z;
  const long c64 = ((t64 + 255) / 256) * 64, c96 = ((t96 + 255) / 256) * 96;
  return c96 <= c64;
}

}  // namespace

extern "C" int mesm_gemm_px_set_tile(int32_t tile) {  // tuning tools: 0 = auto, 64, 96
  g_px_tile = tile;
  return MESM_OK;
}
extern "C" int mesm_gemm_px_set_ring(int32_t mode) {  // tuning tools: 0 = auto, 1 | 2 = ring depth with 4 waves, 8 = 8 waves
  g_px_ring = mode;
  return MESM_OK;
}

extern "C" int mesm_gemm_px_supported(const MesmGemmArgs* args) {
  if (!args) return 0;
  const MesmGemmArgs& a = *args;
  if (a.A2 || a.B2 || a.a_act != MESM_ACT_NONE || a.b_act != MESM_ACT_NONE || a.a_drop_p > 0.f || a.b_drop_p > 0.f) return 0;
  if (a.a_layout == O_ && a.b_layout == R_) return 0;
  return 1;
}

extern "C" int mesm_gemm_px(const MesmGemmArgs* args, const MesmPlanes* pa, const MesmPlanes* pb, void* stream) {
  if (!args || !pa || !pb) return MESM_EINVAL;
  MesmGemmArgs a = *args;
  if (!mesm_gemm_px_supported(&a)) return MESM_EINVAL;
  if (!a.C || a.M <= 0 || a.N <= 0 || a.K <= 0) return MESM_EINVAL;
  int rc = px_check_planes(*pa);
  if (rc == MESM_OK) rc = px_check_planes(*pb);
  if (rc != MESM_OK) return rc;
  // plane extents against the logical problem: outer extents cover M / N, the (padded) reduce extents agree and cover K
  const int a_outer = a.a_layout == R_ ? pa->rows : pa->cols, a_k = a.a_layout == R_ ? pa->cols : pa->rows;
  const int b_outer = a.b_layout == R_ ? pb->rows : pb->cols, b_k = a.b_layout == R_ ? pb->cols : pb->rows;
  if (a_outer < ((a.M + 31) & ~31) || b_outer < ((a.N + 31) & ~31) || a_k != b_k || a_k < a.K || a_k >= a.K + 32) return MESM_EINVAL;
  if (a.e_actgrad != MESM_ACT_NONE && !a.aux) return MESM_EINVAL;
  if (a.e_actgrad == MESM_ACT_PRELU && a.dslope && !a.dslope_ws) return MESM_EINVAL;
  if ((a.e_act == MESM_ACT_PRELU || a.e_actgrad == MESM_ACT_PRELU) && !a.slope) return MESM_EINVAL;
  if (a.pre_out && (a.split_k > 1 || a.accumulate != 0)) return MESM_EINVAL;
  if (a.e_drop_p < 0.f || a.e_drop_p >= 1.f) return MESM_EINVAL;
  if (a.split_k < 1) a.split_k = 1;
  if (a.split_k > 1) {
    if (a.e_act != MESM_ACT_NONE || a.e_actgrad != MESM_ACT_NONE || a.e_drop_p > 0.f) return MESM_EINVAL;
    a.accumulate = 2;
    const int max_split = (a_k + 63) / 64;
    if (a.split_k > max_split) a.split_k = max_split;
  }
  if (a.accumulate < 0 || a.accumulate > 2) return MESM_EINVAL;
  if (a.out_scale == 0.0f) a.out_scale = 1.0f;
  a.reserved0 = g_px_debug;  // tuning: 1 = no refill loads, 2 = no matrix instructions (wrong results, timing only)
  hipStream_t s = (hipStream_t)stream;
  const bool t96 = px_use96(a);
  if (a.a_layout == R_ && a.b_layout == R_) return t96 ? px_launch<R_, R_, 3>(a, *pa, *pb, s) : px_launch<R_, R_, 2>(a, *pa, *pb, s);
  if (a.a_layout == R_ && a.b_layout == O_) return t96 ? px_launch<R_, O_, 3>(a, *pa, *pb, s) : px_launch<R_, O_, 2>(a, *pa, *pb, s);
  return px_launch<O_, O_, 2>(a, *pa, *pb, s);
}

extern "C" int mesm_split_planes(const float* x, int64_t ld, int32_t rows, int32_t cols, const MesmPlanes* out, void* stream) {
  if (!x || !out || rows <= 0 || cols <= 0 || ld < cols) return MESM_EINVAL;
  const int rc = px_check_planes(*out);
  if (rc != MESM_OK) return rc;
  if (out->rows < rows || out->cols < cols) return MESM_EINVAL;
  SplitDesc d;
  d.x = x; d.ld = ld; d.rows = rows; d.cols = cols; d.out = *out; d.start = 0; d.pad_ = 0;
  const int64_t waves = (int64_t)(out->rows >> 3) * ((out->cols + 63) >> 6);
  hipLaunchKernelGGL(px_split_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, d);
  return mesm_launch_status();
}

extern "C" int64_t mesm_split_desc_size(void) { return (int64_t)sizeof(SplitDesc); }

// host-side helper: fill entry `idx` of a descriptor table (host memory, mesm_split_desc_size() bytes per entry) and
// return the number of workgroups the tensor takes; the caller accumulates `start`, uploads the table and launches
// mesm_split_planes_table(table_dev, n, total_workgroups)
extern "C" int32_t mesm_split_desc_fill(void* table_host, int32_t idx, const float* x, int64_t ld, int32_t rows, int32_t cols,
                                        const MesmPlanes* out, int32_t start) {
  if (!table_host || !x || !out || px_check_planes(*out) != MESM_OK || out->rows < rows || out->cols < cols) return -1;
  SplitDesc* d = reinterpret_cast<SplitDesc*>(table_host) + idx;
  d->x = x; d->ld = ld; d->rows = rows; d->cols = cols; d->out = *out; d->start = start; d->pad_ = 0;
  const int64_t waves = (int64_t)(out->rows >> 3) * ((out->cols + 63) >> 6);
  return (int32_t)((waves + 3) / 4);
}

extern "C" int mesm_split_planes_table(const void* table_dev, int32_t n, int32_t total_workgroups, void* stream) {
  if (!table_dev || n <= 0 || total_workgroups <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(px_split_table_kernel, dim3((unsigned)total_workgroups), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const SplitDesc*>(table_dev), (int)n);
  return mesm_launch_status();
}
