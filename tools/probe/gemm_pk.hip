// Persistent ("stream-K") form of the grouped split-bf16 GEMM launch (round 5).
//
// gemm_wstage64_group_kernel (gemm.hip) gives every 64 x 64 output tile of every problem of a call its own workgroup.
// Its steady state is fine -- 8192 x 256 x 1024 as 512 workgroups on the 512 slots of the chip: 30 us, 142 TF -- but the
// step's calls have 272 ... 792 tiles: 300 equal tiles put two workgroups on 44 CUs and one on the other 212, and the
// launch lasts as long as two rounds (4800 x 256 x 1024: 33 us where 256 tiles take 17); and every workgroup pays its own
// ramp (arguments, first loads in flight, first loads landed: ~1.8 us) before its first matrix instruction.
//
// Here the grid is FIXED (PK_GRID workgroups: two per CU) and the work is a flat list of UNITS -- one unit = one 128-deep
// slice of the reduce range of one 64 x 64 tile (32 reduce indices for each of the four waves: one stage of the k-split
// body) -- ordered (problem, k-slice of a split-K problem, tile, stage).  Workgroup at position i takes the units
// [i W / G, (i + 1) W / G): every workgroup the same number of stages whatever the tile count.  Positions are dealt so
// that the workgroups of an XCD (blockIdx % 8, observed placement: speed only) own one CONTIGUOUS range: the column
// tiles of a row tile, and the two halves of a tile that is cut between two workgroups, meet in one L2.
//
// A range is a run of PIECES, a piece = consecutive stages of one tile:
//   * a whole tile: the k-split body as before, epilogue from registers;
//   * a tile cut by a range boundary: every piece but the one that holds the tile's FIRST stage is a contributor -- it is
//     the first thing its workgroup does, its 64 x 64 partial sum goes to the workgroup's slot of a workspace (16 KB,
//     write-through stores) and a flag is raised; the piece with the first stage is the LAST thing its workgroup does
//     (ranges are walked front to back), so by the time it has multiplied its own stages the contributors' partials have
//     been sitting in memory for the whole launch: it adds them (in position order: deterministic), lowers the flags
//     and runs the tile's epilogue.  A finisher only ever waits for workgroups that owe nothing to anybody before they
//     raise their flag, so the scheme cannot deadlock as long as all PK_GRID workgroups are resident (two 64 KB / 256
//     thread workgroups per CU: checked against the occupancy query at first use); the wait is bounded anyway and
//     a time-out is reported through mesm_gemm_pk_status();
//   * problems that accumulate with atomics (split-K weight gradients, shared gradient buffers): every piece adds its
//     partial product straight into C, bias / residual / K-tail ride with the piece that holds reduce index 0.
// After a piece's cross-wave reduction the slabs are free: the NEXT piece's first stage is issued before the epilogue
// (or the hand-off) of the current one, so the first-load latency of a tile hides behind its predecessor's stores.
//
// Visibility of a partial between workgroups (MI355X_MICROARCH.md, "Workgroup dispatch, XCD placement & inter-workgroup
// visibility"): payload and flag move with sc0 sc1 stores / loads (write-through, L1-bypassing) -- no agent-scope fence,
// whose write-back / invalidate every other workgroup of the XCD would pay for (DESIGN.md section 7, round 4 probe).
// -DMESM_PK_FENCE=1 builds the fenced form (plain accesses + release / acquire) for A/B.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "gemm_ws.hpp"

#ifndef MESM_PK_FENCE
#define MESM_PK_FENCE 0
#endif

namespace {

constexpr int PK_MACRO = 128;        // reduce indices of one unit (4 waves x one 32-deep stage)
constexpr int PK_GRID_MAX = 512;     // two workgroups per CU
constexpr int PK_SLOT = 64 * 64;     // floats of one partial tile
constexpr int PK_FLAG_STRIDE = 32;   // one flag per 128-byte line
constexpr unsigned PK_SPIN_LIMIT = 1u << 21;

struct PkArgs {
  MesmGemmArgs p[GROUP_MAX];
  int ustart[GROUP_MAX + 1];  // first unit of every problem
  int S[GROUP_MAX];           // units per (tile, k-slice)
  int cut[GROUP_MAX];         // 0 = tiles stay whole, 1 = may be cut (hand-off through the workspace), 2 = atomics
  int c0;                     // fixed cost of a tile in units (see pk_unit_begin): a run of S stages weighs S + c0
  int n;
  int W;                      // = ustart[n]
  float* ws;                  // PK_GRID_MAX slots of PK_SLOT floats
  unsigned* flags;            // PK_GRID_MAX flags, PK_FLAG_STRIDE apart
  unsigned* status;           // != 0: a finisher gave up waiting (results of that launch are wrong)
};

struct Piece {
  int gi;    // problem
  int unit;  // (k-slice, tile) index inside the problem
  int s0;    // first stage of the piece inside its run
  int len;   // stages
  int S;     // stages of the whole (tile, slice)
  int u;     // global index of the piece's first (weighted) unit
  int vlen;  // weighted units the piece covers (u + vlen = the next piece)
  int V;     // weighted units of the whole run = S + c0
  bool whole;
};

// First unit of the workgroup at position pos: pos W / G, SNAPPED so that a hand-off is only ever paid where it pays.
// Cutting a tile between two workgroups costs both of them a cross-wave reduction and -- unless the problem accumulates
// with atomics -- a 16 KB round trip through memory on the critical path of the finisher (measured: ~6 us): worth it for
// a tile with a long reduce range (>= PK_CUT_MIN stages) cut into pieces of >= PK_PIECE_MIN stages, not for the
// K = 256 tiles (2 stages), which stay whole: their boundary goes to the nearer end of the tile.
// Shares are equal in WEIGHTED units: a tile costs its workgroup a fixed ~c0 stages' worth of time whatever its reduce
// range (first loads, cross-wave reduction, epilogue: per-tile launches of the step fit t = 7.9 us + 2.8 us x stages at
// two workgroups per CU), so a run of S stages weighs S + c0 -- without it the workgroups that got the short-K
// tiles of a mixed call (dX 4864 x 1024 x 256 beside its split-K dW) ran twice as long as the others.
constexpr int PK_CUT_MIN = 8, PK_PIECE_MIN = 2;
__device__ __forceinline__ int pk_unit_begin(const char* ka, int n, int pos, int W, int G) {
  const int raw = (int)(((int64_t)pos * W) / G);
  if (raw <= 0 || raw >= W) return raw < 0 ? 0 : (raw > W ? W : raw);
  int gi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k) {
    const int st = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, ustart) + k * sizeof(int));
    if (k < n && raw >= st) gi = k;
  }
  const int first = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, ustart) + (size_t)gi * sizeof(int));
  const int S = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, S) + (size_t)gi * sizeof(int));
  const int cut = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, cut) + (size_t)gi * sizeof(int));
  const int c0 = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, c0));
  const int V = S + c0;
  const int o = (raw - first) % V, rs = raw - o;
  if (cut == 0) return 2 * o >= V ? rs + V : rs;          // whole tiles only
  if (o < c0 + PK_PIECE_MIN) return rs;                   // (the fixed cost goes with the piece that holds stage 0)
  if (V - o < PK_PIECE_MIN) return rs + V;
  return raw;
}

// the piece that starts at global (weighted) unit u (u < u1 <= W): a run of S stages occupies V = S + c0 units, the
// first c0 of them standing for the tile's fixed cost (they belong to the piece that holds stage 0)
__device__ __forceinline__ Piece pk_decode(const char* ka, int n, int u, int u1) {
  Piece pc;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < GROUP_MAX; ++k) {
    const int st = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, ustart) + k * sizeof(int));
    if (k < n && u >= st) gi = k;
  }
  const int first = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, ustart) + (size_t)gi * sizeof(int));
  const int S = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, S) + (size_t)gi * sizeof(int));
  const int c0 = *reinterpret_cast<const int*>(ka + offsetof(PkArgs, c0));
  const int V = S + c0;
  const int local = u - first;
  pc.gi = gi;
  pc.S = S;
  pc.V = V;
  pc.unit = local / V;
  const int o0 = local - pc.unit * V;
  const int room = V - o0, want = u1 - u;
  pc.vlen = room < want ? room : want;
  pc.s0 = o0 > c0 ? o0 - c0 : 0;
  const int s1 = o0 + pc.vlen > c0 ? o0 + pc.vlen - c0 : 0;
  pc.len = s1 - pc.s0;
  pc.whole = o0 == 0 && pc.vlen == V;
  pc.u = u;
  return pc;
}

struct PieceGeom {
  int m0, n0, z, by;
  int KM;
  int ka, kb;      // reduce range of the piece
  int k0, k1, nst; // ... of this wave
  bool first;      // the piece holds the tile's reduce index 0 (bias, residual, K tail, column-sum tail ride with it)
};

__device__ __forceinline__ PieceGeom pk_geom(const MesmGemmArgs& p, const Piece& pc, int wave) {
  PieceGeom g;
  const int mt = (p.M + 63) / 64, nt = (p.N + 63) / 64, T = mt * nt;
  g.z = pc.unit / T;
  const int t = pc.unit - g.z * T;
  int bx, by;
  if (nt > mt) {  // walk the LONGER tile axis slowest: a contiguous range then re-reads the smaller operand only
    by = t / mt;
    bx = t - by * mt;
  } else {
    bx = t / nt;
    by = t - bx * nt;
  }
  g.m0 = bx * 64;
  g.n0 = by * 64;
  g.by = by;
  g.KM = gemm_kmain(p);
  int kbeg = 0, kend = g.KM;
  if (p.split_k > 1) {
    int chunk = (p.K + p.split_k - 1) / p.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    kbeg = g.z * chunk;
    kend = kbeg + chunk < g.KM ? kbeg + chunk : g.KM;
    if (kbeg > g.KM) kbeg = g.KM;
  }
  g.ka = kbeg + pc.s0 * PK_MACRO;
  g.ka = g.ka < kend ? g.ka : kend;
  g.kb = kbeg + (pc.s0 + pc.len) * PK_MACRO;
  g.kb = g.kb < kend ? g.kb : kend;
  const int kw = (((g.kb - g.ka + 3) >> 2) + 31) & ~31;
  g.k0 = g.ka + wave * kw;
  g.k1 = g.k0 + kw < g.kb ? g.k0 + kw : g.kb;
  g.nst = g.k1 > g.k0 ? (g.k1 - g.k0 + 31) >> 5 : 0;
  g.first = g.z == 0 && pc.s0 == 0;
  return g;
}

__device__ __forceinline__ void pk_issue_rt(int layout, const float* base, int64_t ld, int o0, int extent, int kb, int k1,
                                            float* slab, int lane) {
  if (layout == MESM_LAYOUT_REDUCE_CONTIG) ws_issue<MESM_LAYOUT_REDUCE_CONTIG>(base, ld, o0, extent, kb, k1, slab, lane);
  else ws_issue<MESM_LAYOUT_OUTER_CONTIG>(base, ld, o0, extent, kb, k1, slab, lane);
}

// stage 0 of a piece into this wave's slabs (the problem's arguments come from the kernarg segment: scalar loads)
__device__ __forceinline__ void pk_issue_first(const char* ka, const Piece& pc, float* mine, int wave, int lane) {
  const MesmGemmArgs p = *reinterpret_cast<const MesmGemmArgs*>(ka + offsetof(PkArgs, p) + (size_t)pc.gi * sizeof(MesmGemmArgs));
  const PieceGeom g = pk_geom(p, pc, wave);
  if (g.nst > 0) {
    pk_issue_rt(p.a_layout, p.A, p.lda, g.m0, p.M, g.k0, g.k1, mine, lane);
    pk_issue_rt(p.a_layout, p.A, p.lda, g.m0 + 32, p.M, g.k0, g.k1, mine + WS_SLAB, lane);
    pk_issue_rt(p.b_layout, p.B, p.ldb, g.n0, p.N, g.k0, g.k1, mine + 2 * WS_SLAB, lane);
    pk_issue_rt(p.b_layout, p.B, p.ldb, g.n0 + 32, p.N, g.k0, g.k1, mine + 3 * WS_SLAB, lane);
  }
}

// ---- cross-workgroup payload: 16 bytes per lane, coherent at the memory side
typedef float f32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pk_store16(float* dst, f32x4 v) {
#if MESM_PK_FENCE
  *reinterpret_cast<f32x4*>(dst) = v;
#else
  asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" : : "v"(dst), "v"(v) : "memory");
#endif
}
// the four 1 KB rows a wave owns of a partial tile: issued together and WAITED FOR inside one asm statement (outputs of
// an asm load are not ready when the statement ends; the compiler may move them before a separate s_waitcnt)
__device__ __forceinline__ void pk_load_partial(const float* src, f32x4 (&v)[4]) {
#if MESM_PK_FENCE
#pragma unroll
  for (int r4 = 0; r4 < 4; ++r4) v[r4] = *reinterpret_cast<const f32x4*>(src + r4 * 256);
#else
  asm volatile(
      "global_load_dwordx4 %0, %4, off sc0 sc1\n\t"
      "global_load_dwordx4 %1, %4, off offset:1024 sc0 sc1\n\t"
      "global_load_dwordx4 %2, %4, off offset:2048 sc0 sc1\n\t"
      "global_load_dwordx4 %3, %4, off offset:3072 sc0 sc1\n\t"
      "s_waitcnt vmcnt(0)"
      : "=&v"(v[0]), "=&v"(v[1]), "=&v"(v[2]), "=&v"(v[3])
      : "v"(src)
      : "memory");
#endif
}

// One piece, start to finish.  Its stage 0 is already in flight in the slabs.  `next` (has_next) is the workgroup's
// following piece: its stage 0 is issued as soon as the slabs are free.
template <int LA, int LB, int BF>
__device__ __forceinline__ void pk_piece(const char* ka, const Piece& pc, int n, int u1, float* L, float* sh4, int pos) {
  // the problem's arguments are read from the kernarg segment TWICE: what the k loop needs (operand pointers, extents)
  // before it, everything again behind it -- one copy held across the loop costs ~60 scalar registers, whose spills
  // push the 256-VGPR budget of two workgroups per CU into scratch
  const MesmGemmArgs pl = *reinterpret_cast<const MesmGemmArgs*>(ka + offsetof(PkArgs, p) + (size_t)pc.gi * sizeof(MesmGemmArgs));
  const MesmGemmArgs& p = pl;
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);  // (wave-uniform: scalar registers)
  const int li = lane & 31, h = lane >> 5;
  const PieceGeom g = pk_geom(p, pc, wave);
  const int m0 = g.m0, n0 = g.n0, k0 = g.k0, k1 = g.k1, nst = g.nst;
  float* mine = L + wave * (4 * WS_SLAB);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.0f;
  float csum[2] = {0.0f, 0.0f};
  const bool do_colsum = (p.colsum != nullptr) && (g.by == 0);

  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float a[2][4][4], b[2][4][4];
    ws_read<LA>(mine, li, h, a[0]);
    ws_read<LA>(mine + WS_SLAB, li, h, a[1]);
    ws_read<LB>(mine + 2 * WS_SLAB, li, h, b[0]);
    ws_read<LB>(mine + 3 * WS_SLAB, li, h, b[1]);
    const int kb = k0 + 32 * st;
    if (st + 1 < nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");  // fragments are in registers: refill the slabs
      const int kn = kb + 32;
      ws_issue<LA>(p.A, p.lda, m0, p.M, kn, k1, mine, lane);
      ws_issue<LA>(p.A, p.lda, m0 + 32, p.M, kn, k1, mine + WS_SLAB, lane);
      ws_issue<LB>(p.B, p.ldb, n0, p.N, kn, k1, mine + 2 * WS_SLAB, lane);
      ws_issue<LB>(p.B, p.ldb, n0 + 32, p.N, kn, k1, mine + 3 * WS_SLAB, lane);
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
    if (do_colsum) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_)
#pragma unroll
          for (int j = 0; j < 4; ++j) csum[t] += a[t][s_][j];
    }
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  const char* kb_ = ka;
  asm volatile("" : "+s"(kb_));  // (a second, independent read of the arguments: see the top of the function)
  const MesmGemmArgs pe = *reinterpret_cast<const MesmGemmArgs*>(kb_ + offsetof(PkArgs, p) + (size_t)pc.gi * sizeof(MesmGemmArgs));
  const int W = *reinterpret_cast<const int*>(kb_ + offsetof(PkArgs, W));
  float* ws = *reinterpret_cast<float* const*>(kb_ + offsetof(PkArgs, ws));
  unsigned* flags = *reinterpret_cast<unsigned* const*>(kb_ + offsetof(PkArgs, flags));
  unsigned* status = *reinterpret_cast<unsigned* const*>(kb_ + offsetof(PkArgs, status));
  const int G = gridDim.x;

  XForm xa = {}, xb = {};
  if (do_colsum) {
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      float c = add_xor32(csum[t]);
      const int gm = m0 + 32 * t + li;
      if (wave == 0 && g.first && g.KM < pe.K) c += tail_colsum<LA, false>(pe, gm, g.KM, xa);
      if (h == 0 && gm < pe.M && c != 0.0f) atomicAdd(pe.colsum + gm, c);
    }
  }
  __syncthreads();  // every wave is done with its slabs: the reduction buffer aliases them
  // the four partial 64 x 64 tiles meet in LDS; wave w then owns sub-tile (w >> 1, w & 1)
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
  __syncthreads();  // the slabs are free again
  if (pc.u + pc.vlen < u1) pk_issue_first(ka, pk_decode(ka, n, pc.u + pc.vlen, u1), mine, wave, lane);

  const float slope = pe.slope ? *pe.slope : 0.0f;
  const uint32_t seed_off = pe.seed_offset ? *pe.seed_offset : 0u;
  const bool whole = pc.whole;
  if (pe.accumulate == 2 || whole) {
    // atomics: every piece adds its share (the first-split terms ride with the piece that holds reduce index 0)
    MesmGemmArgs q = pe;
    if (!whole && q.split_k <= 1) q.split_k = 2;  // (tile16_epilogue: first_split = split_k <= 1 || bz == 0)
    tile16_epilogue<LA, LB, false>(q, sum, m0 + 32 * (wave >> 1), n0 + 32 * (wave & 1), slope, seed_off, g.first ? 0 : 1, sh4,
                                   pc.unit, g.KM, xa, xb);
    return;
  }
  if (pc.s0 > 0) {
    // contributor: the partial tile goes to this workgroup's slot, then the flag
    float* slot = ws + (size_t)pos * PK_SLOT;
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4)
      pk_store16(slot + ((wave * 4 + r4) * 64 + lane) * 4, f32x4{sum[4 * r4], sum[4 * r4 + 1], sum[4 * r4 + 2], sum[4 * r4 + 3]});
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
#if MESM_PK_FENCE
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
      __hip_atomic_store(flags + (size_t)pos * PK_FLAG_STRIDE, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return;
  }
  // finisher: the tile's other pieces belong to the workgroups at the following positions
  const int uend = pc.u + pc.V;  // first unit behind this (tile, slice): a finisher starts at its run's first unit
  const int n_ = *reinterpret_cast<const int*>(kb_ + offsetof(PkArgs, n));
  int qb = pk_unit_begin(kb_, n_, pos + 1, W, G);
  for (int q = pos + 1; q < G && qb < uend; ++q) {
    const int qe = pk_unit_begin(kb_, n_, q + 1, W, G);
    const bool empty = qe == qb;  // (a position whose share was snapped away owes nothing)
    qb = qe;
    if (empty) continue;
    unsigned* flag = flags + (size_t)q * PK_FLAG_STRIDE;
    if (tid == 0) {
      unsigned spins = 0;
      while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
        if (++spins > PK_SPIN_LIMIT) {
          __hip_atomic_store(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(4);
      }
#if MESM_PK_FENCE
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    }
    __syncthreads();
    const float* slot = ws + (size_t)q * PK_SLOT;
    f32x4 v[4];
    pk_load_partial(slot + (wave * 4 * 64 + lane) * 4, v);
#pragma unroll
    for (int r4 = 0; r4 < 4; ++r4) {
      sum[4 * r4] += v[r4].x; sum[4 * r4 + 1] += v[r4].y; sum[4 * r4 + 2] += v[r4].z; sum[4 * r4 + 3] += v[r4].w;
    }
    __syncthreads();  // every wave has its share: the flag can go down for the next launch
    if (tid == 0) __hip_atomic_store(flag, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  tile16_epilogue<LA, LB, false>(pe, sum, m0 + 32 * (wave >> 1), n0 + 32 * (wave & 1), slope, seed_off, 0, sh4, pc.unit, g.KM,
                                 xa, xb);
}

template <int BF>
__global__ __launch_bounds__(NTHREADS, 2) void gemm_pk_kernel(const PkArgs g, const SideRed sr) {
  side_reduce(sr);
  __shared__ __attribute__((aligned(16))) float L[4 * 4 * WS_SLAB];  // 4 waves x 4 slabs = 64 KB
  __shared__ float sh4[8];
  const int G = gridDim.x, bid = blockIdx.x;
  const int pos = (bid & 7) * (G >> 3) + (bid >> 3);  // an XCD's workgroups own one contiguous range (G % 8 == 0)
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const int W = g.W, n = g.n;
  const int u1 = pk_unit_begin(ka, n, pos + 1, W, G);
  int u = pk_unit_begin(ka, n, pos, W, G);
  if (u >= u1) return;
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  pk_issue_first(ka, pk_decode(ka, n, u, u1), L + wave * (4 * WS_SLAB), wave, lane);
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG, O = MESM_LAYOUT_OUTER_CONTIG;
  while (u < u1) {
    const Piece cur = pk_decode(ka, n, u, u1);
    const char* pp = ka + offsetof(PkArgs, p) + (size_t)cur.gi * sizeof(MesmGemmArgs);
    const int la = *reinterpret_cast<const int*>(pp + offsetof(MesmGemmArgs, a_layout));
    const int lb = *reinterpret_cast<const int*>(pp + offsetof(MesmGemmArgs, b_layout));
    const int sel = (la == O ? 2 : 0) + (lb == O ? 1 : 0);
    if (sel == 0) pk_piece<R, R, BF>(ka, cur, n, u1, L, sh4, pos);
    else if (sel == 1) pk_piece<R, O, BF>(ka, cur, n, u1, L, sh4, pos);
    else if (sel == 2) pk_piece<O, R, BF>(ka, cur, n, u1, L, sh4, pos);
    else pk_piece<O, O, BF>(ka, cur, n, u1, L, sh4, pos);
    u += cur.vlen;
  }
}

// ------------------------------------------------------------------------------------------------ host side
struct PkState {
  float* ws = nullptr;
  unsigned* flags = nullptr;
  unsigned* status = nullptr;
  int grid = 0;      // 0: not initialised; < 0: unavailable on this device
  int max_grid = 0;  // what the occupancy query admits (all workgroups of a launch must be resident)
};
PkState g_pk[16];

// MESM_GEMM_PK: 1 (default) = grouped split-bf16 launches run as the persistent kernel, 0 = one workgroup per tile
int g_pk_on = []() { const char* e = getenv("MESM_GEMM_PK"); return e ? atoi(e) : 1; }();
// MESM_GEMM_PK_GRID: workgroups of the persistent grid (default: 2 per CU)
int g_pk_grid = []() { const char* e = getenv("MESM_GEMM_PK_GRID"); return e ? atoi(e) : 0; }();

// MESM_GEMM_PK_CUT: tiles of problems with at least this many 128-deep stages may be cut between workgroups (default 8)
int g_pk_cut_min = []() { const char* e = getenv("MESM_GEMM_PK_CUT"); return e ? atoi(e) : PK_CUT_MIN; }();

// MESM_GEMM_PK_C0: fixed cost of a tile in stages (the weight of a run of S stages is S + c0)
int g_pk_c0 = []() { const char* e = getenv("MESM_GEMM_PK_C0"); return e ? atoi(e) : 3; }();

PkState* pk_state(hipStream_t s) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return nullptr;
  PkState& st = g_pk[dev];
  if (st.grid != 0) return st.grid > 0 ? &st : nullptr;
  hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) {
    (void)hipGetLastError();
    return nullptr;  // no allocation under capture: the caller takes the one-workgroup-per-tile launch this time
  }
  int per_cu = 0, cus = 0;
  hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, gemm_pk_kernel<6>, NTHREADS, 0);
  if (e == hipSuccess) e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
  int grid = e == hipSuccess ? (per_cu < 2 ? per_cu : 2) * cus : 0;
  grid = (grid < PK_GRID_MAX ? grid : PK_GRID_MAX) & ~7;
  if (grid < 8) {
    (void)hipGetLastError();
    st.grid = -1;
    return nullptr;
  }
  void* base = nullptr;
  const size_t ws_bytes = (size_t)PK_GRID_MAX * PK_SLOT * sizeof(float);
  const size_t fl_bytes = (size_t)(PK_GRID_MAX + 1) * PK_FLAG_STRIDE * sizeof(unsigned);
  if (hipMalloc(&base, ws_bytes + fl_bytes) != hipSuccess || hipMemset(base, 0, ws_bytes + fl_bytes) != hipSuccess ||
      hipDeviceSynchronize() != hipSuccess) {
    (void)hipGetLastError();
    st.grid = -1;
    return nullptr;
  }
  st.ws = reinterpret_cast<float*>(base);
  st.flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(base) + ws_bytes);
  st.status = st.flags + (size_t)PK_GRID_MAX * PK_FLAG_STRIDE;
  st.grid = grid;
  st.max_grid = grid;
  if (g_pk_grid >= 8 && g_pk_grid <= grid) st.grid = g_pk_grid & ~7;
  return &st;
}

}  // namespace

// units of one problem: (tile, k-slice) runs of S stages
static void pk_units(const MesmGemmArgs& a, int& S, int64_t& units, int64_t& slots) {
  // mirrors gemm_kmain / pk_geom
  const bool a_red = a.a_layout == MESM_LAYOUT_REDUCE_CONTIG, b_red = a.b_layout == MESM_LAYOUT_REDUCE_CONTIG;
  int km = a.K;
  if (a_red || b_red) km &= ~3;
  if ((!a_red && (a.M & 3)) || (!b_red && (a.N & 3))) {
    const int lim = (a.K - 1) & ~3;
    km = km < lim ? km : lim;
  }
  if (km < 0) km = 0;
  const int64_t T = (int64_t)((a.M + 63) / 64) * ((a.N + 63) / 64);
  int Z = 1, range = km;
  if (a.split_k > 1) {
    int chunk = (a.K + a.split_k - 1) / a.split_k;
    chunk = ((chunk + BK_MAX - 1) / BK_MAX) * BK_MAX;
    Z = km > 0 ? (km + chunk - 1) / chunk : 1;
    if (Z > a.split_k) Z = a.split_k;
    range = chunk < km ? chunk : km;
  }
  S = (range + PK_MACRO - 1) / PK_MACRO;
  if (S < 1) S = 1;
  slots = T * Z;
  units = T * Z * S;
}

// the grouped split-bf16 launch as ONE persistent kernel; returns MESM_OK, an error, or 1 = "not available, use the
// per-tile launch" (first call under stream capture, device without room for the grid, switched off)
int mesm_gemm_pk_launch(const MesmGemmArgs* list, int n, const void* side_red, hipStream_t s, int64_t* dslope_slots) {
  if (!g_pk_on || n < 1 || n > GROUP_MAX) return 1;
  PkState* st = pk_state(s);
  if (!st) return 1;
  PkArgs g;
  g.n = n;
  g.ustart[0] = 0;
  for (int i = 0; i < n; ++i) {
    g.p[i] = list[i];
    int64_t units = 0, slots = 0;
    pk_units(list[i], g.S[i], units, slots);
    units = units / g.S[i] * (g.S[i] + g_pk_c0);
    if (g.ustart[i] + units > (int64_t)1 << 30) return 1;
    g.ustart[i + 1] = g.ustart[i] + (int)units;
    g.cut[i] = list[i].accumulate == 2 ? 2 : (g.S[i] >= g_pk_cut_min ? 1 : 0);
    if (dslope_slots) dslope_slots[i] = slots;
  }
  for (int i = n; i < GROUP_MAX; ++i) {
    g.ustart[i + 1] = g.ustart[n];
    g.S[i] = 1;
    g.cut[i] = 0;
  }
  g.W = g.ustart[n];
  g.c0 = g_pk_c0;
  g.ws = st->ws;
  g.flags = st->flags;
  g.status = st->status;
  int grid = st->grid;
  const int runs_w = g.W / (1 + g_pk_c0);
  if (runs_w < grid) grid = (runs_w & ~7) < 8 ? 8 : (runs_w & ~7);  // (no more workgroups than one-stage tiles)
  SideRed sr;
  memcpy(&sr, side_red, sizeof(sr));
  hipLaunchKernelGGL(gemm_pk_kernel<6>, dim3(grid), dim3(NTHREADS), 0, s, g, sr);
  return mesm_launch_status();
}

// 0 = every finisher of every launch so far found its contributors; reading it synchronises the device
extern "C" int mesm_gemm_pk_status(void) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16 || g_pk[dev].grid <= 0) return 0;
  unsigned v = 0;
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(&v, g_pk[dev].status, sizeof(v), hipMemcpyDeviceToHost) != hipSuccess)
    return MESM_ELAUNCH;
  return (int)v;
}

// tuning tools: switch the persistent form on / off, set its grid (0 = keep) between calls of one process
extern "C" int mesm_gemm_pk_set(int32_t on, int32_t grid, int32_t cut_min, int32_t c0) {
  if (on >= 0) g_pk_on = on;
  if (cut_min > 0) g_pk_cut_min = cut_min;
  if (c0 >= 0) g_pk_c0 = c0;
  if (grid > 0) {
    g_pk_grid = grid;
    for (auto& st : g_pk)
      if (st.grid > 0) st.grid = ((grid < st.max_grid ? grid : st.max_grid) & ~7) < 8 ? 8 : ((grid < st.max_grid ? grid : st.max_grid) & ~7);
  }
  return MESM_OK;
}
