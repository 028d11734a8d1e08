// Assembly kernels of MESM.forward: the places where the reference concatenates, repeats, selects and masks
// tensors between the transformer blocks (model/model.py:184-207, 260-299, 307-325; model/transformer.py:174-205)
// and where autograd would otherwise add the gradients of the pieces back together.  Each of these was 2-8
// element-wise / copy launches per site (~220 per step, 12 % of the step for no arithmetic); here every site is
// ONE launch forward and ONE backward, and the gradient sums happen while the data moves.
// All HBM-bound byte movers: rows are copied with 16-byte accesses, one workgroup per (row, tensor).
#include "common.hpp"

namespace {

// ---------------------------------------------------------------------------------------------------------
// stack_rows: dst (2N rows) = [ src ; src[idx] ]  for up to 8 tensors of any row size in one launch
// (idx == NULL: the second half repeats the first).  The positive and the negative pass of the model run
// stacked along the batch: rows [0, N) = the pairs, rows [N, 2N) = every pair's video with the
// words / mask of its negative query (model.py:260-299, neg_index from sample_outclass_neg).
constexpr int STACK_MAX = 8;
struct StackArgs {
  const unsigned char* src[STACK_MAX];
  unsigned char* dst[STACK_MAX];
  int64_t row_bytes[STACK_MAX];
  int gather[STACK_MAX];  // second half: 1 = src[idx[j]], 0 = src[j]
  const int64_t* idx;
  int N, n;
};

__global__ __launch_bounds__(256) void stack_rows_kernel(const StackArgs a) {
  const int t = blockIdx.y, r = blockIdx.x;
  int sr = r < a.N ? r : r - a.N;
  if (r >= a.N && a.gather[t]) sr = (int)a.idx[r - a.N];
  const int64_t rb = a.row_bytes[t];
  const unsigned char* s = a.src[t] + (int64_t)sr * rb;
  unsigned char* d = a.dst[t] + (int64_t)r * rb;
  if ((rb & 15) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
    const int64_t n16 = rb >> 4;
    for (int64_t i = threadIdx.x; i < n16; i += 256) mesm_store_wt16(reinterpret_cast<uint4*>(d) + i, reinterpret_cast<const uint4*>(s)[i]);
  } else {
    for (int64_t i = threadIdx.x; i < rb; i += 256) d[i] = s[i];
  }
}

// backward of one float tensor: dx[i, :] = d2[i, :] + sum over j with idx[j] == i of d2[N + j, :]
// (idx == NULL: j == i).  Deterministic: every output row gathers its contributors (N <= a few hundred).
__device__ __forceinline__ void unstack_rows_body(const float* __restrict__ d2, const int64_t* __restrict__ idx,
                                                  float* __restrict__ dx, int N, int64_t R, int bx, int by) {
  const int i = bx;
  const int64_t c0 = ((int64_t)by * 256 + threadIdx.x) * 4;
  if (c0 >= R) return;
  float4 acc = *reinterpret_cast<const float4*>(d2 + (int64_t)i * R + c0);
  if (idx == nullptr) {
    const float4 b = *reinterpret_cast<const float4*>(d2 + (int64_t)(N + i) * R + c0);
    acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
  } else {
    for (int j = 0; j < N; ++j)
      if ((int)idx[j] == i) {
        const float4 b = *reinterpret_cast<const float4*>(d2 + (int64_t)(N + j) * R + c0);
        acc.x += b.x; acc.y += b.y; acc.z += b.z; acc.w += b.w;
      }
  }
  mesm_store_wt4(dx + (int64_t)i * R + c0, acc);
}

__global__ __launch_bounds__(256) void unstack_rows_kernel(const float* __restrict__ d2, const int64_t* __restrict__ idx,
                                                           float* __restrict__ dx, int N, int64_t R) {
  unstack_rows_body(d2, idx, dx, N, R, blockIdx.x, blockIdx.y);
}

// ---------------------------------------------------------------------------------------------------------
// prepend_token: xo (B, L+1, D) = [ tok ; x ],  optionally po = [ ptok ; pos ] and xp = xo + po, and the key
// padding mask [ first ; pad ].  tok / ptok are one D-vector for all rows (tok_per_row = 0: the global token
// and its position embedding, transformer.py:185-188) or one per batch row (tok_per_row = 1: the reconstructed
// sentence token in front of the words, model.py:221-224).
__global__ __launch_bounds__(256) void prepend_fwd_kernel(const float* __restrict__ tok, const float* __restrict__ x,
                                                          const float* __restrict__ ptok, const float* __restrict__ pos,
                                                          const uint8_t* __restrict__ pad, float* __restrict__ xo,
                                                          float* __restrict__ po, float* __restrict__ xp,
                                                          uint8_t* __restrict__ pado, int L, int D, int tok_per_row,
                                                          int first_pad) {
  // (a wave per row, four rows per workgroup: one 64-thread workgroup per row was 4,928 workgroups for 5 MB)
  const int b = blockIdx.x, l = blockIdx.y * 4 + (threadIdx.x >> 6);  // l in [0, L]
  if (l > L) return;
  const int64_t orow = ((int64_t)b * (L + 1) + l) * D;
  const float* xs = l == 0 ? tok + (tok_per_row ? (int64_t)b * D : 0) : x + ((int64_t)b * L + l - 1) * D;
  const float* ps = nullptr;
  if (po || xp) ps = l == 0 ? ptok : pos + ((int64_t)b * L + l - 1) * D;
  for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(xs + c);
    mesm_store_wt4(xo + orow + c, v);
    if (ps) {
      const float4 q = *reinterpret_cast<const float4*>(ps + c);
      if (po) mesm_store_wt4(po + orow + c, q);
      if (xp) mesm_store_wt4(xp + orow + c, make_float4(v.x + q.x, v.y + q.y, v.z + q.z, v.w + q.w));
    }
  }
  if (pado && (threadIdx.x & 63) == 0) pado[(int64_t)b * (L + 1) + l] = l == 0 ? (uint8_t)first_pad : pad[(int64_t)b * L + l - 1];
}

// backward: g = dxo (+ dxp) [+ dpo for the position outputs]
//   dx[b, l]   = g[b, 1 + l]                               (rows of the sequence)
//   dtok       (+)= g[b, 0]          per row: plain store;  shared: atomic add over b into the gradient view
//   dptok      (+)= (dxp + dpo)[b, 0]
__global__ __launch_bounds__(256) void prepend_bwd_kernel(const float* __restrict__ dxo, const float* __restrict__ dxp,
                                                          const float* __restrict__ dpo, float* __restrict__ dx,
                                                          float* __restrict__ dtok, float* __restrict__ dptok, int L, int D,
                                                          int tok_per_row) {
  const int b = blockIdx.x, l = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (l > L) return;
  if (l == 0 && !tok_per_row) {
    // the shared token's gradient: the wave of every 8th batch row sums the token rows of its 8 batch rows (loads in flight
    // together) and adds once per column -- an atomic per batch row was 64-way contention on 512 addresses, 8 of the
    // launch's 11 us; ONE wave walking all 64 rows was a 64-deep chain of loads, 50 us
    if (b & 7) return;
    const int B = (int)gridDim.x;
    for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
      float4 g = make_float4(0.f, 0.f, 0.f, 0.f), gp = g;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int bb = b + u;
        if (bb >= B) break;
        const int64_t o = (int64_t)bb * (L + 1) * D + c;
        if (dxo) { const float4 u = *reinterpret_cast<const float4*>(dxo + o); g.x += u.x; g.y += u.y; g.z += u.z; g.w += u.w; }
        if (dxp) {
          const float4 u = *reinterpret_cast<const float4*>(dxp + o);
          g.x += u.x; g.y += u.y; g.z += u.z; g.w += u.w;
          gp.x += u.x; gp.y += u.y; gp.z += u.z; gp.w += u.w;
        }
        if (dptok && dpo) { const float4 u = *reinterpret_cast<const float4*>(dpo + o); gp.x += u.x; gp.y += u.y; gp.z += u.z; gp.w += u.w; }
      }
      if (dtok) { atomicAdd(dtok + c, g.x); atomicAdd(dtok + c + 1, g.y); atomicAdd(dtok + c + 2, g.z); atomicAdd(dtok + c + 3, g.w); }
      if (dptok) { atomicAdd(dptok + c, gp.x); atomicAdd(dptok + c + 1, gp.y); atomicAdd(dptok + c + 2, gp.z); atomicAdd(dptok + c + 3, gp.w); }
    }
    return;
  }
  const int64_t orow = ((int64_t)b * (L + 1) + l) * D;
  for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), gp = g;
    if (dxo) g = *reinterpret_cast<const float4*>(dxo + orow + c);
    if (dxp) {
      gp = *reinterpret_cast<const float4*>(dxp + orow + c);
      g.x += gp.x; g.y += gp.y; g.z += gp.z; g.w += gp.w;
    }
    if (l > 0) {
      if (dx) mesm_store_wt4(dx + ((int64_t)b * L + l - 1) * D + c, g);
    } else {
      if (dtok) {
        if (tok_per_row) {
          mesm_store_wt4(dtok + (int64_t)b * D + c, g);
        } else {
          atomicAdd(dtok + c, g.x); atomicAdd(dtok + c + 1, g.y); atomicAdd(dtok + c + 2, g.z); atomicAdd(dtok + c + 3, g.w);
        }
      }
      if (dptok) {
        if (dpo) {
          const float4 q = *reinterpret_cast<const float4*>(dpo + orow + c);
          gp.x += q.x; gp.y += q.y; gp.z += q.z; gp.w += q.w;
        }
        atomicAdd(dptok + c, gp.x); atomicAdd(dptok + c + 1, gp.y); atomicAdd(dptok + c + 2, gp.z); atomicAdd(dptok + c + 3, gp.w);
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------------
// split_token: mem (B, L+1, D) -> g (B, D) = mem[:, 0], loc (B, L, D) = mem[:, 1:], and dec (Bd, L, D) = loc[:Bd]
// (the decoder only sees the positive half, model.py:295).  Backward: dmem[b, 0] = dg[b];
// dmem[b, 1 + l] = dloc[b, l] + (b < Bd ? ddec[b, l] : 0); missing gradients count as zero.
__global__ __launch_bounds__(256) void split_fwd_kernel(const float* __restrict__ mem, float* __restrict__ g,
                                                        float* __restrict__ loc, float* __restrict__ dec, int L, int D,
                                                        int Bd) {
  const int b = blockIdx.x, l = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (l > L) return;
  const float* s = mem + ((int64_t)b * (L + 1) + l) * D;
  for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
    const float4 v = *reinterpret_cast<const float4*>(s + c);
    if (l == 0) {
      mesm_store_wt4(g + (int64_t)b * D + c, v);
    } else {
      mesm_store_wt4(loc + ((int64_t)b * L + l - 1) * D + c, v);
      if (dec && b < Bd) mesm_store_wt4(dec + ((int64_t)b * L + l - 1) * D + c, v);
    }
  }
}

__global__ __launch_bounds__(256) void split_bwd_kernel(const float* __restrict__ dg, const float* __restrict__ dloc,
                                                        const float* __restrict__ ddec, float* __restrict__ dmem, int L,
                                                        int D, int Bd) {
  const int b = blockIdx.x, l = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (l > L) return;
  float* d = dmem + ((int64_t)b * (L + 1) + l) * D;
  for (int c = (threadIdx.x & 63) * 4; c < D; c += 256) {
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (l == 0) {
      if (dg) v = *reinterpret_cast<const float4*>(dg + (int64_t)b * D + c);
    } else {
      const int64_t o = ((int64_t)b * L + l - 1) * D + c;
      if (dloc) v = *reinterpret_cast<const float4*>(dloc + o);
      if (ddec && b < Bd) {
        const float4 w = *reinterpret_cast<const float4*>(ddec + o);
        v.x += w.x; v.y += w.y; v.z += w.z; v.w += w.w;
      }
    }
    mesm_store_wt4(d + c, v);
  }
}

// ---------------------------------------------------------------------------------------------------------
// token_mix: y[r, :] = m2[r] ? tok2 : (m1[r] ? tok1 : x[r, :])  -- MESM._replace_unknown + _mask_words
// (model.py:361-394: unknown words, then the drawn MLM positions, are replaced by learned tokens) and
// SegSenRecon's masked sentence slot (model.py:493-501).  m2 / tok2 optional.  Backward: dx = dy where neither
// mask is set, else 0; dtok1 += sum of dy over (m1 & ~m2) rows, dtok2 += sum over m2 rows (atomics, one
// partial per workgroup of 32 rows).
constexpr int TM_ROWS = 8;
__device__ __forceinline__ void token_mix_fwd_body(const float* __restrict__ x, const uint8_t* __restrict__ m1,
                                                   const float* __restrict__ tok1, const uint8_t* __restrict__ m2,
                                                   const float* __restrict__ tok2, float* __restrict__ y,
                                                   int64_t rows, int D, int bx) {
  const int64_t r = (int64_t)bx * 4 + (threadIdx.x >> 6);
  if (r >= rows) return;
  const float* s = x + r * D;
  if (m2 && m2[r]) s = tok2;
  else if (m1[r]) s = tok1;
  for (int c = (threadIdx.x & 63) * 4; c < D; c += 256)
    mesm_store_wt4(y + r * D + c, *reinterpret_cast<const float4*>(s + c));
}

__global__ __launch_bounds__(256) void token_mix_fwd_kernel(const float* __restrict__ x, const uint8_t* __restrict__ m1,
                                                            const float* __restrict__ tok1, const uint8_t* __restrict__ m2,
                                                            const float* __restrict__ tok2, float* __restrict__ y,
                                                            int64_t rows, int D) {
  token_mix_fwd_body(x, m1, tok1, m2, tok2, y, rows, D, blockIdx.x);
}

__device__ __forceinline__ void token_mix_bwd_body(const float* __restrict__ dy, const uint8_t* __restrict__ m1,
                                                   const uint8_t* __restrict__ m2, float* __restrict__ dx,
                                                   float* __restrict__ dtok1, float* __restrict__ dtok2,
                                                   int64_t rows, int D, int bx) {
  const int64_t r0 = (int64_t)bx * TM_ROWS;
  const int64_t r1 = (r0 + TM_ROWS) < rows ? (r0 + TM_ROWS) : rows;
  // the rows' mask bytes first, into LDS: read inside the loop they made every iteration wait for a global load
  __shared__ uint8_t mk1[TM_ROWS], mk2[TM_ROWS];
  if (threadIdx.x < TM_ROWS && r0 + threadIdx.x < r1) {
    mk1[threadIdx.x] = m1[r0 + threadIdx.x];
    mk2[threadIdx.x] = m2 ? m2[r0 + threadIdx.x] : 0;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 256) {
    float s1 = 0.0f, s2 = 0.0f;
    float gv[TM_ROWS];  // the workgroup's rows of this column, all in flight together
#pragma unroll
    for (int u = 0; u < TM_ROWS; ++u) gv[u] = r0 + u < r1 ? dy[(r0 + u) * D + c] : 0.0f;
#pragma unroll
    for (int u = 0; u < TM_ROWS; ++u) {
      if (r0 + u < r1) {
        const float g = gv[u];
        const bool b2 = mk2[u] != 0, b1 = mk1[u] != 0;
        if (dx) dx[(r0 + u) * D + c] = (b1 || b2) ? 0.0f : g;
        if (b2) s2 += g;
        else if (b1) s1 += g;
      }
    }
    if (dtok1 && s1 != 0.0f) atomicAdd(dtok1 + c, s1);
    if (dtok2 && s2 != 0.0f) atomicAdd(dtok2 + c, s2);
  }
}

__global__ __launch_bounds__(256) void token_mix_bwd_kernel(const float* __restrict__ dy, const uint8_t* __restrict__ m1,
                                                            const uint8_t* __restrict__ m2, float* __restrict__ dx,
                                                            float* __restrict__ dtok1, float* __restrict__ dtok2,
                                                            int64_t rows, int D) {
  token_mix_bwd_body(dy, m1, m2, dx, dtok1, dtok2, rows, D, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------
// gather_rows: y[j, :] = valid[j] ? x[idx[j], :] : 0 (valid == NULL: all), optionally L2-normalised with
// F.normalize's eps (norm = 1: y / max(||y||, 1e-12), the reconstructed sentence token model.py:486).
// Gathers the ground-truth clips for the MLM branch (model.py:312-325: projed_video_feat[clip_mask] re-padded)
// and the masked slot of every pair (model.py:485).  Backward scatters through the INVERSE map built on the
// host (inv[i] = j with idx[j] == i, or -1): dx[i, :] = inv[i] >= 0 ? dy-through-the-normalisation : 0, every
// source row written exactly once -- no zero fill, no atomics.  (Rows gathered more than once are not
// supported by this kernel; those sites keep index_add.)
__device__ __forceinline__ void gather_rows_fwd_body(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                     const uint8_t* __restrict__ valid, float* __restrict__ y,
                                                     float* __restrict__ rnorm, int64_t rows, int D, int norm, int bx) {
  const int lane = threadIdx.x & 63;
  const int64_t j = (int64_t)bx * 4 + (threadIdx.x >> 6);
  if (j >= rows) return;
  const bool ok = valid == nullptr || valid[j] != 0;
  const float* s = x + idx[j] * D;
  float inv = 1.0f;
  if (norm) {
    float q = 0.0f;
    if (ok)
      for (int c = lane; c < D; c += 64) q += s[c] * s[c];
    const float nrm = fmaxf(sqrtf(wave_sum(q)), 1e-12f);
    inv = 1.0f / nrm;
    if (lane == 0 && rnorm) rnorm[j] = nrm;
  }
  for (int c = lane; c < D; c += 64) y[j * D + c] = ok ? s[c] * inv : 0.0f;
}

__global__ __launch_bounds__(256) void gather_rows_fwd_kernel(const float* __restrict__ x, const int64_t* __restrict__ idx,
                                                              const uint8_t* __restrict__ valid, float* __restrict__ y,
                                                              float* __restrict__ rnorm, int64_t rows, int D, int norm) {
  gather_rows_fwd_body(x, idx, valid, y, rnorm, rows, D, norm, blockIdx.x);
}

__device__ __forceinline__ void gather_rows_bwd_body(const float* __restrict__ dy, const float* __restrict__ y,
                                                     const float* __restrict__ rnorm, const int64_t* __restrict__ inv,
                                                     const uint8_t* __restrict__ valid, float* __restrict__ dx,
                                                     int64_t src_rows, int D, int norm, int bx) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)bx * 4 + (threadIdx.x >> 6);
  if (i >= src_rows) return;
  const int64_t j = inv[i];
  const bool ok = j >= 0 && (valid == nullptr || valid[j] != 0);
  if (!ok) {
    for (int c = lane; c < D; c += 64) dx[i * D + c] = 0.0f;
    return;
  }
  if (!norm) {
    for (int c = lane; c < D; c += 64) dx[i * D + c] = dy[j * D + c];
    return;
  }
  // y = x / n  (n = max(||x||, eps)): dx = (dy - y <dy, y>) / n   (for ||x|| < eps the clamp is constant: dx = dy / n)
  const float n = rnorm[j];
  float dot = 0.0f;
  for (int c = lane; c < D; c += 64) dot += dy[j * D + c] * y[j * D + c];
  dot = wave_sum(dot);
  if (n <= 1e-12f) dot = 0.0f;
  for (int c = lane; c < D; c += 64) dx[i * D + c] = (dy[j * D + c] - y[j * D + c] * dot) / n;
}

__global__ __launch_bounds__(256) void gather_rows_bwd_kernel(const float* __restrict__ dy, const float* __restrict__ y,
                                                              const float* __restrict__ rnorm, const int64_t* __restrict__ inv,
                                                              const uint8_t* __restrict__ valid, float* __restrict__ dx,
                                                              int64_t src_rows, int D, int norm) {
  gather_rows_bwd_body(dy, y, rnorm, inv, valid, dx, src_rows, D, norm, blockIdx.x);
}

// ---- backward of a Linear with at most 4 output features (the last layers of the box / class / anchor heads,
// transformer.py:395-398, model.py:118-119): dX = dz W (masked by x > 0 when x is a ReLU's output), dW += dz^T x,
// db += colsum(dz) in ONE launch.  As GEMMs these are a 1-2 wide product and a 1-2 deep one, each a launch of its own
// that nothing else could share (extent < 4 on the vectorised axis).  Workgroup = 8 rows x 256 columns, thread =
// column: dz of the rows sits in LDS, W's J rows in registers, the 8 rows of x in flight together.
constexpr int SK_ROWS = 8;
__global__ __launch_bounds__(256) void skinny_linear_bwd_kernel(const float* __restrict__ dz, const float* __restrict__ x,
                                                                const float* __restrict__ w, float* __restrict__ dx,
                                                                float* __restrict__ dw, float* __restrict__ db,
                                                                int64_t M, int K, int J, int relu_mask) {
  __shared__ float dzs[SK_ROWS][4];
  const int64_t r0 = (int64_t)blockIdx.x * SK_ROWS;
  const int nr = M - r0 < SK_ROWS ? (int)(M - r0) : SK_ROWS;
  const int c = blockIdx.y * 256 + threadIdx.x;
  if (threadIdx.x < SK_ROWS * 4) {
    const int r = threadIdx.x >> 2, j = threadIdx.x & 3;
    dzs[r][j] = (r < nr && j < J) ? dz[(r0 + r) * J + j] : 0.0f;
  }
  __syncthreads();
  if (c < K) {
    float wv[4], acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { wv[j] = j < J ? w[(int64_t)j * K + c] : 0.0f; acc[j] = 0.0f; }
    for (int rb = 0; rb < nr; rb += 8) {
      float xv[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) xv[u] = rb + u < nr ? x[(r0 + rb + u) * K + c] : 0.0f;
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (rb + u < nr) {
          float o = 0.0f;
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float d = dzs[rb + u][j];
            acc[j] += d * xv[u];
            o += d * wv[j];
          }
          if (dx) dx[(r0 + rb + u) * K + c] = (relu_mask && !(xv[u] > 0.0f)) ? 0.0f : o;
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j)
      if (j < J) atomicAdd(dw + (int64_t)j * K + c, acc[j]);
  }
  if (db && blockIdx.y == 0 && threadIdx.x < J) {
    float t = 0.0f;
    for (int r = 0; r < nr; ++r) t += dzs[r][threadIdx.x];
    atomicAdd(db + threadIdx.x, t);
  }
}


// out[i, :] = a[i, :] + b[i % rows_b, :]  (one float4 per thread): the first query of the enhance encoder,
// projected video + sine position embedding for both stacked passes (model.py:175-180, 281-286)
__global__ __launch_bounds__(256) void add_wrap_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                       float* __restrict__ out, int64_t n4, int64_t nb4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    const float4 u = reinterpret_cast<const float4*>(a)[i], v = reinterpret_cast<const float4*>(b)[i % nb4];
    reinterpret_cast<float4*>(out)[i] = make_float4(u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
  }
}

// out = src[0] + ... + src[k - 1], k <= 8 (one float4 per thread): the gradient of a tensor with several consumers in ONE
// launch instead of the autograd engine's k - 1 pairwise adds (a decoder layer's output feeds the box head, the next
// layer's anchor / scale heads, the final norm and the next layer twice: transformer.py:370-394)
struct AddNSrc {
  const float* p[8];
};
__global__ __launch_bounds__(256) void add_n_kernel(const AddNSrc s, int k, float* __restrict__ out, int64_t n4) {
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (int64_t)gridDim.x * 256) {
    float4 t = reinterpret_cast<const float4*>(s.p[0])[i];
#pragma unroll
    for (int j = 1; j < 8; ++j)
      if (j < k) {
        const float4 u = reinterpret_cast<const float4*>(s.p[j])[i];
        t.x += u.x; t.y += u.y; t.z += u.z; t.w += u.w;
      }
    reinterpret_cast<float4*>(out)[i] = t;
  }
}

// ---------------------------------------------------------------------------------------------------------
// Grouped launch: up to GLUE_MAX INDEPENDENT assembly problems of one launch phase -- the stacked copies, token mixes and
// clip gathers in front of the enhance stage (model.py:260-325) were six launches of ~5 us forward and as many backward
// -- as workgroup ranges of ONE grid.  Every member is a 256-thread byte mover; a problem's workgroups see their index
// inside the problem's own (gx, gy) grid.  The same device functions as the plain launches: identical results.
constexpr int GLUE_MAX = 16;
struct GlueGroup {
  MesmGlueArgs p[GLUE_MAX];
  int start[GLUE_MAX + 1];  // first workgroup of every problem
  int gx[GLUE_MAX];         // the problem's grid is (gx, (start[k + 1] - start[k]) / gx)
  int n;
};

__global__ __launch_bounds__(256) void glue_group_kernel(const GlueGroup g) {
  const int bid = blockIdx.x;
  int gi = 0;
#pragma unroll
  for (int k = 1; k < GLUE_MAX; ++k)
    if (k < g.n && bid >= g.start[k]) gi = k;
  // (read from the kernarg segment with a wave-uniform offset: indexing g.p[gi] would copy the array to scratch)
  const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
  const int first = *reinterpret_cast<const int*>(ka + offsetof(GlueGroup, start) + (size_t)gi * sizeof(int));
  const int gx = *reinterpret_cast<const int*>(ka + offsetof(GlueGroup, gx) + (size_t)gi * sizeof(int));
  const MesmGlueArgs& a = *reinterpret_cast<const MesmGlueArgs*>(ka + offsetof(GlueGroup, p) + (size_t)gi * sizeof(MesmGlueArgs));
  const int local = bid - first;
  const int by = local / gx, bx = local - by * gx;
  switch (a.op) {
    case MESM_GLUE_TOKEN_MIX_FWD:
      token_mix_fwd_body((const float*)a.p[0], (const uint8_t*)a.p[1], (const float*)a.p[2], (const uint8_t*)a.p[3],
                         (const float*)a.p[4], (float*)a.p[5], a.n[0], a.i[0], bx);
      break;
    case MESM_GLUE_TOKEN_MIX_BWD:
      token_mix_bwd_body((const float*)a.p[0], (const uint8_t*)a.p[1], (const uint8_t*)a.p[2], (float*)a.p[3], (float*)a.p[4],
                         (float*)a.p[5], a.n[0], a.i[0], bx);
      break;
    case MESM_GLUE_GATHER_ROWS_FWD:
      gather_rows_fwd_body((const float*)a.p[0], (const int64_t*)a.p[1], (const uint8_t*)a.p[2], (float*)a.p[3], (float*)a.p[4],
                           a.n[0], a.i[0], a.i[1], bx);
      break;
    case MESM_GLUE_GATHER_ROWS_BWD:
      gather_rows_bwd_body((const float*)a.p[0], (const float*)a.p[1], (const float*)a.p[2], (const int64_t*)a.p[3],
                           (const uint8_t*)a.p[4], (float*)a.p[5], a.n[0], a.i[0], a.i[1], bx);
      break;
    case MESM_GLUE_UNSTACK_ROWS:
      unstack_rows_body((const float*)a.p[0], (const int64_t*)a.p[1], (float*)a.p[2], a.i[0], a.n[0], bx, by);
      break;
    case MESM_GLUE_STACK_ROWS: {  // one tensor of a stack_rows call: dst (2N rows) = [src ; src[idx]] or [src ; src]
      const int N = a.i[0], r = bx;
      int sr = r < N ? r : r - N;
      if (r >= N && a.p[2]) sr = (int)((const int64_t*)a.p[2])[r - N];
      const int64_t rb = a.n[0];
      const unsigned char* s = (const unsigned char*)a.p[0] + (int64_t)sr * rb;
      unsigned char* d = (unsigned char*)a.p[1] + (int64_t)r * rb;
      if ((rb & 15) == 0 && ((((uintptr_t)s) | ((uintptr_t)d)) & 15) == 0) {
        const int64_t n16 = rb >> 4;
        for (int64_t i = threadIdx.x; i < n16; i += 256) mesm_store_wt16(reinterpret_cast<uint4*>(d) + i, reinterpret_cast<const uint4*>(s)[i]);
      } else {
        for (int64_t i = threadIdx.x; i < rb; i += 256) d[i] = s[i];
      }
      break;
    }
    case MESM_GLUE_ADD_TILE: {  // out[i] = a[i % na] + b[i % nb] (float4 units)
      const float4* pa = (const float4*)a.p[0];
      const float4* pb = (const float4*)a.p[1];
      float4* po = (float4*)a.p[2];
      const int64_t n4 = a.n[0], na4 = a.n[1], nb4 = ((const int64_t*)a.i)[0];
      for (int64_t i = (int64_t)bx * 256 + threadIdx.x; i < n4; i += (int64_t)gx * 256) {
        const float4 u = pa[i % na4], v = pb[i % nb4];
        mesm_store_wt4(reinterpret_cast<float*>(po + i), u.x + v.x, u.y + v.y, u.z + v.z, u.w + v.w);
      }
      break;
    }
    case MESM_GLUE_GATHER_ADD: {  // y[j] = valid[j] ? a[idx[j]] + b[idx[j]] : 0 (a wave per row)
      const float* pa = (const float*)a.p[0];
      const float* pb = (const float*)a.p[1];
      const int64_t* idx = (const int64_t*)a.p[2];
      const uint8_t* valid = (const uint8_t*)a.p[3];
      float* y = (float*)a.p[4];
      const int D = a.i[0];
      const int64_t j = (int64_t)bx * 4 + (threadIdx.x >> 6);
      if (j < a.n[0]) {
        const bool ok = valid == nullptr || valid[j] != 0;
        const int64_t r = idx[j];
        for (int c = threadIdx.x & 63; c < D; c += 64) y[j * D + c] = ok ? pa[r * D + c] + pb[r * D + c] : 0.0f;
      }
      break;
    }
    default:
      break;
  }
}

// The zero-initialised scratch of a step in ONE launch: up to FILL_MAX byte ranges (16-byte units) cleared together -- the
// pool of gradient tensors that kernels accumulate into, and the remainder ROWS of the 4800 / 4864-row GEMM outputs whose
// last tiles are split along K and meet by atomic adds (kernels.ZeroPool).
constexpr int FILL_MAX = 32;
struct FillRanges {
  uint4* p[FILL_MAX];
  int64_t n16[FILL_MAX];
};
__global__ __launch_bounds__(256) void fill_ranges_kernel(const FillRanges r) {
  uint4* d = r.p[0];
  int64_t n = r.n16[0];
#pragma unroll
  for (int k = 1; k < FILL_MAX; ++k)
    if ((int)blockIdx.y == k) { d = r.p[k]; n = r.n16[k]; }
  const uint4 z = make_uint4(0u, 0u, 0u, 0u);
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) mesm_store_wt16(d + i, z);
}

// First node of a captured step: the step's host draws come from a ring of `slots` buffers in PINNED HOST memory
// (read over the host link by this kernel: ~1 KB) instead of a host-to-device copy between two graph replays (a copy
// on the stream between two graph launches cost ~50 us of idle queue per step, tools/host_cost.py), and the dropout
// seed offset of the step is advanced (it used to be an element-wise launch of its own).  The slot is chosen by a
// DEVICE counter that this kernel advances: the host publishes replay n's draws into slot n % slots before it
// launches replay n and never runs more than slots - 1 replays ahead (graphed.GraphedStep).
__global__ __launch_bounds__(256) void step_begin_kernel(const uint4* __restrict__ ring, int32_t slot_vec, int32_t slots,
                                                         uint4* __restrict__ dst, int32_t* __restrict__ pull_ctr,
                                                         int32_t* __restrict__ seed_ctr) {
  const int c = *pull_ctr;
  const uint4* src = ring + (size_t)(c % slots) * slot_vec;
  for (int i = threadIdx.x; i < slot_vec; i += 256) dst[i] = src[i];
  __syncthreads();  // every thread has read the counter
  if (threadIdx.x == 0) {
    *pull_ctr = c + 1;
    if (seed_ctr) *seed_ctr += 1;
  }
}

}  // namespace

extern "C" int mesm_step_begin(const void* host_ring, int32_t slot_bytes, int32_t slots, void* dst, int32_t* pull_ctr,
                               int32_t* seed_ctr, void* stream) {
  if (!host_ring || !dst || !pull_ctr || slots <= 0 || slot_bytes <= 0 || (slot_bytes & 15)) return MESM_EINVAL;
  if ((((uintptr_t)host_ring) | ((uintptr_t)dst)) & 15) return MESM_EALIGN;
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, (const uint4*)host_ring, slot_bytes / 16,
                     slots, (uint4*)dst, pull_ctr, seed_ctr);
  return mesm_launch_status();
}

extern "C" int mesm_stack_rows(const void* const* src, void* const* dst, const int64_t* row_bytes, const int32_t* gather,
                               int32_t n, const int64_t* idx, int32_t N, void* stream) {
  if (!src || !dst || !row_bytes || !gather || n <= 0 || n > STACK_MAX || N <= 0) return MESM_EINVAL;
  StackArgs a;
  for (int t = 0; t < n; ++t) {
    if (!src[t] || !dst[t] || row_bytes[t] <= 0) return MESM_EINVAL;
    if (gather[t] && !idx) return MESM_EINVAL;
    a.src[t] = (const unsigned char*)src[t];
    a.dst[t] = (unsigned char*)dst[t];
    a.row_bytes[t] = row_bytes[t];
    a.gather[t] = gather[t];
  }
  a.idx = idx;
  a.N = N;
  a.n = n;
  hipLaunchKernelGGL(stack_rows_kernel, dim3(2 * N, n), dim3(256), 0, (hipStream_t)stream, a);
  return mesm_launch_status();
}

extern "C" int mesm_unstack_rows(const float* d2, const int64_t* idx, float* dx, int32_t N, int64_t R, void* stream) {
  if (!d2 || !dx || N <= 0 || R <= 0 || (R & 3)) return MESM_EINVAL;
  if ((((uintptr_t)d2) | ((uintptr_t)dx)) & 15) return MESM_EALIGN;
  hipLaunchKernelGGL(unstack_rows_kernel, dim3(N, (unsigned)((R / 4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     d2, idx, dx, N, R);
  return mesm_launch_status();
}

extern "C" int mesm_prepend_fwd(const float* tok, const float* x, const float* ptok, const float* pos,
                                const uint8_t* pad, float* xo, float* po, float* xp, uint8_t* pado, int32_t B,
                                int32_t L, int32_t D, int32_t tok_per_row, int32_t first_pad, void* stream) {
  if (!tok || !x || !xo || B <= 0 || L <= 0 || D <= 0 || (D & 3)) return MESM_EINVAL;
  if ((po || xp) && (!ptok || !pos)) return MESM_EINVAL;
  if (pado && !pad) return MESM_EINVAL;
  hipLaunchKernelGGL(prepend_fwd_kernel, dim3(B, (L + 4) / 4), dim3(256), 0, (hipStream_t)stream, tok, x, ptok, pos, pad, xo,
                     po, xp, pado, L, D, tok_per_row, first_pad);
  return mesm_launch_status();
}

extern "C" int mesm_prepend_bwd(const float* dxo, const float* dxp, const float* dpo, float* dx, float* dtok,
                                float* dptok, int32_t B, int32_t L, int32_t D, int32_t tok_per_row, void* stream) {
  if ((!dxo && !dxp) || B <= 0 || L <= 0 || D <= 0 || (D & 3)) return MESM_EINVAL;
  hipLaunchKernelGGL(prepend_bwd_kernel, dim3(B, (L + 4) / 4), dim3(256), 0, (hipStream_t)stream, dxo, dxp, dpo, dx, dtok,
                     dptok, L, D, tok_per_row);
  return mesm_launch_status();
}

extern "C" int mesm_split_token_fwd(const float* mem, float* g, float* loc, float* dec, int32_t B, int32_t L, int32_t D,
                                    int32_t Bd, void* stream) {
  if (!mem || !g || !loc || B <= 0 || L <= 0 || D <= 0 || (D & 3) || Bd < 0 || Bd > B) return MESM_EINVAL;
  hipLaunchKernelGGL(split_fwd_kernel, dim3(B, (L + 4) / 4), dim3(256), 0, (hipStream_t)stream, mem, g, loc, dec, L, D, Bd);
  return mesm_launch_status();
}

extern "C" int mesm_split_token_bwd(const float* dg, const float* dloc, const float* ddec, float* dmem, int32_t B,
                                    int32_t L, int32_t D, int32_t Bd, void* stream) {
  if (!dmem || B <= 0 || L <= 0 || D <= 0 || (D & 3) || Bd < 0 || Bd > B) return MESM_EINVAL;
  hipLaunchKernelGGL(split_bwd_kernel, dim3(B, (L + 4) / 4), dim3(256), 0, (hipStream_t)stream, dg, dloc, ddec, dmem, L, D, Bd);
  return mesm_launch_status();
}

extern "C" int mesm_token_mix_fwd(const float* x, const uint8_t* m1, const float* tok1, const uint8_t* m2,
                                  const float* tok2, float* y, int64_t rows, int32_t D, void* stream) {
  if (!x || !m1 || !tok1 || !y || rows <= 0 || D <= 0 || (D & 3)) return MESM_EINVAL;
  if ((m2 == nullptr) != (tok2 == nullptr)) return MESM_EINVAL;
  hipLaunchKernelGGL(token_mix_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, m1,
                     tok1, m2, tok2, y, rows, D);
  return mesm_launch_status();
}

extern "C" int mesm_token_mix_bwd(const float* dy, const uint8_t* m1, const uint8_t* m2, float* dx, float* dtok1,
                                  float* dtok2, int64_t rows, int32_t D, void* stream) {
  if (!dy || !m1 || rows <= 0 || D <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(token_mix_bwd_kernel, dim3((unsigned)((rows + TM_ROWS - 1) / TM_ROWS)), dim3(256), 0,
                     (hipStream_t)stream, dy, m1, m2, dx, dtok1, dtok2, rows, D);
  return mesm_launch_status();
}

extern "C" int mesm_gather_rows_fwd(const float* x, const int64_t* idx, const uint8_t* valid, float* y, float* rnorm,
                                    int64_t rows, int32_t D, int32_t normalize, void* stream) {
  if (!x || !idx || !y || rows <= 0 || D <= 0 || (normalize && !rnorm)) return MESM_EINVAL;
  hipLaunchKernelGGL(gather_rows_fwd_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x, idx,
                     valid, y, rnorm, rows, D, normalize);
  return mesm_launch_status();
}

extern "C" int mesm_gather_rows_bwd(const float* dy, const float* y, const float* rnorm, const int64_t* inv,
                                    const uint8_t* valid, float* dx, int64_t src_rows, int32_t D, int32_t normalize,
                                    void* stream) {
  if (!dy || !inv || !dx || src_rows <= 0 || D <= 0 || (normalize && (!y || !rnorm))) return MESM_EINVAL;
  hipLaunchKernelGGL(gather_rows_bwd_kernel, dim3((unsigned)((src_rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, dy,
                     y, rnorm, inv, valid, dx, src_rows, D, normalize);
  return mesm_launch_status();
}

extern "C" int mesm_skinny_linear_bwd(const float* dz, const float* x, const float* w, float* dx, float* dw, float* db,
                                      int64_t M, int32_t K, int32_t J, int32_t relu_mask, void* stream) {
  if (!dz || !x || !w || !dw || M < 0 || K <= 0 || J < 1 || J > 4) return MESM_EINVAL;
  if (M == 0) return MESM_OK;
  const int64_t rb = (M + SK_ROWS - 1) / SK_ROWS;
  if (rb > 0x7fffffff) return MESM_EINVAL;
  hipLaunchKernelGGL(skinny_linear_bwd_kernel, dim3((unsigned)rb, (unsigned)((K + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, dz, x, w, dx, dw, db, M, K, J, relu_mask);
  return mesm_launch_status();
}

extern "C" int mesm_add_n(const float* const* srcs, int32_t k, float* out, int64_t n, void* stream) {
  if (!srcs || !out || k < 1 || k > 8 || n <= 0 || (n & 3)) return MESM_EINVAL;
  AddNSrc s = {};
  uintptr_t al = (uintptr_t)out;
  for (int j = 0; j < k; ++j) {
    if (!srcs[j]) return MESM_EINVAL;
    s.p[j] = srcs[j];
    al |= (uintptr_t)srcs[j];
  }
  if (al & 15) return MESM_EALIGN;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_n_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, s, (int)k, out, n / 4);
  return mesm_launch_status();
}

extern "C" int mesm_add_wrap(const float* a, const float* b, float* out, int64_t n, int64_t nb, void* stream) {
  if (!a || !b || !out || n <= 0 || nb <= 0 || (n & 3) || (nb & 3) || (n % nb)) return MESM_EINVAL;
  if ((((uintptr_t)a) | ((uintptr_t)b) | ((uintptr_t)out)) & 15) return MESM_EALIGN;
  int64_t blocks = (n / 4 + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(add_wrap_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, a, b, out, n / 4, nb / 4);
  return mesm_launch_status();
}

extern "C" int mesm_glue_group(const MesmGlueArgs* list, int32_t n, void* stream) {
  if (!list || n <= 0 || n > 64) return MESM_EINVAL;
  hipStream_t s = (hipStream_t)stream;
  GlueGroup g;
  g.n = 0;
  g.start[0] = 0;
  int rc = MESM_OK;
  auto flush = [&]() {
    if (g.n == 0) return;
    hipLaunchKernelGGL(glue_group_kernel, dim3((unsigned)g.start[g.n]), dim3(256), 0, s, g);
    rc = mesm_launch_status();
    g.n = 0;
  };
  for (int k = 0; k < n && rc == MESM_OK; ++k) {
    const MesmGlueArgs& a = list[k];
    int64_t gx = 0, gy = 1;
    const int D = a.i[0];
    switch (a.op) {
      case MESM_GLUE_TOKEN_MIX_FWD:  // p: x, m1, tok1, m2, tok2, y; n[0] rows; i[0] D
        if (!a.p[0] || !a.p[1] || !a.p[2] || !a.p[5] || a.n[0] <= 0 || D <= 0 || (D & 3) || ((a.p[3] == nullptr) != (a.p[4] == nullptr)))
          return MESM_EINVAL;
        gx = (a.n[0] + 3) / 4;
        break;
      case MESM_GLUE_TOKEN_MIX_BWD:  // p: dy, m1, m2, dx, dtok1, dtok2
        if (!a.p[0] || !a.p[1] || a.n[0] <= 0 || D <= 0) return MESM_EINVAL;
        gx = (a.n[0] + TM_ROWS - 1) / TM_ROWS;
        break;
      case MESM_GLUE_GATHER_ROWS_FWD:  // p: x, idx, valid, y, rnorm; n[0] rows; i[0] D, i[1] normalize
        if (!a.p[0] || !a.p[1] || !a.p[3] || a.n[0] <= 0 || D <= 0 || (a.i[1] && !a.p[4])) return MESM_EINVAL;
        gx = (a.n[0] + 3) / 4;
        break;
      case MESM_GLUE_GATHER_ROWS_BWD:  // p: dy, y, rnorm, inv, valid, dx; n[0] source rows
        if (!a.p[0] || !a.p[3] || !a.p[5] || a.n[0] <= 0 || D <= 0 || (a.i[1] && (!a.p[1] || !a.p[2]))) return MESM_EINVAL;
        gx = (a.n[0] + 3) / 4;
        break;
      case MESM_GLUE_UNSTACK_ROWS:  // p: d2, idx, dx; i[0] N; n[0] R
        if (!a.p[0] || !a.p[2] || a.i[0] <= 0 || a.n[0] <= 0 || (a.n[0] & 3)) return MESM_EINVAL;
        if ((((uintptr_t)a.p[0]) | ((uintptr_t)a.p[2])) & 15) return MESM_EALIGN;
        gx = a.i[0];
        gy = (a.n[0] / 4 + 255) / 256;
        break;
      case MESM_GLUE_STACK_ROWS:  // p: src, dst, idx (NULL: repeat); i[0] N; n[0] row bytes
        if (!a.p[0] || !a.p[1] || a.i[0] <= 0 || a.n[0] <= 0) return MESM_EINVAL;
        gx = 2 * (int64_t)a.i[0];
        break;
      case MESM_GLUE_ADD_TILE: {  // p: a, b, out; n[0] elements of out, n[1] of a, (i[0], i[1]) = int64 elements of b
        const int64_t nb = ((const int64_t*)a.i)[0];
        if (!a.p[0] || !a.p[1] || !a.p[2] || a.n[0] <= 0 || a.n[1] <= 0 || nb <= 0 || ((a.n[0] | a.n[1] | nb) & 3)) return MESM_EINVAL;
        if ((((uintptr_t)a.p[0]) | ((uintptr_t)a.p[1]) | ((uintptr_t)a.p[2])) & 15) return MESM_EALIGN;
        gx = (a.n[0] / 4 + 255) / 256;
        if (gx > 2048) gx = 2048;
        break;
      }
      case MESM_GLUE_GATHER_ADD:  // p: a, b, idx, valid, y; n[0] rows; i[0] D
        if (!a.p[0] || !a.p[1] || !a.p[2] || !a.p[4] || a.n[0] <= 0 || D <= 0) return MESM_EINVAL;
        gx = (a.n[0] + 3) / 4;
        break;
      default:
        return MESM_EINVAL;
    }
    if (gx <= 0 || gx * gy > (1 << 24)) return MESM_EINVAL;
    MesmGlueArgs q = a;
    if (a.op == MESM_GLUE_ADD_TILE) {  // (device form: float4 units)
      q.n[0] = a.n[0] / 4;
      q.n[1] = a.n[1] / 4;
      ((int64_t*)q.i)[0] = ((const int64_t*)a.i)[0] / 4;
    }
    g.p[g.n] = q;
    g.gx[g.n] = (int)gx;
    g.start[g.n + 1] = g.start[g.n] + (int)(gx * gy);
    if (++g.n == GLUE_MAX) flush();
  }
  if (rc == MESM_OK) flush();
  return rc;
}

extern "C" int mesm_fill_ranges(void* const* ptrs, const int64_t* nbytes, int32_t n, void* stream) {
  if (!ptrs || !nbytes || n <= 0 || n > FILL_MAX) return MESM_EINVAL;
  FillRanges r = {};
  int64_t most = 0;
  for (int k = 0; k < n; ++k) {
    if (!ptrs[k] || nbytes[k] <= 0 || (nbytes[k] & 15)) return MESM_EINVAL;
    if (((uintptr_t)ptrs[k]) & 15) return MESM_EALIGN;
    r.p[k] = (uint4*)ptrs[k];
    r.n16[k] = nbytes[k] >> 4;
    if (r.n16[k] > most) most = r.n16[k];
  }
  int64_t gx = (most + 255) / 256;
  if (gx > 1024) gx = 1024;
  hipLaunchKernelGGL(fill_ranges_kernel, dim3((unsigned)gx, (unsigned)n), dim3(256), 0, (hipStream_t)stream, r);
  return mesm_launch_status();
}
