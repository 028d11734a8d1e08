// Frozen text encoders of the MESM forward (reference model/text_encoder.py): the fp16 CLIP text
// transformer (CLIPTextEncoder.forward :340-354, ResidualAttentionBlock :168-189, QuickGELU :163-165,
// fp16-safe LayerNorm :154-160, weights converted by convert_weights :373-394) and the embedding
// lookup of GloveTextEncoder (:432-454), plus the pooling of MESM.CLIP_encode_text /
// GloVe_encode_text (model/model.py:103-143).  Forward only (the reference runs them under no_grad).
//
// Arithmetic = the reference's: activations and Linear / attention weights are IEEE fp16 in memory,
// every Linear accumulates in fp32 (v_mfma_f32_32x32x8_f16) and rounds ONCE to fp16 after the bias,
// QuickGELU and the residual add round after each fp16 op like the eager tensor ops do, LayerNorm runs
// in fp32 on the widened row and rounds its output to fp16.  The attention core reuses the fp32 kernel
// of attention.hip (MESM_MASK_CAUSAL) on the fp16-rounded q, k, v: exact products of fp16 values,
// fp32 softmax, one rounding of the output -- what the fp16 SDPA path does up to its internal
// rounding of the probabilities.
#include <hip/hip_fp16.h>

#include "common.hpp"

namespace {

typedef _Float16 h16;
typedef _Float16 h16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ __forceinline__ float r16(float x) { return (float)(h16)x; }  // round to nearest fp16

// x[n, l, :] = fp16(tok[ids[n, l], :]) + fp16(pos[l, :])   (text_encoder.py:342-344; both tables are fp32
// parameters: convert_weights leaves nn.Embedding / nn.Parameter alone, forward casts with .type(fp16))
__global__ __launch_bounds__(256) void clip_embed_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                         const float* __restrict__ pos, h16* __restrict__ x,
                                                         int64_t rows, int L, int D, int vocab) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  int64_t id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  const int l = (int)(row % L);
  for (int c = threadIdx.x; c < D; c += blockDim.x) {
    const float t = r16(tok[id * D + c]), q = r16(pos[(int64_t)l * D + c]);
    x[row * D + c] = (h16)(t + q);
  }
}

// y = fp16( LN_fp32( float(x) ) ): one wave per row, D <= 64 * 32.
__global__ __launch_bounds__(256) void ln_f16_kernel(const h16* __restrict__ x, const float* __restrict__ gamma,
                                                     const float* __restrict__ beta, h16* __restrict__ y,
                                                     int64_t rows, int D, float eps) {
  const int lane = threadIdx.x & 63;
  const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= rows) return;
  const h16* xr = x + row * D;
  float s = 0.0f;
  for (int c = lane; c < D; c += 64) s += (float)xr[c];
  const float mean = wave_sum(s) / (float)D;
  float v = 0.0f;
  for (int c = lane; c < D; c += 64) {
    const float d = (float)xr[c] - mean;
    v += d * d;
  }
  const float rstd = rsqrtf(wave_sum(v) / (float)D + eps);
  for (int c = lane; c < D; c += 64)
    y[row * D + c] = (h16)(((float)xr[c] - mean) * rstd * gamma[c] + beta[c]);
}

// C[M, N] = epi( A[M, K] @ W[N, K]^T + bias[N] ), fp16 operands, fp32 accumulate.
//   AT  = h16 or float (float: the fp32 attention output, rounded to fp16 while staged)
//   epi: v = fp16(acc + bias); QuickGELU: v = fp16(v * fp16(sigmoid(fp16(1.702 v)))); residual: v = fp16(v + res)
//   out: fp16 (C16) or the same fp16-rounded values widened to fp32 (C32; feeds the attention kernel)
constexpr int GBM = 64, GBN = 64, GBK = 32, GPAD = 4;
constexpr int GS = GBK + GPAD;  // LDS row stride in halfs (72 B: ds_read_b64 of 32 rows is conflict-free)

template <typename AT>
__device__ __forceinline__ h16x8 load8(const AT* p);
template <>
__device__ __forceinline__ h16x8 load8<h16>(const h16* p) {
  return *reinterpret_cast<const h16x8*>(p);
}
template <>
__device__ __forceinline__ h16x8 load8<float>(const float* p) {
  const float4 a = *reinterpret_cast<const float4*>(p), b = *reinterpret_cast<const float4*>(p + 4);
  h16x8 r;
  r[0] = (h16)a.x; r[1] = (h16)a.y; r[2] = (h16)a.z; r[3] = (h16)a.w;
  r[4] = (h16)b.x; r[5] = (h16)b.y; r[6] = (h16)b.z; r[7] = (h16)b.w;
  return r;
}

template <typename AT>
__global__ __launch_bounds__(256) void gemm_f16_kernel(const AT* __restrict__ A, int64_t lda, const h16* __restrict__ W,
                                                       int64_t ldw, const h16* __restrict__ bias,
                                                       const h16* __restrict__ res, int64_t ldr, h16* __restrict__ C16,
                                                       float* __restrict__ C32, int64_t ldc, int M, int N, int K,
                                                       int gelu) {
  __shared__ __attribute__((aligned(16))) h16 As[2][GBM * GS];
  __shared__ __attribute__((aligned(16))) h16 Bs[2][GBN * GS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int m0 = blockIdx.x * GBM, n0 = blockIdx.y * GBN;
  // staging: thread t moves 8 halfs of row t / 4, columns (t % 4) * 8 of the A tile and of the W tile
  const int sr = tid >> 2, sc = (tid & 3) * 8;
  const bool a_ok = m0 + sr < M, b_ok = n0 + sr < N;
  const AT* ap = A + (int64_t)(a_ok ? m0 + sr : 0) * lda + sc;
  const h16* bp = W + (int64_t)(b_ok ? n0 + sr : 0) * ldw + sc;
  h16x8 zero;
#pragma unroll
  for (int i = 0; i < 8; ++i) zero[i] = (h16)0.0f;

  f32x16 acc;
#pragma unroll
  for (int r = 0; r < 16; ++r) acc[r] = 0.0f;

  h16x8 ra = a_ok ? load8<AT>(ap) : zero, rb = b_ok ? *reinterpret_cast<const h16x8*>(bp) : zero;
  *reinterpret_cast<h16x8*>(&As[0][sr * GS + sc]) = ra;
  *reinterpret_cast<h16x8*>(&Bs[0][sr * GS + sc]) = rb;
  __syncthreads();
  const int nk = K / GBK;
  for (int kt = 0; kt < nk; ++kt) {
    const int buf = kt & 1;
    if (kt + 1 < nk) {
      ra = a_ok ? load8<AT>(ap + (int64_t)(kt + 1) * GBK) : zero;
      rb = b_ok ? *reinterpret_cast<const h16x8*>(bp + (int64_t)(kt + 1) * GBK) : zero;
    }
    const h16* as = &As[buf][(wm * 32 + (lane & 31)) * GS + 4 * (lane >> 5)];
    const h16* bs = &Bs[buf][(wn * 32 + (lane & 31)) * GS + 4 * (lane >> 5)];
#pragma unroll
    for (int kk = 0; kk < GBK; kk += 8) {
      const h16x4 a = *reinterpret_cast<const h16x4*>(as + kk);
      const h16x4 b = *reinterpret_cast<const h16x4*>(bs + kk);
      acc = __builtin_amdgcn_mfma_f32_32x32x8f16(a, b, acc, 0, 0, 0);
    }
    if (kt + 1 < nk) {
      *reinterpret_cast<h16x8*>(&As[buf ^ 1][sr * GS + sc]) = ra;
      *reinterpret_cast<h16x8*>(&Bs[buf ^ 1][sr * GS + sc]) = rb;
    }
    __syncthreads();
  }
  // accumulator register r of lane l: row (r / 4) * 8 + (l / 32) * 4 + r % 4, column l % 32
  const int col = n0 + wn * 32 + (lane & 31);
  if (col >= N) return;
  const float bv = bias ? (float)bias[col] : 0.0f;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const int row = m0 + wm * 32 + (r >> 2) * 8 + (lane >> 5) * 4 + (r & 3);
    if (row >= M) continue;
    float v = r16(acc[r] + bv);
    if (gelu) {
      const float t = r16(1.702f * v);
      const float s = r16(1.0f / (1.0f + __expf(-t)));
      v = r16(v * s);
    }
    if (res) v = r16(v + (float)res[(int64_t)row * ldr + col]);
    if (C16) C16[(int64_t)row * ldc + col] = (h16)v;
    else C32[(int64_t)row * ldc + col] = v;
  }
}

// MESM.CLIP_encode_text / GloVe_encode_text tail (model.py:118-134, 138-143): first Lw tokens, pads zeroed,
// sentence = sum / count of the UN-normalised words, then both L2-normalised (eps 1e-5) when `normalize`.
// One workgroup per pair; x is (N, Lx, D) fp16 or fp32 with Lx >= Lw; mask is (N, Lm) with Lm >= Lw.
template <typename XT>
__global__ __launch_bounds__(256) void text_pool_kernel(const XT* __restrict__ x, const uint8_t* __restrict__ mask,
                                                        int Lx, int Lm, int Lw, int D, int normalize,
                                                        float* __restrict__ words, float* __restrict__ sent) {
  extern __shared__ float sm[];  // D sentence accumulators + Lw row norms + 8 scratch
  float* sacc = sm;
  float* rn = sm + D;
  float* scratch = rn + Lw;
  const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const XT* xb = x + (int64_t)n * Lx * D;
  const uint8_t* mb = mask + (int64_t)n * Lm;
  float* wb = words + (int64_t)n * Lw * D;
  for (int l = wave; l < Lw; l += 4) {  // row norms
    float s = 0.0f;
    if (mb[l])
      for (int c = lane; c < D; c += 64) {
        const float v = (float)xb[(int64_t)l * D + c];
        s += v * v;
      }
    s = wave_sum(s);
    if (lane == 0) rn[l] = fmaxf(sqrtf(s), 1e-5f);
  }
  int cnt = 0;
  for (int l = 0; l < Lw; ++l) cnt += mb[l] ? 1 : 0;
  __syncthreads();
  float part = 0.0f;
  for (int c = tid; c < D; c += 256) {
    float s = 0.0f;
    for (int l = 0; l < Lw; ++l) {
      const float v = mb[l] ? (float)xb[(int64_t)l * D + c] : 0.0f;
      s += v;
      wb[(int64_t)l * D + c] = normalize ? v / rn[l] : v;
    }
    s /= (float)cnt;
    sacc[c] = s;
    part += s * s;
  }
  part = wave_sum(part);
  if (lane == 0) scratch[wave] = part;
  __syncthreads();
  const float nrm = fmaxf(sqrtf(scratch[0] + scratch[1] + scratch[2] + scratch[3]), 1e-5f);
  for (int c = tid; c < D; c += 256) sent[(int64_t)n * D + c] = normalize ? sacc[c] / nrm : sacc[c];
}

// rows of an fp32 table by int64 index (GloveTextEncoder.forward, text_encoder.py:446-454)
__global__ __launch_bounds__(256) void embed_rows_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tab,
                                                         float* __restrict__ out, int64_t rows, int D, int vocab) {
  const int64_t row = blockIdx.x;
  if (row >= rows) return;
  int64_t id = ids[row];
  if (id < 0) id = 0;
  if (id >= vocab) id = vocab - 1;
  for (int c = threadIdx.x; c < D; c += blockDim.x) out[row * D + c] = tab[id * D + c];
}

}  // namespace

extern "C" int mesm_clip_embed(const int64_t* ids, const float* tok, const float* pos, void* x, int64_t rows,
                               int32_t L, int32_t D, int32_t vocab, void* stream) {
  if (!ids || !tok || !pos || !x || rows <= 0 || L <= 0 || D <= 0 || vocab <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(clip_embed_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, ids, tok, pos,
                     (h16*)x, rows, L, D, vocab);
  return mesm_launch_status();
}

extern "C" int mesm_layernorm_f16(const void* x, const float* gamma, const float* beta, void* y, int64_t rows,
                                  int32_t D, float eps, void* stream) {
  if (!x || !gamma || !beta || !y || rows <= 0 || D <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(ln_f16_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     (const h16*)x, gamma, beta, (h16*)y, rows, D, eps);
  return mesm_launch_status();
}

extern "C" int mesm_gemm_f16(const void* A, int32_t a_is_f32, int64_t lda, const void* W, int64_t ldw,
                             const void* bias, const void* residual, int64_t ldr, void* C, int32_t c_is_f32,
                             int64_t ldc, int32_t M, int32_t N, int32_t K, int32_t quick_gelu, void* stream) {
  if (!A || !W || !C || M <= 0 || N <= 0 || K <= 0) return MESM_EINVAL;
  if (K % GBK != 0) return MESM_EINVAL;
  if (lda % 8 != 0 || ldw % 8 != 0 || ((uintptr_t)A % 16) != 0 || ((uintptr_t)W % 16) != 0) return MESM_EALIGN;
  dim3 grid((M + GBM - 1) / GBM, (N + GBN - 1) / GBN);
  h16* c16 = c_is_f32 ? nullptr : (h16*)C;
  float* c32 = c_is_f32 ? (float*)C : nullptr;
  if (a_is_f32)
    hipLaunchKernelGGL(gemm_f16_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, (const float*)A, lda,
                       (const h16*)W, ldw, (const h16*)bias, (const h16*)residual, ldr, c16, c32, ldc, M, N, K,
                       quick_gelu);
  else
    hipLaunchKernelGGL(gemm_f16_kernel<h16>, grid, dim3(256), 0, (hipStream_t)stream, (const h16*)A, lda,
                       (const h16*)W, ldw, (const h16*)bias, (const h16*)residual, ldr, c16, c32, ldc, M, N, K,
                       quick_gelu);
  return mesm_launch_status();
}

extern "C" int mesm_text_pool(const void* x, int32_t x_is_f16, const uint8_t* mask, int32_t N, int32_t Lx, int32_t Lm,
                              int32_t Lw, int32_t D, int32_t normalize, float* words, float* sent, void* stream) {
  if (!x || !mask || !words || !sent || N <= 0 || Lw <= 0 || Lw > Lx || Lw > Lm || D <= 0) return MESM_EINVAL;
  const size_t shm = (size_t)(D + Lw + 8) * sizeof(float);
  if (x_is_f16)
    hipLaunchKernelGGL(text_pool_kernel<h16>, dim3(N), dim3(256), shm, (hipStream_t)stream, (const h16*)x, mask, Lx, Lm,
                       Lw, D, normalize, words, sent);
  else
    hipLaunchKernelGGL(text_pool_kernel<float>, dim3(N), dim3(256), shm, (hipStream_t)stream, (const float*)x, mask, Lx,
                       Lm, Lw, D, normalize, words, sent);
  return mesm_launch_status();
}

extern "C" int mesm_embed_rows(const int64_t* ids, const float* table, float* out, int64_t rows, int32_t D,
                               int32_t vocab, void* stream) {
  if (!ids || !table || !out || rows <= 0 || D <= 0 || vocab <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(embed_rows_kernel, dim3((unsigned)rows), dim3(256), 0, (hipStream_t)stream, ids, table, out, rows,
                     D, vocab);
  return mesm_launch_status();
}
