// Small HBM-bound kernels: sine position encodings, dropout, activation/bias backward.
#include "common.hpp"

namespace {

constexpr float TWO_PI_F = 6.283185307179586f;  // float32(2*math.pi), as torch casts the scalar

// ---- PositionEmbeddingSine (position_encoding.py:51-72) ----
constexpr int SP_ROWS = 8;  // positions per workgroup
__global__ __launch_bounds__(256) void sine_pos_kernel(const uint8_t* __restrict__ mask,
                                                      float* __restrict__ out, int L, int D) {
  __shared__ float xs[SP_ROWS + 1];  // inclusive prefix counts of this chunk's rows, then the total
  const int b = blockIdx.x;
  const int l0 = blockIdx.y * SP_ROWS;
  const uint8_t* m = mask + (int64_t)b * L;
  __shared__ uint8_t ms[1024];  // the mask row, staged once: the counting loop below is a chain of byte loads otherwise
  if (L <= 1024) {
    for (int t = threadIdx.x; t < L; t += blockDim.x) ms[t] = m[t];
    __syncthreads();
    m = ms;
  }
  // inclusive prefix count of valid clips (exact in fp32, like cumsum(dtype=float32)): 64 positions per ballot of wave 0
  // (a thread per prefix walking the mask bytes was a 75-deep chain in front of everything else)
  __shared__ unsigned long long bal[16];
  if (L <= 1024) {
    if (threadIdx.x < 64)
      for (int c0 = 0; c0 < L; c0 += 64) {
        const int t = c0 + (int)threadIdx.x;
        const unsigned long long bits = __ballot(t < L && m[t] != 0);
        if (threadIdx.x == 0) bal[c0 >> 6] = bits;
      }
    __syncthreads();
    if (threadIdx.x <= SP_ROWS) {
      int upto = threadIdx.x == SP_ROWS ? L - 1 : l0 + (int)threadIdx.x;
      upto = upto < L ? upto : L - 1;
      int c = 0;
      for (int q = 0; q < (upto >> 6); ++q) c += __popcll(bal[q]);
      const int r = upto & 63;
      c += __popcll(bal[upto >> 6] & (r == 63 ? ~0ull : ((1ull << (r + 1)) - 1ull)));
      xs[threadIdx.x] = (float)c;
    }
  } else if (threadIdx.x <= SP_ROWS) {
    const int upto = threadIdx.x == SP_ROWS ? L - 1 : l0 + threadIdx.x;
    int c = 0;
    for (int t = 0; t <= upto && t < L; ++t) c += m[t] != 0;
    xs[threadIdx.x] = (float)c;
  }
  __syncthreads();
  const float last = xs[SP_ROWS];
  for (int i = threadIdx.x; i < D; i += blockDim.x) {
    const float e = (float)(2 * (i / 2)) / (float)D;
    const float dim_t = powf(10000.0f, e);
#pragma unroll
    for (int r = 0; r < SP_ROWS; ++r) {
      const int l = l0 + r;
      if (l < L) {
        const float v = xs[r] / (last + 1e-6f) * TWO_PI_F / dim_t;
        out[((int64_t)b * L + l) * D + i] = (i & 1) ? cosf(v) : sinf(v);
      }
    }
  }
}

// ---- gen_sineembed_for_position (transformer.py:43-59) ----
__global__ __launch_bounds__(256) void query_sine_fwd_kernel(const float* __restrict__ ref,
                                                            float* __restrict__ out, int64_t R,
                                                            int D) {
  const int half = D / 2;
  const int64_t total = R * D;
  for (int64_t idx = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (int64_t)gridDim.x * blockDim.x) {
    int64_t r = idx / D;
    int i = (int)(idx % D);
    int which = i >= half;
    int ii = which ? i - half : i;
    float e = (float)(2 * (ii / 2)) / (float)half;
    float dim_t = powf(10000.0f, e);
    float v = ref[r * 2 + which] * TWO_PI_F / dim_t;
    out[idx] = (ii & 1) ? cosf(v) : sinf(v);
  }
}

// one wave per reference point
__global__ __launch_bounds__(256) void query_sine_bwd_kernel(const float* __restrict__ ref,
                                                            const float* __restrict__ dout,
                                                            float* __restrict__ dref, int64_t R,
                                                            int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int half = D / 2;
  float g0 = 0.0f, g1 = 0.0f;
  for (int i = lane; i < D; i += 64) {
    int which = i >= half;
    int ii = which ? i - half : i;
    float e = (float)(2 * (ii / 2)) / (float)half;
    float dim_t = powf(10000.0f, e);
    float x = ref[r * 2 + which] * TWO_PI_F;
    float v = x / dim_t;
    float dvdr = TWO_PI_F / dim_t;
    float d = (ii & 1) ? -sinf(v) : cosf(v);
    float g = dout[r * D + i] * d * dvdr;
    if (which) g1 += g; else g0 += g;
  }
  g0 = wave_sum(g0);
  g1 = wave_sum(g1);
  if (lane == 0) {
    dref[r * 2] += g0;
    dref[r * 2 + 1] += g1;
  }
}

__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x,
                                                     float* __restrict__ y, int64_t n,
                                                     uint32_t thresh, float inv_keep,
                                                     uint32_t seed,
                                                     const uint32_t* __restrict__ seed_offset) {
  if (seed_offset) seed += *seed_offset;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n;
       i += (int64_t)gridDim.x * blockDim.x)
    y[i] = mesm_dropout_apply(x[i], (uint32_t)i, seed, thresh, inv_keep);
}

// y = dropout(act(x)): the FFN hidden activation, materialised once (transformer.py:537: as an
// operand transform of linear2 and of its dW GEMM it was recomputed by every output tile)
__global__ __launch_bounds__(256) void act_dropout_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                         int64_t n4, int act, const float* __restrict__ slope_p,
                                                         uint32_t thresh, float inv_keep, uint32_t seed,
                                                         const uint32_t* __restrict__ seed_offset) {
  if (seed_offset) seed += *seed_offset;
  const float slope = slope_p ? *slope_p : 0.0f;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (int64_t)gridDim.x * blockDim.x) {
    float4 v = reinterpret_cast<const float4*>(x)[i];
    float o[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      o[e] = mesm_act(o[e], act, slope);
      if (thresh) o[e] = mesm_dropout_apply(o[e], (uint32_t)(4 * i + e), seed, thresh, inv_keep);
    }
    reinterpret_cast<float4*>(y)[i] = make_float4(o[0], o[1], o[2], o[3]);
  }
}


// ---- decoder reference-point arithmetic (transformer.py:36-40, 373-376, 392-397; model.py:250) ----
// out = sigmoid(delta + inverse_sigmoid(ref)), inverse_sigmoid(x) = log(clamp(x01, eps) / clamp(1 - x01, eps)),
// x01 = clamp(x, 0, 1), eps = 1e-3: eight ATen launches per call (and twice that in backward) as one.
__device__ __forceinline__ float inv_sigmoid(float x, float eps, float& dinv) {
  const float xc = fminf(fmaxf(x, 0.0f), 1.0f);
  const float in01 = (x >= 0.0f && x <= 1.0f) ? 1.0f : 0.0f;  // clamp passes the gradient inside [0, 1]
  const float x1 = fmaxf(xc, eps), x2 = fmaxf(1.0f - xc, eps);
  dinv = in01 * ((xc >= eps ? 1.0f / x1 : 0.0f) + ((1.0f - xc) >= eps ? 1.0f / x2 : 0.0f));
  return logf(x1 / x2);
}

__global__ __launch_bounds__(256) void ref_update_fwd_kernel(const float* __restrict__ delta,
                                                            const float* __restrict__ ref, float* __restrict__ out,
                                                            int64_t n, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float d;
  const float t = delta[i] + inv_sigmoid(ref[i], eps, d);
  out[i] = 1.0f / (1.0f + expf(-t));
}

__global__ __launch_bounds__(256) void ref_update_bwd_kernel(const float* __restrict__ out,
                                                            const float* __restrict__ ref,
                                                            const float* __restrict__ dout, float* __restrict__ ddelta,
                                                            float* __restrict__ dref, int64_t n, float eps) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  const float s = out[i];
  const float dpre = dout[i] * s * (1.0f - s);
  float d;
  inv_sigmoid(ref[i], eps, d);
  ddelta[i] = dpre;
  if (dref) dref[i] = dpre * d;
}

// ---- the decoder's first reference points: ref[n, q, :] = sigmoid(p[q, :]) for every pair n
// (transformer.py:197, 361: query_embed.weight repeated over the batch, then .sigmoid()).  As torch ops this was
// sigmoid + expand-copy forward and sum + sigmoid_backward + accumulate backward: five launches for 20 numbers.
__global__ __launch_bounds__(256) void ref_init_fwd_kernel(const float* __restrict__ p, float* __restrict__ out,
                                                          int64_t total, int QC) {
  const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total) return;
  out[i] = 1.0f / (1.0f + expf(-p[i % QC]));
}

// dp[j] += s (1 - s) sum_n dout[n, j]   (one workgroup; dp is the parameter's gradient view)
__global__ __launch_bounds__(256) void ref_init_bwd_kernel(const float* __restrict__ out, const float* __restrict__ dout,
                                                          float* __restrict__ dp, int N, int QC) {
  for (int j = threadIdx.x; j < QC; j += 256) {
    float acc = 0.0f;
    int n = 0;
    for (; n + 8 <= N; n += 8) {
      float t[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) t[u] = dout[(int64_t)(n + u) * QC + j];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += t[u];
    }
    for (; n < N; ++n) acc += dout[(int64_t)n * QC + j];
    const float sg = out[j];
    dp[j] += acc * sg * (1.0f - sg);
  }
}

// out[r, :] = qsine[r, :] * (scale ? scale[r, :] : 1) * sigmoid(anchor[r]) / ref[r, 1]   (one wave per row)
__global__ __launch_bounds__(256) void qsine_scale_fwd_kernel(const float* __restrict__ qsine,
                                                             const float* __restrict__ scale,
                                                             const float* __restrict__ anchor,
                                                             const float* __restrict__ ref, float* __restrict__ out,
                                                             int64_t R, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float f = (1.0f / (1.0f + expf(-anchor[r]))) / ref[r * 2 + 1];
  for (int c = lane; c < D; c += 64) {
    const float q = qsine[r * D + c];
    out[r * D + c] = q * (scale ? scale[r * D + c] : 1.0f) * f;
  }
}

__global__ __launch_bounds__(256) void qsine_scale_bwd_kernel(
    const float* __restrict__ qsine, const float* __restrict__ scale, const float* __restrict__ anchor,
    const float* __restrict__ ref, const float* __restrict__ dout, float* __restrict__ dqsine,
    float* __restrict__ dscale, float* __restrict__ danchor, float* __restrict__ dref, int64_t R, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const float sg = 1.0f / (1.0f + expf(-anchor[r]));
  const float w = ref[r * 2 + 1];
  const float f = sg / w;
  float df = 0.0f;
  for (int c = lane; c < D; c += 64) {
    const float q = qsine[r * D + c], g = dout[r * D + c];
    const float sc = scale ? scale[r * D + c] : 1.0f;
    dqsine[r * D + c] = g * sc * f;
    if (dscale) dscale[r * D + c] = g * q * f;
    df += g * q * sc;
  }
  df = wave_sum(df);
  if (lane == 0) {
    danchor[r] = df * sg * (1.0f - sg) / w;
    dref[r * 2] = 0.0f;
    dref[r * 2 + 1] = -df * sg / (w * w);
  }
}

// ---- the decoder's reference points at a layer boundary, ONE launch (transformer.py:343-397) ----
// What the loop does between two decoder layers is a chain of tiny dependent tensors -- the refined reference point
// (n, nq, 2), its sine embedding (the next layer's ref_point_head input) and that embedding modulated by the
// query_scale / ref_anchor_head outputs (the next layer's ca_qpos_sine_proj input) -- which as kernels of their own
// (ref_update | query_sine | qsine_scale; in front of layer 0: ref_init | query_sine) cost a ~4.7 us launch each.
//   INIT (p):     ref[r, c] = sigmoid(p[(r % Q) * 2 + c])                                    r = pair * Q + query
//   NEXT (delta): ref[r, c] = sigmoid(delta[r, c] + inverse_sigmoid(prev[r, c]))
//   qsine[r, :]   = gen_sineembed_for_position(ref[r, :])                       (transformer.py:43-59)
//   qscaled[r, :] = qsine[r, :] * scale[r, :] * sigmoid(anchor[r]) / ref[r, 1]  (transformer.py:366-376; when anchor)
// Same expressions as the separate kernels above (bit-identical results).  One wave per row.
__global__ __launch_bounds__(256) void ref_step_fwd_kernel(const float* __restrict__ p, int QC,
                                                          const float* __restrict__ delta,
                                                          const float* __restrict__ prev, float eps,
                                                          const float* __restrict__ scale,
                                                          const float* __restrict__ anchor, float* __restrict__ ref_out,
                                                          float* __restrict__ qsine, float* __restrict__ qscaled,
                                                          int64_t R, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  float rc[2];
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    if (p) {
      rc[c] = 1.0f / (1.0f + expf(-p[(r * 2 + c) % QC]));
    } else {
      float d;
      const float t = delta[r * 2 + c] + inv_sigmoid(prev[r * 2 + c], eps, d);
      rc[c] = 1.0f / (1.0f + expf(-t));
    }
  }
  if (lane < 2) ref_out[r * 2 + lane] = lane ? rc[1] : rc[0];
  const float f = anchor ? (1.0f / (1.0f + expf(-anchor[r]))) / rc[1] : 0.0f;
  const int half = D / 2;
  for (int i = lane; i < D; i += 64) {
    const int which = i >= half;
    const int ii = which ? i - half : i;
    const float e = (float)(2 * (ii / 2)) / (float)half;
    const float dim_t = powf(10000.0f, e);
    const float v = (which ? rc[1] : rc[0]) * TWO_PI_F / dim_t;
    const float q = (ii & 1) ? cosf(v) : sinf(v);
    qsine[r * D + i] = q;
    if (anchor) qscaled[r * D + i] = q * (scale ? scale[r * D + i] : 1.0f) * f;
  }
}

// NEXT backward: d delta / d prev from d ref (the refined point is detached before the sine embedding is taken:
// nothing reaches it through qsine / qscaled), d scale / d anchor from d qscaled.  Either gradient may be absent.
__global__ __launch_bounds__(256) void ref_step_bwd_kernel(
    const float* __restrict__ ref, const float* __restrict__ prev, const float* __restrict__ dref_out, float eps,
    const float* __restrict__ qsine, const float* __restrict__ scale, const float* __restrict__ anchor,
    const float* __restrict__ dqscaled, float* __restrict__ ddelta, float* __restrict__ dprev,
    float* __restrict__ dscale, float* __restrict__ danchor, int64_t R, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  if (ddelta && lane < 2) {
    const int64_t i = r * 2 + lane;
    const float s = ref[i];
    const float dpre = dref_out ? dref_out[i] * s * (1.0f - s) : 0.0f;
    float d;
    inv_sigmoid(prev[i], eps, d);
    ddelta[i] = dpre;
    if (dprev) dprev[i] = dpre * d;
  }
  if (danchor) {
    const float sg = 1.0f / (1.0f + expf(-anchor[r]));
    const float w = ref[r * 2 + 1];
    const float f = sg / w;
    float df = 0.0f;
    if (dqscaled) {
      for (int c = lane; c < D; c += 64) {
        const float q = qsine[r * D + c], g = dqscaled[r * D + c];
        const float sc = scale ? scale[r * D + c] : 1.0f;
        if (dscale) dscale[r * D + c] = g * q * f;
        df += g * q * sc;
      }
      df = wave_sum(df);
    } else if (dscale) {
      for (int c = lane; c < D; c += 64) dscale[r * D + c] = 0.0f;
    }
    if (lane == 0) danchor[r] = df * sg * (1.0f - sg) / w;
  }
}

// INIT backward: dp[j] += sum_n s (1 - s) (d ref_a + d ref_b + d ref_c + query_sine_bwd(d qsine + d qsine2))[n, j] -- the
// initial reference points have three consumers besides their sine embedding (the stacked output, the layer-0 width
// modulation, the first refinement) and the embedding has two (ref_point_head, the modulation); their gradients are
// summed HERE instead of by element-wise launches of the autograd engine.  One wave per row (the embedding's
// transcendental functions are the cost: a single workgroup looping over the rows took 142 us), the pairs' terms of a
// (query, coordinate) meet by float atomics in the parameter's gradient view, like every weight gradient of the step.
__global__ __launch_bounds__(256) void ref_init_sine_bwd_kernel(
    const float* __restrict__ ref, const float* __restrict__ da, const float* __restrict__ db,
    const float* __restrict__ dc, const float* __restrict__ dqsine, const float* __restrict__ dqsine2,
    float* __restrict__ dp, int64_t R, int QC, int D) {
  const int lane = threadIdx.x & 63;
  const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= R) return;
  const int half = D / 2;
  float g0 = 0.0f, g1 = 0.0f;
  if (dqsine || dqsine2) {
    for (int i = lane; i < D; i += 64) {
      const int which = i >= half;
      const int ii = which ? i - half : i;
      const float e = (float)(2 * (ii / 2)) / (float)half;
      const float dim_t = powf(10000.0f, e);
      const float x = ref[r * 2 + which] * TWO_PI_F;
      const float v = x / dim_t;
      const float dvdr = TWO_PI_F / dim_t;
      const float d = (ii & 1) ? -sinf(v) : cosf(v);
      const float go = (dqsine ? dqsine[r * D + i] : 0.0f) + (dqsine2 ? dqsine2[r * D + i] : 0.0f);
      const float g = go * d * dvdr;
      if (which) g1 += g; else g0 += g;
    }
    g0 = wave_sum(g0);
    g1 = wave_sum(g1);
  }
  if (lane < 2) {
    float g = lane ? g1 : g0;
    const int64_t i = r * 2 + lane;
    if (da) g += da[i];
    if (db) g += db[i];
    if (dc) g += dc[i];
    const float sg = ref[i];
    atomicAdd(dp + (int)(i % QC), g * sg * (1.0f - sg));
  }
}

constexpr int AB_ROWS = 32;  // rows per workgroup when column sums are accumulated (few atomics);
                             // 4 when there is nothing to reduce (row chains are dependent loads)

__global__ __launch_bounds__(256) void act_bias_bwd_kernel(
    const float* __restrict__ dy, const float* __restrict__ ref, float* __restrict__ dz,
    float* __restrict__ dbias, const float* __restrict__ slope_p, float* __restrict__ dslope,
    int64_t rows, int cols, int act, int rpb) {
  const int c = blockIdx.y * 256 + threadIdx.x;
  const int64_t r0 = (int64_t)blockIdx.x * rpb;
  const int64_t r1 = (r0 + rpb) < rows ? (r0 + rpb) : rows;
  const float slope = slope_p ? *slope_p : 0.0f;
  float colsum = 0.0f, ds = 0.0f;
  if (c < cols) {
#pragma unroll 4
    for (int64_t r = r0; r < r1; ++r) {
      float g = dy[r * cols + c];
      if (act == MESM_ACT_RELU) {
        g = ref[r * cols + c] > 0.0f ? g : 0.0f;
      } else if (act == MESM_ACT_PRELU) {
        float z = ref[r * cols + c];
        if (z <= 0.0f) {
          ds += g * z;
          g *= slope;
        }
      }
      if (dz) dz[r * cols + c] = g;
      colsum += g;
    }
    if (dbias) atomicAdd(dbias + c, colsum);
  }
  if (act == MESM_ACT_PRELU && dslope) {
    ds = wave_sum(ds);
    if ((threadIdx.x & 63) == 0 && ds != 0.0f) atomicAdd(dslope, ds);
  }
}

}  // namespace

extern "C" int mesm_sine_pos_fwd(const uint8_t* mask, float* out, int32_t B, int32_t L,
                                 int32_t D, void* stream) {
  if (!mask || !out || B <= 0 || L <= 0 || D <= 0 || (D & 1)) return MESM_EINVAL;
  if (L > 8192) return MESM_EINVAL;
  hipLaunchKernelGGL(sine_pos_kernel, dim3(B, (L + SP_ROWS - 1) / SP_ROWS), dim3(256), 0,
                     (hipStream_t)stream, mask, out, L, D);
  return mesm_launch_status();
}

extern "C" int mesm_query_sine_fwd(const float* ref, float* out, int64_t R, int32_t D,
                                   void* stream) {
  if (!ref || !out || R <= 0 || D <= 0 || (D % 4)) return MESM_EINVAL;
  int64_t total = R * D;
  int64_t blocks = (total + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL(query_sine_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0,
                     (hipStream_t)stream, ref, out, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_query_sine_bwd(const float* ref, const float* dout, float* dref, int64_t R,
                                   int32_t D, void* stream) {
  if (!ref || !dout || !dref || R <= 0 || D <= 0 || (D % 4)) return MESM_EINVAL;
  hipLaunchKernelGGL(query_sine_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0,
                     (hipStream_t)stream, ref, dout, dref, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_dropout(const float* x, float* y, int64_t n, float p, uint32_t seed,
                            const uint32_t* seed_offset, void* stream) {
  if (!x || !y || n < 0 || p < 0.f || p >= 1.f) return MESM_EINVAL;
  if (n == 0) return MESM_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x,
                     y, n, mesm_drop_threshold(p), 1.0f / (1.0f - p), seed, seed_offset);
  return mesm_launch_status();
}

extern "C" int mesm_act_bias_bwd(const float* dy, const float* ref, float* dz, float* dbias,
                                 const float* slope, float* dslope, int64_t rows, int32_t cols,
                                 int32_t act, void* stream) {
  if (!dy || rows <= 0 || cols <= 0) return MESM_EINVAL;
  if (act != MESM_ACT_NONE && !ref) return MESM_EINVAL;
  if (act == MESM_ACT_PRELU && !slope) return MESM_EINVAL;
  const int rpb = (dbias || dslope) ? AB_ROWS : 4;
  dim3 grid((unsigned)((rows + rpb - 1) / rpb), (cols + 255) / 256);
  hipLaunchKernelGGL(act_bias_bwd_kernel, grid, dim3(256), 0, (hipStream_t)stream, dy, ref, dz,
                     dbias, slope, dslope, rows, cols, act, rpb);
  return mesm_launch_status();
}

// ---- inference windows (eval.py:63-78): [start, end, foreground score] per (pair, query) ----
__global__ __launch_bounds__(256) void windows_kernel(const float* __restrict__ logits,
                                                     const float* __restrict__ spans,
                                                     const float* __restrict__ duration, float* __restrict__ out,
                                                     int N, int Q) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= N * Q) return;
  const float l0 = logits[2 * i], l1 = logits[2 * i + 1];
  const float m = fmaxf(l0, l1);
  const float e0 = expf(l0 - m), e1 = expf(l1 - m);
  const float cx = spans[2 * i], w = spans[2 * i + 1];
  const float d = duration[i / Q];
  // separately rounded operations, like the reference's tensor ops (no fused multiply-add contraction:
  // the rows are rounded to 4 decimals / clip multiples afterwards and compared exactly)
  const float hw = __fmul_rn(0.5f, w);
  out[3 * i] = __fmul_rn(__fsub_rn(cx, hw), d);
  out[3 * i + 1] = __fmul_rn(__fadd_rn(cx, hw), d);
  out[3 * i + 2] = e0 / (e0 + e1);
}

extern "C" int mesm_windows(const float* logits, const float* spans, const float* duration, float* out,
                            int32_t N, int32_t Q, void* stream) {
  if (!logits || !spans || !duration || !out || N <= 0 || Q <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(windows_kernel, dim3((N * Q + 255) / 256), dim3(256), 0, (hipStream_t)stream, logits, spans,
                     duration, out, N, Q);
  return mesm_launch_status();
}

extern "C" int mesm_ref_update_fwd(const float* delta, const float* ref, float* out, int64_t n, float eps,
                                   void* stream) {
  if (!delta || !ref || !out || n <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(ref_update_fwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     delta, ref, out, n, eps);
  return mesm_launch_status();
}

extern "C" int mesm_ref_update_bwd(const float* out, const float* ref, const float* dout, float* ddelta,
                                   float* dref, int64_t n, float eps, void* stream) {
  if (!out || !ref || !dout || !ddelta || n <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(ref_update_bwd_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     out, ref, dout, ddelta, dref, n, eps);
  return mesm_launch_status();
}

extern "C" int mesm_ref_init_fwd(const float* p, float* out, int32_t N, int32_t QC, void* stream) {
  if (!p || !out || N <= 0 || QC <= 0) return MESM_EINVAL;
  const int64_t total = (int64_t)N * QC;
  hipLaunchKernelGGL(ref_init_fwd_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, p, out,
                     total, QC);
  return mesm_launch_status();
}

extern "C" int mesm_ref_init_bwd(const float* out, const float* dout, float* dp, int32_t N, int32_t QC, void* stream) {
  if (!out || !dout || !dp || N <= 0 || QC <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(ref_init_bwd_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, out, dout, dp, N, QC);
  return mesm_launch_status();
}

extern "C" int mesm_ref_step_fwd(const float* p, int32_t QC, const float* delta, const float* prev, float eps,
                                 const float* scale, const float* anchor, float* ref_out, float* qsine, float* qscaled,
                                 int64_t R, int32_t D, void* stream) {
  if (!ref_out || !qsine || R <= 0 || D <= 0 || (D & 1)) return MESM_EINVAL;
  if (p ? (QC <= 0 || (QC & 1) || delta || prev) : (!delta || !prev)) return MESM_EINVAL;
  if (anchor ? !qscaled : (scale != nullptr)) return MESM_EINVAL;
  hipLaunchKernelGGL(ref_step_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, p, QC, delta,
                     prev, eps, scale, anchor, ref_out, qsine, qscaled, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_ref_step_bwd(const float* ref, const float* prev, const float* dref_out, float eps,
                                 const float* qsine, const float* scale, const float* anchor, const float* dqscaled,
                                 float* ddelta, float* dprev, float* dscale, float* danchor, int64_t R, int32_t D,
                                 void* stream) {
  if (!ref || R <= 0 || D <= 0 || (D & 1)) return MESM_EINVAL;
  if (ddelta ? !prev : (dprev != nullptr)) return MESM_EINVAL;
  if (danchor ? (!anchor || !qsine) : (dscale != nullptr)) return MESM_EINVAL;
  if (!ddelta && !danchor) return MESM_OK;
  hipLaunchKernelGGL(ref_step_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, prev,
                     dref_out, eps, qsine, scale, anchor, dqscaled, ddelta, dprev, dscale, danchor, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_ref_init_sine_bwd(const float* ref, const float* da, const float* db, const float* dc,
                                      const float* dqsine, const float* dqsine2, float* dp, int32_t N, int32_t QC,
                                      int32_t D, void* stream) {
  if (!ref || !dp || N <= 0 || QC <= 0 || (QC & 1) || D <= 0 || (D & 1)) return MESM_EINVAL;
  const int64_t R = (int64_t)N * (QC / 2);
  hipLaunchKernelGGL(ref_init_sine_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream, ref, da, db,
                     dc, dqsine, dqsine2, dp, R, QC, D);
  return mesm_launch_status();
}

extern "C" int mesm_qsine_scale_fwd(const float* qsine, const float* scale, const float* anchor,
                                    const float* ref, float* out, int64_t R, int32_t D, void* stream) {
  if (!qsine || !anchor || !ref || !out || R <= 0 || D <= 0) return MESM_EINVAL;
  hipLaunchKernelGGL(qsine_scale_fwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     qsine, scale, anchor, ref, out, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_qsine_scale_bwd(const float* qsine, const float* scale, const float* anchor,
                                    const float* ref, const float* dout, float* dqsine, float* dscale,
                                    float* danchor, float* dref, int64_t R, int32_t D, void* stream) {
  if (!qsine || !anchor || !ref || !dout || !dqsine || !danchor || !dref || R <= 0 || D <= 0) return MESM_EINVAL;
  if ((scale == nullptr) != (dscale == nullptr)) return MESM_EINVAL;
  hipLaunchKernelGGL(qsine_scale_bwd_kernel, dim3((unsigned)((R + 3) / 4)), dim3(256), 0, (hipStream_t)stream,
                     qsine, scale, anchor, ref, dout, dqsine, dscale, danchor, dref, R, D);
  return mesm_launch_status();
}

extern "C" int mesm_act_dropout(const float* x, float* y, int64_t n, int32_t act, const float* slope,
                                float p, uint32_t seed, const uint32_t* seed_offset, void* stream) {
  if (!x || !y || n <= 0 || (n & 3) || p < 0.f || p >= 1.f) return MESM_EINVAL;
  if (act == MESM_ACT_PRELU && !slope) return MESM_EINVAL;
  if (((uintptr_t)x | (uintptr_t)y) & 15) return MESM_EALIGN;
  const int64_t n4 = n / 4;
  int64_t blocks = (n4 + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(act_dropout_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, x, y, n4,
                     act, slope, p > 0.f ? mesm_drop_threshold(p) : 0u, 1.0f / (1.0f - p), seed, seed_offset);
  return mesm_launch_status();
}

extern "C" int mesm_abi_version(void) { return 10; }  // 10: mesm_ref_step_fwd / _bwd, mesm_ref_init_sine_bwd; 9: mesm_criterion_fwd / _bwd, mesm_glue_group, mesm_fill_ranges, mesm_add_n, split-K with dropout / ReLU-gradient epilogues (plane GEMM entries removed); 8: mesm_ref_init_*; 7: mesm_skinny_linear_bwd; 6: MesmAttnArgs.mask_mod, MesmLnArgs + group entries, *_nv, mesm_ddp_*, match_q = -1
extern "C" const char* mesm_arch(void) { return "gfx950"; }
