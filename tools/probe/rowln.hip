// Row-owning GEMM + bias + residual + LayerNorm in ONE kernel (the "32 x 256 tile" of DESIGN.md section 8.2, asked for
// since round 2): z = LayerNorm(res + x W^T + b) for outputs of width 256 -- the out-projection -> + x -> norm and
// FFN2 -> + x -> norm pattern of every post-norm layer (/root/reference/model/transformer.py:534-540, 646-649, 754, 793-796).
// A workgroup owns 32 complete rows: 8 waves = 2 halves of the reduce range x 4 column groups of 64; every wave stages
// its own A (32 x 32) and B (64 x 32) slabs by LDS-DMA (no barrier in the k loop, as in the k-split kernels), splits
// them into three exact bf16 terms and issues the six products on v_mfma_f32_32x32x16_bf16; the two halves meet in LDS, the
// four column groups exchange row sums (two passes: mean, then centred squares) and write y = res + x W^T + b (the
// LayerNorm input the backward needs), z, mean, rstd.
// This program times it against the production pair (mesm_gemm_f32 with the residual epilogue, then mesm_layernorm_fwd2)
// on the step's shapes and checks z against that pair.
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mesm_amd/csrc tools/probe/rowln.hip -o tools/probe/rowln \
//        -L mesm_amd -lmesm_gfx950 -Wl,-rpath,'$ORIGIN/../../mesm_amd'
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include <algorithm>

#include "gemm_ws.hpp"

namespace {

constexpr int RL_N = 256;

struct RowLnArgs {
  const float* A; int64_t lda;      // M x K, reduce-contiguous
  const float* W; int64_t ldw;      // 256 x K, reduce-contiguous
  const float* bias;                // 256
  const float* res; int64_t ldr;    // M x 256 or null
  const float* gamma; const float* beta;
  float* y; float* z; float* mean; float* rstd;
  int M, K;
  float eps;
};

// sum over the 32 lanes that share lane >> 5 (every lane of the half ends with the total)
__device__ __forceinline__ float half_sum(float v) {
  v = sum_within<16>(v);
  return add_xor16(v);
}

__global__ __launch_bounds__(512) void rowln_kernel(const RowLnArgs p) {
  extern __shared__ __attribute__((aligned(16))) float L[];  // 8 waves x 3 slabs (12 KB each) = 96 KB
  __shared__ float part[4][32];
  const int tid = threadIdx.x;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int li = lane & 31, h = lane >> 5;
  const int kh = wave >> 2, cw = wave & 3;
  const int m0 = blockIdx.x * 32, n0 = cw * 64;
  const int khalf = ((p.K / 2) + 31) & ~31;
  const int k0 = kh * khalf, k1 = (k0 + khalf < p.K) ? k0 + khalf : p.K;
  const int nst = k1 > k0 ? (k1 - k0 + 31) >> 5 : 0;
  float* mine = L + wave * (3 * WS_SLAB);
  constexpr int R = MESM_LAYOUT_REDUCE_CONTIG;
  auto issue = [&](int st) {
    const int kb = k0 + 32 * st;
    ws_issue<R>(p.A, p.lda, m0, p.M, kb, k1, mine, lane);
    ws_issue<R>(p.W, p.ldw, n0, RL_N, kb, k1, mine + WS_SLAB, lane);
    ws_issue<R>(p.W, p.ldw, n0 + 32, RL_N, kb, k1, mine + 2 * WS_SLAB, lane);
  };
  f32x16 acc[2];
#pragma unroll
  for (int j = 0; j < 2; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.0f;
  if (nst > 0) issue(0);
  for (int st = 0; st < nst; ++st) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float a[4][4], b[2][4][4];
    ws_read<R>(mine, li, h, a);
    ws_read<R>(mine + WS_SLAB, li, h, b[0]);
    ws_read<R>(mine + 2 * WS_SLAB, li, h, b[1]);
    if (st + 1 < nst) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      issue(st + 1);
    }
    SplitFrag<6> sa, sb[2];
    sa.make(a);
    sb[0].make(b[0]);
    sb[1].make(b[1]);
    acc[0] = split_mma<6>(sa, sb[0], acc[0]);
    acc[1] = split_mma<6>(sa, sb[1], acc[1]);
  }
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
  __syncthreads();  // the slabs are free: the k-half hand-over buffer aliases them
  if (kh == 1) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4)
        reinterpret_cast<float4*>(L)[((cw * 2 + j) * 4 + r4) * 64 + lane] =
            make_float4(acc[j][4 * r4], acc[j][4 * r4 + 1], acc[j][4 * r4 + 2], acc[j][4 * r4 + 3]);
  }
  __syncthreads();
  float v[2][16];
  float s[16];
  const bool own = kh == 0;
  if (own) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 32 * j + li;
      const float bv = p.bias ? p.bias[col] : 0.0f;
#pragma unroll
      for (int r4 = 0; r4 < 4; ++r4) {
        const float4 u = reinterpret_cast<const float4*>(L)[((cw * 2 + j) * 4 + r4) * 64 + lane];
        v[j][4 * r4] = acc[j][4 * r4] + u.x + bv;
        v[j][4 * r4 + 1] = acc[j][4 * r4 + 1] + u.y + bv;
        v[j][4 * r4 + 2] = acc[j][4 * r4 + 2] + u.z + bv;
        v[j][4 * r4 + 3] = acc[j][4 * r4 + 3] + u.w + bv;
      }
      if (p.res) {
        float rr[16];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          int row = m0 + 4 * h + (r & 3) + 8 * (r >> 2);
          row = row < p.M ? row : p.M - 1;
          rr[r] = p.res[(int64_t)row * p.ldr + col];
        }
#pragma unroll
        for (int r = 0; r < 16; ++r) v[j][r] += rr[r];
      }
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = m0 + 4 * h + (r & 3) + 8 * (r >> 2);
        if (row < p.M) p.y[(int64_t)row * RL_N + col] = v[j][r];
      }
    }
    // pass 1: row sums over this wave's 64 columns -> part[cw][row]
#pragma unroll
    for (int r = 0; r < 16; ++r) s[r] = half_sum(v[0][r] + v[1][r]);
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[cw][4 * h + (r & 3) + 8 * (r >> 2)] = s[r];
    }
  }
  __syncthreads();
  float mu[16];
  if (own) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int lr = 4 * h + (r & 3) + 8 * (r >> 2);
      mu[r] = (part[0][lr] + part[1][lr] + part[2][lr] + part[3][lr]) * (1.0f / RL_N);
    }
  }
  __syncthreads();
  if (own) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const float d0 = v[0][r] - mu[r], d1 = v[1][r] - mu[r];
      s[r] = half_sum(d0 * d0 + d1 * d1);
    }
    if (li == 0) {
#pragma unroll
      for (int r = 0; r < 16; ++r) part[cw][4 * h + (r & 3) + 8 * (r >> 2)] = s[r];
    }
  }
  __syncthreads();
  if (own) {
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      const int col = n0 + 32 * j + li;
      const float g = p.gamma[col], be = p.beta[col];
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int lr = 4 * h + (r & 3) + 8 * (r >> 2);
        const int row = m0 + lr;
        const float rs = rsqrtf((part[0][lr] + part[1][lr] + part[2][lr] + part[3][lr]) * (1.0f / RL_N) + p.eps);
        if (row < p.M) {
          p.z[(int64_t)row * RL_N + col] = (v[j][r] - mu[r]) * rs * g + be;
          if (cw == 0 && j == 0 && li == 0) {
            p.mean[row] = mu[r];
            p.rstd[row] = rs;
          }
        }
      }
    }
  }
}

}  // namespace

static float* dalloc(size_t n) {
  float* p = nullptr;
  if (hipMalloc(&p, n * sizeof(float)) != hipSuccess) { fprintf(stderr, "hipMalloc failed\n"); exit(1); }
  return p;
}

static void fill(float* d, size_t n, unsigned seed, float scale) {
  std::vector<float> h(n);
  unsigned x = seed * 2654435761u + 12345u;
  for (size_t i = 0; i < n; ++i) {
    x = x * 1664525u + 1013904223u;
    h[i] = ((float)((x >> 8) & 0xFFFF) / 32768.0f - 1.0f) * scale;
  }
  hipMemcpy(d, h.data(), n * sizeof(float), hipMemcpyHostToDevice);
}

template <typename F>
static double bench(F f, int reps) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 20; ++i) f();
  hipDeviceSynchronize();
  std::vector<double> t;
  for (int rnd = 0; rnd < 5; ++rnd) {
    hipEventRecord(e0, 0);
    for (int i = 0; i < reps; ++i) f();
    hipEventRecord(e1, 0);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    t.push_back(ms * 1e3 / reps);
  }
  std::sort(t.begin(), t.end());
  return t[2];
}

int main() {
  hipFuncSetAttribute(reinterpret_cast<const void*>(rowln_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * 3 * WS_SLAB * 4);
  const int shapes[][2] = {{4800, 256}, {4800, 1024}, {2400, 256}, {2400, 1024}, {320, 256}, {320, 1024}, {1056, 256}};
  printf("rows x 256 x K:  one kernel (gemm + bias + residual + LayerNorm) | mesm_gemm_f32 + mesm_layernorm_fwd2 | max |dz|\n");
  for (auto& sh : shapes) {
    const int M = sh[0], K = sh[1];
    float *A = dalloc((size_t)M * K), *W = dalloc((size_t)256 * K), *bias = dalloc(256), *res = dalloc((size_t)M * 256);
    float *gamma = dalloc(256), *beta = dalloc(256);
    float *y1 = dalloc((size_t)M * 256), *z1 = dalloc((size_t)M * 256), *mean1 = dalloc(M), *rstd1 = dalloc(M);
    float *y2 = dalloc((size_t)M * 256), *z2 = dalloc((size_t)M * 256), *mean2 = dalloc(M), *rstd2 = dalloc(M);
    fill(A, (size_t)M * K, 1, 1.0f); fill(W, (size_t)256 * K, 2, 0.1f); fill(bias, 256, 3, 0.5f); fill(res, (size_t)M * 256, 4, 1.0f);
    fill(gamma, 256, 5, 1.0f); fill(beta, 256, 6, 0.5f);
    RowLnArgs p = {A, K, W, K, bias, res, 256, gamma, beta, y1, z1, mean1, rstd1, M, K, 1e-5f};
    auto fused = [&]() { hipLaunchKernelGGL(rowln_kernel, dim3((M + 31) / 32), dim3(512), 8 * 3 * WS_SLAB * 4, 0, p); };
    MesmGemmArgs g = {};
    g.A = A; g.B = W; g.C = y2; g.M = M; g.N = 256; g.K = K;
    g.a_layout = MESM_LAYOUT_REDUCE_CONTIG; g.b_layout = MESM_LAYOUT_REDUCE_CONTIG;
    g.lda = K; g.ldb = K; g.ldc = 256; g.bias = bias; g.residual = res; g.ldr = 256; g.out_scale = 1.0f; g.split_k = 1;
    auto pair = [&]() {
      mesm_gemm_f32(&g, nullptr);
      mesm_layernorm_fwd2(y2, gamma, beta, z2, mean2, rstd2, M, 256, 1e-5f, 0.0f, 0, nullptr, nullptr, nullptr, nullptr);
    };
    auto gemm_only = [&]() { mesm_gemm_f32(&g, nullptr); };
    fused(); pair();
    hipDeviceSynchronize();
    std::vector<float> a((size_t)M * 256), b((size_t)M * 256);
    hipMemcpy(a.data(), z1, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), z2, b.size() * 4, hipMemcpyDeviceToHost);
    double md = 0;
    for (size_t i = 0; i < a.size(); ++i) md = std::max(md, (double)std::fabs(a[i] - b[i]));
    const double tf = bench(fused, 300), tp = bench(pair, 300), tg = bench(gemm_only, 300);
    printf("%5d x 256 x %4d:  %7.2f us | %7.2f us (gemm alone %7.2f) | %.2e   (%d workgroups)\n", M, K, tf, tp, tg, md, (M + 31) / 32);
    hipFree(A); hipFree(W); hipFree(bias); hipFree(res); hipFree(gamma); hipFree(beta);
    hipFree(y1); hipFree(z1); hipFree(mean1); hipFree(rstd1); hipFree(y2); hipFree(z2); hipFree(mean2); hipFree(rstd2);
  }
  return 0;
}
