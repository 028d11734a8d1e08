// probe: per-node cost of a serial HIP-graph chain for kernels of increasing content
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k_empty(float* p) {}
__global__ __launch_bounds__(256) void k_store(float* p) { p[blockIdx.x * 256 + threadIdx.x] = 1.0f; }
__global__ __launch_bounds__(256) void k_mfma(float* p, int n) {
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float a = threadIdx.x, b = 1.0f;
  for (int i = 0; i < n; ++i) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc, 0, 0, 0);
  float s = 0; for (int r = 0; r < 16; ++r) s += acc[r];
  p[blockIdx.x * 256 + threadIdx.x] = s;
}
// each lane loads `nv` float4 (rows of 256 floats, like the frag kernel) then stores a sum
__global__ __launch_bounds__(256) void k_load(const float* a, float* p, int nv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* r = a + (size_t)((blockIdx.x % 150) * 32 + (lane & 31)) * 256 + wave * 64 + (lane >> 5) * 4;
  float s = 0;
  for (int i = 0; i < nv; ++i) { float4 x = *(const float4*)(r + 8 * i); s += x.x + x.y + x.z + x.w; }
  p[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k_load_mfma(const float* a, const float* b, float* p, int nv) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float* ra = a + (size_t)(blockIdx.x * 32 + (lane & 31)) * 256 + wave * 64 + (lane >> 5) * 4;
  const float* rb = b + (size_t)(blockIdx.y * 32 + (lane & 31)) * 256 + wave * 64 + (lane >> 5) * 4;
  f32x16 acc; for (int r = 0; r < 16; ++r) acc[r] = 0.f;
  float4 xa[8], xb[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) { xa[i] = *(const float4*)(ra + 8 * i); xb[i] = *(const float4*)(rb + 8 * i); }
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].x, xb[i].x, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].y, xb[i].y, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].z, xb[i].z, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xa[i].w, xb[i].w, acc, 0, 0, 0);
  }
  __shared__ float red[4 * 16 * 64];
  for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = acc[r];
  __syncthreads();
  for (int rr = 0; rr < 4; ++rr) {
    int r = wave * 4 + rr;
    float t = 0; for (int w = 0; w < 4; ++w) t += red[(w * 16 + r) * 64 + lane];
    int row = blockIdx.x * 32 + 4 * (lane >> 5) + (r & 3) + 8 * (r >> 2);
    p[(size_t)row * 256 + blockIdx.y * 32 + (lane & 31)] = t;
  }
}
template <class F> double chain(hipStream_t s, int n, F launch) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) launch(i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  const int reps = 20;
  for (int i = 0; i < reps; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / reps / n;
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  hipStream_t s; hipStreamCreate(&s);
  const int NS = 8; float *a[NS], *b[NS], *p[NS];
  for (int i = 0; i < NS; ++i) { hipMalloc(&a[i], 4800 * 256 * 4); hipMalloc(&b[i], 256 * 256 * 4); hipMalloc(&p[i], 4800 * 256 * 4);
    hipMemset(a[i], 0, 4800 * 256 * 4); hipMemset(b[i], 0, 256 * 256 * 4); }
  const int n = 64;
  printf("empty 1 WG            : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s, p[i % NS]); }));
  printf("empty 600 WG          : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_empty, dim3(600), dim3(256), 0, s, p[i % NS]); }));
  printf("store 8 WG            : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_store, dim3(8), dim3(256), 0, s, p[i % NS]); }));
  printf("store 600 WG          : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_store, dim3(600), dim3(256), 0, s, p[i % NS]); }));
  for (int m : {8, 32, 128})
    printf("mfma x%-3d 600 WG      : %6.2f us/node\n", m, chain(s, n, [&](int i) { hipLaunchKernelGGL(k_mfma, dim3(600), dim3(256), 0, s, p[i % NS], m); }));
  printf("mfma x32 8 WG         : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_mfma, dim3(8), dim3(256), 0, s, p[i % NS], 32); }));
  for (int nv : {1, 8})
    printf("load %d x16B/lane 75 WG: %6.2f us/node\n", nv, chain(s, n, [&](int i) { hipLaunchKernelGGL(k_load, dim3(75), dim3(256), 0, s, a[i % NS], p[i % NS], nv); }));
  printf("load 8 x16B 600 WG(dup): %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_load, dim3(75 * 8), dim3(256), 0, s, a[i % NS] , p[i % NS], 8); }));
  printf("load+mfma 2400x256x256: %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_load_mfma, dim3(75, 8), dim3(256), 0, s, a[i % NS], b[i % NS], p[i % NS], 8); }));
  printf("load+mfma 32x256x256  : %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_load_mfma, dim3(1, 8), dim3(256), 0, s, a[i % NS], b[i % NS], p[i % NS], 8); }));
  printf("load+mfma 4800x256x256: %6.2f us/node\n", chain(s, n, [&](int i) { hipLaunchKernelGGL(k_load_mfma, dim3(150, 8), dim3(256), 0, s, a[i % NS], b[i % NS], p[i % NS], 8); }));
  return 0;
}
