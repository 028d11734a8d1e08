// probe: does preloading kernel arguments into SGPRs (-mllvm -amdgpu-kernarg-preload-count=16) shorten a
// graph node on this firmware?  The kernel's first instruction needs its arguments (pointer + sizes).
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ __launch_bounds__(256) void k(const float* a, float* p, int n, int ld, float s, int m) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i < n) p[(size_t)(i / ld) * ld + i % ld] = a[i] * s + (float)m;
}
double chain(hipStream_t s, int n, int wgs, float** a, float** p) {
  hipGraph_t g; hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
  for (int i = 0; i < n; ++i) hipLaunchKernelGGL(k, dim3(wgs), dim3(256), 0, s, a[i % 8], p[i % 8], wgs * 256, 256, 2.0f, i);
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  for (int i = 0; i < 3; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  auto t0 = std::chrono::steady_clock::now();
  for (int i = 0; i < 20; ++i) hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
  hipGraphExecDestroy(ge); hipGraphDestroy(g);
  return us / 20 / n;
}
int main() {
  hipStream_t s; hipStreamCreate(&s);
  float *a[8], *p[8];
  for (int i = 0; i < 8; ++i) { hipMalloc(&a[i], 1200 * 256 * 4); hipMalloc(&p[i], 1200 * 256 * 4); hipMemset(a[i], 0, 1200 * 256 * 4); }
  for (int wgs : {1, 8, 300, 1200}) printf("%4d WG: %6.3f us/node\n", wgs, chain(s, 512, wgs, a, p));
  return 0;
}
