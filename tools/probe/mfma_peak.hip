// probe: sustained v_mfma_f32_32x32x2_f32 rate, registers only, N waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
  f32x16 acc[NACC];
  for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = a0 + threadIdx.x, b = b0 + threadIdx.x * 0.5f;
  for (int i = 0; i < iters; ++i) {
#pragma unroll
    for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
  }
  float s = 0; for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks, int iters) {
  float* d; hipMalloc(&d, blocks * 256 * 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k<NACC>, dim3(blocks), dim3(256), 0, 0, d, iters, 1.f, 2.f);
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  double flops = (double)blocks * 4 * iters * NACC * 32 * 32 * 2 * 2;
  printf("NACC=%d blocks=%5d iters=%6d: %8.3f ms  %7.1f TFLOP/s  (%.1f cycles/MFMA/SIMD @2.4GHz, waves/SIMD=%.1f)\n", NACC, blocks, iters, ms,
         flops / ms / 1e9, ms * 1e-3 * 2.4e9 / ((double)iters * NACC * (blocks / 256.0 > 1 ? blocks / 256.0 : 1)), blocks / 256.0);
  hipFree(d);
}
int main() {
  run<1>(256, 20000); run<1>(512, 20000); run<1>(1024, 20000);
  run<4>(256, 5000); run<4>(512, 5000);
  run<1>(256, 200); run<1>(152, 512); run<1>(608, 128);
  return 0;
}
