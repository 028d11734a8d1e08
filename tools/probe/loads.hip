// probe: L2 -> CU load throughput for the access shapes a 32x32-tile GEMM can use (39 MB re-read
// out of a 2.4 MB + 256 KB working set, 600 workgroups x 64 KB), graph chain of 32 launches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <chrono>
// A: [4800][256] floats, B: [256][256].  Each WG (bx, by) touches A rows bx*32.., B rows by*32..
// MODE 0: fragment shape: lane (i,h) reads 16 B at row i, k = w*64 + s*8 + h*4   (32 B per row per instr)
// MODE 1: row shape: a wave instruction reads 1 KB contiguous = one full row (64 lanes x 16 B)
// MODE 2: row shape, 256 B per row x 4 rows per instruction
template <int MODE, bool DISTINCT>
__global__ __launch_bounds__(256) void k_loads(const float* __restrict__ a, const float* __restrict__ b, float* p) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bx = blockIdx.x, by = blockIdx.y;
  const float* A = a + (size_t)(DISTINCT ? (bx * 8 + by) % 150 : bx) * 32 * 256;
  const float* B = DISTINCT ? a + (size_t)((bx * 8 + by + 75) % 150) * 32 * 256 : b + (size_t)by * 32 * 256;
  float4 x[16];
  if (MODE == 0) {
    const float* ra = A + (lane & 31) * 256 + wave * 64 + (lane >> 5) * 4;
    const float* rb = B + (lane & 31) * 256 + wave * 64 + (lane >> 5) * 4;
#pragma unroll
    for (int s = 0; s < 8; ++s) { x[s] = *(const float4*)(ra + 8 * s); x[8 + s] = *(const float4*)(rb + 8 * s); }
  } else if (MODE == 1) {
    // wave w reads rows w*8 .. w*8+7 of A and of B, one full row per instruction
#pragma unroll
    for (int s = 0; s < 8; ++s) {
      x[s] = *(const float4*)(A + (wave * 8 + s) * 256 + lane * 4);
      x[8 + s] = *(const float4*)(B + (wave * 8 + s) * 256 + lane * 4);
    }
  } else {
#pragma unroll
    for (int s = 0; s < 8; ++s) {  // instr s of wave w: rows (s*4 + lane/16), cols w*64 + (lane%16)*4
      x[s] = *(const float4*)(A + (s * 4 + (lane >> 4)) * 256 + wave * 64 + (lane & 15) * 4);
      x[8 + s] = *(const float4*)(B + (s * 4 + (lane >> 4)) * 256 + wave * 64 + (lane & 15) * 4);
    }
  }
  float s = 0;
#pragma unroll
  for (int i = 0; i < 16; ++i) s += x[i].x + x[i].y + x[i].z + x[i].w;
  if (s == 123.456f) p[threadIdx.x] = s;
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
  const int NS = 8; float *a[NS], *b[NS], *p;
  for (int i = 0; i < NS; ++i) { hipMalloc(&a[i], 4800 * 256 * 4); hipMalloc(&b[i], 256 * 256 * 4);
    hipMemset(a[i], 0, 4800 * 256 * 4); hipMemset(b[i], 0, 256 * 256 * 4); }
  hipMalloc(&p, 4096);
  const int n = 32;
#define RUN(MODE, DIST, GX, label) { double t = chain(s, n, [&](int i) { hipLaunchKernelGGL((k_loads<MODE, DIST>), dim3(GX, 8), dim3(256), 0, s, a[i % NS], b[i % NS], p); }); \
    printf("%-44s grid(%3d,8): %6.2f us/node  -> %5.1f TB/s (launch floor 1.7 us excluded: %5.1f TB/s)\n", label, GX, t, GX * 8 * 65536.0 / t / 1e6, GX * 8 * 65536.0 / (t - 1.7) / 1e6); }
  RUN(0, false, 75, "fragment shape (32 B/row/instr), shared");
  RUN(0, true, 75, "fragment shape, distinct tiles per WG");
  RUN(1, false, 75, "row shape 1 KB/instr, shared");
  RUN(1, true, 75, "row shape 1 KB/instr, distinct");
  RUN(2, false, 75, "row shape 4 x 256 B/instr, shared");
  RUN(0, false, 150, "fragment shape, shared");
  RUN(1, false, 150, "row shape 1 KB/instr, shared");
  RUN(2, false, 150, "row shape 4 x 256 B/instr, shared");
  return 0;
}
