// What does the END of a kernel that writes a large output cost, by store flavour?  (round 6: the GEMM epilogues write
// 5-39 MB per launch with plain 4-byte stores; at the kernel boundary the XCD L2s write their dirty lines back.)
// A chain of N dependent launches in a HIP graph; every workgroup (256 threads) spins ~`spin` clocks ("the tile's
// work"), then writes its 64 x 64 fp32 tile (16 KB) of a (tiles x 4096) output:
//   mode 0  plain 4-byte stores, 128 contiguous bytes per half wave (the MFMA C layout of the GEMM epilogues)
//   mode 1  plain 16-byte stores (a wave instruction = 4 rows x 256 B)
//   mode 2  16-byte stores with sc1 (write-through, line dropped from the XCD's L2)
//   mode 3  4-byte stores with sc1
//   mode 4  16-byte stores with nt
//   mode 5  no stores (the chain's floor)
// A consumer pass can follow every writer (mode | 8): each workgroup reads ITS OWN tile back (same blockIdx -> same XCD:
// plain stores leave the line in that L2, sc1 drops it) -- the price write-through may have for the next kernel.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/store_tail.hip -o tools/probe/store_tail ; run: store_tail
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void writer(float* __restrict__ out, int spin, float seed) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long long t0 = clock64();
  float v = seed + tid;
  while (clock64() - t0 < spin) v = v * 1.0001f + 0.5f;
  float* tile = out + (size_t)blockIdx.x * 4096;
  if (MODE == 0 || MODE == 3) {
    // wave w owns rows [16 w, 16 w + 16) of the 64 x 64 tile: per instruction two rows of 32 columns... keep the GEMM's shape:
    // lane -> column (lane & 31) of a 32-wide half, 16 registers -> rows
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      float* p = tile + (size_t)(16 * wave + r) * 64 + (lane & 31) + 32 * (lane >> 5);
      if (MODE == 0) *p = v + r;
      else __hip_atomic_store(p, v + r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  } else if (MODE == 1 || MODE == 2 || MODE == 4) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float* p = tile + (size_t)(16 * wave + 4 * r) * 64 + lane * 4;
      f32x4 x = {v + r, v, v, v};
      if (MODE == 1) *reinterpret_cast<f32x4*>(p) = x;
      else if (MODE == 2) asm volatile("global_store_dwordx4 %0, %1, off sc1" ::"v"(p), "v"(x) : "memory");
      else asm volatile("global_store_dwordx4 %0, %1, off nt" ::"v"(p), "v"(x) : "memory");
    }
  } else {
    if (v == 12345.678f) tile[tid] = v;
  }
}

__global__ __launch_bounds__(256) void reader(const float* __restrict__ in, float* __restrict__ sink) {
  const float* tile = in + (size_t)blockIdx.x * 4096;
  f32x4 a = {0, 0, 0, 0};
#pragma unroll
  for (int r = 0; r < 4; ++r) a += *reinterpret_cast<const f32x4*>(tile + (size_t)r * 1024 + threadIdx.x * 4);
  if (a[0] + a[1] + a[2] + a[3] == 12345.678f) sink[0] = a[0];
}

template <int MODE>
void launch_writer(float* out, int tiles, int spin, hipStream_t s) {
  hipLaunchKernelGGL(writer<MODE>, dim3(tiles), dim3(256), 0, s, out, spin, 1.0f);
}

double chain(int mode, bool read_back, float* out, float* sink, int tiles, int spin, int n) {
  hipStream_t s;
  hipStreamCreate(&s);
  hipGraph_t g;
  hipGraphExec_t ge;
  hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal);
  for (int i = 0; i < n; ++i) {
    switch (mode) {
      case 0: launch_writer<0>(out, tiles, spin, s); break;
      case 1: launch_writer<1>(out, tiles, spin, s); break;
      case 2: launch_writer<2>(out, tiles, spin, s); break;
      case 3: launch_writer<3>(out, tiles, spin, s); break;
      case 4: launch_writer<4>(out, tiles, spin, s); break;
      default: launch_writer<5>(out, tiles, spin, s); break;
    }
    if (read_back) hipLaunchKernelGGL(reader, dim3(tiles), dim3(256), 0, s, out, sink);
  }
  hipStreamEndCapture(s, &g);
  hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
  hipGraphLaunch(ge, s);
  hipStreamSynchronize(s);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  double best = 1e30;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0, s);
    hipGraphLaunch(ge, s);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  hipGraphExecDestroy(ge);
  hipGraphDestroy(g);
  hipStreamDestroy(s);
  return best * 1e3 / n;
}

int main() {
  const int max_tiles = 4800;
  float *out, *sink;
  hipMalloc(&out, (size_t)max_tiles * 4096 * sizeof(float));
  hipMalloc(&sink, 64);
  hipMemset(out, 0, (size_t)max_tiles * 4096 * sizeof(float));
  const char* names[] = {"plain 4 B", "plain 16 B", "sc1 16 B", "sc1 4 B", "nt 16 B", "no stores"};
  const int tiles_list[] = {300, 512, 1200, 2400, 4800};
  const int spins[] = {4000, 12000};
  for (int spin : spins)
    for (int tiles : tiles_list) {
      printf("tiles %4d (%5.1f MB)  spin %5d clk:", tiles, tiles * 16384.0 / 1e6, spin);
      for (int mode = 0; mode < 6; ++mode) printf("  %s %6.2f", names[mode], chain(mode, false, out, sink, tiles, spin, 200));
      printf("  us per launch\n");
      printf("            + reader pass             :");
      for (int mode = 0; mode < 6; ++mode) printf("  %s %6.2f", names[mode], chain(mode, true, out, sink, tiles, spin, 100));
      printf("  us per writer + reader\n");
      fflush(stdout);
    }
  return 0;
}
