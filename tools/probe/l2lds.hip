// How many bytes per clock can one CU pull out of its XCD's L2 (the budget that sizes the plane-GEMM tiles)?
// Every workgroup streams an L2-resident window of `foot` bytes round and round; modes:
//   0  LDS-DMA (global_load_lds_dwordx4): 1 KB per wave instruction into a per-wave LDS ring, DEPTH instructions in flight
//   1  global_load_dwordx4 into registers, DEPTH instructions in flight
// build: hipcc --offload-arch=gfx950 -O3 tools/probe/l2lds.hip -o tools/probe/l2lds ; run: l2lds
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int DEPTH, int MODE>
__global__ __launch_bounds__(1024) void stream_kernel(const char* __restrict__ src, size_t foot, int iters, float* sink) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  // each workgroup owns a window; waves interleave 1 KB pieces inside it
  // `foot` bytes shared by ALL workgroups (L2-resident when <= ~2 MB per XCD), every workgroup starting elsewhere in it
  const char* base = src;
  const unsigned pmask = (unsigned)(foot / 1024) - 1;  // foot is a power of two
  char* mine = lds + wave * (DEPTH * 1024);
  f32x4 acc = {0, 0, 0, 0};
  unsigned piece = wave + blockIdx.x * 37u * nw;
  if (MODE == 0) {
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = base + (size_t)((piece & pmask) * 1024u + lane * 16u);
        piece += nw;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                         (__attribute__((address_space(3))) void*)(mine + d * 1024), 16, 0, 0);
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    acc[0] = *reinterpret_cast<float*>(mine + lane * 16);
  } else {
    for (int it = 0; it < iters; ++it) {
      f32x4 v[DEPTH];
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) {
        const char* g = base + (size_t)((piece & pmask) * 1024u + lane * 16u);
        piece += nw;
        asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[d]) : "v"(g) : "memory");
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
      for (int d = 0; d < DEPTH; ++d) acc += v[d];
    }
  }
  if (acc[0] + acc[1] + acc[2] + acc[3] == 12345.678f) sink[0] = acc[0];
}

template <int DEPTH, int MODE>
double run(const char* src, int waves, int wgs, size_t foot, int iters, float* sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  const size_t shmem = MODE == 0 ? (size_t)waves * DEPTH * 1024 : 0;
  hipFuncSetAttribute(reinterpret_cast<const void*>(stream_kernel<DEPTH, MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(wgs), dim3(waves * 64), shmem, 0, src, foot, 4, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((stream_kernel<DEPTH, MODE>), dim3(wgs), dim3(waves * 64), shmem, 0, src, foot, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  if (hipGetLastError() != hipSuccess) return -1;
  const double bytes = (double)wgs * waves * DEPTH * 1024.0 * iters;
  return bytes / (ms * 1e-3) / 1e9;  // GB/s chip-wide
}

int main() {
  char* src;
  float* sink;
  hipMalloc(&src, (size_t)80 << 20);
  hipMemset(src, 1, (size_t)80 << 20);
  hipMalloc(&sink, 64);
  printf("mode waves/WG WGs depth foot(KB)  chip GB/s  per-CU GB/s  B/clk/CU@2.4GHz\n");
  const size_t foots[] = {1 << 20, 8 << 20};
  for (size_t foot : foots)
    for (int wgs : {256, 512})
      for (int waves : {4, 8, 16}) {
        struct R { const char* name; int depth; double v; };
        std::vector<R> rs;
        const int it = 2000;
        rs.push_back({"ldsdma", 4, run<4, 0>(src, waves, wgs, foot, it, sink)});
        if (waves * 8 * 1024 * (wgs / 256) <= 160 * 1024) rs.push_back({"ldsdma", 8, run<8, 0>(src, waves, wgs, foot, it, sink)});
        if (waves * 16 * 1024 * (wgs / 256) <= 160 * 1024) rs.push_back({"ldsdma", 16, run<16, 0>(src, waves, wgs, foot, it, sink)});
        rs.push_back({"vgpr", 4, run<4, 1>(src, waves, wgs, foot, it, sink)});
        rs.push_back({"vgpr", 8, run<8, 1>(src, waves, wgs, foot, it, sink)});
        for (auto& r : rs)
          printf("%-6s %2d %4d %3d %5zu  %9.0f  %8.1f  %6.1f\n", r.name, waves, wgs, r.depth, foot >> 10, r.v, r.v / 256, r.v / 256 / 2.4);
      }
  return 0;
}
