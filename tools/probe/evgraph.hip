// probe: can HIP events bracket a kernel inside a captured graph and be read after replay?
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void spin(float* p, int n) { float x = p[threadIdx.x]; for (int i = 0; i < n; ++i) x = x * 1.0001f + 0.5f; p[threadIdx.x] = x; }
#define CK(x) do { hipError_t e = (x); printf("%-55s -> %s\n", #x, hipGetErrorName(e)); } while (0)
int main() {
  float* d; hipMalloc(&d, 4096);
  hipStream_t s; hipStreamCreate(&s);
  for (int variant = 0; variant < 2; ++variant) {
    printf("== variant %d (%s)\n", variant, variant ? "plain hipEventRecord" : "hipEventRecordWithFlags(External)");
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    if (variant == 0) CK(hipEventRecordWithFlags(e0, s, hipEventRecordExternal)); else CK(hipEventRecord(e0, s));
    hipLaunchKernelGGL(spin, dim3(1), dim3(64), 0, s, d, 200000);
    if (variant == 0) CK(hipEventRecordWithFlags(e1, s, hipEventRecordExternal)); else CK(hipEventRecord(e1, s));
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipGraphLaunch(ge, s));
      CK(hipStreamSynchronize(s));
      float ms = -1; hipError_t e = hipEventElapsedTime(&ms, e0, e1);
      printf("   replay %d: elapsed -> %s, %.3f ms\n", rep, hipGetErrorName(e), ms);
    }
  }
  return 0;
}
