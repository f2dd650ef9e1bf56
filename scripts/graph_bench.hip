// Does a hipGraph shorten a chain of dependent small kernels on gfx950?  100 kernels of ~2 us on one stream:
// direct launches against one captured graph launched repeatedly.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(long cycles, int* sink) {
  long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
int main() {
  hipStream_t s; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  hipEvent_t t0, t1; CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int N = 100;
  for (long cyc : {200L, 4000L, 40000L}) {
    for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s>>>(cyc, nullptr);
    CK(hipStreamSynchronize(s));
    CK(hipEventRecord(t0, s));
    for (int r = 0; r < 10; ++r) for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s>>>(cyc, nullptr);
    CK(hipEventRecord(t1, s)); CK(hipEventSynchronize(t1));
    float ms_direct; CK(hipEventElapsedTime(&ms_direct, t0, t1));
    hipGraph_t g; hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int i = 0; i < N; ++i) spin<<<1, 64, 0, s>>>(cyc, nullptr);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    CK(hipGraphLaunch(ge, s)); CK(hipStreamSynchronize(s));
    CK(hipEventRecord(t0, s));
    for (int r = 0; r < 10; ++r) CK(hipGraphLaunch(ge, s));
    CK(hipEventRecord(t1, s)); CK(hipEventSynchronize(t1));
    float ms_graph; CK(hipEventElapsedTime(&ms_graph, t0, t1));
    printf("kernel body %6ld cycles: direct %.2f us per kernel, graph %.2f us per kernel\n", cyc, ms_direct * 1e3 / (10 * N),
           ms_graph * 1e3 / (10 * N));
    CK(hipGraphExecDestroy(ge)); CK(hipGraphDestroy(g));
  }
  return 0;
}
