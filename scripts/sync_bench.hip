// What do stream-synchronisation primitives cost between two dependent small kernels?
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <chrono>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__global__ void spin(long cycles, int* sink) {
  long t0 = clock64();
  while (clock64() - t0 < cycles) {}
  if (sink && threadIdx.x == 1000) *sink = 1;
}
int main() {
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  hipEvent_t ev[256], evt[256], t0, t1;
  for (int i = 0; i < 256; ++i) { CK(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming)); CK(hipEventCreate(&evt[i])); }
  CK(hipEventCreate(&t0)); CK(hipEventCreate(&t1));
  const int N = 100; const long cyc = 20000;  // ~10 us kernels (clock64 ticks at 100 MHz? measured below)
  uint32_t *sigA = nullptr, *sigB = nullptr;  // signal memory for hipStreamWriteValue32 / hipStreamWaitValue32
  CK(hipExtMallocWithFlags((void**)&sigA, 8, hipMallocSignalMemory)); CK(hipExtMallocWithFlags((void**)&sigB, 8, hipMallocSignalMemory));
  uint32_t epoch = 0;
  for (int mode = 0; mode < 8; ++mode) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipDeviceSynchronize());
      CK(hipEventRecord(t0, s));
      for (int i = 0; i < N; ++i) {
        spin<<<1, 64, 0, s>>>(cyc, nullptr);
        if (mode == 1) CK(hipEventRecord(ev[i], s));                     // record (no-timing event)
        if (mode == 2) CK(hipEventRecord(evt[i], s));                    // record (timing event)
        if (mode == 3) { CK(hipEventRecord(ev[i], s)); CK(hipStreamWaitEvent(s, ev[i], 0)); }  // same-stream wait
        if (mode == 4 && i > 0) CK(hipStreamWaitEvent(s, ev[128 + i - 1], 0));  // wait on long-complete event of s2
        if (mode == 4) { spin<<<1, 64, 0, s2>>>(100, nullptr); CK(hipEventRecord(ev[128 + i], s2)); }
        if (mode == 5) {  // ping-pong: s -> s2 -> s each iteration
          CK(hipEventRecord(ev[i], s)); CK(hipStreamWaitEvent(s2, ev[i], 0));
          spin<<<1, 64, 0, s2>>>(cyc, nullptr);
          CK(hipEventRecord(ev[128 + i], s2)); CK(hipStreamWaitEvent(s, ev[128 + i], 0));
        }
        if (mode == 7) {  // ping-pong through memory values instead of events
          ++epoch;
          CK(hipStreamWriteValue32(s, sigA, epoch, 0)); CK(hipStreamWaitValue32(s2, sigA, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
          spin<<<1, 64, 0, s2>>>(cyc, nullptr);
          CK(hipStreamWriteValue32(s2, sigB, epoch, 0)); CK(hipStreamWaitValue32(s, sigB, epoch, hipStreamWaitValueGte, 0xFFFFFFFFu));
        }
        if (mode == 6) {  // fork only: s2 waits on s each iteration, s never waits
          CK(hipEventRecord(ev[i], s)); CK(hipStreamWaitEvent(s2, ev[i], 0));
          spin<<<1, 64, 0, s2>>>(100, nullptr);
        }
      }
      CK(hipEventRecord(t1, s)); CK(hipEventSynchronize(t1)); CK(hipDeviceSynchronize());
      float ms; CK(hipEventElapsedTime(&ms, t0, t1));
      if (rep == 1) printf("mode %d: %.2f us per iteration\n", mode, ms * 1e3 / N);
    }
  }
  return 0;
}
