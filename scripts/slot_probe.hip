// When does a workgroup with the diagonal-block kernel's footprint get a CU slot beside a SATURATING launch of GEMM-like workgroups
// (168 registers, three per CU, a long queue behind them)?  The dispatcher refills every slot a finished workgroup frees with the
// next one of the same launch, so a kernel that needs MORE than one freed slot's worth of LDS / registers waits until the queue is
// empty.  Swept here: the probe's dynamic LDS (bytes) and register count against bulk workgroups of 48 KB and of 32 KB.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 scripts/slot_probe.hip -o scripts/slot_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// ~`ticks` of the 100 MHz clock of busy waiting with 168 registers and `lds` bytes of dynamic LDS
__global__ __launch_bounds__(256, 3) void bulk_like(long long ticks, int* sink) {
  extern __shared__ unsigned char lds[];
  asm volatile("" ::: "v160");
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(32);
  if (threadIdx.x == 0 && sink) lds[0] = 1;
}
template <int REGS>
__global__ __launch_bounds__(256) void probe(long long* started) {
  extern __shared__ unsigned char lds[];
  if (REGS > 200) asm volatile("" ::: "v227");
  else asm volatile("" ::: "v158");
  if (threadIdx.x == 0) { started[blockIdx.x] = wall_clock64(); lds[0] = 1; }
}
__global__ void stamp(long long* t) { *t = wall_clock64(); }

template <int REGS>
int run(size_t bulk_lds, size_t probe_lds) {
  long long *started, *t0;
  CK(hipMalloc(&started, 64 * 8)); CK(hipMalloc(&t0, 8));
  hipStream_t sb, sp;
  int lo, hi; CK(hipDeviceGetStreamPriorityRange(&lo, &hi));
  CK(hipStreamCreateWithPriority(&sb, hipStreamNonBlocking, lo)); CK(hipStreamCreateWithFlags(&sp, hipStreamNonBlocking));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&bulk_like), hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&probe<REGS>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  // 768 slots x 20 rounds of 200 us = 4 ms of saturating bulk work
  bulk_like<<<768 * 20, 256, bulk_lds, sb>>>(20000, nullptr);
  // give the bulk launch 1 ms to fill the GPU, then launch the probe workgroups
  bulk_like<<<1, 256, 0, sp>>>(100000, nullptr);
  stamp<<<1, 1, 0, sp>>>(t0);
  probe<REGS><<<32, 256, probe_lds, sp>>>(started);
  CK(hipDeviceSynchronize());
  long long h[64], ht0;
  CK(hipMemcpy(h, started, 32 * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&ht0, t0, 8, hipMemcpyDeviceToHost));
  double mn = 1e30, mx = 0, sum = 0;
  for (int i = 0; i < 32; ++i) { const double us = (h[i] - ht0) / 100.0; mn = us < mn ? us : mn; mx = us > mx ? us : mx; sum += us; }
  printf("bulk workgroups of %6zu B LDS (168 registers), probe %3d registers + %6zu B LDS: its 32 workgroups start %8.1f / %8.1f / %8.1f us (min / mean / max) after launch\n",
         bulk_lds, REGS > 200 ? 228 : 160, probe_lds, mn, sum / 32, mx);
  CK(hipFree(started)); CK(hipFree(t0)); CK(hipStreamDestroy(sb)); CK(hipStreamDestroy(sp));
  return 0;
}

int main() {
  for (size_t bulk : {(size_t)49152, (size_t)32768})
    for (size_t p : {(size_t)97808, (size_t)97280, (size_t)96768, (size_t)95744, (size_t)81920, (size_t)65536, (size_t)49152, (size_t)16384}) {
      if (run<160>(bulk, p)) return 1;
      if (p == 97808 || p == 65536) if (run<228>(bulk, p)) return 1;
    }
  return 0;
}
