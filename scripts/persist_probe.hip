// Feasibility probe for a persistent panel-chain kernel:
//  (1) a resident kernel and the command processor of ANOTHER stream hand control back and forth through memory
//      words: the kernel stores flagA = i (system scope) and polls flagB >= i; the host has queued on s2, for every i,
//      hipStreamWaitValue32(flagA >= i) -> a small kernel -> hipStreamWriteValue32(flagB, i).  Round trip per hop?
//  (2) inside one launch: workgroup 0 publishes a 32 KB tile (sc1 / atomic stores) + flag, workgroup 1 polls, acquires,
//      reads it back and publishes the next flag: latency of one producer -> consumer hop, same and different XCD.
// Every spin is bounded: on timeout the kernel records it and leaves.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__device__ __forceinline__ bool wait_ge(const uint32_t* p, uint32_t v, int scope_system, uint32_t* tmo) {
  for (long spins = 0; spins < 20000000; ++spins) {
    const uint32_t x = scope_system ? __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)
                                    : __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (x >= v) return true;
    __builtin_amdgcn_s_sleep(2);
  }
  atomicAdd(tmo, 1u);
  return false;
}

__global__ void pingpong_host(uint32_t* flagA, uint32_t* flagB, int n, uint32_t* tmo, long long* ticks) {
  if (threadIdx.x != 0) return;
  const long long t0 = wall_clock64();
  for (int i = 1; i <= n; ++i) {
    __hip_atomic_store(flagA, (uint32_t)i, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (!wait_ge(flagB, (uint32_t)i, 1, tmo)) break;
  }
  *ticks = wall_clock64() - t0;
}
__global__ void tiny(int* sink) { if (threadIdx.x == 999) *sink = 1; }

// (2) chain of hops between workgroups of one launch; payload = 4096 doubles
__global__ __launch_bounds__(256) void hop_kernel(double* buf, uint32_t* flags, int nhops, uint32_t* tmo, long long* ticks, double* check) {
  const int wg = blockIdx.x, nwg = gridDim.x, t = threadIdx.x;
  __shared__ int ok;
  long long t0 = 0;
  if (wg == 0 && t == 0) t0 = wall_clock64();
  double acc = 0.0;
  for (int h = 0; h < nhops; ++h) {
    const int owner = h % nwg;
    if (wg == owner) {
      if (h > 0) {  // consume hop h-1
        if (t == 0) {
          ok = wait_ge(&flags[h - 1], 1u, 0, tmo);
          __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (!ok) return;
        const double* src = buf + (size_t)((h - 1) & 1) * 4096;
        for (int i = t; i < 4096; i += 256) acc += src[i];
      }
      double* dst = buf + (size_t)(h & 1) * 4096;
      for (int i = t; i < 4096; i += 256) __hip_atomic_store(dst + i, (double)(h + 1) + 1e-6 * i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      __syncthreads();
      if (t == 0) __hip_atomic_store(&flags[h], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
  // wait for the last hop so that the kernel's duration covers the whole chain
  if (wg == 0) {
    if (t == 0) {
      ok = wait_ge(&flags[nhops - 1], 1u, 0, tmo);
      *ticks = wall_clock64() - t0;
    }
    __syncthreads();
  }
  if (acc != 0.0) atomicAdd(check, acc);
}

int main() {
  hipStream_t s, s2; CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  uint32_t *flagA = nullptr, *flagB = nullptr, *tmo = nullptr; long long* ticks; int* sink;
  CK(hipExtMallocWithFlags((void**)&flagA, 8, hipMallocSignalMemory));  // signal memory comes in 8-byte allocations
  CK(hipExtMallocWithFlags((void**)&flagB, 8, hipMallocSignalMemory));
  CK(hipMalloc(&tmo, 64)); CK(hipMalloc(&ticks, 8)); CK(hipMalloc(&sink, 4));
  const int n = 200;
  for (int rep = 0; rep < 2; ++rep) {
    CK(hipMemset(flagA, 0, 8)); CK(hipMemset(flagB, 0, 8)); CK(hipMemset(tmo, 0, 64));
    CK(hipDeviceSynchronize());
    pingpong_host<<<1, 64, 0, s>>>(flagA, flagB, n, tmo, ticks);
    for (int i = 1; i <= n; ++i) {
      CK(hipStreamWaitValue32(s2, flagA, (uint32_t)i, hipStreamWaitValueGte, 0xFFFFFFFFu));
      tiny<<<1, 64, 0, s2>>>(sink);
      CK(hipStreamWriteValue32(s2, flagB, (uint32_t)i, 0));
    }
    CK(hipDeviceSynchronize());
    long long h_ticks; uint32_t h_tmo;
    CK(hipMemcpy(&h_ticks, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
    printf("kernel <-> stream ping-pong: %.2f us per round trip (timeouts %u)\n", h_ticks / 100.0 / n, h_tmo);
  }
  // (2)
  double* buf; uint32_t* flags; double* check;
  const int nhops = 400;
  CK(hipMalloc(&buf, 2 * 4096 * 8)); CK(hipMalloc(&flags, nhops * 4)); CK(hipMalloc(&check, 8));
  for (int nwg : {2, 9, 64}) {
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipMemset(flags, 0, nhops * 4)); CK(hipMemset(tmo, 0, 64)); CK(hipMemset(check, 0, 8));
      hop_kernel<<<nwg, 256, 0, s>>>(buf, flags, nhops, tmo, ticks, check);
      CK(hipDeviceSynchronize());
      long long h_ticks; uint32_t h_tmo; double h_check;
      CK(hipMemcpy(&h_ticks, ticks, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(&h_tmo, tmo, 4, hipMemcpyDeviceToHost));
      CK(hipMemcpy(&h_check, check, 8, hipMemcpyDeviceToHost));
      double expect = 0; for (int h = 0; h + 1 < nhops; ++h) for (int i = 0; i < 4096; ++i) expect += (double)(h + 1) + 1e-6 * i;
      if (rep) printf("in-kernel hop over %d workgroups: %.2f us per hop (32 KB payload, timeouts %u, payload check %s)\n", nwg,
                      h_ticks / 100.0 / nhops, h_tmo, fabs(h_check - expect) < 1e-6 * expect ? "ok" : "STALE");
    }
  }
  return 0;
}
