// Probe of global_load_lds_dwordx4 semantics on gfx950: which LDS bytes does lane l write, and is the LDS base
// (M0) honoured beyond 64 KB?  Standalone, tiny buffers.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;
__global__ void probe(const double* __restrict__ g, double* out, int base_elems) {
  extern __shared__ double s[];
  const int t = threadIdx.x;
  for (int i = t; i < base_elems + 256; i += 64) s[i] = -1.0;
  __syncthreads();
  // lane t fetches granule (63 - t): a permutation, so the LDS order shows the lane -> LDS map
  __builtin_amdgcn_global_load_lds((gptr)(g + 2 * (63 - t)), (lptr)(s + base_elems), 16, 0, 0);
  __builtin_amdgcn_s_waitcnt(0);
  __syncthreads();
  for (int i = t; i < 256; i += 64) out[i] = s[base_elems + i];
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  double h[128], *g, *o, r[256];
  for (int i = 0; i < 128; ++i) h[i] = i;
  hipMalloc(&g, sizeof(h)); hipMalloc(&o, sizeof(r));
  hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  for (int base : {0, 4096, 8192 + 512, 12288}) {  // LDS byte offsets 0, 32 KB, 68 KB, 96 KB
    const size_t bytes = (size_t)(base + 256) * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(&probe), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    hipMemset(o, 0, sizeof(r));
    probe<<<1, 64, bytes>>>(g, o, base);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("base %6zu B (%s): ", (size_t)base * 8, hipGetErrorString(e));
    for (int i = 0; i < 12; ++i) printf("%g ", r[i]);
    printf("... [126..131]: ");
    for (int i = 126; i < 132; ++i) printf("%g ", r[i]);
    printf("\n");
  }
  return 0;
}
