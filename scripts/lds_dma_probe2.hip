// Does global_load_lds_dwordx4 read its address VGPRs after later instructions may have overwritten them?
#include <hip/hip_runtime.h>
#include <stdio.h>
__global__ void probe(const double* __restrict__ g, double* out, int nops) {
  extern __shared__ double s[];
  const int t = threadIdx.x;
  for (int i = t; i < 256; i += 64) s[i] = -1.0;
  __syncthreads();
  const double* a1 = g + 2 * (63 - t);  // reversed map: what the load should use
  const double* a2 = g + 2 * t;         // identity map: what the clobber writes
  if (nops == 0)
    asm volatile("s_mov_b32 m0, 0\n s_nop 0\n global_load_lds_dwordx4 %0, off\n v_mov_b64 %0, %1\n s_waitcnt vmcnt(0)"
                 : "+v"(a1) : "v"(a2) : "memory");
  else
    asm volatile("s_mov_b32 m0, 0\n s_nop 0\n global_load_lds_dwordx4 %0, off\n s_nop 7\n v_mov_b64 %0, %1\n s_waitcnt vmcnt(0)"
                 : "+v"(a1) : "v"(a2) : "memory");
  __syncthreads();
  for (int i = t; i < 128; i += 64) out[i] = s[i];
}
int main() {
  setvbuf(stdout, nullptr, _IONBF, 0);
  double h[128], *g, *o, r[128];
  for (int i = 0; i < 128; ++i) h[i] = i;
  (void)hipMalloc(&g, sizeof(h)); (void)hipMalloc(&o, sizeof(r));
  (void)hipMemcpy(g, h, sizeof(h), hipMemcpyHostToDevice);
  for (int nops : {0, 1}) {
    probe<<<1, 64, 4096>>>(g, o, nops);
    hipError_t e = hipDeviceSynchronize();
    (void)hipMemcpy(r, o, sizeof(r), hipMemcpyDeviceToHost);
    printf("nops=%d (%s): ", nops, hipGetErrorString(e));
    for (int i = 0; i < 8; ++i) printf("%g ", r[i]);
    printf(" (126 127 124 ... = address read at issue; 0 1 2 3 ... = read after the overwrite)\n");
  }
  return 0;
}
