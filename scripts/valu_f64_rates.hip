// Issue cost of the fp64 vector instructions the Gram kernels' exp / sqrt are made of, relative to v_fma_f64:
// 8 independent dependency chains per lane, one wave per SIMD and four (full occupancy of the issue port).
// build: hipcc -O3 --offload-arch=gfx950 scripts/valu_f64_rates.hip -o scripts/valu_f64_rates
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
enum { FMA, MUL, ADD, RNDNE, CVT_I32, CVT_F64_I32, LDEXP, RSQ, RCP, MAXF, LSHL_ADD, FMA32, NOPS };
const char* names[NOPS] = {"v_fma_f64", "v_mul_f64", "v_add_f64", "v_rndne_f64", "v_cvt_i32_f64", "v_cvt_f64_i32", "v_ldexp_f64", "v_rsq_f64",
                           "v_rcp_f64", "v_max_f64", "v_lshl_add_u32", "v_fma_f32"};
template <int OP>
__global__ __launch_bounds__(256) void k(double* out, int iters, double seed) {
  double x[8];
  int n[8];
  float f[8];
  for (int i = 0; i < 8; ++i) { x[i] = seed + threadIdx.x * 1e-3 + i; n[i] = threadIdx.x + i; f[i] = (float)x[i]; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if (OP == FMA) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(x[i]) : "v"(seed));
      if (OP == MUL) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
      if (OP == ADD) asm volatile("v_add_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
      if (OP == RNDNE) asm volatile("v_rndne_f64 %0, %0" : "+v"(x[i]));
      if (OP == CVT_I32) asm volatile("v_cvt_i32_f64 %0, %1" : "=v"(n[i]) : "v"(x[i]));
      if (OP == CVT_F64_I32) asm volatile("v_cvt_f64_i32 %0, %1" : "=v"(x[i]) : "v"(n[i]));
      if (OP == LDEXP) asm volatile("v_ldexp_f64 %0, %0, %1" : "+v"(x[i]) : "v"(n[i]));
      if (OP == RSQ) asm volatile("v_rsq_f64 %0, %0" : "+v"(x[i]));
      if (OP == RCP) asm volatile("v_rcp_f64 %0, %0" : "+v"(x[i]));
      if (OP == MAXF) asm volatile("v_max_f64 %0, %0, %1" : "+v"(x[i]) : "v"(seed));
      if (OP == LSHL_ADD) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(n[i]) : "v"(n[(i + 1) & 7]));
      if (OP == FMA32) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
    }
  }
  double s = 0;
  for (int i = 0; i < 8; ++i) s += x[i] + n[i] + f[i];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int OP>
static float run(double* o, int wgs, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  k<OP><<<wgs, 256>>>(o, 10, 1.0000001); hipDeviceSynchronize();
  hipEventRecord(e0); k<OP><<<wgs, 256>>>(o, iters, 1.0000001); hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1); return ms;
}
int main() {
  double* o; CK(hipMalloc(&o, 4096 * 256 * 8));
  const int iters = 20000;
  for (int wgs : {256, 1024}) {
    float t[NOPS];
    t[FMA] = run<FMA>(o, wgs, iters); t[MUL] = run<MUL>(o, wgs, iters); t[ADD] = run<ADD>(o, wgs, iters); t[RNDNE] = run<RNDNE>(o, wgs, iters);
    t[CVT_I32] = run<CVT_I32>(o, wgs, iters); t[CVT_F64_I32] = run<CVT_F64_I32>(o, wgs, iters); t[LDEXP] = run<LDEXP>(o, wgs, iters);
    t[RSQ] = run<RSQ>(o, wgs, iters); t[RCP] = run<RCP>(o, wgs, iters); t[MAXF] = run<MAXF>(o, wgs, iters); t[LSHL_ADD] = run<LSHL_ADD>(o, wgs, iters);
    t[FMA32] = run<FMA32>(o, wgs, iters);
    printf("%d workgroups x 4 waves (%s):\n", wgs, wgs == 256 ? "one wave per SIMD" : "four waves per SIMD");
    for (int i = 0; i < NOPS; ++i) printf("  %-16s %7.3f ms   %.2f x v_fma_f64\n", names[i], t[i], t[i] / t[FMA]);
  }
  return 0;
}
