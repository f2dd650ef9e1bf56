// Issue-rate peaks of the low-precision matrix instructions an error-free fp64 split (Ozaki scheme) would run on:
// v_mfma_i32_16x16x64_i8, v_mfma_i32_32x32x32_i8, v_mfma_f32_16x16x32_bf16 -- operands in registers, RANDOM bit patterns
// (the clock a chip holds depends on the data), every CU busy, next to the fp64 instruction the product uses.
// build: hipcc -O3 --offload-arch=gfx950 scripts/mfma_peak_lowp.hip -o scripts/mfma_peak_lowp
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef int i4 __attribute__((ext_vector_type(4)));
typedef int i16 __attribute__((ext_vector_type(16)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef double d4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
__device__ __forceinline__ unsigned rnd(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }

template <int NACC>
__global__ __launch_bounds__(256, 2) void k_i8_16(int* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  i4 a[4], b[4], acc[NACC];
  for (int i = 0; i < 4; ++i) { a[i] = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)}; b[i] = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)}; }
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_16x16x64_i8(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  int t = 0;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_i8_32(int* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  i4 a[2], b[2];
  i16 acc[NACC];
  for (int i = 0; i < 2; ++i) { a[i] = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)}; b[i] = {(int)rnd(s), (int)rnd(s), (int)rnd(s), (int)rnd(s)}; }
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) acc[i][j] = 0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_i32_32x32x32_i8(a[i & 1], b[i >> 1], acc[i], 0, 0, 0);
  int t = 0;
  for (int i = 0; i < NACC; ++i)
    for (int j = 0; j < 16; ++j) t += acc[i][j];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_bf16(float* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  bf8 a[4], b[4];
  f4 acc[NACC];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 8; ++j) { a[i][j] = (__bf16)((float)(rnd(s) >> 8) * (1.0f / 8388608.0f) - 1.0f); b[i][j] = (__bf16)((float)(rnd(s) >> 8) * (1.0f / 8388608.0f) - 1.0f); }
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  float t = 0;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}
template <int NACC>
__global__ __launch_bounds__(256, 2) void k_f64(double* out, int iters) {
  unsigned s = threadIdx.x * 2654435761u + blockIdx.x;
  d4 acc[NACC];
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = (double)rnd(s) / 4294967296.0 - 0.5; b[i] = (double)rnd(s) / 4294967296.0 - 0.5; }
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  double t = 0;
  for (int i = 0; i < NACC; ++i) t += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = t;
}

int main() {
  void* o; CK(hipMalloc(&o, 2048 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  const int wgs = 512;
#define RUN(NAME, LAUNCH, OPS_PER_MFMA, NACC, ITERS)                                                                  \
  { LAUNCH(100); CK(hipDeviceSynchronize()); CK(hipEventRecord(e0)); LAUNCH(ITERS); CK(hipEventRecord(e1));            \
    CK(hipEventSynchronize(e1)); CK(hipEventElapsedTime(&ms, e0, e1));                                                \
    printf("%-28s %8.1f T(FL)OP/s  (%d WGs x 4 waves, %d accumulators, %.1f ms)\n", NAME,                              \
           (double)wgs * 4 * ITERS * NACC * (double)(OPS_PER_MFMA) / ms / 1e9, wgs, NACC, ms); }
#define L_I8_16(IT) k_i8_16<16><<<wgs, 256>>>((int*)o, IT)
#define L_I8_32(IT) k_i8_32<4><<<wgs, 256>>>((int*)o, IT)
#define L_BF16(IT) k_bf16<16><<<wgs, 256>>>((float*)o, IT)
#define L_F64(IT) k_f64<16><<<wgs, 256>>>((double*)o, IT)
  RUN("i32_16x16x64_i8", L_I8_16, 2.0 * 16 * 16 * 64, 16, 40000)
  RUN("i32_32x32x32_i8", L_I8_32, 2.0 * 32 * 32 * 32, 4, 40000)
  RUN("f32_16x16x32_bf16", L_BF16, 2.0 * 16 * 16 * 32, 16, 40000)
  RUN("f64_16x16x4_f64", L_F64, 2.0 * 16 * 16 * 4, 16, 20000)
  return 0;
}
