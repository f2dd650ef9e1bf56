// Pure-MFMA issue rate (operands in registers): the practical fp64 / fp32 matrix peak of this device.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
template <int NACC>
__global__ __launch_bounds__(256, 2) void k64(double* out, int iters, double a0, double b0) {
  d4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  double a[4], b[4];
  for (int i = 0; i < 4; ++i) { a[i] = a0 + threadIdx.x * 1e-9 + i; b[i] = b0 + i * 0.5; }
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i & 3], b[i >> 2], acc[i], 0, 0, 0);
  double s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
__global__ __launch_bounds__(256) void k32(float* out, int iters, float a0, float b0) {
  f4 acc[NACC];
  for (int i = 0; i < NACC; ++i) acc[i] = {0, 0, 0, 0};
  float a = a0 + threadIdx.x * 1e-6f, b = b0;
  for (int it = 0; it < iters; ++it)
#pragma unroll
    for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[i], 0, 0, 0);
  float s = 0;
  for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
  double* o; CK(hipMalloc(&o, 2048 * 256 * 8));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const int iters = 20000, NACC = 16;
  for (int wgs : {256, 512, 1024}) {
    k64<NACC><<<wgs, 256>>>(o, 100, 0.5, 0.25); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k64<NACC><<<wgs, 256>>>(o, iters, 0.5, 0.25); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fp64 mfma 16x16x4, %4d WGs x 4 waves: %.1f TFLOP/s\n", wgs, (double)wgs * 4 * iters * NACC * 2048.0 / ms / 1e9);
    k32<NACC><<<wgs, 256>>>((float*)o, 100, 0.5f, 0.25f); CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0)); k32<NACC><<<wgs, 256>>>((float*)o, iters, 0.5f, 0.25f); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("fp32 mfma 16x16x4, %4d WGs x 4 waves: %.1f TFLOP/s\n", wgs, (double)wgs * 4 * iters * NACC * 2048.0 / ms / 1e9);
  }
  return 0;
}
