// Can ONE compute unit be kept free for the latency-critical diagonal-block kernel of the factorisation's panel chain
// while bulk GEMM launches saturate the GPU -- without a persistent server and without any signalling through the command
// processor?  A HOLDER workgroup sleeps on one CU with a register allocation chosen so that no 240-VGPR bulk wave fits
// beside it (512 - 280 = 232 < 240) but the diagonal-block kernel's waves (228 -> 232) still do: the dispatcher then has
// exactly one CU left for every diagonal-block launch.  Measured here: dependent launches of the PRODUCT kernel
// (potrf_diag_fast_kernel<double>) on a stream, (a) idle GPU, (b) beside saturating GEMM launches, (c) as (b) + holder.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Idiscontinuum_amd/csrc scripts/cu_holder_probe.hip -o scripts/cu_holder_probe
#include <stdio.h>
#include <chrono>
#include <thread>
#include <vector>
#include "dgp_diag.h"
#include "dgp_gemm.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

__global__ __launch_bounds__(256, 2) void gemm_kernel(const double* A, const double* B, double* C, long n, int ktiles) {
  using G = TileGemm<double, true, true, 128, 128>;
  __shared__ double smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  G::run(A + bi * 128 * n, n, B + bj * 128 * n, n, ktiles, smem, acc);
  double* out = C + bi * 128 * n + bj * 128;
  G::foreach (acc, [&](int r, int c, double& v) { out[(long)r * n + c] = v; });
}

// 4 waves (one per SIMD), 280 registers each, asleep until *flag != 0 or max_ticks (100 MHz) have passed
__global__ __launch_bounds__(256, 1) void cu_holder_kernel(const int* flag, long long max_ticks, unsigned* where) {
  asm volatile("" ::: "v255", "a23");
  if (threadIdx.x == 0 && where) where[0] = __smid();
  const long long t0 = wall_clock64();
  while (__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0 && wall_clock64() - t0 < max_ticks) __builtin_amdgcn_s_sleep(100);
}
__global__ void set_flag(int* flag, int v) { *flag = v; }
// the diagonal-block kernel's footprint (228 registers, 96.7 KB of LDS), reporting where it was placed
__global__ __launch_bounds__(256) void footprint_kernel(unsigned* out, int i) {
  extern __shared__ unsigned char fp_lds[];
  asm volatile("" ::: "v163", "a63");
  if (threadIdx.x == 0) { out[i] = __smid(); fp_lds[0] = 1; }
}
__global__ void restore_block(const double* src, double* dst, long ld) {
  for (int i = threadIdx.x; i < 128 * 128; i += 256) dst[(long)(i / 128) * ld + i % 128] = src[i];
}

int main() {
  const long ld = 1024, n = 8192;
  std::vector<double> blk(128 * 128), G(128 * 128);
  srand(1);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < 128; ++k) s += G[i * 128 + k] * G[j * 128 + k] / 128;
      blk[i * 128 + j] = s;
    }
  double *A0, *B0, *Tm, *logdet, *GA, *GB, *GC; int *info, *flag; unsigned* where;
  CK(hipMalloc(&A0, ld * ld * 8)); CK(hipMalloc(&B0, 128 * 128 * 8)); CK(hipMalloc(&Tm, ld * ld * 8)); CK(hipMalloc(&logdet, 64));
  CK(hipMalloc(&info, 4096)); CK(hipMalloc(&flag, 4)); CK(hipMalloc(&where, 8));
  CK(hipMalloc(&GA, n * n * 8)); CK(hipMalloc(&GB, n * n * 8)); CK(hipMalloc(&GC, n * n * 8));
  CK(hipMemcpy(B0, blk.data(), 128 * 128 * 8, hipMemcpyHostToDevice));
  std::vector<double> r(n * n);
  for (auto& v : r) v = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(GA, r.data(), n * n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(GB, r.data(), n * n * 8, hipMemcpyHostToDevice));
  hipStream_t s1, s2, s3;
  int least, greatest; CK(hipDeviceGetStreamPriorityRange(&least, &greatest));
  CK(hipStreamCreateWithPriority(&s1, hipStreamNonBlocking, greatest)); CK(hipStreamCreateWithPriority(&s2, hipStreamNonBlocking, least));
  CK(hipStreamCreateWithFlags(&s3, hipStreamNonBlocking));
  const size_t bytes = potrf_diag_fast_smem<double>();
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_fast_kernel<double, double>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  const int reps = 200;
  for (int mode = 0; mode < 3; ++mode) {
    CK(hipMemset(flag, 0, 4)); CK(hipMemset(info, 0, 4096));
    CK(hipDeviceSynchronize());
    if (mode == 2) cu_holder_kernel<<<1, 256, 0, s3>>>(flag, 200000000LL, where);  // first: it needs an empty CU
    if (mode >= 1)
      for (int k = 0; k < 2; ++k) gemm_kernel<<<dim3(n / 128, n / 128), 256, 0, s2>>>(GA, GB, GC, n, (int)(n / 16));
    std::this_thread::sleep_for(std::chrono::milliseconds(3));  // the GEMM has filled the GPU
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipEventRecord(e0, s1));
    for (int i = 0; i < reps; ++i) {
      restore_block<<<1, 256, 0, s1>>>(B0, A0, ld);
      potrf_diag_fast_kernel<double, double><<<1, 256, bytes, s1>>>(A0, ld, 0, Tm, logdet, info, 0, 0, 1, 16, nullptr);
    }
    CK(hipEventRecord(e1, s1));
    CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    set_flag<<<1, 1, 0, s1>>>(flag, 1);
    CK(hipDeviceSynchronize());
    unsigned w = 0; if (mode == 2) CK(hipMemcpy(&w, where, 4, hipMemcpyDeviceToHost));
    int inf; CK(hipMemcpy(&inf, info, 4, hipMemcpyDeviceToHost));
    printf("%-56s %.1f us per (restore + diagonal-block) launch pair, info %d\n",
           mode == 0 ? "GPU otherwise idle:" : mode == 1 ? "beside saturating GEMM launches:" : "beside saturating GEMM launches, CU holder resident:", ms * 1e3 / reps, inf);
    if (mode == 2) printf("  holder sat on hardware CU id 0x%x\n", w);
    if (mode >= 1) {  // where do workgroups with the diagonal-block kernel's footprint land under this load?
      CK(hipMemset(flag, 0, 4));
      if (mode == 2) cu_holder_kernel<<<1, 256, 0, s3>>>(flag, 200000000LL, where);
      for (int k = 0; k < 2; ++k) gemm_kernel<<<dim3(n / 128, n / 128), 256, 0, s2>>>(GA, GB, GC, n, (int)(n / 16));
      std::this_thread::sleep_for(std::chrono::milliseconds(3));
      unsigned* places; CK(hipMalloc(&places, 64 * 4));
      CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&footprint_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
      CK(hipEventRecord(e0, s1));
      for (int i = 0; i < 64; ++i) footprint_kernel<<<1, 256, bytes, s1>>>(places, i);
      CK(hipEventRecord(e1, s1));
      CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms, e0, e1));
      set_flag<<<1, 1, 0, s1>>>(flag, 1);
      CK(hipDeviceSynchronize());
      unsigned pl[64]; CK(hipMemcpy(pl, places, 256, hipMemcpyDeviceToHost));
      printf("  64 empty launches with the diagonal-block footprint: %.1f us each; CU ids:", ms * 1e3 / 64);
      for (int i = 0; i < 64; ++i) printf(" %x", pl[i]);
      printf("\n");
    }
  }
  return 0;
}
