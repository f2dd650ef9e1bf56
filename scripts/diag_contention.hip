// How much does a saturating GEMM launch slow the diagonal-block factorisation of a RESIDENT workgroup -- sharing its CU
// with one GEMM workgroup (96.7 KB of LDS) or owning the CU (140 KB of LDS: no GEMM workgroup fits beside it)?
// Built against csrc + scripts/persistent_chain.patch (potrf_diag_block as a device function).
#include <stdio.h>
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

__global__ __launch_bounds__(256) void diag_loop(const double* A0, double* Aw, double* Tm, double* logdet, int* info, long ld,
                                                 int reps, long long* ticks) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  long long total = 0, worst = 0;
  for (int r = 0; r < reps; ++r) {
    for (int i = threadIdx.x; i < 128 * 128; i += 256) Aw[(long)(i / 128) * ld + i % 128] = A0[(long)(i / 128) * ld + i % 128];
    __syncthreads();
    const long long t0 = wall_clock64();
    potrf_diag_block<double, false>(Aw, ld, 0, Tm, logdet, info, 1, smem);
    __syncthreads();
    const long long dt = wall_clock64() - t0;
    total += dt;
    worst = dt > worst ? dt : worst;
  }
  if (threadIdx.x == 0) { ticks[0] = total; ticks[1] = worst; }
}

int main() {
  const long ld = 1024, n = 8192;
  std::vector<double> h(ld * ld, 0.0), G(128 * 128);
  srand(1);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < 128; ++i)
    for (int j = 0; j < 128; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < 128; ++k) s += G[i * 128 + k] * G[j * 128 + k] / 128;
      h[i * ld + j] = s;
    }
  double *A0, *Aw, *Tm, *logdet, *GA, *GB, *GC; int* info; long long* ticks;
  CK(hipMalloc(&A0, ld * ld * 8)); CK(hipMalloc(&Aw, ld * ld * 8)); CK(hipMalloc(&Tm, ld * ld * 8)); CK(hipMalloc(&logdet, 8));
  CK(hipMalloc(&info, 1024)); CK(hipMalloc(&ticks, 16));
  CK(hipMalloc(&GA, n * n * 8)); CK(hipMalloc(&GB, n * n * 8)); CK(hipMalloc(&GC, n * n * 8));
  CK(hipMemcpy(A0, h.data(), ld * ld * 8, hipMemcpyHostToDevice));
  std::vector<double> r(n * n);
  for (auto& v : r) v = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(GA, r.data(), n * n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(GB, r.data(), n * n * 8, hipMemcpyHostToDevice));
  hipStream_t s1, s2; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking)); CK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
  const int reps = 400;
  for (size_t lds : {potrf_diag_fast_smem<double>(), (size_t)140 * 1024}) {
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&diag_loop), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    for (int load = 0; load < 2; ++load) {
      CK(hipMemset(info, 0, 1024));
      CK(hipDeviceSynchronize());
      diag_loop<<<1, 256, lds, s1>>>(A0, Aw, Tm, logdet, info, ld, reps, ticks);   // resident first
      if (load)
        for (int k = 0; k < 3; ++k) gemm_kernel<<<dim3(n / 128, n / 128), 256, 0, s2>>>(GA, GB, GC, n, (int)(n / 16));
      CK(hipDeviceSynchronize());
      long long t[2]; CK(hipMemcpy(t, ticks, 16, hipMemcpyDeviceToHost));
      printf("LDS %6zu B, %s: %.1f us per diagonal block (worst %.1f) over %d blocks\n", lds, load ? "GEMM saturating the GPU" : "GPU otherwise idle",
             t[0] / 100.0 / reps, t[1] / 100.0, reps);
    }
  }
  return 0;
}
