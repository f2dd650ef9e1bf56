// Steady-state rate of the TileGemm core per operand layout (KC/IC) and dtype; standalone executable.
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_gemm.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, bool AKC, bool BKC, int BM, int BN>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T* A, const T* B, T* C, long n, int ktiles) {
  using G = TileGemm<T, AKC, BKC, BM, BN>;
  __shared__ T smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + bi * BM * n : A + bi * BM;
  const T* b = BKC ? B + bj * BN * n : B + bj * BN;
  G::run(a, n, b, n, ktiles, smem, acc);
  T* out = C + bi * BM * n + bj * BN;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * n + c] = v; });
}

template <typename T, bool AKC, bool BKC, int BM, int BN>
int bench(const char* name, long n, int K) {
  T *A, *B, *C;
  CK(hipMalloc(&A, n * n * sizeof(T))); CK(hipMalloc(&B, n * n * sizeof(T))); CK(hipMalloc(&C, n * n * sizeof(T)));
  std::vector<T> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (T)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(A, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  dim3 grid(n / BN, n / BM);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  gemm_kernel<T, AKC, BKC, BM, BN><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) gemm_kernel<T, AKC, BKC, BM, BN><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-28s n=%ld K=%d tile %dx%d: %.3f ms  %.1f TFLOP/s\n", name, n, K, BM, BN, ms, 2.0 * n * n * K / ms / 1e9);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  return 0;
}

// syrk-like: lower-triangular tile grid, C -= A_i A_j^T with K = ktiles*16 (panel at column offset 0)
template <typename T, int MODE>
__global__ __launch_bounds__(256, 2) void syrk_like(T* A, long n, int ktiles, int jbeg) {
  using G = TileGemm<T, true, true>;
  __shared__ T smem[G::SMEM_ELEMS];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  bi += jbeg; bj += jbeg;
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + (long)bi * 128 * n + (long)bj * 128;
  if (MODE == 0) G::foreach (acc, [&](int r, int c, T& v) { v = -C[(long)r * n + c]; });
  else G::zero(acc);
  G::run(A + (long)bi * 128 * n, n, A + (long)bj * 128 * n, n, ktiles, smem, acc);
  if (MODE == 0) G::foreach (acc, [&](int r, int c, T& v) { C[(long)r * n + c] = -v; });
  else G::foreach (acc, [&](int r, int c, T& v) { C[(long)r * n + c] = v; });
}
template <typename T, int MODE>
int bench_syrk(const char* name, long n, int K) {
  T* A; CK(hipMalloc(&A, n * n * sizeof(T)));
  CK(hipMemset(A, 0, n * n * sizeof(T)));
  const int jbeg = K / 128, m = (int)(n / 128) - jbeg;
  const unsigned grid = m * (m + 1) / 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  syrk_like<T, MODE><<<grid, 256>>>(A, n, K / 16, jbeg); CK(hipDeviceSynchronize());
  const int reps = 10;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) syrk_like<T, MODE><<<grid, 256>>>(A, n, K / 16, jbeg);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-28s n=%ld K=%d tiles=%u: %.3f ms  %.1f TFLOP/s\n", name, n, K, grid, ms, 2.0 * grid * 128.0 * 128.0 * K / ms / 1e9);
  CK(hipFree(A));
  return 0;
}

int main() {
  bench_syrk<double, 0>("f64 syrk RMW K=128", 8192, 128);
  bench_syrk<double, 1>("f64 syrk store-only K=128", 8192, 128);
  bench_syrk<double, 0>("f64 syrk RMW K=256", 8192, 256);
  bench_syrk<double, 0>("f64 syrk RMW K=512", 8192, 512);
  bench_syrk<double, 0>("f64 syrk RMW K=1024", 8192, 1024);
  bench_syrk<double, 0>("f64 syrk RMW K=2048", 8192, 2048);
  bench_syrk<double, 1>("f64 syrk store-only K=256", 8192, 256);
  bench_syrk<double, 1>("f64 syrk store-only K=1024", 8192, 1024);
  bench_syrk<double, 0>("f64 syrk RMW K=128 n=4096", 4096, 128);
  // many rounds of tiles (the batched plan's bulk launches have ~21): steady-state cost of the read-modify-write
  bench_syrk<double, 0>("f64 syrk RMW K=256 n=16384", 16384, 256);
  bench_syrk<double, 1>("f64 syrk store-only K=256 n=16384", 16384, 256);
  bench_syrk<double, 0>("f64 syrk RMW K=512 n=16384", 16384, 512);
  bench_syrk<double, 1>("f64 syrk store-only K=512 n=16384", 16384, 512);
  bench_syrk<double, 0>("f64 syrk RMW K=1024 n=16384", 16384, 1024);
  bench_syrk<double, 1>("f64 syrk store-only K=1024 n=16384", 16384, 1024);
  bench_syrk<double, 0>("f64 syrk RMW K=4096 n=16384", 16384, 4096);

  const long n = 8192;
  bench<double, true, true, 128, 128>("f64 KC/KC", n, 8192);
  bench<double, true, false, 128, 128>("f64 KC/IC", n, 8192);
  bench<double, false, false, 128, 128>("f64 IC/IC", n, 8192);
  bench<double, true, true, 128, 128>("f64 KC/KC shortK", n, 128);
  bench<double, false, false, 128, 128>("f64 IC/IC shortK", n, 128);
  bench<double, true, true, 64, 64>("f64 KC/KC 64", n, 8192);
  bench<double, false, false, 64, 64>("f64 IC/IC 64", n, 8192);
  bench<float, true, true, 128, 128>("f32 KC/KC", n, 8192);
  bench<float, false, false, 128, 128>("f32 IC/IC", n, 8192);
  return 0;
}
