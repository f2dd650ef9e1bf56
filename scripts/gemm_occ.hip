// Does the tile core gain from MORE co-resident workgroups with smaller tiles?  (128x128 @ 2 per CU is the product's.)
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_gemm.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, bool AKC, bool BKC, int BM, int BN, int OCC, int BK = 16, int PF = 1>
__global__ __launch_bounds__(256, OCC) void gemm_kernel(const T* A, const T* B, T* C, long n, int ktiles) {
  using G = TileGemm<T, AKC, BKC, BM, BN, BK>;
  __shared__ T smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + bi * BM * n : A + bi * BM;
  const T* b = BKC ? B + bj * BN * n : B + bj * BN;
  G::template run<PF>(a, n, b, n, ktiles, smem, acc);
  T* out = C + bi * BM * n + bj * BN;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * n + c] = v; });
}

template <typename T, bool AKC, bool BKC, int BM, int BN, int OCC, int BK = 16, int PF = 1>
int bench(const char* name, long n, int K) {
  T *A, *B, *C;
  CK(hipMalloc(&A, n * n * sizeof(T))); CK(hipMalloc(&B, n * n * sizeof(T))); CK(hipMalloc(&C, n * n * sizeof(T)));
  std::vector<T> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (T)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(A, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  CK(hipMemcpy(B, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  dim3 grid(n / BN, n / BM);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_kernel<T, AKC, BKC, BM, BN, OCC, BK, PF>, 256, 0));
  gemm_kernel<T, AKC, BKC, BM, BN, OCC, BK, PF><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) gemm_kernel<T, AKC, BKC, BM, BN, OCC, BK, PF><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("%-22s tile %3dx%3d BK %d PF %d bounds %d (occupancy query %d/CU): %.3f ms  %.1f TFLOP/s\n", name, BM, BN, BK, PF, OCC, occ, ms, 2.0 * n * n * K / ms / 1e9);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  return 0;
}

int main() {
  const long n = 8192;
  bench<float, true, true, 128, 128, 2, 16, 1>("f32 KC/KC", n, 8192);
  bench<float, true, true, 128, 128, 2, 16, 2>("f32 KC/KC", n, 8192);
  bench<float, true, true, 128, 128, 2, 16, 4>("f32 KC/KC", n, 8192);
  bench<float, true, true, 128, 128, 2, 32, 1>("f32 KC/KC", n, 8192);
  bench<float, true, true, 128, 128, 2, 32, 2>("f32 KC/KC", n, 8192);
  bench<float, false, false, 128, 128, 2, 16, 4>("f32 IC/IC", n, 8192);
  bench<float, false, false, 128, 128, 2, 32, 1>("f32 IC/IC", n, 8192);
  bench<float, false, false, 128, 128, 2, 32, 2>("f32 IC/IC", n, 8192);
  bench<float, true, false, 128, 128, 2, 32, 2>("f32 KC/IC", n, 8192);
  bench<double, true, true, 128, 128, 2, 16, 1>("f64 KC/KC", n, 8192);
  bench<double, true, true, 128, 128, 2, 32, 1>("f64 KC/KC", n, 8192);
  bench<double, false, false, 128, 128, 2, 16, 2>("f64 IC/IC", n, 8192);
  bench<double, false, false, 128, 128, 2, 32, 1>("f64 IC/IC", n, 8192);
  return 0;
}
