// one long-K GEMM per operand layout (for rocprofv3 --pmc): f64 KC/KC, KC/IC, IC/IC at n = 8192, K = 8192
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_gemm.h"
using namespace dgp;
template <typename T, bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_kernel(const T* A, const T* B, T* C, long n, int ktiles) {
  using G = TileGemm<T, AKC, BKC, 128, 128>;
  __shared__ T smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + bi * 128 * n : A + bi * 128;
  const T* b = BKC ? B + bj * 128 * n : B + bj * 128;
  G::run(a, n, b, n, ktiles, smem, acc);
  T* out = C + bi * 128 * n + bj * 128;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * n + c] = v; });
}
int main() {
  const long n = 8192;
  double *A, *B, *C;
  hipMalloc(&A, n * n * 8); hipMalloc(&B, n * n * 8); hipMalloc(&C, n * n * 8);
  hipMemset(A, 0, n * n * 8); hipMemset(B, 0, n * n * 8);
  dim3 grid(n / 128, n / 128);
  gemm_kernel<double, true, true><<<grid, 256>>>(A, B, C, n, n / 16);
  gemm_kernel<double, true, false><<<grid, 256>>>(A, B, C, n, n / 16);
  gemm_kernel<double, false, false><<<grid, 256>>>(A, B, C, n, n / 16);
  hipDeviceSynchronize();
  printf("done\n");
  return 0;
}
