// dgp_selftest.hip -- diagnostic entry point of the C ABI: a grid of 128 x 128 output tiles through EITHER tile-GEMM core.
//
// The O(n^3) stages run on two cores with one contract (dgp_gemm.h::TileGemm, register-staged, and
// dgp_gemm_dma.h::DmaGemm, direct-to-LDS at three workgroups per CU): same accumulator layout, same order of the k-sum,
// hence BITWISE equal results.  Every parity statement about the 128-tile kernels rests on that equality, so it is
// testable through the ABI (tests/test_gpu_gemm_cores.py): all four operand layouts, both precisions, k-ranges that
// are not a multiple of the direct-to-LDS ring (3 chunks of 64 bytes of k), against each other and against a dense
// product.  Nothing in the reference corresponds (its GEMMs are LAPACK / rocBLAS calls made by gpytorch underneath
// src/discontinuum/engines/gpytorch.py:350-353).
#include "../../include/dgp_hip.h"
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"

namespace dgp {

// C[128 bm + i][128 bn + j] = sum_{k < 16 ktiles} opA(128 bm + i, k) opB(128 bn + j, k)
//   KC operand: op(i, k) = p[i ld + k];  IC operand: op(i, k) = p[k ld + i]
template <typename T, bool AKC, bool BKC, bool DMA>
__global__ __launch_bounds__(256, (TileCore<T, AKC, BKC, 128, 128, 1, DMA>::OCC)) void tile_gemm_kernel(
    const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles, T* __restrict__ C, long ldc, int reverse) {
  using K = TileCore<T, AKC, BKC, 128, 128, 1, DMA>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int bm = blockIdx.y, bn = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + (long)bm * 128 * lda : A + (long)bm * 128;
  const T* b = BKC ? B + (long)bn * 128 * ldb : B + (long)bn * 128;
  if (reverse) K::template run<true>(a, lda, b, ldb, ktiles, smem, acc);  // k-tiles of 16 from the last to the first
  else K::run(a, lda, b, ldb, ktiles, smem, acc);
  T* out = C + (long)bm * 128 * ldc + (long)bn * 128;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ldc + c] = v; });
}

template <typename T, bool DMA>
static int launch(int a_kc, int b_kc, const T* A, long lda, const T* B, long ldb, int ktiles, T* C, long ldc, dim3 grid,
                  hipStream_t s, int rev) {
  if (a_kc && b_kc) tile_gemm_kernel<T, true, true, DMA><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else if (a_kc) tile_gemm_kernel<T, true, false, DMA><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else if (b_kc) tile_gemm_kernel<T, false, true, DMA><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else tile_gemm_kernel<T, false, false, DMA><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  return (int)hipGetLastError();
}

}  // namespace dgp

extern "C" int dgp_debug_tile_gemm(int dtype, int core, int a_kc, int b_kc, const void* A, int64_t lda, const void* B,
                                   int64_t ldb, int64_t k, void* C, int64_t ldc, int tiles_m, int tiles_n, int reverse,
                                   void* stream) {
  using namespace dgp;
  if ((dtype != DGP_F64 && dtype != DGP_F32) || (core != 0 && core != 1) || !A || !B || !C || k < 16 || k % 16 != 0 ||
      k > (1 << 24) || tiles_m < 1 || tiles_n < 1 || tiles_m > 4096 || tiles_n > 4096 || lda < 1 || ldb < 1 || ldc < 128L * tiles_n)
    return DGP_E_ARG;
  // the direct-to-LDS loads fetch 16 bytes per lane: operand rows must start 16-byte aligned
  const int64_t epu = dtype == DGP_F64 ? 2 : 4;
  if (lda % epu || ldb % epu || ((uintptr_t)A & 15) || ((uintptr_t)B & 15)) return DGP_E_ARG;
  const dim3 grid((unsigned)tiles_n, (unsigned)tiles_m);
  hipStream_t s = (hipStream_t)stream;
  const int kt = (int)(k / 16);
  if (dtype == DGP_F64)
    return core ? launch<double, true>(a_kc, b_kc, (const double*)A, lda, (const double*)B, ldb, kt, (double*)C, ldc, grid, s, reverse)
                : launch<double, false>(a_kc, b_kc, (const double*)A, lda, (const double*)B, ldb, kt, (double*)C, ldc, grid, s, reverse);
  return core ? launch<float, true>(a_kc, b_kc, (const float*)A, lda, (const float*)B, ldb, kt, (float*)C, ldc, grid, s, reverse)
              : launch<float, false>(a_kc, b_kc, (const float*)A, lda, (const float*)B, ldb, kt, (float*)C, ldc, grid, s, reverse);
}
