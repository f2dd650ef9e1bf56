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
// variant: 0 plain map; 1 interleaved 16-row / 16-column groups (IL); 2..5 IL + zero-work skipping TRI_ROW_LE / TRI_ROW_GE /
// TRI_COL_LE / TRI_LOWER (dgp_gemm_dma.h) -- the caller's operands must then HAVE the zero structure the mode assumes in the
// block of 128 k's that is visited last (TRI_LOWER: only the sub-tiles with row group >= column group are defined)
template <typename T, bool AKC, bool BKC, bool DMA, int VARIANT>
__global__ __launch_bounds__(256, (TileCore<T, AKC, BKC, 128, 128, 1, DMA>::OCC)) void tile_gemm_kernel(
    const T* __restrict__ A, long lda, const T* __restrict__ B, long ldb, int ktiles, T* __restrict__ C, long ldc, int reverse) {
  using K = TileCore<T, AKC, BKC, 128, 128, 1, DMA, (VARIANT > 0)>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int bm = blockIdx.y, bn = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + (long)bm * 128 * lda : A + (long)bm * 128;
  const T* b = BKC ? B + (long)bn * 128 * ldb : B + (long)bn * 128;
  constexpr int TRI = VARIANT >= 2 ? VARIANT - 1 : 0;  // TriMode
  T* out = C + (long)bm * 128 * ldc + (long)bn * 128;
  auto store = [&]() { K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ldc + c] = v; }); };
  if (reverse) K::template run_tri<true, TRI>(a, lda, b, ldb, ktiles, smem, acc, store);  // k-tiles of 16 from the last to the first
  else K::template run_tri<false, TRI>(a, lda, b, ldb, ktiles, smem, acc, store);
}

template <typename T, bool DMA, int VARIANT>
static int launch(int a_kc, int b_kc, const T* A, long lda, const T* B, long ldb, int ktiles, T* C, long ldc, dim3 grid,
                  hipStream_t s, int rev) {
  if (a_kc && b_kc) tile_gemm_kernel<T, true, true, DMA, VARIANT><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else if (a_kc) tile_gemm_kernel<T, true, false, DMA, VARIANT><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else if (b_kc) tile_gemm_kernel<T, false, true, DMA, VARIANT><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  else tile_gemm_kernel<T, false, false, DMA, VARIANT><<<grid, 256, 0, s>>>(A, lda, B, ldb, ktiles, C, ldc, rev);
  return (int)hipGetLastError();
}
template <typename T>
static int launch_variant(int core, int variant, int a_kc, int b_kc, const T* A, long lda, const T* B, long ldb, int ktiles, T* C,
                          long ldc, dim3 grid, hipStream_t s, int rev) {
  if (!core) return launch<T, false, 0>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
  switch (variant) {
    case 0: return launch<T, true, 0>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
    case 1: return launch<T, true, 1>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
    case 2: return launch<T, true, 2>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
    case 3: return launch<T, true, 3>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
    case 4: return launch<T, true, 4>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
    default: return launch<T, true, 5>(a_kc, b_kc, A, lda, B, ldb, ktiles, C, ldc, grid, s, rev);
  }
}

}  // namespace dgp

extern "C" int dgp_debug_tile_gemm(int dtype, int core, int a_kc, int b_kc, const void* A, int64_t lda, const void* B,
                                   int64_t ldb, int64_t k, void* C, int64_t ldc, int tiles_m, int tiles_n, int reverse,
                                   int variant, void* stream) {
  using namespace dgp;
  if (variant < 0 || variant > 5 || (variant > 0 && core != 1) || (variant >= 2 && k % 128 != 0)) return DGP_E_ARG;
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
    return launch_variant<double>(core, variant, a_kc, b_kc, (const double*)A, lda, (const double*)B, ldb, kt, (double*)C, ldc, grid, s, reverse);
  return launch_variant<float>(core, variant, a_kc, b_kc, (const float*)A, lda, (const float*)B, ldb, kt, (float*)C, ldc, grid, s, reverse);
}
