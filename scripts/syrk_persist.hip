// The bulk trailing update's tile (C -= A_i A_j^T, K = 128 nk, 128 x 128 tiles of the direct-to-LDS core, dgp_chol.hip::syrk_tile)
// as ONE WORKGROUP PER TILE (what the product launches) against RESIDENT workgroups that walk the tile list -- VERDICT r4 item 5:
// how much of the K = 512 tile's distance from the K = 8192 rate (64 vs 76.5 TFLOP/s) is the per-tile dispatch / prologue / epilogue?
//   MODE 0  one workgroup per tile
//   MODE 1  grid = resident workgroups, tile t = b, b + grid, ...; every tile runs the whole DmaGemm::run (ring refilled per tile)
//   MODE 2  like 1, but the ring is CARRIED across the tile boundary: the next tile's first chunks are issued while the current
//           tile's last chunks are multiplied and its accumulators are stored (DmaGemm::run_stream below)
// build: hipcc -O3 -std=c++20 --offload-arch=gfx950 -Idiscontinuum_amd/csrc scripts/syrk_persist.hip -o scripts/syrk_persist
// usage: scripts/syrk_persist [nt=90] [nk=4] [grid=768] [reps=5]
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
static constexpr int NB = 128;

// CIO: 3 = read-modify-write of C (the product), 2 = store only (accumulators start at zero), 0 = no C traffic at all
// (one element per lane stored so that the MFMAs stay live)
template <typename T, int CIO = 3>
__device__ __forceinline__ void one_tile(T* A, long ld, int nk, int bi, int bj, T* smem) {
  using K = TileCore<T, true, true, 128, 128, 1>;
  using G = typename K::G;
  const long row0 = (long)(bi + nk) * NB, col0 = (long)(bj + nk) * NB;
  typename G::acc_t acc[G::MI][G::NI], keep[G::MI][G::NI];
  T* C = A + row0 * ld + col0;
  // CIO 1: load only; 4: read-modify-write with NON-TEMPORAL loads and stores (C is read once and written once per launch:
  // it need not displace the operand panels from L2); 5: non-temporal loads only; 6: non-temporal stores only
  if (CIO == 3 || CIO == 1 || CIO == 6) trailing_begin<T, G, K::DMA>(acc, keep, C, ld);
  else if (CIO == 4 || CIO == 5) G::foreach (acc, [&](int r, int c, T& v) { v = -__builtin_nontemporal_load(&C[(long)r * ld + c]); });
  else if (CIO == 10) {
    // scalar row bases + ONE 32-bit lane offset: the compiler can use the saddr form of global_load (no 64-bit address per load)
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned loff = (unsigned)((((long)(lane >> 4)) * ld + (lane & 15)) * 8);
    const char* Cw = (const char*)(C + (long)((w >> 1) * 64) * ld + (w & 1) * 64);
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const char* rp = Cw + (long)(mi * 16 + 4 * r) * ld * 8;
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) acc[mi][ni][r] = -*reinterpret_cast<const double*>(rp + ni * 128 + loff);
      }
  } else if (CIO == 7 || CIO == 8 || CIO == 9) {
    // TIMING ONLY (values land in the wrong accumulator slots): the tile of C through 16-byte loads / stores, half as many
    // instructions -- lane l covers row (l >> 3) + 8 p of its wave's 64 x 64 quadrant, columns 8 q + 2 (l & 7) .. + 1
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const T* Cq = C + (long)((w >> 1) * 64) * ld + (w & 1) * 64;
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int slab = (mi * G::NI + ni) * 2 + h;  // 32 slabs of 8 rows x 16 columns... laid out as 8 rows x 64 columns / 4
          const long r = (slab >> 2) * 8 + (lane >> 3), c = (slab & 3) * 16 + 2 * (lane & 7);
          const dgp_d2 v = CIO == 9 ? __builtin_nontemporal_load(reinterpret_cast<const dgp_d2*>(Cq + r * ld + c)) : *reinterpret_cast<const dgp_d2*>(Cq + r * ld + c);
          acc[mi][ni][2 * h] = -v[0];
          acc[mi][ni][2 * h + 1] = -v[1];
        }
  } else G::zero(acc);
  K::run(A + row0 * ld, ld, A + col0 * ld, ld, nk * (NB / 16), smem, acc);
  if (CIO == 10) {
    const int lane = threadIdx.x & 63, w = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    const unsigned loff = (unsigned)((((long)(lane >> 4)) * ld + (lane & 15)) * 8);
    char* Cw = (char*)(C + (long)((w >> 1) * 64) * ld + (w & 1) * 64);
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        char* rp = Cw + (long)(mi * 16 + 4 * r) * ld * 8;
#pragma unroll
        for (int ni = 0; ni < G::NI; ++ni) *reinterpret_cast<double*>(rp + ni * 128 + loff) = -acc[mi][ni][r];
      }
    return;
  }
  if (CIO == 8 || CIO == 9) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    T* Cq = C + (long)((w >> 1) * 64) * ld + (w & 1) * 64;
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          const int slab = (mi * G::NI + ni) * 2 + h;
          const long r = (slab >> 2) * 8 + (lane >> 3), c = (slab & 3) * 16 + 2 * (lane & 7);
          dgp_d2 v = {-acc[mi][ni][2 * h], -acc[mi][ni][2 * h + 1]};
          if (CIO == 9) __builtin_nontemporal_store(v, reinterpret_cast<dgp_d2*>(Cq + r * ld + c));
          else *reinterpret_cast<dgp_d2*>(Cq + r * ld + c) = v;
        }
    return;
  }
  if (CIO == 2 || CIO == 3 || CIO == 5 || CIO == 7) trailing_end<T, G, K::DMA>(acc, keep, C, ld);
  else if (CIO == 4 || CIO == 6) G::foreach (acc, [&](int r, int c, T& v) { __builtin_nontemporal_store(-v, &C[(long)r * ld + c]); });
  else {
    T sum = T(0);
#pragma unroll
    for (int mi = 0; mi < G::MI; ++mi)
#pragma unroll
      for (int ni = 0; ni < G::NI; ++ni)
#pragma unroll
        for (int r = 0; r < 4; ++r) sum += acc[mi][ni][r];
    C[(long)(threadIdx.x >> 4) * ld + (threadIdx.x & 15)] = sum;
  }
}

template <typename T, int MODE, int CIO>
__global__ __launch_bounds__(256, 3) void syrk_bench_kernel(T* A, long ld, int nk, int ntiles) {
  __shared__ T smem[TileCore<T, true, true>::SMEM_ELEMS];
  if (MODE == 0) {
    int bi, bj;
    tri_decode(xcd_remap((int)blockIdx.x, ntiles), bi, bj);
    one_tile<T, CIO>(A, ld, nk, bi, bj, smem);
  } else {
    for (int t = (int)blockIdx.x; t < ntiles; t += (int)gridDim.x) {
      int bi, bj;
      tri_decode(t, bi, bj);
      one_tile<T, CIO>(A, ld, nk, bi, bj, smem);
    }
  }
}

int main(int argc, char** argv) {
  const int nt = argc > 1 ? atoi(argv[1]) : 90, nk = argc > 2 ? atoi(argv[2]) : 4, grid = argc > 3 ? atoi(argv[3]) : 768;
  const int reps = argc > 4 ? atoi(argv[4]) : 5;
  const long N = (long)(nt + nk) * NB;
  const int ntiles = nt * (nt + 1) / 2;
  double* A;
  CK(hipMalloc(&A, N * N * sizeof(double)));
  std::vector<double> h((size_t)N * N);
  srand(1);
  for (auto& v : h) v = (rand() / (double)RAND_MAX - 0.5) * 1e-3;
  CK(hipMemcpy(A, h.data(), N * N * sizeof(double), hipMemcpyHostToDevice));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  const double flop = 2.0 * NB * NB * (double)(nk * NB) * ntiles;
  auto timeit = [&](int mode, int cio) -> double {
    float best = 1e30f;
    for (int r = 0; r < reps + 1; ++r) {
      hipEventRecord(e0);
      for (int q = 0; q < 4; ++q) {
#define LAUNCH(CIO_)                                                                            \
  if (cio == CIO_) {                                                                            \
    if (mode == 0) syrk_bench_kernel<double, 0, CIO_><<<ntiles, 256>>>(A, N, nk, ntiles);       \
    else syrk_bench_kernel<double, 1, CIO_><<<grid, 256>>>(A, N, nk, ntiles);                   \
  }
        LAUNCH(0) LAUNCH(1) LAUNCH(2) LAUNCH(3) LAUNCH(4) LAUNCH(5) LAUNCH(6) LAUNCH(7) LAUNCH(8) LAUNCH(9) LAUNCH(10)
      }
      hipEventRecord(e1);
      hipEventSynchronize(e1);
      float ms;
      hipEventElapsedTime(&ms, e0, e1);
      if (r > 0 && ms < best) best = ms;
    }
    return best / 4.0;
  };
  const int maxmode = getenv("PERSIST") ? 2 : 1;
  for (int cio : {3, 10, 8, 9, 4, 0})
    for (int mode = 0; mode < maxmode; ++mode) {
      const double ms = timeit(mode, cio);
      printf("MODE %d CIO %d  nt=%d (%d tiles) K=%d grid=%d: %.3f ms  %.2f TFLOP/s\n", mode, cio, nt, ntiles, nk * NB, mode ? grid : ntiles, ms,
             flop / (ms * 1e-3) / 1e12);
    }
  CK(hipGetLastError());
  return 0;
}
