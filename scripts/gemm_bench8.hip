// Feasibility bench: 128 x 256 output tile per workgroup of EIGHT waves (2 x 4), against the 4-wave 128 x 128 core,
// for the bulk update's shape (C -= A_i A_j^T, both operands k-contiguous, K = 256 ... 8192).  Standalone.
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_gemm.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

// LDS image [row][17] for both operands; 512 threads stage A (128 x 16: 4 elems/thread) and B (256 x 16: 8/thread)
template <int RMW>
__global__ __launch_bounds__(512, 1) void gemm8(const double* __restrict__ A, const double* __restrict__ B,
                                                double* __restrict__ C, long n, int ktiles) {
  constexpr int BM = 128, BN = 256, S = 17;
  __shared__ double sA[BM * S], sB[BN * S];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = (w >> 2) * 64, wn = (w & 3) * 64;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const double* a = A + bi * BM * n;
  const double* b = B + bj * BN * n;
  // staging: A: thread t -> row t/4, k offset (t%4)*4 (4 elems); B: row t/2, k offset (t%2)*8 (8 elems)
  const int ar = t >> 2, ac = (t & 3) * 4, br = t >> 1, bc = (t & 1) * 8;
  dgp_d4 acc[4][4];
  double* Ct = C + bi * BM * n + bj * BN;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        acc[mi][ni][r] = RMW ? -Ct[(long)(wm + mi * 16 + Mfma<double>::crow(lane, r)) * n + wn + ni * 16 + (lane & 15)] : 0.0;
  dgp_d2 ra[2], rb[4];
  auto load = [&](int kt) {
    const double* pa = a + (long)ar * n + kt * 16 + ac;
    ra[0] = *(const dgp_d2*)pa; ra[1] = *(const dgp_d2*)(pa + 2);
    const double* pb = b + (long)br * n + kt * 16 + bc;
#pragma unroll
    for (int v = 0; v < 4; ++v) rb[v] = *(const dgp_d2*)(pb + 2 * v);
  };
  load(0);
  for (int kt = 0; kt < ktiles; ++kt) {
    __syncthreads();
#pragma unroll
    for (int v = 0; v < 2; ++v) { sA[ar * S + ac + 2 * v] = ra[v][0]; sA[ar * S + ac + 2 * v + 1] = ra[v][1]; }
#pragma unroll
    for (int v = 0; v < 4; ++v) { sB[br * S + bc + 2 * v] = rb[v][0]; sB[br * S + bc + 2 * v + 1] = rb[v][1]; }
    __syncthreads();
    if (kt + 1 < ktiles) load(kt + 1);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = sA[(wm + mi * 16 + (lane & 15)) * S + ks * 4 + (lane >> 4)];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = sB[(wn + ni * 16 + (lane & 15)) * S + ks * 4 + (lane >> 4)];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Mfma<double>::mma(fa[mi], fb[ni], acc[mi][ni]);
    }
  }
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        Ct[(long)(wm + mi * 16 + Mfma<double>::crow(lane, r)) * n + wn + ni * 16 + (lane & 15)] = RMW ? -acc[mi][ni][r] : acc[mi][ni][r];
}

template <int RMW>
__global__ __launch_bounds__(256, 2) void gemm4(const double* A, const double* B, double* C, long n, int ktiles) {
  using G = TileGemm<double, true, true, 128, 128>;
  __shared__ double smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  double* out = C + bi * 128 * n + bj * 128;
  if (RMW) G::foreach (acc, [&](int r, int c, double& v) { v = -out[(long)r * n + c]; });
  else G::zero(acc);
  G::run(A + bi * 128 * n, n, B + bj * 128 * n, n, ktiles, smem, acc);
  G::foreach (acc, [&](int r, int c, double& v) { out[(long)r * n + c] = RMW ? -v : v; });
}

int main() {
  const long n = 8192;
  double *A, *B, *C, *D;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&D, n * n * 8));
  std::vector<double> h(n * n), c0(n * n), c1(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int K : {256, 512, 8192}) {
    for (int rmw = 0; rmw < 2; ++rmw) {
      float ms4, ms8;
      CK(hipMemset(C, 0, n * n * 8)); CK(hipMemset(D, 0, n * n * 8));
      dim3 g4(n / 128, n / 128), g8(n / 256, n / 128);
      auto run4 = [&]() { if (rmw) gemm4<1><<<g4, 256>>>(A, B, C, n, K / 16); else gemm4<0><<<g4, 256>>>(A, B, C, n, K / 16); };
      auto run8 = [&]() { if (rmw) gemm8<1><<<g8, 512>>>(A, B, D, n, K / 16); else gemm8<0><<<g8, 512>>>(A, B, D, n, K / 16); };
      run4(); run8(); CK(hipDeviceSynchronize());
      CK(hipMemcpy(c0.data(), C, n * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c1.data(), D, n * n * 8, hipMemcpyDeviceToHost));
      double md = 0; for (long i = 0; i < n * n; i += 97) md = fmax(md, fabs(c0[i] - c1[i]));
      const int reps = K > 1000 ? 3 : 10;
      CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) run4(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms4, e0, e1)); ms4 /= reps;
      CK(hipEventRecord(e0)); for (int r = 0; r < reps; ++r) run8(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
      CK(hipEventElapsedTime(&ms8, e0, e1)); ms8 /= reps;
      printf("K=%5d %s: 4-wave 128x128 %.1f TFLOP/s | 8-wave 128x256 %.1f TFLOP/s   (max diff %.2g)\n", K, rmw ? "C -= AB^T" : "C  = AB^T",
             2.0 * n * n * K / ms4 / 1e9, 2.0 * n * n * K / ms8 / 1e9, md);
    }
  }
  return 0;
}
