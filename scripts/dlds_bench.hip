// Feasibility bench: fp64 128x128 tile GEMM (A, B both k-contiguous) whose operands go global -> LDS
// directly (global_load_lds_dwordx4), NSTAGE-deep LDS ring, one barrier per k-tile; compared with the
// register-staged TileGemm core.  Standalone executable.
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_gemm.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef __attribute__((address_space(1))) const void* gptr;
typedef __attribute__((address_space(3))) void* lptr;

template <int N>
__device__ __forceinline__ void wait_vm() {  // s_waitcnt vmcnt(N) only
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
}

// LDS image of one operand k-tile (128 rows x 16 k): 16-byte granule (row, kp) = k's (2kp, 2kp+1) at
// granule index kp * 128 + row; a wave instruction fills 64 consecutive granules (64 rows of one kp).
template <int NSTAGE>
__global__ __launch_bounds__(256, 2) void gemm_dlds(const double* __restrict__ A, const double* __restrict__ B,
                                                    double* __restrict__ C, long n, int ktiles) {
  extern __shared__ double smem[];  // NSTAGE x (A 2048 + B 2048) doubles
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const double* a0 = A + (bi * 128 + lane) * n;
  const double* a1 = a0 + 64 * n;
  const double* b0 = B + (bj * 128 + lane) * n;
  const double* b1 = b0 + 64 * n;
  auto issue = [&](int kt, int stage) {
    double* sA = smem + stage * 4096;
    double* sB = sA + 2048;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = w * 4 + j, kp = w * 2 + (j >> 1);
      __builtin_amdgcn_global_load_lds((gptr)(((j & 1) ? a1 : a0) + kt * 16 + kp * 2), (lptr)(sA + c * 128), 16, 0, 0);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = w * 4 + j, kp = w * 2 + (j >> 1);
      __builtin_amdgcn_global_load_lds((gptr)(((j & 1) ? b1 : b0) + kt * 16 + kp * 2), (lptr)(sB + c * 128), 16, 0, 0);
    }
  };
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = dgp_d4{0, 0, 0, 0};
#pragma unroll
  for (int s = 0; s < NSTAGE - 1; ++s)
    if (s < ktiles) issue(s, s);
  const int r = lane & 15, kq = lane >> 4;
  const int foff = ((kq >> 1) * 128 + r) * 2 + (kq & 1);
  int stage = 0;
  for (int kt = 0; kt < ktiles; ++kt) {
    if (kt + NSTAGE - 2 < ktiles) wait_vm<(NSTAGE - 2) * 8>();
    else wait_vm<0>();
    __syncthreads();
    if (kt + NSTAGE - 1 < ktiles) {
      int ns = stage + NSTAGE - 1;
      if (ns >= NSTAGE) ns -= NSTAGE;
      issue(kt + NSTAGE - 1, ns);
    }
    const double* sA = smem + stage * 4096;
    const double* sB = sA + 2048;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      double fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = sA[foff + (ks * 2 * 128 + wm + mi * 16) * 2];
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = sB[foff + (ks * 2 * 128 + wn + ni * 16) * 2];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = Mfma<double>::mma(fa[mi], fb[ni], acc[mi][ni]);
    }
    if (++stage == NSTAGE) stage = 0;
  }
  double* out = C + bi * 128 * n + bj * 128;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int q = 0; q < 4; ++q)
        out[(long)(wm + mi * 16 + Mfma<double>::crow(lane, q)) * n + wn + ni * 16 + (lane & 15)] = acc[mi][ni][q];
}

__global__ __launch_bounds__(256, 2) void gemm_ref(const double* A, const double* B, double* C, long n, int ktiles) {
  using G = TileGemm<double, true, true, 128, 128>;
  __shared__ double smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  G::run(A + bi * 128 * n, n, B + bj * 128 * n, n, ktiles, smem, acc);
  double* out = C + bi * 128 * n + bj * 128;
  G::foreach (acc, [&](int r, int c, double& v) { out[(long)r * n + c] = v; });
}

template <int NSTAGE>
int run(long n, int K, const double* A, const double* B, double* C, double* Cref, std::vector<double>& h0, std::vector<double>& h1) {
  const size_t bytes = (size_t)NSTAGE * 4096 * sizeof(double);
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_dlds<NSTAGE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes));
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  printf("launch NSTAGE=%d\n", NSTAGE);
  gemm_dlds<NSTAGE><<<grid, 256, bytes>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  printf("ran\n");
  CK(hipMemcpy(h0.data(), C, n * n * sizeof(double), hipMemcpyDeviceToHost));
  double maxd = 0;
  for (long i = 0; i < n * n; ++i) { double d = fabs(h0[i] - h1[i]); if (d > maxd) maxd = d; }
  const int reps = 5;
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) gemm_dlds<NSTAGE><<<grid, 256, bytes>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= reps;
  printf("direct-to-LDS NSTAGE=%d n=%ld K=%d: %.3f ms  %.1f TFLOP/s   max|diff vs register-staged| = %.3g\n", NSTAGE, n, K, ms,
         2.0 * n * n * K / ms / 1e9, maxd);
  return 0;
}

int main(int argc, char** argv) {
  setvbuf(stdout, nullptr, _IONBF, 0);
  const long n = argc > 1 ? atol(argv[1]) : 8192;
  for (int K : {(int)n, 256}) {
    double *A, *B, *C, *Cref;
    CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&Cref, n * n * 8));
    std::vector<double> h(n * n), h0(n * n), h1(n * n);
    for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
    for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
    CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
    dim3 grid(n / 128, n / 128);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    printf("A=%p B=%p C=%p Cref=%p\n", (void*)A, (void*)B, (void*)C, (void*)Cref);
    gemm_ref<<<grid, 256>>>(A, B, Cref, n, K / 16); CK(hipDeviceSynchronize());
    printf("ref ok\n");
    CK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) gemm_ref<<<grid, 256>>>(A, B, Cref, n, K / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
    printf("register-staged         n=%ld K=%d: %.3f ms  %.1f TFLOP/s\n", n, K, ms, 2.0 * n * n * K / ms / 1e9);
    CK(hipMemcpy(h1.data(), Cref, n * n * 8, hipMemcpyDeviceToHost));
    if (run<2>(n, K, A, B, C, Cref, h0, h1)) return 1;
    if (run<3>(n, K, A, B, C, Cref, h0, h1)) return 1;
    if (run<4>(n, K, A, B, C, Cref, h0, h1)) return 1;
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(Cref));
  }
  return 0;
}
