// A barrier-free variant of the tile core: every WAVE stages its own 64 x 16 operand sub-panels in a private LDS region
// (global -> registers -> LDS -> MFMA fragments), so no __syncthreads couples the four waves of a workgroup -- LDS
// instructions of one wave execute in order, which is all the write -> read hand-over inside a wave needs.  Price: each
// operand sub-panel is loaded by the two waves that share it (2x the L1/L2 requests, 2x the LDS writes).
// Compared here with the product's TileGemm (two barriers per k-tile) on the same C = A^T B / C = A B^T problems.
// build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -Idiscontinuum_amd/csrc scripts/gemm_wave.hip -o scripts/gemm_wave
#include <stdio.h>
#include <vector>
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, bool AKC, bool BKC, int PF>
__global__ __launch_bounds__(256, 2) void gemm_ref_kernel(const T* A, const T* B, T* C, long n, int ktiles) {
  using G = TileGemm<T, AKC, BKC, 128, 128>;
  __shared__ T smem[G::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* a = AKC ? A + bi * 128 * n : A + bi * 128;
  const T* b = BKC ? B + bj * 128 * n : B + bj * 128;
  G::template run<PF>(a, n, b, n, ktiles, smem, acc);
  T* out = C + bi * 128 * n + bj * 128;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * n + c] = v; });
}

// ---- wave-private staging, fp64 ------------------------------------------------------------------------------------
// IC operand (op(i, k) = p[k ld + i]): LDS image [16 k][64 i] doubles, element (k, i) at k*64 + (i ^ 16 (k & 1)): the four
//   k's of a fragment read land in both 128-byte halves of the 256-byte bank row without padding.
// KC operand (op(i, k) = p[i ld + k]): LDS image [64 i][16 k + 1 pad]: lane l loads a 16-byte pair (i = l / 8 + 8 v, k = 2 (l & 7)).
template <bool KC, int NK>
struct WaveTile {
  static constexpr int NV = NK / 2;  // 16-byte loads per lane per stage
  static constexpr int ELEMS = KC ? 64 * (NK + 1) : NK * 64;
  static __device__ __forceinline__ void load(const double* __restrict__ p, long ld, dgp_d2 (&r)[NV], int lane) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (KC) r[v] = *reinterpret_cast<const dgp_d2*>(p + (long)((128 / NK) * v + lane / (NK / 2)) * ld + 2 * (lane % (NK / 2)));
      else r[v] = *reinterpret_cast<const dgp_d2*>(p + (long)(2 * v + (lane >> 5)) * ld + 2 * (lane & 31));
    }
  }
  static __device__ __forceinline__ void store(double* __restrict__ s, const dgp_d2 (&r)[NV], int lane) {
#pragma unroll
    for (int v = 0; v < NV; ++v) {
      if (KC) {
        double* d = s + ((128 / NK) * v + lane / (NK / 2)) * (NK + 1) + 2 * (lane % (NK / 2));
        d[0] = r[v][0];
        d[1] = r[v][1];
      } else {
        const int k = 2 * v + (lane >> 5), i = 2 * (lane & 31);
        *reinterpret_cast<dgp_d2*>(s + k * 64 + (i ^ (16 * (k & 1)))) = r[v];
      }
    }
  }
  static __device__ __forceinline__ double frag(const double* __restrict__ s, int mi, int ks, int lane) {
    const int k = ks * 4 + (lane >> 4), i = mi * 16 + (lane & 15);
    if (KC) return s[i * (NK + 1) + k];
    return s[k * 64 + (i ^ (16 * (k & 1)))];
  }
};

template <bool AKC, bool BKC, int NK>
__global__ __launch_bounds__(256, 2) void gemm_wave_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                           int ktiles) {
  using WA = WaveTile<AKC, NK>;
  using WB = WaveTile<BKC, NK>;
  extern __shared__ double smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  double* sA = smem + w * (WA::ELEMS + WB::ELEMS);
  double* sB = sA + WA::ELEMS;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const double* a = AKC ? A + (bi * 128 + wm) * n : A + bi * 128 + wm;
  const double* b = BKC ? B + (bj * 128 + wn) * n : B + bj * 128 + wn;
  const long stepA = AKC ? NK : NK * n, stepB = BKC ? NK : NK * n;
  const int stages = ktiles * 16 / NK;
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  dgp_d2 ra[WA::NV], rb[WB::NV];
  WA::load(a, n, ra, lane);
  WB::load(b, n, rb, lane);
  for (int kt = 0; kt < stages; ++kt) {
    WA::store(sA, ra, lane);
    WB::store(sB, rb, lane);
    __builtin_amdgcn_wave_barrier();  // compiler-only: the fragment reads below stay behind the stores
    a += stepA;
    b += stepB;
    if (kt + 1 < stages) {
      WA::load(a, n, ra, lane);
      WB::load(b, n, rb, lane);
    }
#pragma unroll
    for (int ks = 0; ks < NK / 4; ++ks) {
      double fa[4], fb[4];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi) fa[mi] = WA::frag(sA, mi, ks, lane);
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) fb[ni] = WB::frag(sB, ni, ks, lane);
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
}

// ---- wave-private staging, DOUBLE-buffered in LDS and software-pipelined: stages of 8 k's (two MFMA k-steps); while the MFMAs
// of stage s run, the wave writes stage s + 1 into its other LDS buffer, issues the global loads of stage s + 2 and reads the
// first fragments of stage s + 1 -- no barrier, no point where the wave has nothing but LDS latency to wait for.
// LDS image of either operand kind: [8 k][64 i] doubles, element (k, i) at k*64 + (i ^ 16 ((k ^ (k >> 2)) & 3)).
template <bool KC>
struct PipeTile {
  static constexpr int ELEMS = 8 * 64;
  static __device__ __forceinline__ int sw(int k) { return 16 * ((k ^ (k >> 2)) & 3); }
  static __device__ __forceinline__ void load(const double* __restrict__ p, long ld, dgp_d2 (&r)[4], int lane) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (KC) r[v] = *reinterpret_cast<const dgp_d2*>(p + (long)(16 * v + (lane >> 2)) * ld + 2 * (lane & 3));
      else r[v] = *reinterpret_cast<const dgp_d2*>(p + (long)(2 * v + (lane >> 5)) * ld + 2 * (lane & 31));
    }
  }
  static __device__ __forceinline__ void store(double* __restrict__ s, const dgp_d2 (&r)[4], int lane) {
#pragma unroll
    for (int v = 0; v < 4; ++v) {
      if (KC) {
        const int i = 16 * v + (lane >> 2), k = 2 * (lane & 3);
        s[k * 64 + (i ^ sw(k))] = r[v][0];
        s[(k + 1) * 64 + (i ^ sw(k + 1))] = r[v][1];
      } else {
        const int k = 2 * v + (lane >> 5), i = 2 * (lane & 31);
        *reinterpret_cast<dgp_d2*>(s + k * 64 + (i ^ sw(k))) = r[v];
      }
    }
  }
  static __device__ __forceinline__ double frag(const double* __restrict__ s, int mi, int ks, int lane) {
    const int k = ks * 4 + (lane >> 4), i = mi * 16 + (lane & 15);
    return s[k * 64 + (i ^ sw(k))];
  }
};

template <bool AKC, bool BKC>
__global__ __launch_bounds__(256, 2) void gemm_pipe_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                           int ktiles) {
  using WA = PipeTile<AKC>;
  using WB = PipeTile<BKC>;
  extern __shared__ double smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  double* sA0 = smem + w * 2048;  // per wave: A0 B0 A1 B1
  double* sB0 = sA0 + 512;
  double* sA1 = sA0 + 1024;
  double* sB1 = sA0 + 1536;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const double* a = AKC ? A + (bi * 128 + wm) * n : A + bi * 128 + wm;
  const double* b = BKC ? B + (bj * 128 + wn) * n : B + bj * 128 + wn;
  const long stepA = AKC ? 8 : 8 * n, stepB = BKC ? 8 : 8 * n;
  const int stages = ktiles * 2;  // even
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  dgp_d2 ra[4], rb[4];
  double fa0[4], fb0[4], fa1[4], fb1[4];
  WA::load(a, n, ra, lane);
  WB::load(b, n, rb, lane);
  a += stepA; b += stepB;
  WA::store(sA0, ra, lane);
  WB::store(sB0, rb, lane);
  WA::load(a, n, ra, lane);
  WB::load(b, n, rb, lane);
  a += stepA; b += stepB;
  __builtin_amdgcn_wave_barrier();
#pragma unroll
  for (int i = 0; i < 4; ++i) fa0[i] = WA::frag(sA0, i, 0, lane), fb0[i] = WB::frag(sB0, i, 0, lane);
  auto mma = [&](const double (&fa)[4], const double (&fb)[4]) {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
  };
  auto stage = [&](double* cA, double* cB, double* nA, double* nB, int kt) {
#pragma unroll
    for (int i = 0; i < 4; ++i) fa1[i] = WA::frag(cA, i, 1, lane), fb1[i] = WB::frag(cB, i, 1, lane);
    mma(fa0, fb0);
    if (kt + 1 < stages) {
      WA::store(nA, ra, lane);
      WB::store(nB, rb, lane);
      __builtin_amdgcn_wave_barrier();
      if (kt + 2 < stages) {
        WA::load(a, n, ra, lane);
        WB::load(b, n, rb, lane);
        a += stepA; b += stepB;
      }
#pragma unroll
      for (int i = 0; i < 4; ++i) fa0[i] = WA::frag(nA, i, 0, lane), fb0[i] = WB::frag(nB, i, 0, lane);
    }
    mma(fa1, fb1);
  };
  for (int kt = 0; kt < stages; kt += 2) {
    stage(sA0, sB0, sA1, sB1, kt);
    stage(sA1, sB1, sA0, sB0, kt + 1);
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
}

template <bool AKC, bool BKC>
int run_pipe(const char* name, long n, int K) {
  double *A, *B, *C, *C2;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&C2, n * n * 8));
  std::vector<double> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 4 * 2048 * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_pipe_kernel<AKC, BKC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_pipe_kernel<AKC, BKC>, 256, lds));
  float ms_ref, ms_w;
  gemm_ref_kernel<double, AKC, BKC, 2><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) gemm_ref_kernel<double, AKC, BKC, 2><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_ref, e0, e1)); ms_ref /= 5;
  gemm_pipe_kernel<AKC, BKC><<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) gemm_pipe_kernel<AKC, BKC><<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_w, e0, e1)); ms_w /= 5;
  std::vector<double> c1(n * n), c2(n * n);
  CK(hipMemcpy(c1.data(), C, n * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, n * n * 8, hipMemcpyDeviceToHost));
  double md = 0;
  for (long i = 0; i < n * n; ++i) md = fmax(md, fabs(c1[i] - c2[i]));
  const double fl = 2.0 * n * n * K / 1e9;
  printf("%-10s pipelined n %ld K %d: core PF2 %.3f ms %.1f TF | wave-private double-buffered %.3f ms %.1f TF (occupancy %d/CU) | max |diff| %.3g\n", name, n, K,
         ms_ref, fl / ms_ref, ms_w, fl / ms_w, occ, md);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2));
  return 0;
}

// ---- wave-private, operands global -> LDS DIRECTLY (global_load_lds_dwordx4: no staging registers, no ds_write -- the parts
// test below prices the LDS stores' register reads at 6-7 % of the MFMA rate), ring of NST stages of 4 k's per wave, no barrier.
// IC operands only: a load instruction fills two k-rows of 64 doubles ([k][64] in LDS, lane order); the XOR swizzle that
// spreads a fragment read's four k's over the banks is applied on the GLOBAL side (lane l fetches i ^ 16 k), which permutes
// 128-byte groups inside the same 512-byte row and leaves coalescing alone.
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
template <int N>
__device__ __forceinline__ void wait_vm() {  // s_waitcnt vmcnt(N) only
  __builtin_amdgcn_s_waitcnt((N & 0xF) | ((N >> 4) << 14) | 0x0F70);
  asm volatile("" ::: "memory");
}

template <int NST>
__global__ __launch_bounds__(256, 2) void gemm_dma_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                          int ktiles) {
  extern __shared__ double smem[];  // per wave NST x (A 256 + B 256) doubles
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  double* ring = smem + w * NST * 512;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const int k0 = lane >> 5, k1 = 2 + (lane >> 5), i2 = 2 * (lane & 31);
  const long off0 = (long)k0 * n + (i2 ^ (16 * k0)), off1 = (long)k1 * n + (i2 ^ (16 * k1));
  const double* a = A + bi * 128 + wm;
  const double* b = B + bj * 128 + wn;
  const long step = 4 * n;
  auto issue = [&](int slot) {
    double* d = ring + slot * 512;
    __builtin_amdgcn_global_load_lds((gptr_t)(a + off0), (lptr_t)(d), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(a + off1), (lptr_t)(d + 128), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(b + off0), (lptr_t)(d + 256), 16, 0, 0);
    __builtin_amdgcn_global_load_lds((gptr_t)(b + off1), (lptr_t)(d + 384), 16, 0, 0);
    a += step;
    b += step;
  };
  const int fk = lane >> 4, fr = lane & 15;
  int foff[4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi) foff[mi] = fk * 64 + ((mi ^ fk) & 3) * 16 + fr;
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  const int S = ktiles * 4;
#pragma unroll
  for (int s = 0; s < NST - 1; ++s) issue(s);
  double f[2][8];
  wait_vm<(NST - 2) * 4>();
#pragma unroll
  for (int i = 0; i < 4; ++i) f[0][i] = ring[foff[i]], f[0][4 + i] = ring[256 + foff[i]];
  int s0 = 0;
  // steady state, branch-free: every stage of the round still has a stage NST - 1 ahead to issue
  for (; s0 + 2 * NST - 2 < S; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      wait_vm<(NST - 3) * 4>();  // stage s + 1 has landed once at most the NST - 3 stages issued after it are outstanding
      issue((u + NST - 1) % NST);  // into the slot of stage s - 1 (its fragments were consumed)
      const double* d = ring + ((u + 1) % NST) * 512;
#pragma unroll
      for (int i = 0; i < 4; ++i) f[(u + 1) & 1][i] = d[foff[i]], f[(u + 1) & 1][4 + i] = d[256 + foff[i]];
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u & 1][mi], f[u & 1][4 + ni], acc[mi][ni], 0, 0, 0);
    }
  }
  for (; s0 < S; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int s = s0 + u;
      if (s + NST - 2 < S) wait_vm<(NST - 3) * 4>();
      else wait_vm<0>();
      if (s + NST - 1 < S) issue((u + NST - 1) % NST);
      if (s + 1 < S) {
        const double* d = ring + ((u + 1) % NST) * 512;
#pragma unroll
        for (int i = 0; i < 4; ++i) f[(u + 1) & 1][i] = d[foff[i]], f[(u + 1) & 1][4 + i] = d[256 + foff[i]];
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni)
          acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[u & 1][mi], f[u & 1][4 + ni], acc[mi][ni], 0, 0, 0);
    }
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
}

// The same with THREE workgroups per CU (the parts test reaches 77 TFLOP/s exactly when its register count lets a third wave onto
// every SIMD): accumulators 128 + one set of fragments 16 + addresses <= 168 VGPRs, ring of 3 stages = 12 KB per wave, 48 KB per
// workgroup.  The fragment reads of the next stage follow the MFMAs of this one; the other two waves of the SIMD cover their latency.
__global__ __launch_bounds__(256, 3) void gemm_dma3_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                           int ktiles) {
  constexpr int NST = 3;
  extern __shared__ double smem[];
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  double* ring = smem + w * NST * 512;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const int k0 = lane >> 5, k1 = 2 + (lane >> 5), i2 = 2 * (lane & 31);
  // uniform base pointers (scalar registers) + one 32-bit byte offset per lane and instruction
  const unsigned off0 = (unsigned)(((long)k0 * n + (i2 ^ (16 * k0))) * 8), off1 = (unsigned)(((long)k1 * n + (i2 ^ (16 * k1))) * 8);
  const char* a = (const char*)(A + bi * 128 + wm);
  const char* b = (const char*)(B + bj * 128 + wn);
  const long step = 4 * n * 8;
  const unsigned ring_lds = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)ring);  // LDS byte address, scalar
  // written as assembly so that the base pointers STAY in scalar registers (the compiler turns a + off into four 64-bit
  // per-lane induction variables otherwise, and the kernel no longer fits 168 registers)
  auto dma = [&](unsigned voff, const char* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory", "m0");
  };
  auto issue = [&](int slot) {
    const unsigned d = ring_lds + slot * 4096;
    dma(off0, a, d);
    dma(off1, a, d + 1024);
    dma(off0, b, d + 2048);
    dma(off1, b, d + 3072);
    a += step;
    b += step;
  };
  const int fk = lane >> 4, fr = lane & 15;
  const int fbase = fk * 64 + fr;  // + ((mi ^ fk) & 3) * 16
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  const int S = ktiles * 4;
  issue(0);
  issue(1);
  double f[8];
  auto frags = [&](int slot) {
    const double* d = ring + slot * 512 + fbase;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = d[((i ^ fk) & 3) * 16], f[4 + i] = d[256 + ((i ^ fk) & 3) * 16];
  };
  auto mma = [&]() {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[mi], f[4 + ni], acc[mi][ni], 0, 0, 0);
  };
  wait_vm<4>();
  frags(0);
  int s0 = 0;
  for (; s0 + 2 * NST - 2 < S; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      mma();                      // stage s
      wait_vm<0>();               // stage s + 1 has landed
      issue((u + 2) % NST);       // stage s + 2 into the slot of stage s - 1
      frags((u + 1) % NST);
    }
  }
  for (; s0 < S; s0 += NST) {
#pragma unroll
    for (int u = 0; u < NST; ++u) {
      const int s = s0 + u;
      if (s < S) mma();
      wait_vm<0>();
      if (s + 2 < S) issue((u + 2) % NST);
      if (s + 1 < S) frags((u + 1) % NST);
    }
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
}

// Workgroup-SHARED ring, direct-to-LDS, three workgroups per CU: a stage (4 k's) is A [4][128] + B [4][128] doubles = 8 KB, wave w
// loads k-row w of both (one 1 KB instruction each), R stages in 48 KB (R = 6: five stages of prefetch distance instead of the
// wave-private ring's two), ONE barrier per stage (it both publishes stage s + 1 and frees the slot of stage s).
template <int R>
__global__ __launch_bounds__(256, 3) void gemm_dmas_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                           int ktiles) {
  extern __shared__ double smem[];  // R x (A 512 + B 512)
  const int t = threadIdx.x, lane = t & 63;
  const int w = __builtin_amdgcn_readfirstlane(t >> 6);
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const unsigned voff = (unsigned)(((2 * lane) ^ (16 * w)) * 8);
  const char* a = (const char*)(A + bi * 128 + (long)w * n);
  const char* b = (const char*)(B + bj * 128 + (long)w * n);
  const long step = 4 * n * 8;
  const unsigned lds0 = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(lptr_t)smem) + w * 1024;
  auto dma = [&](unsigned vo, const char* base, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(vo), "s"(base), "s"(lds) : "memory", "m0");
  };
  auto issue = [&](int slot) {
    dma(voff, a, lds0 + slot * 8192);
    dma(voff, b, lds0 + slot * 8192 + 4096);
    a += step;
    b += step;
  };
  const int fk = lane >> 4;
  const int fa0 = fk * 128 + wm + (lane & 15), fb0 = 512 + fk * 128 + wn + (lane & 15);
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  double f[8];
  auto frags = [&](int slot) {
    const double* d = smem + slot * 1024;
#pragma unroll
    for (int i = 0; i < 4; ++i) f[i] = d[fa0 + ((i ^ fk) & 3) * 16], f[4 + i] = d[fb0 + ((i ^ fk) & 3) * 16];
  };
  auto mma = [&]() {
#pragma unroll
    for (int mi = 0; mi < 4; ++mi)
#pragma unroll
      for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[mi], f[4 + ni], acc[mi][ni], 0, 0, 0);
  };
  const int S = ktiles * 4;
#pragma unroll
  for (int s = 0; s < R - 1; ++s)
    if (s < S) issue(s);
  if (S >= R - 1) wait_vm<2 * (R - 2)>();
  else wait_vm<0>();
  __syncthreads();
  frags(0);
  int s0 = 0;
  for (; s0 + 2 * R - 2 < S; s0 += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      mma();                       // stage s
      wait_vm<2 * (R - 3)>();      // this wave's rows of stage s + 1 have landed (stages s + 2 .. s + R - 2 may be in flight)
      __syncthreads();             // ... everyone's have; and everyone holds stage s in registers: its slot is free
      issue((u + R - 1) % R);      // stage s + R - 1 into the slot of stage s - 1
      frags((u + 1) % R);
    }
  }
  for (; s0 < S; s0 += R) {
#pragma unroll
    for (int u = 0; u < R; ++u) {
      const int s = s0 + u;
      if (s < S) mma();
      wait_vm<0>();
      __syncthreads();
      if (s + R - 1 < S) issue((u + R - 1) % R);
      if (s + 1 < S) frags((u + 1) % R);
    }
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
}

template <int NST>
int run_dma(long n, int K) {
  double *A, *B, *C, *C2;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&C2, n * n * 8));
  std::vector<double> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = NST >= 10 ? (size_t)(NST - 10) * 8192 : 4 * (NST ? NST : 3) * 512 * 8;
  auto kern = NST >= 10 ? gemm_dmas_kernel<(NST >= 10 ? NST - 10 : 6)> : NST ? gemm_dma_kernel<(NST && NST < 10 ? NST : 4)> : gemm_dma3_kernel;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, 256, lds));
  float ms_ref, ms_w;
  gemm_ref_kernel<double, false, false, 2><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) gemm_ref_kernel<double, false, false, 2><<<grid, 256>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_ref, e0, e1)); ms_ref /= 5;
  kern<<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) kern<<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_w, e0, e1)); ms_w /= 5;
  std::vector<double> c1(n * n), c2(n * n);
  CK(hipMemcpy(c1.data(), C, n * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, n * n * 8, hipMemcpyDeviceToHost));
  double md = 0;
  for (long i = 0; i < n * n; ++i) md = fmax(md, fabs(c1[i] - c2[i]));
  const double fl = 2.0 * n * n * K / 1e9;
  printf("f64 IC/IC direct-to-LDS ring %d n %ld K %d: core PF2 %.3f ms %.1f TF | wave-private DMA %.3f ms %.1f TF (LDS %zu B, occupancy %d/CU) | max |diff| %.3g\n", NST, n, K,
         ms_ref, fl / ms_ref, ms_w, fl / ms_w, lds, occ, md);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2));
  return 0;
}

// which part of the k-loop costs what: the wave-private IC/IC loop with its parts switched off one by one
// (results are garbage when a part is off; only the rate matters)
__device__ long long g_clk[4];  // shader-clock and 100 MHz wall-clock ticks of one workgroup in the middle of the grid
template <bool LOADS, bool STORES, bool READS>
__global__ __launch_bounds__(256, 2) void parts_kernel(const double* __restrict__ A, const double* __restrict__ B, double* __restrict__ C, long n,
                                                       int ktiles) {
  constexpr int NK = 8;
  const bool probe = blockIdx.x == gridDim.x / 2 && blockIdx.y == gridDim.y / 2 && threadIdx.x == 0;
  long long c0 = 0, w0 = 0;
  if (probe) c0 = clock64(), w0 = wall_clock64();
  using WA = WaveTile<false, NK>;
  extern __shared__ double smem[];
  const int t = threadIdx.x, lane = t & 63, w = t >> 6;
  const int wm = (w >> 1) * 64, wn = (w & 1) * 64;
  double* sA = smem + w * (2 * WA::ELEMS);
  double* sB = sA + WA::ELEMS;
  const long bi = blockIdx.y, bj = blockIdx.x;
  const double* a = A + bi * 128 + wm;
  const double* b = B + bj * 128 + wn;
  const long step = NK * n;
  const int stages = ktiles * 16 / NK;
  dgp_d4 acc[4][4];
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[mi][ni][r] = 0.0;
  dgp_d2 ra[WA::NV], rb[WA::NV];
  WA::load(a, n, ra, lane);
  WA::load(b, n, rb, lane);
  if (!STORES) {
    WA::store(sA, ra, lane);
    WA::store(sB, rb, lane);
  }
  double fa[4], fb[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) fa[i] = ra[i][0], fb[i] = rb[i][1];
  for (int kt = 0; kt < stages; ++kt) {
    if (STORES) {
      WA::store(sA, ra, lane);
      WA::store(sB, rb, lane);
    }
    __builtin_amdgcn_wave_barrier();
    a += step;
    b += step;
    if (LOADS && kt + 1 < stages) {
      WA::load(a, n, ra, lane);
      WA::load(b, n, rb, lane);
    }
#pragma unroll
    for (int ks = 0; ks < NK / 4; ++ks) {
      if (READS) {
#pragma unroll
        for (int mi = 0; mi < 4; ++mi) fa[mi] = WA::frag(sA, mi, ks, lane);
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) fb[ni] = WA::frag(sB, ni, ks, lane);
      }
#pragma unroll
      for (int mi = 0; mi < 4; ++mi)
#pragma unroll
        for (int ni = 0; ni < 4; ++ni) acc[mi][ni] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[mi], fb[ni], acc[mi][ni], 0, 0, 0);
    }
    __builtin_amdgcn_wave_barrier();
  }
  double* out = C + (bi * 128 + wm) * n + bj * 128 + wn;
#pragma unroll
  for (int mi = 0; mi < 4; ++mi)
#pragma unroll
    for (int ni = 0; ni < 4; ++ni)
#pragma unroll
      for (int r = 0; r < 4; ++r) out[(long)(mi * 16 + (lane >> 4) + 4 * r) * n + ni * 16 + (lane & 15)] = acc[mi][ni][r];
  if (probe) g_clk[0] = clock64() - c0, g_clk[1] = wall_clock64() - w0;
}

template <bool LOADS, bool STORES, bool READS>
int parts(const double* A, const double* B, double* C, long n, int K) {
  const size_t lds = 4 * 2 * WaveTile<false, 8>::ELEMS * 8;
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  parts_kernel<LOADS, STORES, READS><<<grid, 256, lds>>>(A, B, C, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < 5; ++r) parts_kernel<LOADS, STORES, READS><<<grid, 256, lds>>>(A, B, C, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 5;
  long long clk[4];
  CK(hipMemcpyFromSymbol(clk, HIP_SYMBOL(g_clk), sizeof(clk)));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, parts_kernel<LOADS, STORES, READS>, 256, lds));
  printf("parts: global loads %d, LDS stores %d, fragment reads %d: %.3f ms %.1f TF   (%d workgroups per CU; probe workgroup: %lld shader clocks in %.1f us = %.0f MHz)\n", LOADS, STORES, READS, ms,
         2.0 * n * n * K / ms / 1e9, occ, clk[0], clk[1] / 100.0, clk[0] / (clk[1] / 100.0));
  return 0;
}

template <bool AKC, bool BKC, int NK>
int run(const char* name, long n, int K) {
  double *A, *B, *C, *C2;
  CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8)); CK(hipMalloc(&C2, n * n * 8));
  std::vector<double> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice));
  for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
  CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const size_t lds = 4 * (WaveTile<AKC, NK>::ELEMS + WaveTile<BKC, NK>::ELEMS) * 8;
  CK(hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm_wave_kernel<AKC, BKC, NK>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
  int occ = 0;
  CK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, gemm_wave_kernel<AKC, BKC, NK>, 256, lds));
  const int reps = 5;
  float ms_ref[2], ms_w;
  for (int pf = 1; pf <= 2; ++pf) {
    auto k = pf == 1 ? gemm_ref_kernel<double, AKC, BKC, 1> : gemm_ref_kernel<double, AKC, BKC, 2>;
    k<<<grid, 256>>>(A, B, C, n, K / 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) k<<<grid, 256>>>(A, B, C, n, K / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_ref[pf - 1], e0, e1)); ms_ref[pf - 1] /= reps;
  }
  gemm_wave_kernel<AKC, BKC, NK><<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  for (int r = 0; r < reps; ++r) gemm_wave_kernel<AKC, BKC, NK><<<grid, 256, lds>>>(A, B, C2, n, K / 16);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms_w, e0, e1)); ms_w /= reps;
  std::vector<double> c1(n * n), c2(n * n);
  CK(hipMemcpy(c1.data(), C, n * n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, n * n * 8, hipMemcpyDeviceToHost));
  double md = 0;
  for (long i = 0; i < n * n; ++i) md = fmax(md, fabs(c1[i] - c2[i]));
  const double fl = 2.0 * n * n * K / 1e9;
  printf("%-10s NK %d n %ld K %d: core PF1 %.3f ms %.1f TF | core PF2 %.3f ms %.1f TF | wave-private %.3f ms %.1f TF (LDS %zu B, occupancy %d/CU) | max |diff| %.3g\n", name, NK, n, K,
         ms_ref[0], fl / ms_ref[0], ms_ref[1], fl / ms_ref[1], ms_w, fl / ms_w, lds, occ, md);
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2));
  return 0;
}

// ---- the PRODUCT's direct-to-LDS core (dgp_gemm_dma.h) against TileGemm, every operand combination, both precisions
template <typename T, bool AKC, bool BKC, int SLOTS = 3>
__global__ __launch_bounds__(256, 3) void gemm_product_kernel(const T* A, const T* B, T* C, long n, int ktiles) {
  using D = DmaGemm<T, AKC, BKC, SLOTS>;
  using G = TileGemm<T, AKC, BKC, 128, 128>;
  __shared__ T smem[D::SMEM_ELEMS];
  const long bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[4][4];
  G::zero(acc);
  const T* a = AKC ? A + bi * 128 * n : A + bi * 128;
  const T* b = BKC ? B + bj * 128 * n : B + bj * 128;
  D::run(a, n, b, n, ktiles, smem, acc);
  T* out = C + bi * 128 * n + bj * 128;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * n + c] = v; });
}

template <typename T, bool AKC, bool BKC, int PF, int SLOTS = 3>
int run_product(const char* name, long n) {
  T *A, *B, *C, *C2;
  CK(hipMalloc(&A, n * n * sizeof(T))); CK(hipMalloc(&B, n * n * sizeof(T))); CK(hipMalloc(&C, n * n * sizeof(T))); CK(hipMalloc(&C2, n * n * sizeof(T)));
  std::vector<T> h(n * n);
  for (long i = 0; i < n * n; ++i) h[i] = (T)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(A, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  for (long i = 0; i < n * n; ++i) h[i] = (T)((double)rand() / RAND_MAX - 0.5);
  CK(hipMemcpy(B, h.data(), n * n * sizeof(T), hipMemcpyHostToDevice));
  dim3 grid(n / 128, n / 128);
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  std::vector<T> c1(n * n), c2(n * n);
  const int reps = getenv("REPS") ? atoi(getenv("REPS")) : 5;  // REPS=60: ~1 s of sustained load per measurement (power / clock steady state)
  for (int K : {512, 1024, 8192}) {
    float ms_ref, ms_d;
    gemm_ref_kernel<T, AKC, BKC, PF><<<grid, 256>>>(A, B, C, n, K / 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) gemm_ref_kernel<T, AKC, BKC, PF><<<grid, 256>>>(A, B, C, n, K / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_ref, e0, e1)); ms_ref /= reps;
    gemm_product_kernel<T, AKC, BKC, SLOTS><<<grid, 256>>>(A, B, C2, n, K / 16);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; ++r) gemm_product_kernel<T, AKC, BKC, SLOTS><<<grid, 256>>>(A, B, C2, n, K / 16);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipEventElapsedTime(&ms_d, e0, e1)); ms_d /= reps;
    CK(hipMemcpy(c1.data(), C, n * n * sizeof(T), hipMemcpyDeviceToHost)); CK(hipMemcpy(c2.data(), C2, n * n * sizeof(T), hipMemcpyDeviceToHost));
    double md = 0;
    for (long i = 0; i < n * n; ++i) md = fmax(md, fabs((double)c1[i] - (double)c2[i]));
    const double fl = 2.0 * n * n * K / 1e9;
    printf("product core %-10s n %ld K %5d: TileGemm (PF %d) %.3f ms %6.1f TF | DmaGemm (ring of %d) %.3f ms %6.1f TF | max |diff| %.3g\n", name, n, K, PF, ms_ref, fl / ms_ref,
           SLOTS, ms_d, fl / ms_d, md);
  }
  CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C)); CK(hipFree(C2));
  return 0;
}

// ---- lauum's launch pattern on the shipped core: lower-triangle tiles (bi >= bj) of S = T^T T with k from bi*128 to n, `sites` matrices
// (blockIdx.z), against the same number of flops as uniform-K tiles.  ORDER: 0 = row by row, long k-ranges first (the product's),
// 1 = the same tiles with every workgroup's k-range cut to the launch's mean (uniform cost, same total flops)
template <int ORDER>
__global__ __launch_bounds__(256, 3) void lauum_like_kernel(const double* T, double* S, long n, int nbk, long site_stride) {
  using D = DmaGemm<double, false, false>;
  using G = TileGemm<double, false, false, 128, 128>;
  __shared__ double smem[D::SMEM_ELEMS];
  T += blockIdx.z * site_stride;
  S += blockIdx.z * site_stride;
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  typename G::acc_t acc[4][4];
  G::zero(acc);
  const int kb = ORDER == 0 ? nbk - bi : (nbk + 2) / 3;  // mean of (nbk - bi) over the triangle ~ nbk / 3 + ...
  const int row = ORDER == 0 ? bi : 0;
  const double* base = T + (long)row * 128 * n;
  D::run(base + (long)bi * 128, n, base + (long)bj * 128, n, kb * 8, smem, acc);
  double* out = S + (long)bi * 128 * n + (long)bj * 128;
  G::foreach (acc, [&](int r, int c, double& v) { out[(long)r * n + c] = v; });
}
int run_lauum_like(long n, int sites) {
  double *T, *S;
  CK(hipMalloc(&T, sites * n * n * 8)); CK(hipMalloc(&S, sites * n * n * 8));
  CK(hipMemset(T, 0, sites * n * n * 8));
  const int nbk = (int)(n / 128), tiles = nbk * (nbk + 1) / 2;
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  double ktot = 0;  // k-blocks summed over the triangle
  for (int bi = 0; bi < nbk; ++bi) ktot += (double)(bi + 1) * (nbk - bi);
  for (int order = 0; order < 2; ++order) {
    auto k = order == 0 ? lauum_like_kernel<0> : lauum_like_kernel<1>;
    const double kb_total = order == 0 ? ktot : (double)tiles * ((nbk + 2) / 3);
    k<<<dim3(tiles, 1, sites), 256>>>(T, S, n, nbk, n * n);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < 3; ++r) k<<<dim3(tiles, 1, sites), 256>>>(T, S, n, nbk, n * n);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1)); ms /= 3;
    printf("lauum-like launch n %ld, %d sites, %d tiles per site, %s: %.3f ms  %.1f TF\n", n, sites, tiles,
           order == 0 ? "triangular k-ranges, long first" : "uniform k-range (the mean)", ms, sites * kb_total * 2.0 * 128 * 128 * 128 / ms / 1e9);
  }
  CK(hipFree(T)); CK(hipFree(S));
  return 0;
}

int main() {
  if (getenv("LAUUM_LIKE")) return run_lauum_like(8192, 32);
  if (!getenv("EXPERIMENTS")) {  // default: the shipped core; EXPERIMENTS=1 runs the variants that led to it
    run_product<double, false, false, 2>("f64 IC/IC", 8192);
    if (getenv("ONLY_FIRST")) return 0;
    run_product<double, true, false, 1>("f64 KC/IC", 8192);
    run_product<double, true, true, 1>("f64 KC/KC", 8192);
    run_product<double, true, true, 1, 2>("f64 KC/KC", 8192);  // the ring of two chunks (32 KB)
    run_product<double, false, true, 1>("f64 IC/KC", 8192);
    run_product<float, false, false, 4>("f32 IC/IC", 8192);
    run_product<float, true, false, 4>("f32 KC/IC", 8192);
    run_product<float, true, true, 2>("f32 KC/KC", 8192);
    run_product<float, false, true, 2>("f32 IC/KC", 8192);
    return 0;
  }
  {
    const long n = 8192;
    double *A, *B, *C;
    CK(hipMalloc(&A, n * n * 8)); CK(hipMalloc(&B, n * n * 8)); CK(hipMalloc(&C, n * n * 8));
    for (int pass = 0; pass < 2; ++pass) {
    if (pass == 0) { CK(hipMemset(A, 0, n * n * 8)); CK(hipMemset(B, 0, n * n * 8)); printf("parts: operands all zero\n"); }
    else {
      std::vector<double> h(n * n);
      for (long i = 0; i < n * n; ++i) h[i] = (double)rand() / RAND_MAX - 0.5;
      CK(hipMemcpy(A, h.data(), n * n * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(B, h.data(), n * n * 8, hipMemcpyHostToDevice));
      printf("parts: operands uniform random\n");
    }
    parts<false, false, false>(A, B, C, n, 8192);
    parts<false, false, true>(A, B, C, n, 8192);
    parts<false, true, true>(A, B, C, n, 8192);
    parts<true, false, true>(A, B, C, n, 8192);
    parts<true, true, false>(A, B, C, n, 8192);
    parts<true, true, true>(A, B, C, n, 8192);
    }
    CK(hipFree(A)); CK(hipFree(B)); CK(hipFree(C));
  }
  for (int K : {512, 8192}) {
    run_dma<4>(8192, K);   // wave-private ring of 4 stages, two workgroups per CU
    run_dma<0>(8192, K);   // wave-private ring of 3 stages, three workgroups per CU
    run_dma<14>(8192, K);  // workgroup-shared ring of 4 stages, three per CU
    run_dma<16>(8192, K);  // ... of 6 stages (spills)
  }
  if (getenv("ALL")) for (int K : {512, 8192}) {
    run_pipe<false, false>("f64 IC/IC", 8192, K);
    run_pipe<true, true>("f64 KC/KC", 8192, K);
    run_pipe<true, false>("f64 KC/IC", 8192, K);
  }
  if (getenv("ALL")) for (int K : {512, 8192}) {
    run<false, false, 16>("f64 IC/IC", 8192, K);
    run<false, false, 8>("f64 IC/IC", 8192, K);
    run<true, true, 16>("f64 KC/KC", 8192, K);
    run<true, true, 8>("f64 KC/KC", 8192, K);
  }
  return 0;
}
