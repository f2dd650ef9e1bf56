// dgp_common.h -- shared device/host helpers for the MI355X (gfx950) exact-GP engine.
//
// Conventions used by every kernel in this directory:
//   * matrices are row-major N x N with leading dimension ld == N, N = round_up(n, 128);
//     only the lower triangle (i >= j) is meaningful.  The pad region (i or j >= n) holds the
//     identity, so chol(blockdiag(K, I)) = blockdiag(L, I): no kernel needs a bounds check and
//     log|K|, K^-1 r are unaffected.
//   * a wavefront is 64 lanes; MFMA is v_mfma_{f64,f32}_16x16x4 (one A and one B element per lane).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGP_TILE 128  // block size of every blocked algorithm == padding quantum of N
#define DGP_MAX_BATCH 8  // sites whose hyperparameters travel BY VALUE in the kernel arguments (larger batches, up to 1024, read a device array)

typedef double dgp_d4 __attribute__((ext_vector_type(4)));
typedef float dgp_f4 __attribute__((ext_vector_type(4)));
typedef double dgp_d2 __attribute__((ext_vector_type(2)));

namespace dgp {

template <typename T>
struct Mfma;

// v_mfma_f64_16x16x4_f64: lane l holds A[i=l&15][k=l>>4], B[k=l>>4][j=l&15];
// C/D: col = l&15, row = (l>>4) + 4*reg   (cdna_hip_programming.md section 3: f64 does NOT use
// the f32 row map).
template <>
struct Mfma<double> {
  using acc_t = dgp_d4;
  static __device__ __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// v_mfma_f32_16x16x4_f32: same A/B maps; C/D: col = l&15, row = (l>>4)*4 + reg.
template <>
struct Mfma<float> {
  using acc_t = dgp_f4;
  static __device__ __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  static __device__ __forceinline__ int crow(int lane, int r) { return (lane >> 4) * 4 + r; }
};

// 16-byte global load/store of 16/sizeof(T) consecutive elements
template <typename T>
struct Vec16;
template <>
struct Vec16<double> {
  typedef dgp_d2 type;
  static constexpr int N = 2;
};
template <>
struct Vec16<float> {
  typedef dgp_f4 type;
  static constexpr int N = 4;
};

// A batched plan keeps B sites' workspaces at a fixed stride: blockIdx.z selects the site.
template <typename T>
__device__ __forceinline__ T* site(T* p, long stride) {
  return p + (long)blockIdx.z * stride;
}
// a ragged batch: site b uses the first ns[b] <= n rows of its slots, the rest is identity padding like rows >= n
__device__ __forceinline__ int site_n(const int* ns, int n) { return ns ? ns[blockIdx.z] : n; }

template <typename T>
__device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
  return v;
}

// map a linear index onto the lower triangle (bi >= bj) of an nb x nb block grid, row by row.
__device__ __forceinline__ void tri_decode(int idx, int& bi, int& bj) {
  int i = (int)((sqrtf(8.0f * (float)idx + 1.0f) - 1.0f) * 0.5f);
  while ((i + 1) * (i + 2) / 2 <= idx) ++i;
  while (i * (i + 1) / 2 > idx) --i;
  bi = i;
  bj = idx - i * (i + 1) / 2;
}

// SUPERTILE order of the lower triangle of an nt x nt tile grid: bands of S tile rows, inside a band S x S supertiles
// left to right (row-major inside each), the band's diagonal supertile (a triangle) last.  A bijection of [0, nt(nt+1)/2)
// for every nt.  Consecutive indices share S row panels and S column panels instead of one row panel and S^2 column
// panels: S times less operand traffic beyond L2 when they run on one XCD (xcd_remap).  tri_decode is the case S = 1.
__device__ __forceinline__ void super_decode(int t, int nt, int S, int& bi, int& bj) {
  int r0, c0;
  tri_decode(t, r0, c0);            // r0 = the tile row of t in plain row-major order: tiles above band R = tri(R S) <= t
  const int R = r0 / S, base = R * S;
  const int h = min(S, nt - base);  // rows of this band (the last one may be short)
  const int u = t - base * (base + 1) / 2;
  if (u < R * h * S) {
    const int C = u / (h * S), w = u - C * h * S;
    bi = base + w / S;
    bj = C * S + w % S;
  } else {
    int li, lj;
    tri_decode(u - R * h * S, li, lj);
    bi = base + li;
    bj = base + lj;
  }
}

// XCD-aware block remap (cdna_hip_programming.md T1, bijective form): hardware deals consecutive
// workgroup ids round-robin over the 8 XCDs, each with a private L2.  After the remap the ids that land
// on one XCD are CONSECUTIVE in the logical tile order, so neighbouring tiles (which share operand
// panels) hit the same L2.  A speed hint only: nothing depends on the actual placement.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
  const int q = nwg >> 3, r = nwg & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

}  // namespace dgp
