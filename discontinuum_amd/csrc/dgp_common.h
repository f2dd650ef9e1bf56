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

// ---- cooperative yield to the panel chain (single-site plans) ------------------------------------------------------------
// The diagonal-block kernel of the factorisation is ONE workgroup of dependent fp64 latency chains on the critical path.  Sharing
// its CU with two or three workgroups of a bulk update makes it 4-8 times slower (in-kernel stamps: 24 -> 120-206 us at
// n = 16384 fp32, 38 -> 90-100 us at n = 8192 fp64; the wait for a slot is 2-15 us of that: scripts/diag_in_situ.py) -- the
// MFMA pipe, the LDS and the issue ports are occupied by waves that do not care about 30 us.  So the block kernel publishes
// the CU it runs on (cu_code) in a word for its lifetime, and every wave of a bulk tile looks at the word every few chunks of
// its k-loop: on that CU it sleeps until the word changes (bounded: 96 x ~2 us).  Two or three workgroups of ~768 pause for
// ~30 us; the chain step that every bulk launch behind it waits for gets the CU to itself.  A hint only: results do not depend
// on it, a stale or foreign word costs at most the bound.
__device__ __forceinline__ unsigned cu_code() {  // (XCC, SE, SH, CU) of this wave + 1: never 0
  const unsigned xcc = __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15u;   // HW_REG_XCC_ID[3:0]
  const unsigned cu = __builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4) & 255u;    // HW_REG_HW_ID[15:8]: CU_ID, SH_ID, SE_ID
  return ((xcc << 8) | cu) + 1u;
}
// Per round of the k-loop a wave (i) looks at ITS copy of the word in LDS and (ii) refreshes the copy with one more
// direct-to-LDS load (global_load_lds_dword, all lanes the same address -> 256 bytes of LDS per wave): the load joins the
// operand loads' in-order vmcnt queue, so the copy is complete a round later without any wait of its own -- a blocking scalar
// load per check measured 2.3x on the bulk update (the memory system is saturated: several us per load).  Only a wave whose
// copy matches enters the sleep loop, which re-reads the real word (scalar, glc: past the scalar cache) until it changes.
// All control flow is inside the assembly block: the caller's register allocation never sees a branch (dgp_gemm_dma.h: joins
// around 128 live accumulators spill).
__device__ __forceinline__ void yield_refresh(const unsigned* word, unsigned lds_byte_addr) {
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %0, %1 sc1" ::"v"(0u), "s"(word), "s"(lds_byte_addr) : "memory", "m0");
}
__device__ __forceinline__ void yield_if_asked(const unsigned* word, unsigned me, unsigned seen) {
  unsigned t, cnt;
  asm volatile(
      "s_cmp_lg_u32 %4, %3\n\t"
      "s_cbranch_scc1 Lyield_done%=\n\t"
      "s_movk_i32 %1, 96\n"
      "Lyield_loop%=:\n\t"
      "s_load_dword %0, %2, 0x0 glc\n\t"
      "s_waitcnt lgkmcnt(0)\n\t"
      "s_cmp_lg_u32 %0, %3\n\t"
      "s_cbranch_scc1 Lyield_done%=\n\t"
      "s_sleep 32\n\t"
      "s_sub_u32 %1, %1, 1\n\t"
      "s_cmp_lg_u32 %1, 0\n\t"
      "s_cbranch_scc1 Lyield_loop%=\n"
      "Lyield_done%=:"
      : "=&s"(t), "=&s"(cnt)
      : "s"(word), "s"(me), "s"(seen)
      : "memory", "scc");
}

}  // namespace dgp
