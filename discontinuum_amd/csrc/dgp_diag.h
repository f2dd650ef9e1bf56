// dgp_diag.h -- the 128x128 diagonal-block kernel of the blocked Cholesky: L_kk and L_kk^-1.
//
// This kernel sits on the sequential critical path of the factorisation (one launch per 128 columns),
// so it is built for latency, not throughput.  The block lives in LDS as 16x16 sub-blocks
// (lower block triangle of L and of L^-1, row stride 17 -> conflict-free MFMA fragment reads):
//   for each 16-column step kb:
//     GJ16    one wave factors the 16x16 pivot block by an in-register Gauss-Jordan sweep: 16 pivots,
//             values broadcast with lane shuffles (ds_bpermute / v_readlane), no barrier, no LDS
//             round trip -> ~350 cycles per pivot (measured) instead of ~1400 for a workgroup-wide sweep.
//             (A sweep that applies each pivot as a rank-1 MFMA on accumulator-layout cells needs no shuffles
//             but measured SLOWER, ~400-470 cycles per pivot: one wave's dependent fp64 chain, ~10 cycles per
//             instruction, bounds both.)
//     panel   L_ik = A_ik L_kk^-T              (MFMA 16x16x4, one sub-block per wave)
//     update  A_ij -= L_ik L_jk^T, i >= j > kb (MFMA)
//   then L^-1 by block forward substitution, one block column per wave (columns w and 7-w balance
//   to 7 sub-steps each), all MFMA, no inter-wave synchronisation.
// Flops ~ 2 * 128^3 / 3 (factor + inverse); the kernel is bounded by the 128 dependent pivots.
#pragma once
#include "dgp_common.h"

namespace dgp {

#define DGP_DB 16                      // sub-block size
#define DGP_DS 17                      // sub-block row stride in LDS (odd: conflict-free fragment reads)
#define DGP_DBLK (DGP_DB * DGP_DS)     // elements per sub-block
#define DGP_DNB 8                      // sub-blocks per dimension (128 / 16)
#define DGP_DTRI (DGP_DNB * (DGP_DNB + 1) / 2)

__device__ __forceinline__ int dtri(int bi, int bj) { return bi * (bi + 1) / 2 + bj; }

__device__ __forceinline__ double dg_rcp(double d) {
  double y = __builtin_amdgcn_rcp(d);
  double e = fma(-d, y, 1.0);
  y = fma(y, e, y);
  e = fma(-d, y, 1.0);
  return fma(y, e, y);
}
__device__ __forceinline__ float dg_rcp(float d) {
  float y = __builtin_amdgcn_rcpf(d);
  float e = fmaf(-d, y, 1.0f);
  return fmaf(y, e, y);
}
__device__ __forceinline__ double dg_rsqrt(double d) {
  double y = __builtin_amdgcn_rsq(d);
  const double h = 0.5 * d;
  double e = fma(-h * y, y, 0.5);
  y = fma(y, e, y);
  e = fma(-h * y, y, 0.5);
  return fma(y, e, y);
}
__device__ __forceinline__ float dg_rsqrt(float d) {
  float y = __builtin_amdgcn_rsqf(d);
  const float h = 0.5f * d;
  float e = fmaf(-h * y, y, 0.5f);
  return fmaf(y, e, y);
}
__device__ __forceinline__ double dg_readlane(double x, int lane) {
  int lo = __double2loint(x), hi = __double2hiint(x);
  lo = __builtin_amdgcn_readlane(lo, lane);
  hi = __builtin_amdgcn_readlane(hi, lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ float dg_readlane(float x, int lane) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(x), lane));
}

// In-register Gauss-Jordan sweep of one symmetric positive definite 16x16 block by ONE wave.
// Lane l owns column j = l & 15 and rows q + 4r (q = l >> 4, r = 0..3).  The whole (symmetric) Schur
// complement is kept up to date, so row k of the cells is also column k of the matrix and a single
// shuffle delivers the pivot row to every lane.  On exit `blkL` holds L (zeros above the diagonal),
// `blkX` holds L^-1 and dv[0..15] the pivots.
template <typename T>
__device__ __forceinline__ void gj16(T* __restrict__ blkL, T* __restrict__ blkX, T* __restrict__ dv, int lane) {
  const int j = lane & 15, q = lane >> 4;
  T c[4], lraw[4];
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    c[r] = blkL[(q + 4 * r) * DGP_DS + j];
    lraw[r] = T(0);
  }
  T dj = T(1);
#pragma unroll
  for (int k = 0; k < 16; ++k) {
    const int rk = k >> 2, qk = k & 3;
    const T d = dg_readlane(c[rk], (qk << 4) | k);       // pivot (uniform)
    const T inv_d = dg_rcp(d);
    const T uj = __shfl(c[rk], (qk << 4) | j, 64);       // cell (k, j): pivot row
    T ui[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) ui[r] = __shfl(c[r], (q << 4) | k, 64);  // cell (q + 4r, k): pivot column
    const bool pivcol = (j == k);
    const T vj = pivcol ? T(1) : uj;
    dj = pivcol ? d : dj;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int i = q + 4 * r;
      const bool act = i > k;
      lraw[r] = (pivcol && i >= k) ? ui[r] : lraw[r];     // column k of L, unscaled (ui of row k is d)
      const T li = act ? ui[r] * inv_d : T(0);
      const T cur = pivcol ? T(0) : c[r];                 // cell (i, k) restarts as the inverse entry
      c[r] = act ? fma(-li, vj, cur) : c[r];
    }
    if (lane == 0) dv[k] = d;
  }
  // scale: L[i][j] = lraw * s_j ; X[i][j] = c * s_i (j < i), s_i (j == i)
  const T sj = dg_rsqrt(dj);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const int i = q + 4 * r;
    const T si = __shfl(sj, i, 64);  // lane i (q = 0, column i) holds s_i
    const T lval = i >= j ? lraw[r] * sj : T(0);
    const T xval = i > j ? c[r] * si : (i == j ? si : T(0));
    blkL[i * DGP_DS + j] = lval;
    blkX[i * DGP_DS + j] = xval;
  }
}

// MFMA fragment helpers on 16x17 LDS sub-blocks
template <typename T>
__device__ __forceinline__ T frag_rc(const T* blk, int ks, int lane) {  // element (row lane&15, col 4ks + lane>>4)
  return blk[(lane & 15) * DGP_DS + 4 * ks + (lane >> 4)];
}
template <typename T>
__device__ __forceinline__ T frag_cr(const T* blk, int ks, int lane) {  // element (row 4ks + lane>>4, col lane&15)
  return blk[(4 * ks + (lane >> 4)) * DGP_DS + (lane & 15)];
}

// An accumulator block (C layout) can feed the next MFMA as its B operand without moving data:
//   fp64: C row = (lane>>4) + 4r  == B row of k-step r            -> natural k order
//   fp32: C row = 4(lane>>4) + r  == B row 4q + r of lane group q -> the A operand must then present
//         column 4q + r in "k-step" r (a consistent permutation of the summation index)
template <typename T>
__device__ __forceinline__ T frag_rc_acc(const T* blk, int r, int lane);
template <>
__device__ __forceinline__ double frag_rc_acc<double>(const double* blk, int r, int lane) {
  return blk[(lane & 15) * DGP_DS + 4 * r + (lane >> 4)];
}
template <>
__device__ __forceinline__ float frag_rc_acc<float>(const float* blk, int r, int lane) {
  return blk[(lane & 15) * DGP_DS + 4 * (lane >> 4) + r];
}

// optional phase timestamps (scripts/diag_bench.hip builds with -DDGP_DIAG_PROFILE)
#ifdef DGP_DIAG_PROFILE
__device__ long long dgp_diag_prof[16];
#define DGP_DIAG_STAMP(i) \
  if (threadIdx.x == 0) dgp_diag_prof[i] = clock64();
#else
#define DGP_DIAG_STAMP(i)
#endif
// optional in-situ log (scripts/diag_in_situ.py builds a second library with -DDGP_DIAG_LOG): per 128-column block the
// 100 MHz wall clock at the workgroup's first and last instruction and where it ran -- the profiler's kernel duration
// minus (end - begin) is the time the launch spent waiting to be placed
#ifdef DGP_DIAG_LOG
__device__ unsigned long long dgp_diag_log[4 * 1024];
#define DGP_DIAG_LOG_BEGIN(k)                                                                                       \
  if (threadIdx.x == 0 && blockIdx.z == 0 && (k) < 1024) {                                                           \
    dgp_diag_log[4 * (k)] = wall_clock64();                                                                          \
    dgp_diag_log[4 * (k) + 1] = ((unsigned long long)(__builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 15) << 16) | \
                                (unsigned long long)(__builtin_amdgcn_s_getreg((7 << 11) | (8 << 6) | 4) & 255);      \
  }
#define DGP_DIAG_LOG_END(k) \
  if (threadIdx.x == 0 && blockIdx.z == 0 && (k) < 1024) dgp_diag_log[4 * (k) + 2] = wall_clock64();
#else
#define DGP_DIAG_LOG_BEGIN(k)
#define DGP_DIAG_LOG_END(k)
#endif

// TS = the matrix's storage type, T = the type the block is factored and inverted in.  T = TS for fp64 plans; fp32
// plans use T = double (MIXED-PRECISION PANEL): the block is promoted when it is loaded into LDS, factored and inverted
// with the fp64 code path (fp64 MFMA, fp64 pivots), and L_kk / L_kk^-1 are rounded to fp32 ONCE when they are stored, so
// that trsm's operand carries one rounding instead of the ~cond(L_kk) eps32 of an fp32 Gauss-Jordan inverse.  The
// log-determinant is accumulated in T; `logdet_hi` (the fp32 plans' double slot, may be null) receives it unrounded.
template <typename TS, typename T = TS>
__global__ __launch_bounds__(256) void potrf_diag_fast_kernel(TS* __restrict__ A, long ld, long k0,
                                                              TS* __restrict__ Tinv, TS* __restrict__ logdet,
                                                              int* __restrict__ info, long bs, long ibs, int init,
                                                              int ninit, double* __restrict__ logdet_hi, int done_index,
                                                              int done_value, int yield_index = -1) {
  A = site(A, bs);
  Tinv = site(Tinv, bs);
  logdet = site(logdet, bs);
  if (logdet_hi) logdet_hi = (double*)((char*)logdet_hi + (long)blockIdx.z * bs * (long)sizeof(TS));
  info = site(info, ibs);
  // the first diagonal block of a factorisation also resets its status words: info[0] (first bad pivot), the finish
  // kernel's ticket and the early-launch queue counters -- nothing else runs on this matrix yet
  if (init)
    for (int i = threadIdx.x; i < ninit; i += 256) info[i] = 0;
  extern __shared__ __attribute__((aligned(16))) unsigned char dg_smem[];
  using acc_t = typename Mfma<T>::acc_t;
  // LDS budget (fp64): 36 + 8 sub-blocks of 16x17 + 128 pivots = 96.7 KB.  It has to stay well below
  // 160 KB - 35 KB so that this workgroup can be placed on a CU that still hosts one workgroup of the
  // concurrent bulk trailing update; needing a whole CU would starve it until the bulk kernel drains.
  T* sL = reinterpret_cast<T*>(dg_smem);         // DGP_DTRI sub-blocks: A, then L, then off-diagonal L^-1
  T* sXd = sL + DGP_DTRI * DGP_DBLK;             // DGP_DNB diagonal sub-blocks of L^-1
  T* dvals = sXd + DGP_DNB * DGP_DBLK;           // 128 pivots
  const int t = threadIdx.x, lane = t & 63;
  const int wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int ti = t >> 4, tj = t & 15;
  TS* Ablk = A + k0 * ld + k0;
  TS* Xblk = Tinv + k0 * ld + k0;
  // this workgroup usually shares its CU with waves of the bulk trailing update: win the issue arbitration
  __builtin_amdgcn_s_setprio(3);
  DGP_DIAG_STAMP(0)
  DGP_DIAG_LOG_BEGIN((int)(k0 / 128))

  // ---- load the lower block triangle (diagonal sub-blocks complete: they are symmetric)
  // (fully unrolled: all 36 global loads are in flight before the first LDS store)
  {
    T tmp[DGP_DTRI];
#pragma unroll
    for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) tmp[dtri(bi, bj)] = (T)Ablk[(long)(16 * bi + ti) * ld + 16 * bj + tj];
#pragma unroll
    for (int b = 0; b < DGP_DTRI; ++b) sL[b * DGP_DBLK + ti * DGP_DS + tj] = tmp[b];
  }
  __syncthreads();
  // from here to the end this workgroup is a chain of dependent latencies: ask the bulk update's waves on THIS CU to sleep
  // meanwhile (dgp_common.h: yield_if_asked; after the barrier, so that the init block's reset of the word is behind us)
  if (yield_index >= 0 && t == 0)
    __hip_atomic_store(reinterpret_cast<unsigned*>(&info[yield_index]), cu_code(), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);

  // ---- blocked right-looking Cholesky on the sub-blocks, software-pipelined: while wave 0 runs the
  // sequential 16-pivot sweep of the NEXT diagonal sub-block (which only needs that one sub-block
  // updated), waves 1-3 apply the rest of the current trailing update.
  auto update_block = [&](int bi, int bj, int kb) {  // A_ij -= L_ik L_jk^T
    T* C = sL + dtri(bi, bj) * DGP_DBLK;
    const T* Li = sL + dtri(bi, kb) * DGP_DBLK;
    const T* Lj = sL + dtri(bj, kb) * DGP_DBLK;
    acc_t acc;
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = C[Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) acc = Mfma<T>::mma(-frag_rc(Li, ks, lane), frag_rc(Lj, ks, lane), acc);
#pragma unroll
    for (int r = 0; r < 4; ++r) C[Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)] = acc[r];
  };
  DGP_DIAG_STAMP(1)
  if (wave == 0) gj16<T>(sL + dtri(0, 0) * DGP_DBLK, sXd, dvals, lane);
  __syncthreads();
  DGP_DIAG_STAMP(2)
  for (int kb = 0; kb < DGP_DNB - 1; ++kb) {
    // panel: L_ik = A_ik * Linv_kk^T
    const T* Xkk = sXd + kb * DGP_DBLK;
    for (int i = kb + 1 + wave; i < DGP_DNB; i += 4) {
      T* blk = sL + dtri(i, kb) * DGP_DBLK;
      acc_t acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) acc = Mfma<T>::mma(frag_rc(blk, ks, lane), frag_rc(Xkk, ks, lane), acc);
#pragma unroll
      for (int r = 0; r < 4; ++r) blk[Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)] = acc[r];
    }
    __syncthreads();
    // trailing update A_ij -= L_ik L_jk^T (kb < j <= i), next pivot block first
    const int m = DGP_DNB - 1 - kb, nblk = m * (m + 1) / 2;
    if (wave == 0) {
      update_block(kb + 1, kb + 1, kb);
      gj16<T>(sL + dtri(kb + 1, kb + 1) * DGP_DBLK, sXd + (kb + 1) * DGP_DBLK, dvals + 16 * (kb + 1), lane);
    } else {
      for (int idx = wave; idx < nblk; idx += 3) {  // idx 0 is the pivot block handled by wave 0
        int bi, bj;
        tri_decode(idx, bi, bj);
        update_block(bi + kb + 1, bj + kb + 1, kb);
      }
    }
    __syncthreads();
  }

  DGP_DIAG_STAMP(3)
  // ---- L is final: stream it to global (zeros above the block diagonal); nothing waits for these stores
#pragma unroll
  for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
    for (int bj = 0; bj < DGP_DNB; ++bj)
      Ablk[(long)(16 * bi + ti) * ld + 16 * bj + tj] = bj <= bi ? (TS)sL[dtri(bi, bj) * DGP_DBLK + ti * DGP_DS + tj] : TS(0);

  DGP_DIAG_STAMP(4)
  // ---- L^-1 by block forward substitution: wave w owns block columns w and 7 - w.  The finished blocks
  // X_cj of the column stay in REGISTERS (accumulator layout == B-operand layout, see frag_rc_acc), so
  // this phase touches LDS only for its A operands (L_ic, Linv_ii) and global memory only for stores.
  const int crow0 = Mfma<T>::crow(lane, 0);
  for (int half = 0; half < 2; ++half) {
    const int j = half == 0 ? wave : DGP_DNB - 1 - wave;
    const T* Xjj = sXd + j * DGP_DBLK;
    acc_t xb[DGP_DNB - 1];  // xb[d - 1] = X_{j+d, j}
#pragma unroll
    for (int di = 1; di < DGP_DNB; ++di) {
      const int i = j + di;
      if (i < DGP_DNB) {
        acc_t acc = {T(0), T(0), T(0), T(0)};
        {  // c = j: X_jj comes from LDS in natural k order
          const T* Lij = sL + dtri(i, j) * DGP_DBLK;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = Mfma<T>::mma(frag_rc(Lij, ks, lane), frag_cr(Xjj, ks, lane), acc);
        }
#pragma unroll
        for (int dc = 1; dc < di; ++dc) {  // c = j + dc: X_cj is a register block
          const T* Lic = sL + dtri(i, j + dc) * DGP_DBLK;
#pragma unroll
          for (int r = 0; r < 4; ++r) acc = Mfma<T>::mma(frag_rc_acc<T>(Lic, r, lane), xb[dc - 1][r], acc);
        }
        const T* Xii = sXd + i * DGP_DBLK;
        acc_t out = {T(0), T(0), T(0), T(0)};
#pragma unroll
        for (int r = 0; r < 4; ++r) out = Mfma<T>::mma(-frag_rc_acc<T>(Xii, r, lane), acc[r], out);
        xb[di - 1] = out;
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Xblk[(long)(16 * i + Mfma<T>::crow(lane, r)) * ld + 16 * j + (lane & 15)] = (TS)out[r];
      }
    }
  }
  (void)crow0;
  DGP_DIAG_STAMP(5)

  // ---- diagonal sub-blocks of L^-1 and the zeros above the block diagonal
#pragma unroll
  for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
    for (int bj = bi; bj < DGP_DNB; ++bj)
      Xblk[(long)(16 * bi + ti) * ld + 16 * bj + tj] = bj == bi ? (TS)sXd[bi * DGP_DBLK + ti * DGP_DS + tj] : TS(0);
  DGP_DIAG_STAMP(6)
  // ---- log-determinant and first bad pivot (fixed-order tree: reproducible).  The scratch lives in the (now dead) sub-block
  // area: with NO static LDS the kernel's footprint is 94.5 KB = 76 allocation units of 1280 B, which is exactly what a CU
  // has free once ONE of three 32 KB-ring bulk workgroups retires (dgp_chol.hip: syrk_kernel RING = 2); with the 1040 bytes
  // these two used to take statically it was one unit more
  __syncthreads();  // every wave is through with sL / sXd
  T* red = sL;
  int& bad = *reinterpret_cast<int*>(sL + 128);
  if (t == 0) bad = 128;
  __syncthreads();
  if (t < 128) {
    const T d = dvals[t];
    red[t] = log(d);
    if (!(d > T(0))) atomicMin(&bad, t);
  }
  __syncthreads();
  for (int sft = 64; sft > 0; sft >>= 1) {
    if (t < sft) red[t] += red[t + sft];
    __syncthreads();
  }
  if (t == 0) {
    logdet[0] = init ? (TS)red[0] : (TS)((T)logdet[0] + red[0]);  // diag kernels of one factorisation run in stream order
    if (logdet_hi) logdet_hi[0] = init ? (double)red[0] : logdet_hi[0] + (double)red[0];
    if (bad < 128) atomicCAS(info, 0, (int)(k0 + bad + 1));
  }
  DGP_DIAG_STAMP(7)
  DGP_DIAG_LOG_END((int)(k0 / 128))
  if (yield_index >= 0 && t == 0)
    __hip_atomic_store(reinterpret_cast<unsigned*>(&info[yield_index]), 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  // split panel chain: publish "this block is factored" for the rest stream's trsm, which was launched before this
  // kernel finished and polls the word (no event record / stream wait on the critical stream).  Every thread's stores of
  // L and L^-1 precede the barrier; the release makes them visible to the agent.
  if (done_index >= 0) {
    __syncthreads();
    if (t == 0) {
      __threadfence();
      __hip_atomic_store(&info[done_index], done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

template <typename T>
inline size_t potrf_diag_fast_smem() {
  return (size_t)((DGP_DTRI + DGP_DNB) * DGP_DBLK + 128) * sizeof(T);
}

}  // namespace dgp
