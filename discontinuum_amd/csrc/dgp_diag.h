// dgp_diag.h -- the 128x128 diagonal-block kernel of the blocked Cholesky: L_kk and L_kk^-1.
//
// This kernel sits on the sequential critical path of the factorisation (one launch per 128 columns),
// so it is built for latency, not throughput.  The block lives in LDS as 16x16 sub-blocks
// (lower block triangle of L and of L^-1, row stride 17 -> conflict-free MFMA fragment reads):
//   for each 16-column step kb:
//     GJ16    one wave factors the 16x16 pivot block by an in-register Gauss-Jordan sweep: 16 pivots,
//             values broadcast with lane shuffles (ds_bpermute / v_readlane), no barrier, no LDS
//             round trip -> ~200 cycles per pivot instead of ~1400 for a workgroup-wide sweep.
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

template <typename T>
__global__ __launch_bounds__(256) void potrf_diag_fast_kernel(T* __restrict__ A, long ld, long k0,
                                                              T* __restrict__ Tinv, T* __restrict__ logdet,
                                                              int* __restrict__ info) {
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
  T* Ablk = A + k0 * ld + k0;
  T* Xblk = Tinv + k0 * ld + k0;
  // this workgroup usually shares its CU with waves of the bulk trailing update: win the issue arbitration
  __builtin_amdgcn_s_setprio(3);

  // ---- load the lower block triangle (diagonal sub-blocks complete: they are symmetric)
  // (fully unrolled: all 36 global loads are in flight before the first LDS store)
  {
    T tmp[DGP_DTRI];
#pragma unroll
    for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) tmp[dtri(bi, bj)] = Ablk[(long)(16 * bi + ti) * ld + 16 * bj + tj];
#pragma unroll
    for (int b = 0; b < DGP_DTRI; ++b) sL[b * DGP_DBLK + ti * DGP_DS + tj] = tmp[b];
  }
  __syncthreads();

  // ---- blocked right-looking Cholesky on the sub-blocks
  for (int kb = 0; kb < DGP_DNB; ++kb) {
    if (wave == 0) gj16<T>(sL + dtri(kb, kb) * DGP_DBLK, sXd + kb * DGP_DBLK, dvals + 16 * kb, lane);
    __syncthreads();
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
    // trailing update: A_ij -= L_ik L_jk^T for kb < j <= i
    const int m = DGP_DNB - 1 - kb;
    for (int idx = wave; idx < m * (m + 1) / 2; idx += 4) {
      int bi, bj;
      tri_decode(idx, bi, bj);
      bi += kb + 1;
      bj += kb + 1;
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
    }
    __syncthreads();
  }

  // ---- L is final: stream it to global (zeros above the block diagonal).  The inverse below reads
  // its L operands back from L2, which frees the LDS copy to receive the off-diagonal blocks of L^-1.
#pragma unroll
  for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
    for (int bj = 0; bj < DGP_DNB; ++bj)
      Ablk[(long)(16 * bi + ti) * ld + 16 * bj + tj] = bj <= bi ? sL[dtri(bi, bj) * DGP_DBLK + ti * DGP_DS + tj] : T(0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // the stores have left this CU ...
  __syncthreads();                                   // ... for every wave, before anyone reads them back

  // ---- L^-1 by block forward substitution: wave w owns block columns w and 7 - w
  const int fr = lane & 15, fq = lane >> 4;
  for (int half = 0; half < 2; ++half) {
    const int j = half == 0 ? wave : DGP_DNB - 1 - wave;
    for (int i = j + 1; i < DGP_DNB; ++i) {
      // all L fragments of block row i, columns j..i-1, in flight at once (L1-bypassing loads: this
      // CU's L1 may still hold the pre-factorisation lines of the block)
      T lf[DGP_DNB - 1][4];
#pragma unroll
      for (int cc = 0; cc < DGP_DNB - 1; ++cc)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
          const int c = j + cc;
          lf[cc][ks] = c < i ? __builtin_nontemporal_load(&Ablk[(long)(16 * i + fr) * ld + 16 * c + 4 * ks + fq]) : T(0);
        }
      acc_t acc = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int cc = 0; cc < DGP_DNB - 1; ++cc) {
        const int c = j + cc;
        if (c < i) {
          const T* Xcj = (c == j) ? sXd + j * DGP_DBLK : sL + dtri(c, j) * DGP_DBLK;
#pragma unroll
          for (int ks = 0; ks < 4; ++ks) acc = Mfma<T>::mma(lf[cc][ks], frag_cr(Xcj, ks, lane), acc);
        }
      }
      // X_ij = -Linv_ii * W : park W in the destination sub-block, read it back as the B operand
      T* Xij = sL + dtri(i, j) * DGP_DBLK;
      const T* Xii = sXd + i * DGP_DBLK;
#pragma unroll
      for (int r = 0; r < 4; ++r) Xij[Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)] = acc[r];
      acc_t out = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) out = Mfma<T>::mma(-frag_rc(Xii, ks, lane), frag_cr(Xij, ks, lane), out);
#pragma unroll
      for (int r = 0; r < 4; ++r) Xij[Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)] = out[r];
    }
  }
  __syncthreads();

  // ---- L^-1 into the diagonal block of Tinv (zeros above the block diagonal)
#pragma unroll
  for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
    for (int bj = 0; bj < DGP_DNB; ++bj) {
      const int o = ti * DGP_DS + tj;
      const T v = bj < bi ? sL[dtri(bi, bj) * DGP_DBLK + o] : (bj == bi ? sXd[bi * DGP_DBLK + o] : T(0));
      Xblk[(long)(16 * bi + ti) * ld + 16 * bj + tj] = v;
    }
  // ---- log-determinant and first bad pivot (fixed-order tree: reproducible)
  __shared__ T red[128];
  __shared__ int bad;
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
    logdet[0] += red[0];  // diag kernels of one factorisation run in stream order
    if (bad < 128) atomicCAS(info, 0, (int)(k0 + bad + 1));
  }
}

template <typename T>
inline size_t potrf_diag_fast_smem() {
  return (size_t)((DGP_DTRI + DGP_DNB) * DGP_DBLK + 128) * sizeof(T);
}

}  // namespace dgp
