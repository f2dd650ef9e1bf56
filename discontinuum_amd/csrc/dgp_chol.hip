// dgp_chol.hip -- blocked right-looking Cholesky of K^ = K + Sigma, the explicit inverse needed by the
// gradient trace terms, and the triangular solves of the marginal likelihood.  Hand-written for
// gfx950; this replaces what the reference gets from gpytorch/linear_operator/LAPACK underneath
// ExactMarginalLogLikelihood (src/discontinuum/engines/gpytorch.py:318, 353, 384).
//
//   potrf   for each 128-wide block column k:
//             potrf_diag   one workgroup: L_kk and L_kk^-1 (dgp_diag.h: LDS-blocked, wave-level pivots, MFMA)
//             trsm         A[i,k] <- A[i,k] L_kk^-T  as an MFMA GEMM against L_kk^-1
//             syrk         A[i,j] -= A[i,k] A[j,k]^T (MFMA), lower tiles only
//           with a one-panel lookahead: the bulk update runs on a second HIP stream, two panels at a time
//           (K = 256), while the panel chain of the next columns proceeds on the caller's stream.
//   trtri   T = L^-1 by log2(N/128) levels of batched MFMA GEMMs  T21 = -T22 (L21 T11)
//   lauum   S = T^T T = K^^-1, one launch, triangular k-range per tile
//   solve   z = T r, quad = z^T z, alpha = T^T z (bandwidth-bound, deterministic two-stage sums)
// Flops per fit: N^3/3 (potrf) + N^3/3 (trtri) + N^3/3 (lauum) -- MFMA roofline.
#include "dgp_diag.h"
#include "dgp_gemm.h"
#include "dgp_internal.h"

namespace dgp {

static constexpr int NB = DGP_TILE;
// k-tiles of register prefetch (TileGemm::run<PF>) for the kernels whose operands mostly miss L2
// (measured at n = 8192: fp64 lauum 2.99 -> 2.87 ms with 2; fp64 trtri gets SLOWER with 2 -- 256 VGPRs; fp32 has room)
template <typename T>
struct Prefetch {
  static constexpr bool F64 = sizeof(T) == 8;
  static constexpr int LAUUM = F64 ? 2 : 4, TRTRI = F64 ? 1 : 4, SYRK = F64 ? 1 : 2;
};

// ------------------------------------------------------------------------------------------
// A[i,k] <- A[i,k] * L_kk^-T  (L_kk^-1 lives in the diagonal block of Tinv).  This kernel sits on the
// sequential panel chain, so it uses 64-row tiles: twice the workgroups, half the per-workgroup latency.
template <typename T>
__global__ __launch_bounds__(256, 2) void trsm_kernel(T* __restrict__ A, const T* __restrict__ Tinv, long ld, int k) {
  // 64 rows x all 128 panel columns per workgroup: a workgroup only ever reads the rows it overwrites
  using G = TileGemm<T, true, true, 64, 128>;
  __shared__ T smem[G::SMEM_ELEMS];
  __builtin_amdgcn_s_setprio(3);  // panel chain: outrank co-resident bulk-update waves
  const long row0 = (long)(k + 1) * NB + (long)blockIdx.x * 64;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  T* Arow = A + row0 * ld + (long)k * NB;
  G::run(Arow, ld, Tinv + (long)k * NB * ld + (long)k * NB, ld, NB / 16, smem, acc);
  G::foreach (acc, [&](int r, int c, T& v) { Arow[(long)r * ld + c] = v; });
}

// A[i,j] -= sum over panels k..k+nk-1 of A[i,p] A[j,p]^T for the lower tiles with block column >= jbeg
// (bulk trailing update; nk = 2 halves the passes over the trailing matrix: 53.6 vs 44.5 TFLOP/s in isolation).
// The accumulators start at -C, so the read of C overlaps the first operand loads and the epilogue
// is store-only:  C_new = -( -C + P_i P_j^T ).
template <typename T>
__global__ __launch_bounds__(256, 2) void syrk_kernel(T* __restrict__ A, long ld, int k, int nk, int jbeg) {
  using G = TileGemm<T, true, true>;
  __shared__ T smem[G::SMEM_ELEMS];
  int bi, bj;
  tri_decode(xcd_remap((int)blockIdx.x, (int)gridDim.x), bi, bj);  // all tiles cost the same: remap freely
  bi += jbeg;
  bj += jbeg;
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + (long)bi * NB * ld + (long)bj * NB;
  G::foreach (acc, [&](int r, int c, T& v) { v = -C[(long)r * ld + c]; });
  G::template run<Prefetch<T>::SYRK>(A + (long)bi * NB * ld + (long)k * NB, ld, A + (long)bj * NB * ld + (long)k * NB, ld,
                                     nk * (NB / 16), smem, acc);
  G::foreach (acc, [&](int r, int c, T& v) { C[(long)r * ld + c] = -v; });
}

// the lookahead column: only block column jcol, 64x64 tiles (latency-critical, see trsm_kernel)
template <typename T>
__global__ __launch_bounds__(256, 2) void syrk_col_kernel(T* __restrict__ A, long ld, int k, int nk, int jcol) {
  using G = TileGemm<T, true, true, 64, 64>;
  __shared__ T smem[G::SMEM_ELEMS];
  __builtin_amdgcn_s_setprio(3);  // panel chain: outrank co-resident bulk-update waves
  const long row0 = (long)jcol * NB + (long)blockIdx.x * 64;
  const long col0 = (long)jcol * NB + (long)blockIdx.y * 64;
  if (col0 > row0 + 63) return;  // strictly upper 64x64 quadrant of the diagonal block
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + row0 * ld + col0;
  G::foreach (acc, [&](int r, int c, T& v) { v = -C[(long)r * ld + c]; });
  G::run(A + row0 * ld + (long)k * NB, ld, A + col0 * ld + (long)k * NB, ld, nk * (NB / 16), smem, acc);
  G::foreach (acc, [&](int r, int c, T& v) { C[(long)r * ld + c] = -v; });
}

template <typename T>
__global__ void zero1_kernel(T* p, int* info) {
  p[0] = T(0);
  info[0] = 0;
}

// launch the LDS-resident diagonal-block kernel (needs > 64 KB of dynamic LDS: opt in once per type)
template <typename T>
static void launch_diag(T* A, long N, long k0, T* Tinv, T* logdet, int* info, hipStream_t s) {
  static bool configured = false;
  const size_t bytes = potrf_diag_fast_smem<T>();
  if (!configured) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_fast_kernel<T>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    configured = true;
  }
  potrf_diag_fast_kernel<T><<<1, 256, bytes, s>>>(A, N, k0, Tinv, logdet, info);
}

template <typename T>
int potrf(T* A, long N, T* Tinv, T* logdet, int* info, int lookahead, hipStream_t s, hipStream_t s2, hipEvent_t* ev,
          hipEvent_t* syrk_ev, int* n_syrk, double* syrk_flop) {
  const int nbk = (int)(N / NB);
  int ns = 0;
  double flop = 0.0;
  zero1_kernel<T><<<1, 1, 0, s>>>(logdet, info);
  auto tri = [](int m) { return (unsigned)(m * (m + 1) / 2); };
  const double tile_flop = 2.0 * NB * NB * NB;
  if (!lookahead || nbk < 4 || s2 == nullptr || ev == nullptr) {
    for (int k = 0; k < nbk; ++k) {
      launch_diag<T>(A, N, (long)k * NB, Tinv, logdet, info, s);
      if (k + 1 < nbk) {
        trsm_kernel<T><<<2 * (nbk - k - 1), 256, 0, s>>>(A, Tinv, N, k);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s);
        syrk_kernel<T><<<tri(nbk - k - 1), 256, 0, s>>>(A, N, k, 1, k + 1);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s);
        flop += tile_flop * tri(nbk - k - 1);
        ++ns;
      }
    }
    if (n_syrk) *n_syrk = ns;
    if (syrk_flop) *syrk_flop = flop;
    return (int)hipGetLastError();
  }
  if (lookahead >= 2) {
    // PAIR-AHEAD schedule.  Pair q = panels (2q, 2q+1).  The chain on stream s factors pair q+1 while the bulk
    // update of pair q runs on s2, so consecutive bulk launches follow each other without a gap:
    //   chain(q):  [wait U[q-2]]  col 2q   <- pair q-1 (K=256)            diag(2q)    trsm(2q)
    //                             col 2q+1 <- pair q-1 + panel 2q (K=384) diag(2q+1)  trsm(2q+1)   record P[q]
    //   bulk(q):   [wait P[q]]    block columns >= 2q+4 <- pair q (K=256)                           record U[q]
    // Column c gets pairs <= c/2-2 from the bulk launches, pair c/2-1 and its own pair's first panel from the chain.
    hipEvent_t* P = ev;
    hipEvent_t* U = ev + nbk;
    const int Q = (nbk + 1) / 2;
    for (int q = 0; q < Q; ++q) {
      if (q >= 2) hipStreamWaitEvent(s, U[q - 2], 0);  // bulk(q-2) exists whenever chain(q) does
      for (int h = 0; h < 2; ++h) {
        const int k = 2 * q + h;
        if (k >= nbk) break;
        const int kfirst = q >= 1 ? 2 * q - 2 : 2 * q;
        const int nk = k - kfirst;  // panels not yet applied to column k
        if (nk > 0) syrk_col_kernel<T><<<dim3(2 * (nbk - k), 2), 256, 0, s>>>(A, N, kfirst, nk, k);
        launch_diag<T>(A, N, (long)k * NB, Tinv, logdet, info, s);
        if (k + 1 < nbk) trsm_kernel<T><<<2 * (nbk - k - 1), 256, 0, s>>>(A, Tinv, N, k);
      }
      if (2 * q + 4 < nbk) {
        hipEventRecord(P[q], s);
        hipStreamWaitEvent(s2, P[q], 0);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s2);
        syrk_kernel<T><<<tri(nbk - 2 * q - 4), 256, 0, s2>>>(A, N, 2 * q, 2, 2 * q + 4);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s2);
        flop += 2.0 * tile_flop * tri(nbk - 2 * q - 4);
        ++ns;
        hipEventRecord(U[q], s2);
      }
    }
    // bulk(q) exists for q <= Q-3 and chain(q+2) has waited on every one of them: s is joined
    if (n_syrk) *n_syrk = ns;
    if (syrk_flop) *syrk_flop = flop;
    return (int)hipGetLastError();
  }
  // One-panel lookahead with PAIRED bulk updates.  Stream s carries the panel chain
  //     column update (k -> k+1)  ->  diag(k+1)  ->  trsm(k+1)
  // and stream s2 (lowest priority) the bulk trailing update, launched once per two panels with
  // K = 256:   after panel k = 2p+1 is ready, panels (2p, 2p+1) update block columns >= 2p+3.
  // Column 2p+1 gets panel 2p, and column 2p+2 gets panels (2p, 2p+1), from the chain itself.
  //   P[k]: panel k ready (recorded for odd k)      U[p]: bulk update of pair p done
  hipEvent_t* P = ev;
  hipEvent_t* U = ev + nbk;
  launch_diag<T>(A, N, 0, Tinv, logdet, info, s);
  trsm_kernel<T><<<2 * (nbk - 1), 256, 0, s>>>(A, Tinv, N, 0);
  int last_u = -1;
  for (int k = 0; k + 1 < nbk; ++k) {
    const bool odd = (k & 1) != 0;
    const int p = k >> 1;
    if (odd && k + 2 < nbk) {
      hipStreamWaitEvent(s2, P[k], 0);
      if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s2);
      syrk_kernel<T><<<tri(nbk - k - 2), 256, 0, s2>>>(A, N, k - 1, 2, k + 2);
      if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s2);
      flop += 2.0 * tile_flop * tri(nbk - k - 2);
      ++ns;
      hipEventRecord(U[p], s2);
    }
    // column k+1 must carry every earlier pair before the chain adds its own panels: pair p-1 is the last
    // bulk launch that touches it (pair p starts at column 2p+3 > k+1)
    if (p >= 1 && last_u < p - 1) {
      hipStreamWaitEvent(s, U[p - 1], 0);
      last_u = p - 1;
    }
    if (odd) syrk_col_kernel<T><<<dim3(2 * (nbk - k - 1), 2), 256, 0, s>>>(A, N, k - 1, 2, k + 1);
    else syrk_col_kernel<T><<<dim3(2 * (nbk - k - 1), 2), 256, 0, s>>>(A, N, k, 1, k + 1);
    launch_diag<T>(A, N, (long)(k + 1) * NB, Tinv, logdet, info, s);
    if (k + 2 < nbk) {
      trsm_kernel<T><<<2 * (nbk - k - 2), 256, 0, s>>>(A, Tinv, N, k + 1);
      if (((k + 1) & 1) != 0 && k + 3 < nbk) hipEventRecord(P[k + 1], s);
    }
  }
  // join: the last bulk launch must precede whatever the caller enqueues next on s
  {
    const int kmax = ((nbk - 3) & 1) ? nbk - 3 : nbk - 4;  // largest odd k with k + 2 < nbk
    const int last_pair = kmax >= 1 ? (kmax - 1) / 2 : -1;
    if (last_pair >= 0 && last_u < last_pair) hipStreamWaitEvent(s, U[last_pair], 0);
  }
  if (n_syrk) *n_syrk = ns;
  if (syrk_flop) *syrk_flop = flop;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// trtri level with half-size m (in tiles of BT rows): group g covers tile rows/cols [lo, hi), split at mid.
//   W[i,j] = sum_{c=j}^{mid-1} L[i,c] T[c,j]      (i in [mid,hi), j in [lo,mid))
//   T[i,j] = - sum_{c=mid}^{i} T[i,c] W[c,j]
// BT = 128 for the large levels; the small levels (few tiles, short k) use 64x64 tiles so that four
// times as many workgroups share the work.
template <typename T, int STEP, int BT>
__global__ __launch_bounds__(256, 2) void trtri_level_kernel(const T* __restrict__ L, T* __restrict__ Tm,
                                                             T* __restrict__ W, long ld, int m, int ntile) {
  using G = TileGemm<T, true, false, BT, BT>;
  __shared__ T smem[G::SMEM_ELEMS];
  const int lo = 2 * m * blockIdx.y, mid = lo + m, hi = min(lo + 2 * m, ntile);
  // longest k-range first: W-step K ~ (mid - j), T-step K ~ (i - mid + 1)
  const int i = STEP == 0 ? mid + blockIdx.x % m : mid + (m - 1 - blockIdx.x / m);
  const int j = STEP == 0 ? lo + blockIdx.x / m : lo + blockIdx.x % m;
  if (i >= hi) return;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  constexpr int KT = BT / 16;  // k-tiles per tile of the reduction dimension
  if (STEP == 0) {
    G::template run<Prefetch<T>::TRTRI>(L + (long)i * BT * ld + (long)j * BT, ld, Tm + (long)j * BT * ld + (long)j * BT, ld,
                              (mid - j) * KT, smem, acc);
    T* out = W + (long)i * BT * ld + (long)j * BT;
    G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; });
  } else {
    G::template run<Prefetch<T>::TRTRI>(Tm + (long)i * BT * ld + (long)mid * BT, ld, W + (long)mid * BT * ld + (long)j * BT, ld,
                              (i - mid + 1) * KT, smem, acc);
    T* out = Tm + (long)i * BT * ld + (long)j * BT;
    G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = -v; });
  }
}

template <typename T, int BT>
static void trtri_level(const T* L, T* Tm, T* W, long N, int mblk, hipStream_t s) {
  const int per = NB / BT;                   // tiles per 128-block
  const int m = mblk * per, ntile = (int)(N / BT);
  const int ngroups = (ntile + 2 * m - 1) / (2 * m);
  dim3 grid((unsigned)(m * m), (unsigned)ngroups);
  trtri_level_kernel<T, 0, BT><<<grid, 256, 0, s>>>(L, Tm, W, N, m, ntile);
  trtri_level_kernel<T, 1, BT><<<grid, 256, 0, s>>>(L, Tm, W, N, m, ntile);
}

template <typename T>
int trtri(const T* L, const T* /*Dinv: already the diagonal blocks of Tm*/, long N, T* Tm, T* W, hipStream_t s) {
  const int nbk = (int)(N / NB);
  for (int m = 1; m < nbk; m *= 2) {
    if (m <= 16) trtri_level<T, 64>(L, Tm, W, N, m, s);  // <= 512 tiles of 128^2 would leave the GPU waiting on the longest one
    else trtri_level<T, 128>(L, Tm, W, N, m, s);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// S[i,j] = sum_{c >= i} T[c,i]^T T[c,j]   (i >= j): K^^-1 = L^-T L^-1
template <typename T>
__global__ __launch_bounds__(256, 2) void lauum_kernel(const T* __restrict__ Tm, T* __restrict__ S, long ld, int nbk) {
  using G = TileGemm<T, false, false>;
  __shared__ T smem[G::SMEM_ELEMS];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);  // ascending bi: the long-K tiles are dispatched first
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* base = Tm + (long)bi * NB * ld;
  G::template run<Prefetch<T>::LAUUM>(base + (long)bi * NB, ld, base + (long)bj * NB, ld, (nbk - bi) * (NB / 16), smem, acc);
  T* out = S + (long)bi * NB * ld + (long)bj * NB;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; });
}

template <typename T>
int lauum(const T* Tm, long N, T* S, hipStream_t s) {
  const int nbk = (int)(N / NB);
  lauum_kernel<T><<<(unsigned)(nbk * (nbk + 1) / 2), 256, 0, s>>>(Tm, S, N, nbk);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// z_i = sum_{j <= i} T[i][j] r_j : one wave per row (coalesced along j)
template <typename T>
__global__ __launch_bounds__(256) void trmv_n_kernel(const T* __restrict__ Tm, long ld, const T* __restrict__ r, int n,
                                                     T* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long jend = (i / NB + 1) * NB;  // the diagonal block is zero above the diagonal
  const T* row = Tm + i * ld;
  T acc = T(0);
  for (long j = lane; j < jend; j += 64) acc += row[j] * (j < n ? r[j] : T(0));
  acc = wave_sum(acc);
  if (lane == 0) z[i] = acc;
}

// partial[c][j] = sum_{i in chunk c, i >= blockrow(j)} T[i][j] z_i  (64 columns per workgroup)
#define DGP_TRMV_CHUNK 512
template <typename T>
__global__ __launch_bounds__(256) void trmv_t_kernel(const T* __restrict__ Tm, long ld, const T* __restrict__ z,
                                                     T* __restrict__ partial) {
  __shared__ T red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + tx;
  const long r0 = (long)blockIdx.y * DGP_TRMV_CHUNK, r1 = min(r0 + (long)DGP_TRMV_CHUNK, ld);
  const long rstart = max(r0, ((long)blockIdx.x * 64 / NB) * NB);
  T acc = T(0);
  for (long i = rstart + ty; i < r1; i += 4) acc += Tm[i * ld + j] * z[i];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) partial[(long)blockIdx.y * ld + j] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

template <typename T>
__global__ __launch_bounds__(256) void trmv_t_reduce_kernel(const T* __restrict__ partial, long N, int nchunks,
                                                            T* __restrict__ alpha) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T acc = T(0);
  const int c0 = (int)((j / NB) * NB / DGP_TRMV_CHUNK);
  for (int c = c0; c < nchunks; ++c) acc += partial[(long)c * N + j];
  alpha[j] = acc;
}

template <typename T>
__global__ __launch_bounds__(256) void sumsq_kernel(const T* __restrict__ z, long N, T* __restrict__ out) {
  __shared__ T red[256];
  T acc = T(0);
  for (long i = threadIdx.x; i < N; i += 256) acc += z[i] * z[i];
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[0] = red[0];
}

// beta = S g for a symmetric matrix stored in its lower triangle (diagonal tiles hold both halves):
// row part  sum_{j <= i} S[i][j] g_j  +  column part  sum_{i > j} S[i][j] g_i
template <typename T>
__global__ __launch_bounds__(256) void symv_row_kernel(const T* __restrict__ S, long ld, const T* __restrict__ g,
                                                       T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const T* row = S + i * ld;
  T acc = T(0);
  for (long j = lane; j <= i; j += 64) acc += row[j] * g[j];
  acc = wave_sum(acc);
  if (lane == 0) out[i] = acc;
}
template <typename T>
__global__ __launch_bounds__(256) void symv_col_kernel(const T* __restrict__ S, long ld, const T* __restrict__ g,
                                                       T* __restrict__ partial) {
  __shared__ T red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + tx;
  const long r0 = (long)blockIdx.y * DGP_TRMV_CHUNK, r1 = min(r0 + (long)DGP_TRMV_CHUNK, ld);
  const long rstart = max(r0, (long)blockIdx.x * 64);
  T acc = T(0);
  for (long i = rstart + ty; i < r1; i += 4) acc += (i > j) ? S[i * ld + j] * g[i] : T(0);
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) partial[(long)blockIdx.y * ld + j] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
template <typename T>
__global__ __launch_bounds__(256) void symv_finish_kernel(const T* __restrict__ partial, long N, int nchunks,
                                                          const T* __restrict__ alpha, int n, T* __restrict__ beta,
                                                          T* __restrict__ dnoise) {
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T acc = beta[j];  // row part
  for (int c = (int)(j / DGP_TRMV_CHUNK); c < nchunks; ++c) acc += partial[(long)c * N + j];
  acc = j < n ? acc : T(0);
  beta[j] = acc;
  if (j < n && dnoise) dnoise[j] = -acc * alpha[j];
}

template <typename T>
int symv_lower(const T* S, long N, const T* g, int n, const T* alpha, T* beta, T* partials, T* dnoise, hipStream_t s) {
  const int nchunks = (int)((N + DGP_TRMV_CHUNK - 1) / DGP_TRMV_CHUNK);
  symv_row_kernel<T><<<(unsigned)(N / 4), 256, 0, s>>>(S, N, g, beta);
  dim3 grid((unsigned)(N / 64), (unsigned)nchunks);
  symv_col_kernel<T><<<grid, 256, 0, s>>>(S, N, g, partials);
  symv_finish_kernel<T><<<(unsigned)((N + 255) / 256), 256, 0, s>>>(partials, N, nchunks, alpha, n, beta, dnoise);
  return (int)hipGetLastError();
}

long solve_partials(long N) { return (N + DGP_TRMV_CHUNK - 1) / DGP_TRMV_CHUNK * N; }

template <typename T>
int solve(const T* Tm, long N, const T* r, int n, T* z, T* alpha, T* partials, T* quad, hipStream_t s) {
  trmv_n_kernel<T><<<(unsigned)(N / 4), 256, 0, s>>>(Tm, N, r, n, z);
  sumsq_kernel<T><<<1, 256, 0, s>>>(z, N, quad);
  const int nchunks = (int)((N + DGP_TRMV_CHUNK - 1) / DGP_TRMV_CHUNK);
  dim3 grid((unsigned)(N / 64), (unsigned)nchunks);
  trmv_t_kernel<T><<<grid, 256, 0, s>>>(Tm, N, z, partials);
  trmv_t_reduce_kernel<T><<<(unsigned)((N + 255) / 256), 256, 0, s>>>(partials, N, nchunks, alpha);
  return (int)hipGetLastError();
}

// dNLL/dnoise_i = 1/2 (S_ii - alpha_i^2)
template <typename T>
__global__ __launch_bounds__(256) void dnoise_kernel(const T* __restrict__ S, const T* __restrict__ alpha, long N,
                                                     int n, T* __restrict__ dnoise) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dnoise[i] = T(0.5) * (S[i * N + i] - alpha[i] * alpha[i]);
}

template <typename T>
int finish(const T* S, const T* alpha, long N, int n, T* dnoise, hipStream_t s) {
  dnoise_kernel<T><<<(unsigned)((n + 255) / 256), 256, 0, s>>>(S, alpha, N, n, dnoise);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// prediction: V = T Ks (N x M, Ks = K(X, X*) padded to M % 128 == 0), var_j = kss_j - sum_i V_ij^2,
// mean_j = sum_i Ks_ij alpha_i      (src/discontinuum/engines/gpytorch.py:621-624)
template <typename T>
__global__ __launch_bounds__(256, 2) void predict_v_kernel(const T* __restrict__ Tm, long N, const T* __restrict__ Ks,
                                                        long M, T* __restrict__ V) {
  using G = TileGemm<T, true, false>;
  __shared__ T smem[G::SMEM_ELEMS];
  const int bi = blockIdx.y, bj = blockIdx.x;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  G::run(Tm + (long)bi * NB * N, N, Ks + (long)bj * NB, M, (bi + 1) * (NB / 16), smem, acc);
  T* out = V + (long)bi * NB * M + (long)bj * NB;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * M + c] = v; });
}

template <typename T>
__global__ __launch_bounds__(256) void predict_reduce_kernel(const T* __restrict__ V, const T* __restrict__ Ks, long N,
                                                             long M, const T* __restrict__ alpha,
                                                             const T* __restrict__ kss, T* __restrict__ mean,
                                                             T* __restrict__ var) {
  __shared__ T r1[4][64], r2[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + tx;
  T s1 = T(0), s2 = T(0);
  for (long i = ty; i < N; i += 4) {
    const T v = V[i * M + j];
    s2 += v * v;
    s1 += Ks[i * M + j] * alpha[i];
  }
  r1[ty][tx] = s1;
  r2[ty][tx] = s2;
  __syncthreads();
  if (ty == 0) {
    mean[j] = r1[0][tx] + r1[1][tx] + r1[2][tx] + r1[3][tx];
    var[j] = kss[j] - (r2[0][tx] + r2[1][tx] + r2[2][tx] + r2[3][tx]);
  }
}

template <typename T>
int predict_var(const T* Tm, long N, const T* Ks, long M, T* V, const T* alpha, const T* kss, T* mean, T* var,
                hipStream_t s) {
  dim3 grid((unsigned)(M / NB), (unsigned)(N / NB));
  predict_v_kernel<T><<<grid, 256, 0, s>>>(Tm, N, Ks, M, V);
  predict_reduce_kernel<T><<<(unsigned)(M / 64), 256, 0, s>>>(V, Ks, N, M, alpha, kss, mean, var);
  return (int)hipGetLastError();
}

// cov[i,j] = Kss[i,j] - sum_k V[k,i] V[k,j]  (i >= j tiles), k over all N rows of V (N x M)
template <typename T>
__global__ __launch_bounds__(256, 2) void posterior_cov_kernel(const T* __restrict__ V, long N, long M, T* __restrict__ cov) {
  using G = TileGemm<T, false, false>;
  __shared__ T smem[G::SMEM_ELEMS];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  typename G::acc_t acc[G::MI][G::NI];
  T* C = cov + (long)bi * NB * M + (long)bj * NB;
  G::foreach (acc, [&](int r, int c, T& v) { v = -C[(long)r * M + c]; });
  G::run(V + (long)bi * NB, M, V + (long)bj * NB, M, (int)(N / 16), smem, acc);
  G::foreach (acc, [&](int r, int c, T& v) { C[(long)r * M + c] = -v; });
}

template <typename T>
int posterior_cov(const T* V, long N, long M, T* cov, hipStream_t s) {
  const int nb = (int)(M / NB);
  posterior_cov_kernel<T><<<(unsigned)(nb * (nb + 1) / 2), 256, 0, s>>>(V, N, M, cov);
  return (int)hipGetLastError();
}

#define DGP_INST(T)                                                                                              \
  template int posterior_cov<T>(const T*, long, long, T*, hipStream_t);                                          \
  template int symv_lower<T>(const T*, long, const T*, int, const T*, T*, T*, T*, hipStream_t);                  \
  template int potrf<T>(T*, long, T*, T*, int*, int, hipStream_t, hipStream_t, hipEvent_t*, hipEvent_t*, int*, double*);                     \
  template int trtri<T>(const T*, const T*, long, T*, T*, hipStream_t);                                          \
  template int lauum<T>(const T*, long, T*, hipStream_t);                                                        \
  template int solve<T>(const T*, long, const T*, int, T*, T*, T*, T*, hipStream_t);                             \
  template int finish<T>(const T*, const T*, long, int, T*, hipStream_t);                                        \
  template int predict_var<T>(const T*, long, const T*, long, T*, const T*, const T*, T*, T*, hipStream_t);
DGP_INST(double)
DGP_INST(float)

}  // namespace dgp
