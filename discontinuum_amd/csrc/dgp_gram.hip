// dgp_gram.hip -- fused kernel-Gram assembly and its hyperparameter-gradient contraction.
//
//   gram_sym    K^ = K(X, X; theta) + diag(noise), lower 64x64 tiles of the padded N x N matrix.
//               Replaces gpytorch's lazy covar_module(x) + likelihood noise that the reference
//               evaluates at src/discontinuum/engines/gpytorch.py:350-353.
//               Bytes per launch: N(N+64)/2 * sizeof(T) written (+ d*N read) -- HBM-write roofline.
//   gram_cross  K(X, X*) rectangular (prediction, engines/gpytorch.py:621-624).
//   gram_grad   dNLL/dtheta_p = 1/2 sum_ij (S - alpha alpha^T)_ij dK_ij/dtheta_p for all p at once:
//               streams S = K^^-1 once (N(N+64)/2 * sizeof(T) read), recomputes every sub-kernel in
//               registers, wave-shuffle + LDS block reduction, deterministic two-stage sum.
// Coordinates are SoA (d x N) so that the strip loads are coalesced; strips are staged in LDS.
#include <mutex>
#include <string.h>
#include "dgp_internal.h"
#include "dgp_models.h"
#include "dgp_gram_shared.h"

namespace dgp {

size_t pre_scratch_bytes(int B) { return B > DGP_MAX_BATCH ? (size_t)B * DGP_PRE_SLOT_BYTES : 0; }

// ---- registry of generic composite models (dgp_composite_define) -------------------------------------------------
#define DGP_C_MAXMODELS 64
static CompositeDesc g_comp[DGP_C_MAXMODELS];
static int g_ncomp = 0;
static std::mutex g_comp_mtx;
static thread_local const CompositeDesc* g_comp_cur = nullptr;
const CompositeDesc* composite_current() { return g_comp_cur; }
static const CompositeDesc* composite_get(int model) {
  const int slot = model - DGP_MODEL_COMPOSITE_BASE;
  return (slot >= 0 && slot < g_ncomp) ? &g_comp[slot] : nullptr;
}

// spec: [d, nterms, then per term: scaled (0/1), nfac, then per factor: type, 2 nu, ard (0/1), ndims, dims...].
// theta order: per term [outputscale if scaled], per factor [lengthscale(s)], [period if periodic].
// Returns the model id (>= DGP_MODEL_COMPOSITE_BASE; an identical description registered before is reused), -2 for a
// malformed / unsupported description, -5 when all DGP_C_MAXMODELS slots hold other structures.
int composite_define(const int* spec, int nspec) {
  CompositeDesc c;
  memset(&c, 0, sizeof(c));
  int i = 0, th = 0;
  auto next = [&](int& v) { if (i >= nspec) return false; v = spec[i++]; return true; };
  int d, nterms;
  if (!next(d) || !next(nterms) || d < 1 || d > DGP_C_DMAX || nterms < 1 || nterms > DGP_C_TMAX) return -2;
  c.d = (unsigned char)d;
  c.nterms = (unsigned char)nterms;
  for (int t = 0; t < nterms; ++t) {
    int scaled, nfac;
    if (!next(scaled) || !next(nfac) || nfac < 1 || nfac > DGP_C_FMAX) return -2;
    c.term[t].os = scaled ? (signed char)th++ : (signed char)-1;
    c.term[t].nfac = (unsigned char)nfac;
    for (int f = 0; f < nfac; ++f) {
      int type, nu2, ard, ndims;
      if (!next(type) || !next(nu2) || !next(ard) || !next(ndims) || ndims < 1 || ndims > d) return -2;
      if (type != DGP_FAC_RBF && type != DGP_FAC_MATERN && type != DGP_FAC_PERIODIC) return -2;
      if (type == DGP_FAC_MATERN && nu2 != 1 && nu2 != 3 && nu2 != 5) return -2;
      if (type == DGP_FAC_PERIODIC && ndims != 1) return -2;  // one column per periodic factor
      CompositeDesc::Fac& fc = c.term[t].fac[f];
      fc.type = (unsigned char)type; fc.nu2 = (unsigned char)nu2; fc.ard = ard ? 1 : 0; fc.ndims = (unsigned char)ndims;
      for (int j = 0; j < ndims; ++j) {
        int col;
        if (!next(col) || col < 0 || col >= d) return -2;
        fc.dims[j] = (unsigned char)col;
      }
      fc.ls = (signed char)th;
      th += ard ? ndims : 1;
      fc.period = type == DGP_FAC_PERIODIC ? (signed char)th++ : (signed char)-1;
      if (th > DGP_MAX_THETA) return -2;
    }
  }
  if (i != nspec) return -2;
  c.ntheta = (unsigned char)th;
  std::lock_guard<std::mutex> lock(g_comp_mtx);
  for (int k = 0; k < g_ncomp; ++k)
    if (memcmp(&g_comp[k], &c, sizeof(c)) == 0) return DGP_MODEL_COMPOSITE_BASE + k;
  if (g_ncomp >= DGP_C_MAXMODELS) return -5;  // registry full (reported as such by dgp_composite_define)
  g_comp[g_ncomp] = c;
  return DGP_MODEL_COMPOSITE_BASE + g_ncomp++;
}

int model_ntheta(int model, int d) {
  if (model == DGP_MODEL_LOADEST) return (d >= 2 && d <= 6) ? 2 * d + 5 : -1;
  if (model == DGP_MODEL_RATING) return d == 2 ? 16 : -1;
  if (const CompositeDesc* c = composite_get(model)) return c->d == d ? c->ntheta : -1;
  return -1;
}

template <typename T>
__global__ __launch_bounds__(256) void pack_x_kernel(const T* __restrict__ X, int n, int d, long N, T* __restrict__ Xt,
                                                     long bs, const int* __restrict__ ns) {
  X = site(X, (long)n * d);
  Xt = site(Xt, bs);
  n = site_n(ns, n);
  long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  for (int c = 0; c < d; ++c) Xt[(long)c * N + i] = i < n ? X[i * d + c] : T(0);
}

// ------------------------------------------------------------------------------------------
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_sym_kernel(const T* __restrict__ Xt, long N, int n, const PreBatch<M> pb,
                                                       const T* __restrict__ noise, T* __restrict__ K, long bs,
                                                       const int* __restrict__ ns, long ks, long noise_stride) {
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  K = site(K, ks);
  noise = site(noise, noise_stride);
  n = site_n(ns, n);
  __shared__ T sfi[M::NF][64], sfj[M::NF][64];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  const int t = threadIdx.x;
  exp_table_init<T>();
  if (t < 64) stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
  else if (t < 128) stage_strip<T, M>(Xt, N, (long)bj * 64, pre, sfj, t - 64);
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T fj[4][M::NF];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fj[b][c] = sfj[c][tx * 4 + b];
  T dummy[M::NTHETA];
  // interior tiles (strictly below the diagonal, no padding rows or columns) skip the per-entry pad / noise selects:
  // a workgroup-uniform branch that almost every tile takes
  const bool interior = bi != bj && (long)bi * 64 + 64 <= n;
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    T out[4];
    if (interior) {
#pragma unroll
      for (int b = 0; b < 4; ++b) out[b] = M::template pair<false>(fi, fj[b], pre, T(0), dummy);
    } else {
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const long gj = (long)bj * 64 + tx * 4 + b;
        T v = M::template pair<false>(fi, fj[b], pre, T(0), dummy);
        if (gi >= n || gj >= n) v = (gi == gj) ? T(1) : T(0);  // identity pad
        else if (gi == gj) v += noise[gi];
        out[b] = v;
      }
    }
    store4<T>(K + gi * N + (long)bj * 64 + tx * 4, out);
  }
}

// Batched plans (blockIdx.z = site): the training coordinates sit at the plan's site stride `bs`, the test coordinates
// and the output in the caller's work area at its own site stride `wbs`; ragged sites have n = ns[site] rows.
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_cross_kernel(const T* __restrict__ Xt, long N, int n,
                                                         const T* __restrict__ Xst, long Mp, int m, const PreBatch<M> pb,
                                                         T* __restrict__ Ks, long bs, long wbs, const int* __restrict__ ns) {
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  Xst = site(Xst, wbs);
  Ks = site(Ks, wbs);
  n = site_n(ns, n);
  __shared__ T sfi[M::NF][64], sfj[M::NF][64];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int t = threadIdx.x;
  exp_table_init<T>();
  if (t < 64) stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
  else if (t < 128) stage_strip<T, M>(Xst, Mp, (long)bj * 64, pre, sfj, t - 64);
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T fj[4][M::NF];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fj[b][c] = sfj[c][tx * 4 + b];
  T dummy[M::NTHETA];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    T out[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long gj = (long)bj * 64 + tx * 4 + b;
      T v = M::template pair<false>(fi, fj[b], pre, T(0), dummy);
      if (gi >= n || gj >= m) v = T(0);
      out[b] = v;
    }
    store4<T>(Ks + gi * Mp + (long)bj * 64 + tx * 4, out);
  }
}

// prior variance k(x*, x*) of the test points (diagonal of K**)
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_diag_kernel(const T* __restrict__ Xst, long Mp, int m, const PreBatch<M> pb,
                                                        T* __restrict__ kss, long wbs) {
  const typename M::Pre& pre = pb.get();
  Xst = site(Xst, wbs);
  kss = site(kss, wbs);
  exp_table_init<T>();
  __syncthreads();
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  T x[M::NX], f[M::NF], dummy[M::NTHETA];
#pragma unroll
  for (int c = 0; c < M::NX; ++c) x[c] = Xst[(long)c * Mp + j];
  M::features(x, pre, f);
  T v = M::template pair<false>(f, f, pre, T(0), dummy);
  kss[j] = j < m ? v : T(0);
}

// ------------------------------------------------------------------------------------------
// MODE 0: weights W_ij = S_ij - a_i a_j            (marginal likelihood:  1/2 tr((K^-1 - a a^T) dK))
// MODE 1: weights W_ij = -(b_i a_j + b_j a_i)       (predictive-mean VJP:  -b^T dK a, symmetrised; S unused)
template <typename T, typename M, int MODE>
__global__ __launch_bounds__(256) void gram_grad_kernel(const T* __restrict__ Xt, long N, int n, const PreBatch<M> pb,
                                                        const T* __restrict__ S, const T* __restrict__ alpha,
                                                        const T* __restrict__ beta, T* __restrict__ partials, long bs,
                                                        const int* __restrict__ ns, long wbs /* MODE 1: stride of beta / partials */) {
  n = site_n(ns, n);
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  if (MODE == 0) S = site(S, bs);
  alpha = site(alpha, bs);
  partials = site(partials, MODE == 0 ? bs : wbs);
  if (MODE == 1) beta = site(beta, wbs);
  __shared__ T sfi[M::NF][64], sfj[M::NF][64], sai[64], saj[64], sbi[64], sbj[64];
  __shared__ T red[4][M::NTHETA];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  const int t = threadIdx.x;
  exp_table_init<T>();
  if (t < 64) {
    stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
    sai[t] = alpha[(long)bi * 64 + t];
    sbi[t] = MODE == 1 ? beta[(long)bi * 64 + t] : T(0);
  } else if (t < 128) {
    stage_strip<T, M>(Xt, N, (long)bj * 64, pre, sfj, t - 64);
    saj[t - 64] = alpha[(long)bj * 64 + t - 64];
    sbj[t - 64] = MODE == 1 ? beta[(long)bj * 64 + t - 64] : T(0);
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T acc[M::NTHETA];
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) acc[p] = T(0);
  // one entry at a time: the derivative expressions are register-hungry, occupancy hides latency.  The column
  // point's features come from LDS per entry -- a register array indexed by the rolled loop's counter would live in
  // scratch memory.
#pragma unroll 1
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    const T ai = sai[ty * 4 + a], bi_ = sbi[ty * 4 + a];
    T sv[4] = {T(0), T(0), T(0), T(0)};
    if (MODE == 0) load4<T>(S + gi * N + (long)bj * 64 + tx * 4, sv);
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
      const int cj = tx * 4 + b;
      const long gj = (long)bj * 64 + cj;
      T fj[M::NF];
#pragma unroll
      for (int c = 0; c < M::NF; ++c) fj[c] = sfj[c][cj];
      const T aj = saj[cj];
      const T svb = b == 0 ? sv[0] : (b == 1 ? sv[1] : (b == 2 ? sv[2] : sv[3]));
      // lower triangle counted once with weight 1 (= 1/2 * 2), diagonal with 1/2, pad with 0
      T w = MODE == 0 ? svb - ai * aj : -(bi_ * aj + sbj[cj] * ai);
      w = (gj > gi || gi >= n) ? T(0) : (gj == gi ? T(0.5) * w : w);
      (void)M::template pair<true>(fi, fj, pre, w, acc);
    }
  }
  M::finalize(acc, pre);  // the pair-independent factors of the derivative sums, once per thread
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) {
    T v = wave_sum(acc[p]);
    if (lane == 0) red[wv][p] = v;
  }
  __syncthreads();
  if (t < M::NTHETA) partials[(long)blockIdx.x * DGP_MAX_THETA + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// dL/dtheta_p += sum_{i train, j test} a_i w_j dk(x_i, x*_j)/dtheta_p   (rectangular, no symmetry)
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_cross_grad_kernel(const T* __restrict__ Xt, long N, int n,
                                                              const T* __restrict__ Xst, long Mp, int m,
                                                              const PreBatch<M> pb, const T* __restrict__ alpha,
                                                              const T* __restrict__ wts, T* __restrict__ partials, long bs,
                                                              long wbs, const int* __restrict__ ns) {
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  alpha = site(alpha, bs);
  Xst = site(Xst, wbs);
  partials = site(partials, wbs);
  wts = site(wts, (long)m);
  n = site_n(ns, n);
  __shared__ T sfi[M::NF][64], sfj[M::NF][64], sai[64], swj[64];
  __shared__ T red[4][M::NTHETA];
  const int bi = blockIdx.y, bj = blockIdx.x;
  const int t = threadIdx.x;
  exp_table_init<T>();
  if (t < 64) {
    stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
    sai[t] = alpha[(long)bi * 64 + t];
  } else if (t < 128) {
    stage_strip<T, M>(Xst, Mp, (long)bj * 64, pre, sfj, t - 64);
    const long j = (long)bj * 64 + t - 64;
    swj[t - 64] = j < m ? wts[j] : T(0);
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T acc[M::NTHETA];
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) acc[p] = T(0);
#pragma unroll 1
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    const T ai = gi < n ? sai[ty * 4 + a] : T(0);
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
      T fj[M::NF];
#pragma unroll
      for (int c = 0; c < M::NF; ++c) fj[c] = sfj[c][tx * 4 + b];
      (void)M::template pair<true>(fi, fj, pre, ai * swj[tx * 4 + b], acc);
    }
  }
  M::finalize(acc, pre);  // the pair-independent factors of the derivative sums, once per thread
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) {
    T v = wave_sum(acc[p]);
    if (lane == 0) red[wv][p] = v;
  }
  __syncthreads();
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if (t < M::NTHETA) partials[blk * DGP_MAX_THETA + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// g_i = sum_j Ks[i][j] w_j  (N x Mp row-major, one wave per row)
template <typename T>
__global__ __launch_bounds__(256) void gemv_rows_kernel(const T* __restrict__ Ks, long N, long Mp, int m,
                                                        const T* __restrict__ w, T* __restrict__ out, long wbs) {
  Ks = site(Ks, wbs);
  out = site(out, wbs);
  w = site(w, (long)m);
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  T acc = T(0);
  for (long j = lane; j < m; j += 64) acc += Ks[i * Mp + j] * w[j];
  acc = wave_sum(acc);
  if (lane == 0) out[i] = acc;
}

// ------------------------------------------------------------------------------------------
// Column-slab variants for ONE matrix distributed over several ranks (dgp_dist.hip).  A rank holds the block columns it
// owns side by side: slab[r * Cl + lc], local 64-column tile lt <-> global tile gt = 2 gblock(lt / 2) + lt % 2.
// gram_slab writes K^ for the owned columns from their diagonal block down (rows >= column block, 64 x 64 tiles).
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_slab_kernel(const T* __restrict__ Xt, long N, int n, const typename M::Pre pre,
                                                        const T* __restrict__ noise, T* __restrict__ slab, long Cl, SlabMap sm) {
  const int bi = blockIdx.x, lt = blockIdx.y;
  const int bj = 2 * sm.gblock(lt >> 1) + (lt & 1);
  if ((long)bj * 64 >= N || bi < bj) return;
  __shared__ T sfi[M::NF][64], sfj[M::NF][64];
  const int t = threadIdx.x;
  exp_table_init<T>();
  if (t < 64) stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
  else if (t < 128) stage_strip<T, M>(Xt, N, (long)bj * 64, pre, sfj, t - 64);
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T fj[4][M::NF];
#pragma unroll
  for (int b = 0; b < 4; ++b)
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fj[b][c] = sfj[c][tx * 4 + b];
  T dummy[M::NTHETA];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    T out[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const long gj = (long)bj * 64 + tx * 4 + b;
      T v = M::template pair<false>(fi, fj[b], pre, T(0), dummy);
      if (gi >= n || gj >= n) v = (gi == gj) ? T(1) : T(0);  // identity pad
      else if (gi == gj) v += noise[gi];
      out[b] = v;
    }
    store4<T>(slab + gi * Cl + (long)lt * 64 + tx * 4, out);
  }
}

// The rank's share of 1/2 sum_ij (S - alpha alpha^T)_ij dK_ij / dtheta: its slab of S = K^^-1 holds, for every owned
// block column, the rows from the top down to the diagonal block (S_ij, i <= j).  Entries above the diagonal count
// once with weight 1 (= 1/2 * 2), the diagonal with 1/2.  Tiles outside that region write zero partials.
template <typename T, typename M>
__global__ __launch_bounds__(256) void gram_grad_slab_kernel(const T* __restrict__ Xt, long N, int n, const typename M::Pre pre,
                                                             const T* __restrict__ S, long Cl, SlabMap sm,
                                                             const T* __restrict__ alpha, T* __restrict__ partials,
                                                             T* __restrict__ dnoise_part) {
  __shared__ T sfi[M::NF][64], sfj[M::NF][64], sai[64], saj[64];
  __shared__ T red[4][M::NTHETA];
  const int bi = blockIdx.x, lt = blockIdx.y;  // row tile, local column tile
  const int bj = 2 * sm.gblock(lt >> 1) + (lt & 1);
  const int t = threadIdx.x;
  const long blk = (long)blockIdx.y * gridDim.x + blockIdx.x;
  if ((long)bj * 64 >= N || bi > bj) {
    if (t < M::NTHETA) partials[blk * DGP_MAX_THETA + t] = T(0);
    return;
  }
  exp_table_init<T>();
  if (t < 64) {
    stage_strip<T, M>(Xt, N, (long)bi * 64, pre, sfi, t);
    sai[t] = alpha[(long)bi * 64 + t];
  } else if (t < 128) {
    stage_strip<T, M>(Xt, N, (long)bj * 64, pre, sfj, t - 64);
    saj[t - 64] = alpha[(long)bj * 64 + t - 64];
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T acc[M::NTHETA];
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) acc[p] = T(0);
#pragma unroll 1
  for (int a = 0; a < 4; ++a) {
    const long gi = (long)bi * 64 + ty * 4 + a;
    T fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    const T ai = sai[ty * 4 + a];
    T sv[4];
    load4<T>(S + gi * Cl + (long)lt * 64 + tx * 4, sv);
#pragma unroll 1
    for (int b = 0; b < 4; ++b) {
      const int cj = tx * 4 + b;
      const long gj = (long)bj * 64 + cj;
      T fj[M::NF];
#pragma unroll
      for (int c = 0; c < M::NF; ++c) fj[c] = sfj[c][cj];
      const T aj = saj[cj];
      const T svb = b == 0 ? sv[0] : (b == 1 ? sv[1] : (b == 2 ? sv[2] : sv[3]));
      T w = svb - ai * aj;
      w = (gi > gj || gj >= n) ? T(0) : (gj == gi ? T(0.5) * w : w);
      if (gi == gj && gi < n) dnoise_part[gi] = T(0.5) * (svb - ai * ai);
      (void)M::template pair<true>(fi, fj, pre, w, acc);
    }
  }
  M::finalize(acc, pre);  // the pair-independent factors of the derivative sums, once per thread
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) {
    T v = wave_sum(acc[p]);
    if (lane == 0) red[wv][p] = v;
  }
  __syncthreads();
  if (t < M::NTHETA) partials[blk * DGP_MAX_THETA + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

// ------------------------------------------------------------------------------------------
// Iterative refinement of fp32 plans (SURVEY.md section 8d's fp32 row; the reference's dtype,
// src/discontinuum/engines/gpytorch.py:221-222): the residual rho = r - (K + diag(noise)) alpha in DOUBLE.  K^ was
// overwritten by its factor, so the covariance is re-evaluated from the stored fp32 coordinates with the fp64
// evaluator M = Model<double> and multiplied into alpha on the fly -- never materialised.  One workgroup per lower
// 64 x 64 tile (bi >= bj): its row sums go to part[bi][bj][0..63] and, for an off-diagonal tile, its column sums (the
// transposed tile) to part[bj][bi][..]; the second stage adds a row block's nb partials in block order, so the result
// does not depend on scheduling.  VALU-bound like gram_sym in fp64 (one pair evaluation per entry of the lower
// triangle): N(N + 64) / 2 evaluations per site.
template <typename TS, typename M>
__device__ __forceinline__ void stage_strip_as_double(const TS* __restrict__ Xt, long N, long base, const typename M::Pre& pre,
                                                      double (*sf)[64], int lane) {
  double x[M::NX], f[M::NF];
#pragma unroll
  for (int c = 0; c < M::NX; ++c) x[c] = (double)Xt[(long)c * N + base + lane];
  M::features(x, pre, f);
#pragma unroll
  for (int c = 0; c < M::NF; ++c) sf[c][lane] = f[c];
}

template <typename TS, typename M>
__global__ __launch_bounds__(256) void gram_matvec_kernel(const TS* __restrict__ Xt, long N, int n, const PreBatch<M> pb,
                                                          const TS* __restrict__ alpha, double* __restrict__ part, long bs,
                                                          long ps, const int* __restrict__ ns) {
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  alpha = site(alpha, bs);
  part = site(part, ps);
  n = site_n(ns, n);
  __shared__ double sfi[M::NF][64], sfj[M::NF][64], sai[64], saj[64];
  __shared__ double racc[16][64], cacc[16][64];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  const int nb = (int)(N / 64);
  const int t = threadIdx.x;
  exp_table_init<double>();
  if (t < 64) {
    stage_strip_as_double<TS, M>(Xt, N, (long)bi * 64, pre, sfi, t);
    const long gi = (long)bi * 64 + t;
    sai[t] = gi < n ? (double)alpha[gi] : 0.0;  // pad rows / columns contribute nothing
  } else if (t < 128) {
    stage_strip_as_double<TS, M>(Xt, N, (long)bj * 64, pre, sfj, t - 64);
    const long gj = (long)bj * 64 + t - 64;
    saj[t - 64] = gj < n ? (double)alpha[gj] : 0.0;
  }
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  double fj[4][M::NF], aj[4], cs[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
  for (int b = 0; b < 4; ++b) {
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fj[b][c] = sfj[c][tx * 4 + b];
    aj[b] = saj[tx * 4 + b];
  }
  double dummy[M::NTHETA];
#pragma unroll
  for (int a = 0; a < 4; ++a) {
    double fi[M::NF];
#pragma unroll
    for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ty * 4 + a];
    const double ai = sai[ty * 4 + a];
    double rsum = 0.0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const double v = M::template pair<false>(fi, fj[b], pre, 0.0, dummy);
      rsum += v * aj[b];
      cs[b] += v * ai;
    }
    racc[tx][ty * 4 + a] = rsum;
  }
#pragma unroll
  for (int b = 0; b < 4; ++b) cacc[ty][tx * 4 + b] = cs[b];
  __syncthreads();
  if (t < 64) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += racc[k][t];
    part[((long)bi * nb + bj) * 64 + t] = v;
  } else if (t < 128 && bi != bj) {
    double v = 0.0;
#pragma unroll
    for (int k = 0; k < 16; ++k) v += cacc[k][t - 64];
    part[((long)bj * nb + bi) * 64 + t - 64] = v;
  }
}

// rho_i = r_i - noise_i alpha_i - sum_b part[i / 64][b][i % 64]   (i < n; 0 in the pad), as double and rounded to TS
template <typename TS>
__global__ __launch_bounds__(256) void resid_reduce_kernel(const double* __restrict__ part, long N, int n,
                                                           const TS* __restrict__ r, const TS* __restrict__ noise,
                                                           const TS* __restrict__ alpha, double* __restrict__ rho64,
                                                           TS* __restrict__ rho32, long bs, long ps, long rs,
                                                           const int* __restrict__ ns) {
  part = site(part, ps);
  rho64 = site(rho64, ps);
  rho32 = site(rho32, rs);
  alpha = site(alpha, bs);
  r = site(r, (long)n);
  noise = site(noise, (long)n);
  n = site_n(ns, n);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= N) return;
  const int nb = (int)(N / 64);
  const double* p = part + (i / 64) * (long)nb * 64 + (i & 63);
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;  // four fixed interleaved chains (latency), joined in a fixed order
  int b = 0;
  for (; b + 4 <= nb; b += 4) {
    s0 += p[(long)b * 64];
    s1 += p[(long)(b + 1) * 64];
    s2 += p[(long)(b + 2) * 64];
    s3 += p[(long)(b + 3) * 64];
  }
  for (; b < nb; ++b) s0 += p[(long)b * 64];
  double v = 0.0;
  if (i < n) v = (double)r[i] - (double)noise[i] * (double)alpha[i] - ((s0 + s1) + (s2 + s3));
  rho64[i] = v;
  rho32[i] = (TS)v;
}

// ------------------------------------------------------------------------------------------
#define DGP_DISPATCH_MODEL(model, d, CALL)                   \
  switch (model) {                                           \
    case DGP_MODEL_LOADEST:                                  \
      switch (d) {                                           \
        case 2: { using M = Loadest<T, 2>; CALL; } break;    \
        case 3: { using M = Loadest<T, 3>; CALL; } break;    \
        case 4: { using M = Loadest<T, 4>; CALL; } break;    \
        case 5: { using M = Loadest<T, 5>; CALL; } break;    \
        case 6: { using M = Loadest<T, 6>; CALL; } break;    \
        default: return -2;                                  \
      }                                                      \
      break;                                                 \
    case DGP_MODEL_RATING:                                   \
      if (d != 2) return -2;                                 \
      { using M = Rating<T>; CALL; }                         \
      break;                                                 \
    default:                                                 \
      g_comp_cur = composite_get(model);                     \
      if (!g_comp_cur || g_comp_cur->d != d) return -2;      \
      switch (d) {                                           \
        case 1: { using M = Composite<T, 1>; CALL; } break;  \
        case 2: { using M = Composite<T, 2>; CALL; } break;  \
        case 3: { using M = Composite<T, 3>; CALL; } break;  \
        case 4: { using M = Composite<T, 4>; CALL; } break;  \
        case 5: { using M = Composite<T, 5>; CALL; } break;  \
        case 6: { using M = Composite<T, 6>; CALL; } break;  \
        default: return -2;                                  \
      }                                                      \
  }

template <typename T>
int pack_x(const T* X, int n, int d, long N, T* Xt, hipStream_t s, Batch bt) {
  pack_x_kernel<T><<<dim3((unsigned)((N + 255) / 256), 1, (unsigned)bt.B), dim3(256), 0, s>>>(X, n, d, N, Xt, bt.ws, bt.ns);
  return (int)hipGetLastError();
}

template <typename T>
int gram_sym(int model, int d, const T* Xt, long N, int n, const double* theta, const T* noise, T* K, hipStream_t s,
             Batch bt, void* pre_scratch, void* pre_staging, long k_stride, long noise_stride, bool pre_ready) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  const long nb = N / 64;
  const unsigned grid = (unsigned)(nb * (nb + 1) / 2);
  const long ks = k_stride >= 0 ? k_stride : bt.ws, nstr = noise_stride >= 0 ? noise_stride : (long)n;
  DGP_DISPATCH_MODEL(model, d, (gram_sym_kernel<T, M><<<dim3(grid, 1, (unsigned)bt.B), dim3(256), 0, s>>>(
                                   Xt, N, n, prepare_batch<M>(theta, nt, bt.B, pre_scratch, !pre_ready, s, pre_staging), noise, K, bt.ws, bt.ns,
                                   ks, nstr)));
  return (int)hipGetLastError();
}

template <typename T>
int gram_cross(int model, int d, const T* Xt, long N, int n, const T* Xst, long Mp, int m, const double* theta,
               T* Ks, hipStream_t s, Batch bt, long wbs, void* pre_scratch, void* pre_staging) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  dim3 grid((unsigned)(Mp / 64), (unsigned)(N / 64), (unsigned)bt.B);
  DGP_DISPATCH_MODEL(model, d, (gram_cross_kernel<T, M><<<grid, dim3(256), 0, s>>>(
                                   Xt, N, n, Xst, Mp, m, prepare_batch<M>(theta, nt, bt.B, pre_scratch, true, s, pre_staging), Ks, bt.ws,
                                   wbs, bt.ns)));
  return (int)hipGetLastError();
}

// pre_ready: the hyperparameters of a batch of more than 8 are already in pre_scratch (gram_cross of the same call)
template <typename T>
int gram_diag(int model, int d, const T* Xst, long Mp, int m, const double* theta, T* kss, hipStream_t s, Batch bt, long wbs,
              void* pre_scratch) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  dim3 grid((unsigned)((Mp + 255) / 256), 1, (unsigned)bt.B);
  DGP_DISPATCH_MODEL(model, d, (gram_diag_kernel<T, M><<<grid, dim3(256), 0, s>>>(
                                   Xst, Mp, m, prepare_batch<M>(theta, nt, bt.B, pre_scratch, false, s), kss, wbs)));
  return (int)hipGetLastError();
}

long gram_grad_partials(long N) {
  const long nb = N / 64;
  return nb * (nb + 1) / 2 * DGP_MAX_THETA;
}

template <typename T>
int gram_grad(int model, int d, const T* Xt, long N, int n, const double* theta, const T* S, const T* alpha,
              T* partials, T* dtheta, hipStream_t s, Batch bt, long dtheta_stride, void* pre_scratch, bool pre_ready,
              void* pre_staging) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  const long nb = N / 64;
  const long nblk = nb * (nb + 1) / 2;
  DGP_DISPATCH_MODEL(model, d,
                     (gram_grad_kernel<T, M, 0><<<dim3((unsigned)nblk, 1, (unsigned)bt.B), dim3(256), 0, s>>>(
                         Xt, N, n, prepare_batch<M>(theta, nt, bt.B, pre_scratch, !pre_ready, s, pre_staging), S, alpha, nullptr,
                         partials, bt.ws, bt.ns, 0)));
  grad_reduce_kernel<T><<<dim3((unsigned)nt, 1, (unsigned)bt.B), dim3(256), 0, s>>>(partials, nblk, nt, dtheta, 0, bt.ws,
                                                                                 dtheta_stride);
  return (int)hipGetLastError();
}

// dtheta = -b^T dK a (symmetric part) + sum_ij a_i w_j dK*_ij   -- the two kernel-gradient terms of the mean VJP.
// Batched: beta, partials and Xst live in the caller's work area (site stride wbs), wts is [B][m], dtheta [B][dstride].
template <typename T>
int mean_vjp_grad(int model, int d, const T* Xt, long N, int n, const T* Xst, long Mp, int m, const double* theta,
                  const T* alpha, const T* beta, const T* wts, T* partials, T* dtheta, hipStream_t s, Batch bt, long wbs,
                  long dstride, void* pre_scratch, void* pre_staging) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  const long nb = N / 64;
  const long nblk = nb * (nb + 1) / 2;
  const unsigned Bz = (unsigned)bt.B;
  DGP_DISPATCH_MODEL(model, d,
                     (gram_grad_kernel<T, M, 1><<<dim3((unsigned)nblk, 1, Bz), dim3(256), 0, s>>>(
                         Xt, N, n, prepare_batch<M>(theta, nt, bt.B, pre_scratch, true, s, pre_staging), nullptr, alpha, beta, partials,
                         bt.ws, bt.ns, wbs)));
  grad_reduce_kernel<T><<<dim3((unsigned)nt, 1, Bz), dim3(256), 0, s>>>(partials, nblk, nt, dtheta, 0, wbs, dstride);
  dim3 grid((unsigned)(Mp / 64), (unsigned)(N / 64), Bz);
  DGP_DISPATCH_MODEL(model, d,
                     (gram_cross_grad_kernel<T, M><<<grid, dim3(256), 0, s>>>(
                         Xt, N, n, Xst, Mp, m, prepare_batch<M>(theta, nt, bt.B, pre_scratch, false, s), alpha, wts, partials, bt.ws, wbs,
                         bt.ns)));
  grad_reduce_kernel<T><<<dim3((unsigned)nt, 1, Bz), dim3(256), 0, s>>>(partials, (long)grid.x * grid.y, nt, dtheta, 1, wbs, dstride);
  return (int)hipGetLastError();
}

template <typename T>
int gram_slab(int model, int d, const T* Xt, long N, int n, const double* theta, const T* noise, T* Aslab, long Cl,
              SlabMap sm, hipStream_t s) {
  if (model_ntheta(model, d) < 0) return -2;
  dim3 grid((unsigned)(N / 64), (unsigned)(Cl / 64));
  DGP_DISPATCH_MODEL(model, d, (gram_slab_kernel<T, M><<<grid, dim3(256), 0, s>>>(Xt, N, n, M::prepare(theta), noise, Aslab, Cl, sm)));
  return (int)hipGetLastError();
}

long gram_grad_slab_partials(long N, long Cl) { return (N / 64) * (Cl / 64) * DGP_MAX_THETA; }

template <typename T>
int gram_grad_slab(int model, int d, const T* Xt, long N, int n, const double* theta, const T* Sslab, long Cl, SlabMap sm,
                   const T* alpha, T* partials, T* dtheta, T* dnoise_part, hipStream_t s) {
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  dim3 grid((unsigned)(N / 64), (unsigned)(Cl / 64));
  DGP_DISPATCH_MODEL(model, d, (gram_grad_slab_kernel<T, M><<<grid, dim3(256), 0, s>>>(Xt, N, n, M::prepare(theta), Sslab, Cl, sm,
                                                                                   alpha, partials, dnoise_part)));
  grad_reduce_kernel<T><<<dim3((unsigned)nt), dim3(256), 0, s>>>(partials, (long)grid.x * grid.y, nt, dtheta, 0, 0, 0);
  return (int)hipGetLastError();
}

template <typename T>
int gemv_rows(const T* Ks, long N, long Mp, int m, const T* w, T* out, hipStream_t s, int B, long wbs) {
  gemv_rows_kernel<T><<<dim3((unsigned)((N + 3) / 4), 1, (unsigned)B), 256, 0, s>>>(Ks, N, Mp, m, w, out, wbs);
  return (int)hipGetLastError();
}

// scratch per site: part (nb^2 x 64 doubles) | rho64 (N doubles) | rho32 (N TS) | delta (N TS); dgp_api.hip carves it
// out of the plan's S buffer, which is free between the solves and lauum
size_t gram_residual_scratch_bytes(long N) {
  const size_t nb = (size_t)(N / 64);
  return nb * nb * 64 * sizeof(double) + (size_t)N * sizeof(double) + 2 * (size_t)N * sizeof(float);
}

template <typename TS>
int gram_residual(int model, int d, const TS* Xt, long N, int n, const double* theta, const TS* noise, const TS* r,
                  const TS* alpha, double* part, double* rho64, TS* rho32, hipStream_t s, Batch bt, long ps, long rs,
                  void* pre_scratch, void* pre_staging) {
  using T = double;  // the evaluator's arithmetic type (DGP_DISPATCH_MODEL instantiates Model<T, ...>)
  const int nt = model_ntheta(model, d);
  if (nt < 0) return -2;
  const long nb = N / 64;
  const unsigned grid = (unsigned)(nb * (nb + 1) / 2);
  DGP_DISPATCH_MODEL(model, d, (gram_matvec_kernel<TS, M><<<dim3(grid, 1, (unsigned)bt.B), dim3(256), 0, s>>>(
                                   Xt, N, n, prepare_batch<M>(theta, nt, bt.B, pre_scratch, true, s, pre_staging), alpha, part, bt.ws, ps,
                                   bt.ns)));
  resid_reduce_kernel<TS><<<dim3((unsigned)((N + 255) / 256), 1, (unsigned)bt.B), dim3(256), 0, s>>>(part, N, n, r, noise, alpha, rho64,
                                                                                                 rho32, bt.ws, ps, rs, bt.ns);
  return (int)hipGetLastError();
}
template int gram_residual<float>(int, int, const float*, long, int, const double*, const float*, const float*, const float*,
                                  double*, double*, float*, hipStream_t, Batch, long, long, void*, void*);

#define DGP_INST(T)                                                                                              \
  template int pack_x<T>(const T*, int, int, long, T*, hipStream_t, Batch);                                      \
  template int gram_sym<T>(int, int, const T*, long, int, const double*, const T*, T*, hipStream_t, Batch, void*, void*, long, long, bool); \
  template int gram_cross<T>(int, int, const T*, long, int, const T*, long, int, const double*, T*, hipStream_t, Batch, long, void*, void*); \
  template int gram_diag<T>(int, int, const T*, long, int, const double*, T*, hipStream_t, Batch, long, void*);    \
  template int gram_grad<T>(int, int, const T*, long, int, const double*, const T*, const T*, T*, T*, hipStream_t, Batch, \
                            long, void*, bool, void*);                                                         \
  template int mean_vjp_grad<T>(int, int, const T*, long, int, const T*, long, int, const double*, const T*, const T*, \
                                const T*, T*, T*, hipStream_t, Batch, long, long, void*, void*);                    \
  template int gemv_rows<T>(const T*, long, long, int, const T*, T*, hipStream_t, int, long);                   \
  template int gram_slab<T>(int, int, const T*, long, int, const double*, const T*, T*, long, SlabMap, hipStream_t); \
  template int gram_grad_slab<T>(int, int, const T*, long, int, const double*, const T*, long, SlabMap, const T*, T*, T*, T*, \
                                 hipStream_t);
DGP_INST(double)
DGP_INST(float)

}  // namespace dgp
