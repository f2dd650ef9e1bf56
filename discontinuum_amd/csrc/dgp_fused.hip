// dgp_fused.hip -- K^^-1 = L^-T L^-1 with the hyperparameter-gradient contraction in its epilogue (round 5).
//
// The backward pass of the reference's objective (src/discontinuum/engines/gpytorch.py:384) needs
//     dNLL/dtheta_p = 1/2 sum_ij (K^^-1 - alpha alpha^T)_ij dK_ij/dtheta_p .
// Until round 4 that was a kernel of its own behind lauum (dgp_gram.hip::gram_grad_kernel): it streams K^^-1 once from HBM
// and evaluates every sub-kernel's derivative per entry -- in fp64 it is VALU-bound (142 instructions per entry, 5.9 ms of
// the 281 ms headline step with every MFMA pipe idle).  Here each 128 x 128 tile of lauum_kernel, right after it has stored
// its tile of S, contracts that tile with dK/dtheta: vector-ALU work of ONE workgroup of a compute unit while the other two
// resident workgroups keep the MFMA pipe busy, and S is read back from the L2 it was just written to instead of from HBM.
//
// Why the tile is re-read and not taken from the accumulators: a wave holds its 64 x 64 fp64 sub-tile in 128 registers and the
// kernel may use 168 (three workgroups per CU, dgp_gemm_dma.h); the derivative expressions need ~100 on their own.  After the
// store the accumulators are dead and the contraction gets the whole register file of the wave; the re-read hits the XCD's
// L2 microseconds after the write (the tile is private to the workgroup: visibility is a fence + barrier).
//
// Roofline: MFMA (the k-loop is lauum_kernel's: N^3/3 flop per matrix); the contraction adds N(N+128)/2 x ~142 VALU
// instructions per matrix that run beside other workgroups' MFMAs.
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"
#include "dgp_gram_shared.h"
#include "dgp_internal.h"
#include "dgp_models.h"

namespace dgp {

static constexpr int NB = DGP_TILE;

template <typename T, typename M>
__global__ __launch_bounds__(256, 3) void lauum_grad_kernel(const T* __restrict__ Tm, T* __restrict__ S, long ld, int nbk, long bs,
                                                            const T* __restrict__ Xt, int n, const PreBatch<M> pb,
                                                            const T* __restrict__ alpha, T* __restrict__ partials,
                                                            const int* __restrict__ ns) {
  Tm = site(Tm, bs);
  S = site(S, bs);
  using K = TileCore<T, false, false, 128, 128, 1, true, true>;  // lauum_kernel's core: direct-to-LDS, interleaved groups
  static_assert(K::DMA && K::IL, "the fused kernel is the 128 x 128 direct-to-LDS tile");
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);  // ascending bi: the long-K tiles are dispatched first
  {
    typename G::acc_t acc[G::MI][G::NI];
    G::zero(acc);
    const T* base = Tm + (long)bi * NB * ld;
    T* out = S + (long)bi * NB * ld + (long)bj * NB;
    auto store = [&]() { K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; }); };
    // (k from the last row block up to the diagonal block, zero-work skipping in the last block: dgp_chol.hip::lauum_kernel)
    if (bi == bj) K::template run_tri<true, TRI_LOWER>(base + (long)bi * NB, ld, base + (long)bj * NB, ld, (nbk - bi) * (NB / 16), smem, acc, store);
    else K::template run_tri<true, TRI_ROW_LE>(base + (long)bi * NB, ld, base + (long)bj * NB, ld, (nbk - bi) * (NB / 16), smem, acc, store);
  }
  // ---- the tile is stored and the accumulators are dead: contract it with dK/dtheta.  The ring is free (every wave is past the
  // last chunk's barrier): it now holds the row / column strips of per-point features and of alpha.
  n = site_n(ns, n);
  const typename M::Pre& pre = pb.get();
  Xt = site(Xt, bs);
  alpha = site(alpha, bs);
  partials = site(partials, bs);
  static_assert((2 * M::NF + 2) * 128 + 4 * M::NTHETA <= K::SMEM_ELEMS, "feature strips do not fit the ring");
  T(*sfi)[128] = reinterpret_cast<T(*)[128]>(smem);
  T(*sfj)[128] = reinterpret_cast<T(*)[128]>(smem + M::NF * 128);
  T* sai = smem + 2 * M::NF * 128;
  T* saj = sai + 128;
  T(*red)[M::NTHETA] = reinterpret_cast<T(*)[M::NTHETA]>(saj + 128);
  const int t = threadIdx.x;
  exp_table_init<T>();
  {
    const int row = t & 127;
    const long base = (long)(t < 128 ? bi : bj) * NB + row;
    T x[M::NX], f[M::NF];
#pragma unroll
    for (int c = 0; c < M::NX; ++c) x[c] = Xt[(long)c * ld + base];
    M::features(x, pre, f);
    T(*sf)[128] = t < 128 ? sfi : sfj;
#pragma unroll
    for (int c = 0; c < M::NF; ++c) sf[c][row] = f[c];
    (t < 128 ? sai : saj)[row] = alpha[base];
  }
  __threadfence();   // this workgroup's stores of the tile are complete and visible to its other waves (L1 invalidated)
  __syncthreads();
  const int ty = t >> 4, tx = t & 15;
  T acc[M::NTHETA];
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) acc[p] = T(0);
  // four 64 x 64 quarters, each walked like gram_grad_kernel walks its tile (one entry at a time: the derivative expressions
  // are register-hungry); a diagonal tile's upper-right quarter lies above the diagonal: nothing to do
#pragma unroll 1
  for (int q = 0; q < 4; ++q) {
    const int qi = q >> 1, qj = q & 1;
    if (bi == bj && qj > qi) continue;
#pragma unroll 1
    for (int a = 0; a < 4; ++a) {
      const int ri = qi * 64 + ty * 4 + a;
      const long gi = (long)bi * NB + ri;
      T fi[M::NF];
#pragma unroll
      for (int c = 0; c < M::NF; ++c) fi[c] = sfi[c][ri];
      const T ai = sai[ri];
      T sv[4];
      load4<T>(S + gi * ld + (long)bj * NB + qj * 64 + tx * 4, sv);
#pragma unroll 1
      for (int b = 0; b < 4; ++b) {
        const int cj = qj * 64 + tx * 4 + b;
        const long gj = (long)bj * NB + cj;
        T fj[M::NF];
#pragma unroll
        for (int c = 0; c < M::NF; ++c) fj[c] = sfj[c][cj];
        const T svb = b == 0 ? sv[0] : (b == 1 ? sv[1] : (b == 2 ? sv[2] : sv[3]));
        // lower triangle counted once with weight 1 (= 1/2 * 2), diagonal with 1/2, pad with 0 -- a SELECT, so whatever a
        // diagonal tile holds above its diagonal (unspecified: TRI_LOWER) never enters
        T w = svb - ai * saj[cj];
        w = (gj > gi || gi >= n) ? T(0) : (gj == gi ? T(0.5) * w : w);
        (void)M::template pair<true>(fi, fj, pre, w, acc);
      }
    }
  }
  M::finalize(acc, pre);
  const int lane = t & 63, wv = t >> 6;
#pragma unroll
  for (int p = 0; p < M::NTHETA; ++p) {
    T v = wave_sum(acc[p]);
    if (lane == 0) red[wv][p] = v;
  }
  __syncthreads();
  if (t < M::NTHETA) partials[(long)blockIdx.x * DGP_MAX_THETA + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
}

#define DGP_DISPATCH_FUSED(model, d, CALL)                   \
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
      return -2;                                             \
  }

// Whether lauum_grad applies: the two fused covariance models (the interpreted composite evaluator keeps its own kernel),
// matrices whose K^^-1 runs in 128 x 128 tiles in the default order.
bool lauum_grad_applies(int model, long N, const Batch& bt) {
  if (model != DGP_MODEL_LOADEST && model != DGP_MODEL_RATING) return false;
  const int nbk = (int)(N / NB);
  const long tiles = (long)nbk * (nbk + 1) / 2;
  return tiles * bt.B > bt.tuning().lauum64_max_tiles && bt.tuning().lauum_super == 0;
}

// S = T^T T and dtheta = 1/2 sum (S - alpha alpha^T) dK/dtheta in one launch + the deterministic second reduction stage.
// alpha must be final (the solves run BEFORE this launch).  `partials`: gram_grad_partials(N) elements per site.
template <typename T>
int lauum_grad(int model, int d, const T* Tm, long N, T* S, const T* Xt, int n, const double* theta, const T* alpha, T* partials,
               T* dtheta, hipStream_t s, Batch bt, long dtheta_stride, void* pre_scratch, bool pre_ready, void* pre_staging) {
  const int nt = model_ntheta(model, d);
  if (nt < 0 || !lauum_grad_applies(model, N, bt)) return -2;
  const int nbk = (int)(N / NB);
  const long tiles = (long)nbk * (nbk + 1) / 2;
  DGP_DISPATCH_FUSED(model, d,
                     (lauum_grad_kernel<T, M><<<dim3((unsigned)tiles, 1, (unsigned)bt.B), dim3(256), 0, s>>>(
                         Tm, S, N, nbk, bt.ws, Xt, n, prepare_batch<M>(theta, nt, bt.B, pre_scratch, !pre_ready, s, pre_staging), alpha,
                         partials, bt.ns)));
  grad_reduce_kernel<T><<<dim3((unsigned)nt, 1, (unsigned)bt.B), dim3(256), 0, s>>>(partials, tiles, nt, dtheta, 0, bt.ws, dtheta_stride);
  return (int)hipGetLastError();
}

#define DGP_INSTANTIATE_FUSED(T)                                                                                               \
  template int lauum_grad<T>(int, int, const T*, long, T*, const T*, int, const double*, const T*, T*, T*, hipStream_t, Batch, \
                             long, void*, bool, void*);
DGP_INSTANTIATE_FUSED(double)
DGP_INSTANTIATE_FUSED(float)

}  // namespace dgp
