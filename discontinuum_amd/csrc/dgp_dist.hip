// dgp_dist.hip -- ONE exact-GP matrix factored, inverted and differentiated across several GPUs (BASELINE config 5,
// SURVEY.md section 8e; nothing in the reference corresponds to this: its engines hold one matrix on one device).
//
// Layout: 1-D block-cyclic by COLUMN GROUPS of W 128-wide panels (GW = 128 W columns), group g on rank g % world.  A
// rank keeps only its own groups, side by side, in three column slabs of N x Cl elements (Cl = its groups x GW; row
// stride Cl):
//     Aslab   K^ then L               (rows from each column's diagonal block down)
//     Tslab   L^-1                    (same region)
//     Sslab   first the running sums X of the inverse's forward substitution (below the diagonal), in the end
//             K^^-1 for the rows from the top down to each column's diagonal block (the transposed half)
// plus two panel buffers for the payloads in flight.  Nothing else of the matrix exists on a rank: n = 65536 fp32 on
// 8 ranks is 3 x 2.1 GB + 2 x 0.13 GB per GPU.  The host side (discontinuum_amd/dist_chol.py) moves the payloads with
// torch.distributed broadcasts (RCCL over xGMI) and the O(n) vectors with all-reduces; everything below is local.
//
// Pass 1, group g (owner g % world), one broadcast of (N - c0) x GW + GW x GW elements:
//     owner      panel chain of the group's columns (the single-GPU diag / trsm / column-update kernels through a base
//                pointer into the slab), inverse of the GW x GW diagonal block, pack -> payload P = [L(c0:, G) ; L_GG^-1]
//     every rank A[i, j] -= L[i, G] L[j, G]^T         for its own block columns j right of the group      (MFMA, K = GW)
//                T[G, H]  = -L_GG^-1 X[G, H]           for its own groups H < g                            (MFMA)
//                X[i, H] +=  L[i, G] T[G, H]           for rows i below the group, own groups H <= g       (MFMA, K = GW)
//     so that L^-1 is built by the SAME broadcasts as L: after the last group every rank holds its columns of both.
// Pass 2, group J, one broadcast of (N - c0) x GW elements of T:  S[J, I] = T[:, J]^T T[:, I] for the rank's own block
//     columns I >= J (k-range from I's diagonal block down: the heavy early steps have the most tiles).
// Flops per rank: N^3 / (3 world) each for L, L^-1 and K^^-1 -- the single-GPU fit step's N^3, divided by the ranks.
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "../../include/dgp_hip.h"
#include "dgp_common.h"
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"
#include "dgp_internal.h"

using namespace dgp;

namespace dgp {
static constexpr int NB = DGP_TILE;

// P[r][j] = src[(r0 + r) * ld + c0 + j], r < rows, j < width (width % 4 == 0); `lower_blocks`: 128-blocks above the block
// diagonal of a square source are written as zeros (they were never computed)
template <typename T>
__global__ __launch_bounds__(256) void pack_kernel(const T* __restrict__ src, long ld, long r0, long c0, long rows, int width,
                                                   T* __restrict__ dst, int lower_blocks) {
  constexpr int VN = Vec16<T>::N;
  using vec_t = typename Vec16<T>::type;
  const long per_row = width / VN;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= rows * per_row) return;
  const long r = idx / per_row, j = (idx % per_row) * VN;
  vec_t v;
  if (lower_blocks && j / NB > r / NB) {
#pragma unroll
    for (int e = 0; e < VN; ++e) v[e] = T(0);
  } else {
    v = *reinterpret_cast<const vec_t*>(src + (r0 + r) * ld + c0 + j);
  }
  *reinterpret_cast<vec_t*>(dst + r * width + j) = v;
}

// A[i, j] -= P[i, :] P[j, :]^T for the rank's block columns right of the group (pass 1).  P row 0 = global row c0.
// One workgroup per LIVE tile: the rank's columns right of the group are enumerated group by group (W columns with
// consecutive global indices bj0 + h, nbk - bj0 - h tiles each, from the diagonal down), so no workgroup is launched for
// the half of a (rows x columns) rectangle that lies above the diagonal (n = 65536 fp32, one rank: 114.7 -> 116.9 TFLOP/s).
struct SlabTiles {
  int lg0, ngl, W, world, rank, nbk;
  __host__ __device__ int group_tiles(int lg) const {  // tiles of local group lg (0 if it lies beyond the matrix)
    const int bj0 = (lg * world + rank) * W;      // N is a multiple of the group width: a group is whole or absent
    return bj0 < nbk ? W * (nbk - bj0) - W * (W - 1) / 2 : 0;
  }
  __host__ __device__ long total() const {
    long t = 0;
    for (int lg = lg0; lg < ngl; ++lg) t += group_tiles(lg);
    return t;
  }
  __device__ bool decode(long t, int& lb, int& bi, int& bj) const {
    for (int lg = lg0; lg < ngl; ++lg) {
      const int c = group_tiles(lg);
      if (t >= c) {
        t -= c;
        continue;
      }
      const int bj0 = (lg * world + rank) * W;
      for (int h = 0; h < W; ++h) {
        const int ch = nbk - bj0 - h;
        if (t < ch) {
          lb = lg * W + h;
          bj = bj0 + h;
          bi = bj + (int)t;
          return true;
        }
        t -= ch;
      }
    }
    return false;
  }
};
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, true, true>::OCC)) void slab_syrk_kernel(T* __restrict__ Aslab, long Cl, const T* __restrict__ P, int GW,
                                                           long c0, SlabTiles st, int cb, int ce) {
  using K = TileCore<T, true, true>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  int lb, bi, bj;
  // plain order (a column's tiles follow each other, the hardware deals them round-robin over the XCDs, which then all read
  // the same column operand): the XCD-contiguous remap of dgp_common.h gives every XCD its own column region and measured
  // 84 instead of 117 TFLOP/s here; a row-major order measured the same 117
  if (!st.decode((long)blockIdx.x, lb, bi, bj)) return;
  if (bj < cb || bj >= ce) return;  // the lookahead split: this launch covers block columns [cb, ce) only
  typename G::acc_t acc[G::MI][G::NI];
  T* C = Aslab + (long)bi * NB * Cl + (long)lb * NB;
  typename G::acc_t keep[G::MI][G::NI];
  trailing_begin<T, G, K::DMA>(acc, keep, C, Cl);
  K::run(P + ((long)bi * NB - c0) * GW, GW, P + ((long)bj * NB - c0) * GW, GW, GW / 16, smem, acc);
  trailing_end<T, G, K::DMA>(acc, keep, C, Cl);
}

// T[G, H] = -Linv X[G, H] for the rank's groups H < g: row block i of the group needs X's row blocks <= i (Linv is
// block lower triangular).  X is read from Sslab, T written to Tslab: no in-place hazard.
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, true, false>::OCC)) void slab_tconv_kernel(const T* __restrict__ Linv, int GW, const T* __restrict__ Sslab,
                                                            T* __restrict__ Tslab, long Cl, long c0) {
  using K = TileCore<T, true, false>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int i = (int)blockIdx.x, lb = (int)blockIdx.y;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  K::run(Linv + (long)i * NB * GW, GW, Sslab + c0 * Cl + (long)lb * NB, Cl, (i + 1) * (NB / 16), smem, acc);
  T* out = Tslab + (c0 + (long)i * NB) * Cl + (long)lb * NB;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * Cl + c] = -v; });
}

// X[i, lb] (+)= P[i, :] T[G, lb] for the rows below the group.  Columns of the group itself (the owner's local blocks
// own0 .. own0 + W - 1) start their sums here: column h of the group only has T[G, h] from its own diagonal block down.
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, true, false>::OCC)) void slab_xacc_kernel(const T* __restrict__ P, int GW, long c0, const T* __restrict__ Tslab,
                                                           T* __restrict__ Sslab, long Cl, int own0, int W) {
  using K = TileCore<T, true, false>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int bi = (int)((c0 + GW) / NB) + (int)blockIdx.x, lb = (int)blockIdx.y;
  const int h = lb - own0;
  const bool fresh = own0 >= 0 && h >= 0 && h < W;
  const long koff = fresh ? (long)h * NB : 0;
  typename G::acc_t acc[G::MI][G::NI];
  T* X = Sslab + (long)bi * NB * Cl + (long)lb * NB;
  // fp32 sums every pass from zero and joins the running sum once (dgp_gemm.h::trailing_begin: K roundings at the
  // magnitude of the running sum would otherwise swallow the small products)
  const bool from_zero = fresh || sizeof(T) == 4;
  if (from_zero) G::zero(acc);
  else G::foreach (acc, [&](int r, int c, T& v) { v = X[(long)r * Cl + c]; });
  K::run(P + ((long)bi * NB - c0) * GW + koff, GW, Tslab + (c0 + koff) * Cl + (long)lb * NB, Cl, (int)((GW - koff) / 16), smem, acc);
  if (from_zero && !fresh) G::foreach (acc, [&](int r, int c, T& v) { X[(long)r * Cl + c] += v; });
  else G::foreach (acc, [&](int r, int c, T& v) { X[(long)r * Cl + c] = v; });
}

// Pass 2: S[bj, bi] = sum_{c >= bi} T[c, bj]^T T[c, bi] for the rank's block columns bi >= bj, bj in the broadcast group
// (Q row 0 = global row c0).  BT = 64 when the step has too few 128-tiles to fill the GPU.
template <typename T, int BT>
__global__ __launch_bounds__(256, (TileCore<T, false, false, BT, BT>::OCC)) void slab_ttt_kernel(const T* __restrict__ Q, int GW, long c0, const T* __restrict__ Tslab,
                                                          T* __restrict__ Sslab, long Cl, long N, SlabMap sm, int lb0) {
  using K = TileCore<T, false, false, BT, BT>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  constexpr int PER = NB / BT;  // sub-tiles per 128-block and dimension
  const int jq = (int)blockIdx.x, lq = (int)blockIdx.y;  // in units of BT
  const int lb = lb0 + lq / PER, bi = sm.gblock(lb);
  const long col_j = c0 + (long)jq * BT;                                   // global column of the Q operand = row of S
  const long col_i = (long)bi * NB + (long)(lq % PER) * BT;                // global column of the T operand
  if ((long)bi * NB >= N || col_i + BT <= col_j) return;                   // strictly below the diagonal: the other half
  const long k0 = (long)bi * NB;                                           // T[:, bi] starts at its diagonal block
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  // k from the last row up to the diagonal block: small-to-large, like the single-GPU lauum_kernel (dgp_gemm.h REV)
  K::template run<true>(Q + (k0 - c0) * GW + (long)jq * BT, GW, Tslab + k0 * Cl + (long)lb * NB + (long)(lq % PER) * BT, Cl,
                        (int)((N - k0) / 16), smem, acc);
  T* out = Sslab + col_j * Cl + (long)lb * NB + (long)(lq % PER) * BT;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * Cl + c] = v; });
}

// z_part[i] = sum over the rank's columns c (block(c) <= block(i)) of T[i, c] r[c]: one wave per row
template <typename T>
__global__ __launch_bounds__(256) void slab_gemv_n_kernel(const T* __restrict__ Tslab, long Cl, long N, SlabMap sm,
                                                          const T* __restrict__ r, int n, T* __restrict__ z) {
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (i >= N) return;
  const int bi = (int)(i / NB);
  const T* row = Tslab + i * Cl;
  T acc = T(0);
  for (long lb = 0; lb * NB < Cl; ++lb) {
    const int gb = sm.gblock((int)lb);
    if (gb > bi) break;  // local blocks ascend with the global ones
    for (int j = lane; j < NB; j += 64) {
      const long c = (long)gb * NB + j;
      acc += row[lb * NB + j] * (c < n ? r[c] : T(0));
    }
  }
  acc = wave_sum(acc);
  if (lane == 0) z[i] = acc;
}

// partial[chunk][lc] = sum_{i in chunk, i >= the column's diagonal block} T[i, lc] z[i]
#define DGP_SLAB_CHUNK 512
template <typename T>
__global__ __launch_bounds__(256) void slab_gemv_t_kernel(const T* __restrict__ Tslab, long Cl, long N, SlabMap sm,
                                                          const T* __restrict__ z, T* __restrict__ partial) {
  __shared__ T red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long lc = (long)blockIdx.x * 64 + tx;
  const long gb = sm.gblock((int)(lc / NB));
  const long r0 = (long)blockIdx.y * DGP_SLAB_CHUNK, r1 = min(r0 + (long)DGP_SLAB_CHUNK, N);
  const long rstart = max(r0, gb * NB);
  T acc = T(0);
  if (gb * NB < N)
    for (long i = rstart + ty; i < r1; i += 4) acc += Tslab[i * Cl + lc] * z[i];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) partial[(long)blockIdx.y * Cl + lc] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}
template <typename T>
__global__ __launch_bounds__(256) void slab_gemv_t_reduce_kernel(const T* __restrict__ partial, long Cl, long N, int nchunks,
                                                                 SlabMap sm, T* __restrict__ alpha_part) {
  const long lc = (long)blockIdx.x * 256 + threadIdx.x;
  if (lc >= Cl) return;
  const long c = (long)sm.gblock((int)(lc / NB)) * NB + lc % NB;
  if (c >= N) return;
  T acc = T(0);
  for (int k = (int)(c / DGP_SLAB_CHUNK); k < nchunks; ++k) acc += partial[(long)k * Cl + lc];  // fixed order
  alpha_part[c] = acc;
}

template <typename T>
__global__ void dist_reset_kernel(T* scal, int* info, int ninfo) {
  if (threadIdx.x < 16) scal[threadIdx.x] = T(0);
  for (int i = threadIdx.x; i < ninfo; i += 64) info[i] = 0;
}
template <typename T>
__global__ void dist_status_kernel(const T* scal, const int* info, T* stat) {
  stat[0] = sizeof(T) == 4 ? (T) * reinterpret_cast<const double*>(scal + 2) : scal[0];  // fp32: the double slot (dgp_diag.h)
  stat[1] = (T)info[0];
}

}  // namespace dgp

// ---------------------------------------------------------------------------------------------
static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int wrap(int rc, const char* where) {
  if (rc > 0) {
    snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString((hipError_t)rc));
    return rc;
  }
  if (rc < 0) return fail(rc, where);
  return 0;
}
extern "C" const char* dgp_dist_last_error(void) { return g_err; }

struct dgp_dist {
  int model, dtype, d, ntheta, rank, world, W;
  int64_t n;
  long N, GW, Cl;
  int nbk, ng, ngl;
  size_t elem;
  char* ws;
  void *Xt, *A, *Tm, *S, *z, *part, *gpart, *scal;
  int* info;
  int have_inputs, stage;  // stage: 0 nothing, 1 gram built, 2 factor + inverse complete, 3 K^^-1 complete
  SlabMap sm() const { return SlabMap{W, world, rank}; }
  int owned_below(int g) const { return g <= rank ? 0 : (g - rank + world - 1) / world; }  // own groups with index < g
};

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }
struct DistLayout {
  size_t Xt, A, Tm, S, z, part, gpart, scal, info, total;
};
static DistLayout dist_layout(const dgp_dist* p) {
  DistLayout L;
  const size_t e = p->elem, N = (size_t)p->N, Cl = (size_t)p->Cl;
  const size_t nchunks = (N + DGP_SLAB_CHUNK - 1) / DGP_SLAB_CHUNK;
  size_t o = 0;
  L.Xt = o; o += align_up(e * N * p->d);
  L.A = o; o += align_up(e * N * Cl);
  L.Tm = o; o += align_up(e * N * Cl);
  L.S = o; o += align_up(e * N * Cl);
  L.z = o; o += align_up(e * N);
  L.part = o; o += align_up(e * nchunks * Cl);
  L.gpart = o; o += align_up(e * (size_t)gram_grad_slab_partials((long)N, (long)Cl));
  L.scal = o; o += align_up(e * 16);
  L.info = o; o += align_up(sizeof(int) * POTRF_INFO_INTS);
  L.total = o;
  return L;
}

#define DIST_CHECK(p)                                 \
  if (!(p)) return fail(DGP_E_ARG, "null handle");    \
  if (!(p)->ws) return fail(DGP_E_WORKSPACE, "no workspace: call dgp_dist_set_workspace")
#define DIST_GROUP(p, g) \
  if ((g) < 0 || (g) >= (p)->ng) return fail(DGP_E_ARG, "group index out of range")
#define BY_DTYPE(p, C64, C32) ((p)->dtype == DGP_F64 ? (C64) : (C32))

template <typename T>
static int dist_factor(dgp_dist* p, int g, void* panel, hipStream_t s) {
  const long c0 = (long)g * p->GW, lc0 = (long)(g / p->world) * p->GW, Cl = p->Cl;
  // base pointers through which (row r, GLOBAL column c of this group) is at [r * Cl + c]
  T* Ag = (T*)p->A + (lc0 - c0);
  T* Tg = (T*)p->Tm + (lc0 - c0);
  T* Sg = (T*)p->S + (lc0 - c0);
  int rc = potrf_group<T>(Ag, Cl, p->nbk, Tg, (T*)p->scal, p->info, g * p->W, p->W, s);
  if (rc) return rc;
  // the inverse of the GW x GW diagonal block: the single-GPU level recursion on that block alone (scratch: the same
  // block of the S slab, not in use yet)
  const long off = c0 * Cl + c0;
  if (p->W > 1 && (rc = trtri<T>(Ag + off, nullptr, p->GW, Tg + off, Sg + off, s, Batch(), Cl))) return rc;
  const long rows = p->N - c0;
  T* P = (T*)panel;
  pack_kernel<T><<<(unsigned)((rows * (p->GW / Vec16<T>::N) + 255) / 256), 256, 0, s>>>((const T*)p->A, Cl, c0, lc0, rows,
                                                                                    (int)p->GW, P, 0);
  pack_kernel<T><<<(unsigned)((p->GW * (p->GW / Vec16<T>::N) + 255) / 256), 256, 0, s>>>((const T*)p->Tm, Cl, c0, lc0, p->GW,
                                                                                     (int)p->GW, P + rows * p->GW, 1);
  return (int)hipGetLastError();
}

template <typename T>
static int dist_update(dgp_dist* p, int g, const void* panel, int cb, int ce, hipStream_t s) {
  const long c0 = (long)g * p->GW;
  const int rows_below = p->nbk - (g + 1) * p->W;
  const int lg0 = p->owned_below(g + 1);  // first local group right of g
  if (rows_below <= 0 || lg0 >= p->ngl) return 0;
  const SlabTiles st{lg0, p->ngl, p->W, p->world, p->rank, p->nbk};
  const long tiles = st.total();
  if (tiles <= 0) return 0;
  slab_syrk_kernel<T><<<dim3((unsigned)tiles), 256, 0, s>>>((T*)p->A, p->Cl, (const T*)panel, (int)p->GW, c0, st, cb, ce);
  return (int)hipGetLastError();
}

template <typename T>
static int dist_invert(dgp_dist* p, int g, const void* panel, hipStream_t s) {
  const long c0 = (long)g * p->GW, rows = p->N - c0;
  const T* P = (const T*)panel;
  const T* Linv = P + rows * p->GW;
  const int nbelow = p->owned_below(g) * p->W;            // local blocks of the rank's groups H < g
  const bool owner = g % p->world == p->rank;
  if (nbelow > 0)
    slab_tconv_kernel<T><<<dim3((unsigned)p->W, (unsigned)nbelow), 256, 0, s>>>(Linv, (int)p->GW, (const T*)p->S, (T*)p->Tm, p->Cl, c0);
  const int rows_below = p->nbk - (g + 1) * p->W;
  const int ncols = nbelow + (owner ? p->W : 0);
  if (rows_below > 0 && ncols > 0)
    slab_xacc_kernel<T><<<dim3((unsigned)rows_below, (unsigned)ncols), 256, 0, s>>>(P, (int)p->GW, c0, (const T*)p->Tm, (T*)p->S,
                                                                                 p->Cl, owner ? nbelow : -1, p->W);
  return (int)hipGetLastError();
}

template <typename T>
static int dist_product(dgp_dist* p, int g, const void* panel, hipStream_t s) {
  const long c0 = (long)g * p->GW;
  const int lg0 = p->owned_below(g);  // first local group with index >= g
  const int nlb = (p->ngl - lg0) * p->W;
  if (nlb <= 0) return 0;
  if ((long)p->W * nlb >= 1024)
    slab_ttt_kernel<T, 128><<<dim3((unsigned)p->W, (unsigned)nlb), 256, 0, s>>>((const T*)panel, (int)p->GW, c0, (const T*)p->Tm,
                                                                             (T*)p->S, p->Cl, p->N, p->sm(), lg0 * p->W);
  else
    slab_ttt_kernel<T, 64><<<dim3((unsigned)(2 * p->W), (unsigned)(2 * nlb)), 256, 0, s>>>((const T*)panel, (int)p->GW, c0,
                                                                                        (const T*)p->Tm, (T*)p->S, p->Cl, p->N,
                                                                                        p->sm(), lg0 * p->W);
  return (int)hipGetLastError();
}

extern "C" {

int dgp_dist_create(int model, int dtype, int64_t n, int d, int rank, int world, int group_panels, dgp_dist** out) {
  if (!out || n <= 0 || n > (1 << 22)) return fail(DGP_E_ARG, "dgp_dist_create: bad n / null out");
  if (dtype != DGP_F64 && dtype != DGP_F32) return fail(DGP_E_ARG, "dgp_dist_create: dtype must be 0 (f64) or 1 (f32)");
  if (world < 1 || rank < 0 || rank >= world) return fail(DGP_E_ARG, "dgp_dist_create: bad rank / world");
  if (group_panels < 1 || group_panels > 8) return fail(DGP_E_ARG, "dgp_dist_create: group_panels must be 1..8");
  const int nt = model_ntheta(model, d);
  if (nt < 0) return fail(DGP_E_MODEL, "dgp_dist_create: unsupported (model, d)");
  dgp_dist* p = new (std::nothrow) dgp_dist();
  if (!p) return fail(DGP_E_ARG, "dgp_dist_create: out of host memory");
  memset(p, 0, sizeof(*p));
  p->model = model; p->dtype = dtype; p->d = d; p->ntheta = nt; p->rank = rank; p->world = world; p->W = group_panels;
  p->n = n;
  p->GW = (long)DGP_TILE_HOST * group_panels;
  p->N = round_up(n, p->GW);  // whole groups: the identity pad takes the rest
  p->nbk = (int)(p->N / DGP_TILE_HOST);
  p->ng = (int)(p->N / p->GW);
  p->ngl = (p->ng + world - 1) / world;
  p->Cl = (long)p->ngl * p->GW;
  p->elem = dtype == DGP_F64 ? 8 : 4;
  *out = p;
  return 0;
}
int dgp_dist_destroy(dgp_dist* p) {
  delete p;
  return 0;
}
int64_t dgp_dist_padded_n(const dgp_dist* p) { return p ? p->N : 0; }
int dgp_dist_groups(const dgp_dist* p) { return p ? p->ng : 0; }
int64_t dgp_dist_slab_columns(const dgp_dist* p) { return p ? p->Cl : 0; }
size_t dgp_dist_workspace_bytes(const dgp_dist* p) { return p ? dist_layout(p).total : 0; }
size_t dgp_dist_panel_elems(const dgp_dist* p, int group) {
  if (!p || group < 0 || group >= p->ng) return 0;
  return (size_t)(p->N - (long)group * p->GW) * p->GW + (size_t)p->GW * p->GW;
}

int dgp_dist_set_workspace(dgp_dist* p, void* dev_ptr, size_t bytes) {
  if (!p || !dev_ptr) return fail(DGP_E_ARG, "dgp_dist_set_workspace: null");
  if (((uintptr_t)dev_ptr & 255) != 0) return fail(DGP_E_ARG, "dgp_dist_set_workspace: pointer must be 256-byte aligned");
  const DistLayout L = dist_layout(p);
  if (bytes < L.total) return fail(DGP_E_WORKSPACE, "dgp_dist_set_workspace: workspace too small");
  p->ws = (char*)dev_ptr;
  p->Xt = p->ws + L.Xt; p->A = p->ws + L.A; p->Tm = p->ws + L.Tm; p->S = p->ws + L.S; p->z = p->ws + L.z;
  p->part = p->ws + L.part; p->gpart = p->ws + L.gpart; p->scal = p->ws + L.scal; p->info = (int*)(p->ws + L.info);
  p->have_inputs = 0;
  p->stage = 0;
  return 0;
}

int dgp_dist_set_inputs(dgp_dist* p, const void* X, void* stream) {
  DIST_CHECK(p);
  if (!X) return fail(DGP_E_ARG, "dgp_dist_set_inputs: null X");
  hipStream_t s = (hipStream_t)stream;
  const int rc = BY_DTYPE(p, pack_x<double>((const double*)X, (int)p->n, p->d, p->N, (double*)p->Xt, s),
                          pack_x<float>((const float*)X, (int)p->n, p->d, p->N, (float*)p->Xt, s));
  p->have_inputs = 1;
  p->stage = 0;
  return wrap(rc, "dgp_dist_set_inputs");
}

int dgp_dist_gram(dgp_dist* p, const double* theta, const void* noise, void* stream) {
  DIST_CHECK(p);
  if (!theta || !noise) return fail(DGP_E_ARG, "dgp_dist_gram: null argument");
  if (!p->have_inputs) return fail(DGP_E_STATE, "dgp_dist_gram: call dgp_dist_set_inputs first");
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (p->dtype == DGP_F64) {
    dist_reset_kernel<double><<<1, 64, 0, s>>>((double*)p->scal, p->info, POTRF_INFO_INTS);
    rc = gram_slab<double>(p->model, p->d, (const double*)p->Xt, p->N, (int)p->n, theta, (const double*)noise, (double*)p->A, p->Cl, p->sm(), s);
  } else {
    dist_reset_kernel<float><<<1, 64, 0, s>>>((float*)p->scal, p->info, POTRF_INFO_INTS);
    rc = gram_slab<float>(p->model, p->d, (const float*)p->Xt, p->N, (int)p->n, theta, (const float*)noise, (float*)p->A, p->Cl, p->sm(), s);
  }
  p->stage = 1;
  return wrap(rc, "dgp_dist_gram");
}

int dgp_dist_factor(dgp_dist* p, int group, void* panel, void* stream) {
  DIST_CHECK(p);
  DIST_GROUP(p, group);
  if (!panel) return fail(DGP_E_ARG, "dgp_dist_factor: null panel");
  if (group % p->world != p->rank) return fail(DGP_E_ARG, "dgp_dist_factor: this rank does not own the group");
  if (p->stage < 1) return fail(DGP_E_STATE, "dgp_dist_factor: call dgp_dist_gram first");
  hipStream_t s = (hipStream_t)stream;
  return wrap(BY_DTYPE(p, dist_factor<double>(p, group, panel, s), dist_factor<float>(p, group, panel, s)), "dgp_dist_factor");
}

int dgp_dist_update(dgp_dist* p, int group, const void* panel, int col_begin, int col_end, void* stream) {
  DIST_CHECK(p);
  DIST_GROUP(p, group);
  if (!panel) return fail(DGP_E_ARG, "dgp_dist_update: null panel");
  if (col_end <= 0) col_end = p->nbk;
  hipStream_t s = (hipStream_t)stream;
  return wrap(BY_DTYPE(p, dist_update<double>(p, group, panel, col_begin, col_end, s),
                       dist_update<float>(p, group, panel, col_begin, col_end, s)), "dgp_dist_update");
}

int dgp_dist_invert(dgp_dist* p, int group, const void* panel, void* stream) {
  DIST_CHECK(p);
  DIST_GROUP(p, group);
  if (!panel) return fail(DGP_E_ARG, "dgp_dist_invert: null panel");
  hipStream_t s = (hipStream_t)stream;
  const int rc = BY_DTYPE(p, dist_invert<double>(p, group, panel, s), dist_invert<float>(p, group, panel, s));
  if (!rc && group == p->ng - 1) p->stage = 2;
  return wrap(rc, "dgp_dist_invert");
}

int dgp_dist_status(dgp_dist* p, void* stat, void* stream) {
  DIST_CHECK(p);
  if (!stat) return fail(DGP_E_ARG, "dgp_dist_status: null");
  hipStream_t s = (hipStream_t)stream;
  if (p->dtype == DGP_F64) dist_status_kernel<double><<<1, 1, 0, s>>>((const double*)p->scal, p->info, (double*)stat);
  else dist_status_kernel<float><<<1, 1, 0, s>>>((const float*)p->scal, p->info, (float*)stat);
  return wrap((int)hipGetLastError(), "dgp_dist_status");
}

int dgp_dist_solve_partial(dgp_dist* p, const void* r, void* z_part, void* stream) {
  DIST_CHECK(p);
  if (!r || !z_part) return fail(DGP_E_ARG, "dgp_dist_solve_partial: null argument");
  if (p->stage < 2) return fail(DGP_E_STATE, "dgp_dist_solve_partial: the inverse factor is not complete");
  hipStream_t s = (hipStream_t)stream;
  const unsigned grid = (unsigned)((p->N + 3) / 4);
  if (p->dtype == DGP_F64)
    slab_gemv_n_kernel<double><<<grid, 256, 0, s>>>((const double*)p->Tm, p->Cl, p->N, p->sm(), (const double*)r, (int)p->n, (double*)z_part);
  else
    slab_gemv_n_kernel<float><<<grid, 256, 0, s>>>((const float*)p->Tm, p->Cl, p->N, p->sm(), (const float*)r, (int)p->n, (float*)z_part);
  return wrap((int)hipGetLastError(), "dgp_dist_solve_partial");
}

int dgp_dist_alpha_partial(dgp_dist* p, const void* z, void* alpha_part, void* stream) {
  DIST_CHECK(p);
  if (!z || !alpha_part) return fail(DGP_E_ARG, "dgp_dist_alpha_partial: null argument");
  if (p->stage < 2) return fail(DGP_E_STATE, "dgp_dist_alpha_partial: the inverse factor is not complete");
  hipStream_t s = (hipStream_t)stream;
  const int nchunks = (int)((p->N + DGP_SLAB_CHUNK - 1) / DGP_SLAB_CHUNK);
  hipError_t e = hipMemsetAsync(alpha_part, 0, p->elem * (size_t)p->N, s);
  if (e != hipSuccess) return wrap((int)e, "dgp_dist_alpha_partial");
  dim3 grid((unsigned)(p->Cl / 64), (unsigned)nchunks);
  if (p->dtype == DGP_F64) {
    slab_gemv_t_kernel<double><<<grid, 256, 0, s>>>((const double*)p->Tm, p->Cl, p->N, p->sm(), (const double*)z, (double*)p->part);
    slab_gemv_t_reduce_kernel<double><<<(unsigned)((p->Cl + 255) / 256), 256, 0, s>>>((const double*)p->part, p->Cl, p->N, nchunks, p->sm(), (double*)alpha_part);
  } else {
    slab_gemv_t_kernel<float><<<grid, 256, 0, s>>>((const float*)p->Tm, p->Cl, p->N, p->sm(), (const float*)z, (float*)p->part);
    slab_gemv_t_reduce_kernel<float><<<(unsigned)((p->Cl + 255) / 256), 256, 0, s>>>((const float*)p->part, p->Cl, p->N, nchunks, p->sm(), (float*)alpha_part);
  }
  return wrap((int)hipGetLastError(), "dgp_dist_alpha_partial");
}

// fp32 handles: the residual of the refinement step (dgp_api.hip::run_refine is the single-plan counterpart).  K^ is
// re-evaluated in double from the inputs, which every rank holds: each rank computes the WHOLE vector -- no exchange, the
// same bits everywhere, n^2 / 2 kernel evaluations against the n^3 / world of the step.  The tile sums (N / 64)^2 x 64
// doubles = N^2 / 8 bytes live in the K^^-1 slab, which nothing has written yet at stage 2.
int dgp_dist_residual(dgp_dist* p, const double* theta, const void* noise, const void* r, const void* alpha, double* rho64,
                      void* rho32, void* stream) {
  DIST_CHECK(p);
  if (!theta || !noise || !r || !alpha || !rho64 || !rho32) return fail(DGP_E_ARG, "dgp_dist_residual: null argument");
  if (p->dtype != DGP_F32) return fail(DGP_E_ARG, "dgp_dist_residual: fp32 handles only (fp64 needs no refinement)");
  if (p->stage != 2)
    return fail(DGP_E_STATE, "dgp_dist_residual: call between the factorisation and the inverse products (the K^^-1 slab is its scratch)");
  const size_t nb = (size_t)(p->N / 64), need = nb * nb * 64 * sizeof(double);
  if (need > p->elem * (size_t)p->N * (size_t)p->Cl)
    return fail(DGP_E_WORKSPACE, "dgp_dist_residual: the K^^-1 slab is smaller than the tile sums (more than 32 ranks)");
  const int rc = gram_residual<float>(p->model, p->d, (const float*)p->Xt, p->N, (int)p->n, theta, (const float*)noise, (const float*)r,
                                      (const float*)alpha, (double*)p->S, rho64, (float*)rho32, (hipStream_t)stream, Batch(), 0, 0,
                                      nullptr, nullptr);
  return wrap(rc, "dgp_dist_residual");
}

int dgp_dist_pack_inverse(dgp_dist* p, int group, void* panel, void* stream) {
  DIST_CHECK(p);
  DIST_GROUP(p, group);
  if (!panel) return fail(DGP_E_ARG, "dgp_dist_pack_inverse: null panel");
  if (group % p->world != p->rank) return fail(DGP_E_ARG, "dgp_dist_pack_inverse: this rank does not own the group");
  if (p->stage < 2) return fail(DGP_E_STATE, "dgp_dist_pack_inverse: the inverse factor is not complete");
  hipStream_t s = (hipStream_t)stream;
  const long c0 = (long)group * p->GW, lc0 = (long)(group / p->world) * p->GW, rows = p->N - c0;
  if (p->dtype == DGP_F64)
    pack_kernel<double><<<(unsigned)((rows * (p->GW / 2) + 255) / 256), 256, 0, s>>>((const double*)p->Tm, p->Cl, c0, lc0, rows, (int)p->GW, (double*)panel, 1);
  else
    pack_kernel<float><<<(unsigned)((rows * (p->GW / 4) + 255) / 256), 256, 0, s>>>((const float*)p->Tm, p->Cl, c0, lc0, rows, (int)p->GW, (float*)panel, 1);
  return wrap((int)hipGetLastError(), "dgp_dist_pack_inverse");
}

int dgp_dist_product(dgp_dist* p, int group, const void* panel, void* stream) {
  DIST_CHECK(p);
  DIST_GROUP(p, group);
  if (!panel) return fail(DGP_E_ARG, "dgp_dist_product: null panel");
  if (p->stage < 2) return fail(DGP_E_STATE, "dgp_dist_product: the inverse factor is not complete");
  hipStream_t s = (hipStream_t)stream;
  const int rc = BY_DTYPE(p, dist_product<double>(p, group, panel, s), dist_product<float>(p, group, panel, s));
  if (!rc && group == p->ng - 1) p->stage = 3;
  return wrap(rc, "dgp_dist_product");
}

int dgp_dist_grad_partial(dgp_dist* p, const double* theta, const void* alpha, void* dtheta_part, void* dnoise_part, void* stream) {
  DIST_CHECK(p);
  if (!theta || !alpha || !dtheta_part || !dnoise_part) return fail(DGP_E_ARG, "dgp_dist_grad_partial: null argument");
  if (p->stage < 3) return fail(DGP_E_STATE, "dgp_dist_grad_partial: K^^-1 is not complete (run every dgp_dist_product)");
  hipStream_t s = (hipStream_t)stream;
  hipError_t e = hipMemsetAsync(dnoise_part, 0, p->elem * (size_t)p->N, s);
  if (e != hipSuccess) return wrap((int)e, "dgp_dist_grad_partial");
  const int rc = BY_DTYPE(p,
      gram_grad_slab<double>(p->model, p->d, (const double*)p->Xt, p->N, (int)p->n, theta, (const double*)p->S, p->Cl, p->sm(),
                             (const double*)alpha, (double*)p->gpart, (double*)dtheta_part, (double*)dnoise_part, s),
      gram_grad_slab<float>(p->model, p->d, (const float*)p->Xt, p->N, (int)p->n, theta, (const float*)p->S, p->Cl, p->sm(),
                            (const float*)alpha, (float*)p->gpart, (float*)dtheta_part, (float*)dnoise_part, s));
  return wrap(rc, "dgp_dist_grad_partial");
}

int dgp_dist_slab(const dgp_dist* p, int which, void** dev_ptr) {
  if (!p || !dev_ptr || !p->ws) return fail(DGP_E_ARG, "dgp_dist_slab: null / no workspace");
  switch (which) {
    case DGP_BUF_A: *dev_ptr = p->A; break;
    case DGP_BUF_T: *dev_ptr = p->Tm; break;
    case DGP_BUF_S: *dev_ptr = p->S; break;
    default: return fail(DGP_E_ARG, "dgp_dist_slab: unknown buffer");
  }
  return 0;
}

}  // extern "C"
