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
#include <stdlib.h>
#include <string.h>
#include <mutex>
#include "dgp_diag.h"
#include "dgp_gemm.h"
#include "dgp_gemm_dma.h"
#include "dgp_internal.h"

namespace dgp {

static constexpr int NB = DGP_TILE;
static constexpr int SMID_TABLE = 4096;  // covers the XCC | SE | CU id bits of __smid()
// k-tiles of register prefetch (TileGemm::run<PF>) for the kernels whose operands mostly miss L2
// (measured at n = 8192: fp64 lauum 2.99 -> 2.87 ms with 2; fp64 trtri gets SLOWER with 2 -- 256 VGPRs; fp32 has room)
template <typename T>
struct Prefetch {
  static constexpr bool F64 = sizeof(T) == 8;
  static constexpr int LAUUM = F64 ? 2 : 4, TRTRI = F64 ? 1 : 4, SYRK = F64 ? 1 : 2;
};

// ------------------------------------------------------------------------------------------
// In-kernel waits of the split panel chain (potrf_split): a kernel polls a progress word that ANOTHER kernel publishes.
// That is only safe while the producer gets dispatched beside the waiter -- true for the hardware queues of distinct
// streams, NOT for a tool that serialises dispatches in queue-ready order (rocprofv3 --pmc does: a waiter granted first
// would spin forever; counter passes on single-site plans must run with DGP_SPLIT_CHAIN=0, scripts/collect_profiles.sh
// exports it), not after a failed producer launch (checked on the host before the waiter is enqueued), and not when the
// waiters alone can fill the machine: ONE site's trsm / crit launches are at most a few hundred workgroups; a 32-site
// batch's are thousands, and a GPU full of polling workgroups leaves the producer no slot (measured: every wait ran into its
// bound, profiles/r04_experiments_batched_split.txt) -- batched plans never poll (split_applies: B == 1).  So every wait
// is BOUNDED: after CHAIN_WAIT_TICKS of the 100 MHz wall clock (2 s -- a whole n = 8192 fit takes 12 ms) the waiter
// writes DGP_INFO_CHAIN_TIMEOUT into info[0] (unless a pivot index is already there), sets info[CHAIN_ABORT] and its workgroup
// returns without touching the matrix; every later waiter sees the abort word and returns at once, so the chain drains in about one budget, the result row carries info < 0 and a NaN
// NLL (finish_kernel) instead of the device hanging.
#define DGP_INFO_CHAIN_TIMEOUT (-7)
#define CHAIN_WAIT_TICKS 200000000LL
__device__ __forceinline__ bool chain_reached(const int* word, int value) {
  return __hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= value;
}
// thread 0 polls, the workgroup follows; -> false: give up (timeout, or an earlier waiter's timeout)
__device__ __forceinline__ bool chain_wait(const int* w0, int v0, const int* w1, int v1, int* info) {
  __shared__ int chain_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    if (!(chain_reached(w0, v0) && (w1 == nullptr || chain_reached(w1, v1)))) {
      const long long t0 = wall_clock64();
      while (!(chain_reached(w0, v0) && (w1 == nullptr || chain_reached(w1, v1)))) {
        if (__hip_atomic_load(info + CHAIN_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {  // a wait ran out elsewhere
          ok = 0;
          break;
        }
        if (wall_clock64() - t0 > CHAIN_WAIT_TICKS) {
          // the code goes only into an EMPTY slot: a pivot index that diag(k) has already published (info > 0, what the
          // caller's jitter ladder / NaN policy keys on) survives; the abort word makes every later waiter leave at once
          atomicCAS(info, 0, DGP_INFO_CHAIN_TIMEOUT);
          __hip_atomic_store(info + CHAIN_ABORT, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          ok = 0;
          break;
        }
        __builtin_amdgcn_s_sleep(8);
      }
    }
    chain_ok = ok;
  }
  __syncthreads();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  return chain_ok != 0;
}

// A[i,k] <- A[i,k] * L_kk^-T  (L_kk^-1 lives in the diagonal block of Tinv).  This kernel sits on the
// sequential panel chain, so it uses 64-row tiles: twice the workgroups, half the per-workgroup latency.
template <typename T>
__global__ __launch_bounds__(256, 2) void trsm_kernel(T* __restrict__ A, const T* __restrict__ Tinv, long ld, int k,
                                                      long bs, int* __restrict__ wait_info = nullptr, int wait_value = 0) {
  // split panel chain: launched on the rest stream BEFORE diag(k) has finished (wait_info = the plan's info words)
  if (wait_info && !chain_wait(wait_info + CHAIN_DIAG_DONE, wait_value, nullptr, 0, wait_info)) return;
  A = site(A, bs);
  Tinv = site(Tinv, bs);
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
// How C joins the accumulators differs by precision: trailing_begin / trailing_end below.
//
// All tiles cost the same and the round cut assumes 512 workgroup slots (Tuning::syrk_slots), so a launch of t tiles runs in
// ceil(t / slots) rounds and its last round is on average half empty.  The first `nfull` tiles (whole rounds) are
// done as 128 x 128 tiles; the rest is cut into `split` pieces each (2: 64 x 128 halves, 4: 64 x 64 quarters),
// which fills the last round at a fraction of its time (SyrkShape picks the cheapest cut).
// POLITE (single-site plans): the waves of a 128 x 128 tile give their CU to the diagonal-block kernel while it runs there
// (dgp_common.h: yield_if_asked); `me` = cu_code() of this wave
struct YieldHook {
  const unsigned* word;
  unsigned me;
  const unsigned* copy;  // this wave's 64 words of LDS
  __device__ __forceinline__ void operator()(int) const {
    const unsigned seen = __builtin_amdgcn_readfirstlane(copy[0]);  // (the first round reads whatever LDS held: a false match costs one scalar load)
    yield_if_asked(word, me, seen);
    yield_refresh(word, (unsigned)(size_t)(dgp_lds_ptr)copy);
  }
};
#ifndef DGP_BULK_C_DEFAULT
#define DGP_BULK_C_DEFAULT 0
#endif
#ifndef DGP_COL_C_DEFAULT
#define DGP_COL_C_DEFAULT 0
#endif
// CMODE: how the tile of C moves (dgp_gemm.h::trailing_begin): bit 0 = non-temporal loads / stores (STREAM), bit 1 = 16-byte
// accesses with a lane-pair swap (WIDE, fp64) -- the bulk update's tiles
template <typename T, int BM, int BN, bool POLITE = false, int CMODE = 0, int RING = 3>
__device__ __forceinline__ void syrk_tile(T* __restrict__ A, long ld, int k, int nk, long row0, long col0,
                                          T* __restrict__ smem, const unsigned* yield_word = nullptr, unsigned me = 0,
                                          const unsigned* yield_copy = nullptr) {
  using K = TileCore<T, true, true, BM, BN, (BM == 128 && BN == 128) ? Prefetch<T>::SYRK : 1, true, false, RING>;
  using G = typename K::G;
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + row0 * ld + col0;
  typename G::acc_t keep[G::MI][G::NI];
  constexpr bool STREAM = (CMODE & 1) != 0, WIDE = (CMODE & 2) != 0;
  trailing_begin<T, G, K::DMA, STREAM, WIDE>(acc, keep, C, ld);
  if constexpr (POLITE)
    K::run_hooked(A + row0 * ld + (long)k * NB, ld, A + col0 * ld + (long)k * NB, ld, nk * (NB / 16), smem, acc, YieldHook{yield_word, me, yield_copy});
  else
    K::run(A + row0 * ld + (long)k * NB, ld, A + col0 * ld + (long)k * NB, ld, nk * (NB / 16), smem, acc);
  trailing_end<T, G, K::DMA, STREAM, WIDE>(acc, keep, C, ld);
}

// RING = 2 (experiment, DGP_BULK_RING=2 with the group panel solve): a 32 KB ring, so that a CU with three of these workgroups
// has 64 KB of LDS free and the diagonal-block kernel (94.5 KB, one workgroup per site, the only chain kernel that must run
// BESIDE a bulk launch) would fit as soon as ONE of them retires.  Measured: it does not change when that kernel is placed --
// the second diagonal block of a group still waits for the bulk launch to drain (2.9 ms at 64 x n = 4096 with either ring,
// profiles/r05_experiments_group_gemm.txt), and two bulk workgroups per CU (DGP_BULK_PAD_BATCH) do not either
template <typename T, bool POLITE = false, int CMODE = 0, int RING = 3>
__global__ __launch_bounds__(256, (TileCore<T, true, true>::OCC)) void syrk_kernel(T* __restrict__ A, long ld, int k, int nk, int jbeg, int nfull,
                                                                                   int split, long bs, int nt = 0, int super = 0,
                                                                                   const unsigned* yield_word = nullptr) {
  A = site(A, bs);
  __shared__ T smem[TileCore<T, true, true, 128, 128, 1, true, false, RING>::SMEM_ELEMS];
  static_assert(TileCore<T, true, true, 64, 128>::SMEM_ELEMS <= TileCore<T, true, true, 128, 128, 1, true, false, RING>::SMEM_ELEMS &&
                    TileCore<T, true, true, 64, 64>::SMEM_ELEMS <= TileCore<T, true, true, 128, 128, 1, true, false, RING>::SMEM_ELEMS,
                "the cut remainder's register-staged tiles must fit the ring's LDS");
  __shared__ unsigned yield_copy[POLITE ? 4 * 64 : 1];  // per wave: its copy of the yield word (YieldHook)
  const int b = (int)blockIdx.x;
  int bi, bj;
  if (b < nfull) {
    // all tiles cost the same: remap freely -- consecutive logical tiles on one XCD; `super` picks the logical order
    if (super > 0) super_decode(xcd_remap(b, nfull), nt, super, bi, bj);
    else tri_decode(xcd_remap(b, nfull), bi, bj);
    syrk_tile<T, 128, 128, POLITE, CMODE, RING>(A, ld, k, nk, (long)(bi + jbeg) * NB, (long)(bj + jbeg) * NB, smem, yield_word, POLITE ? cu_code() : 0u,
                                   yield_copy + (POLITE ? 64 * __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6) : 0));
    return;
  }
  const int sub = b - nfull;
  if (super > 0) super_decode(nfull + sub / split, nt, super, bi, bj);
  else tri_decode(nfull + sub / split, bi, bj);
  const long row0 = (long)(bi + jbeg) * NB, col0 = (long)(bj + jbeg) * NB;
  const int part = sub % split;
  if (split == 2) {
    syrk_tile<T, 64, 128, false, CMODE>(A, ld, k, nk, row0 + 64 * part, col0, smem);
  } else {
    if (bi == bj && part == 1) return;  // strictly upper quadrant of a diagonal tile
    syrk_tile<T, 64, 64, false, CMODE>(A, ld, k, nk, row0 + 64 * (part >> 1), col0 + 64 * (part & 1), smem);
  }
}

// The selectors' defaults (Tuning, dgp_internal.h).  syrk_slots: 512 measured best for both precisions -- with the
// direct-to-LDS core the GPU holds 768 of these workgroups at a time, and cutting the rounds at 768 changes nothing measurable.
const Tuning& default_tuning() {
  static const Tuning t = [] {
    Tuning v;
    v.lauum64_max_tiles = getenv("DGP_LAUUM64") ? atoi(getenv("DGP_LAUUM64")) : 1000;
    v.syrk_slots = getenv("DGP_SYRK_SLOTS") ? atoi(getenv("DGP_SYRK_SLOTS")) : 512;
    v.trtri_small = getenv("DGP_TRTRI_SMALL") ? atol(getenv("DGP_TRTRI_SMALL")) : 1024;
    v.syrk_super = getenv("DGP_SYRK_ORDER") ? atoi(getenv("DGP_SYRK_ORDER")) : 0;
    v.lauum_super = getenv("DGP_LAUUM_ORDER") ? atoi(getenv("DGP_LAUUM_ORDER")) : 0;
    v.chain_yield = getenv("DGP_CHAIN_YIELD") ? atoi(getenv("DGP_CHAIN_YIELD")) : 1;
    // off: measured on one box in alternating processes (scripts/env_ab.py, wall ms per step, 0 / 1): 32 x n = 8192 273.9 /
    // 275.5, 64 x n = 4096 79.5 / 80.1, one site n = 8192 11.86 / 11.86 -- the contraction's fp64 vector work costs the fused
    // launch what it cost alone (lauum 81.0 -> 85.9 ms for 5.8 ms of gram_grad): it does not hide under the other
    // workgroups' MFMAs
    v.fused_grad = getenv("DGP_FUSED_GRAD") ? atoi(getenv("DGP_FUSED_GRAD")) : 0;
    // off: built, parity-green (tests/test_gpu_stages.py with the option on) and measured neutral -- 64 x n = 4096 80.8 / 80.3 ms
    // per step, 32 x n = 8192 280.2 / 280.1, 128 x n = 2048 26.1 / 26.0 (scripts/env_ab.py, one box): the factorisation's wall
    // time is the SUM of its full-GPU kernels at their rates whichever way the chain is cut (EXPERIMENTS.md, round 5)
    v.group_gemm = getenv("DGP_GROUP_GEMM") ? atoi(getenv("DGP_GROUP_GEMM")) : 0;
    if (v.syrk_slots < 1) v.syrk_slots = 1;
    return v;
  }();
  return t;
}
// grid shape of one bulk launch: whole rounds of full tiles + the remainder cut into `split` pieces
struct SyrkShape {
  int nfull, split;
  unsigned grid;
  explicit SyrkShape(int ntiles, int slots = 512) {
    if (slots < 1) slots = 1;  // (slots / batch of a large batch)
    // relative cost of one round of 128x128 / 64x128 / 64x64 tiles (the small ones reach ~87 % of the big tile's rate)
    const double cost[5] = {0, 1.0, 0.55, 0, 0.30};
    const int whole = ntiles / slots * slots, rem = ntiles - whole;
    nfull = ntiles;
    split = 1;
    double best = (ntiles + slots - 1) / slots * cost[1];
    for (int g = 2; g <= 4; g += 2) {
      const double c = whole / slots * cost[1] + (g * rem + slots - 1) / slots * cost[g];
      if (rem > 0 && c < best - 1e-9) {
        best = c;
        nfull = whole;
        split = g;
      }
    }
    grid = (unsigned)(nfull + split * (ntiles - nfull));
  }
};

// the lookahead columns: block columns jcol .. jcol + ncol - 1 only, 64x64 tiles (latency-critical, see trsm_kernel);
// blockIdx.z = site * ncol + column
template <typename T>
__global__ __launch_bounds__(256, 2) void syrk_col_kernel(T* __restrict__ A, long ld, int k, int nk, int jcol, int nbk,
                                                          int ncol, long bs) {
  using G = TileGemm<T, true, true, 64, 64>;
  __shared__ T smem[G::SMEM_ELEMS];
  __builtin_amdgcn_s_setprio(3);  // panel chain: outrank co-resident bulk-update waves
  A += (long)((int)blockIdx.z / ncol) * bs;
  const int jc = jcol + (int)blockIdx.z % ncol;
  if ((int)blockIdx.x >= 2 * (nbk - jc)) return;  // the second column is one block shorter
  const long row0 = (long)jc * NB + (long)blockIdx.x * 64;
  const long col0 = (long)jc * NB + (long)blockIdx.y * 64;
  if (col0 > row0 + 63) return;  // strictly upper 64x64 quadrant of the diagonal block
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + row0 * ld + col0;
  typename G::acc_t keep[G::MI][G::NI];
  trailing_begin<T, G>(acc, keep, C, ld);
  G::run(A + row0 * ld + (long)k * NB, ld, A + col0 * ld + (long)k * NB, ld, nk * (NB / 16), smem, acc);
  trailing_end<T, G>(acc, keep, C, ld);
}

// the same columns in 128 x 128 tiles of the direct-to-LDS core, for batched plans: there the group's column update is
// thousands of tiles and not latency-critical (the sites fill each other's gaps); bitwise the same sums
template <typename T, int CMODE = 0>
__global__ __launch_bounds__(256, (TileCore<T, true, true>::OCC)) void syrk_col128_kernel(T* __restrict__ A, long ld, int k, int nk, int jcol,
                                                                                          int nbk, int ncol, long bs) {
  __shared__ T smem[TileCore<T, true, true>::SMEM_ELEMS];
  A += (long)((int)blockIdx.z / ncol) * bs;
  const int jc = jcol + (int)blockIdx.z % ncol;
  if ((int)blockIdx.x >= nbk - jc) return;  // the later columns are shorter
  syrk_tile<T, 128, 128, false, CMODE>(A, ld, k, nk, (long)(jc + (int)blockIdx.x) * NB, (long)jc * NB, smem);
}

// launch the LDS-resident diagonal-block kernel (needs > 64 KB of dynamic LDS: opt in once per instantiation).
// DGP_F32_DIAG64=1 factors the diagonal blocks of fp32 matrices with the MIXED-PRECISION instantiation (block promoted
// to fp64 in LDS, dgp_diag.h).  Off by default -- measured over 18 matrices (profiles/r03_fp32_error_sources.txt): once
// the trailing updates sum from zero (dgp_gemm.h::trailing_begin) the fp64 block changes neither the log-determinant
// nor the quadratic form's error (which is then set by L being STORED in fp32), and costs 2 % (n = 16384) to 12 %
// (n = 2048) of a fit step.  Kept as a measurement knob.
// Runs `configure` once per device (the calling thread's current one) UNDER the lock and marks the device configured only
// when it returned hipSuccess: a second host thread that drives the same device (thread ranks, two Python threads with a
// plan each) either finds the opt-in done or waits for it -- it can never launch with > 64 KB of dynamic LDS before the
// attribute is set (with the flag set first and the call outside the lock it could).  -> configure's error, else hipSuccess.
struct DeviceOnce {
  std::mutex mtx;
  bool seen[64] = {false};
  template <typename F>
  hipError_t run(F&& configure) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return configure();  // unknown: configure every time (cheap)
    std::lock_guard<std::mutex> lock(mtx);
    if (seen[dev]) return hipSuccess;
    const hipError_t e = configure();
    if (e == hipSuccess) seen[dev] = true;
    return e;
  }
};
static bool f32_diag64() {
  static const bool v = [] {
    const char* e = getenv("DGP_F32_DIAG64");
    return e ? atoi(e) != 0 : false;
  }();
  return v;
}
// single-site plans: the bulk update's waves yield their CU to the diagonal-block kernel (dgp_common.h: yield_if_asked)
static bool yields(const Batch& bt) { return bt.B == 1 && bt.tuning().chain_yield != 0; }
// one bulk launch of the trailing update (SyrkShape), polite or not
template <typename T>
static void launch_bulk(T* A, long N, int k, int nk, int jbeg, const SyrkShape& sh, int nt, int* info, hipStream_t s, const Batch& bt) {
  const int nbk = (int)(N / NB);
  const dim3 grid(sh.grid, 1, (unsigned)bt.B);
  // One site, chain-bound sizes: 14 KB of unused dynamic LDS per workgroup hold the bulk update at TWO workgroups per CU, so
  // that trsm / the column updates of the chain find a slot on every CU at once instead of waiting for tiles to retire (first
  // panel of a pair at n = 8192: trsm 59 -> 33 us, column update 77 -> 30).  Measured (step, ms): n = 6144 6.72 -> 6.69,
  // 8192 11.93 -> 11.89, 12288 33.3 -> 32.7, 16384 fp64 71.6 -> 70.9, 16384 fp32 38.95 -> 38.55; n >= 24576 (bound by the
  // bulk launches) and n <= 4096 (one round of tiles anyway): nothing or slightly worse -- hence the window.
  static const int pad_env = getenv("DGP_BULK_LDS_PAD") ? std::min(16384, std::max(0, atoi(getenv("DGP_BULK_LDS_PAD")))) : 14336;
  // (batched plans: 64 x 4096 797 -> 789 fits/s, 32 x 8192 115.6 -> 115.0, 128 x 2048 no change -- their chain kernels are wide enough)
  const size_t pad = (bt.B == 1 && nbk >= 40 && nbk <= 160) ? (size_t)pad_env : 0;
  // DGP_BULK_C: how the tiles of C move -- 0 plain 8-byte accesses (the form until round 4), 1 non-temporal, 2 16-byte accesses with
  // a lane-pair swap (fp64), 3 both.  Measured on the tile alone (scripts/syrk_persist.hip, fp64, TFLOP/s at K = 512 / 256): 64.0 /
  // 52.7, 68.6 / 60.9, 69.7 / 62.2, 70.6 / 64.1 -- the read-modify-write of C, not the dispatch or the ring fill, is what
  // separates a short-K tile from the long-K rate (74.4 with no C traffic).  In situ every variant ties or LOSES (default 0):
  // the non-temporal form moves the bulk launches' sum 79.2 -> 75.7 ms and the step not at all; the 16-byte form needs a
  // lane-pair swap on 128 live accumulator registers, which this kernel (168 registers, 20 already spilled) only affords by
  // spilling 99 registers around every tile (400 bytes of scratch per lane -- as much traffic as the tile itself): bulk sum
  // 77.3 -> 82.4 ms, step 275.4 -> 280.2 (32 x n = 8192), 11.90 -> 12.55 ms (one site).  EXPERIMENTS.md round 5.
  static const int cmode_env = getenv("DGP_BULK_C") ? atoi(getenv("DGP_BULK_C")) & 3 : DGP_BULK_C_DEFAULT;
  const int cmode = sizeof(T) == 8 ? cmode_env : (cmode_env & 1);  // (fp32 tiles read C in the epilogue: no wide form)
  const unsigned* yw = reinterpret_cast<const unsigned*>(info + CHAIN_YIELD);
  // batched plans on the group panel solve: the 32 KB ring (syrk_kernel: RING) when DGP_BULK_RING=2 (experiment)
  static const int ring_env = getenv("DGP_BULK_RING") ? atoi(getenv("DGP_BULK_RING")) : 3;
  static const int pad_batch = getenv("DGP_BULK_PAD_BATCH") ? std::min(65536, std::max(0, atoi(getenv("DGP_BULK_PAD_BATCH")))) : 0;  // experiment
  if (bt.B >= 4 && bt.tuning().group_gemm && bt.W != nullptr && ring_env == 2) {
    syrk_kernel<T, false, 0, 2><<<grid, 256, (size_t)pad_batch, s>>>(A, N, k, nk, jbeg, sh.nfull, sh.split, bt.ws, nt, bt.tuning().syrk_super, nullptr);
    return;
  }
  const size_t lds_pad = (bt.B >= 4 && pad_batch > 0) ? (size_t)pad_batch : pad;
#define DGP_LAUNCH_BULK(POLITE_, CM_)                                                                                              \
  syrk_kernel<T, POLITE_, CM_><<<grid, 256, lds_pad, s>>>(A, N, k, nk, jbeg, sh.nfull, sh.split, bt.ws, nt, bt.tuning().syrk_super, \
                                                         POLITE_ ? yw : nullptr)
  if (yields(bt)) {
    switch (cmode) {
      case 1: DGP_LAUNCH_BULK(true, 1); break;
      case 2: DGP_LAUNCH_BULK(true, 2); break;
      case 3: DGP_LAUNCH_BULK(true, 3); break;
      default: DGP_LAUNCH_BULK(true, 0);
    }
  } else {
    switch (cmode) {
      case 1: DGP_LAUNCH_BULK(false, 1); break;
      case 2: DGP_LAUNCH_BULK(false, 2); break;
      case 3: DGP_LAUNCH_BULK(false, 3); break;
      default: DGP_LAUNCH_BULK(false, 0);
    }
  }
#undef DGP_LAUNCH_BULK
}
template <typename TS, typename TC>
static void launch_diag_as(TS* A, long N, long k0, TS* Tinv, TS* logdet, int* info, hipStream_t s, Batch bt, bool init,
                           double* logdet_hi, int done_index, int done_value) {
  // the opt-in for > 64 KB of dynamic LDS belongs to the CURRENT DEVICE's function object: keyed by device, so that a
  // process that drives plans on several GPUs opts in on each (a failure surfaces as the launch error below it)
  const size_t bytes = potrf_diag_fast_smem<TC>();
  static DeviceOnce configured;
  // (a failed opt-in is retried by the next call and surfaces here as the launch error: invalid value for > 64 KB)
  (void)configured.run([&] {
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&potrf_diag_fast_kernel<TS, TC>),
                               hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
  });
  potrf_diag_fast_kernel<TS, TC><<<dim3(1, 1, (unsigned)bt.B), 256, bytes, s>>>(
      A, N, k0, Tinv, logdet, info, bt.ws, bt.ws * (long)sizeof(TS) / (long)sizeof(int), init ? 1 : 0, POTRF_INFO_INTS, logdet_hi,
      done_index, done_value, yields(bt) ? CHAIN_YIELD : -1);
}
// the fp32 plans' unrounded log-determinant lives in the scalar block right behind (log-det, quad): element 2..3 as ONE double
template <typename T>
static double* logdet_hi_slot(T* logdet) {
  return sizeof(T) == 4 ? reinterpret_cast<double*>(logdet + 2) : nullptr;
}
template <typename T>
static void launch_diag(T* A, long N, long k0, T* Tinv, T* logdet, int* info, hipStream_t s, Batch bt, bool init = false,
                        int done_index = -1, int done_value = 0) {
  if (sizeof(T) == 4 && f32_diag64())
    launch_diag_as<T, double>(A, N, k0, Tinv, logdet, info, s, bt, init, logdet_hi_slot(logdet), done_index, done_value);
  else
    launch_diag_as<T, T>(A, N, k0, Tinv, logdet, info, s, bt, init, logdet_hi_slot(logdet), done_index, done_value);
}

// ---- the group's panel solve as ONE GEMM (round 5; batched plans) ---------------------------------------------------------
// The group-ahead schedule below factors a group of G panels column by column over the FULL height of the matrix: per panel a
// column update (K = 128 h), the diagonal block, a trsm (K = 128) -- 2 G - 1 launches that each sweep the group's G x (rows
// below) tiles with a short k-range: the chain of a batch is thousands of 64-row tiles at 45-60 % of the MFMA rate, and for
// mid-size sites (64 x n = 4096) it, not the bulk update, bounds the factorisation (potrf wall 27.7 ms, bulk launches 16.1).
// But only the G x G-block DIAGONAL block D of the group has to be factored panel by panel.  With L_D final and T_D = L_D^-1
// (its diagonal 128-blocks come from the diagonal-block kernel, the rest from the first log2(G) levels of the inverse's own
// recursion, restricted to D: work the inverse stage would do anyway), the rows below are
//     L[i, group] = A[i, group] T_D^T,        column block j:  sum over the column blocks c <= j of A[i, c] T_D[j, c]^T
// -- the same flops (G (G + 1) / 2 tile products of K = 128 per row tile) in ONE launch of 128 x 128 tiles on the direct-to-LDS
// core with k-ranges of 128 .. 128 G, in place (column block j is written after every product that reads it; j descends).
// Not bitwise the panel-by-panel chain (a different association of the same sums): batched plans against single-site plans
// agree to rounding, ~1e-13 relative in fp64 (tests/test_gpu_stages.py, test_gpu_headline_shape.py hold 1e-11).
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, true, true>::OCC)) void trsm_group_kernel(T* __restrict__ A, const T* __restrict__ Tm, long ld,
                                                                                         int k0, int G, long bs) {
  A = site(A, bs);
  Tm = site(Tm, bs);
  using K = TileCore<T, true, true, 128, 128, 1>;
  using Gm = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  T* Arow = A + (long)(k0 + G + (int)blockIdx.x) * NB * ld + (long)k0 * NB;  // this workgroup's 128 rows of the group's columns
  const T* TD = Tm + (long)k0 * NB * ld + (long)k0 * NB;                      // T_D, lower triangular, zeros above the diagonal
#pragma unroll 1
  for (int j = G - 1; j >= 0; --j) {
    typename Gm::acc_t acc[Gm::MI][Gm::NI];
    Gm::zero(acc);
    K::run(Arow, ld, TD + (long)j * NB * ld, ld, (j + 1) * (NB / 16), smem, acc);  // (every wave is past its last operand read)
    T* out = Arow + (long)j * NB;
    K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; });
  }
}
// the first log2(G) levels of the inverse's recursion for the diagonal group block at block column k0 (defined behind
// trtri_level): T_D's off-diagonal blocks into Tm, scratch in W
template <typename T>
static void trtri_group(const T* L, T* Tm, T* W, long N, int k0, int G, hipStream_t s, Batch bt);

template <typename T>
int potrf(T* A, long N, T* Tinv, T* logdet, int* info, int lookahead, hipStream_t s, hipStream_t s2, hipEvent_t* ev,
          hipEvent_t* syrk_ev, int* n_syrk, double* syrk_flop, int nck, const int* ck_blocks, hipEvent_t* ck_ev,
          void (*on_ck)(void*, int), void* ck_ctx, Batch bt, int q_stop, PotrfCarry* carry) {
  const int nbk = (int)(N / NB);
  // checkpoint c: recorded on s once the first ck_blocks[c] block columns of L are final (every schedule records
  // every checkpoint, at the latest when the factorisation is complete)
  int ck_next = carry ? carry->ck_next : 0;
  auto checkpoint = [&](int cols_final) {
    while (ck_next < nck && ck_blocks[ck_next] <= cols_final) {
      hipEventRecord(ck_ev[ck_next], s);
      if (on_ck) on_ck(ck_ctx, ck_next);  // the caller enqueues its dependent work NOW, not after the whole schedule
      ++ck_next;
    }
  };
  int ns = 0;
  double flop = 0.0;
  const unsigned Bz = (unsigned)bt.B;
  auto tri = [](int m) { return (unsigned)(m * (m + 1) / 2); };
  const double tile_flop = 2.0 * NB * NB * NB;
  if (!lookahead || nbk < 4 || s2 == nullptr || ev == nullptr) {
    for (int k = 0; k < nbk; ++k) {
      launch_diag<T>(A, N, (long)k * NB, Tinv, logdet, info, s, bt, k == 0);
      if (k + 1 < nbk) {
        trsm_kernel<T><<<dim3(2 * (nbk - k - 1), 1, Bz), 256, 0, s>>>(A, Tinv, N, k, bt.ws);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s);
        {
          const SyrkShape sh((int)tri(nbk - k - 1), bt.tuning().syrk_slots / bt.B);
          syrk_kernel<T><<<dim3(sh.grid, 1, Bz), 256, 0, s>>>(A, N, k, 1, k + 1, sh.nfull, sh.split, bt.ws, nbk - k - 1, bt.tuning().syrk_super);  // (same stream as the chain: nothing to yield to)
        }
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s);
        flop += tile_flop * tri(nbk - k - 1) * bt.B;
        ++ns;
      }
    }
    checkpoint(nbk);
    if (n_syrk) *n_syrk = ns;
    if (syrk_flop) *syrk_flop = flop;
    return (int)hipGetLastError();
  }
  {
    // GROUP-AHEAD schedule.  Group q = G consecutive panels (G = 2: pairs, for one site; G = 4 for batched plans,
    // where the chain is shared by the batch and the bulk update gains from K = 512).  The chain on stream s factors
    // group q+1 while the bulk update of group q runs on s2, so consecutive bulk launches follow each other without
    // waiting for a panel:
    //   chain(q):  [wait U[q-2]]  the G columns of group q <- group q-1 (K = 128 G, one launch)   record P[q-1]
    //              for each panel k of the group:  column k <- the group's earlier panels (K = 128 h)
    //                                              diag(k)  trsm(k)
    //   bulk(q-1): [wait P[q-1]]  block columns >= G(q+1) <- group q-1 (K = 128 G)               record U[q-1]
    // Column c (group qc) gets groups <= qc-2 from the bulk launches, group qc-1 and its own group's earlier panels
    // from the chain.  bulk(q-1) is released only once the chain's G-column update for group q is through: released
    // together, the bulk launch takes every CU slot first and that update (the widest chain kernel) runs ~100 us
    // instead of ~20; the chain, not the bulk stream, is what the first half of a single site's factorisation
    // waits for.
    const int G = lookahead < 2 ? 2 : (lookahead > 8 ? 8 : lookahead);
    hipEvent_t* P = ev;
    hipEvent_t* U = ev + nbk;
    const int Q = (nbk + G - 1) / G;
    // q_stop: hand over to the split chain (potrf_split) after chain(q_stop - 1): pair q_stop - 1 is factored, its
    // updates of the later columns (chain and bulk) are left to the caller, bulk(q_stop - 2) may still be running
    const int Qrun = (q_stop >= 0 && q_stop < Q) ? q_stop : Q;
    for (int q = 0; q < Qrun; ++q) {
      const int k0 = G * q, ncol = nbk - k0 < G ? nbk - k0 : G;
      if (q >= 2) hipStreamWaitEvent(s, U[q - 2], 0);  // bulk(q-2) exists whenever chain(q) does
      if (q >= 1) {
        static const int col128 = getenv("DGP_COL128") ? atoi(getenv("DGP_COL128")) : 4;  // batch size from which the group's columns use 128-tiles
        if (bt.B >= col128) {
          // DGP_COL_C: the group's look-ahead column update likewise (its tiles ARE re-read soon, by the chain: only the wide form, 2, is a candidate)
          static const int ccol = getenv("DGP_COL_C") ? atoi(getenv("DGP_COL_C")) & 3 : DGP_COL_C_DEFAULT;
          const dim3 cgrid(nbk - k0, 1, ncol * Bz);
          switch (sizeof(T) == 8 ? ccol : (ccol & 1)) {
            case 1: syrk_col128_kernel<T, 1><<<cgrid, 256, 0, s>>>(A, N, k0 - G, G, k0, nbk, ncol, bt.ws); break;
            case 2: syrk_col128_kernel<T, 2><<<cgrid, 256, 0, s>>>(A, N, k0 - G, G, k0, nbk, ncol, bt.ws); break;
            case 3: syrk_col128_kernel<T, 3><<<cgrid, 256, 0, s>>>(A, N, k0 - G, G, k0, nbk, ncol, bt.ws); break;
            default: syrk_col128_kernel<T, 0><<<cgrid, 256, 0, s>>>(A, N, k0 - G, G, k0, nbk, ncol, bt.ws);
          }
        }
        else syrk_col_kernel<T><<<dim3(2 * (nbk - k0), 2, ncol * Bz), 256, 0, s>>>(A, N, k0 - G, G, k0, nbk, ncol, bt.ws);
      }
      if (q >= 1 && k0 + G < nbk) {  // bulk(q-1): columns >= G(q+1) exist
        hipEventRecord(P[q - 1], s);
        hipStreamWaitEvent(s2, P[q - 1], 0);
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s2);
        {
          const SyrkShape sh((int)tri(nbk - k0 - G), bt.tuning().syrk_slots / bt.B);
          launch_bulk<T>(A, N, k0 - G, G, k0 + G, sh, nbk - k0 - G, info, s2, bt);
        }
        if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s2);
        flop += (double)G * tile_flop * tri(nbk - k0 - G) * bt.B;
        ++ns;
        hipEventRecord(U[q - 1], s2);
      }
      // the group's panel solve as one GEMM (trsm_group_kernel): batches from 4 sites (the chain is then wide enough to be
      // throughput-, not latency-bound), full groups with rows below them, G a power of two (the inverse's levels)
      // (one site: the groups of 4 / 8 panels that large matrices start with -- G >= 4 means N >= 12288, beyond the early inverse,
      // which would share the scratch)
      const bool group_gemm = bt.tuning().group_gemm && bt.W != nullptr && ncol == G && k0 + G < nbk && (G & (G - 1)) == 0 && G >= 4 &&
                              nck == 0;
      const int kend = group_gemm ? k0 + G : nbk;  // the panel-by-panel chain covers block rows < kend
      for (int h = 0; h < ncol; ++h) {
        const int k = k0 + h;
        if (h >= 1) syrk_col_kernel<T><<<dim3(2 * (kend - k), 2, Bz), 256, 0, s>>>(A, N, k0, h, k, kend, 1, bt.ws);
        launch_diag<T>(A, N, (long)k * NB, Tinv, logdet, info, s, bt, k == 0);
        if (k + 1 < kend) trsm_kernel<T><<<dim3(2 * (kend - k - 1), 1, Bz), 256, 0, s>>>(A, Tinv, N, k, bt.ws);
        if (!group_gemm) checkpoint(k + 1);
      }
      if (group_gemm) {
        trtri_group<T>(A, Tinv, (T*)bt.W, N, k0, G, s, bt);
        trsm_group_kernel<T><<<dim3((unsigned)(nbk - k0 - G), 1, Bz), 256, 0, s>>>(A, Tinv, N, k0, G, bt.ws);
        checkpoint(k0 + G);
      }
    }
    // bulk(q) exists for q <= Q-3 and chain(q+2) has waited on every one of them: s is joined
    if (Qrun == Q) checkpoint(nbk);
    if (carry) {
      carry->ck_next = ck_next;
      carry->ns = ns;
      carry->flop = flop;
    }
    if (n_syrk) *n_syrk = ns;
    if (syrk_flop) *syrk_flop = flop;
    return (int)hipGetLastError();
  }
}

// ------------------------------------------------------------------------------------------
// SPLIT PANEL CHAIN (one site, pairs of panels).  The schedule above runs, per panel, diag -> trsm -> column update one
// after the other on one stream: ~74 us, of which the block kernel is 36.  But the next diagonal block only needs ITS OWN
// 128 rows of the panel and its own 128 x 128 tile of the update: everything else of trsm / column update is off the
// critical path.  Here the chain is split over two streams:
//   critical stream s   crit(k): block (k,k) -= its pending panels (the trsm of its own 128 rows of panel k-1 from a
//                                SNAPSHOT of those rows + the tile update; for even k also panel k-2, whose rows are final),
//                                ten small workgroups;   then diag(k)
//   rest stream c2      R(k):    trsm(k) on all rows below the block (in place), then the column updates of the next
//                                column(s) -- without the tile crit owns, and leaving a snapshot of the NEXT block's rows
//                                for crit(k+2) -- and the release of the bulk update of the finished pair
// with crit(k) waiting for R(k-2) and R(k) for diag(k): R(k-1) and crit(k) + diag(k) overlap.  Every element still
// receives its panels in ascending order and each pass continues the same k-ordered fma chain, so in fp64 the factor is
// BITWISE the one of the single-stream schedule (tests/test_gpu_stages.py).

// snapshot[r][c] = A[(row_block 128 + r) ld + col_block 128 + c]  (128 x 128, row stride 128)
template <typename T>
__global__ __launch_bounds__(256) void snap_copy_kernel(const T* __restrict__ A, long ld, int row_block, int col_block,
                                                        T* __restrict__ snap) {
  const T* src = A + (long)row_block * NB * ld + (long)col_block * NB;
  for (int i = threadIdx.x + blockIdx.x * 256; i < NB * NB; i += 256 * gridDim.x) snap[i] = src[(long)(i / NB) * ld + i % NB];
}

// crit(k): the 128 x 128 diagonal block (k,k) receives its pending panels.  Workgroup = one 32 x 32 tile (ti >= tj) of
// the block, wave = one 16 x 16 sub-block of the tile.
//   panel k-1:  P = snap X^T with X = L_{k-1,k-1}^-1 (the diagonal block of Tinv), snap = rows of block k of panel k-1
//               BEFORE their trsm (R(k-1) overwrites them in place meanwhile, with bitwise the same P);
//   panel k-2 (two_pending): P = A[block k rows, block k-2 columns], final.
// LDS: X as 36 sub-blocks of 16 x 17 (lower block triangle) + the tile's four 16-row strips of HALF of P (16 sub-blocks).
template <typename T>
__global__ __launch_bounds__(256) void crit_kernel(T* __restrict__ A, long ld, int k, const T* __restrict__ Tinv,
                                                   const T* __restrict__ snap, int nfinal, int first_final, int solve_,
                                                   int* __restrict__ info, int need_rest, int need_bulk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char crit_smem[];
  using acc_t = typename Mfma<T>::acc_t;
  T* sX = reinterpret_cast<T*>(crit_smem);   // DGP_DTRI sub-blocks
  T* sP = sX + DGP_DTRI * DGP_DBLK;          // 4 strips x 4 sub-blocks (one half of the panel's columns at a time)
  int ti, tj;
  tri_decode((int)blockIdx.x, ti, tj);       // 32 x 32 tile of the block's lower triangle
  const int t = threadIdx.x, lane = t & 63, wave = __builtin_amdgcn_readfirstlane(t >> 6);
  const int e_r = t >> 4, e_c = t & 15;      // element of a 16 x 16 sub-block this thread stages
  __builtin_amdgcn_s_setprio(3);
  // The rest stream's R(k-2) (and, for an even block, the bulk update of pair k/2 - 2) must have finished.  They were
  // enqueued long before this launch and have normally finished long ago -- an event wait in front of this kernel would
  // cost the critical stream ~11 us per panel for nothing (measured: profiles/r03_split_chain.txt), so the kernel checks
  // two counters itself: flags[0] = rest steps finished, flags[1] = bulk launches finished (one-thread signal kernels
  // behind them).
  // The producers never depend on anything this kernel does; the wait is bounded (chain_wait above).
  if (!chain_wait(info + CHAIN_FLAG0, need_rest, info + CHAIN_FLAG0 + 1, need_bulk, info)) return;
  T* Akk = A + (long)k * NB * ld + (long)k * NB;
  // strip s (0..3) = 16 rows of the block: strips 0,1 = the tile's rows (ti), strips 2,3 = its columns' rows (tj)
  const int srow = (wave < 2 ? 2 * ti + wave : 2 * tj + (wave - 2)) * 16;  // first row (inside the block) of this wave's strip
  // the accumulator: this wave's 16 x 16 sub-block (a, b) of the tile; strictly upper sub-blocks of a diagonal tile are never read
  const int sa = wave >> 1, sb = wave & 1;
  const bool live = !(ti == tj && sa < sb);
  T* Csub = Akk + (long)(32 * ti + 16 * sa) * ld + 32 * tj + 16 * sb;
  acc_t acc = {T(0), T(0), T(0), T(0)};
  if (sizeof(T) == 8 && live)
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = Csub[(long)Mfma<T>::crow(lane, r) * ld + (lane & 15)];
  // ---- everything this kernel reads from global memory is requested up front
  // nfinal panels from first_final on have FINAL rows in A and are applied first (0: odd block; 1: even block, panel k-2;
  // G: the first block after the hand-over from a schedule with groups of G panels, which solved all of them: no
  // snapshot, no trsm here -- solve_ = 0); then panel k-1 from the snapshot (solve_ = 1)
  const bool solve = solve_ != 0;
  T xreg[DGP_DTRI];  // X = L_{k-1,k-1}^-1, lower block triangle
  T af[DGP_DNB][4];  // this wave's 16 rows of the snapshot as MFMA A fragments
  if (solve) {
    const T* Xg = Tinv + (long)(k - 1) * NB * ld + (long)(k - 1) * NB;
#pragma unroll
    for (int bi = 0; bi < DGP_DNB; ++bi)
#pragma unroll
      for (int bj = 0; bj <= bi; ++bj) xreg[dtri(bi, bj)] = Xg[(long)(16 * bi + e_r) * ld + 16 * bj + e_c];
#pragma unroll
    for (int kc = 0; kc < DGP_DNB; ++kc)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) af[kc][ks] = snap[(long)(srow + (lane & 15)) * NB + 16 * kc + 4 * ks + (lane >> 4)];
  }
  // The panel's 128 columns are processed in two halves (LDS: X + 4 strips x 4 sub-blocks = 113 KB in fp64, so that the
  // workgroup fits on a CU beside ONE bulk workgroup: with 148 KB it could only start once a whole CU had drained and
  // starved behind a saturating bulk launch -- measured 277 us instead of 15).
  auto apply = [&](int half) {  // acc -= P[strip sa of ti] P[strip sb of tj]^T over columns 64 half .. 64 half + 63, ascending
    if (!live) return;
    const T* Pi = sP + (sa)*4 * DGP_DBLK;
    const T* Pj = sP + (2 + sb) * 4 * DGP_DBLK;
#pragma unroll
    for (int kb = 0; kb < 4; ++kb)
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const T a = frag_rc(Pi + kb * DGP_DBLK, ks, lane), b = frag_rc(Pj + kb * DGP_DBLK, ks, lane);
        acc = sizeof(T) == 8 ? Mfma<T>::mma(-a, b, acc) : Mfma<T>::mma(a, b, acc);  // fp32: summed from zero, joined below
      }
    (void)half;
  };
  for (int f = 0; f < nfinal; ++f) {  // panels k-2 (, k-1): final rows of L, this wave's strip straight into LDS (lane = one column of a half)
    const T* Pg = A + ((long)k * NB + srow) * ld + (long)(first_final + f) * NB + lane;
    T pr[2][16];
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) pr[h][rr] = Pg[(long)rr * ld + 64 * h];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
#pragma unroll
      for (int rr = 0; rr < 16; ++rr) sP[(wave * 4 + (lane >> 4)) * DGP_DBLK + rr * DGP_DS + (lane & 15)] = pr[h][rr];
      __syncthreads();
      apply(h);
      __syncthreads();
    }
  }
  if (solve) {
  // ---- panel k-1: X into LDS
#pragma unroll
  for (int b = 0; b < DGP_DTRI; ++b) sX[b * DGP_DBLK + e_r * DGP_DS + e_c] = xreg[b];
  __syncthreads();
  // P[strip, jb] = sum_{kc <= jb} snap[strip, kc] X[jb, kc]^T   (k ascending: the order of trsm_kernel's chain; the
  // zero blocks kc > jb add nothing)
#pragma unroll
  for (int h = 0; h < 2; ++h) {
#pragma unroll
    for (int j4 = 0; j4 < 4; ++j4) {
      const int jb = 4 * h + j4;
      acc_t p = {T(0), T(0), T(0), T(0)};
#pragma unroll
      for (int kc = 0; kc <= jb; ++kc)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) p = Mfma<T>::mma(af[kc][ks], frag_rc(sX + dtri(jb, kc) * DGP_DBLK, ks, lane), p);
#pragma unroll
      for (int r = 0; r < 4; ++r) sP[(wave * 4 + j4) * DGP_DBLK + Mfma<T>::crow(lane, r) * DGP_DS + (lane & 15)] = p[r];
    }
    __syncthreads();
    apply(h);
    __syncthreads();
  }
  }
  if (live)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      T* c = &Csub[(long)Mfma<T>::crow(lane, r) * ld + (lane & 15)];
      *c = sizeof(T) == 8 ? acc[r] : *c - acc[r];
    }
}
__global__ void chain_signal_kernel(int* word, int value, int* also_zero = nullptr) {
  if (threadIdx.x == 0) {
    if (also_zero) {  // = the plan's info block: a stale status (info[0]) and a stale abort word would make the waiters leave
      __hip_atomic_store(also_zero, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      __hip_atomic_store(also_zero + CHAIN_ABORT, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    __hip_atomic_store(word, value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
  }
}
template <typename T>
static size_t crit_smem_bytes() { return (size_t)(DGP_DTRI + 16) * DGP_DBLK * sizeof(T); }

// Column updates of the rest stream: up to two jobs per launch (blockIdx.z), each one block column jc receiving panels
// k .. k + nk - 1 on its rows from block row_lo down (row_lo = jc: with the diagonal block; jc + 1: without -- crit owns
// it), 64 x 64 tiles like syrk_col_kernel.  Tiles of block row snap_rb are also stored to `snap` (the rows the next
// crit kernel will solve itself).
struct ChainJob {
  int jc, row_lo, k, nk, snap_rb;
};
struct ChainJobs {
  ChainJob j[2];
};
template <typename T>
__device__ __forceinline__ void chain_col_tile(T* __restrict__ A, long ld, const ChainJob& jb, long row0, long col0,
                                               T* __restrict__ snap, T* __restrict__ smem) {
  using G = TileGemm<T, true, true, 64, 64>;
  {
  typename G::acc_t acc[G::MI][G::NI];
  T* C = A + row0 * ld + col0;
  typename G::acc_t keep[G::MI][G::NI];
  trailing_begin<T, G>(acc, keep, C, ld);
  G::run(A + row0 * ld + (long)jb.k * NB, ld, A + col0 * ld + (long)jb.k * NB, ld, jb.nk * (NB / 16), smem, acc);
  trailing_end<T, G>(acc, keep, C, ld);
  if (jb.snap_rb >= 0 && row0 / NB == jb.snap_rb) {
    T* sn = snap + (row0 - (long)jb.snap_rb * NB) * NB + (long)blockIdx.y * 64;
    G::foreach (acc, [&](int r, int c, T& v) { sn[(long)r * NB + c] = sizeof(T) == 8 ? -v : v; });
  }
  }
}

template <typename T>
__global__ __launch_bounds__(256, 2) void chain_col_kernel(T* __restrict__ A, long ld, int nbk, ChainJobs jobs,
                                                           T* __restrict__ snap, int* __restrict__ info, int done_value) {
  using G = TileGemm<T, true, true, 64, 64>;
  __shared__ T smem[G::SMEM_ELEMS];
  __builtin_amdgcn_s_setprio(3);
  const ChainJob jb = jobs.j[blockIdx.z];
  const long row0 = (long)jb.row_lo * NB + (long)blockIdx.x * 64;
  const long col0 = (long)jb.jc * NB + (long)blockIdx.y * 64;
  // (workgroups beyond the matrix or in the strictly upper 64 x 64 quadrant of a diagonal block only take their ticket)
  if (row0 < (long)nbk * NB && col0 <= row0 + 63) chain_col_tile<T>(A, ld, jb, row0, col0, snap, smem);
  // the LAST workgroup to finish publishes "R(k) is through" (flags[0] = done_value) for crit(k + 2): no signal launch
  __syncthreads();
  if (threadIdx.x == 0) {
    __threadfence();
    const int total = (int)(gridDim.x * gridDim.y * gridDim.z);
    if (atomicAdd(&info[CHAIN_TICKET], 1) == total - 1) {
      info[CHAIN_TICKET] = 0;
      __hip_atomic_store(&info[CHAIN_FLAG0], done_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
  }
}

// k_start (a multiple of G_old): the block columns before it are factored by the single-stream GROUP schedule (potrf with q_stop,
// groups of G_old panels: pairs below 96 block columns, 4 or 8 above) --
// while a bulk launch is many rounds long the factorisation is bound by it, and crit's ten workgroups would queue behind
// it like every other chain kernel -- and the split chain takes over from there.
template <typename T>
int potrf_split(T* A, long N, T* Tinv, T* logdet, int* info, T* snap, hipStream_t s, hipStream_t c2, hipStream_t s2,
                hipEvent_t* ev /* 3 nbk */, hipEvent_t* syrk_ev, int* n_syrk, double* syrk_flop, int nck, const int* ck_blocks,
                hipEvent_t* ck_ev, void (*on_ck)(void*, int), void* ck_ctx, int k_start, int G_old, const Tuning* tune) {
  const int nbk = (int)(N / NB);
  Batch bt;
  bt.tune = tune;
  if (G_old < 2) G_old = 2;
  if (k_start < 0) k_start = 0;
  k_start = (k_start + G_old - 1) / G_old * G_old;  // a group boundary of the schedule that runs first (G_old is even)
  if (k_start + 4 > nbk)  // nothing left for the split chain
    return potrf<T>(A, N, Tinv, logdet, info, G_old, s, s2, ev, syrk_ev, n_syrk, syrk_flop, nck, ck_blocks, ck_ev, on_ck, ck_ctx, bt);
  PotrfCarry carry;
  hipEvent_t* ED = ev;            // diag(k) done, on s
  hipEvent_t* ER = ev + nbk;      // R(k) done, on c2
  hipEvent_t* U = ev + 2 * nbk;   // bulk(q) done, on s2
  const size_t cbytes = crit_smem_bytes<T>();
  static DeviceOnce configured;  // one per instantiation (T), keyed by device
  {
    const hipError_t ce = configured.run([&] {
      return hipFuncSetAttribute(reinterpret_cast<const void*>(&crit_kernel<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cbytes);
    });
    if (ce != hipSuccess) return (int)ce;
  }
  int ck_next = 0;
  int ns = 0, last_u = -1;
  double flop = 0.0;
  int* flags = info + CHAIN_FLAG0;  // zeroed with the other status words by the first diagonal-block kernel
  if (k_start > 0) {
    const int q0 = k_start / G_old;
    const int rc = potrf<T>(A, N, Tinv, logdet, info, G_old, s, s2, ev, syrk_ev, n_syrk, syrk_flop, nck, ck_blocks, ck_ev, on_ck,
                            ck_ctx, bt, q0, &carry);
    if (rc) return rc;
    ck_next = carry.ck_next;
    ns = carry.ns;
    flop = carry.flop;
  }
  auto checkpoint = [&](int cols_final, hipStream_t st) {
    while (ck_next < nck && ck_blocks[ck_next] <= cols_final) {
      hipEventRecord(ck_ev[ck_next], st);
      if (on_ck) on_ck(ck_ctx, ck_next);
      ++ck_next;
    }
  };
  auto tri = [](int m) { return (unsigned)(m * (m + 1) / 2); };
  const double tile_flop = 2.0 * NB * NB * NB;
  auto bulk_exists = [&](int q) { return q >= 0 && 2 * q + 4 < nbk; };
  // block columns >= 2q + 4 <- panels kfirst .. kfirst + nk - 1 (pair q: kfirst = 2q, nk = 2; at the hand-over the last
  // group of the schedule that ran first), behind the rest step recorded in ER[2q + 1]
  auto bulk = [&](int q, int kfirst = -1, int nk = 2) {
    const int k = 2 * q + 1;
    if (kfirst < 0) kfirst = k - 1;
    hipStreamWaitEvent(s2, ER[k], 0);
    if (syrk_ev) hipEventRecord(syrk_ev[2 * ns], s2);
    const SyrkShape sh((int)tri(nbk - k - 3), bt.tuning().syrk_slots);
    launch_bulk<T>(A, N, kfirst, nk, k + 3, sh, nbk - k - 3, info, s2, bt);
    if (syrk_ev) hipEventRecord(syrk_ev[2 * ns + 1], s2);
    flop += (double)nk * tile_flop * tri(nbk - k - 3);
    ++ns;
    chain_signal_kernel<<<1, 64, 0, s2>>>(flags + 1, q + 1);  // bulk(q) finished
    hipEventRecord(U[q], s2);
    last_u = q;
  };
  if (k_start == 0) {
    // The rest stream's first trsm polls info[CHAIN_DIAG_DONE], which still holds the LAST step's final value until this
    // factorisation's first diagonal-block kernel resets the status words: clear it here, on s, and let c2 start behind
    // that (one event per factorisation; without it trsm(0) could run ahead of diag(0) on stale data)
    chain_signal_kernel<<<1, 64, 0, s>>>(info + CHAIN_DIAG_DONE, 0, info);  // (and info[0] / info[CHAIN_ABORT]: a stale time-out would make the waiters leave)
    hipEventRecord(ED[0], s);
    hipStreamWaitEvent(c2, ED[0], 0);
    snap_copy_kernel<T><<<8, 256, 0, s>>>(A, N, 1, 0, snap + (long)NB * NB);  // rows of block 1 of panel 0, before their trsm
  } else {
    // hand-over from the group schedule after its chain(q0 - 1): what R(k_start - 1) would have done except its trsm -- the
    // chain's own two columns <- the last group (G_old panels), the snapshot for crit(k_start + 1) -- then the counters and
    // that group's bulk update of the columns from k_start + 2 on
    const int k = k_start - 1, q0 = k_start / G_old;
    hipEvent_t* Uold = ev + nbk;  // the group schedule's U events (its P events are ev[0 .. q0))
    hipEventRecord(ED[k_start], s);  // any free slot of ED: the chain of the pair schedule ends here
    hipStreamWaitEvent(c2, ED[k_start], 0);
    if (q0 >= 2) {
      hipStreamWaitEvent(c2, Uold[q0 - 2], 0);
      hipStreamWaitEvent(s2, Uold[q0 - 2], 0);
    }
    ChainJobs jobs;
    jobs.j[0] = ChainJob{k + 1, k + 2, k_start - G_old, G_old, k + 2};
    jobs.j[1] = ChainJob{k + 2, k + 2, k_start - G_old, G_old, -1};
    chain_col_kernel<T><<<dim3(2 * (nbk - k - 2), 2, 2), 256, 0, c2>>>(A, N, nbk, jobs, snap + (long)((k + 2) & 1) * NB * NB, info, k + 1);
    hipEventRecord(ER[k], c2);
    chain_signal_kernel<<<1, 64, 0, s2>>>(flags + 1, k_start / 2 - 1);  // the group schedule's bulk launches are through
    if (k + 3 < nbk) bulk(k_start / 2 - 1, k_start - G_old, G_old);
  }
  for (int k = k_start; k < nbk; ++k) {
    // ---- critical stream
    if (k >= 1) {
      // crit(k) needs R(k-2) and, for an even block, bulk(k/2 - 2): checked inside the kernel (flags), not by stream waits
      const int need_rest = k >= 2 ? k - 1 : 0, need_bulk = ((k & 1) == 0 && k >= 4 && bulk_exists(k / 2 - 2)) ? k / 2 - 1 : 0;
      const bool handed = k == k_start && k_start > 0;  // every pending panel of this block was solved by the group schedule
      const int nfinal = handed ? G_old : ((k & 1) ? 0 : 1), first_final = handed ? k_start - G_old : k - 2;
      crit_kernel<T><<<10, 256, cbytes, s>>>(A, N, k, Tinv, snap + (long)(k & 1) * NB * NB, nfinal, first_final, handed ? 0 : 1, info,
                                            need_rest, need_bulk);
    }
    // diag(k) publishes info[CHAIN_DIAG_DONE] = k + 1 itself; the rest stream's trsm(k) is launched right away and polls
    // that word: no event record / stream wait between the critical kernels (each costs the stream ~3 us)
    launch_diag<T>(A, N, (long)k * NB, Tinv, logdet, info, s, bt, k == 0, CHAIN_DIAG_DONE, k + 1);
    // trsm(k) polls a word that diag(k) publishes: never enqueue the waiter behind a producer whose launch failed
    // (both critical kernels need opted-in dynamic LDS) -- that would turn an error code into a 2 s stall
    if (const hipError_t le = hipGetLastError(); le != hipSuccess) return (int)le;
    if (k + 1 >= nbk) break;
    // ---- rest stream
    trsm_kernel<T><<<dim3(2 * (nbk - k - 1), 1, 1), 256, 0, c2>>>(A, Tinv, N, k, 0, info, k + 1);
    const int q = k >> 1;
    ChainJobs jobs;
    if ((k & 1) == 0) {  // first panel of pair q: column k+1 (below its diagonal block) <- panel k; snapshot block row k+2
      jobs.j[0] = ChainJob{k + 1, k + 2, k, 1, k + 2};
      jobs.j[1] = jobs.j[0];
      if (k + 2 < nbk)
        chain_col_kernel<T><<<dim3(2 * (nbk - k - 2), 2, 1), 256, 0, c2>>>(A, N, nbk, jobs, snap + (long)((k + 2) & 1) * NB * NB, info, k + 1);
    } else {             // second panel: columns k+1 (below its diagonal block; snapshot block row k+2) and k+2 (whole) <- pair q
      if (q >= 1) hipStreamWaitEvent(c2, U[q - 1], 0);
      jobs.j[0] = ChainJob{k + 1, k + 2, k - 1, 2, k + 2};
      jobs.j[1] = ChainJob{k + 2, k + 2, k - 1, 2, -1};
      if (k + 2 < nbk)
        chain_col_kernel<T><<<dim3(2 * (nbk - k - 2), 2, 2), 256, 0, c2>>>(A, N, nbk, jobs, snap + (long)((k + 2) & 1) * NB * NB, info, k + 1);
    }
    if (k + 2 >= nbk) chain_signal_kernel<<<1, 64, 0, c2>>>(flags, k + 1);  // R(k) finished (otherwise chain_col's last workgroup says so)
    hipEventRecord(ER[k], c2);
    checkpoint(k + 1, c2);
    if ((k & 1) == 1 && k + 3 < nbk) bulk(q);  // released behind the chain's own updates
  }
  // every R(k) was waited for by crit(k+2) except the last one; every bulk launch by the crit of a later even block or not at all
  if (nbk >= 2) hipStreamWaitEvent(s, ER[nbk - 2], 0);
  if (last_u >= 0) hipStreamWaitEvent(s, U[last_u], 0);  // s2 runs its launches in order
  checkpoint(nbk, s);
  if (n_syrk) *n_syrk = ns;
  if (syrk_flop) *syrk_flop = flop;
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// The panel chain of W consecutive block columns starting at block column k0, on a matrix addressed through
// (A, ld) -- the owner's step of the DISTRIBUTED factorisation (dgp_dist.hip), where A points into a rank's column
// slab such that A[r * ld + c] is the global element (r, c) for the columns of this group: every kernel of the chain
// only touches columns of the group, so the single-GPU kernels apply unchanged.  Tinv likewise (L_kk^-1 lands in
// its diagonal blocks).
template <typename T>
int potrf_group(T* A, long ld, int nbk, T* Tinv, T* logdet, int* info, int k0, int W, hipStream_t s) {
  const Batch bt;
  for (int h = 0; h < W && k0 + h < nbk; ++h) {
    const int k = k0 + h;
    if (h >= 1) syrk_col_kernel<T><<<dim3(2 * (nbk - k), 2, 1), 256, 0, s>>>(A, ld, k0, h, k, nbk, 1, 0);
    launch_diag<T>(A, ld, (long)k * NB, Tinv, logdet, info, s, bt);
    if (k + 1 < nbk) trsm_kernel<T><<<dim3(2 * (nbk - k - 1), 1, 1), 256, 0, s>>>(A, Tinv, ld, k, 0);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// trtri level with half-size m (in tiles of BT rows): group g covers tile rows/cols [lo, hi), split at mid.
//   W[i,j] = sum_{c=j}^{mid-1} L[i,c] T[c,j]      (i in [mid,hi), j in [lo,mid))
//   T[i,j] = - sum_{c=mid}^{i} T[i,c] W[c,j]
// BT = 128 for the large levels; the small levels (few tiles, short k) use 64x64 tiles so that four
// times as many workgroups share the work.
template <typename T, int STEP, int BT, bool DMA = true>
__device__ __forceinline__ void trtri_tile(const T* __restrict__ L, T* __restrict__ Tm, T* __restrict__ W, long ld,
                                           int m, int lo, int mid, int hi, int tile, T* __restrict__ smem) {
  using K = TileCore<T, true, false, BT, BT, Prefetch<T>::TRTRI, DMA, true>;  // interleaved groups: balanced zero-work skipping
  using G = typename K::G;
  constexpr int KT = BT / 16;  // k-tiles per tile of the reduction dimension
  // A group that the matrix edge cuts off has R = hi - mid < m tile rows: its R m tiles are the FIRST tile indices, so
  // that they are dealt over all XCDs (workgroup ids go round-robin over the 8 XCDs: with the full groups' i = mid + tile % m
  // map, a group with one tile row -- 2^k + 1 block columns -- would run on every m-th id only, i.e. on ONE XCD when
  // 8 | m: measured 126 ms instead of 5 for the top level of 65 block columns x 32 sites).
  const int R = min(m, hi - mid);
  if (R <= 0 || tile >= R * m) return;
  // longest k-range first: W-step K ~ (mid - j), T-step K ~ (i - mid + 1)
  const int i = STEP == 0 ? mid + tile % R : mid + (R - 1 - tile / m);
  const int j = STEP == 0 ? lo + tile / R : lo + tile % m;
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  // Both products end their k-range in a diagonal block of the triangular T (round 4): the W-step walks k DOWNWARDS from
  // mid - 1 to j (T[c, j] decays away from the diagonal: small-to-large, and the triangular T[j, j] comes last: columns
  // jcol > k are structurally zero), the T-step upwards from mid to i (T[i, i] last: rows irow < k are zero).  The
  // direct-to-LDS core leaves those MFMAs out (TriMode); the register-staged 64-tiles compute them (same results).
  if (STEP == 0) {
    T* out = W + (long)i * BT * ld + (long)j * BT;
    K::template run_tri<true, TRI_COL_LE>(L + (long)i * BT * ld + (long)j * BT, ld, Tm + (long)j * BT * ld + (long)j * BT, ld, (mid - j) * KT, smem, acc,
                                          [&]() { K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; }); });
  } else {
    T* out = Tm + (long)i * BT * ld + (long)j * BT;
    K::template run_tri<false, TRI_ROW_GE>(Tm + (long)i * BT * ld + (long)mid * BT, ld, W + (long)mid * BT * ld + (long)j * BT, ld, (i - mid + 1) * KT, smem, acc,
                                           [&]() { K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = -v; }); });
  }
}

// QUEUE = false: one workgroup per tile.
// QUEUE = true: the EARLY launches that share the GPU with the factorisation's panel chain.  Few persistent
// workgroups (about one per CU) pull (group, tile) pairs from a queue (ctr[0]); a workgroup that finds itself on a
// RESERVED compute unit (table indexed by the hardware CU id) leaves at once, so those CUs stay free for the
// latency-critical chain kernels -- co-resident GEMM waves slow the diagonal-block kernel 2-5x.  Correctness does
// not depend on where workgroups land: a reserved-CU workgroup only leaves once some other workgroup has
// announced (ctr[1]) that it will drain the queue; until then it works like any other.
// (the queue-driven launches keep the register-staged core at two workgroups per CU: they share the GPU with the panel chain,
// whose kernels -- 228 registers, 97-110 KB of LDS -- only fit a CU that holds at most one GEMM workgroup; with three
// 168-register workgroups per CU they wait for two to drain instead of one.  n = 8192, one site: fit step 12.4 ms with
// the direct-to-LDS core here, 11.9 with the register-staged one)
template <typename T, int STEP, int BT, bool QUEUE>
__global__ __launch_bounds__(256, (TileCore<T, true, false, BT, BT, 1, !QUEUE>::OCC)) void trtri_level_kernel(const T* __restrict__ L, T* __restrict__ Tm,
                                                             T* __restrict__ W, long ld, int m, int ntile, int g0,
                                                             int ngroups, int* __restrict__ ctr,
                                                             const unsigned char* __restrict__ resv, long bs) {
  __shared__ T smem[TileCore<T, true, false, BT, BT, 1, !QUEUE>::SMEM_ELEMS];
  if (!QUEUE) {
    L = site(L, bs);  // the queue-driven early launches are single-site only
    Tm = site(Tm, bs);
    W = site(W, bs);
    const int lo = 2 * m * (g0 + (int)blockIdx.y), mid = lo + m, hi = min(lo + 2 * m, ntile);
    trtri_tile<T, STEP, BT, true>(L, Tm, W, ld, m, lo, mid, hi, (int)blockIdx.x, smem);
  } else {
    __shared__ int next_item;
    const int nitems = m * m * ngroups;
    const bool on_reserved = resv != nullptr && resv[__smid() & (SMID_TABLE - 1)] != 0;
    if (threadIdx.x == 0 && !on_reserved) atomicAdd(&ctr[1], 1);
#pragma unroll 1
    for (;;) {
      if (threadIdx.x == 0) {
        int it = nitems;
        if (!on_reserved || __hip_atomic_load(&ctr[1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0)
          it = atomicAdd(&ctr[0], 1);
        next_item = it;
      }
      __syncthreads();
      const int it = __builtin_amdgcn_readfirstlane(next_item);
      __syncthreads();
      if (it >= nitems) return;  // every workgroup gets here once the queue is empty
      const int g = it % ngroups, tile = it / ngroups;  // tile-major: the long-k tiles of every group first
      const int lo = 2 * m * (g0 + g), mid = lo + m, hi = min(lo + 2 * m, ntile);
      trtri_tile<T, STEP, BT, false>(L, Tm, W, ld, m, lo, mid, hi, tile, smem);
    }
  }
}

// ---- reserved compute units ------------------------------------------------------------------
// A probe launch lists the hardware ids (__smid: XCC | SE | CU) of the CUs this process can use; every
// (count / nreserve)-th one, in id order, is marked in a small per-device table (spread over XCDs and SEs).
__global__ void smid_probe_kernel(unsigned* out) {
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < 2000) __builtin_amdgcn_s_sleep(8);  // ~20 us: long enough to fill every CU
  if (threadIdx.x == 0) out[blockIdx.x] = __smid();
}

const unsigned char* reserved_cu_table(int nreserve, int* n_cu) {
  struct Entry {
    int built;
    unsigned char* table;
    int ncu, nres;
  };
  static Entry cache[64];
  static std::mutex mtx;  // plans on several host threads may reach their first fit together
  std::lock_guard<std::mutex> lock(mtx);
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return nullptr;
  Entry& e = cache[dev];
  if (!e.built) {
    e.built = 1;
    const int nprobe = 8192;
    unsigned* d = nullptr;
    if (hipMalloc(&d, nprobe * sizeof(unsigned)) != hipSuccess) return nullptr;
    smid_probe_kernel<<<nprobe, 64>>>(d);
    unsigned* h = new unsigned[nprobe];
    const bool ok = hipMemcpy(h, d, nprobe * sizeof(unsigned), hipMemcpyDeviceToHost) == hipSuccess;
    (void)hipFree(d);
    unsigned char host[SMID_TABLE];
    memset(host, 0, sizeof(host));
    int ncu = 0;
    if (ok)
      for (int i = 0; i < nprobe; ++i)
        if (h[i] < (unsigned)SMID_TABLE && !host[h[i]]) {
          host[h[i]] = 1;
          ++ncu;
        }
    delete[] h;
    e.ncu = ncu;
    if (nreserve > 0 && ncu >= 2 * nreserve) {
      const int stride = ncu / nreserve;
      int rank = 0, nres = 0;
      for (int id = 0; id < SMID_TABLE; ++id) {
        if (!host[id]) continue;
        host[id] = (rank % stride == stride / 2 && nres < nreserve) ? 2 : 1;
        nres += host[id] == 2;
        ++rank;
      }
      for (int id = 0; id < SMID_TABLE; ++id) host[id] = host[id] == 2;
      e.nres = nres;
      if (hipMalloc(&e.table, SMID_TABLE) != hipSuccess ||
          hipMemcpy(e.table, host, SMID_TABLE, hipMemcpyHostToDevice) != hipSuccess)
        e.table = nullptr;
    }
  }
  if (n_cu) *n_cu = e.ncu;
  return e.table;
}

// groups [g0, g1) of the level with half-size mblk (in 128-blocks), one step
// `early` != null: queue-driven launch of at most early->wg_cap workgroups that keeps the reserved CUs free
struct EarlyLaunch {
  int wg_cap;
  int* ctr;                   // next free (queue, workers) counter pair, zeroed at the start of the factorisation
  int pairs_left;
  const unsigned char* resv;  // reserved-CU table or null
};

template <typename T, int STEP>
static void trtri_level(const T* L, T* Tm, T* W, long N, long ld, int mblk, int g0, int g1, EarlyLaunch* early,
                        hipStream_t s, Batch bt) {
  if (g1 <= g0) return;
  const int ng = g1 - g0;
  // fewer than ~two rounds (1024) of 128^2 tiles would leave the GPU waiting on the longest one: then use 64^2 tiles
  // (four times as many, at ~87 % of the big tile's rate); a batch multiplies the tile count
  // (counting the tiles that exist: the last group of a level is cut off at the matrix edge -- nbk = 40 has a top
  // level of 8 x 32 blocks, not 32 x 32)
  const int nbk128 = (int)(N / NB);
  long tiles = 0;
  for (int g = g0; g < g1; ++g) {
    const int rows = nbk128 - (2 * mblk * g + mblk);
    tiles += (long)(rows < mblk ? (rows > 0 ? rows : 0) : mblk) * mblk;
  }
  const bool small = tiles * bt.B < bt.tuning().trtri_small;
  const int m = small ? 2 * mblk : mblk, ntile = (int)(N / (small ? 64 : 128));
  const bool queue = early != nullptr && early->pairs_left > 0 && m * m * ng > early->wg_cap;
  if (!queue) {
    const dim3 grid((unsigned)(m * m), (unsigned)ng, (unsigned)bt.B);
    if (small)
      trtri_level_kernel<T, STEP, 64, false><<<grid, 256, 0, s>>>(L, Tm, W, ld, m, ntile, g0, ng, nullptr, nullptr, bt.ws);
    else
      trtri_level_kernel<T, STEP, 128, false><<<grid, 256, 0, s>>>(L, Tm, W, ld, m, ntile, g0, ng, nullptr, nullptr, bt.ws);
    return;
  }
  int* ctr = early->ctr;
  early->ctr += 2;
  --early->pairs_left;
  const unsigned grid = (unsigned)early->wg_cap;
  if (small) trtri_level_kernel<T, STEP, 64, true><<<grid, 256, 0, s>>>(L, Tm, W, ld, m, ntile, g0, ng, ctr, early->resv, 0);
  else trtri_level_kernel<T, STEP, 128, true><<<grid, 256, 0, s>>>(L, Tm, W, ld, m, ntile, g0, ng, ctr, early->resv, 0);
}

// Launch every step of the level recursion that the first `ready` block columns of L (and the diagonal blocks
// of T that the factorisation wrote with them) allow and that `st` has not seen yet.  A group [lo, hi) split
// at mid needs hi <= ready for its T-step, but only mid <= ready for its W-step (W = L21 T11 reads columns
// < mid of L, which are final for ALL rows once their panels are): so while the factorisation is still
// working down its sequential tail, most of the inverse's GEMM work can already run on the idle CUs.
// All launches of one TrtriProgress must go to the same stream (level order = stream order).  wg_cap > 0 limits
// the workgroups per launch (persistent, queue-driven, see trtri_level_kernel): early launches share the GPU with
// the panel chain; `ctr` = nctr_pairs zeroed (queue, workers) int pairs, `reserve_cus` CUs are left to the chain.
template <typename T>
int trtri_advance(const T* L, long N, T* Tm, T* W, int ready, TrtriProgress* st, hipStream_t s, int wg_cap, int* ctr,
                  int nctr_pairs, int reserve_cus, Batch bt, long ld) {
  if (ld <= 0) ld = N;
  const int nbk = (int)(N / NB);
  if (ready > nbk) ready = nbk;
  EarlyLaunch el{wg_cap, ctr + 2 * st->pairs_used, nctr_pairs - st->pairs_used,
                 reserve_cus > 0 ? reserved_cu_table(reserve_cus, nullptr) : nullptr};
  EarlyLaunch* early = (wg_cap > 0 && ctr != nullptr && bt.B == 1) ? &el : nullptr;
  int lvl = 0;
  for (int m = 1; m < nbk && lvl < TrtriProgress::MAXLVL; m *= 2, ++lvl) {
    const int ngroups = (nbk + 2 * m - 1) / (2 * m);
    int full = 0;  // groups whose whole range [lo, hi) is final
    while (full < ngroups && (2 * m * (full + 1) < nbk ? 2 * m * (full + 1) : nbk) <= ready) ++full;
    int wcan = full;  // groups whose W-step can run
    if (wcan < ngroups && 2 * m * wcan + m <= ready) ++wcan;
    trtri_level<T, 0>(L, Tm, W, N, ld, m, st->wdone[lvl], wcan, early, s, bt);
    trtri_level<T, 1>(L, Tm, W, N, ld, m, st->gdone[lvl], full, early, s, bt);
    if (wcan > st->wdone[lvl]) st->wdone[lvl] = wcan;
    if (full > st->gdone[lvl]) st->gdone[lvl] = full;
  }
  if (early) st->pairs_used = nctr_pairs - el.pairs_left;
  return (int)hipGetLastError();
}

template <typename T>
static void trtri_group(const T* L, T* Tm, T* W, long N, int k0, int G, hipStream_t s, Batch bt) {
  for (int m = 1; m < G; m *= 2) {  // level with half-size m: the groups of 2 m blocks inside [k0, k0 + G)
    const int g0 = k0 / (2 * m), g1 = (k0 + G) / (2 * m);
    trtri_level<T, 0>(L, Tm, W, N, N, m, g0, g1, nullptr, s, bt);
    trtri_level<T, 1>(L, Tm, W, N, N, m, g0, g1, nullptr, s, bt);
  }
}

template <typename T>
int trtri(const T* L, const T* /*Dinv: already the diagonal blocks of Tm*/, long N, T* Tm, T* W, hipStream_t s, Batch bt,
          long ld) {
  TrtriProgress st;
  return trtri_advance<T>(L, N, Tm, W, (int)(N / NB), &st, s, 0, nullptr, 0, 0, bt, ld);
}

// ------------------------------------------------------------------------------------------
// S[i,j] = sum_{c >= i} T[c,i]^T T[c,j]   (i >= j): K^^-1 = L^-T L^-1
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, false, false>::OCC)) void lauum_kernel(const T* __restrict__ Tm, T* __restrict__ S, long ld,
                                                                                      int nbk, long bs, int super = 0) {
  Tm = site(Tm, bs);
  S = site(S, bs);
  using K = TileCore<T, false, false, 128, 128, Prefetch<T>::LAUUM, true, true>;  // interleaved groups (TriSpec skipping)
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  int bi, bj;
  if (super > 0) {
    // EXPERIMENT (Tuning::lauum_super): super x super supertiles, dealt round-robin over the 8 XCDs in row order (long
    // k-ranges first, every XCD gets the same mix); workgroup b runs on XCD b % 8 and is that XCD's (b / 8)-th tile
    const int x = (int)blockIdx.x & 7, j = (int)blockIdx.x >> 3, s2 = super * super;
    const int nsr = (nbk + super - 1) / super, st = (j / s2) * 8 + x;
    if (st >= nsr * (nsr + 1) / 2) return;
    int R, C;
    tri_decode(st, R, C);
    const int l = j % s2;
    bi = R * super + l / super;
    bj = C * super + l % super;
    if (bi >= nbk || bj > bi) return;
  } else {
    tri_decode(blockIdx.x, bi, bj);  // ascending bi: the long-K tiles are dispatched first
  }
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* base = Tm + (long)bi * NB * ld;
  // k from the LAST row block up to the diagonal block (REV): the rows of L^-1 far below the diagonal hold its smallest
  // entries, and thousands of tiny products added to a sum that already holds the diagonal block's large ones are
  // rounded away in fp32 -- small-to-large keeps them (dgp_gemm.h::run)
  // ... and the diagonal block T[bi, bi] comes LAST: rows i > k of it are structurally zero (operand A); of a diagonal tile
  // only the 16 x 16 sub-tiles on and below the diagonal are needed (the others hold unspecified values: nothing reads S
  // above its diagonal blocks' diagonal): the direct-to-LDS core leaves those MFMAs out (TriMode, dgp_gemm_dma.h)
  T* out = S + (long)bi * NB * ld + (long)bj * NB;
  auto store = [&]() { K::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; }); };
  if (bi == bj) K::template run_tri<true, TRI_LOWER>(base + (long)bi * NB, ld, base + (long)bj * NB, ld, (nbk - bi) * (NB / 16), smem, acc, store);
  else K::template run_tri<true, TRI_ROW_LE>(base + (long)bi * NB, ld, base + (long)bj * NB, ld, (nbk - bi) * (NB / 16), smem, acc, store);
}

// the same product in 64 x 64 tiles: four times the workgroups, for matrices whose 128 x 128 tiles do not fill the CUs
template <typename T>
__global__ __launch_bounds__(256, 2) void lauum64_kernel(const T* __restrict__ Tm, T* __restrict__ S, long ld, int nb64,
                                                         long bs) {
  Tm = site(Tm, bs);
  S = site(S, bs);
  using G = TileGemm<T, false, false, 64, 64>;
  __shared__ T smem[G::SMEM_ELEMS];
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  const T* base = Tm + (long)bi * 64 * ld;
  G::template run<1, true>(base + (long)bi * 64, ld, base + (long)bj * 64, ld, (nb64 - bi) * (64 / 16), smem, acc);
  T* out = S + (long)bi * 64 * ld + (long)bj * 64;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * ld + c] = v; });
}

template <typename T>
int lauum(const T* Tm, long N, T* S, hipStream_t s, Batch bt) {
  const int nbk = (int)(N / NB);
  const int tiles = nbk * (nbk + 1) / 2;
  if ((long)tiles * bt.B <= bt.tuning().lauum64_max_tiles) {  // fewer 128-tiles than CUs
    const int nb64 = 2 * nbk;
    lauum64_kernel<T><<<dim3((unsigned)(nb64 * (nb64 + 1) / 2), 1, (unsigned)bt.B), 256, 0, s>>>(Tm, S, N, nb64, bt.ws);
  } else {
    const int sup = bt.tuning().lauum_super;
    unsigned grid = (unsigned)tiles;
    if (sup > 0) {
      const int nsr = (nbk + sup - 1) / sup, nst = nsr * (nsr + 1) / 2;
      grid = (unsigned)((nst + 7) / 8 * 8 * sup * sup);
    }
    lauum_kernel<T><<<dim3(grid, 1, (unsigned)bt.B), 256, 0, s>>>(Tm, S, N, nbk, bt.ws, sup);
  }
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// z_i = sum_{j <= i} T[i][j] r_j : one wave per row (coalesced along j)
template <typename T>
__global__ __launch_bounds__(256) void trmv_n_kernel(const T* __restrict__ Tm, long ld, const T* __restrict__ r, int n,
                                                     T* __restrict__ z, long bs, const int* __restrict__ ns, long rs) {
  Tm = site(Tm, bs);
  z = site(z, bs);
  r = site(r, rs);  // site stride of r: n for the caller's residuals, the scratch stride for the refinement's rho
  n = site_n(ns, n);
  const int lane = threadIdx.x & 63;
  const long i = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long jend = (i / NB + 1) * NB;  // the diagonal block is zero above the diagonal
  const T* row = Tm + i * ld;
  T acc = T(0);
  for (long j = lane; j < jend; j += 64) acc += row[j] * (j < n ? r[j] : T(0));
  acc = wave_sum(acc);
  if (lane == 0) z[i] = acc;
}

// partial[c][j] = sum_{i in chunk c, i >= blockrow(j)} T[i][j] z_i  (64 columns per workgroup).  Row chunks of 512 for
// large matrices; 128 below N = 2048, where the 4 x N/128 dependent loads per thread of a 512-chunk are the whole
// latency of the kernel (N = 384: 19 -> 6 us)
#define DGP_TRMV_CHUNK 512
static inline int trmv_chunk(long N) { return N < 2048 ? 128 : DGP_TRMV_CHUNK; }
template <typename T>
__global__ __launch_bounds__(256) void trmv_t_kernel(const T* __restrict__ Tm, long ld, const T* __restrict__ z,
                                                     T* __restrict__ partial, long bs, int chunk) {
  Tm = site(Tm, bs);
  z = site(z, bs);
  partial = site(partial, bs);
  __shared__ T red[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + tx;
  const long r0 = (long)blockIdx.y * chunk, r1 = min(r0 + (long)chunk, ld);
  const long rstart = max(r0, ((long)blockIdx.x * 64 / NB) * NB);
  T acc = T(0);
  for (long i = rstart + ty; i < r1; i += 4) acc += Tm[i * ld + j] * z[i];
  red[ty][tx] = acc;
  __syncthreads();
  if (ty == 0) partial[(long)blockIdx.y * ld + j] = red[0][tx] + red[1][tx] + red[2][tx] + red[3][tx];
}

// alpha[j] = sum over the row chunks; the LAST workgroup instead computes quad = z^T z (one launch less; quad == null:
// the refinement's second solve, whose quadratic form comes from refine_finish_kernel)
template <typename T>
__global__ __launch_bounds__(256) void trmv_t_reduce_kernel(const T* __restrict__ partial, long N, int nchunks,
                                                            T* __restrict__ alpha, long bs, int chunk,
                                                            const T* __restrict__ z, T* __restrict__ quad, long as = -1) {
  if (blockIdx.x == gridDim.x - 1) {
    if (quad == nullptr) return;
    z = site(z, bs);
    quad = site(quad, bs);
    // accumulated in double whatever T is (fp32 plans: the sum of up to 2^20 squares would otherwise lose ~3 digits);
    // fp32 plans also keep the unrounded value in the scalar block's second double slot (quad = scal + 1 -> scal + 4)
    __shared__ double red[256];
    double acc = 0.0;
    for (long i = threadIdx.x; i < N; i += 256) acc += (double)z[i] * (double)z[i];
    red[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
      if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
      __syncthreads();
    }
    if (threadIdx.x == 0) {
      quad[0] = (T)red[0];
      if (sizeof(T) == 4) *reinterpret_cast<double*>(quad + 3) = red[0];
    }
    return;
  }
  partial = site(partial, bs);
  alpha = site(alpha, as >= 0 ? as : bs);
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T acc = T(0);
  const int c0 = (int)((j / NB) * NB / chunk);
  for (int c = c0; c < nchunks; ++c) acc += partial[(long)c * N + j];
  alpha[j] = acc;
}

// beta = S g for a symmetric matrix stored in its lower triangle (nothing above the diagonal is read):
// row part  sum_{j <= i} S[i][j] g_j  +  column part  sum_{i > j} S[i][j] g_i
template <typename T>
__global__ __launch_bounds__(256) void symv_row_kernel(const T* __restrict__ S, long ld, const T* __restrict__ g,
                                                       T* __restrict__ out, long bs, long wbs) {
  S = site(S, bs);
  g = site(g, wbs);
  out = site(out, wbs);
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
                                                       T* __restrict__ partial, long bs, long wbs) {
  S = site(S, bs);
  g = site(g, wbs);
  partial = site(partial, wbs);
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
                                                          T* __restrict__ dnoise, long bs, long wbs, const int* __restrict__ ns) {
  partial = site(partial, wbs);
  beta = site(beta, wbs);
  alpha = site(alpha, bs);
  if (dnoise) dnoise = site(dnoise, (long)n);
  const int nfull = n;
  n = site_n(ns, n);
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= N) return;
  T acc = beta[j];  // row part
  for (int c = (int)(j / DGP_TRMV_CHUNK); c < nchunks; ++c) acc += partial[(long)c * N + j];
  acc = j < n ? acc : T(0);
  beta[j] = acc;
  if (j < nfull && dnoise) dnoise[j] = j < n ? -acc * alpha[j] : T(0);
}

template <typename T>
int symv_lower(const T* S, long N, const T* g, int n, const T* alpha, T* beta, T* partials, T* dnoise, hipStream_t s,
               Batch bt, long wbs) {
  const int nchunks = (int)((N + DGP_TRMV_CHUNK - 1) / DGP_TRMV_CHUNK);
  const unsigned Bz = (unsigned)bt.B;
  symv_row_kernel<T><<<dim3((unsigned)(N / 4), 1, Bz), 256, 0, s>>>(S, N, g, beta, bt.ws, wbs);
  dim3 grid((unsigned)(N / 64), (unsigned)nchunks, Bz);
  symv_col_kernel<T><<<grid, 256, 0, s>>>(S, N, g, partials, bt.ws, wbs);
  symv_finish_kernel<T><<<dim3((unsigned)((N + 255) / 256), 1, Bz), 256, 0, s>>>(partials, N, nchunks, alpha, n, beta, dnoise, bt.ws,
                                                                                wbs, bt.ns);
  return (int)hipGetLastError();
}

long solve_partials(long N) { return (N + trmv_chunk(N) - 1) / trmv_chunk(N) * N; }

template <typename T>
int solve(const T* Tm, long N, const T* r, int n, T* z, T* alpha, T* partials, T* quad, hipStream_t s, Batch bt) {
  const unsigned Bz = (unsigned)bt.B;
  trmv_n_kernel<T><<<dim3((unsigned)(N / 4), 1, Bz), 256, 0, s>>>(Tm, N, r, n, z, bt.ws, bt.ns, (long)n);
  const int chunk = trmv_chunk(N), nchunks = (int)((N + chunk - 1) / chunk);
  dim3 grid((unsigned)(N / 64), (unsigned)nchunks, Bz);
  trmv_t_kernel<T><<<grid, 256, 0, s>>>(Tm, N, z, partials, bt.ws, chunk);
  trmv_t_reduce_kernel<T><<<dim3((unsigned)((N + 255) / 256) + 1, 1, Bz), 256, 0, s>>>(partials, N, nchunks, alpha, bt.ws, chunk,
                                                                                   z, quad);
  return (int)hipGetLastError();
}

// ---- iterative refinement of the fp32 solve (dgp_internal.h::refine_solve) ------------------------------------------
// alpha <- alpha0 + delta (rounded once to T) and quad = sum_i r_i alpha0_i + rho_i (alpha0_i + delta_i) in double: with
// e = K^^-1 r - alpha0 and K^ e = rho,  r^T K^^-1 r = r^T alpha0 + rho^T alpha0 + rho^T e  exactly, and delta ~ e enters
// only through rho^T delta -- an error of delta costs |rho| |error|, not |r| |error| as in r^T (alpha0 + delta).
// One workgroup per site, fixed summation order.
template <typename T>
__global__ __launch_bounds__(256) void refine_finish_kernel(T* __restrict__ alpha, const T* __restrict__ delta,
                                                            const double* __restrict__ rho, const T* __restrict__ r, int n,
                                                            T* __restrict__ quad, long bs, long ps, long rs,
                                                            const int* __restrict__ ns) {
  alpha = site(alpha, bs);
  quad = site(quad, bs);
  delta = site(delta, rs);
  rho = site(rho, ps);
  r = site(r, (long)n);
  n = site_n(ns, n);
  __shared__ double red[256];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 256) {
    const double a0 = (double)alpha[i], a1 = a0 + (double)delta[i];
    acc += (double)r[i] * a0 + rho[i] * a1;
    alpha[i] = (T)a1;
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    quad[0] = (T)red[0];
    if (sizeof(T) == 4) *reinterpret_cast<double*>(quad + 3) = red[0];  // the scalar block's unrounded slot (solve())
  }
}

template <typename T>
int refine_solve(const T* Tm, long N, const T* r, int n, const double* rho64, const T* rho32, T* z, T* delta, T* alpha,
                 T* partials, T* quad, hipStream_t s, Batch bt, long ps, long rs) {
  const unsigned Bz = (unsigned)bt.B;
  trmv_n_kernel<T><<<dim3((unsigned)(N / 4), 1, Bz), 256, 0, s>>>(Tm, N, rho32, n, z, bt.ws, bt.ns, rs);
  const int chunk = trmv_chunk(N), nchunks = (int)((N + chunk - 1) / chunk);
  dim3 grid((unsigned)(N / 64), (unsigned)nchunks, Bz);
  trmv_t_kernel<T><<<grid, 256, 0, s>>>(Tm, N, z, partials, bt.ws, chunk);
  // delta sits in the caller's scratch at site stride rs: the reduce kernel strides its output by `bs`
  trmv_t_reduce_kernel<T><<<dim3((unsigned)((N + 255) / 256) + 1, 1, Bz), 256, 0, s>>>(partials, N, nchunks, delta, bt.ws, chunk,
                                                                                   z, nullptr, rs);
  refine_finish_kernel<T><<<dim3(1, 1, Bz), 256, 0, s>>>(alpha, delta, rho64, r, n, quad, bt.ws, ps, rs, bt.ns);
  return (int)hipGetLastError();
}

// dNLL/dnoise_i = 1/2 (S_ii - alpha_i^2)
template <typename T>
__global__ __launch_bounds__(256) void dnoise_kernel(const T* __restrict__ S, const T* __restrict__ alpha, long N,
                                                     int n, T* __restrict__ dnoise, long bs, const int* __restrict__ ns) {
  S = site(S, bs);
  alpha = site(alpha, bs);
  dnoise = site(dnoise, (long)n);
  const int nb = site_n(ns, n);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dnoise[i] = i < nb ? T(0.5) * (S[i * N + i] - alpha[i] * alpha[i]) : T(0);
}

template <typename T>
int finish(const T* S, const T* alpha, long N, int n, T* dnoise, hipStream_t s, Batch bt) {
  dnoise_kernel<T><<<dim3((unsigned)((n + 255) / 256), 1, (unsigned)bt.B), 256, 0, s>>>(S, alpha, N, n, dnoise, bt.ws, bt.ns);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// prediction: V = T Ks (N x M, Ks = K(X, X*) padded to M % 128 == 0), var_j = kss_j - sum_i V_ij^2,
// mean_j = sum_i Ks_ij alpha_i      (src/discontinuum/engines/gpytorch.py:621-624)
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, true, false>::OCC)) void predict_v_kernel(const T* __restrict__ Tm, long N, const T* __restrict__ Ks,
                                                        long M, T* __restrict__ V, long bs, long wbs) {
  Tm = site(Tm, bs);  // batched plans: blockIdx.z = site; Ks, V in the caller's work area (site stride wbs)
  Ks = site(Ks, wbs);
  V = site(V, wbs);
  using K = TileCore<T, true, false>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int bi = gridDim.y - 1 - blockIdx.y, bj = blockIdx.x;  // longest row blocks first
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  K::run(Tm + (long)bi * NB * N, N, Ks + (long)bj * NB, M, (bi + 1) * (NB / 16), smem, acc);
  T* out = V + (long)bi * NB * M + (long)bj * NB;
  G::foreach (acc, [&](int r, int c, T& v) { out[(long)r * M + c] = v; });
}

// column sums over a slab of rows: part[0][z][j] = sum_i Ks_ij alpha_i, part[1][z][j] = sum_i V_ij^2, i in slab z.
// 64 columns x 4 row lanes per workgroup; (M/64) x PREDICT_SPLIT workgroups keep every CU reading.
template <typename T>
__global__ __launch_bounds__(256) void predict_partial_kernel(const T* __restrict__ V, const T* __restrict__ Ks, long N,
                                                              long M, const T* __restrict__ alpha, T* __restrict__ part, long bs,
                                                              long wbs) {
  V = site(V, wbs);
  Ks = site(Ks, wbs);
  part = site(part, wbs);
  alpha = site(alpha, bs);
  __shared__ T r1[4][64], r2[4][64];
  const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
  const long j = (long)blockIdx.x * 64 + tx;
  const long rows = N / PREDICT_SPLIT;  // N % 128 == 0 and PREDICT_SPLIT divides 128
  const long i0 = (long)blockIdx.y * rows, i1 = i0 + rows;
  T s1 = T(0), s2 = T(0);
  for (long i = i0 + ty; i < i1; i += 4) {
    const T v = V[i * M + j];
    s2 += v * v;
    s1 += Ks[i * M + j] * alpha[i];
  }
  r1[ty][tx] = s1;
  r2[ty][tx] = s2;
  __syncthreads();
  if (ty == 0) {
    part[(long)blockIdx.y * M + j] = r1[0][tx] + r1[1][tx] + r1[2][tx] + r1[3][tx];
    part[((long)PREDICT_SPLIT + blockIdx.y) * M + j] = r2[0][tx] + r2[1][tx] + r2[2][tx] + r2[3][tx];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void predict_finish_kernel(const T* __restrict__ part, long M, const T* __restrict__ kss,
                                                             T* __restrict__ mean, T* __restrict__ var, long wbs) {
  part = site(part, wbs);
  kss = site(kss, wbs);
  mean = site(mean, wbs);
  var = site(var, wbs);
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= M) return;
  T s1 = T(0), s2 = T(0);
  for (int z = 0; z < PREDICT_SPLIT; ++z) {  // fixed order: results do not depend on scheduling
    s1 += part[(long)z * M + j];
    s2 += part[((long)PREDICT_SPLIT + z) * M + j];
  }
  mean[j] = s1;
  var[j] = kss[j] - s2;
}

template <typename T>
int predict_var(const T* Tm, long N, const T* Ks, long M, T* V, const T* alpha, const T* kss, T* part, T* mean, T* var,
                hipStream_t s, Batch bt, long wbs) {
  const unsigned Bz = (unsigned)bt.B;
  dim3 grid((unsigned)(M / NB), (unsigned)(N / NB), Bz);
  predict_v_kernel<T><<<grid, 256, 0, s>>>(Tm, N, Ks, M, V, bt.ws, wbs);
  predict_partial_kernel<T><<<dim3((unsigned)(M / 64), PREDICT_SPLIT, Bz), 256, 0, s>>>(V, Ks, N, M, alpha, part, bt.ws, wbs);
  predict_finish_kernel<T><<<dim3((unsigned)((M + 255) / 256), 1, Bz), 256, 0, s>>>(part, M, kss, mean, var, wbs);
  return (int)hipGetLastError();
}

// cov[i,j] = Kss[i,j] - sum_k V[k,i] V[k,j]  (i >= j tiles), k over all N rows of V (N x M)
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, false, false>::OCC)) void posterior_cov_kernel(const T* __restrict__ V, long N, long M, T* __restrict__ cov, long wbs) {
  using K = TileCore<T, false, false>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  V = site(V, wbs);  // batched plans: V in the caller's work area, cov [batch][M][M]
  cov = site(cov, M * M);
  int bi, bj;
  tri_decode(blockIdx.x, bi, bj);
  typename G::acc_t acc[G::MI][G::NI];
  T* C = cov + (long)bi * NB * M + (long)bj * NB;
  typename G::acc_t keep[G::MI][G::NI];
  trailing_begin<T, G, K::DMA>(acc, keep, C, M);
  K::run(V + (long)bi * NB, M, V + (long)bj * NB, M, (int)(N / 16), smem, acc);
  trailing_end<T, G, K::DMA>(acc, keep, C, M);
}

template <typename T>
int posterior_cov(const T* V, long N, long M, T* cov, hipStream_t s, int B, long wbs) {
  const int nb = (int)(M / NB);
  posterior_cov_kernel<T><<<dim3((unsigned)(nb * (nb + 1) / 2), 1, (unsigned)B), 256, 0, s>>>(V, N, M, cov, wbs);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------
// Draws from N(mean, L L^T) for sample() (src/discontinuum/engines/gpytorch.py:575-580, `f_preds.sample`):
//     out[q][j] = mean[j] + sum_{k <= j} L[j][k] Z[k][q]
// as the transposed product Z^T L^T, so that the draw-major output rows are written contiguously: operand A = Z read
// transposed (op(q, k) = Z[k ldz + q]), operand B = L by rows (op(j, k) = L[j ld + k]); the k-range of point block bj
// stops at its diagonal block (the factorisation leaves zeros above the diagonal inside it).  Loads need no bounds:
// L is M x M with the identity pad, Z is M x Q with Q % 128 == 0; only the stores are clipped to ndraw x m.
template <typename T>
__global__ __launch_bounds__(256, (TileCore<T, false, true>::OCC)) void sample_draws_kernel(const T* __restrict__ L, long ld, const T* __restrict__ Z,
                                                              long ldz, const T* __restrict__ mean, int m, int ndraw,
                                                              T* __restrict__ out) {
  using K = TileCore<T, false, true>;
  using G = typename K::G;
  __shared__ T smem[K::SMEM_ELEMS];
  const int bq = blockIdx.x, bj = gridDim.y - 1 - blockIdx.y;  // longest k-ranges first
  typename G::acc_t acc[G::MI][G::NI];
  G::zero(acc);
  K::run(Z + (long)bq * NB, ldz, L + (long)bj * NB * ld, ld, (bj + 1) * (NB / 16), smem, acc);
  G::foreach (acc, [&](int r, int c, T& v) {
    const long q = (long)bq * NB + r, j = (long)bj * NB + c;
    if (q < ndraw && j < m) out[q * m + j] = v + (mean ? mean[j] : T(0));
  });
}

template <typename T>
int sample_draws(const T* L, long M, const T* Z, long Q, const T* mean, int m, int ndraw, T* out, hipStream_t s) {
  sample_draws_kernel<T><<<dim3((unsigned)(Q / NB), (unsigned)(M / NB)), 256, 0, s>>>(L, M, Z, Q, mean, m, ndraw, out);
  return (int)hipGetLastError();
}

#define DGP_INST(T)                                                                                              \
  template int posterior_cov<T>(const T*, long, long, T*, hipStream_t, int, long);                                          \
  template int sample_draws<T>(const T*, long, const T*, long, const T*, int, int, T*, hipStream_t);             \
  template int symv_lower<T>(const T*, long, const T*, int, const T*, T*, T*, T*, hipStream_t, Batch, long);     \
  template int potrf<T>(T*, long, T*, T*, int*, int, hipStream_t, hipStream_t, hipEvent_t*, hipEvent_t*, int*, double*, int, \
                        const int*, hipEvent_t*, void (*)(void*, int), void*, Batch, int, PotrfCarry*);                                                                                        \
  template int potrf_split<T>(T*, long, T*, T*, int*, T*, hipStream_t, hipStream_t, hipStream_t, hipEvent_t*, hipEvent_t*, int*, double*, \
                              int, const int*, hipEvent_t*, void (*)(void*, int), void*, int, int, const Tuning*);         \
  template int potrf_group<T>(T*, long, int, T*, T*, int*, int, int, hipStream_t);                                \
  template int trtri_advance<T>(const T*, long, T*, T*, int, TrtriProgress*, hipStream_t, int, int*, int, int, Batch, long);                   \
  template int trtri<T>(const T*, const T*, long, T*, T*, hipStream_t, Batch, long);                                        \
  template int lauum<T>(const T*, long, T*, hipStream_t, Batch);                                                      \
  template int solve<T>(const T*, long, const T*, int, T*, T*, T*, T*, hipStream_t, Batch);                           \
  template int refine_solve<T>(const T*, long, const T*, int, const double*, const T*, T*, T*, T*, T*, T*, hipStream_t, Batch, long, long); \
  template int finish<T>(const T*, const T*, long, int, T*, hipStream_t, Batch);                                      \
  template int predict_var<T>(const T*, long, const T*, long, T*, const T*, const T*, T*, T*, T*, hipStream_t, Batch, long);
DGP_INST(double)
DGP_INST(float)

}  // namespace dgp

#ifdef DGP_DIAG_LOG
// measurement builds only (scripts/diag_in_situ.py): the diagonal-block kernel's in-situ log, 4 x 1024 words
extern "C" int dgp_debug_diag_log(unsigned long long* out_host) {
  return (int)hipMemcpyFromSymbol(out_host, HIP_SYMBOL(dgp::dgp_diag_log), sizeof(unsigned long long) * 4 * 1024);
}
#endif
