// dgp_internal.h -- host-side launcher declarations shared by the .hip translation units.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define DGP_TILE_HOST 128  // == DGP_TILE in dgp_common.h

namespace dgp {

inline long round_up(long n, long q) { return (n + q - 1) / q * q; }

// A batched plan carries B <= DGP_MAX_BATCH_SITES sites in lockstep: every fit-step kernel is launched once with
// gridDim.z = B and finds its site's buffers at blockIdx.z * stride (workspace buffers: `ws` elements of the plan's
// dtype; caller arrays: n or DGP_OUT_LEN).  B = 1 is the plain single-site plan.
#define DGP_MAX_BATCH_HOST 8     // == DGP_MAX_BATCH in dgp_common.h: hyperparameters by value up to here
#define DGP_MAX_BATCH_SITES 1024  // largest batch of a plan
// Tile-shape selectors of the O(n^3) stages.  Plan-level (dgp_plan_set_option) so that a test can force the kernels the
// benchmark shapes run -- lauum_kernel, the 128-tile rounds of the bulk update, the 128-tile inverse levels -- at sizes
// the dense oracle reaches; the defaults are the measured optima (environment overrides read once: DGP_LAUUM64,
// DGP_SYRK_SLOTS, DGP_TRTRI_SMALL).
struct Tuning {
  int lauum64_max_tiles;  // K^^-1 = L^-T L^-1 in 64 x 64 tiles while (128-tiles x batch) <= this (1000)
  int syrk_slots;         // workgroup slots of a bulk-update round: whole rounds in 128 x 128 tiles, the rest cut (512)
  long trtri_small;       // an inverse level with fewer 128-tiles (x batch) than this runs in 64 x 64 tiles (1024)
  int syrk_super;         // tile order of the bulk update: 0 = rows of the trailing matrix, S > 0 = S x S supertiles (experiment)
  int lauum_super;        // tile order of K^^-1 = L^-T L^-1: 0 = rows, S > 0 = S x S supertiles dealt round-robin over the XCDs
  int chain_yield;        // single-site plans: bulk-update waves leave their CU to the diagonal-block kernel while it runs there (1)
  int fused_grad;         // the gradient contraction runs in the epilogue of K^^-1's 128 x 128 tiles (dgp_fused.hip) when it applies (1)
  int group_gemm;         // batched plans: a panel group's rows below its diagonal block are solved by ONE GEMM against the inverted
                          // G x G-block diagonal block instead of G trsm + G - 1 column-update launches (dgp_chol.hip::potrf)
};
const Tuning& default_tuning();
struct Batch {
  int B = 1;
  long ws = 0;
  const int* ns = nullptr;  // device array of the sites' own sizes n_b <= n (ragged batch), or null: all n
  const Tuning* tune = nullptr;  // null: default_tuning()
  void* W = nullptr;             // N x N scratch per site (the plan's S buffer: free during the factorisation), or null: the
                                 // group schedule's panel solve as ONE GEMM against the inverted diagonal group block needs it
  const Tuning& tuning() const { return tune ? *tune : default_tuning(); }
};
int model_ntheta(int model, int d);  // number of constrained kernel hyperparameters, -1 if unsupported
int composite_define(const int* spec, int nspec);  // register a generic composite model (dgp_models.h), -> model id or < 0

// ---- dgp_gram.hip ---------------------------------------------------------------------------
template <typename T>
int pack_x(const T* X, int n, int d, long N, T* Xt, hipStream_t s, Batch bt = Batch());
template <typename T>
int gram_sym(int model, int d, const T* Xt, long N, int n, const double* theta, const T* noise, T* K, hipStream_t s,
             Batch bt = Batch(), void* pre_scratch = nullptr /* pre_scratch_bytes(B) of device memory, B > 8 */,
             void* pre_staging = nullptr /* pinned host memory of the same size that outlives the copy, or null: blocking copy */,
             long k_stride = -1 /* site stride of K (default bt.ws) */, long noise_stride = -1 /* of noise (default n) */,
             bool pre_ready = false /* pre_scratch already holds this theta (gram_cross of the same call) */);
// The inference launchers take a Batch like the fit-step ones (gridDim.z = sites): training-side arrays at the plan's
// site stride bt.ws, everything in the caller's work area (test coordinates, cross Gram, partials ...) at `wbs` elements.
template <typename T>
int gram_cross(int model, int d, const T* Xt, long N, int n, const T* Xst, long M, int m, const double* theta,
               T* Ks, hipStream_t s, Batch bt = Batch(), long wbs = 0, void* pre_scratch = nullptr, void* pre_staging = nullptr);
template <typename T>
int gram_diag(int model, int d, const T* Xst, long M, int m, const double* theta, T* kss, hipStream_t s, Batch bt = Batch(),
              long wbs = 0, void* pre_scratch = nullptr);
template <typename T>
int gram_grad(int model, int d, const T* Xt, long N, int n, const double* theta, const T* S, const T* alpha,
              T* partials, T* dtheta, hipStream_t s, Batch bt = Batch(), long dtheta_stride = 0,
              void* pre_scratch = nullptr, bool pre_ready = false /* gram_sym of this step filled pre_scratch */,
              void* pre_staging = nullptr);
size_t pre_scratch_bytes(int B);  // device scratch for the hyperparameters of a batch of B (0 up to 8)
// rho = r - (K(X, X; theta) + diag(noise)) alpha in DOUBLE, the covariance re-evaluated pair by pair from the stored
// (TS = float) coordinates -- K^ itself was overwritten by its factor.  Lower 64 x 64 tiles only (each tile feeds its
// row block and, transposed, its column block); `part` = (N/64)^2 x 64 doubles of scratch per site (site stride ps
// doubles), rho64 / rho32 N elements per site at strides ps / rs.  Deterministic (fixed summation order).
template <typename TS>
int gram_residual(int model, int d, const TS* Xt, long N, int n, const double* theta, const TS* noise, const TS* r,
                  const TS* alpha, double* part, double* rho64, TS* rho32, hipStream_t s, Batch bt, long ps, long rs,
                  void* pre_scratch, void* pre_staging);
size_t gram_residual_scratch_bytes(long N);  // part + rho64 + rho32 + delta, per site
long gram_grad_partials(long N);
// ---- dgp_fused.hip: K^^-1 = L^-T L^-1 with the gradient contraction in the epilogue of every 128 x 128 tile
bool lauum_grad_applies(int model, long N, const Batch& bt);
template <typename T>
int lauum_grad(int model, int d, const T* Tm, long N, T* S, const T* Xt, int n, const double* theta, const T* alpha, T* partials,
               T* dtheta, hipStream_t s, Batch bt, long dtheta_stride, void* pre_scratch, bool pre_ready, void* pre_staging);
template <typename T>
int mean_vjp_grad(int model, int d, const T* Xt, long N, int n, const T* Xst, long Mp, int m, const double* theta,
                  const T* alpha, const T* beta, const T* wts, T* partials, T* dtheta, hipStream_t s, Batch bt = Batch(),
                  long wbs = 0, long dstride = 0, void* pre_scratch = nullptr, void* pre_staging = nullptr);
template <typename T>
int gemv_rows(const T* Ks, long N, long Mp, int m, const T* w, T* out, hipStream_t s, int B = 1, long wbs = 0);

// ---- dgp_chol.hip ---------------------------------------------------------------------------
struct PotrfCarry {  // state of a factorisation that one schedule hands to the next (potrf with q_stop -> potrf_split)
  int ck_next = 0, ns = 0;
  double flop = 0.0;
};
template <typename T>
int potrf(T* A, long N, T* Dinv, T* logdet, int* info, int lookahead, hipStream_t s, hipStream_t s2, hipEvent_t* ev,
          hipEvent_t* syrk_ev /* 2 per bulk launch, or null */, int* n_syrk, double* syrk_flop,
          int nck = 0, const int* ck_blocks = nullptr /* ascending */, hipEvent_t* ck_ev = nullptr,
          void (*on_ck)(void* ctx, int idx) = nullptr /* called right after checkpoint idx is recorded */,
          void* ck_ctx = nullptr, Batch bt = Batch(), int q_stop = -1 /* stop after chain(q_stop - 1): potrf_split */,
          struct PotrfCarry* carry = nullptr);
// the split panel chain (one site, fewer than 96 block columns): critical tile on `s`, rest of the chain on `c2`,
// bulk updates on `s2`; ev holds 3 N/128 events; snap = 2 x 128 x 128 elements of workspace
template <typename T>
int potrf_split(T* A, long N, T* Dinv, T* logdet, int* info, T* snap, hipStream_t s, hipStream_t c2, hipStream_t s2,
                hipEvent_t* ev, hipEvent_t* syrk_ev, int* n_syrk, double* syrk_flop, int nck = 0,
                const int* ck_blocks = nullptr, hipEvent_t* ck_ev = nullptr, void (*on_ck)(void* ctx, int idx) = nullptr,
                void* ck_ctx = nullptr, int k_start = 0 /* block columns before it by the single-stream group schedule */,
                int G_old = 2 /* panels per group of that schedule */, const Tuning* tune = nullptr);
// progress of the level recursion of trtri when it is issued piecewise (trtri_advance)
struct TrtriProgress {
  static constexpr int MAXLVL = 16;
  int wdone[MAXLVL] = {0}, gdone[MAXLVL] = {0};  // per level: groups whose W-step / both steps are launched
  int pairs_used = 0;                            // counter pairs consumed by queue-driven launches
};
template <typename T>
int trtri_advance(const T* L, long N, T* Tm, T* W, int ready_blocks, TrtriProgress* st, hipStream_t s, int wg_cap,
                  int* ctr /* info + EARLY_CTR0, or null */, int nctr_pairs, int reserve_cus, Batch bt = Batch(),
                  long ld = 0 /* leading dimension of L, Tm, W if not N */);
// info[0] = potrf status; info[EARLY_CTR0 + 2i ..] = (tile queue, worker count) of the i-th early inverse launch
#define CHAIN_FLAG0 2  /* info[2], info[3]: the split chain's progress counters (rest steps / bulk launches finished) */
#define EARLY_CTR0 4
#define EARLY_CTR_PAIRS 125
#define POTRF_INFO_INTS (EARLY_CTR0 + 2 * EARLY_CTR_PAIRS + 4)
#define CHAIN_ABORT (POTRF_INFO_INTS - 4)      /* != 0: a bounded wait of the split chain ran out -- every later waiter leaves at once */
#define CHAIN_YIELD (POTRF_INFO_INTS - 3)      /* cu_code() of the running diagonal-block kernel, else 0 (dgp_common.h: yield_if_asked) */
#define CHAIN_DIAG_DONE (POTRF_INFO_INTS - 2)  /* diagonal blocks finished (split chain: the rest stream's trsm waits on it) */
#define CHAIN_TICKET (POTRF_INFO_INTS - 1)     /* workgroup ticket of the rest stream's column-update kernel */
// per-device table of compute units kept free of early-inverse workgroups (null if unavailable)
const unsigned char* reserved_cu_table(int nreserve, int* n_cu);
template <typename T>
int trtri(const T* L, const T* Dinv, long N, T* Tm, T* W, hipStream_t s, Batch bt = Batch(), long ld = 0);
template <typename T>
int lauum(const T* Tm, long N, T* S, hipStream_t s, Batch bt = Batch());
template <typename T>
int solve(const T* Tm, long N, const T* r, int n, T* z, T* alpha, T* partials, T* quad, hipStream_t s,
          Batch bt = Batch());
template <typename T>
int finish(const T* S, const T* alpha, long N, int n, T* dnoise, hipStream_t s, Batch bt = Batch());
// fp32 plans, after solve(): one step of iterative refinement with an fp64 residual (dgp_gram.hip::gram_residual).
//   delta = T^T (T rho32)  (the fp32 factor),  alpha <- alpha + delta,  quad = r^T alpha0 + rho^T (alpha0 + delta) in double
// (second-order accurate in the error of delta).  z and `partials` are solve()'s scratch; rho32 / delta: N elements each.
template <typename T>
int refine_solve(const T* Tm, long N, const T* r, int n, const double* rho64, const T* rho32, T* z, T* delta, T* alpha,
                 T* partials, T* quad, hipStream_t s, Batch bt, long scratch_stride /* site stride of rho64 (in doubles) */,
                 long rho32_stride /* site stride of rho32 / delta, in elements */);
// row slabs of the prediction's column reductions (part: 2 x PREDICT_SPLIT x M elements of workspace)
#define PREDICT_SPLIT 32
template <typename T>
int predict_var(const T* Tm, long N, const T* Ks, long M, T* V, const T* alpha, const T* kss, T* part, T* mean, T* var,
                hipStream_t s, Batch bt = Batch(), long wbs = 0);
long solve_partials(long N);
// ---- one matrix over several GPUs (dgp_dist.hip): the panel chain of W block columns on a slab-addressed matrix
template <typename T>
int potrf_group(T* A, long ld, int nbk, T* Tinv, T* logdet, int* info, int k0, int W, hipStream_t s);
// Block-cyclic column ownership: groups of W 128-wide panels, group g on rank g % world; a rank keeps its groups side
// by side in a slab of Cl columns (local column block lb <-> global block column gblock(lb)).
struct SlabMap {
  int W, world, rank;
  __host__ __device__ int gblock(int lb) const { return ((lb / W) * world + rank) * W + lb % W; }
};
template <typename T>
int gram_slab(int model, int d, const T* Xt, long N, int n, const double* theta, const T* noise, T* Aslab, long Cl,
              SlabMap sm, hipStream_t s);
// partial gradient over the tiles of K^^-1 this rank holds (rows <= columns of its own block columns) and
// 1/2 (diag K^^-1 - alpha^2) for its own columns (dnoise_part: N entries, others untouched)
template <typename T>
int gram_grad_slab(int model, int d, const T* Xt, long N, int n, const double* theta, const T* Sslab, long Cl, SlabMap sm,
                   const T* alpha, T* partials, T* dtheta, T* dnoise_part, hipStream_t s);
long gram_grad_slab_partials(long N, long Cl);
template <typename T>
int symv_lower(const T* S, long N, const T* g, int n, const T* alpha, T* beta, T* partials, T* dnoise, hipStream_t s,
               Batch bt = Batch(), long wbs = 0);
// cov (M x M) = Kss - V^T V, lower tiles; Kss already holds K(Xs, Xs) (identity pad)
template <typename T>
int posterior_cov(const T* V, long N, long M, T* cov, hipStream_t s, int B = 1, long wbs = 0);

// out (ndraw x m) = mean + (L Z)^T : L is M x M lower (identity pad), Z is M x Q standard normals, Q % 128 == 0
template <typename T>
int sample_draws(const T* L, long M, const T* Z, long Q, const T* mean, int m, int ndraw, T* out, hipStream_t s);

}  // namespace dgp
