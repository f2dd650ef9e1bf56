/* dgp_hip.h -- C ABI of libdgp_hip.so, the MI355X (gfx950) exact-GP marginal-likelihood engine.
 *
 * The reference (thodson-usgs/discontinuum) has NO FFI for this path: its boundary is the Python
 * class contract of `MarginalGPyTorch` (src/discontinuum/engines/gpytorch.py:36-626) and the hot-path
 * arithmetic is delegated to gpytorch.  Each entry point below therefore cites the reference call site
 * whose work it replaces.  Plain pointers and sizes only: device pointers are owned by the caller
 * (e.g. torch tensors), `stream` is a hipStream_t passed as void*, host pointers are read before
 * the call returns.  All functions are asynchronous on `stream` and return 0 on success, a negative
 * DGP_E* code for bad arguments, or a positive hipError_t.  No exceptions cross the ABI; a
 * non-positive-definite matrix is reported through the `info` slot of the output vector
 * (index of the first failing pivot, 1-based; the NLL is then NaN) so that the caller's NaN/exception
 * guard (engines/gpytorch.py:352-382) keeps working.
 *
 * dtype: 0 = float64, 1 = float32 (sizeof element = 8 / 4; every device array below has that type).
 * model: 0 = loadest-gp composite kernel, d columns (time first), 2d+5 constrained hyperparameters
 *            (src/loadest_gp/models/gpytorch.py:61-128)
 *        1 = rating-gp composite kernel, d = 2 (time, stage), 16 constrained hyperparameters
 *            (src/rating_gp/models/gpytorch.py:205-372, src/rating_gp/models/kernels.py:242-382)
 *        >= 16: a generic composite model registered with dgp_composite_define (below)
 *        parameter order: DESIGN.md section "Hyperparameter vectors".
 */
#ifndef DGP_HIP_H
#define DGP_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DGP_F64 0
#define DGP_F32 1
#define DGP_MODEL_LOADEST 0
#define DGP_MODEL_RATING 1

#define DGP_E_ARG (-1)       /* null pointer / bad size / bad dtype */
#define DGP_E_MODEL (-2)     /* unsupported (model, d) */
#define DGP_E_WORKSPACE (-3) /* workspace missing or too small */
#define DGP_E_STATE (-4)     /* call order violated (e.g. predict before factorize) */

/* output vector of dgp_fit_step / dgp_factorize, in elements of the plan dtype */
#define DGP_OUT_NLL 0    /* 1/2 r^T K^^-1 r + 1/2 log|K^| + n/2 log 2 pi */
#define DGP_OUT_QUAD 1   /* r^T K^^-1 r */
#define DGP_OUT_LOGDET 2 /* log|K^| */
#define DGP_OUT_INFO 3   /* 0, or 1-based index of the first non-positive pivot; -7: an internal wait of the factorisation's
                            split panel chain timed out (the NLL is NaN; see dgp_chol.hip::chain_wait) */
#define DGP_OUT_DTHETA 4 /* d NLL / d theta_p, p = 0 .. ntheta-1 */
#define DGP_OUT_SUM_DR 28 /* sum_i d NLL / d r_i (gradient of a constant prior mean is its negative); fit step only */
#define DGP_OUT_DR_W0 29  /* sum_i d NLL / d r_i * w0[i], and w1 in the next slot: see dgp_plan_set_dr_weights */
#define DGP_OUT_SUM_DNOISE 31 /* sum_i d NLL / d noise_i (gradient of a homoskedastic noise term); fit step only */
#define DGP_OUT_LEN 32

/* buffers exposed by dgp_plan_buffer (tests and profiling) */
#define DGP_BUF_XT 0    /* coordinates, SoA d x N */
#define DGP_BUF_A 1     /* K^ then its Cholesky factor L (lower), N x N row-major */
#define DGP_BUF_T 2     /* L^-1 (lower) */
#define DGP_BUF_S 3     /* K^^-1 (lower) */
#define DGP_BUF_Z 4     /* L^-1 r */
#define DGP_BUF_ALPHA 5 /* K^^-1 r */
#define DGP_BUF_INFO 6  /* int32: info of the last factorisation */
#define DGP_BUF_SCAL 7  /* scalars: [0] log|K^| accumulated by the diagonal-block kernels, [1] r^T K^^-1 r; float32 plans also keep
                           both UNROUNDED as doubles at elements [2..3] and [4..5] (the NLL's three terms are added in double) */

typedef struct dgp_plan dgp_plan;

int dgp_version(void);
const char* dgp_last_error(void);
/* number of constrained kernel hyperparameters of (model, d); <0 if unsupported */
int dgp_model_ntheta(int model, int d);
/* A GENERIC model: any sum of (optionally scaled) products of stationary factors -- RBF, Matern(nu = 1/2, 3/2, 5/2),
 * Periodic -- on subsets of the d <= 6 input columns, e.g. the reference's covariance with its unused trend term
 * (src/loadest_gp/models/gpytorch.py:78-88) switched on.  `spec` (host ints) describes the tree:
 *     d, nterms, then per term:  scaled (0/1), nfactors (<= 3), then per factor:
 *         type (0 RBF, 1 Matern, 2 Periodic), 2 nu (Matern: 1, 3, 5; else 0), ard (0/1), ndims, the ndims column indices
 * (<= 6 terms, Periodic factors on one column).  The constrained hyperparameters follow the same order: per term
 * [outputscale if scaled], per factor [lengthscale: one, or one per column if ard], [period if Periodic]; at most 24.
 * *model_out (>= 16) is then accepted wherever a model id is (dgp_plan_create, dgp_dist_create, dgp_model_ntheta).  An
 * interpreted evaluator: slower per matrix entry than the two fused models, same kernels otherwise. */
int dgp_composite_define(const int* spec_host, int nspec, int* model_out);
/* padded order N = round_up(n, 128) used by every N x N buffer */
int64_t dgp_padded_n(int64_t n);

/* A plan fixes (model, dtype, n, d) and owns host-side resources only (up to two internal HIP streams
 * and events).  Device memory is the caller's: query the size, allocate, hand it over. */
int dgp_plan_create(int model, int dtype, int64_t n, int d, dgp_plan** out);
int dgp_plan_destroy(dgp_plan* plan);
size_t dgp_plan_workspace_bytes(const dgp_plan* plan);
/* Optional, before dgp_plan_set_workspace: carry `batch` (1..1024) independent sites of the same (model, dtype, n, d)
 * in lockstep -- every kernel of a fit step is launched once for all of them (gridDim.z = batch), which amortises
 * the sequential panel chain and the launch rate over the batch (the reference analogue is its map over sites,
 * examples/nwqn-loadest-example/nwqn-loadest-example.py:156-159).  The workspace grows by the same factor and the
 * arrays of dgp_set_inputs / dgp_fit_step / dgp_factorize become batch-major: X[batch][n][d], theta[batch][ntheta],
 * r / noise / dr / dnoise [batch][n], out[batch][DGP_OUT_LEN].  dgp_predict, dgp_posterior_cov, dgp_predict_mean and
 * dgp_mean_vjp also accept batched plans: every site works at its own m points -- Xs[batch][m][d], theta[batch][ntheta]
 * -> mean / var [batch][m], cov [batch][M][M]; dgp_mean_vjp: w[batch][m] -> dtheta[batch][ntheta], dr / dnoise [batch][n]
 * -- with ONE launch sequence for all sites (gridDim.z = batch, like the fit step); their work areas are batch times the
 * single-site size (the *_workspace_bytes queries account for it).  The stage-level entries dgp_stage_grad and
 * dgp_cross_gram need batch == 1. */
int dgp_plan_set_batch(dgp_plan* plan, int batch);
int dgp_plan_batch(const dgp_plan* plan);
/* Ragged batches (after dgp_plan_set_workspace, before dgp_set_inputs): site b has sizes[b] <= n observations; it
 * uses the first sizes[b] rows of its [n]-sized slots in X / r / noise (the rest is ignored), its padding is handled
 * like the plan's own (identity), its NLL carries sizes[b]/2 log(2 pi), and dr / dnoise are zero beyond sizes[b]. */
int dgp_plan_set_site_sizes(dgp_plan* plan, const int64_t* sizes_host, void* stream);
int dgp_plan_set_workspace(dgp_plan* plan, void* dev_ptr, size_t bytes);
/* Reductions for a parametric prior mean mu(x; phi) that lives on the host side (rating-gp's power law,
 * src/rating_gp/models/gpytorch.py:28-40): with w_dev = two device vectors [2][n] (e.g. d mu_i / d phi_k), every
 * following dgp_fit_step also writes sum_i dNLL/dr_i w_k[i] to out[DGP_OUT_DR_W0 + k], next to out[DGP_OUT_SUM_DR]
 * and out[DGP_OUT_SUM_DNOISE] -- the host gets its mean / noise gradients from the one result row instead of reducing
 * dr and dnoise itself.  The vectors are read when the step runs; NULL clears.  Batched plans: [batch][2][n]. */
int dgp_plan_set_dr_weights(dgp_plan* plan, const void* w_dev);
/* Concurrency inside one fit step.  0: everything in order on the caller's stream.  1: the bulk trailing
 * updates of the factorisation run on a second (lowest-priority) stream beside the panel chain, and -- single-site
 * plans -- the chain itself is split: the next diagonal block's own rows and tile on the caller's stream, the rest of
 * the chain on a third (highest-priority) stream.  2 (default): additionally the inverse's level recursion is issued
 * on another stream behind checkpoints of the factorisation, filling the CUs its sequential tail leaves idle -- best
 * for ONE plan per GPU; callers that keep several plans in flight on one GPU should select 1 (the other plans already
 * fill the idle CUs).  The internal streams belong to the CALLER's stream: every plan driven from one stream shares
 * one set (a process has few hardware queues), plans driven from different streams get a set each; they live as long
 * as the process.  Results are ordered on the caller's stream whatever the level. */
int dgp_plan_set_lookahead(dgp_plan* plan, int level);
/* Plan-level options (any time; they take effect at the next call).  The first three select the TILE SHAPE of the three
 * O(n^3) stages, which the library otherwise derives from the problem size -- the parity tests use them to run the
 * kernels of the benchmark shapes (128 x 128 tiles of the direct-to-LDS core) at sizes the dense CPU oracle reaches;
 * results must not depend on them beyond rounding.  Nothing in the reference corresponds (gpytorch picks its own
 * LAPACK / CG paths, engines/gpytorch.py:350-353).
 *   DGP_OPT_LAUUM64_MAX_TILES  K^^-1 = L^-T L^-1 runs in 64 x 64 tiles while (128-tiles x batch) <= value (default 1000;
 *                              0: always the 128 x 128 kernel)
 *   DGP_OPT_SYRK_SLOTS         workgroup slots of one round of the bulk trailing update: whole rounds run as 128 x 128
 *                              tiles, the remainder is cut into 64-wide pieces (default 512)
 *   DGP_OPT_TRTRI_SMALL        a level of the inverse with fewer than `value` 128-tiles (x batch) runs in 64 x 64 tiles
 *                              (default 1024; 0: always 128 x 128)
 *   DGP_OPT_REFINE             float32 plans only (default 1): after the triangular solves, ONE step of iterative
 *                              refinement -- residual r - K^ alpha in float64 with K^ re-evaluated on the fly, correction
 *                              through the float32 factor -- so that alpha, the quadratic form and everything computed
 *                              from alpha (gradients, dnoise, predictive mean) carry ~cond(K^) eps32 SQUARED instead of
 *                              cond(K^) eps32; the reference trains in float32 (engines/gpytorch.py:221-222). */
#define DGP_OPT_LAUUM64_MAX_TILES 0
#define DGP_OPT_SYRK_SLOTS 1
#define DGP_OPT_TRTRI_SMALL 2
#define DGP_OPT_REFINE 3
/* tile ORDER of the bulk update / of K^^-1 = L^-T L^-1 (measurement knobs, default 0 = rows of the triangle): S > 0 runs
 * S x S supertiles per XCD -- S times fewer distinct operand panels in flight per L2.  Same tiles, same sums: results are
 * bitwise those of the default order. */
#define DGP_OPT_SYRK_ORDER 4
#define DGP_OPT_LAUUM_ORDER 5
/* single-site plans (default 1): while the diagonal-block kernel of the panel chain runs, the waves of the concurrent bulk
 * update that share ITS compute unit sleep (a word in the plan's status block names the CU; bounded at ~0.2 ms) -- the
 * block kernel is one workgroup of dependent latencies on the critical path and runs 3-5 times slower beside them.  A
 * scheduling hint: results are bitwise the same with 0. */
#define DGP_OPT_CHAIN_YIELD 6
/* default 0 (a measured alternative, not faster: csrc/dgp_fused.hip): 1 = when K^^-1 = L^-T L^-1 runs in 128 x 128 tiles (see
 * DGP_OPT_LAUUM64_MAX_TILES) and the model is one of the two fused covariance functions, every tile contracts itself with
 * dK/dtheta right after it is stored (one launch instead of lauum + gram_grad; K^^-1 is never re-read from HBM).  Same sums in
 * a different order: the gradients agree to rounding (1e-11 relative in fp64), everything else is bitwise the same.  The
 * backward pass of engines/gpytorch.py:384. */
#define DGP_OPT_FUSED_GRAD 7
/* batched plans of 4 or more sites (default 0: measured neutral; 1 =): the factorisation's panel GROUPS (4 panels) solve their rows below the group's
 * diagonal block with ONE GEMM against that block's inverse -- L[i, group] = A[i, group] T_D^T -- instead of panel-by-panel trsm
 * and column-update launches over the full height; 0 = the panel-by-panel chain (the form until round 4).  A different
 * association of the same sums: results agree to rounding (fp64 ~1e-13).  Replaces part of what gpytorch's Cholesky does
 * at engines/gpytorch.py:350-353. */
#define DGP_OPT_GROUP_GEMM 8
int dgp_plan_set_option(dgp_plan* plan, int key, int64_t value);
int dgp_plan_get_option(const dgp_plan* plan, int key, int64_t* value_out);
int dgp_plan_buffer(const dgp_plan* plan, int which, void** dev_ptr, int64_t* ld);
/* batched plans: site b's copy of every dgp_plan_buffer buffer starts this many bytes after site b - 1's (tests) */
size_t dgp_plan_site_stride_bytes(const dgp_plan* plan);

/* Training inputs X (n x d row-major, device) -> internal SoA copy.  Replaces the train_x tensor
 * handed to ExactGP at engines/gpytorch.py:221-235. */
int dgp_set_inputs(dgp_plan* plan, const void* X_dev, void* stream);

/* One fit step = one evaluation of the data term of the objective and ALL its gradients, the work of
 * `output = model(train_x); nll = -mll(output, train_y); objective.backward()` at
 * engines/gpytorch.py:350-384 (minus the O(P) prior / constraint algebra, which stays in torch):
 *   theta_host  ntheta constrained kernel hyperparameters (host, double)
 *   r_dev       residual y - mean(X) (n)          noise_dev  diagonal of Sigma (n)
 *   out_dev     DGP_OUT_LEN elements, layout DGP_OUT_*
 *   dr_dev      d NLL / d r = alpha = K^^-1 r (n)
 *   dnoise_dev  d NLL / d noise_i = 1/2 (K^^-1_ii - alpha_i^2) (n)
 * Leaves L, L^-1, K^^-1, alpha in the plan buffers. */
int dgp_fit_step(dgp_plan* plan, const double* theta_host, const void* r_dev, const void* noise_dev,
                 void* out_dev, void* dr_dev, void* dnoise_dev, void* stream);

/* Value only (no K^^-1, no gradient): Gram, Cholesky, L^-1, alpha.  out_dev as above with dtheta = 0.
 * This is the eval-mode cache build of ExactGP (engines/gpytorch.py:618-622). */
int dgp_factorize(dgp_plan* plan, const double* theta_host, const void* r_dev, const void* noise_dev,
                  void* out_dev, void* stream);

/* Workspace for dgp_predict on m test points. */
size_t dgp_predict_workspace_bytes(const dgp_plan* plan, int64_t m);
/* Posterior at Xs (m x d row-major, device) from the factorisation currently held by the plan:
 *   mean_dev[j] = K(x*_j, X) alpha            (add the mean function on the host)
 *   var_dev[j]  = k(x*_j, x*_j) - || L^-1 K(X, x*_j) ||^2   (latent f; add likelihood noise on the host)
 * Replaces `self.likelihood(self.model(x))` .mean/.variance at engines/gpytorch.py:621-624. */
int dgp_predict(dgp_plan* plan, const double* theta_host, const void* Xs_dev, int64_t m, void* work_dev,
                size_t work_bytes, void* mean_dev, void* var_dev, void* stream);

/* Full latent posterior covariance for sample() (engines/gpytorch.py:575-580):
 *   cov_dev (M x M row-major, M = dgp_padded_n(m), lower triangle valid, identity pad)
 *     = K(Xs, Xs) - V^T V,  V = L^-1 K(X, Xs);   mean_dev[j] = K(x*_j, X) alpha  (m entries).
 * work_dev as for dgp_predict (dgp_predict_workspace_bytes). */
int dgp_posterior_cov(dgp_plan* plan, const double* theta_host, const void* Xs_dev, int64_t m, void* work_dev,
                      size_t work_bytes, void* mean_dev, void* cov_dev, void* stream);

/* Draws from N(mean, L L^T) -- what `f_preds.sample(torch.Size([n]))` does at engines/gpytorch.py:575-580:
 *     out_dev[q][j] = mean_dev[j] + sum_{k <= j} L[j][k] Z[k][q]        (ndraw x m row-major, q = draw)
 * L_dev   M x M row-major, M = dgp_padded_n(m): a lower-triangular factor with zeros above the diagonal inside its
 *         diagonal 128-blocks and the identity in the pad -- e.g. DGP_BUF_A of an order-m plan after dgp_stage_potrf
 *         of the matrix dgp_posterior_cov wrote there (blocks above the block diagonal are never read);
 * Z_dev   M x Q row-major standard normals, Q = dgp_padded_n(ndraw) (the pad only feeds entries that are not stored);
 * mean_dev m entries or NULL.  One MFMA launch (the same tile core as the factorisation); needs no plan. */
int dgp_sample_draws(int dtype, const void* L_dev, int64_t m, const void* Z_dev, int64_t ndraw, const void* mean_dev,
                     void* out_dev, void* stream);

/* Predictive mean only, and its vector-Jacobian product -- what the rating-gp monotonicity penalty
 * differentiates (src/rating_gp/models/gpytorch.py:130-187: mean of likelihood(model(x_grid)) with grad).
 *   dgp_predict_mean : mean_dev[j] = K(x*_j, X) alpha                                  (m entries)
 *   dgp_mean_vjp     : given w_dev[j] = dLoss/dmean[j], with g = K(X, X*) w and beta = K^^-1 g:
 *        dtheta_dev[p] = sum_ij alpha_i w_j dK*_ij/dtheta_p - beta^T (dK/dtheta_p) alpha   (ntheta entries)
 *        dr_dev        = beta                       (dLoss/dr,     n entries)
 *        dnoise_dev    = -beta * alpha              (dLoss/dnoise, n entries)
 * Both use the factorisation (alpha, K^^-1) left in the plan by dgp_fit_step at the SAME theta. */
size_t dgp_mean_vjp_workspace_bytes(const dgp_plan* plan, int64_t m);
int dgp_predict_mean(dgp_plan* plan, const double* theta_host, const void* Xs_dev, int64_t m, void* work_dev,
                     size_t work_bytes, void* mean_dev, void* stream);
int dgp_mean_vjp(dgp_plan* plan, const double* theta_host, const void* Xs_dev, int64_t m, const void* w_dev,
                 void* work_dev, size_t work_bytes, void* dtheta_dev, void* dr_dev, void* dnoise_dev, void* stream);

/* In-library HIP-event timing of the stages of dgp_fit_step (events are recorded on the stream each
 * kernel is launched on, including the internal lookahead stream).  dgp_plan_get_timing synchronises
 * on the events of the most recent fit step and fills ms_out[DGP_TIME_COUNT]. */
#define DGP_TIME_GRAM 0      /* gram_sym kernel */
#define DGP_TIME_POTRF 1     /* whole factorisation, wall time on the caller's stream */
#define DGP_TIME_SYRK_SUM 2  /* sum of the bulk trailing-update (syrk) launches */
#define DGP_TIME_SYRK_N 3    /* number of those launches */
#define DGP_TIME_TRTRI 4     /* all trtri level launches */
#define DGP_TIME_LAUUM 5     /* lauum kernel */
#define DGP_TIME_SOLVE 6     /* triangular solves */
#define DGP_TIME_GRAD 7      /* gram_grad + reduction */
#define DGP_TIME_SYRK_FLOP 8 /* algorithmic flops of those bulk launches (not a time) */
#define DGP_TIME_COUNT 9
int dgp_plan_set_timing(dgp_plan* plan, int enabled);
int dgp_plan_get_timing(dgp_plan* plan, double* ms_out);

/* ---- ONE matrix distributed over `world` GPUs (BASELINE config 5; nothing in the reference corresponds) ----------
 * 1-D block-cyclic by column groups of `group_panels` 128-wide panels, group g on rank g % world.  A rank holds ONLY its
 * own groups -- three column slabs (K^ -> L, L^-1, K^^-1) of N x (its groups x 128 group_panels) elements, N =
 * dgp_dist_padded_n() -- plus whatever panel buffers the caller allocates for the payloads in flight
 * (dgp_dist_panel_elems(group) elements).  The caller moves the payloads (torch.distributed broadcast = RCCL over xGMI;
 * discontinuum_amd/dist_chol.py) and sums the O(n) vectors; all O(n^2) / O(n^3) work is in these entry points.
 *
 *   dgp_dist_gram                         every rank: its columns of K^ (the inputs are replicated)
 *   for g = 0 .. groups-1:   dgp_dist_factor(g, panel)        owner: panel chain, diagonal-block inverse, pack
 *                            <broadcast panel from rank g % world>
 *                            dgp_dist_update(g, panel, cb, ce) every rank: its block columns in [cb, ce) right of g
 *                                                              (ce <= 0: to the end; the owner of g + 1 brings that group
 *                                                              up to date first and factors it while the rest runs)
 *                            dgp_dist_invert(g, panel)         every rank: its columns of L^-1 advance by group g
 *   dgp_dist_status -> (local log-determinant, local info)     sum / max over the ranks
 *   dgp_dist_solve_partial(r) -> z_part (N)                    sum over the ranks: z = L^-1 r ;  r^T K^^-1 r = z^T z
 *   dgp_dist_alpha_partial(z) -> alpha_part (N)                sum over the ranks: alpha = K^^-1 r
 *   fp32 handles, one refinement step (the single plan's DGP_OPT_REFINE):
 *     dgp_dist_residual(alpha) -> rho64, rho32 (N)             every rank, no exchange: rho = r - K^ alpha, K^ re-evaluated
 *                                                              in double; then delta = K^^-1 rho32 by the two calls above,
 *                                                              alpha += delta, r^T K^^-1 r = r^T alpha0 + rho^T (alpha0 + delta)
 *   for g = 0 .. groups-1:   dgp_dist_pack_inverse(g, panel)  owner: its columns of L^-1 from the diagonal down
 *                            <broadcast>
 *                            dgp_dist_product(g, panel)        every rank: K^^-1 [group g rows, its columns >= g]
 *   dgp_dist_grad_partial(theta, alpha) -> dtheta_part (DGP_OUT_LEN, first ntheta valid), dnoise_part (N, zeros outside
 *                            the rank's columns)               sum over the ranks: dNLL/dtheta, 1/2 (diag K^^-1 - alpha^2)
 * Together: the NLL and ALL gradients of one fit step (engines/gpytorch.py:350-384) with N^3 / world flops per rank. */
typedef struct dgp_dist dgp_dist;
const char* dgp_dist_last_error(void);
int dgp_dist_create(int model, int dtype, int64_t n, int d, int rank, int world, int group_panels, dgp_dist** out);
int dgp_dist_destroy(dgp_dist* h);
int64_t dgp_dist_padded_n(const dgp_dist* h);     /* round_up(n, 128 group_panels) */
int dgp_dist_groups(const dgp_dist* h);
int64_t dgp_dist_slab_columns(const dgp_dist* h); /* columns of each of the rank's three slabs */
size_t dgp_dist_workspace_bytes(const dgp_dist* h);
size_t dgp_dist_panel_elems(const dgp_dist* h, int group);
int dgp_dist_set_workspace(dgp_dist* h, void* dev_ptr, size_t bytes);
int dgp_dist_set_inputs(dgp_dist* h, const void* X_dev, void* stream);
int dgp_dist_gram(dgp_dist* h, const double* theta_host, const void* noise_dev, void* stream);
int dgp_dist_factor(dgp_dist* h, int group, void* panel_dev, void* stream);
int dgp_dist_update(dgp_dist* h, int group, const void* panel_dev, int col_begin, int col_end, void* stream);
int dgp_dist_invert(dgp_dist* h, int group, const void* panel_dev, void* stream);
int dgp_dist_status(dgp_dist* h, void* stat_dev /* 2 elements */, void* stream);
int dgp_dist_solve_partial(dgp_dist* h, const void* r_dev, void* z_part_dev, void* stream);
int dgp_dist_alpha_partial(dgp_dist* h, const void* z_dev, void* alpha_part_dev, void* stream);
/* fp32 only; between the factorisation and dgp_dist_pack_inverse (the K^^-1 slab is its scratch: N^2 / 8 bytes, i.e. at
 * most 32 ranks).  rho64: N doubles; rho32: N floats (the same vector rounded, the right-hand side of the solves). */
int dgp_dist_residual(dgp_dist* h, const double* theta_host, const void* noise_dev, const void* r_dev, const void* alpha_dev,
                      double* rho64_dev, void* rho32_dev, void* stream);
int dgp_dist_pack_inverse(dgp_dist* h, int group, void* panel_dev, void* stream);
int dgp_dist_product(dgp_dist* h, int group, const void* panel_dev, void* stream);
int dgp_dist_grad_partial(dgp_dist* h, const double* theta_host, const void* alpha_dev, void* dtheta_part_dev,
                          void* dnoise_part_dev, void* stream);
/* the rank's slab `which` = DGP_BUF_A / DGP_BUF_T / DGP_BUF_S (tests) */
int dgp_dist_slab(const dgp_dist* h, int which, void** dev_ptr);

/* ---- single stages on the plan buffers, for parity tests and per-kernel profiling ---- */
int dgp_stage_gram(dgp_plan* plan, const double* theta_host, const void* noise_dev, void* stream);
int dgp_stage_potrf(dgp_plan* plan, void* stream);  /* A: K^ -> L ; T diag blocks <- L_kk^-1 */
int dgp_stage_trtri(dgp_plan* plan, void* stream);  /* T <- L^-1 */
int dgp_stage_lauum(dgp_plan* plan, void* stream);  /* S <- T^T T */
int dgp_stage_solve(dgp_plan* plan, const void* r_dev, void* stream); /* z, alpha, quad */
int dgp_stage_grad(dgp_plan* plan, const double* theta_host, void* dtheta_dev, void* stream);
/* rectangular K(X, Xs) into caller memory (N x M row-major, M = dgp_padded_n(m)); Xs as in dgp_predict;
 * work_dev holds the SoA copy of Xs (d * M elements) */
int dgp_cross_gram(dgp_plan* plan, const double* theta_host, const void* Xs_dev, int64_t m, void* work_dev,
                   void* Ks_dev, void* stream);

/* ---- diagnostics ----
 * One grid of 128 x 128 output tiles through either tile-GEMM core of the O(n^3) stages (core 0: register-staged
 * dgp_gemm.h::TileGemm; 1: direct-to-LDS dgp_gemm_dma.h::DmaGemm), for the test that they are BITWISE equal:
 *     C[128 bm + i][128 bn + j] = sum_{kk < k} opA(128 bm + i, kk) opB(128 bn + j, kk),   bm < tiles_m, bn < tiles_n
 *     x_kc != 0: op(i, kk) = p[i ld + kk] (k-contiguous);  x_kc == 0: op(i, kk) = p[kk ld + i]
 * k a multiple of 16; A, B 16-byte aligned with ld a multiple of 16 bytes; all device pointers of `dtype`.
 * reverse != 0: the k-tiles of 16 are summed from the last to the first (ascending inside a tile) -- the order of
 * K^^-1 = L^-T L^-1, whose terms decay along k (small-to-large summation; csrc/dgp_gemm.h).
 * variant (core 1 only): 0 = the plain accumulator map; 1 = 16-row / 16-column groups dealt alternately to the wave rows /
 * columns; 2..5 = that map + zero-work skipping in the block of 128 k's visited LAST (k a multiple of 128): 2 operand A is
 * op(i, kk) = 0 for i > kk there, 3 op(i, kk) = 0 for kk > i, 4 operand B is op(j, kk) = 0 for j > kk, 5 the output is a
 * diagonal tile of a symmetric product (only the 16 x 16 sub-tiles with row group >= column group are specified).  With
 * operands that have that structure the specified results are bitwise those of variant 0. */
int dgp_debug_tile_gemm(int dtype, int core, int a_kc, int b_kc, const void* A_dev, int64_t lda, const void* B_dev,
                        int64_t ldb, int64_t k, void* C_dev, int64_t ldc, int tiles_m, int tiles_n, int reverse, int variant,
                        void* stream);

/* Shader-clock probe: `nwg` one-wave workgroups (8 or more reach every XCD) stay resident for `seconds` (<= 5) on one of
 * the library's INTERNAL streams of the caller's `stream` (no new stream: a process has few hardware queues) and write
 * (d s_memtime, d s_memrealtime) -- shader cycles and ticks of the 100 MHz wall clock -- to out_dev[2 i], out_dev[2 i + 1]
 * (uint64).  Enqueue the load to be clocked on `stream` meanwhile and synchronise the DEVICE before reading: the clock the
 * chip held is d s_memtime / d s_memrealtime x 100 MHz (MI355X lowers it under MFMA-dense load).  bench.py uses it for
 * `roofline.clock_mhz`; no product kernel carries a stamp. */
int dgp_debug_clock_probe(void* out_dev, int nwg, double seconds, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DGP_HIP_H */
