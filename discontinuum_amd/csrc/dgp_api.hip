// dgp_api.hip -- the C ABI declared in include/dgp_hip.h: plan bookkeeping and stage sequencing.
#include <mutex>
#include <new>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <utility>
#include <vector>

#include "../../include/dgp_hip.h"
#include "dgp_common.h"
#include "dgp_internal.h"

using namespace dgp;

static thread_local char g_err[256] = "";
static int fail(int code, const char* msg) {
  snprintf(g_err, sizeof(g_err), "%s", msg);
  return code;
}
static int hipfail(hipError_t e, const char* where) {
  snprintf(g_err, sizeof(g_err), "%s: %s", where, hipGetErrorString(e));
  return (int)e;
}

// Small host -> device uploads (site sizes, the hyperparameters of large batches) go through PINNED staging slots that
// the plan owns: an asynchronous copy from pageable memory is only safe while the runtime happens to stage it before
// returning.  A slot is reused after the event recorded behind its last copy has completed.
struct PinnedRing {
  static constexpr int SLOTS = 4;
  void* buf[SLOTS];
  size_t cap[SLOTS];
  hipEvent_t ev[SLOTS];
  int busy[SLOTS], have_ev[SLOTS], next;
  void* acquire(size_t bytes) {
    const int i = next;
    if (busy[i]) {
      (void)hipEventSynchronize(ev[i]);
      busy[i] = 0;
    }
    if (cap[i] < bytes) {
      if (buf[i]) (void)hipHostFree(buf[i]);
      buf[i] = nullptr;
      cap[i] = 0;
      if (hipHostMalloc(&buf[i], bytes, hipHostMallocDefault) != hipSuccess) return nullptr;
      cap[i] = bytes;
    }
    return buf[i];
  }
  void commit(hipStream_t s) {  // after the copy out of the slot handed out last has been enqueued on s
    const int i = next;
    if (!have_ev[i]) have_ev[i] = hipEventCreateWithFlags(&ev[i], hipEventDisableTiming) == hipSuccess;
    if (have_ev[i] && hipEventRecord(ev[i], s) == hipSuccess) busy[i] = 1;
    else (void)hipStreamSynchronize(s);
    next = (next + 1) % SLOTS;
  }
  void destroy() {
    for (int i = 0; i < SLOTS; ++i) {
      if (busy[i]) (void)hipEventSynchronize(ev[i]);
      if (have_ev[i]) (void)hipEventDestroy(ev[i]);
      if (buf[i]) (void)hipHostFree(buf[i]);
    }
  }
};

struct dgp_plan {
  int model, dtype, d, ntheta;
  PinnedRing ring;
  int* nsite_host;        // host copy of the sites' sizes (B > 1): the inference entry points walk the sites on the host
  int B;                  // sites carried in lockstep (1 = plain plan); site b's buffers sit b * site_bytes further on
  size_t site_bytes;
  void* pre;              // device scratch for the batch's hyperparameters (B > 8), after the last site
  void* pre2;             // a second one for the fp64 evaluator of the refinement (fp32 plans), so that `pre` stays valid
  int pre_ready;          // the Gram build of the current step has uploaded them
  Tuning tune;            // tile-shape selectors (dgp_plan_set_option)
  int refine;             // fp32 plans: one step of iterative refinement with an fp64 residual after the solves
  int* nsite;             // device: the sites' own sizes (B > 1), after the hyperparameter scratch
  const void* dr_w;       // caller's [2][n] weight vectors for the out[DGP_OUT_DR_W0..] reductions, or null
  int64_t n, N;
  size_t elem;
  char* ws;
  size_t ws_bytes;
  // carved workspace
  void *Xt, *A, *Tm, *S, *z, *alpha, *gpart, *spart, *scal, *snap;
  int* info;
  hipStream_t sc;         // rest of the split panel chain (dgp_chol.hip::potrf_split)
  int lookahead, early;   // potrf schedule; issue the inverse's level recursion behind the factorisation's checkpoints
  int have_inputs, have_factor, have_inverse;
  hipStream_t s2, s3;     // bulk trailing updates; early inverse work
  hipEvent_t xev[8];      // checkpoints of the factorisation and fork/join events for s3
  int have_xev;
  hipEvent_t* ev;
  int nev;
  // optional HIP-event timing of the fit-step stages
  int timing, n_syrk, timed_valid;
  double syrk_flop;
  hipEvent_t* tev;   // 2 per stage (start, stop)
  hipEvent_t* sev;   // 2 per bulk syrk launch
  int nsev;
};
enum { TS_GRAM = 0, TS_POTRF, TS_TRTRI, TS_LAUUM, TS_SOLVE, TS_GRAD, TS_COUNT };

static size_t align_up(size_t x) { return (x + 255) & ~(size_t)255; }

struct Layout {
  size_t Xt, A, Tm, S, z, alpha, gpart, spart, scal, info, snap, total;
};
static Layout layout(const dgp_plan* p) {
  Layout L;
  const size_t e = p->elem, N = (size_t)p->N;
  size_t o = 0;
  L.Xt = o; o += align_up(e * N * p->d);
  L.A = o; o += align_up(e * N * N);
  L.Tm = o; o += align_up(e * N * N);
  L.S = o; o += align_up(e * N * N);
  L.z = o; o += align_up(e * N);
  L.alpha = o; o += align_up(e * N);
  L.gpart = o; o += align_up(e * (size_t)gram_grad_partials(p->N));
  L.spart = o; o += align_up(e * (size_t)solve_partials(p->N));
  L.scal = o; o += align_up(e * 16);
  L.info = o; o += align_up(sizeof(int) * POTRF_INFO_INTS);
  L.snap = o; o += align_up(e * 2 * DGP_TILE_HOST * DGP_TILE_HOST);  // the split chain's two snapshot blocks
  L.total = o;
  return L;
}

static int default_lookahead() { return 2; }  // lookahead on: the group-ahead schedule of dgp_chol.hip::potrf
// panels per group of that schedule: pairs for one site (chain-bound), larger groups (K = 512 bulk updates) for a batch
// One site: pairs while the factorisation is chain-bound; from ~96 block columns on it is bound by its bulk updates, which
// gain from a longer K per read-modify-write pass over the trailing matrix (measured, one site: n = 16384 fp32 43.1 -> 42.1 ms
// per fit step with groups of 4, fp64 79.2 -> 77.2; n = 65536 fp32 2305 -> 2239 (4) -> 2216 ms (8); n = 8192: 12.44 -> 12.50
// with 4).  Batched plans: 4, 6 and 8 give the same factorisation time (n = 8192 x 32: 101.7 / 101.9 / 102.5 ms).
static int group_size(int lookahead, int batch, long nbk) {
  if (!lookahead) return 0;
  if (const char* e = getenv("DGP_GROUP")) return atoi(e) < 2 ? 2 : atoi(e);
  if (batch >= 4) return 4;
  return nbk >= 256 ? 8 : (nbk >= 96 ? 4 : 2);
}

// workgroups the EARLY inverse launches may occupy (one per CU): they share the GPU with the panel chain
static int early_wg_cap() {
  const char* e = getenv("DGP_EARLY_WG_CAP");
  return e ? atoi(e) : 384;
}
#define EARLY_WG_CAP early_wg_cap()
// compute units those launches leave to the panel chain (0 = none); DGP_EARLY_RESERVED_CUS overrides (tuning only)
static int early_reserved_cus() {
  const char* e = getenv("DGP_EARLY_RESERVED_CUS");
  const int v = e ? atoi(e) : 64;
  return v < 0 ? 0 : (v > 128 ? 128 : v);
}
#define EARLY_RESERVED_CUS early_reserved_cus()

extern "C" {

int dgp_version(void) { return 1; }
const char* dgp_last_error(void) { return g_err; }
int dgp_model_ntheta(int model, int d) { return model_ntheta(model, d); }
int dgp_composite_define(const int* spec, int nspec, int* model_out) {
  if (!spec || nspec < 2 || !model_out) return fail(DGP_E_ARG, "dgp_composite_define: null argument");
  const int id = composite_define(spec, nspec);
  if (id == -5) return fail(DGP_E_MODEL, "dgp_composite_define: the registry is full (64 distinct kernel structures per process)");
  if (id < 0) return fail(DGP_E_MODEL, "dgp_composite_define: malformed or unsupported kernel description");
  *model_out = id;
  return 0;
}
int64_t dgp_padded_n(int64_t n) { return round_up(n, DGP_TILE_HOST); }

int dgp_plan_create(int model, int dtype, int64_t n, int d, dgp_plan** out) {
  if (!out || n <= 0 || n > (1 << 20)) return fail(DGP_E_ARG, "dgp_plan_create: bad n / null out");
  if (dtype != DGP_F64 && dtype != DGP_F32) return fail(DGP_E_ARG, "dgp_plan_create: dtype must be 0 (f64) or 1 (f32)");
  const int nt = model_ntheta(model, d);
  if (nt < 0) return fail(DGP_E_MODEL, "dgp_plan_create: unsupported (model, d)");
  dgp_plan* p = new (std::nothrow) dgp_plan();
  if (!p) return fail(DGP_E_ARG, "dgp_plan_create: out of host memory");
  memset(p, 0, sizeof(*p));
  p->model = model;
  p->dtype = dtype;
  p->d = d;
  p->ntheta = nt;
  p->n = n;
  p->N = round_up(n, DGP_TILE_HOST);
  p->elem = dtype == DGP_F64 ? 8 : 4;
  p->B = 1;
  p->lookahead = default_lookahead();
  p->early = getenv("DGP_NO_EARLY_TRTRI") ? 0 : 1;
  p->tune = default_tuning();
  p->refine = (dtype == DGP_F32 && !(getenv("DGP_NO_REFINE") && atoi(getenv("DGP_NO_REFINE")))) ? 1 : 0;
  *out = p;
  return 0;
}

int dgp_plan_destroy(dgp_plan* p) {
  if (!p) return 0;
  if (p->ev) {
    for (int i = 0; i < p->nev; ++i) (void)hipEventDestroy(p->ev[i]);
    delete[] p->ev;
  }
  if (p->tev) {
    for (int i = 0; i < 2 * TS_COUNT; ++i) (void)hipEventDestroy(p->tev[i]);
    delete[] p->tev;
  }
  if (p->sev) {
    for (int i = 0; i < p->nsev; ++i) (void)hipEventDestroy(p->sev[i]);
    delete[] p->sev;
  }
  // s2 / sc / s3 belong to the caller-stream's StreamSet (process lifetime), not to the plan
  if (p->have_xev)
    for (int i = 0; i < 8; ++i) (void)hipEventDestroy(p->xev[i]);
  p->ring.destroy();
  delete[] p->nsite_host;
  delete p;
  return 0;
}

size_t dgp_plan_workspace_bytes(const dgp_plan* p) {
  return p ? layout(p).total * (size_t)p->B + 2 * align_up(pre_scratch_bytes(p->B)) + (p->B > 1 ? align_up(sizeof(int) * p->B) : 0)
           : 0;
}

int dgp_plan_set_batch(dgp_plan* p, int batch) {
  if (!p) return fail(DGP_E_ARG, "null plan");
  if (batch < 1 || batch > DGP_MAX_BATCH_SITES) return fail(DGP_E_ARG, "dgp_plan_set_batch: batch must be 1..1024");
  if (p->ws) return fail(DGP_E_STATE, "dgp_plan_set_batch: call before dgp_plan_set_workspace");
  p->B = batch;
  return 0;
}
int dgp_plan_batch(const dgp_plan* p) { return p ? p->B : 0; }

int dgp_plan_set_site_sizes(dgp_plan* p, const int64_t* sizes, void* stream) {
  if (!p || !sizes) return fail(DGP_E_ARG, "dgp_plan_set_site_sizes: null argument");
  if (!p->ws) return fail(DGP_E_WORKSPACE, "plan has no workspace: call dgp_plan_set_workspace");
  if (p->B == 1) {
    if (sizes[0] != p->n) return fail(DGP_E_ARG, "dgp_plan_set_site_sizes: an unbatched plan has the size it was created with");
    return 0;
  }
  for (int b = 0; b < p->B; ++b)
    if (sizes[b] < 1 || sizes[b] > p->n) return fail(DGP_E_ARG, "dgp_plan_set_site_sizes: sizes must be in 1..n");
  int* v = (int*)p->ring.acquire(sizeof(int) * (size_t)p->B);
  if (!v) return fail(DGP_E_ARG, "dgp_plan_set_site_sizes: out of pinned host memory");
  for (int b = 0; b < p->B; ++b) p->nsite_host[b] = v[b] = (int)sizes[b];
  hipError_t e = hipMemcpyAsync(p->nsite, v, sizeof(int) * (size_t)p->B, hipMemcpyHostToDevice, (hipStream_t)stream);
  p->ring.commit((hipStream_t)stream);
  if (e != hipSuccess) return hipfail(e, "dgp_plan_set_site_sizes");
  p->have_factor = 0;
  return 0;
}

int dgp_plan_set_dr_weights(dgp_plan* p, const void* w_dev) {
  if (!p) return fail(DGP_E_ARG, "dgp_plan_set_dr_weights: null plan");
  p->dr_w = w_dev;
  return 0;
}

int dgp_plan_set_workspace(dgp_plan* p, void* dev_ptr, size_t bytes) {
  if (!p || !dev_ptr) return fail(DGP_E_ARG, "dgp_plan_set_workspace: null");
  const Layout L = layout(p);
  if (bytes < dgp_plan_workspace_bytes(p))
    return fail(DGP_E_WORKSPACE, "dgp_plan_set_workspace: workspace too small");
  p->site_bytes = L.total;
  if (((uintptr_t)dev_ptr & 255) != 0) return fail(DGP_E_ARG, "dgp_plan_set_workspace: pointer must be 256-byte aligned");
  p->ws = (char*)dev_ptr;
  p->ws_bytes = bytes;
  p->Xt = p->ws + L.Xt;
  p->A = p->ws + L.A;
  p->Tm = p->ws + L.Tm;
  p->S = p->ws + L.S;
  p->z = p->ws + L.z;
  p->alpha = p->ws + L.alpha;
  p->gpart = p->ws + L.gpart;
  p->spart = p->ws + L.spart;
  p->scal = p->ws + L.scal;
  p->info = (int*)(p->ws + L.info);
  p->snap = p->ws + L.snap;
  p->pre = pre_scratch_bytes(p->B) ? (void*)(p->ws + L.total * (size_t)p->B) : nullptr;
  p->pre2 = p->pre ? (void*)((char*)p->pre + align_up(pre_scratch_bytes(p->B))) : nullptr;
  p->nsite = nullptr;
  p->dr_w = nullptr;
  if (p->B > 1) {  // every site starts at the full size n
    p->nsite = (int*)(p->ws + L.total * (size_t)p->B + 2 * align_up(pre_scratch_bytes(p->B)));
    delete[] p->nsite_host;
    p->nsite_host = new (std::nothrow) int[(size_t)p->B];
    if (!p->nsite_host) return fail(DGP_E_ARG, "dgp_plan_set_workspace: out of host memory");
    for (int b = 0; b < p->B; ++b) p->nsite_host[b] = (int)p->n;
    hipError_t e = hipMemcpy(p->nsite, p->nsite_host, sizeof(int) * (size_t)p->B, hipMemcpyHostToDevice);  // blocking
    if (e != hipSuccess) return hipfail(e, "dgp_plan_set_workspace: hipMemcpy");
  }
  p->have_inputs = p->have_factor = 0;
  return 0;
}

int dgp_plan_set_option(dgp_plan* p, int key, int64_t value) {
  if (!p) return fail(DGP_E_ARG, "dgp_plan_set_option: null plan");
  switch (key) {
    case DGP_OPT_LAUUM64_MAX_TILES:
      if (value < 0 || value > (1 << 30)) return fail(DGP_E_ARG, "dgp_plan_set_option: value out of range");
      p->tune.lauum64_max_tiles = (int)value;
      return 0;
    case DGP_OPT_SYRK_SLOTS:
      if (value < 1 || value > (1 << 20)) return fail(DGP_E_ARG, "dgp_plan_set_option: value out of range");
      p->tune.syrk_slots = (int)value;
      return 0;
    case DGP_OPT_TRTRI_SMALL:
      if (value < 0) return fail(DGP_E_ARG, "dgp_plan_set_option: value out of range");
      p->tune.trtri_small = (long)value;
      return 0;
    case DGP_OPT_SYRK_ORDER:
    case DGP_OPT_LAUUM_ORDER:
      if (value < 0 || value > 64) return fail(DGP_E_ARG, "dgp_plan_set_option: value out of range");
      (key == DGP_OPT_SYRK_ORDER ? p->tune.syrk_super : p->tune.lauum_super) = (int)value;
      return 0;
    case DGP_OPT_CHAIN_YIELD:
      p->tune.chain_yield = value ? 1 : 0;
      return 0;
    case DGP_OPT_FUSED_GRAD:
      p->tune.fused_grad = value ? 1 : 0;
      return 0;
    case DGP_OPT_GROUP_GEMM:
      p->tune.group_gemm = value ? 1 : 0;
      return 0;
    case DGP_OPT_REFINE:
      if (p->dtype != DGP_F32 && value) return fail(DGP_E_ARG, "dgp_plan_set_option: refinement applies to float32 plans");
      p->refine = value ? 1 : 0;
      return 0;
    default:
      return fail(DGP_E_ARG, "dgp_plan_set_option: unknown option");
  }
}
int dgp_plan_get_option(const dgp_plan* p, int key, int64_t* value) {
  if (!p || !value) return fail(DGP_E_ARG, "dgp_plan_get_option: null argument");
  switch (key) {
    case DGP_OPT_LAUUM64_MAX_TILES: *value = p->tune.lauum64_max_tiles; return 0;
    case DGP_OPT_SYRK_SLOTS: *value = p->tune.syrk_slots; return 0;
    case DGP_OPT_TRTRI_SMALL: *value = p->tune.trtri_small; return 0;
    case DGP_OPT_REFINE: *value = p->refine; return 0;
    case DGP_OPT_SYRK_ORDER: *value = p->tune.syrk_super; return 0;
    case DGP_OPT_LAUUM_ORDER: *value = p->tune.lauum_super; return 0;
    case DGP_OPT_CHAIN_YIELD: *value = p->tune.chain_yield; return 0;
    case DGP_OPT_FUSED_GRAD: *value = p->tune.fused_grad; return 0;
    case DGP_OPT_GROUP_GEMM: *value = p->tune.group_gemm; return 0;
    default: return fail(DGP_E_ARG, "dgp_plan_get_option: unknown option");
  }
}

int dgp_plan_set_lookahead(dgp_plan* p, int level) {
  if (!p) return fail(DGP_E_ARG, "null plan");
  p->lookahead = level ? default_lookahead() : 0;
  p->early = level >= 2 && !getenv("DGP_NO_EARLY_TRTRI");
  return 0;
}

size_t dgp_plan_site_stride_bytes(const dgp_plan* p) { return p ? layout(p).total : 0; }

int dgp_plan_buffer(const dgp_plan* p, int which, void** dev_ptr, int64_t* ld) {
  if (!p || !dev_ptr || !p->ws) return fail(DGP_E_ARG, "dgp_plan_buffer: null / no workspace");
  void* q = nullptr;
  switch (which) {
    case DGP_BUF_XT: q = p->Xt; break;
    case DGP_BUF_A: q = p->A; break;
    case DGP_BUF_T: q = p->Tm; break;
    case DGP_BUF_S: q = p->S; break;
    case DGP_BUF_Z: q = p->z; break;
    case DGP_BUF_ALPHA: q = p->alpha; break;
    case DGP_BUF_INFO: q = p->info; break;
    case DGP_BUF_SCAL: q = p->scal; break;
    default: return fail(DGP_E_ARG, "dgp_plan_buffer: unknown buffer");
  }
  *dev_ptr = q;
  if (ld) *ld = p->N;
  return 0;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// The library's internal streams belong to the CALLER'S stream, not to a plan: every plan driven from one stream (the
// usual case: torch's current stream) uses the same three -- bulk updates (lowest priority), rest of the split panel
// chain (highest), early inverse (lowest).  A process has few hardware queues (4 by default); streams beyond that share
// one, and kernels of streams that share a queue run one after the other.  With a stream set per PLAN, a process that had
// created a second plan found the rest stream of the split chain behind the bulk stream in one queue: the critical
// kernels then wait for whole bulk launches (measured: a single n = 8192 fit went from 12.3 to 20 ms as soon as a batched
// plan existed in the process).  Plans driven from DIFFERENT streams (sites.py::fit_sites) still get a set each.  The
// streams live as long as the process.
struct StreamSet {
  hipStream_t bulk = nullptr, rest = nullptr, early = nullptr;
};
static StreamSet* stream_set(hipStream_t caller) {
  static std::mutex mtx;
  static std::vector<std::pair<std::pair<int, hipStream_t>, StreamSet*>> sets;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(mtx);
  for (auto& e : sets)
    if (e.first.first == dev && e.first.second == caller) return e.second;
  StreamSet* st = new (std::nothrow) StreamSet();
  if (!st) return nullptr;
  sets.push_back({{dev, caller}, st});
  return st;
}
static int make_stream(hipStream_t* out, bool high) {
  if (*out) return 0;
  int least = 0, greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
  hipError_t e = hipStreamCreateWithPriority(out, hipStreamNonBlocking, high ? greatest : least);
  return e == hipSuccess ? 0 : hipfail(e, "hipStreamCreateWithPriority");
}

// ---- shader-clock probe ---------------------------------------------------------------------------------------------
// MI355X lowers its clock under an MFMA-dense load (MI355X_MICROARCH.md "DVFS give-back"), so a kernel's fraction of the
// 2.4 GHz peak mixes two things: how full it keeps the MFMA pipe and which clock the chip held.  This probe separates
// them WITHOUT touching a product kernel: a few one-wave workgroups stay resident for `ticks` of the 100 MHz wall clock
// and stamp the shader-cycle counter (s_memtime) and the wall clock (s_memrealtime) at both ends; the caller runs the
// load (e.g. back-to-back lauum launches) on another stream meanwhile.  clock = d s_memtime / d s_memrealtime x 100 MHz
// per workgroup (consecutive workgroup ids land on consecutive XCDs: 8 or more cover every XCD).  The stamps go to a
// buffer of their own; no product value is computed from them.
__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, long long ticks) {
  const long long r0 = wall_clock64();
  const long long s0 = clock64();
  while (wall_clock64() - r0 < ticks) __builtin_amdgcn_s_sleep(64);
  const long long s1 = clock64();
  const long long r1 = wall_clock64();
  if (threadIdx.x == 0) {
    out[2 * blockIdx.x] = (unsigned long long)(s1 - s0);
    out[2 * blockIdx.x + 1] = (unsigned long long)(r1 - r0);
  }
}


static int ensure_async(dgp_plan* p, hipStream_t s) {
  if (!p->lookahead) return 0;
  // the bulk trailing updates run at the LOWEST priority so that the latency-critical panel chain on
  // the caller's stream gets the CUs first whenever both have workgroups ready
  StreamSet* st = stream_set(s);
  if (!st) return fail(DGP_E_ARG, "out of host memory");
  int rc = make_stream(&st->bulk, false);
  if (rc) return rc;
  p->s2 = st->bulk;
  if (p->ev) return 0;
  hipError_t e;
  p->nev = 3 * (int)(p->N / DGP_TILE_HOST);  // group-ahead schedule: P, U; split chain: ED, ER, U
  p->ev = new (std::nothrow) hipEvent_t[p->nev];
  if (!p->ev) return fail(DGP_E_ARG, "out of host memory");
  for (int i = 0; i < p->nev; ++i) {
    e = hipEventCreateWithFlags(&p->ev[i], hipEventDisableTiming);
    if (e != hipSuccess) return hipfail(e, "hipEventCreateWithFlags");
  }
  return 0;
}

// The split panel chain (critical tile on the caller's stream, rest of the chain on `sc`): one site.  Large matrices start in the
// group schedule (groups of 2 / 4 / 8 panels, group_size()) and hand over for their chain-bound tail (split_start()).
// DGP_SPLIT_CHAIN=0 selects the single-stream chain (A/B measurements, the bitwise-equality tests); read at every call.
static bool split_applies(const dgp_plan* p) {
  const char* e = getenv("DGP_SPLIT_CHAIN");
  if (e && atoi(e) == 0) return false;
  const long nbk = p->N / DGP_TILE_HOST;
  return p->B == 1 && p->lookahead && nbk >= 4 && group_size(p->lookahead, p->B, nbk) % 2 == 0;  // (DGP_GROUP may ask for odd groups)
}
// first block column of the split chain: the earliest even k from which a bulk launch (all tiles right of the pair) is at
// most DGP_SPLIT_TILES tiles (default 768 = 1.5 rounds of the 512 workgroup slots): before that the factorisation is bound
// by its bulk launches and every chain kernel queues behind them whatever the schedule
static int split_start(const dgp_plan* p) {
  const char* e = getenv("DGP_SPLIT_TILES");
  const long cap = e ? atol(e) : 768;
  const int nbk = (int)(p->N / DGP_TILE_HOST);
  int k = 0;
  while (k + 4 <= nbk && (long)(nbk - k - 3) * (nbk - k - 2) / 2 > cap) k += 2;
  return k;
}
static int ensure_split(dgp_plan* p, hipStream_t s) {
  if (!split_applies(p)) return 0;
  StreamSet* st = stream_set(s);
  if (!st) return fail(DGP_E_ARG, "out of host memory");
  const int rc = make_stream(&st->rest, true);
  if (rc) return rc;
  p->sc = st->rest;
  return 0;
}

// the third stream exists only for plans that use it: every extra stream per plan costs throughput once
// several plans share the GPU (measured 86 -> 78 fits/s with two plans), so batched callers select level 1
static bool early_applies(const dgp_plan* p) {
  if (p->B != 1) return false;  // a batched plan's sites already fill each other's idle CUs
  // measured (fp64, one site): n = 2048 -1 %, 4096 -3 %, 8192 -6 %, 12288 +0.3 %, 16384 +0.8 %, 32768 (fp32) +6 %:
  // beyond ~10k the factorisation is bound by its bulk updates, not by the panel chain, and has no idle tail
  const int nbk = (int)(p->N / DGP_TILE_HOST);
  return p->early && p->lookahead >= 2 && nbk >= 16 && nbk <= 80;
}
static int ensure_early(dgp_plan* p, hipStream_t s) {
  if (!early_applies(p)) return 0;  // an idle extra stream is not free either (n = 32768: 293 -> 305 ms)
  StreamSet* st = stream_set(s);
  if (!st) return fail(DGP_E_ARG, "out of host memory");
  const int rc = make_stream(&st->early, false);
  if (rc) return rc;
  p->s3 = st->early;
  if (p->have_xev) return 0;
  hipError_t e;
  for (int i = 0; i < 8; ++i) {
    e = hipEventCreateWithFlags(&p->xev[i], hipEventDisableTiming);
    if (e != hipSuccess) return hipfail(e, "hipEventCreateWithFlags");
  }
  p->have_xev = 1;
  return 0;
}

static int ensure_timing(dgp_plan* p) {
  if (!p->timing || p->tev) return 0;
  p->tev = new (std::nothrow) hipEvent_t[2 * TS_COUNT];
  p->nsev = 2 * (int)(p->N / DGP_TILE_HOST);
  p->sev = new (std::nothrow) hipEvent_t[p->nsev];
  if (!p->tev || !p->sev) return fail(DGP_E_ARG, "out of host memory");
  for (int i = 0; i < 2 * TS_COUNT; ++i) {
    hipError_t e = hipEventCreate(&p->tev[i]);
    if (e != hipSuccess) return hipfail(e, "hipEventCreate");
  }
  for (int i = 0; i < p->nsev; ++i) {
    hipError_t e = hipEventCreate(&p->sev[i]);
    if (e != hipSuccess) return hipfail(e, "hipEventCreate");
  }
  return 0;
}
static void tick(dgp_plan* p, int stage, int stop, hipStream_t s) {
  if (p->timing && p->tev) (void)hipEventRecord(p->tev[2 * stage + stop], s);
}

// The tail of a fit step in ONE launch: dnoise_i = 1/2 (S_ii - alpha_i^2), dr = alpha, and the result row -- NLL,
// log-determinant, quadratic form, pivot status, and the reductions of dNLL/dr (= alpha) and dNLL/dnoise that a host-side
// mean / noise model needs for its gradients (sum dr, sum dr w0, sum dr w1, sum dnoise), so that the host never reduces
// device vectors.  Every workgroup handles 256 observations and leaves its partial sums in `part`; the last one to
// finish (ticket counter info[1], reset for the next step) adds them up in workgroup order and writes the row: the
// result does not depend on scheduling.  zero_grad (dgp_factorize): one workgroup, scalars only.
template <typename T>
__global__ __launch_bounds__(256) void finish_kernel(const T* scal, int* info, long n, int ntheta, int zero_grad, T* out,
                                                     long bs, long ibs, const int* ns, const T* alpha, const T* w, const T* S,
                                                     long N, T* dnoise, T* dr, T* part) {
  const long nfull = n;
  n = site_n(ns, (int)n);
  scal = site(scal, bs);
  info = site(info, ibs);
  alpha = site(alpha, bs);
  S = site(S, bs);
  part = site(part, bs);
  if (dnoise) dnoise = site(dnoise, nfull);
  if (dr) dr = site(dr, nfull);
  if (w) w = site(w, 2 * nfull);  // [site][2][n]
  out = site(out, (long)DGP_OUT_LEN);
  const int t = threadIdx.x, lane = t & 63, wv = t >> 6;
  __shared__ T red[4][4];
  __shared__ int last;
  T v[4] = {T(0), T(0), T(0), T(0)};
  if (!zero_grad) {
    const long i = (long)blockIdx.x * 256 + t;
    if (i < nfull) {
      const T a = alpha[i];
      const T dn = i < n ? T(0.5) * (S[i * N + i] - a * a) : T(0);
      if (dnoise) dnoise[i] = dn;
      if (dr) dr[i] = a;
      if (i < n) {
        v[0] = a;
        if (w) {
          v[1] = a * w[i];
          v[2] = a * w[nfull + i];
        }
        v[3] = dn;
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const T sum = wave_sum(v[q]);
      if (lane == 0) red[wv][q] = sum;
    }
    __syncthreads();
    if (t < 4) part[(long)blockIdx.x * 4 + t] = red[0][t] + red[1][t] + red[2][t] + red[3][t];
    __threadfence();
  }
  if (t == 0) last = (gridDim.x == 1) || (atomicAdd(&info[1], 1) == (int)gridDim.x - 1);
  __syncthreads();
  if (!last) return;
  __threadfence();
  if (t == 0) info[1] = 0;
  if (t < 4) {
    T total = T(0);
    if (!zero_grad)
      for (unsigned g = 0; g < gridDim.x; ++g) total += ((const volatile T*)part)[(long)g * 4 + t];  // other workgroups' stores
    red[0][t] = total;
  }
  __syncthreads();
  if (t == 0) {
    // fp32 plans: the log-determinant (mixed-precision panel, dgp_diag.h) and the quadratic form were accumulated in
    // double and sit unrounded in the scalar block's double slots (elements 2..3 and 4..5); the three terms of the NLL
    // -- which cancel to a small number when the noise is small -- are added in double and rounded once
    double logdet = (double)scal[0], quad = (double)scal[1];
    if (sizeof(T) == 4) {
      logdet = *reinterpret_cast<const double*>(scal + 2);
      quad = *reinterpret_cast<const double*>(scal + 4);
    }
    if (info[0] < 0) quad = __builtin_nan("");  // the split chain gave up waiting (dgp_chol.hip::chain_wait): no factor
    out[DGP_OUT_NLL] = (T)(0.5 * quad + 0.5 * logdet + 0.5 * 1.83787706640934548356 * (double)n);
    out[DGP_OUT_QUAD] = (T)quad;
    out[DGP_OUT_LOGDET] = (T)logdet;
    out[DGP_OUT_INFO] = (T)info[0];
  }
  if (zero_grad && t >= DGP_OUT_DTHETA && t < DGP_OUT_LEN) out[t] = T(0);
  if (!zero_grad && t >= DGP_OUT_DTHETA + ntheta && t < DGP_OUT_LEN) out[t] = t >= DGP_OUT_SUM_DR ? red[0][t - DGP_OUT_SUM_DR] : T(0);
}


template <typename T>
static Batch batch_of(const dgp_plan* p) {
  Batch bt;
  bt.B = p->B;
  bt.ws = (long)(p->site_bytes / sizeof(T));  // layout offsets are multiples of 256 bytes
  bt.ns = p->nsite;
  bt.tune = &p->tune;
  bt.W = p->S;  // scratch of the factorisation's group inverse (S is free until trtri / lauum use it)
  return bt;
}
template <typename T>
static int run_gram(dgp_plan* p, const double* theta, const void* noise, hipStream_t s) {
  p->pre_ready = 1;
  void* staging = p->pre ? p->ring.acquire(pre_scratch_bytes(p->B)) : nullptr;
  const int rc = gram_sym<T>(p->model, p->d, (const T*)p->Xt, p->N, (int)p->n, theta, (const T*)noise, (T*)p->A, s,
                             batch_of<T>(p), p->pre, staging);
  if (staging) p->ring.commit(s);
  return rc;
}
template <typename T>
static int run_potrf(dgp_plan* p, hipStream_t s) {
  int rc = ensure_async(p, s);
  if (rc) return rc;
  if ((rc = ensure_timing(p)) || (rc = ensure_split(p, s))) return rc;
  if (split_applies(p) && p->sc)
    return potrf_split<T>((T*)p->A, p->N, (T*)p->Tm, (T*)p->scal, p->info, (T*)p->snap, s, p->sc, p->s2, p->ev,
                          p->timing ? p->sev : nullptr, &p->n_syrk, &p->syrk_flop, 0, nullptr, nullptr, nullptr, nullptr, split_start(p), group_size(p->lookahead, p->B, p->N / DGP_TILE_HOST), &p->tune);
  return potrf<T>((T*)p->A, p->N, (T*)p->Tm, (T*)p->scal, p->info, group_size(p->lookahead, p->B, p->N / DGP_TILE_HOST), s, p->s2, p->ev,
                  p->timing ? p->sev : nullptr, &p->n_syrk, &p->syrk_flop, 0, nullptr, nullptr, nullptr, nullptr,
                  batch_of<T>(p));
}
template <typename T>
static int run_trtri(dgp_plan* p, hipStream_t s) {
  return trtri<T>((const T*)p->A, nullptr, p->N, (T*)p->Tm, (T*)p->S /* scratch W aliases S */, s, batch_of<T>(p));
}
template <typename T>
static int run_lauum(dgp_plan* p, hipStream_t s) {
  return lauum<T>((const T*)p->Tm, p->N, (T*)p->S, s, batch_of<T>(p));
}
template <typename T>
static int run_solve(dgp_plan* p, const void* r, hipStream_t s) {
  return solve<T>((const T*)p->Tm, p->N, (const T*)r, (int)p->n, (T*)p->z, (T*)p->alpha, (T*)p->spart,
                  (T*)p->scal + 1, s, batch_of<T>(p));
}
// fp32 plans: one step of iterative refinement of alpha and of the quadratic form with an fp64 residual (SURVEY.md section
// 8d's fp32 row as written; the reference trains in float32, src/discontinuum/engines/gpytorch.py:221-222, 350-353).
// The factor is stored in fp32, so alpha0 = T^T T r carries cond(K^) eps32 (2.4e-3 at config 3) and the quadratic form
// 1e-5 of itself -- more than the whole NLL tolerance when its three terms cancel.  rho = r - K^ alpha0 in double with K^
// re-evaluated on the fly (gram_residual), delta from the fp32 factor, alpha = alpha0 + delta, quad second-order accurate
// (refine_finish_kernel).  The scratch lives in S, which is free between the solves and lauum (trtri's W is consumed).
template <typename T>
static int run_refine(dgp_plan* p, const double* theta, const void* r, const void* noise, hipStream_t s) {
  if constexpr (sizeof(T) != 4) {
    return 0;
  } else {
    if (!p->refine) return 0;
    const long N = p->N;
    const size_t nb = (size_t)(N / 64), part_bytes = nb * nb * 64 * sizeof(double);
    if (gram_residual_scratch_bytes(N) > (size_t)N * N * sizeof(float)) return 0;  // (never: N >= 128)
    char* base = (char*)p->S;
    double* part = (double*)base;
    double* rho64 = (double*)(base + part_bytes);
    float* rho32 = (float*)(base + part_bytes + (size_t)N * sizeof(double));
    float* delta = rho32 + N;
    const Batch bt = batch_of<float>(p);
    const long ps = bt.ws / 2;  // site stride of S in doubles (bt.ws floats, a multiple of 64)
    void* staging = p->pre2 ? p->ring.acquire(pre_scratch_bytes(p->B)) : nullptr;
    int rc = gram_residual<float>(p->model, p->d, (const float*)p->Xt, N, (int)p->n, theta, (const float*)noise, (const float*)r,
                                  (const float*)p->alpha, part, rho64, rho32, s, bt, ps, bt.ws, p->pre2, staging);
    if (staging) p->ring.commit(s);
    if (rc) return rc;
    return refine_solve<float>((const float*)p->Tm, N, (const float*)r, (int)p->n, rho64, rho32, (float*)p->z, delta,
                               (float*)p->alpha, (float*)p->spart, (float*)p->scal + 1, s, bt, ps, bt.ws);
  }
}

template <typename T>
static int run_grad(dgp_plan* p, const double* theta, void* dtheta, hipStream_t s) {
  return gram_grad<T>(p->model, p->d, (const T*)p->Xt, p->N, (int)p->n, theta, (const T*)p->S, (const T*)p->alpha,
                      (T*)p->gpart, (T*)dtheta, s, batch_of<T>(p), DGP_OUT_LEN, p->pre, p->pre_ready != 0);
}

// DGP_SOLVE_OVERLAP: 0 = solves in front of lauum on the caller's stream (the order until round 4); 1 (default) = on the bulk
// stream (lowest priority: the trmv workgroups fill whatever slots lauum's grid leaves); 2 = on the rest stream (highest
// priority).  Needs the plan's event pool (lookahead >= 1).  Measured on one box in alternating processes
// (scripts/env_ab.py, wall ms per step, 0 / 1 / 2): 32 x n = 8192 274.3 / 273.8 / 274.4 (lauum itself 81.0 -> 83.3 with the
// solves beside it: its three workgroups per CU leave a trmv wave no registers, so the solves take slots, not idle
// resources), 64 x n = 4096 80.56 / 79.30 / 79.59 (-1.6 %), one site n = 8192 11.85 / 11.84 / 11.81; bitwise the same results.
// By default only for 1024 <= N <= 6144: at the headline size the step gains 0.2 % (below the 1 % bar) while lauum -- the kernel the
// roofline is quoted on -- reads 3-4 % slower for sharing the GPU (84.8 against 82.4 ms alone, profiles/r05_lauum_three_ways.txt);
// an explicit DGP_SOLVE_OVERLAP applies to every size.
template <typename T>
static int solve_overlap_mode(dgp_plan* p, hipStream_t s, hipStream_t* out) {
  static const int env = getenv("DGP_SOLVE_OVERLAP") ? atoi(getenv("DGP_SOLVE_OVERLAP")) : -1;
  // (and not below N = 1024: at the reference's own site size, n = 300, the step is ~0.3 ms of launches and the two cross-stream
  // dependencies cost more than the overlap gains: 0.306 -> 0.334 ms; n = 1024 0.705 -> 0.681, 2048 1.319 -> 1.270,
  // 128 x n = 2048 26.24 -> 25.62)
  const int mode = env >= 0 ? env : ((p->N >= 1024 && p->N <= 6144) ? 1 : 0);
  if (mode <= 0 || !p->lookahead || !p->ev || p->nev < 2) return 0;
  if (sizeof(T) == 4 && p->refine) return 0;
  if (mode == 1 && p->s2) {
    *out = p->s2;
    return 1;
  }
  StreamSet* st = stream_set(s);
  if (!st || make_stream(&st->rest, true)) return 0;
  *out = st->rest;
  return 2;
}

template <typename T>
static int fit_step(dgp_plan* p, const double* theta, const void* r, const void* noise, void* out, void* dr,
                    void* dnoise, int with_grad, hipStream_t s) {
  int rc;
  if ((rc = ensure_timing(p))) return rc;
  tick(p, TS_GRAM, 0, s);
  if ((rc = run_gram<T>(p, theta, noise, s))) return rc;
  tick(p, TS_GRAM, 1, s);
  if ((rc = ensure_async(p, s)) || (rc = ensure_early(p, s))) return rc;
  tick(p, TS_POTRF, 0, s);
  const int nbk = (int)(p->N / DGP_TILE_HOST);
  const bool early = early_applies(p) && p->s3 != nullptr;
  const Batch bt = batch_of<T>(p);
  if (early) {
    // The factorisation's tail is a sequential panel chain that leaves most CUs idle, and the inverse's level
    // recursion only needs the block columns that are already final: issue it piecewise on s3 behind
    // checkpoints at 1/2, 3/4 and 7/8 of the columns (W-steps as soon as the left half of a group is final).
    struct Early {
      dgp_plan* p;
      int ck[3];
      TrtriProgress st;
      int rc;
    } ctx{p, {nbk / 2, 3 * nbk / 4, 7 * nbk / 8}, {}, 0};
    if (const char* e = getenv("DGP_EARLY_CK")) {  // tuning: three checkpoints in sixteenths of the columns
      int a16 = 8, b16 = 12, c16 = 14;
      if (sscanf(e, "%d,%d,%d", &a16, &b16, &c16) == 3) {
        ctx.ck[0] = a16 * nbk / 16;
        ctx.ck[1] = b16 * nbk / 16;
        ctx.ck[2] = c16 * nbk / 16;
      }
    }
    auto on_ck = [](void* v, int c) {  // runs inside the factorisation's enqueue loop, right after checkpoint c
      Early* e = (Early*)v;
      hipStreamWaitEvent(e->p->s3, e->p->xev[c], 0);
      const int rc = trtri_advance<T>((const T*)e->p->A, e->p->N, (T*)e->p->Tm, (T*)e->p->S, e->ck[c], &e->st,
                                      e->p->s3, EARLY_WG_CAP, e->p->info + EARLY_CTR0, EARLY_CTR_PAIRS,
                                      EARLY_RESERVED_CUS, batch_of<T>(e->p));
      if (rc && !e->rc) e->rc = rc;
    };
    if ((rc = ensure_timing(p)) || (rc = ensure_split(p, s))) return rc;
    if (split_applies(p) && p->sc)
      rc = potrf_split<T>((T*)p->A, p->N, (T*)p->Tm, (T*)p->scal, p->info, (T*)p->snap, s, p->sc, p->s2, p->ev,
                          p->timing ? p->sev : nullptr, &p->n_syrk, &p->syrk_flop, 3, ctx.ck, p->xev, on_ck, &ctx, split_start(p), group_size(p->lookahead, p->B, p->N / DGP_TILE_HOST), &p->tune);
    else
      rc = potrf<T>((T*)p->A, p->N, (T*)p->Tm, (T*)p->scal, p->info, group_size(p->lookahead, p->B, p->N / DGP_TILE_HOST), s, p->s2, p->ev,
                    p->timing ? p->sev : nullptr, &p->n_syrk, &p->syrk_flop, 3, ctx.ck, p->xev, on_ck, &ctx, bt);
    if (rc || (rc = ctx.rc)) return rc;
    tick(p, TS_POTRF, 1, s);
    tick(p, TS_TRTRI, 0, s);
    hipEventRecord(p->xev[3], s);  // factorisation complete (s has joined the bulk stream)
    hipStreamWaitEvent(p->s3, p->xev[3], 0);
    if ((rc = trtri_advance<T>((const T*)p->A, p->N, (T*)p->Tm, (T*)p->S, nbk, &ctx.st, p->s3, 0, nullptr, 0, 0, bt)))
      return rc;
    hipEventRecord(p->xev[4], p->s3);
    hipStreamWaitEvent(s, p->xev[4], 0);
    tick(p, TS_TRTRI, 1, s);
  } else {
    if ((rc = run_potrf<T>(p, s))) return rc;
    tick(p, TS_POTRF, 1, s);
    tick(p, TS_TRTRI, 0, s);
    if ((rc = run_trtri<T>(p, s))) return rc;
    tick(p, TS_TRTRI, 1, s);
  }
  // The two triangular solves (HBM-bound: z = T r, alpha = T^T z read T twice) need T and r only, and K^^-1 = T^T T
  // (MFMA-bound) needs T only: with a second stream at hand the solves run BESIDE lauum instead of in front of it and join
  // before the gradient contraction, which needs both.  fp32 plans with the refinement keep the order: its scratch is S.
  // (the FUSED kernel -- K^^-1 with the gradient contraction in its epilogue, dgp_fused.hip -- needs alpha before it
  // starts: the solves then stay in front of it on the caller's stream)
  const bool fused = with_grad && p->tune.fused_grad && lauum_grad_applies(p->model, p->N, bt);
  hipStream_t ss = s;
  const int ov = (with_grad && !fused) ? solve_overlap_mode<T>(p, s, &ss) : 0;
  if (ov) {
    hipEventRecord(p->ev[0], s);
    hipStreamWaitEvent(ss, p->ev[0], 0);
  }
  tick(p, TS_SOLVE, 0, ss);
  if ((rc = run_solve<T>(p, r, ss))) return rc;
  if ((rc = run_refine<T>(p, theta, r, noise, ss))) return rc;
  tick(p, TS_SOLVE, 1, ss);
  if (ov) hipEventRecord(p->ev[1], ss);
  p->timed_valid = 0;
  p->have_inverse = 0;
  if (!with_grad && ov) hipStreamWaitEvent(s, p->ev[1], 0);
  if (fused) {
    // TIME_LAUUM then covers the fused launch (N^3/3 flop on MFMA + the contraction's vector work beside it), TIME_GRAD the
    // second reduction stage alone
    tick(p, TS_LAUUM, 0, s);
    void* staging = (p->pre && !p->pre_ready) ? p->ring.acquire(pre_scratch_bytes(p->B)) : nullptr;
    rc = lauum_grad<T>(p->model, p->d, (const T*)p->Tm, p->N, (T*)p->S, (const T*)p->Xt, (int)p->n, theta, (const T*)p->alpha,
                       (T*)p->gpart, (T*)out + DGP_OUT_DTHETA, s, bt, DGP_OUT_LEN, p->pre, p->pre_ready != 0, staging);
    if (staging) p->ring.commit(s);
    if (rc) return rc;
    p->pre_ready = 1;
    tick(p, TS_LAUUM, 1, s);
    tick(p, TS_GRAD, 0, s);
    tick(p, TS_GRAD, 1, s);
    p->timed_valid = p->timing && p->tev;
    p->have_inverse = 1;
  } else if (with_grad) {
    tick(p, TS_LAUUM, 0, s);
    if ((rc = run_lauum<T>(p, s))) return rc;
    tick(p, TS_LAUUM, 1, s);
    if (ov) hipStreamWaitEvent(s, p->ev[1], 0);
    tick(p, TS_GRAD, 0, s);
    if ((rc = run_grad<T>(p, theta, (T*)out + DGP_OUT_DTHETA, s))) return rc;
    tick(p, TS_GRAD, 1, s);
    p->timed_valid = p->timing && p->tev;
    p->have_inverse = 1;
  }
  const unsigned nwg = with_grad ? (unsigned)((p->n + 255) / 256) : 1u;
  finish_kernel<T><<<dim3(nwg, 1, (unsigned)bt.B), 256, 0, s>>>(
      (const T*)p->scal, p->info, (long)p->n, p->ntheta, !with_grad, (T*)out, bt.ws, bt.ws * (long)sizeof(T) / (long)sizeof(int), bt.ns,
      (const T*)p->alpha, (const T*)(with_grad ? p->dr_w : nullptr), (const T*)p->S, p->N, (T*)(with_grad ? dnoise : nullptr),
      (T*)(with_grad ? dr : nullptr), (T*)p->spart);
  p->have_factor = 1;
  return (int)hipGetLastError();
}

// ---- inference on the factorisation a plan holds.  Batched plans run every launch once for all sites (gridDim.z =
// sites, like the fit step): the caller's work area holds one slice per site (site stride = the single-site size).
static size_t predict_site_bytes(const dgp_plan* p, int64_t m) {
  const size_t M = (size_t)round_up(m, DGP_TILE_HOST), e = p->elem;
  return align_up(e * M * p->d) + 2 * align_up(e * (size_t)p->N * M) + 3 * align_up(e * M) + align_up(e * 2 * PREDICT_SPLIT * M);
}
// hyperparameters of a batch of more than 8 sites travel through the plan's device scratch + a pinned staging slot
struct PreSlot {
  dgp_plan* p;
  void* staging;
  hipStream_t s;
  PreSlot(dgp_plan* plan, hipStream_t st) : p(plan), staging(plan->pre ? plan->ring.acquire(pre_scratch_bytes(plan->B)) : nullptr), s(st) {
    plan->pre_ready = 0;  // the fit step's copy of the hyperparameters is overwritten
  }
  ~PreSlot() {
    if (staging) p->ring.commit(s);
  }
};

template <typename T>
static int cross(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* Ks, hipStream_t s, long wbs = 0) {
  const long M = round_up(m, DGP_TILE_HOST);
  Batch wb;  // the test points: [B][m][d] -> SoA in the work area
  wb.B = p->B;
  wb.ws = wbs;
  int rc = pack_x<T>((const T*)Xs, (int)m, p->d, M, (T*)work, s, wb);
  if (rc) return rc;
  PreSlot slot(p, s);
  return gram_cross<T>(p->model, p->d, (const T*)p->Xt, p->N, (int)p->n, (const T*)work, M, (int)m, theta, (T*)Ks, s, batch_of<T>(p), wbs,
                       p->pre, slot.staging);
}

template <typename T>
__global__ void copy_rows_kernel(const T* src, long n, T* dst, long src_stride) {  // dst [site][n] <- src at src_stride
  src = site(src, src_stride);
  dst = site(dst, n);
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

template <typename T>
static int predict_common(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* mean, T** V_out, T** Xst_out,
                          T** vpad_out, long* wbs_out, hipStream_t s) {
  const long M = round_up(m, DGP_TILE_HOST);
  const size_t e = sizeof(T);
  const long wbs = p->B > 1 ? (long)(predict_site_bytes(p, m) / e) : 0;
  char* w = (char*)work;
  T* Xst = (T*)w; w += align_up(e * M * p->d);
  T* Ks = (T*)w; w += align_up(e * (size_t)p->N * M);
  T* V = (T*)w; w += align_up(e * (size_t)p->N * M);
  T* kss = (T*)w; w += align_up(e * M);
  T* mpad = (T*)w; w += align_up(e * M);
  T* vpad = (T*)w; w += align_up(e * M);
  T* part = (T*)w;
  const unsigned Bz = (unsigned)p->B;
  int rc = cross<T>(p, theta, Xs, m, Xst, Ks, s, wbs);
  if (rc) return rc;
  Batch wb;
  wb.B = p->B;
  if ((rc = gram_diag<T>(p->model, p->d, Xst, M, (int)m, theta, kss, s, wb, wbs, p->pre))) return rc;
  if ((rc = predict_var<T>((const T*)p->Tm, p->N, Ks, M, V, (const T*)p->alpha, kss, part, mpad, vpad, s, batch_of<T>(p), wbs))) return rc;
  copy_rows_kernel<T><<<dim3((unsigned)((m + 255) / 256), 1, Bz), 256, 0, s>>>(mpad, m, (T*)mean, wbs);
  *V_out = V;
  *Xst_out = Xst;
  *vpad_out = vpad;
  *wbs_out = wbs;
  return (int)hipGetLastError();
}

template <typename T>
static int predict(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* mean, void* var,
                   hipStream_t s) {
  T *V, *Xst, *vpad;
  long wbs;
  int rc = predict_common<T>(p, theta, Xs, m, work, mean, &V, &Xst, &vpad, &wbs, s);
  if (rc) return rc;
  copy_rows_kernel<T><<<dim3((unsigned)((m + 255) / 256), 1, (unsigned)p->B), 256, 0, s>>>(vpad, m, (T*)var, wbs);
  return (int)hipGetLastError();
}

template <typename T>
static int post_cov(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* mean, void* cov,
                    hipStream_t s) {
  const long M = round_up(m, DGP_TILE_HOST);
  T *V, *Xst, *vpad;
  long wbs;
  int rc = predict_common<T>(p, theta, Xs, m, work, mean, &V, &Xst, &vpad, &wbs, s);
  if (rc) return rc;
  // zero "noise" for K(Xs, Xs); a batched plan's sites keep theirs at the work area's site stride
  hipError_t he = hipMemset2DAsync(vpad, sizeof(T) * (size_t)(wbs > 0 ? wbs : M), 0, sizeof(T) * M, (size_t)p->B, s);
  if (he != hipSuccess) return (int)he;
  Batch wb;
  wb.B = p->B;
  wb.ws = wbs;
  if ((rc = gram_sym<T>(p->model, p->d, Xst, M, (int)m, theta, vpad, (T*)cov, s, wb, p->pre, nullptr, M * M, wbs, true))) return rc;
  return posterior_cov<T>(V, p->N, M, (T*)cov, s, p->B, wbs);
}

struct VjpLayout {
  size_t Xst, Ks, g, beta, part, spart, total;
};
static VjpLayout vjp_layout(const dgp_plan* p, int64_t m) {
  const size_t M = (size_t)round_up(m, DGP_TILE_HOST), e = p->elem, N = (size_t)p->N;
  const size_t nb = N / 64, blocks_sym = nb * (nb + 1) / 2, blocks_cross = (M / 64) * nb;
  VjpLayout L;
  size_t o = 0;
  L.Xst = o; o += align_up(e * M * p->d);
  L.Ks = o; o += align_up(e * N * M);
  L.g = o; o += align_up(e * (N > M ? N : M));
  L.beta = o; o += align_up(e * N);
  L.part = o; o += align_up(e * 24 * (blocks_sym > blocks_cross ? blocks_sym : blocks_cross));
  L.spart = o; o += align_up(e * (size_t)solve_partials(p->N));
  L.total = o;
  return L;
}

template <typename T>
__global__ void matvec_cols_kernel(const T* Ks, long N, long Mp, int n, const T* alpha, T* out, long bs, long wbs, const int* ns) {
  // out[j] = sum_i Ks[i][j] alpha_i : one thread per column, coalesced across columns
  Ks = site(Ks, wbs);
  out = site(out, wbs);
  alpha = site(alpha, bs);
  n = site_n(ns, n);
  const long j = (long)blockIdx.x * 256 + threadIdx.x;
  if (j >= Mp) return;
  T acc = T(0);
  for (long i = 0; i < n; ++i) acc += Ks[i * Mp + j] * alpha[i];
  out[j] = acc;
}

template <typename T>
static int predict_mean(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* mean,
                        hipStream_t s) {
  const VjpLayout L = vjp_layout(p, m);
  const long M = round_up(m, DGP_TILE_HOST);
  const long wbs = p->B > 1 ? (long)(L.total / sizeof(T)) : 0;
  const Batch bt = batch_of<T>(p);
  char* w = (char*)work;
  int rc = cross<T>(p, theta, Xs, m, w + L.Xst, w + L.Ks, s, wbs);
  if (rc) return rc;
  matvec_cols_kernel<T><<<dim3((unsigned)((M + 255) / 256), 1, (unsigned)p->B), 256, 0, s>>>((const T*)(w + L.Ks), p->N, M, (int)p->n,
                                                                                           (const T*)p->alpha, (T*)(w + L.g), bt.ws, wbs, bt.ns);
  copy_rows_kernel<T><<<dim3((unsigned)((m + 255) / 256), 1, (unsigned)p->B), 256, 0, s>>>((const T*)(w + L.g), m, (T*)mean, wbs);
  return (int)hipGetLastError();
}

template <typename T>
static int mean_vjp(dgp_plan* p, const double* theta, const void* Xs, int64_t m, const void* wts, void* work,
                    void* dtheta, void* dr, void* dnoise, hipStream_t s) {
  const VjpLayout L = vjp_layout(p, m);
  const long M = round_up(m, DGP_TILE_HOST);
  const long wbs = p->B > 1 ? (long)(L.total / sizeof(T)) : 0;
  const Batch bt = batch_of<T>(p);
  char* w = (char*)work;
  T* Xst = (T*)(w + L.Xst);
  T* Ks = (T*)(w + L.Ks);
  T* g = (T*)(w + L.g);
  T* beta = (T*)(w + L.beta);
  int rc = cross<T>(p, theta, Xs, m, Xst, Ks, s, wbs);
  if (rc) return rc;
  if ((rc = gemv_rows<T>(Ks, p->N, M, (int)m, (const T*)wts, g, s, p->B, wbs))) return rc;   // g = K(X, X*) w
  if ((rc = symv_lower<T>((const T*)p->S, p->N, g, (int)p->n, (const T*)p->alpha, beta, (T*)(w + L.spart),
                          (T*)dnoise, s, bt, wbs)))                                         // beta = K^^-1 g
    return rc;
  if (dr) copy_rows_kernel<T><<<dim3((unsigned)((p->n + 255) / 256), 1, (unsigned)p->B), 256, 0, s>>>(beta, p->n, (T*)dr, wbs);
  PreSlot slot(p, s);
  return mean_vjp_grad<T>(p->model, p->d, (const T*)p->Xt, p->N, (int)p->n, Xst, M, (int)m, theta, (const T*)p->alpha,
                          beta, (const T*)wts, (T*)(w + L.part), (T*)dtheta, s, bt, wbs, p->B > 1 ? p->ntheta : 0, p->pre, slot.staging);
}

#define DGP_BY_DTYPE(p, CALL64, CALL32) ((p)->dtype == DGP_F64 ? (CALL64) : (CALL32))
#define DGP_CHECK_PLAN(p)                                                        \
  if (!(p)) return fail(DGP_E_ARG, "null plan");                                 \
  if (!(p)->ws) return fail(DGP_E_WORKSPACE, "plan has no workspace: call dgp_plan_set_workspace")
#define DGP_SINGLE_SITE(p) \
  if ((p)->B != 1)         \
  return fail(DGP_E_STATE, "batched plans support dgp_set_inputs / dgp_fit_step / dgp_factorize / dgp_predict / dgp_predict_mean / dgp_mean_vjp only")
static int wrap(int rc, const char* where) {
  if (rc > 0) return hipfail((hipError_t)rc, where);
  if (rc < 0) return fail(rc, where);
  return 0;
}

extern "C" {

int dgp_set_inputs(dgp_plan* p, const void* X, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!X) return fail(DGP_E_ARG, "dgp_set_inputs: null X");
  hipStream_t s = (hipStream_t)stream;
  int rc = DGP_BY_DTYPE(p, pack_x<double>((const double*)X, (int)p->n, p->d, p->N, (double*)p->Xt, s, batch_of<double>(p)),
                        pack_x<float>((const float*)X, (int)p->n, p->d, p->N, (float*)p->Xt, s, batch_of<float>(p)));
  p->have_inputs = 1;
  p->have_factor = 0;
  return wrap(rc, "dgp_set_inputs");
}

int dgp_fit_step(dgp_plan* p, const double* theta, const void* r, const void* noise, void* out, void* dr,
                 void* dnoise, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !r || !noise || !out) return fail(DGP_E_ARG, "dgp_fit_step: null argument");
  if (!p->have_inputs) return fail(DGP_E_STATE, "dgp_fit_step: call dgp_set_inputs first");
  hipStream_t s = (hipStream_t)stream;
  int rc = DGP_BY_DTYPE(p, fit_step<double>(p, theta, r, noise, out, dr, dnoise, 1, s),
                        fit_step<float>(p, theta, r, noise, out, dr, dnoise, 1, s));
  return wrap(rc, "dgp_fit_step");
}

int dgp_factorize(dgp_plan* p, const double* theta, const void* r, const void* noise, void* out, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !r || !noise || !out) return fail(DGP_E_ARG, "dgp_factorize: null argument");
  if (!p->have_inputs) return fail(DGP_E_STATE, "dgp_factorize: call dgp_set_inputs first");
  hipStream_t s = (hipStream_t)stream;
  int rc = DGP_BY_DTYPE(p, fit_step<double>(p, theta, r, noise, out, nullptr, nullptr, 0, s),
                        fit_step<float>(p, theta, r, noise, out, nullptr, nullptr, 0, s));
  return wrap(rc, "dgp_factorize");
}

size_t dgp_predict_workspace_bytes(const dgp_plan* p, int64_t m) {
  if (!p || m <= 0) return 0;
  return predict_site_bytes(p, m) * (size_t)p->B;
}

int dgp_predict(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, size_t work_bytes,
                void* mean, void* var, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !Xs || !work || !mean || !var || m <= 0) return fail(DGP_E_ARG, "dgp_predict: null argument");
  if (!p->have_factor) return fail(DGP_E_STATE, "dgp_predict: no factorisation in the plan (call dgp_factorize)");
  if (work_bytes < dgp_predict_workspace_bytes(p, m)) return fail(DGP_E_WORKSPACE, "dgp_predict: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rc = DGP_BY_DTYPE(p, predict<double>(p, theta, Xs, m, work, mean, var, s), predict<float>(p, theta, Xs, m, work, mean, var, s));
  return wrap(rc, "dgp_predict");
}

int dgp_posterior_cov(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, size_t work_bytes,
                      void* mean, void* cov, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !Xs || !work || !mean || !cov || m <= 0) return fail(DGP_E_ARG, "dgp_posterior_cov: null argument");
  if (!p->have_factor) return fail(DGP_E_STATE, "dgp_posterior_cov: no factorisation in the plan (call dgp_factorize)");
  if (work_bytes < dgp_predict_workspace_bytes(p, m)) return fail(DGP_E_WORKSPACE, "dgp_posterior_cov: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  int rc = DGP_BY_DTYPE(p, post_cov<double>(p, theta, Xs, m, work, mean, cov, s),
                        post_cov<float>(p, theta, Xs, m, work, mean, cov, s));
  return wrap(rc, "dgp_posterior_cov");
}

size_t dgp_mean_vjp_workspace_bytes(const dgp_plan* p, int64_t m) { return (p && m > 0) ? vjp_layout(p, m).total * (size_t)p->B : 0; }

int dgp_predict_mean(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, size_t work_bytes,
                     void* mean, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !Xs || !work || !mean || m <= 0) return fail(DGP_E_ARG, "dgp_predict_mean: null argument");
  if (!p->have_factor) return fail(DGP_E_STATE, "dgp_predict_mean: no factorisation in the plan");
  if (work_bytes < dgp_mean_vjp_workspace_bytes(p, m)) return fail(DGP_E_WORKSPACE, "dgp_predict_mean: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  const int rc = DGP_BY_DTYPE(p, predict_mean<double>(p, theta, Xs, m, work, mean, s), predict_mean<float>(p, theta, Xs, m, work, mean, s));
  return wrap(rc, "dgp_predict_mean");
}

int dgp_mean_vjp(dgp_plan* p, const double* theta, const void* Xs, int64_t m, const void* wts, void* work,
                 size_t work_bytes, void* dtheta, void* dr, void* dnoise, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !Xs || !wts || !work || !dtheta || m <= 0) return fail(DGP_E_ARG, "dgp_mean_vjp: null argument");
  if (!p->have_factor || !p->have_inverse)
    return fail(DGP_E_STATE, "dgp_mean_vjp: needs K^^-1 and alpha from dgp_fit_step at the same theta");
  if (work_bytes < dgp_mean_vjp_workspace_bytes(p, m)) return fail(DGP_E_WORKSPACE, "dgp_mean_vjp: workspace too small");
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, mean_vjp<double>(p, theta, Xs, m, wts, work, dtheta, dr, dnoise, s),
                           mean_vjp<float>(p, theta, Xs, m, wts, work, dtheta, dr, dnoise, s)), "dgp_mean_vjp");
}

int dgp_sample_draws(int dtype, const void* L, int64_t m, const void* Z, int64_t ndraw, const void* mean, void* out,
                     void* stream) {
  if (dtype != DGP_F64 && dtype != DGP_F32) return fail(DGP_E_ARG, "dgp_sample_draws: dtype must be 0 (f64) or 1 (f32)");
  if (!L || !Z || !out || m <= 0 || ndraw <= 0 || m > (1 << 20) || ndraw > (1 << 24))
    return fail(DGP_E_ARG, "dgp_sample_draws: null argument / bad size");
  const long M = round_up(m, DGP_TILE_HOST), Q = round_up(ndraw, DGP_TILE_HOST);
  hipStream_t s = (hipStream_t)stream;
  const int rc = dtype == DGP_F64
                     ? sample_draws<double>((const double*)L, M, (const double*)Z, Q, (const double*)mean, (int)m, (int)ndraw, (double*)out, s)
                     : sample_draws<float>((const float*)L, M, (const float*)Z, Q, (const float*)mean, (int)m, (int)ndraw, (float*)out, s);
  return wrap(rc, "dgp_sample_draws");
}

int dgp_plan_set_timing(dgp_plan* p, int enabled) {
  if (!p) return fail(DGP_E_ARG, "null plan");
  p->timing = enabled ? 1 : 0;
  p->timed_valid = 0;
  return 0;
}

int dgp_plan_get_timing(dgp_plan* p, double* ms_out) {
  if (!p || !ms_out) return fail(DGP_E_ARG, "dgp_plan_get_timing: null argument");
  if (!p->timing || !p->tev || !p->timed_valid)
    return fail(DGP_E_STATE, "dgp_plan_get_timing: enable timing and run dgp_fit_step first");
  static const int map[TS_COUNT] = {DGP_TIME_GRAM, DGP_TIME_POTRF, DGP_TIME_TRTRI, DGP_TIME_LAUUM, DGP_TIME_SOLVE,
                                    DGP_TIME_GRAD};
  for (int i = 0; i < DGP_TIME_COUNT; ++i) ms_out[i] = 0.0;
  hipError_t e = hipEventSynchronize(p->tev[2 * TS_GRAD + 1]);
  if (e != hipSuccess) return hipfail(e, "hipEventSynchronize");
  for (int st = 0; st < TS_COUNT; ++st) {
    float ms = 0.f;
    e = hipEventElapsedTime(&ms, p->tev[2 * st], p->tev[2 * st + 1]);
    if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime");
    ms_out[map[st]] = ms;
  }
  double sum = 0.0;
  for (int i = 0; i < p->n_syrk; ++i) {
    float ms = 0.f;
    e = hipEventSynchronize(p->sev[2 * i + 1]);
    if (e == hipSuccess) e = hipEventElapsedTime(&ms, p->sev[2 * i], p->sev[2 * i + 1]);
    if (e != hipSuccess) return hipfail(e, "hipEventElapsedTime(syrk)");
    sum += ms;
  }
  ms_out[DGP_TIME_SYRK_SUM] = sum;
  ms_out[DGP_TIME_SYRK_N] = p->n_syrk;
  ms_out[DGP_TIME_SYRK_FLOP] = p->syrk_flop;
  return 0;
}

int dgp_stage_gram(dgp_plan* p, const double* theta, const void* noise, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!theta || !noise) return fail(DGP_E_ARG, "dgp_stage_gram: null argument");
  if (!p->have_inputs) return fail(DGP_E_STATE, "dgp_stage_gram: call dgp_set_inputs first");
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, run_gram<double>(p, theta, noise, s), run_gram<float>(p, theta, noise, s)), "dgp_stage_gram");
}
int dgp_stage_potrf(dgp_plan* p, void* stream) {
  DGP_CHECK_PLAN(p);
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, run_potrf<double>(p, s), run_potrf<float>(p, s)), "dgp_stage_potrf");
}
int dgp_stage_trtri(dgp_plan* p, void* stream) {
  DGP_CHECK_PLAN(p);
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, run_trtri<double>(p, s), run_trtri<float>(p, s)), "dgp_stage_trtri");
}
int dgp_stage_lauum(dgp_plan* p, void* stream) {
  DGP_CHECK_PLAN(p);
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, run_lauum<double>(p, s), run_lauum<float>(p, s)), "dgp_stage_lauum");
}
int dgp_stage_solve(dgp_plan* p, const void* r, void* stream) {
  DGP_CHECK_PLAN(p);
  if (!r) return fail(DGP_E_ARG, "dgp_stage_solve: null r");
  hipStream_t s = (hipStream_t)stream;
  int rc = DGP_BY_DTYPE(p, run_solve<double>(p, r, s), run_solve<float>(p, r, s));
  if (!rc) p->have_factor = 1;
  return wrap(rc, "dgp_stage_solve");
}
int dgp_stage_grad(dgp_plan* p, const double* theta, void* dtheta, void* stream) {
  DGP_CHECK_PLAN(p);
  DGP_SINGLE_SITE(p);
  if (!theta || !dtheta) return fail(DGP_E_ARG, "dgp_stage_grad: null argument");
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, run_grad<double>(p, theta, dtheta, s), run_grad<float>(p, theta, dtheta, s)), "dgp_stage_grad");
}
int dgp_cross_gram(dgp_plan* p, const double* theta, const void* Xs, int64_t m, void* work, void* Ks, void* stream) {
  DGP_CHECK_PLAN(p);
  DGP_SINGLE_SITE(p);
  if (!theta || !Xs || !work || !Ks || m <= 0) return fail(DGP_E_ARG, "dgp_cross_gram: null argument");
  hipStream_t s = (hipStream_t)stream;
  return wrap(DGP_BY_DTYPE(p, cross<double>(p, theta, Xs, m, work, Ks, s), cross<float>(p, theta, Xs, m, work, Ks, s)),
              "dgp_cross_gram");
}

// The probe runs on the caller-stream's EARLY-INVERSE stream (lowest priority), not on a stream of its own: a process has
// few hardware queues, and one more stream pushes a later single-site plan's rest / bulk streams into a shared queue
// (measured: bench.py's single-site loop 12.1 -> 18.5 ms after a probe on a fresh torch stream).  The load runs on
// `stream` itself (and the library's other internal streams), so the two overlap.
int dgp_debug_clock_probe(void* out_dev, int nwg, double seconds, void* stream) {
  if (!out_dev || nwg < 1 || nwg > 1024 || !(seconds > 0.0) || seconds > 5.0) return fail(DGP_E_ARG, "dgp_debug_clock_probe: bad argument");
  StreamSet* st = stream_set((hipStream_t)stream);
  if (!st) return fail(DGP_E_ARG, "out of host memory");
  const int rc = make_stream(&st->early, false);
  if (rc) return rc;
  clock_probe_kernel<<<dim3((unsigned)nwg), 64, 0, st->early>>>((unsigned long long*)out_dev, (long long)(seconds * 1e8));
  return wrap((int)hipGetLastError(), "dgp_debug_clock_probe");
}

}  // extern "C"
