/* A host in plain C on the C ABI of libdgp_hip.so: no Python, no torch -- the boundary a C/C++/other-language caller
 * of the engine would use (include/dgp_hip.h).
 *
 *   gcc -std=c11 -O2 -D__HIP_PLATFORM_AMD__ -I include -I /opt/rocm/include examples/c_host.c -L discontinuum_amd -ldgp_hip \
 *       -L /opt/rocm/lib -lamdhip64 -Wl,-rpath,$PWD/discontinuum_amd -Wl,-rpath,/opt/rocm/lib -o build/c_host
 *   build/c_host [n]     ->  one line:  n NLL logdet quad info dtheta[0] sum_dr mean[0] var[0]
 *
 * Inputs come from a 64-bit LCG so that tests/test_gpu_cabi.py can rebuild them in numpy and compare with the Python
 * path and the oracle. */
#include <hip/hip_runtime_api.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include "dgp_hip.h"

static uint64_t lcg_state = 12345;
static double lcg_uniform(void) { /* in (0, 1) */
  lcg_state = lcg_state * 6364136223846793005ULL + 1442695040888963407ULL;
  return ((double)(lcg_state >> 11) + 0.5) / 9007199254740992.0;
}

#define HIP(x)                                                                      \
  do {                                                                              \
    hipError_t e_ = (x);                                                            \
    if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } \
  } while (0)
#define DGP(x)                                                                      \
  do {                                                                              \
    if ((x) != 0) { fprintf(stderr, "%s: %s\n", #x, dgp_last_error()); return 3; }  \
  } while (0)

int main(int argc, char** argv) {
  const int64_t n = argc > 1 ? atoll(argv[1]) : 200;
  const int d = 3, m = 5;
  const int ntheta = dgp_model_ntheta(DGP_MODEL_LOADEST, d);
  double *X = malloc(sizeof(double) * n * d), *r = malloc(sizeof(double) * n), *noise = malloc(sizeof(double) * n);
  double theta[32], out[DGP_OUT_LEN], Xs[5 * 3], mean[5], var[5];
  for (int64_t i = 0; i < n; ++i) {
    X[i * d + 0] = -16.0 + 32.0 * (double)i / (double)n + 0.01 * lcg_uniform(); /* increasing times */
    for (int j = 1; j < d; ++j) X[i * d + j] = 4.0 * lcg_uniform() - 2.0;
    r[i] = 2.0 * lcg_uniform() - 1.0;
    noise[i] = 0.01;
  }
  for (int p = 0; p < ntheta; ++p) theta[p] = 0.5 + 0.05 * p;
  for (int i = 0; i < m * d; ++i) Xs[i] = X[(i / d) * 7 * d + i % d] + 0.05;

  dgp_plan* plan = NULL;
  DGP(dgp_plan_create(DGP_MODEL_LOADEST, DGP_F64, n, d, &plan));
  const size_t ws_bytes = dgp_plan_workspace_bytes(plan);
  void *ws, *dX, *dr, *dnoise, *dout, *ddr, *ddnoise, *dXs, *dwork, *dmean, *dvar;
  HIP(hipMalloc(&ws, ws_bytes)); /* hipMalloc returns 256-byte aligned memory */
  DGP(dgp_plan_set_workspace(plan, ws, ws_bytes));
  HIP(hipMalloc(&dX, sizeof(double) * n * d));
  HIP(hipMalloc(&dr, sizeof(double) * n));
  HIP(hipMalloc(&dnoise, sizeof(double) * n));
  HIP(hipMalloc(&dout, sizeof(double) * DGP_OUT_LEN));
  HIP(hipMalloc(&ddr, sizeof(double) * n));
  HIP(hipMalloc(&ddnoise, sizeof(double) * n));
  HIP(hipMemcpy(dX, X, sizeof(double) * n * d, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dr, r, sizeof(double) * n, hipMemcpyHostToDevice));
  HIP(hipMemcpy(dnoise, noise, sizeof(double) * n, hipMemcpyHostToDevice));
  hipStream_t stream;
  HIP(hipStreamCreate(&stream));
  DGP(dgp_set_inputs(plan, dX, stream));
  DGP(dgp_fit_step(plan, theta, dr, dnoise, dout, ddr, ddnoise, stream));
  HIP(hipMemcpyAsync(out, dout, sizeof(out), hipMemcpyDeviceToHost, stream));
  /* prediction from the factorisation the fit step left in the plan */
  const size_t pw = dgp_predict_workspace_bytes(plan, m);
  HIP(hipMalloc(&dwork, pw));
  HIP(hipMalloc(&dXs, sizeof(Xs)));
  HIP(hipMalloc(&dmean, sizeof(mean)));
  HIP(hipMalloc(&dvar, sizeof(var)));
  HIP(hipMemcpyAsync(dXs, Xs, sizeof(Xs), hipMemcpyHostToDevice, stream));
  DGP(dgp_predict(plan, theta, dXs, m, dwork, pw, dmean, dvar, stream));
  HIP(hipMemcpyAsync(mean, dmean, sizeof(mean), hipMemcpyDeviceToHost, stream));
  HIP(hipMemcpyAsync(var, dvar, sizeof(var), hipMemcpyDeviceToHost, stream));
  HIP(hipStreamSynchronize(stream));
  printf("%lld %.17g %.17g %.17g %d %.17g %.17g %.17g %.17g\n", (long long)n, out[DGP_OUT_NLL], out[DGP_OUT_LOGDET],
         out[DGP_OUT_QUAD], (int)out[DGP_OUT_INFO], out[DGP_OUT_DTHETA], out[DGP_OUT_SUM_DR], mean[0], var[0]);
  DGP(dgp_plan_destroy(plan));
  hipFree(ws); hipFree(dX); hipFree(dr); hipFree(dnoise); hipFree(dout); hipFree(ddr); hipFree(ddnoise);
  hipFree(dwork); hipFree(dXs); hipFree(dmean); hipFree(dvar);
  free(X); free(r); free(noise);
  return 0;
}
