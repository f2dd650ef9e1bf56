// Isolated timing + check of the diagonal-block kernel (L_kk and L_kk^-1) for block sizes 128 / 64 / 32.
#include <math.h>
#include <stdio.h>
#include <vector>
#include "../discontinuum_amd/csrc/dgp_chol.hip"
using namespace dgp;
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

template <typename T, int BS, bool FAST = false>
int run(const char* name) {
  const long ld = 1024;
  std::vector<T> h(ld * ld, 0.0); std::vector<double> G(BS * BS);
  srand(1);
  for (auto& v : G) v = (double)rand() / RAND_MAX - 0.5;
  for (int i = 0; i < BS; ++i)
    for (int j = 0; j < BS; ++j) {
      double s = (i == j) ? 0.5 : 0.0;
      for (int k = 0; k < BS; ++k) s += G[i * BS + k] * G[j * BS + k] / BS;
      h[i * ld + j] = (T)s;
    }
  T *A, *Tm, *logdet; int* info;
  CK(hipMalloc(&A, ld * ld * sizeof(T))); CK(hipMalloc(&Tm, ld * ld * sizeof(T))); CK(hipMalloc(&logdet, 8)); CK(hipMalloc(&info, 4));
  CK(hipMemset(logdet, 0, 8)); CK(hipMemset(info, 0, 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float best = 1e9;
  for (int rep = 0; rep < 5; ++rep) {
    CK(hipMemcpy(A, h.data(), ld * ld * sizeof(T), hipMemcpyHostToDevice));
    CK(hipEventRecord(e0));
    launch_diag<T>(A, ld, 0, Tm, logdet, info, 0, Batch());
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    if (ms < best) best = ms;
  }
  std::vector<T> L(ld * ld), X(ld * ld);
  CK(hipMemcpy(L.data(), A, ld * ld * sizeof(T), hipMemcpyDeviceToHost));
  CK(hipMemcpy(X.data(), Tm, ld * ld * sizeof(T), hipMemcpyDeviceToHost));
  double e1max = 0, e2max = 0;
  for (int i = 0; i < BS; ++i)
    for (int j = 0; j < BS; ++j) {
      double s = 0, t = 0;
      for (int k = 0; k < BS; ++k) { s += L[i * ld + k] * L[j * ld + k]; t += X[i * ld + k] * L[k * ld + j]; }
      e1max = fmax(e1max, fabs(s - h[i * ld + j]));
      e2max = fmax(e2max, fabs(t - (i == j)));
    }
  printf("%-10s BS=%d: %.1f us   |LL^T-A|=%.2e  |XL-I|=%.2e\n", name, BS, best * 1e3, e1max, e2max);
#ifdef DGP_DIAG_PROFILE
  long long st[16];
  CK(hipMemcpyFromSymbol(st, HIP_SYMBOL(dgp_diag_prof), sizeof(st)));
  const char* ph[] = {"load A -> LDS", "first GJ16", "main loop (7 block steps)", "store L", "inverse phase", "store diag X", "logdet"};
  for (int i = 0; i < 7; ++i) printf("    %-28s %8lld cycles\n", ph[i], st[i + 1] - st[i]);
  printf("    total %lld cycles (wave 0 thread 0, shader clock)\n", st[7] - st[0]);
#endif
  return 0;
}
int main() { run<double,128,true>("f64"); run<float,128,true>("f32"); return 0; }
