// Does v_mfma_f32_16x16x4_f32 accumulate with a rounding BIAS?  One wave chains K/4 MFMAs on a 16 x K by K x 16 product;
// the result is compared with (a) the exact sum in double, (b) a k-ordered fmaf chain on the VALU (what
// cdna_hip_programming.md says the MFMA is bit-for-bit).  Reports the MEAN SIGNED relative error over the 256 outputs
// and many trials: a round-to-nearest chain has mean ~ 0, a truncating one ~ -K * 2^-25.
// build: hipcc -O2 --offload-arch=gfx950 scripts/mfma_f32_bias.hip -o scripts/mfma_f32_bias
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ void chain(const float* A, const float* B, int K, float* Cm, float* Cf, float c0) {
  const int lane = threadIdx.x, i = lane & 15, q = lane >> 4;
  const float* a = A + (size_t)blockIdx.x * 16 * K;
  const float* b = B + (size_t)blockIdx.x * 16 * K;
  f4 acc = {c0, c0, c0, c0};
  for (int k = 0; k < K; k += 4) acc = __builtin_amdgcn_mfma_f32_16x16x4f32(a[i * K + k + q], b[i * K + k + q], acc, 0, 0, 0);
  // C: col = lane & 15, row = 4 q + r ; A row index = i (lane & 15 of the A operand), B col index = lane & 15
  for (int r = 0; r < 4; ++r) Cm[(size_t)blockIdx.x * 256 + (4 * q + r) * 16 + i] = acc[r];
  // the same sums as a k-ordered fmaf chain
  for (int r = 0; r < 4; ++r) {
    const int row = 4 * q + r, col = i;
    float s = c0;
    for (int k = 0; k < K; ++k) s = fmaf(a[row * K + k], b[col * K + k], s);
    Cf[(size_t)blockIdx.x * 256 + row * 16 + col] = s;
  }
}

int main(int argc, char** argv) {
  const int trials = 256;
  for (int mode = 0; mode < 3; ++mode)
    for (int K : {64, 512, 4096, 16384}) {
      std::vector<float> A((size_t)trials * 16 * K), B(A.size());
      srand(1 + mode);
      for (size_t x = 0; x < A.size(); ++x) {
        float u = rand() / (float)RAND_MAX, v = rand() / (float)RAND_MAX;
        if (mode == 1) { u = 2 * u - 1; v = 2 * v - 1; }        // signed products
        A[x] = u; B[x] = v;
      }
      const float c0 = mode == 2 ? -0.26f * K : 0.f;  // mode 2: start at -C and climb towards ~0 like the Schur update
      float *dA, *dB, *dC, *dF;
      hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, A.size() * 4); hipMalloc(&dC, trials * 256 * 4); hipMalloc(&dF, trials * 256 * 4);
      hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), A.size() * 4, hipMemcpyHostToDevice);
      chain<<<trials, 64>>>(dA, dB, K, dC, dF, c0);
      std::vector<float> C(trials * 256), F(trials * 256);
      hipMemcpy(C.data(), dC, C.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(F.data(), dF, F.size() * 4, hipMemcpyDeviceToHost);
      double sm = 0, sf = 0, am = 0, scale = 0; long same = 0;
      for (int t = 0; t < trials; ++t)
        for (int row = 0; row < 16; ++row)
          for (int col = 0; col < 16; ++col) {
            double ex = c0, mag = fabs(c0);
            for (int k = 0; k < K; ++k) { double pr = (double)A[((size_t)t * 16 + row) * K + k] * B[((size_t)t * 16 + col) * K + k]; ex += pr; mag += fabs(pr); }
            const float m = C[(size_t)t * 256 + row * 16 + col], f = F[(size_t)t * 256 + row * 16 + col];
            sm += (m - ex) / mag; sf += (f - ex) / mag; am += fabs(m - ex) / mag; scale += 1; same += (m == f);
          }
      printf("mode %d K %5d: MFMA mean signed err / sum|ab| %+.3e  (mean |err| %.3e)   fmaf chain %+.3e   bitwise equal to the fmaf chain: %ld of %d\n",
             mode, K, sm / scale, am / scale, sf / scale, same, trials * 256);
      hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dF);
    }
  return 0;
}
