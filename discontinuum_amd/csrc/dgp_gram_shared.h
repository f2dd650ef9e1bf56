// dgp_gram_shared.h -- pieces of the Gram kernels that the fused K^^-1 + gradient kernel (dgp_fused.hip) shares with dgp_gram.hip:
// staging of a strip of per-point features in LDS, 4-element vector loads / stores, the deterministic second reduction stage.
#pragma once
#include "dgp_internal.h"
#include "dgp_models.h"

namespace dgp {

template <typename T, typename M>
__device__ __forceinline__ void stage_strip(const T* __restrict__ Xt, long N, long base, const typename M::Pre& pre,
                                            T (*sf)[64], int lane) {
  T x[M::NX], f[M::NF];
#pragma unroll
  for (int c = 0; c < M::NX; ++c) x[c] = Xt[(long)c * N + base + lane];
  M::features(x, pre, f);
#pragma unroll
  for (int c = 0; c < M::NF; ++c) sf[c][lane] = f[c];
}

template <typename T>
__device__ __forceinline__ void store4(T* dst, const T (&v)[4]);
template <>
__device__ __forceinline__ void store4<double>(double* dst, const double (&v)[4]) {
  dgp_d2 a = {v[0], v[1]}, b = {v[2], v[3]};
  reinterpret_cast<dgp_d2*>(dst)[0] = a;
  reinterpret_cast<dgp_d2*>(dst)[1] = b;
}
template <>
__device__ __forceinline__ void store4<float>(float* dst, const float (&v)[4]) {
  dgp_f4 a = {v[0], v[1], v[2], v[3]};
  *reinterpret_cast<dgp_f4*>(dst) = a;
}
template <typename T>
__device__ __forceinline__ void load4(const T* src, T (&v)[4]);
template <>
__device__ __forceinline__ void load4<double>(const double* src, double (&v)[4]) {
  dgp_d2 a = reinterpret_cast<const dgp_d2*>(src)[0], b = reinterpret_cast<const dgp_d2*>(src)[1];
  v[0] = a[0]; v[1] = a[1]; v[2] = b[0]; v[3] = b[1];
}
template <>
__device__ __forceinline__ void load4<float>(const float* src, float (&v)[4]) {
  dgp_f4 a = *reinterpret_cast<const dgp_f4*>(src);
  v[0] = a[0]; v[1] = a[1]; v[2] = a[2]; v[3] = a[3];
}

// deterministic second stage: one block, fixed summation order
template <typename T>
__global__ __launch_bounds__(256) void grad_reduce_kernel(const T* __restrict__ partials, long nblk, int nt,
                                                          T* __restrict__ out, int accumulate, long bs, long os) {
  // one workgroup per hyperparameter; fixed strided order + fixed tree => bitwise reproducible
  partials = site(partials, bs);
  out = site(out, os);
  __shared__ T red[256];
  const int p = blockIdx.x;
  T v = T(0);
  for (long b = threadIdx.x; b < nblk; b += 256) v += partials[b * DGP_MAX_THETA + p];
  red[threadIdx.x] = v;
  __syncthreads();
  for (int s = 128; s > 0; s >>= 1) {
    if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) out[p] = accumulate ? out[p] + red[0] : red[0];
  (void)nt;
}


}  // namespace dgp
