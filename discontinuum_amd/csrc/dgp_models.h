// dgp_models.h -- per-pair evaluators of discontinuum's two composite covariance functions and
// their derivatives w.r.t. the CONSTRAINED hyperparameters (softplus / interval chain rule and the
// priors stay on the host, in torch).  Everything here is evaluated in registers per matrix entry.
//
//   Loadest<T, D>  Scale(Periodic x Matern52)[time] + Scale(RBF-ARD)[covariates] + Scale(Matern32-ARD)[all]
//                  reference: src/loadest_gp/models/gpytorch.py:61-128
//   Rating<T>      g g' (shift1 + shift2) + (1-g)(1-g')' bend + base + periodic, Matern factors on
//                  (time, log(stage + 1e-6)), g = 1/(1+exp(20 (stage - b)))
//                  reference: src/rating_gp/models/gpytorch.py:205-372, src/rating_gp/models/kernels.py:242-382
//
// gpytorch formulas: RBF exp(-1/2 r^2); Matern32 (1+sqrt3 r)exp(-sqrt3 r);
// Matern52 (1+sqrt5 r+5r^2/3)exp(-sqrt5 r); Periodic exp(-2 sin^2(pi dx/p)/l).
// Parameter order = oracle/gp_oracle.py::loadest_gram / rating_gram.
#pragma once
#include <vector>
#include "dgp_common.h"

#define DGP_MAX_THETA 24
#define DGP_MODEL_LOADEST 0
#define DGP_MODEL_RATING 1

namespace dgp {

// ---- per-pair elementary functions.  In fp64 the Gram kernels are VALU-bound (SURVEY section 8d prices them against
// HBM), so every exponent / root of a covariance term goes through a version that drops what cannot happen here
// (positive or non-finite arguments, denormal results) -- the library versions spend a third of their instructions on
// those cases.  The periodic term needs no per-pair sine at all: sin and cos of pi t / p are per-POINT features
// (evaluated in double once per 64 x 64 tile strip) and the pair uses the angle-difference identities.
// e^x for x <= 0 in double: x = n ln2 / 64 + r with |r| <= ln2 / 128, e^x = 2^(n >> 6) * 2^((n & 63) / 64) * e^r.  The 64
// table entries 2^(j/64) (correctly rounded) sit in LDS -- a per-lane gather, ds_read_b64 -- and e^r is a degree-5
// polynomial (truncation r^6 / 720 < 3.5e-17 relative).  13 VALU + 1 LDS instruction against 21 VALU for the table-free
// version (same range reduction by ln2, degree-13 polynomial) this replaces.  Every kernel that evaluates a covariance
// in double calls exp_table_init() before the barrier that precedes its first pair().
static __constant__ double dgp_exp_tab_c[64] = {
    1.0, 1.0108892860517005, 1.0218971486541166, 1.0330248790212284,
    1.0442737824274138, 1.0556451783605572, 1.0671404006768237, 1.0787607977571199,
    1.0905077326652577, 1.102382583307841, 1.1143867425958924, 1.1265216186082418,
    1.1387886347566916, 1.1511892299529827, 1.1637248587775775, 1.1763969916502812,
    1.189207115002721, 1.202156731452703, 1.215247359980469, 1.22848053610687,
    1.241857812073484, 1.255380757024691, 1.2690509571917332, 1.2828700160787783,
    1.2968395546510096, 1.3109612115247644, 1.3252366431597413, 1.339667524053303,
    1.3542555469368927, 1.3690024229745905, 1.383909881963832, 1.3989796725383112,
    1.4142135623730951, 1.42961333839197, 1.4451808069770467, 1.460917794180647,
    1.4768261459394993, 1.4929077282912648, 1.5091644275934228, 1.5255981507445384,
    1.5422108254079407, 1.559004400237837, 1.5759808451078865, 1.593142151342267,
    1.6104903319492543, 1.6280274218573478, 1.645755478153965, 1.6636765803267364,
    1.681792830507429, 1.7001063537185235, 1.718619298122478, 1.7373338352737062,
    1.7562521603732995, 1.7753764925265212, 1.7947090750031072, 1.8142521755003989,
    1.8340080864093424, 1.8539791250833855, 1.8741676341103, 1.8945759815869656,
    1.9152065613971474, 1.9360617934922943, 1.9571441241754002, 1.978456026387951};
static __shared__ double dgp_exp_tab[64];
template <typename T>
__device__ __forceinline__ void exp_table_init() {
  if (sizeof(T) == 8 && threadIdx.x < 64) dgp_exp_tab[threadIdx.x] = dgp_exp_tab_c[threadIdx.x];
}
__device__ __forceinline__ double exp_nonpos(double x) {
  x = x < -708.0 ? -708.0 : x;  // below e^-708 = 3e-308 the result only matters as "zero"; a NaN stays a NaN
  const double n = __builtin_rint(x * 92.33248261689366);
  double r = __builtin_fma(n, -0.010830424695086549, x);  // ln2 / 64 in two parts: n * hi is exact for |n| < 2^20
  r = __builtin_fma(n, -1.162596423439437e-12, r);
  const int ni = (int)n;
  double p = 1.0 / 120.0;
  p = __builtin_fma(p, r, 1.0 / 24.0);
  p = __builtin_fma(p, r, 1.0 / 6.0);
  p = __builtin_fma(p, r, 0.5);
  p = __builtin_fma(p, r, 1.0);
  p = __builtin_fma(p, r, 1.0);
  return ldexp(dgp_exp_tab[ni & 63] * p, ni >> 6);
}
// float: the hardware transcendental (v_exp_f32 = 2^x, 1 ulp) on x log2(e) -- two instructions where the library expf
// spends twelve on range reduction, ldexp and special cases that cannot occur for x <= 0 (a result below 2^-126 flushes to
// 0, which is what it stands for; a NaN stays a NaN).  The argument's rounding costs |x| eps32 relative to a value e^x.
__device__ __forceinline__ float exp_nonpos(float x) { return __builtin_amdgcn_exp2f(x * 1.44269504088896340736f); }

__device__ __forceinline__ double sqrt_nonneg(double x) {
  // sqrt of a finite x >= 0 (a scaled squared distance): v_rsq_f64 + two coupled Newton steps, no range scaling
  x = x < 1e-280 ? 1e-280 : x;  // sqrt(0) -> 1e-140: indistinguishable from 0 in every use, rsq stays finite; NaN stays NaN
  const double y = __builtin_amdgcn_rsq(x);
  double g = x * y, h = 0.5 * y;
  const double e = __builtin_fma(-h, g, 0.5);
  g = __builtin_fma(g, e, g);
  h = __builtin_fma(h, e, h);
  double d = __builtin_fma(-g, g, x);
  g = __builtin_fma(d, h, g);
  d = __builtin_fma(-g, g, x);
  return __builtin_fma(d, h, g);
}
__device__ __forceinline__ float sqrt_nonneg(float x) { return __builtin_amdgcn_sqrtf(x); }  // v_sqrt_f32, 1 ulp; no denormal rescue

// (sin, cos)(pi t / p) of one point, in double whatever T is: the features carry ~1e-16 (fp64) / 6e-8 (fp32) absolute
// error, uniformly over all pairs -- the direct fp32 evaluation of sin(pi (t_i - t_j) / p) is only that good near
// the diagonal
template <typename T>
__device__ __forceinline__ void phase_features(T t, double inv_p, T* s, T* c) {
  double sd, cd;
  sincospi((double)t * inv_p, &sd, &cd);
  *s = (T)sd;
  *c = (T)cd;
}

template <typename T>
struct MaternTerm {  // value = poly * exp(-q); d value / d lengthscale = dpoly * exp(-q) / l
  T q, poly, dpoly;
};
template <typename T>
__device__ __forceinline__ MaternTerm<T> matern52(T absdelta, T inv_l) {
  MaternTerm<T> m;
  m.q = T(2.23606797749978969641) * absdelta * inv_l;
  m.poly = T(1) + m.q + m.q * m.q * T(1.0 / 3.0);
  m.dpoly = (T(1) + m.q) * m.q * m.q * T(1.0 / 3.0);
  return m;
}
template <typename T>
__device__ __forceinline__ MaternTerm<T> matern32_q(T q) {
  MaternTerm<T> m;
  m.q = q;
  m.poly = T(1) + q;
  m.dpoly = q * q;
  return m;
}

// ------------------------------------------------------------------------------------------
template <typename T, int D>
struct Loadest {
  static constexpr int NX = D;      // raw coordinate columns (time first)
  static constexpr int NF = D + 2;  // per-point features: the coordinates, sin and cos of pi t / p
  static constexpr int NTHETA = 2 * D + 5;
  struct Pre {
    double inv_p_d;
    T os1, inv_lp, inv_p, inv_lm, os2, os3;
    T inv_l2[D - 1];
    T inv_l3[D];
  };
  // host side, in double: kernels receive Pre by value (uniform -> SGPRs, no per-thread divides)
  static Pre prepare(const double* th) {
    Pre p;
    p.os1 = (T)th[0];
    p.inv_lp = (T)(1.0 / th[1]);
    p.inv_p_d = 1.0 / th[2];
    p.inv_p = (T)p.inv_p_d;
    p.inv_lm = (T)(1.0 / th[3]);
    p.os2 = (T)th[4];
    for (int j = 0; j < D - 1; ++j) p.inv_l2[j] = (T)(1.0 / th[5 + j]);
    p.os3 = (T)th[4 + D];
    for (int j = 0; j < D; ++j) p.inv_l3[j] = (T)(1.0 / th[5 + D + j]);
    return p;
  }
  static __device__ __forceinline__ void features(const T (&x)[NX], const Pre& p, T (&f)[NF]) {
#pragma unroll
    for (int j = 0; j < D; ++j) f[j] = x[j];
    phase_features<T>(x[0], p.inv_p_d, &f[D], &f[D + 1]);
  }
  template <bool GRAD>
  static __device__ __forceinline__ T pair(const T (&fi)[NF], const T (&fj)[NF], const Pre& p, T w, T (&acc)[NTHETA]) {
    const T dt = fi[0] - fj[0];
    const T s = fi[D] * fj[D + 1] - fi[D + 1] * fj[D];  // sin(pi dt / p)
    const T c = fi[D + 1] * fj[D + 1] + fi[D] * fj[D];  // cos(pi dt / p)
    const T s2 = s * s;
    const MaternTerm<T> m5 = matern52(fabs(dt), p.inv_lm);
    const T e1 = exp_nonpos(T(-2) * s2 * p.inv_lp - m5.q);
    const T base1 = e1 * m5.poly;
    T sq2 = T(0), sq3, z3[D], z2[D - 1];
    z3[0] = dt * p.inv_l3[0];
    sq3 = z3[0] * z3[0];
#pragma unroll
    for (int j = 1; j < D; ++j) {
      const T d = fi[j] - fj[j];
      z2[j - 1] = d * p.inv_l2[j - 1];
      z3[j] = d * p.inv_l3[j];
      sq2 += z2[j - 1] * z2[j - 1];
      sq3 += z3[j] * z3[j];
    }
    const T base2 = exp_nonpos(T(-0.5) * sq2);
    const MaternTerm<T> m3 = matern32_q(T(1.73205080756887729353) * sqrt_nonneg(sq3));
    const T e3 = exp_nonpos(-m3.q);
    const T base3 = m3.poly * e3;
    const T k1 = p.os1 * base1, k2 = p.os2 * base2;
    if (GRAD) {
      // RAW sums: every factor that does not depend on the pair (outputscales, inverse lengthscales, 2, 4 pi ...) is
      // applied ONCE per thread by finalize() -- 18 VALU instructions per entry here instead of ~45
      const T wb1 = w * base1, we1 = w * e1, wb2 = w * base2, we3 = w * e3;
      acc[0] += wb1;
      acc[1] += wb1 * s2;
      acc[2] += wb1 * (dt * s * c);
      acc[3] += we1 * m5.dpoly;
      acc[4] += wb2;
#pragma unroll
      for (int j = 0; j < D - 1; ++j) acc[5 + j] += wb2 * (z2[j] * z2[j]);
      acc[4 + D] += w * base3;
#pragma unroll
      for (int j = 0; j < D; ++j) acc[5 + D + j] += we3 * (z3[j] * z3[j]);
    }
    return k1 + k2 + p.os3 * base3;
  }
  // the pair-independent factors of the derivative sums (see pair<true>)
  static __device__ __forceinline__ void finalize(T (&acc)[NTHETA], const Pre& p) {
    acc[1] *= p.os1 * T(2) * p.inv_lp * p.inv_lp;
    acc[2] *= p.os1 * T(4.0 * 3.14159265358979323846) * p.inv_lp * p.inv_p * p.inv_p;
    acc[3] *= p.os1 * p.inv_lm;
#pragma unroll
    for (int j = 0; j < D - 1; ++j) acc[5 + j] *= p.os2 * p.inv_l2[j];
#pragma unroll
    for (int j = 0; j < D; ++j) acc[5 + D + j] *= T(3) * p.os3 * p.inv_l3[j];
  }
};

// ------------------------------------------------------------------------------------------
template <typename T>
struct Rating {
  static constexpr int NX = 2;  // (time, stage)
  static constexpr int NF = 5;  // (time, log(stage + 1e-6), gate g(stage), sin and cos of pi t / p)
  static constexpr int NTHETA = 16;
  struct Pre {
    T b;
    T os_a[2], inv_ls_a[2], inv_lt_a[2];  // cov_shift #1, #2: Matern52(stage) x Matern32(time)
    T os_u, inv_ls_u, inv_lt_u;           // cov_bend: Matern52(stage) x Matern52(time)
    T os_b, inv_ls_b;                     // cov_base: Matern52(stage)
    T os_p, inv_lp, inv_p, inv_lm;        // cov_periodic: Periodic(time) x Matern52(time)
    double inv_p_d;
  };
  static Pre prepare(const double* th) {
    Pre p;
    p.b = (T)th[0];
    for (int a = 0; a < 2; ++a) {
      p.os_a[a] = (T)th[1 + 3 * a];
      p.inv_ls_a[a] = (T)(1.0 / th[2 + 3 * a]);
      p.inv_lt_a[a] = (T)(1.0 / th[3 + 3 * a]);
    }
    p.os_u = (T)th[7];
    p.inv_ls_u = (T)(1.0 / th[8]);
    p.inv_lt_u = (T)(1.0 / th[9]);
    p.os_b = (T)th[10];
    p.inv_ls_b = (T)(1.0 / th[11]);
    p.os_p = (T)th[12];
    p.inv_lp = (T)(1.0 / th[13]);
    p.inv_p_d = 1.0 / th[14];
    p.inv_p = (T)p.inv_p_d;
    p.inv_lm = (T)(1.0 / th[15]);
    return p;
  }
  static __device__ __forceinline__ void features(const T (&x)[NX], const Pre& p, T (&f)[NF]) {
    f[0] = x[0];
    f[1] = log(x[1] + T(1e-6));                        // LogWarpKernel, kernels.py:374-382
    f[2] = T(1) / (T(1) + exp(T(20) * (x[1] - p.b)));  // SigmoidKernel, kernels.py:311-312 (a = 20)
    phase_features<T>(x[0], p.inv_p_d, &f[3], &f[4]);
  }
  template <bool GRAD>
  static __device__ __forceinline__ T pair(const T (&fi)[NF], const T (&fj)[NF], const Pre& p, T w, T (&acc)[NTHETA]) {
    const T dt = fi[0] - fj[0], adt = fabs(dt), adw = fabs(fi[1] - fj[1]);
    const T gi = fi[2], gj = fj[2], hi = T(1) - gi, hj = T(1) - gj;
    const T gg = gi * gj, hh = hi * hj;
    T lower = T(0);
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      const MaternTerm<T> ms = matern52(adw, p.inv_ls_a[a]);
      const MaternTerm<T> mt = matern32_q(T(1.73205080756887729353) * adt * p.inv_lt_a[a]);
      const T e = exp_nonpos(-ms.q - mt.q);
      const T base = e * ms.poly * mt.poly;
      lower += p.os_a[a] * base;
      if (GRAD) {  // raw sums, finalize() applies os / inverse lengthscales
        const T wg = w * gg, wge = wg * e;
        acc[1 + 3 * a] += wg * base;
        acc[2 + 3 * a] += wge * (ms.dpoly * mt.poly);
        acc[3 + 3 * a] += wge * (ms.poly * mt.dpoly);
      }
    }
    const MaternTerm<T> us = matern52(adw, p.inv_ls_u), ut = matern52(adt, p.inv_lt_u);
    const T eu = exp_nonpos(-us.q - ut.q);
    const T baseu = eu * us.poly * ut.poly;
    const T upper = p.os_u * baseu;
    const MaternTerm<T> bs = matern52(adw, p.inv_ls_b);
    const T eb = exp_nonpos(-bs.q);
    const T baseb = eb * bs.poly;
    const T s = fi[3] * fj[4] - fi[4] * fj[3];  // sin(pi dt / p)
    const T c = fi[4] * fj[4] + fi[3] * fj[3];  // cos(pi dt / p)
    const T s2 = s * s;
    const MaternTerm<T> pm = matern52(adt, p.inv_lm);
    const T ep = exp_nonpos(T(-2) * s2 * p.inv_lp - pm.q);
    const T basep = ep * pm.poly;
    const T kp = p.os_p * basep;
    if (GRAD) {
      // d gate / d b = a g (1 - g); the inverted gate has the opposite sign
      const T gpi = T(20) * gi * hi, gpj = T(20) * gj * hj;
      acc[0] += w * (lower * (gpi * gj + gi * gpj) - upper * (gpi * hj + hi * gpj));
      const T wh = w * hh, whe = wh * eu, wbp = w * basep;
      acc[7] += wh * baseu;
      acc[8] += whe * (us.dpoly * ut.poly);
      acc[9] += whe * (us.poly * ut.dpoly);
      acc[10] += w * baseb;
      acc[11] += (w * eb) * bs.dpoly;
      acc[12] += wbp;
      acc[13] += wbp * s2;
      acc[14] += wbp * (dt * s * c);
      acc[15] += (w * ep) * pm.dpoly;
    }
    return gg * lower + hh * upper + p.os_b * baseb + kp;
  }
  static __device__ __forceinline__ void finalize(T (&acc)[NTHETA], const Pre& p) {
#pragma unroll
    for (int a = 0; a < 2; ++a) {
      acc[2 + 3 * a] *= p.os_a[a] * p.inv_ls_a[a];
      acc[3 + 3 * a] *= p.os_a[a] * p.inv_lt_a[a];
    }
    acc[8] *= p.os_u * p.inv_ls_u;
    acc[9] *= p.os_u * p.inv_lt_u;
    acc[11] *= p.os_b * p.inv_ls_b;
    acc[13] *= p.os_p * T(2) * p.inv_lp * p.inv_lp;
    acc[14] *= p.os_p * T(4.0 * 3.14159265358979323846) * p.inv_lp * p.inv_p * p.inv_p;
    acc[15] *= p.os_p * p.inv_lm;
  }
};

// ------------------------------------------------------------------------------------------
// Composite<T, D>: the GENERIC evaluator -- any sum of (optionally scaled) products of stationary factors
//     K = sum_t sigma_t^2 prod_f k_tf,   k in { RBF, Matern(nu = 1/2, 3/2, 5/2), Periodic } on a subset of the D columns,
// one shared or one per-dimension (ARD) lengthscale per factor -- e.g. the reference's unused trend term
// ScaleKernel(RBF(time)) (src/loadest_gp/models/gpytorch.py:78-88) added to its covariance, or any other model a user
// composes from the gpytorch-shaped classes of discontinuum_amd.gp.kernels.  The tree is described by a small table
// (CompositeDesc, registered once through dgp_composite_define) that the pair evaluator INTERPRETS: a slow path next to
// the two fused models (every derivative goes to a dynamically indexed accumulator, the periodic factor pays a sinpi
// per pair), but the same kernels, the same C ABI and the same parity tests.
#define DGP_C_TMAX 6   // terms
#define DGP_C_FMAX 3   // factors per term
#define DGP_C_DMAX 6   // active columns per factor
#define DGP_MODEL_COMPOSITE_BASE 16
enum { DGP_FAC_RBF = 0, DGP_FAC_MATERN = 1, DGP_FAC_PERIODIC = 2 };
struct CompositeDesc {
  unsigned char nterms, ntheta, d, pad;
  struct Fac {
    unsigned char type, nu2 /* 2 nu: 1, 3, 5 */, ndims, ard;
    signed char ls /* index of the (first) lengthscale in theta */, period /* index of the period, or -1 */;
    unsigned char dims[DGP_C_DMAX];
  };
  struct Term {
    signed char os;  // index of the outputscale in theta, or -1: unscaled
    unsigned char nfac;
    Fac fac[DGP_C_FMAX];
  } term[DGP_C_TMAX];
};
const CompositeDesc* composite_current();  // descriptor of the model being dispatched (host, thread-local; dgp_gram.hip)

template <typename T, int D>
struct Composite {
  static constexpr int NX = D, NF = D, NTHETA = DGP_MAX_THETA;
  struct Pre {
    CompositeDesc desc;
    T pv[DGP_MAX_THETA];  // outputscale -> sigma^2, lengthscale -> 1 / l, period -> 1 / p
  };
  static Pre prepare(const double* th) {
    Pre p;
    p.desc = *composite_current();
    for (int i = 0; i < DGP_MAX_THETA; ++i) p.pv[i] = T(0);
    for (int t = 0; t < p.desc.nterms; ++t) {
      const CompositeDesc::Term& tm = p.desc.term[t];
      if (tm.os >= 0) p.pv[tm.os] = (T)th[tm.os];
      for (int f = 0; f < tm.nfac; ++f) {
        const CompositeDesc::Fac& fc = tm.fac[f];
        for (int j = 0; j < (fc.ard ? fc.ndims : 1); ++j) p.pv[fc.ls + j] = (T)(1.0 / th[fc.ls + j]);
        if (fc.period >= 0) p.pv[fc.period] = (T)(1.0 / th[fc.period]);
      }
    }
    return p;
  }
  static __device__ __forceinline__ void features(const T (&x)[NX], const Pre&, T (&f)[NF]) {
#pragma unroll
    for (int j = 0; j < D; ++j) f[j] = x[j];
  }
  // value of one factor; *aux = the factor's common derivative weight: d v / d l_j = aux * z_j^2 * inv_l_j (stationary
  // factors, z_j = delta_j / l_j); for the periodic factor aux = v and the caller uses s, c
  static __device__ __forceinline__ T factor(const CompositeDesc::Fac& fc, const T (&fi)[NF], const T (&fj)[NF], const T* pv,
                                             T* aux, T* sper, T* cper, T* dper) {
    if (fc.type == DGP_FAC_PERIODIC) {
      const T dt = fi[fc.dims[0]] - fj[fc.dims[0]];
      double sd, cd;
      sincospi((double)dt * (double)pv[fc.period], &sd, &cd);
      const T s = (T)sd;
      const T v = exp_nonpos(T(-2) * s * s * pv[fc.ls]);
      *aux = v;
      *sper = s;
      *cper = (T)cd;
      *dper = dt;
      return v;
    }
    T sq = T(0);
    for (int j = 0; j < fc.ndims; ++j) {
      const T z = (fi[fc.dims[j]] - fj[fc.dims[j]]) * pv[fc.ls + (fc.ard ? j : 0)];
      sq += z * z;
    }
    if (fc.type == DGP_FAC_RBF) {
      const T v = exp_nonpos(T(-0.5) * sq);
      *aux = v;
      return v;
    }
    const T r = sqrt_nonneg(sq);
    if (fc.nu2 == 1) {  // exp(-r);  d/dl_j = exp(-r) z_j^2 / (r l_j)
      const T e = exp_nonpos(-r);
      *aux = sq > T(1e-60) ? e / r : T(0);
      return e;
    }
    if (fc.nu2 == 3) {  // (1 + sqrt3 r) exp(-sqrt3 r);  d/dl_j = 3 exp(-sqrt3 r) z_j^2 / l_j
      const T q = T(1.73205080756887729353) * r;
      const T e = exp_nonpos(-q);
      *aux = T(3) * e;
      return (T(1) + q) * e;
    }
    const T q = T(2.23606797749978969641) * r;  // (1 + q + q^2 / 3) exp(-q);  d/dl_j = 5/3 (1 + q) exp(-q) z_j^2 / l_j
    const T e = exp_nonpos(-q);
    *aux = T(5.0 / 3.0) * (T(1) + q) * e;
    return (T(1) + q + q * q * T(1.0 / 3.0)) * e;
  }
  template <bool GRAD>
  static __device__ __forceinline__ T pair(const T (&fi)[NF], const T (&fj)[NF], const Pre& p, T w, T (&acc)[NTHETA]) {
    T k = T(0);
    for (int t = 0; t < p.desc.nterms; ++t) {
      const CompositeDesc::Term& tm = p.desc.term[t];
      T v[DGP_C_FMAX], aux[DGP_C_FMAX], sp[DGP_C_FMAX], cp[DGP_C_FMAX], dp[DGP_C_FMAX];
      T prod = T(1);
      for (int f = 0; f < tm.nfac; ++f) {
        v[f] = factor(tm.fac[f], fi, fj, p.pv, &aux[f], &sp[f], &cp[f], &dp[f]);
        prod *= v[f];
      }
      const T os = tm.os >= 0 ? p.pv[tm.os] : T(1);
      k += os * prod;
      if (GRAD) {
        if (tm.os >= 0) acc[tm.os] += w * prod;
        for (int f = 0; f < tm.nfac; ++f) {
          const CompositeDesc::Fac& fc = tm.fac[f];
          T others = w * os;
          for (int g = 0; g < tm.nfac; ++g) others *= (g == f) ? T(1) : v[g];
          if (fc.type == DGP_FAC_PERIODIC) {
            const T inv_l = p.pv[fc.ls], inv_p = p.pv[fc.period];
            acc[fc.ls] += others * aux[f] * T(2) * sp[f] * sp[f] * inv_l * inv_l;
            acc[fc.period] += others * aux[f] * T(4.0 * 3.14159265358979323846) * dp[f] * sp[f] * cp[f] * inv_l * inv_p * inv_p;
          } else {
            for (int j = 0; j < fc.ndims; ++j) {
              const int q = fc.ls + (fc.ard ? j : 0);
              const T z = (fi[fc.dims[j]] - fj[fc.dims[j]]) * p.pv[q];
              acc[q] += others * aux[f] * z * z * p.pv[q];
            }
          }
        }
      }
    }
    return k;
  }
  static __device__ __forceinline__ void finalize(T (&)[NTHETA], const Pre&) {}  // the interpreter applies every factor per entry
};

// Hyperparameters of a batch, blockIdx.z selects the site: up to DGP_MAX_BATCH sites travel by value in the kernel
// argument segment (SGPR loads, nothing to upload); larger batches read them from a device array (`dev`) that the
// launcher fills with one small asynchronous copy per fit step.
template <typename M>
struct PreBatch {
  typename M::Pre p[DGP_MAX_BATCH];
  const typename M::Pre* dev;
  __device__ __forceinline__ const typename M::Pre& get() const { return dev ? dev[blockIdx.z] : p[blockIdx.z]; }
};
// bytes of device scratch a batch of B needs for its hyperparameters (0: they fit the argument segment)
#define DGP_PRE_SLOT_BYTES 512
// upload = false reuses what an earlier launcher of the same fit step put into `scratch`.  `staging` is PINNED host
// memory of at least B slots that stays untouched until the copy has run (the API layer hands out a ring slot per step,
// dgp_api.hip::PinnedRing); without it the upload is a blocking copy -- never an asynchronous copy from pageable memory
// whose lifetime ends with this call.
template <typename M>
inline PreBatch<M> prepare_batch(const double* theta, int ntheta, int B, void* scratch, bool upload, hipStream_t s,
                                 void* staging = nullptr) {
  static_assert(sizeof(typename M::Pre) <= DGP_PRE_SLOT_BYTES, "Pre does not fit its scratch slot");
  PreBatch<M> pb;
  pb.dev = nullptr;
  if (B <= DGP_MAX_BATCH || scratch == nullptr) {
    for (int b = 0; b < DGP_MAX_BATCH; ++b) pb.p[b] = M::prepare(theta + (long)(b < B ? b : 0) * ntheta);
    return pb;
  }
  for (int b = 0; b < DGP_MAX_BATCH; ++b) pb.p[b] = M::prepare(theta);
  pb.dev = (const typename M::Pre*)scratch;
  if (upload) {
    const size_t bytes = sizeof(typename M::Pre) * (size_t)B;
    if (staging) {
      typename M::Pre* host = (typename M::Pre*)staging;
      for (int b = 0; b < B; ++b) host[b] = M::prepare(theta + (long)b * ntheta);
      (void)hipMemcpyAsync(scratch, host, bytes, hipMemcpyHostToDevice, s);
    } else {
      std::vector<typename M::Pre> host((size_t)B);
      for (int b = 0; b < B; ++b) host[b] = M::prepare(theta + (long)b * ntheta);
      (void)hipStreamSynchronize(s);
      (void)hipMemcpy(scratch, host.data(), bytes, hipMemcpyHostToDevice);  // blocking: the vector dies with this call
    }
  }
  return pb;
}

}  // namespace dgp
