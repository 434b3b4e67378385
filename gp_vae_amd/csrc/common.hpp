// common.hpp — shared host/device helpers for libgpvae_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "../../include/gpvae.h"

namespace gpvae {

// ---- host-side error plumbing -------------------------------------------------------------------
void set_error(const char *fmt, ...);  // defined in api.hip (thread-local buffer)

#define GPVAE_REQUIRE(cond, argidx, ...)     \
  do {                                       \
    if (!(cond)) {                           \
      ::gpvae::set_error(__VA_ARGS__);       \
      return -(argidx);                      \
    }                                        \
  } while (0)

// A launch error (bad configuration) is reported as argument 1 = the stream.
#define GPVAE_CHECK_LAUNCH(name)                                                     \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      ::gpvae::set_error("%s: launch failed: %s", name, hipGetErrorString(e__));     \
      return -1;                                                                     \
    }                                                                                \
  } while (0)

static inline int64_t cdiv(int64_t a, int64_t b) { return (a + b - 1) / b; }
static inline int64_t round_up(int64_t a, int64_t b) { return cdiv(a, b) * b; }
static inline bool aligned16(const void *p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

// ---- device-side helpers --------------------------------------------------------------------------
template <typename T> struct Vec4;
template <> struct Vec4<float> { using type = float4; };
template <> struct Vec4<double> { using type = double4; };

template <typename T> __device__ __forceinline__ T dexp(T v);
template <> __device__ __forceinline__ float dexp<float>(float v) { return expf(v); }
template <> __device__ __forceinline__ double dexp<double>(double v) { return exp(v); }
template <typename T> __device__ __forceinline__ T dsqrt(T v);
template <> __device__ __forceinline__ float dsqrt<float>(float v) { return sqrtf(v); }
template <> __device__ __forceinline__ double dsqrt<double>(double v) { return sqrt(v); }

template <typename T> __device__ __forceinline__ T dnan();
template <> __device__ __forceinline__ float dnan<float>() { return __builtin_nanf(""); }
template <> __device__ __forceinline__ double dnan<double>() { return __builtin_nan(""); }

// Wave-wide sum (64 lanes), result in every lane.
template <typename T> __device__ __forceinline__ T wave_sum(T v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

// ---- MFMA traits ---------------------------------------------------------------------------------
// One MFMA computes a TM x TM tile and consumes KI consecutive k per instruction; lane l supplies
// A[l % TM][l / TM] and B[l / TM][l % TM] (cdna_hip_programming.md §3).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

template <typename T> struct Mfma;
template <> struct Mfma<float> {
  static constexpr int TM = 32, KI = 2, NACC = 16;
  using acc_t = f32x16;
  __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0);
  }
  // accumulator register r of lane l holds C[row][col]
  __device__ static __forceinline__ int row(int lane, int r) {
    return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
  }
  __device__ static __forceinline__ int col(int lane) { return lane & 31; }
};
template <> struct Mfma<double> {
  static constexpr int TM = 16, KI = 4, NACC = 4;
  using acc_t = f64x4;
  __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
  __device__ static __forceinline__ int col(int lane) { return lane & 15; }
};

// 16x16x4 tiles for the small blocked updates inside the Cholesky panel (both types): lane l
// supplies A[l % 16][l / 16] and B[l / 16][l % 16]; accumulator register r of lane l holds
// C[row(l, r)][l % 16].
typedef float f32x4v __attribute__((ext_vector_type(4)));
template <typename T> struct Mfma16;
template <> struct Mfma16<float> {
  using acc_t = f32x4v;
  __device__ static __forceinline__ acc_t mma(float a, float b, acc_t c) {
    return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int r) { return 4 * (lane >> 4) + r; }
};
template <> struct Mfma16<double> {
  using acc_t = f64x4;
  __device__ static __forceinline__ acc_t mma(double a, double b, acc_t c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
  }
  __device__ static __forceinline__ int row(int lane, int r) { return (lane >> 4) + 4 * r; }
};

// exp(-0.5 (sqrt(s) / ls)^2), the reference's order of operations (norm, divide, square, exp).
// f32: with libm's sqrtf / IEEE division / expf the kernel is bound by the vector ALU at 2.6 TB/s
// of stores (about 45 instructions per entry), not by HBM.  Here: v_sqrt_f32 (1 ulp), a multiply
// by 1/ls, and exp as v_exp_f32 of x log2(e) with the product's rounding error carried along
// (2^(hi+lo) = 2^hi (1 + lo ln 2)): 3 ulp on the entry, about 14 instructions.
template <typename T> struct GramMath {
  __device__ static __forceinline__ T expq(T s, T inv_ls) {
    const T q = dsqrt<T>(s) * inv_ls;
    return dexp<T>(T(-0.5) * (q * q));
  }
};
// f64: libm's sqrt and exp made the builder ALU-bound at 0.39 of the HBM rate (about 105 double-rate instructions
// per entry).  Here: (sqrt(s) / ls)^2 as s (1 / ls)^2 — the argument moves by 3e-16 relative, the entry by
// |x| 3e-16 — and exp(x) = 2^k e^r with k = rint(x log2 e), r = x - k ln 2 in two pieces (|r| <= 0.347) and a
// degree-13 Taylor polynomial (remainder 4e-18): about 25 instructions, 1 ulp.
template <> struct GramMath<double> {
  __device__ static __forceinline__ double expq(double s, double inv_ls) {
    const double x = -0.5 * (s * (inv_ls * inv_ls));
    if (x < -745.2) return 0.0;  // below the smallest subnormal (also s = inf)
    const double k = __builtin_rint(x * 1.4426950408889634074);
    double r = __builtin_fma(k, -6.93147180369123816490e-01, x);
    r = __builtin_fma(k, -1.90821492927058770002e-10, r);
    double p = 1.6059043836821613e-10;  // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);    // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);   // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);   // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);  // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);    // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);   // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);   // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);   // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);  // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);  // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_ldexp(p, (int)k);  // NaN in: NaN out
  }
};
template <> struct GramMath<float> {
  __device__ static __forceinline__ float expq(float s, float inv_ls) {
    const float q = __builtin_amdgcn_sqrtf(s) * inv_ls;
    const float x = -0.5f * (q * q);
    const float L2E = 1.44269502162933349609375f, L2E_LO = 1.92596299112661746e-8f;
    const float hi = x * L2E;
    const float lo = fmaf(x, L2E, -hi) + x * L2E_LO;
    const float e = __builtin_amdgcn_exp2f(hi);
    return fmaf(e, lo * 0.693147182464599609375f, e);
  }
};

// 1/sqrt(x): v_rsq_f32 is accurate to 1 ulp; v_rsq_f64 is a ~26-bit seed refined by two Newton
// steps.  Negative or NaN input gives NaN, zero gives +inf — both poison the column, as wanted.
template <typename T> __device__ __forceinline__ T fast_rsqrt(T x);
template <> __device__ __forceinline__ float fast_rsqrt<float>(float x) {
  return __builtin_amdgcn_rsqf(x);
}
template <> __device__ __forceinline__ double fast_rsqrt<double>(double x) {
  double r = __builtin_amdgcn_rsq(x);
#pragma unroll
  for (int it = 0; it < 2; ++it) {
    const double e = fma(-x * r, r, 1.0);
    r = fma(0.5 * r, e, r);
  }
  return r;
}

// Pivot scaling with LAPACK-grade rounding.  Every entry of a column of L is scaled by the same
// 1/L_jj, so an error in it is SYSTEMATIC: on a nearly rank-one matrix (long lengthscale,
// K ~ var 11^T + jitter I) a 1-ulp bias in 1/sqrt(piv) perturbs the Schur complement by
// 2 ulp * var * n in norm — at n = 300, jitter 1e-4, f32 that ate the whole margin and reported
// "not positive definite" where LAPACK's spotrf succeeds.  Hence: L_jj = correctly rounded
// sqrt(piv) (Newton on the v_rsq seed, Markstein correction), rd = its refined reciprocal, and
// entries = correctly rounded s / L_jj (one fma correction of s * rd).
// piv <= 0, NaN or 0 still poison the column (NaN / inf), as wanted.
template <typename T> __device__ __forceinline__ void pivot_scale(T piv, T &ljj, T &rd) {
  T y = fast_rsqrt<T>(piv);
  const T e = fma(-(piv * y), y, T(1));
  y = fma(T(0.5) * y, e, y);                 // 1/sqrt(piv), refined
  const T l0 = piv * y;
  ljj = fma(fma(-l0, l0, piv), T(0.5) * y, l0);  // sqrt(piv), correctly rounded
  rd = fma(fma(-ljj, y, T(1)), y, y);        // 1/ljj
}
template <typename T> __device__ __forceinline__ T div_scaled(T s, T ljj, T rd) {
  const T q = s * rd;
  return fma(fma(-q, ljj, s), rd, q);  // s / ljj, correctly rounded (one correction step)
}

// value of lane j's v in every lane (v_readlane)
template <typename T> __device__ __forceinline__ T lane_bcast(T v, int j);
template <> __device__ __forceinline__ float lane_bcast<float>(float v, int j) {
  return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), j));
}
template <> __device__ __forceinline__ double lane_bcast<double>(double v, int j) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), j);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), j);
  return __hiloint2double(hi, lo);
}

// The scalar part of VAEInference.regression's potential (src/model/inference.py:243-254 in NumPyro's unconstrained
// space), shared by vae_potential_kernel (vae.hip) and the fused NUTS iteration (nuts.hip) so that both evaluate it
// with the very same operations: sq = sum of squared residuals at the observed columns, zz = |z|^2, u = log noise.
// Returns U; du = dU/du; iv = 1 / sigma^2 (the scale of the residuals' gradient).
template <typename T>
__device__ __forceinline__ T vae_potential_scalar(T sq, T zz, T u, T s0p, int n_obs, int z, T &du, T &iv) {
#pragma clang fp contract(off)  // no fused multiply-adds here: which products fuse depends on the inlining context,
                                // and the two callers must agree to the last bit (chains are compared bit for bit)
  const T sigma = dexp<T>(u);
  iv = T(1) / (sigma * sigma);
  const T hl2p = T(0.9189385332046727), rs = sigma / s0p;
  const T log_lik = T(-0.5) * sq * iv - (T)n_obs * (u + hl2p);
  const T log_pz = T(-0.5) * zz - (T)z * hl2p;
  const T log_ps = T(0.6931471805599453) - hl2p - (T)log((double)s0p) - T(0.5) * rs * rs + u;
  du = rs * rs - T(1) + (T)n_obs - sq * iv;
  return -(log_lik + log_pz + log_ps);
}

}  // namespace gpvae
