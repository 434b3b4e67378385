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
