// krig.hip — kriging prediction.
//
// Replaces GPInference.gp_prediction (/root/reference/src/model/inference.py:427-460), keeping its
// literal quirks: A = k_xx(+jitter I) + sigma I (sigma, not sigma^2: inference.py:450); jitter also
// lands on k_px when n_test == n_obs (the kernel's "square" rule, gp.py:79-80); diag(k_pp) =
// var + jitter; variance clipped at 0 (inference.py:452).
//
// The reference forms inv(A) by LU and three dense products, one of them n_test x n_test of which
// only the diagonal is used.  Here:  A = L L^T;  [k_px; y^T] L^-T comes out of the SAME panel sweep
// as trailing rows of gpvae_potrf (no separate triangular solve);  with X = k_px L^-T, w = L^-1 y:
//   mean_p = X[p,:] . w        v_p = (var + jitter) - |X[p,:]|^2
// one streaming pass over X (krig_finish_kernel, HBM-bound).  Identical in exact arithmetic.
#include "common.hpp"

namespace gpvae {

// blockIdx.y = problem.  A_b += sigma_b I  (k_xx + eye * sigma, inference.py:450) and the shared
// observations copied behind problem b's k_px rows (the right-hand side that rides along)
template <typename T>
__global__ void add_diag_kernel(T *A, int64_t lda, int64_t strideA, int64_t n, const T *sigma,
                                T *yrow, int64_t strideX, const T *y_obs) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t b = blockIdx.y;
  if (i < n) {
    A[b * strideA + i * lda + i] += sigma[b];
    yrow[b * strideX + i] = y_obs[i];
  }
}

// one wave per test point: dot(X_p, w) and |X_p|^2 over n_obs
template <typename T>
__global__ __launch_bounds__(256) void krig_finish_kernel(int64_t n_obs, int64_t n_test,
                                                          const T *X, int64_t ldx, int64_t strideX,
                                                          const T *var, T jitter, const T *xi,
                                                          T *mean, T *sample, T *var_out) {
  const int lane = threadIdx.x & 63;
  const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (p >= n_test) return;
  const int64_t b = blockIdx.y;  // problem: outputs are [batch][n_test], var/xi per problem
  X += b * strideX;
  var += b;
  if (xi) xi += b * n_test;
  if (mean) mean += b * n_test;
  if (sample) sample += b * n_test;
  if (var_out) var_out += b * n_test;
  const T *row = X + p * ldx;
  const T *w = X + n_test * ldx;
  T dot = T(0), sq = T(0);
  for (int64_t k = lane; k < n_obs; k += 64) {
    const T v = row[k];
    dot += v * w[k];
    sq += v * v;
  }
  dot = wave_sum<T>(dot);
  sq = wave_sum<T>(sq);
  if (lane == 0) {
    const T v = (var[0] + jitter) - sq;  // diag(k_pp - k_px A^-1 k_px^T)
    const T vc = v > T(0) ? v : T(0);    // jnp.clip(..., a_min=0.0)   (NaN stays NaN below)
    if (mean) mean[p] = dot;
    if (var_out) var_out[p] = (v != v) ? v : vc;
    if (sample) sample[p] = xi ? dot + dsqrt<T>((v != v) ? v : vc) * xi[p] : dot;
  }
}

struct KrigWork {
  int64_t lda, a_elems, x_elems;
};
static inline KrigWork krig_layout(int64_t n_obs, int64_t n_test) {
  KrigWork w;
  w.lda = round_up(n_obs, 4);
  w.a_elems = n_obs * w.lda;
  w.x_elems = (n_test + 1) * w.lda;
  return w;
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int64_t gpvae_krig_batched_workspace_bytes(int dtype, int64_t n_obs, int64_t n_test,
                                                      int batch) {
  const KrigWork w = krig_layout(n_obs, n_test);
  if (batch < 1) batch = 1;
  return (int64_t)batch * (w.a_elems + w.x_elems) * (dtype == GPVAE_F64 ? 8 : 4) + 256 +
         gpvae_potrf_workspace_bytes(dtype, n_obs, n_test + 1, batch);
}

extern "C" int64_t gpvae_krig_workspace_bytes(int dtype, int64_t n_obs, int64_t n_test) {
  return gpvae_krig_batched_workspace_bytes(dtype, n_obs, n_test, 1);
}

extern "C" int gpvae_krig_predict_batched(void *stream, int dtype, int d, int64_t n_obs,
                                          int64_t n_test, int batch, const void *x_obs,
                                          const void *x_test, const void *y_obs, const void *var,
                                          const void *ls, const void *sigma, double jitter,
                                          const void *xi, void *mean, void *sample, void *var_out,
                                          int32_t *info, void *work) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_krig_predict: bad dtype");
  GPVAE_REQUIRE(d == 1 || d == 2, 3, "gpvae_krig_predict: d must be 1 or 2");
  GPVAE_REQUIRE(n_obs >= 1, 4, "gpvae_krig_predict: n_obs < 1");
  GPVAE_REQUIRE(n_test >= 0, 5, "gpvae_krig_predict: n_test < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 6, "gpvae_krig_predict: batch out of range [0,65535]");
  if (batch == 0) return 0;
  GPVAE_REQUIRE(x_obs && y_obs && var && ls && sigma, 7, "gpvae_krig_predict: NULL input");
  GPVAE_REQUIRE(info, 18, "gpvae_krig_predict: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 19, "gpvae_krig_predict: work NULL or not 16-byte aligned");
  if (n_test > 0) GPVAE_REQUIRE(x_test, 8, "gpvae_krig_predict: x_test is NULL");
  hipStream_t st = (hipStream_t)stream;
  const size_t es = dtype == GPVAE_F64 ? 8 : 4;
  const KrigWork w = krig_layout(n_obs, n_test);
  char *A = (char *)work;                                // [batch][n_obs][lda]
  char *X = A + (size_t)batch * w.a_elems * es;          // [batch][n_test + 1][lda]
  int rc = gpvae_gram_se(stream, dtype, d, n_obs, n_obs, x_obs, x_obs, batch, var, ls, jitter, A,
                         w.lda, w.a_elems);
  if (rc) return rc;
  if (n_test > 0) {
    rc = gpvae_gram_se(stream, dtype, d, n_test, n_obs, x_test, x_obs, batch, var, ls, jitter, X,
                       w.lda, w.x_elems);
    if (rc) return rc;
  }
  const dim3 dgrid((unsigned)cdiv(n_obs, 256), (unsigned)batch);
  char *yrow = X + (size_t)n_test * w.lda * es;
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((add_diag_kernel<float>), dgrid, dim3(256), 0, st, (float *)A, w.lda,
                       w.a_elems, n_obs, (const float *)sigma, (float *)yrow, w.x_elems,
                       (const float *)y_obs);
  else
    hipLaunchKernelGGL((add_diag_kernel<double>), dgrid, dim3(256), 0, st, (double *)A, w.lda,
                       w.a_elems, n_obs, (const double *)sigma, (double *)yrow, w.x_elems,
                       (const double *)y_obs);
  char *pw = X + (size_t)batch * w.x_elems * es;
  pw = (char *)(((uintptr_t)pw + 15) & ~(uintptr_t)15);
  rc = gpvae_potrf(stream, dtype, n_obs, batch, A, w.lda, w.a_elems, n_test + 1, X, w.lda,
                   w.x_elems, info, 0, pw);
  if (rc) return rc;
  if (n_test > 0) {
    const dim3 grid((unsigned)cdiv(n_test, 4), (unsigned)batch);
    if (dtype == GPVAE_F32)
      hipLaunchKernelGGL((krig_finish_kernel<float>), grid, dim3(256), 0, st, n_obs, n_test,
                         (const float *)X, w.lda, w.x_elems, (const float *)var, (float)jitter,
                         (const float *)xi, (float *)mean, (float *)sample, (float *)var_out);
    else
      hipLaunchKernelGGL((krig_finish_kernel<double>), grid, dim3(256), 0, st, n_obs, n_test,
                         (const double *)X, w.lda, w.x_elems, (const double *)var, jitter,
                         (const double *)xi, (double *)mean, (double *)sample, (double *)var_out);
  }
  GPVAE_CHECK_LAUNCH("gpvae_krig_predict");
  return 0;
}

extern "C" int gpvae_krig_predict(void *stream, int dtype, int d, int64_t n_obs, int64_t n_test,
                                  const void *x_obs, const void *x_test, const void *y_obs,
                                  const void *var, const void *ls, const void *sigma, double jitter,
                                  const void *xi, void *mean, void *sample, void *var_out,
                                  int32_t *info, void *work) {
  const int rc = gpvae_krig_predict_batched(stream, dtype, d, n_obs, n_test, 1, x_obs, x_test, y_obs,
                                            var, ls, sigma, jitter, xi, mean, sample, var_out, info,
                                            work);
  // argument positions after `batch` are one lower in this entry point
  return rc <= -7 ? rc + 1 : rc;
}
