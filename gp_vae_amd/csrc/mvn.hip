// mvn.hip — zero-mean multivariate-normal log-density with a squared-exponential covariance, and its
// gradient.
//
// Replaces what NUTS differentiates in GPInference.gp_regression
// (/root/reference/src/model/inference.py:400-424): k = kernel(x, x, var, ls) (jitter 1e-5 on the
// diagonal), f ~ MultivariateNormal(0, k).  For a batch of chains b (own f_b, var_b, ls_b, shared x):
//   logp  = -1/2 |L^-1 f|^2 - sum_i log L_ii - n/2 log 2 pi                    K = L L^T
//   dlogp/df   = -alpha,                      alpha = K^-1 f
//   dlogp/dvar = 1/(2 var)   sum_ij G_ij K0_ij,       G = alpha alpha^T - K^-1,  K0 = K - jitter I
//              = 1/(2 var) [ (|w|^2 - n) - jitter (|alpha|^2 - |U|_F^2) ],   w = L^-1 f, U = L^-T
//     (sum_ij G_ij K_ij = f^T alpha - tr(I); the form with K0 entries loses digits in the float32
//      cancellation between alpha_i alpha_j K0_ij and K^-1_ij K0_ij, the one with norms does not)
//   dlogp/dls  = 1/(2 ls^3)  sum_ij G_ij K0_ij r_ij^2,       r_ij = |x_i - x_j|
// The reference gets these by reverse-mode through jnp.linalg.cholesky; here they are closed form.
//
// One batched Gram build, ONE batched gpvae_potrf with n+1 extra rows [I; f^T]: the rows come back as
// U = L^-T and w^T = (L^-1 f)^T, so K^-1 = U U^T and alpha = U w need no further triangular solve.
// mvn_finish_kernel (one wave per row): alpha, -alpha -> grad_f, the three terms of logp.
// mvn_grad_kernel: 128x128 tiles of U U^T on the MFMA pipeline (PipeKC, both operands K-contiguous
// rows of U), epilogue G_ij K0_ij (1, r^2) with K0 recomputed from the points, reduced to two scalars
// per chain.  Bound: MFMA (n^3 flop per chain for U U^T, n^3 for the ride-along solve, n^3/3 factor).
#include "gemm_core.hpp"

namespace gpvae {

template <typename T> __device__ __forceinline__ T dlog(T v);
template <> __device__ __forceinline__ float dlog<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double dlog<double>(double v) { return log(v); }

struct MvnWork {
  int64_t lda, a_elems, x_elems;
};
static inline MvnWork mvn_layout(int64_t n) {
  MvnWork w;
  w.lda = round_up(n, 4);
  w.a_elems = n * w.lda;
  w.x_elems = (n + 1) * w.lda;
  return w;
}

// Sums over rows and tiles are kept as parts (one per row / per tile) and added in a fixed order by
// mvn_reduce_kernel: no floating-point atomics, so a chain's numbers are the same bits on every call and do
// not depend on which other chains share the batch.
// X_b = [I_n ; f_b^T].  grid (rows, batch)
template <typename T>
__global__ void mvn_fill_kernel(int64_t n, T *X, int64_t ldx, int64_t strideX, const T *f) {
  const int64_t b = blockIdx.y;
  const int64_t i = blockIdx.x;  // row 0..n
  T *row = X + b * strideX + i * ldx;
  for (int64_t k = threadIdx.x; k < n; k += blockDim.x)
    row[k] = (i < n) ? (k == i ? T(1) : T(0)) : f[b * n + k];
}

// one wave per row i: alpha_i = U[i,:] . w ; the row's part of logp: -(log L_ii + w_i^2 / 2) ; its part of the
// d/dvar sum: (w_i^2 - 1) - jitter (alpha_i^2 - |U[i,:]|^2).  grid (ceil(n/4), batch)
template <typename T>
__global__ __launch_bounds__(256) void mvn_finish_kernel(int64_t n, const T *A, int64_t lda,
                                                         int64_t strideA, const T *X, int64_t ldx,
                                                         int64_t strideX, const int32_t *info,
                                                         T *alpha, T *grad_f, T jitter, T *part,
                                                         int64_t pstride) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t b = blockIdx.y;
  if (i >= n) return;
  const T *U = X + b * strideX;
  const T *w = U + n * ldx;
  T dot = T(0), usq = T(0);
  for (int64_t k = i + lane; k < n; k += 64) {  // U is upper triangular
    const T u = U[i * ldx + k];
    dot += u * w[k];
    usq += u * u;
  }
  dot = wave_sum<T>(dot);
  usq = wave_sum<T>(usq);
  if (lane == 0) {
    const bool bad = info[b] != 0;
    const T a = bad ? dnan<T>() : dot;
    alpha[b * n + i] = a;
    if (grad_f) grad_f[b * n + i] = -a;
    const T lii = A[b * strideA + i * lda + i];
    T *pb = part + b * pstride;
    pb[i] = bad ? dnan<T>() : -(dlog<T>(lii) + T(0.5) * w[i] * w[i]);
    pb[n + i] = (w[i] * w[i] - T(1)) - jitter * (a * a - usq);
  }
}

// tiles of K^-1 = U U^T against K0 and K0 r^2.  grid (tiles, tiles, batch)
template <typename T, int D, bool VECM>
__global__ __launch_bounds__(NTHREADS, 2) void mvn_grad_kernel(
    int64_t n, const T *X, int64_t ldx, int64_t strideX, const T *x, const T *var_p, const T *ls_p,
    const T *alpha, T *part, int64_t pstride) {
  using M = Mfma<T>;
  constexpr int BM = 128, BK = TileK<T>::BK, F = 64 / M::TM;
  using LA = LoaderKC<T, BM, BK, VECM>;
  using Pipe = PipeKC<T, F, F, BK, BM, BM, LA, LA>;
  __shared__ __align__(16) T lds[Pipe::LDS_ELEMS];
  __shared__ T red[8];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.z;
  const T *U = X + b * strideX;
  const int64_t r0 = (int64_t)blockIdx.x * BM, q0 = (int64_t)blockIdx.y * BM;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;
  // U[i][k] = 0 for k < i: the product only needs k >= max(r0, q0), rounded down to a k-tile
  const int64_t kmin = (r0 > q0 ? r0 : q0) / BK * BK;

  LA la, lb;
  la.init([&](int s, int64_t &rg) -> const T * { rg = r0 + s; return (r0 + s < n) ? U + (r0 + s) * ldx : nullptr; }, U, tid);
  lb.init([&](int s, int64_t &rg) -> const T * { rg = q0 + s; return (q0 + s < n) ? U + (q0 + s) * ldx : nullptr; }, U, tid);
  typename M::acc_t acc[F][F];
  zero_acc<T, F, F>(acc);
  Pipe pipe;
  if (kmin < n) pipe.run(la, lb, lds, kmin, n, wm, wn, lane, tid, acc);

  const T var = var_p[b], ls = ls_p[b];
  T s2 = T(0);  // (the d/dvar sum comes from mvn_finish_kernel's norms)
#pragma unroll
  for (int fn = 0; fn < F; ++fn) {
    const int64_t j = q0 + wn + fn * M::TM + M::col(lane);
    const bool jin = j < n;
    T xj[D], aj = jin ? alpha[b * n + j] : T(0);
#pragma unroll
    for (int e = 0; e < D; ++e) xj[e] = x[(jin ? j : 0) * D + e];
#pragma unroll
    for (int fm = 0; fm < F; ++fm)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int64_t i = r0 + wm + fm * M::TM + M::row(lane, r);
        const bool in = jin && i < n;
        T d2 = T(0);
#pragma unroll
        for (int e = 0; e < D; ++e) {
          const T dl = x[(in ? i : 0) * D + e] - xj[e];
          d2 += dl * dl;
        }
        const T q = dsqrt<T>(d2) / ls;  // the Gram kernel's own arithmetic (gp.py:74-77)
        const T k0 = var * dexp<T>(T(-0.5) * (q * q));
        const T g = in ? (alpha[b * n + (in ? i : 0)] * aj - acc[fm][fn][r]) * k0 : T(0);
        s2 += g * d2;
      }
  }
  s2 = wave_sum<T>(s2);
  if (lane == 0) red[wave] = s2;
  __syncthreads();
  if (tid == 0)
    part[b * pstride + 2 * n + (int64_t)blockIdx.y * gridDim.x + blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// one workgroup per chain: logp = -n/2 log 2pi + sum of the row parts; grad_var = s1 / (2 var),
// grad_ls = s2 / (2 ls^3) from the row / tile parts (NaN when the factorisation failed).  Thread t adds parts
// t, t + 256, ... in order, then a fixed tree over the 256 threads.
template <typename T>
__device__ __forceinline__ T block_sum_ordered(const T *p, int64_t cnt, T *red, int tid) {
  T a = T(0);
  for (int64_t i = tid; i < cnt; i += 256) a += p[i];
  __syncthreads();  // red may still be read by the previous call
  red[tid] = a;
  __syncthreads();
  for (int w = 128; w > 0; w >>= 1) {
    if (tid < w) red[tid] += red[tid + w];
    __syncthreads();
  }
  return red[0];
}
template <typename T>
__global__ __launch_bounds__(256) void mvn_reduce_kernel(int64_t n, int64_t ntile2, const T *var, const T *ls,
                                                         const int32_t *info, const T *part, int64_t pstride,
                                                         T *logp, T *grad_var, T *grad_ls) {
  __shared__ T red[256];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const T *pb = part + b * pstride;
  const bool bad = info[b] != 0;
  const T lp = block_sum_ordered<T>(pb, n, red, tid);
  if (tid == 0) logp[b] = T(-0.9189385332046727) * (T)n + lp;
  if (grad_var != nullptr) {
    const T s1 = block_sum_ordered<T>(pb + n, n, red, tid);
    if (tid == 0) grad_var[b] = bad ? dnan<T>() : s1 / (T(2) * var[b]);
  }
  if (grad_ls != nullptr) {
    const T s2 = block_sum_ordered<T>(pb + 2 * n, ntile2, red, tid);
    if (tid == 0) grad_ls[b] = bad ? dnan<T>() : s2 / (T(2) * ls[b] * ls[b] * ls[b]);
  }
}


// ---- a10 as one call: potential of gp_regression / regression and its gradient ---------------------------
// q [chains, ldq] = (f [n], log var, log ls, log sigma) in NumPyro's unconstrained space
// (inference.py:368-396, 407-424): priors LogNormal(0, 0.1), InverseGamma(4, 1), HalfNormal(0.1), latent
// f ~ MVN(0, K), Gaussian likelihood on f[lik_idx] (lik_idx == NULL: every point).
// gp_prep_kernel: var = e^u, ls = e^u into the workspace, f gathered contiguous.
// gp_finish_kernel (one workgroup per chain): likelihood sums, priors, Jacobians -> U, dU/dq.
template <typename T>
__global__ void gp_prep_kernel(int64_t n, const T *q, int64_t ldq, T *f, T *var, T *ls) {
  const int64_t b = blockIdx.y;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) f[b * n + i] = q[b * ldq + i];
  if (i == 0) {
    var[b] = dexp<T>(q[b * ldq + n]);
    ls[b] = dexp<T>(q[b * ldq + n + 1]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void gp_finish_kernel(int64_t n, int64_t n_lik, const T *q, int64_t ldq,
                                                        const T *y_lik, const int64_t *lik_idx,
                                                        const T *logp, const T *gf, const T *gv,
                                                        const T *gl, T *U, T *dU, int64_t lddu) {
  __shared__ T red[4];
  const int64_t b = blockIdx.x;
  const int tid = threadIdx.x;
  const T *qb = q + b * ldq;
  T *db = dU + b * lddu;
  const T uv = qb[n], ul = qb[n + 1], us = qb[n + 2];
  const T var = dexp<T>(uv), ls = dexp<T>(ul), sig = dexp<T>(us);
  const T isig = T(1) / sig;
  for (int64_t i = tid; i < n; i += 256) db[i] = -gf[b * n + i];  // dU/df = -dlogMVN/df (+ likelihood below)
  __syncthreads();
  T sq = T(0);
  if (lik_idx == nullptr) {
    for (int64_t j = tid; j < n_lik; j += 256) {
      const T r = (y_lik[j] - qb[j]) * isig;
      sq += r * r;
      db[j] -= r * isig;
    }
  } else {
    // f[lik_idx] may name a point more than once (the reference's gather counts it once per occurrence):
    // entry i of dU belongs to thread i % 256, which adds its observations in index order — one writer per
    // entry, the same bits on every call.  An index outside [0, n) contributes nothing (memory safety only:
    // the host wrappers reject such index sets).
    for (int64_t j = 0; j < n_lik; ++j) {
      const int64_t i = lik_idx[j];
      if ((uint64_t)i >= (uint64_t)n) continue;
      const T r = (y_lik[j] - qb[i]) * isig;
      if ((j & 255) == tid) sq += r * r;
      if ((i & 255) == tid) db[i] -= r * isig;
    }
  }
  sq = wave_sum<T>(sq);
  if ((tid & 63) == 0) red[tid >> 6] = sq;
  __syncthreads();
  if (tid == 0) {
    sq = red[0] + red[1] + red[2] + red[3];
    const T s0 = T(0.1), hl2p = T(0.9189385332046727), m = (T)n_lik;
    const T lp_var = -dlog<T>(s0) - hl2p - uv * uv / (T(2) * s0 * s0);                     // LogNormal(0,.1) + Jacobian
    const T lp_sig = T(0.6931471805599453) - hl2p - dlog<T>(s0) - T(0.5) * (sig / s0) * (sig / s0) + us;  // HalfNormal(.1)
    const T lp_ls = T(-1.791759469228055) - T(4) * ul - T(1) / ls;                          // InverseGamma(4,1): -lgamma(4)
    const T lp_lik = T(-0.5) * sq - m * us - m * hl2p;
    U[b] = -(lp_var + lp_sig + lp_ls + logp[b] + lp_lik);
    db[n] = -(-uv / (s0 * s0) + gv[b] * var);
    db[n + 1] = -(T(-4) + T(1) / ls + gl[b] * ls);
    db[n + 2] = -(T(1) - (sig / s0) * (sig / s0) + sq - m);
  }
}

template <typename T>
static int run_mvn(hipStream_t st, int dtype, int d, int64_t n, int batch, const void *x,
                   const void *f, const void *var, const void *ls, double jitter, void *logp,
                   void *grad_f, void *grad_var, void *grad_ls, int32_t *info, void *work) {
  const MvnWork w = mvn_layout(n);
  T *A = (T *)work;                         // [batch][n][lda]
  T *X = A + (int64_t)batch * w.a_elems;    // [batch][n + 1][lda]
  T *alpha = X + (int64_t)batch * w.x_elems;  // [batch][n]
  const int64_t nt = cdiv(n, 128), pstride = 2 * n + nt * nt;
  T *part = alpha + (int64_t)batch * n;       // [batch][n row parts of logp | n of d/dvar | nt^2 tile parts of d/dls]
  char *pw = (char *)(part + pstride * (int64_t)batch);
  pw = (char *)(((uintptr_t)pw + 15) & ~(uintptr_t)15);
  int rc = gpvae_gram_se((void *)st, dtype, d, n, n, x, x, batch, var, ls, jitter, A, w.lda, w.a_elems);
  if (rc) return rc;
  hipLaunchKernelGGL((mvn_fill_kernel<T>), dim3((unsigned)(n + 1), (unsigned)batch), dim3(256), 0, st,
                     n, X, w.lda, w.x_elems, (const T *)f);
  rc = gpvae_potrf((void *)st, dtype, n, batch, A, w.lda, w.a_elems, n + 1, X, w.lda, w.x_elems, info,
                   0, pw);
  if (rc) return rc;
  hipLaunchKernelGGL((mvn_finish_kernel<T>), dim3((unsigned)cdiv(n, 4), (unsigned)batch), dim3(256),
                     0, st, n, A, w.lda, w.a_elems, X, w.lda, w.x_elems, info, alpha, (T *)grad_f,
                     (T)jitter, part, pstride);
  if (grad_ls) {
    const dim3 grid((unsigned)cdiv(n, 128), (unsigned)cdiv(n, 128), (unsigned)batch);
    const bool vec = kc_vec_ok<T>(X, w.lda) && (w.x_elems % VecT<T>::N) == 0;
    auto launch = [&](auto dd, auto vv) {
      hipLaunchKernelGGL((mvn_grad_kernel<T, decltype(dd)::value, decltype(vv)::value>), grid,
                         dim3(NTHREADS), 0, st, n, X, w.lda, w.x_elems, (const T *)x, (const T *)var,
                         (const T *)ls, alpha, part, pstride);
    };
    if (d == 1) {
      if (vec) launch(std::integral_constant<int, 1>{}, std::true_type{});
      else launch(std::integral_constant<int, 1>{}, std::false_type{});
    } else {
      if (vec) launch(std::integral_constant<int, 2>{}, std::true_type{});
      else launch(std::integral_constant<int, 2>{}, std::false_type{});
    }
  }
  hipLaunchKernelGGL((mvn_reduce_kernel<T>), dim3((unsigned)batch), dim3(256), 0, st, n, nt * nt, (const T *)var,
                     (const T *)ls, info, (const T *)part, pstride, (T *)logp, (T *)grad_var, (T *)grad_ls);
  GPVAE_CHECK_LAUNCH("gpvae_mvn_logp");
  return 0;
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int64_t gpvae_mvn_workspace_bytes(int dtype, int64_t n, int batch) {
  if (n <= 0 || batch <= 0) return 0;
  const MvnWork w = mvn_layout(n);
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  const int64_t nt = cdiv(n, 128);
  return (int64_t)batch * (w.a_elems + w.x_elems + 3 * n + nt * nt) * es + 256 +
         gpvae_potrf_workspace_bytes(dtype, n, n + 1, batch);
}

extern "C" int gpvae_mvn_logp(void *stream, int dtype, int d, int64_t n, int batch, const void *x,
                              const void *f, const void *var, const void *ls, double jitter,
                              void *logp, void *grad_f, void *grad_var, void *grad_ls,
                              int32_t *info, void *work) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_mvn_logp: bad dtype %d", dtype);
  GPVAE_REQUIRE(d == 1 || d == 2, 3, "gpvae_mvn_logp: d must be 1 or 2");
  GPVAE_REQUIRE(n >= 1, 4, "gpvae_mvn_logp: n < 1");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 5, "gpvae_mvn_logp: batch out of range [0,65535]");
  if (batch == 0) return 0;
  GPVAE_REQUIRE(x && f && var && ls, 6, "gpvae_mvn_logp: NULL input");
  GPVAE_REQUIRE(logp, 11, "gpvae_mvn_logp: logp is NULL");
  GPVAE_REQUIRE(info, 15, "gpvae_mvn_logp: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 16, "gpvae_mvn_logp: work NULL or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32 ? run_mvn<float>(st, dtype, d, n, batch, x, f, var, ls, jitter, logp,
                                             grad_f, grad_var, grad_ls, info, work)
                            : run_mvn<double>(st, dtype, d, n, batch, x, f, var, ls, jitter, logp,
                                              grad_f, grad_var, grad_ls, info, work);
}

extern "C" int64_t gpvae_gp_potential_workspace_bytes(int dtype, int64_t n, int chains) {
  if (n <= 0 || chains <= 0) return 0;
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  // f [chains, n], var, ls, logp, grad_var, grad_ls [chains], grad_f [chains, n], then gpvae_mvn_logp's workspace
  return round_up((int64_t)chains * (2 * n + 5) * es, 256) + gpvae_mvn_workspace_bytes(dtype, n, chains);
}

template <typename T>
static int run_gp_potential(hipStream_t st, int dtype, int d, int64_t n, int chains, const void *x, const void *q,
                            int64_t ldq, const void *y_lik, const int64_t *lik_idx, int64_t n_lik, double jitter,
                            void *U, void *dU, int64_t lddu, int32_t *info, void *work) {
  T *f = (T *)work, *var = f + (int64_t)chains * n, *ls = var + chains, *lp = ls + chains, *gv = lp + chains,
    *gl = gv + chains, *gf = gl + chains;
  void *mwork = (char *)work + round_up((int64_t)chains * (2 * n + 5) * (int64_t)sizeof(T), 256);
  hipLaunchKernelGGL((gp_prep_kernel<T>), dim3((unsigned)cdiv(n, 256), (unsigned)chains), dim3(256), 0, st, n,
                     (const T *)q, ldq, f, var, ls);
  if (int rc = run_mvn<T>(st, dtype, d, n, chains, x, f, var, ls, jitter, lp, gf, gv, gl, info, mwork)) return rc;
  hipLaunchKernelGGL((gp_finish_kernel<T>), dim3((unsigned)chains), dim3(256), 0, st, n, n_lik, (const T *)q, ldq,
                     (const T *)y_lik, lik_idx, lp, gf, gv, gl, (T *)U, (T *)dU, lddu);
  GPVAE_CHECK_LAUNCH("gpvae_gp_potential");
  return 0;
}

extern "C" int gpvae_gp_potential(void *stream, int dtype, int d, int64_t n, int chains, const void *x,
                                  const void *q, int64_t ldq, const void *y_lik, const int64_t *lik_idx,
                                  int64_t n_lik, double jitter, void *U, void *dU, int64_t lddu,
                                  int32_t *info, void *work) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_gp_potential: bad dtype %d", dtype);
  GPVAE_REQUIRE(d == 1 || d == 2, 3, "gpvae_gp_potential: d must be 1 or 2");
  GPVAE_REQUIRE(n >= 1, 4, "gpvae_gp_potential: n < 1");
  GPVAE_REQUIRE(chains >= 0 && chains <= 65535, 5, "gpvae_gp_potential: chains out of range [0,65535]");
  if (chains == 0) return 0;
  GPVAE_REQUIRE(x && q && y_lik, 6, "gpvae_gp_potential: NULL input");
  GPVAE_REQUIRE(ldq >= n + 3, 8, "gpvae_gp_potential: ldq < n + 3");
  GPVAE_REQUIRE(n_lik >= 0 && n_lik <= n && (lik_idx != nullptr || n_lik == n), 11,
                "gpvae_gp_potential: n_lik out of range, or lik_idx NULL with n_lik != n");
  GPVAE_REQUIRE(U && dU, 13, "gpvae_gp_potential: U / dU is NULL");
  GPVAE_REQUIRE(lddu >= n + 3, 15, "gpvae_gp_potential: lddu < n + 3");
  GPVAE_REQUIRE(info, 16, "gpvae_gp_potential: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 17, "gpvae_gp_potential: work NULL or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? run_gp_potential<float>(st, dtype, d, n, chains, x, q, ldq, y_lik, lik_idx, n_lik, jitter, U, dU, lddu, info, work)
             : run_gp_potential<double>(st, dtype, d, n, chains, x, q, ldq, y_lik, lik_idx, n_lik, jitter, U, dU, lddu, info, work);
}
