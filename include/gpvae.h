/*
 * gpvae.h — C ABI of libgpvae_hip.so: the MI355X (gfx950) hot path of edenx/gp-vae.
 *
 * The reference (/root/reference, pure JAX/NumPyro) has no FFI of its own; its seams are Python
 * callables.  Each entry point below names the reference call site whose arithmetic
 * it replaces (file:line under /root/reference).  INTEGRATION.md shows the ctypes / JAX-FFI binding
 * a maintainer would add on the reference side.
 *
 * Conventions (all entry points):
 *   - plain C, no torch / no C++ types; every pointer except `stream` is a DEVICE pointer owned by
 *     the caller; matrices are row-major with an explicit leading dimension in ELEMENTS; a matrix
 *     with `rows` rows and leading dimension `ld` must be backed by rows*ld readable elements (the
 *     16-byte loads of the last row may touch its padding), and a dense `ld == cols` matrix may be
 *     read up to one 16-byte vector past a row's end INTO THE NEXT ROW of the same array — never past
 *     the array (loads past it are suppressed by the buffer range check);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); calls are asynchronous and
 *     stream-ordered, never synchronise the device and never allocate device memory
 *     (workspaces are caller-provided; sizes come from the *_workspace_bytes queries);
 *   - `dtype` selects the arithmetic type: GPVAE_F32 (the reference's type) or GPVAE_F64;
 *   - return value: 0 = launched; <0 = -(1-based index of the first bad argument); numerical
 *     failures are reported through device-side `info` words (LAPACK style), never through the
 *     return value, because the call does not wait for the device;
 *   - gpvae_last_error() returns a thread-local, human-readable reason for the last non-zero return.
 */
#ifndef GPVAE_H
#define GPVAE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: workspace contents changed (round 2): gpvae_potrf_workspace_bytes also covers the bf16 planes of the
 * current 256-column panels (float32, n % 256 == 0, n >= 512, m_extra == 0: trailing updates on the bf16 matrix
 * pipe; a group of up to four panels since round 3 — the query says how much), decoder / mvn workspaces hold per-slice slabs instead of atomically accumulated buffers.  A caller must
 * size every workspace with the *_workspace_bytes query of the library it runs against. */
#define GPVAE_ABI_VERSION 2

enum { GPVAE_F32 = 0, GPVAE_F64 = 1 };

/* gpvae_potrf flags */
enum {
  GPVAE_POTRF_ZERO_UPPER = 1, /* clear the strict upper triangle (JAX returns a clean lower factor) */
  GPVAE_POTRF_NAN_ON_FAIL = 2, /* a non-PD matrix comes back all-NaN (jnp.linalg.cholesky semantics) */
  /* A hint, not a change of result: the call runs beside other work on the GPU (the training loop's factor stream
   * beside draws and ELBO).  The bf16-plane route then holds fewer CUs: per 256-column outer block the four panel
   * launches cover the block's own rows only (at most two workgroups per matrix) and ONE launch takes all rows
   * below it through the four 64-column steps (potrf.hip potrf_rows_kernel) — about half the chip time the panel
   * workgroups' LDS keeps from a co-running kernel, for a dependent chain that is ~8 % longer when the call runs
   * alone (16 x 4096: 4.1 -> 4.45 ms alone; the training step 0.780 -> 0.771 ms).  No effect on other routes. */
  GPVAE_POTRF_SHARE_CHIP = 4
};

int gpvae_abi_version(void);
const char *gpvae_last_error(void);

/* ---- a1: squared-exponential Gram matrix --------------------------------------------------------
 * Replaces exp_kernel2 (src/model/gp.py:44-81) incl. check_d's 1-D reshape (gp.py:30-41):
 *   K[b][i][j] = var[b] * exp(-0.5 * (||x_i - z_j||_2 / ls[b])^2)  (+ jitter on i==j iff n_x == n_z,
 *   the reference's "square => add jitter*I" rule, gp.py:79-80).
 * x: [n_x, d], z: [n_z, d] (d in {1,2}; contiguous), var, ls: device arrays of length `batch`.
 * K: [batch][n_x][ldk]; matrix b starts at K + b*strideK elements.
 * Bound: HBM write bandwidth; algorithmic bytes = batch*n_x*n_z*sizeof(T).                        */
int gpvae_gram_se(void *stream, int dtype, int d, int64_t n_x, int64_t n_z, const void *x,
                  const void *z, int batch, const void *var, const void *ls, double jitter,
                  void *K, int64_t ldk, int64_t strideK);

/* ---- a2: aggregated (cell-averaged) kernel -------------------------------------------------------
 * Replaces agg_kernel_grid (src/model/gp.py:85-127) with the Uniform draws made an input:
 *   K*[i][j] = (1/m^2) sum_{a,b} var*exp(-0.5*(||p_ia - p_jb||/ls)^2) + jitter/m
 * pts: [n_cells, m, d] sample locations (origin_i + U/n, gp.py:103-113).  The "+jitter/m on every
 * entry" term is what the literal code computes (every m x m block is square);
 * pass jitter = 0 to get the intended average.  var, ls: device scalars.                          */
int gpvae_gram_agg(void *stream, int dtype, int d, int64_t n_cells, int m, const void *pts,
                   const void *var, const void *ls, double jitter, void *K, int64_t ldk);

/* ---- a3: batched blocked Cholesky (+ trailing rows) ----------------------------------------------
 * Replaces the Cholesky inside dist.MultivariateNormal(covariance_matrix=k) (gp.py:190-193,
 * inference.py:378-383, 417-422) and the jnp.linalg.inv of gp_prediction (inference.py:450).
 * In place: the lower triangle of A[b] ([n, lda]) is overwritten by L with A = L L^T.
 * If m_extra > 0, X[b] ([m_extra, ldx]) is overwritten by X L^-T (the right-side triangular solve
 * the kriging path needs: rows of k_px and y^T) — the same panel sweep, no second pass over L.
 * The strict upper triangle is not referenced, except that it is cleared inside the 64x64
 * diagonal blocks (and everywhere with GPVAE_POTRF_ZERO_UPPER).
 * info[b] (int32): 0 = ok, j>0 = leading minor j not positive definite (LAPACK potrf).
 * work: gpvae_potrf_workspace_bytes() bytes, 16-byte aligned.  Contents: the factored diagonal blocks,
 * parked there until the last launch (late workgroups of a panel launch never see a half-updated A),
 * and — float32 with n % 256 == 0, n >= 512 and m_extra == 0 — behind them batch * 3 * n * 256 g bf16
 * values: the current group of g 256-column panels as three bf16 planes (exact 8+8+8-bit split of every
 * float; g = 1 below n = 1024, 2 below 4096, else 4).  On that route the trailing updates C -= P P^T run on
 * the bf16 matrix pipe from those planes (six piece products per float product, f32 accumulation:
 * float32-accurate, csrc/b3.hpp), delayed so that all columns right of a group are updated once per group
 * with K = 256 g (same sums in the same order: bit-identical to g = 1); every other shape and dtype uses
 * the f32 / f64 MFMA.  Always size `work` with the query below.                                       */
int64_t gpvae_potrf_workspace_bytes(int dtype, int64_t n, int64_t m_extra, int batch);
int gpvae_potrf(void *stream, int dtype, int64_t n, int batch, void *A, int64_t lda,
                int64_t strideA, int64_t m_extra, void *X, int64_t ldx, int64_t strideX,
                int32_t *info, int flags, void *work);

/* ---- a9/a10: stand-alone triangular solve against an existing factor ----------------------------
 * The hot paths do not call it: every solve of gp_prediction and of the MVN potential rides
 * gpvae_potrf's extra rows.  For callers that keep L and meet new right-hand sides later, as the
 * reference re-uses jnp.linalg.inv(k_xx + sigma I) for k_px and y (inference.py:450-455).
 * L: [n, ldl] lower factor (strict upper part not referenced); X: [m, ldx], in place:
 *   trans = 1:  X <- X L^-T   (row x -> y with L y = x);     trans = 0:  X <- X L^-1   (L^T y = x);
 * both in turn: X A^-1 with A = L L^T.  m n^2 flop per call; every element of X has one writer per
 * launch (no atomics): the same bits on every call.                                               */
int gpvae_trsm(void *stream, int dtype, int trans, int64_t n, int64_t m, const void *L, int64_t ldl,
               void *X, int64_t ldx);

/* ---- a3/a4: MVN draws ---------------------------------------------------------------------------
 * Replaces f = L eps; y = f + sigma*eps' of GP.sample under Predictive (gp.py:190-194,
 * vae.py:160-162,223-226): for nb draws sharing one factor,
 *   F[r][i] = sum_{j<=i} E[r][j] * L[i][j];   Y = F + sigma * E2   (E2 may be NULL => Y = F).
 * Only the lower triangle of L is read (the strict upper part may hold anything).
 * F may be NULL (only y is wanted).  sigma: device scalar.                                         */
int gpvae_draw(void *stream, int dtype, int64_t n, int64_t nb, const void *L, int64_t ldl,
               const void *E, int64_t lde, const void *E2, int64_t lde2, const void *sigma,
               void *F, int64_t ldf, void *Y, int64_t ldy);

/* ---- a4: one draw per factor (batched factors) ----------------------------------------------------
 * Replaces Predictive(gp.sample, num_samples=nb) with hyper-parameters left None
 * (src/model/vae.py:226, src/test/toy_examples.py:88; priors gp.py:171-176): every sample b has its
 * own K_b, L_b and noise:   F[b][i] = sum_{j<=i} E[b][j] * L_b[i][j];   Y[b] = F[b] + sigma[b]*E2[b].
 * L: [nb][n][ldl], factor b at L + b*strideL (only the lower triangle is read); E, E2, F, Y: [nb, ld];
 * sigma: device array [nb].  E2 NULL => Y = F; F or Y may be NULL.  n*sizeof(T) <= 64 KB.
 * Bound: HBM (every factor is read once: nb*n^2/2*sizeof(T) bytes).                                */
int gpvae_draw_batched(void *stream, int dtype, int64_t n, int64_t nb, const void *L, int64_t ldl,
                       int64_t strideL, const void *E, int64_t lde, const void *E2, int64_t lde2,
                       const void *sigma, void *F, int64_t ldf, void *Y, int64_t ldy);

/* ---- f3: counter-based standard normals (Philox4x32-10 + Box-Muller) -------------------------------
 * Stands in for jax.random.normal (threefry; not reproducible without JAX) in
 * throughput runs; element i depends only on (seed, stream_id, i).  The integers are exact
 * (Random123 known answers); f64 normals use libm, f32 normals the hardware log2 / sqrt / sin / cos
 * instructions (absolute error <= 3e-5 against the exact Box-Muller value).                        */
int gpvae_randn(void *stream, int dtype, int64_t count, uint64_t seed, uint64_t stream_id,
                void *out);

/* ---- a3/a4 on the bf16 matrix pipe ("bf16 x 3"): float32 results at 2.5x the f32 MFMA rate --------
 * Same arithmetic contract as gpvae_draw in GPVAE_F32 (gp.py:190-194, vae.py:160-162): every float
 * operand is split EXACTLY into three bf16 pieces held as three planes in memory, and each product is
 * accumulated in f32 from the six piece products of weight >= 2^-16 (error 7e-8 of sum|e l|, below the
 * f32 MFMA's own 1.4e-7; gp_vae_amd/csrc/b3.hpp).  A "b3" array of a [rows, cols] float matrix is
 * bf16 out3[3][...]: piece p of element (i, j) at out3[p*plane + i*ld3 + j]; the pieces sum to the
 * float bit for bit.
 *   gpvae_split_b3   X [batch][rows, ldx] (f32) -> b3 planes; lower != 0 writes zeros above the
 *                    diagonal (what gpvae_draw_b3 wants of a Cholesky factor).  batch b at
 *                    X + b*strideX and out3 + b*stride3.
 *   gpvae_randn_b3   the normals of gpvae_randn(GPVAE_F32, count, seed, stream_id) written straight
 *                    as b3 planes (the f32 array is never materialised); count % 4 == 0.
 *   gpvae_draw_b3    F = E L^T, Y = F + sigma*E2 from b3 operands: L3 [n, ldl] (zeros above the
 *                    diagonal), E3 [nb, lde]; E2, F, Y as in gpvae_draw (f32).  Needs n % 128 == 0,
 *                    nb % 128 == 0 (gpvae_draw_b3_supported; 256 x 128 tiles, 128 x 128 ones when those
 *                    would leave CUs idle); leading dimensions and plane strides multiples of 8, bases
 *                    16-byte aligned.  Bound: bf16 MFMA (6 n^2 nb flop issued).
 *   gpvae_potrf_b3   gpvae_potrf(GPVAE_F32, no extra rows) that ALSO leaves the factor as b3 planes
 *                    L3 [batch][3][n, ld3] (zeros wherever gpvae_draw_b3 reads above the diagonal):
 *                    the panel launches write the planes as they go and the bf16 x 3 trailing updates
 *                    read them, so the draw's operand costs no extra pass over L.  n % 256 == 0,
 *                    n >= 512 (gpvae_potrf_b3_supported).  A, info, flags as gpvae_potrf; with
 *                    GPVAE_POTRF_NAN_ON_FAIL a failed matrix's planes are NaN on the diagonal blocks
 *                    (every draw from them is NaN).  work: gpvae_potrf_b3_workspace_bytes().        */
int gpvae_potrf_b3_supported(int64_t n);
int64_t gpvae_potrf_b3_workspace_bytes(int64_t n, int batch);
int gpvae_potrf_b3(void *stream, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                   int32_t *info, int flags, void *work, void *L3, int64_t ld3, int64_t plane,
                   int64_t stride3);
int gpvae_split_b3(void *stream, int64_t rows, int64_t cols, int batch, const void *X, int64_t ldx,
                   int64_t strideX, int lower, void *out3, int64_t ld3, int64_t plane,
                   int64_t stride3);
int gpvae_randn_b3(void *stream, int64_t count, uint64_t seed, uint64_t stream_id, void *out3,
                   int64_t plane);
int gpvae_draw_b3_supported(int64_t n, int64_t nb);
int gpvae_draw_b3(void *stream, int64_t n, int64_t nb, const void *L3, int64_t ldl, int64_t planeL,
                  const void *E3, int64_t lde, int64_t planeE, const void *E2, int64_t lde2,
                  const void *sigma, void *F, int64_t ldf, void *Y, int64_t ldy);

/* ---- the same draw on two fp16 planes ("fp16 x 2"): half the matrix work of bf16 x 3 -----------------------
 * Opt-in variant of gpvae_draw_b3 (gp.py:190-194, vae.py:160-162).  Each float operand x, scaled by a power of
 * two into fp16's range, is kept as two fp16 planes h + l (11 + 11 significand bits: x is rounded to 22 bits,
 * relative error <= 2^-22), and a product is accumulated in f32 from a0 b1 + a1 b0 + a0 b0.  Measured on a
 * K = 4096 product: max |err| / sum|a b| = 8e-8, the same as bf16 x 3 and below the f32 MFMA's 1.4e-7 (the f32
 * accumulation dominates); single products carry up to 5e-7 instead of 6e-8.  Not bit-compatible with the
 * bf16 x 3 / f32 results; every f32 parity bound of the path holds (tests).
 *   gpvae_split_h2   X [batch][rows, ldx] (f32) -> planes out2[2][...] of X * 2^e_b, e_b chosen from bound[b]
 *                    (device array: an upper bound of |entries| of matrix b — sqrt(max diag K) for a Cholesky
 *                    factor of K) so that nothing overflows; inv_scale[b] = 2^-e_b (device, f32) is written for
 *                    the draw.  lower != 0 writes zeros above the diagonal.
 *   gpvae_randn_h2   the normals of gpvae_randn(GPVAE_F32, count, seed, stream_id), times 2^12, as two planes.
 *   gpvae_draw_h2    F = E L^T, Y = F + sigma*E2 from such planes (shapes as gpvae_draw_b3_supported); the
 *                    accumulators are rescaled by inv_scale_L[0] * 2^-12 (exact).  Bound: fp16 MFMA
 *                    (3 n^2 nb flop issued).
 *   gpvae_potrf_h2   gpvae_potrf_b3 with the factor left as fp16 x 2 planes L2 [batch][2][n, ld2] of L * 2^e_b
 *                    (zeros wherever gpvae_draw_h2 reads above the diagonal) and the trailing updates on those
 *                    planes.  e_b comes from the largest diagonal entry of matrix b (|L_ij| <= sqrt(K_ii));
 *                    inv_scale[b] = 2^-e_b (device, f32) is written for the draw.  Shapes, A, info, flags as
 *                    gpvae_potrf_b3; work: gpvae_potrf_h2_workspace_bytes().  The update C - P P^T then carries
 *                    the operands' 2^-22 rounding: backward error max|L L^T - K| / max K_ii is 0.8-1.5x
 *                    gpvae_potrf's on the path's Gram matrices (measured, tests) and up to 4x (1.6e-7 against
 *                    4e-8) on near-singular K = var 11^T + 1e-4 I.                                         */
int64_t gpvae_potrf_h2_workspace_bytes(int64_t n, int batch);
int gpvae_potrf_h2(void *stream, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                   int32_t *info, int flags, void *work, void *L2, int64_t ld2, int64_t plane,
                   int64_t stride2, float *inv_scale);
int gpvae_split_h2(void *stream, int64_t rows, int64_t cols, int batch, const void *X, int64_t ldx,
                   int64_t strideX, int lower, const void *bound, void *out2, int64_t ld2, int64_t plane,
                   int64_t stride2, void *inv_scale);
int gpvae_randn_h2(void *stream, int64_t count, uint64_t seed, uint64_t stream_id, void *out2, int64_t plane);
int gpvae_draw_h2(void *stream, int64_t n, int64_t nb, const void *L2, int64_t ldl, int64_t planeL,
                  const void *inv_scale_L, const void *E2h, int64_t lde, int64_t planeE, const void *E2,
                  int64_t lde2, const void *sigma, void *F, int64_t ldf, void *Y, int64_t ldy);

/* ---- f3: hyper-prior samplers on the device -------------------------------------------------------
 * Replaces the three numpyro.sample draws of the training loop and of GP.sample
 * (src/model/vae.py:157-159, gp.py:171-176): for i in [0, count)
 *   var[i] ~ LogNormal(0, var_scale), sigma[i] ~ HalfNormal(sigma_scale),
 *   ls[i] ~ InverseGamma(ls_a, ls_b) (GPVAE_PRIOR_INVGAMMA; the code's prior is (1, 0.1), the kriging
 *           model's (4, 1): inference.py:372) or Beta(ls_a, ls_b) (GPVAE_PRIOR_BETA; README.md:17).
 * Sample i depends only on (seed, stream_id, first_index + i): batching does not change the draws.
 * Any of ls / var / sigma may be NULL.  Stands in for JAX's threefry stream like gpvae_randn.       */
enum { GPVAE_PRIOR_INVGAMMA = 0, GPVAE_PRIOR_BETA = 1 };
int gpvae_hyper_sample(void *stream, int dtype, int64_t count, uint64_t seed, uint64_t stream_id,
                       int64_t first_index, int ls_prior, double ls_a, double ls_b, double var_scale,
                       double sigma_scale, void *ls, void *var, void *sigma);

/* ---- f4: the LGCP observation of PoiGP.sample ------------------------------------------------------
 * Replaces `rate = exp(f); numpyro.sample("y", dist.Poisson(rate))` (src/model/gp.py:249-250) for
 * count elements: rate[i] = exp(f[i]) (dtype), y[i] ~ Poisson(rate[i]) (int32, JAX's default integer).
 * Element i draws from the Philox4x32-10 counters (first_index + i, attempt) of (seed, stream_id), so a
 * sample does not depend on how calls are batched: inversion for rate < 10, Hoermann's transformed
 * rejection (PTRS) above.  y = -1 where the rate is NaN, 2^31 - 1 where it exceeds that.  rate or y
 * may be NULL.  Stands in for JAX's threefry stream like gpvae_randn.                               */
int gpvae_poisson_exp(void *stream, int dtype, int64_t count, uint64_t seed, uint64_t stream_id,
                      int64_t first_index, const void *f, void *rate, int32_t *y);

/* ---- a9: kriging prediction ---------------------------------------------------------------------
 * Replaces GPInference.gp_prediction (src/model/inference.py:427-460), literal quirks kept:
 *   A = k_xx + jitter*I + sigma*I   (sigma, not sigma^2: inference.py:450)
 *   mean = k_px A^-1 y_obs;  v = diag(k_pp - k_px A^-1 k_px^T), diag(k_pp) = var + jitter;
 *   sample = mean + sqrt(max(v,0)) * xi.
 * x_obs [n_obs,d], x_test [n_test,d], y_obs [n_obs], xi [n_test] (standard normals; may be NULL =>
 * sample = mean); var, ls, sigma: device scalars.  Outputs mean, sample, var_out: [n_test] (any may
 * be NULL).  Jitter is also added to k_px's "diagonal" when n_test == n_obs (the reference's
 * square rule).  work: gpvae_krig_workspace_bytes() bytes.  info: int32, potrf status.            */
int64_t gpvae_krig_workspace_bytes(int dtype, int64_t n_obs, int64_t n_test);
int gpvae_krig_predict(void *stream, int dtype, int d, int64_t n_obs, int64_t n_test,
                       const void *x_obs, const void *x_test, const void *y_obs, const void *var,
                       const void *ls, const void *sigma, double jitter, const void *xi, void *mean,
                       void *sample, void *var_out, int32_t *info, void *work);

/* ---- f2: the same prediction for a batch of hyper-parameter draws ----------------------------------
 * Replaces the vmap of gp_prediction over posterior samples in GPInference.gp_fit
 * (src/model/inference.py:511-521): the points and observations are shared, problem b has its own
 * var[b], ls[b], sigma[b] (device arrays of `batch`) and normals xi[b][n_test] (NULL => sample =
 * mean).  Outputs mean, sample, var_out: [batch][n_test]; info: int32[batch].  One batched Gram
 * build, ONE batched factorisation with the k_px rows and y riding along, one finishing pass.
 * work: gpvae_krig_batched_workspace_bytes() bytes (about batch*(n_obs + n_test + 1)*n_obs*s).     */
int64_t gpvae_krig_batched_workspace_bytes(int dtype, int64_t n_obs, int64_t n_test, int batch);
int gpvae_krig_predict_batched(void *stream, int dtype, int d, int64_t n_obs, int64_t n_test,
                               int batch, const void *x_obs, const void *x_test, const void *y_obs,
                               const void *var, const void *ls, const void *sigma, double jitter,
                               const void *xi, void *mean, void *sample, void *var_out,
                               int32_t *info, void *work);
/* gpvae_krig_predict_batched takes one of two routes: batches of at least 8 problems with n_obs <= 160 (float32) /
 * 96 (float64) run ONE WORKGROUP PER PROBLEM with the Gram matrix, its factor and the factor's inverse resident
 * in LDS and the test points streamed through in tiles (what the reference's vmap over posterior draws asks
 * for: inference.py:511-521, n_obs = 10..100 x thousands of draws; `work` is not touched on this route);
 * everything else goes through the batched Gram + gpvae_potrf kernels.  gpvae_krig_set_path(1) forces the
 * second route (tests and benchmarks compare the two), 0 restores the choice.  No counterpart in the reference. */
void gpvae_krig_set_path(int force_big);

/* ---- a10 / f1: MVN log-density with squared-exponential covariance, and its gradient ---------------
 * Replaces what NUTS differentiates in GPInference.gp_regression (src/model/inference.py:400-424):
 *   k = kernel(x, x, var, ls) (+ jitter I);  f ~ MultivariateNormal(0, k)     (inference.py:414-422)
 * for a batch of chains: x [n,d] shared, f [batch][n], var[batch], ls[batch] (device arrays).
 *   logp[b]     = log N(f_b; 0, K_b)
 *   grad_f      [batch][n]  = -K^-1 f          (may be NULL)
 *   grad_var[b], grad_ls[b] = d logp / d var, d logp / d ls   (closed form; may be NULL)
 * info: int32[batch] potrf status; a chain whose K is not positive definite gets NaN outputs.
 * work: gpvae_mvn_workspace_bytes() bytes (about batch*(2n+1)*n*s).  The reference obtains the
 * gradient by reverse-mode through jnp.linalg.cholesky; values agree, the arithmetic differs.
 * Reproducible: row and tile sums are added in a fixed order (no floating-point atomics), and a chain's
 * results do not depend on which other chains share the batch.                                    */
int64_t gpvae_mvn_workspace_bytes(int dtype, int64_t n, int batch);
int gpvae_mvn_logp(void *stream, int dtype, int d, int64_t n, int batch, const void *x,
                   const void *f, const void *var, const void *ls, double jitter, void *logp,
                   void *grad_f, void *grad_var, void *grad_ls, int32_t *info, void *work);

/* ---- a10 as one call: potential energy of gp_regression / regression and its gradient -------------------
 * Replaces what NUTS evaluates per leapfrog step for GPInference.gp_regression (inference.py:400-424)
 * and, with lik_idx, for GPInference.regression (inference.py:359-397), in NumPyro's unconstrained space:
 * q [chains, ldq] = (f [n], log kernel_var, log kernel_length, log kernel_sigma); sites kernel_var ~
 * LogNormal(0, 0.1), kernel_sigma ~ HalfNormal(0.1), kernel_length ~ InverseGamma(4, 1),
 * f ~ MVN(0, kernel(x, x, var, ls) + jitter I), y_lik[j] ~ Normal(f[lik_idx[j]], sigma) (lik_idx NULL:
 * every point, n_lik == n).  U[c] = potential energy, dU [chains, lddu] = dU/dq.  gpvae_mvn_logp plus two
 * small kernels (exp / gather, and priors + likelihood + Jacobians); a chain whose K is not positive
 * definite gets NaN and info > 0.  work: gpvae_gp_potential_workspace_bytes().                         */
int64_t gpvae_gp_potential_workspace_bytes(int dtype, int64_t n, int chains);
int gpvae_gp_potential(void *stream, int dtype, int d, int64_t n, int chains, const void *x, const void *q,
                       int64_t ldq, const void *y_lik, const int64_t *lik_idx, int64_t n_lik,
                       double jitter, void *U, void *dU, int64_t lddu, int32_t *info, void *work);

/* ---- a5+a6: VAE ELBO forward (+ backward) ---------------------------------------------------------
 * Replaces vae_encoder/vae_decoder (src/model/vae.py:72-96) and vae_model+vae_guide under
 * Trace_ELBO (vae.py:98-137,213-218): one-particle ELBO summed over batch and dims, obs noise
 * `obs_sigma` (0.1 at vae.py:117).
 * Parameters live in ONE flat device array `params` in this order (weights stored (in,out) row
 * major as stax does):  W1[n,h0] b1[h0] W2[h0,h1] b2[h1] Wmu[h1,z] bmu[z] Wsd[h1,z] bsd[z]
 *                       V1[z,h1] c1[h1] V2[h1,h0] c2[h0] V3[h0,n] c3[n]
 * (gpvae_vae_param_count gives the length).  Y: [B, ldy] data, eps_z: [B, z] standard normals.
 * loss: device scalar (overwritten).  grads: flat array like params (overwritten) or NULL for a
 * forward-only evaluation (svi.evaluate, vae.py:194).  work: gpvae_vae_workspace_bytes() bytes; its
 * contents on entry do not matter.  Reproducible: every sum across workgroups (split-K products, weight
 * gradients over the batch, the loss) is kept as partial sums in `work` and added in a fixed order —
 * no floating-point atomics — so the same inputs give the same bits on every call.                 */
int64_t gpvae_vae_param_count(int64_t n, int h0, int h1, int z);
int64_t gpvae_vae_workspace_bytes(int dtype, int64_t B, int64_t n, int h0, int h1, int z);
int gpvae_vae_elbo(void *stream, int dtype, int64_t B, int64_t n, int h0, int h1, int z,
                   const void *params, const void *Y, int64_t ldy, const void *eps_z,
                   double obs_sigma, void *loss, void *grads, void *work);

/* ---- a5: encoder alone ----------------------------------------------------------------------------
 * Replaces the apply function of vae_encoder() (src/model/vae.py:72-85) and the guide's sample
 * z ~ Normal(z_loc, z_std) (vae.py:134-135): mu = enc_mu(Y), sd = exp(enc_logsd(Y)) — the stax.Exp
 * head returns the standard deviation — and, when eps_z and Z are given, Z = mu + sd * eps_z.
 * enc_params: the encoder head of the flat layout above (W1 b1 W2 b2 Wmu bmu Wsd bsd; a pointer to
 * the full flat vector serves).  Y: [B, ldy]; eps_z, mu, sd, Z: [B, z] contiguous.                 */
int64_t gpvae_encoder_workspace_bytes(int dtype, int64_t B, int64_t n, int h0, int h1, int z);
int gpvae_encoder_fwd(void *stream, int dtype, int64_t B, int64_t n, int h0, int h1, int z,
                      const void *enc_params, const void *Y, int64_t ldy, const void *eps_z,
                      void *mu, void *sd, void *Z, void *work);

/* ---- a7: Adam -----------------------------------------------------------------------------------
 * Replaces numpyro.optim.Adam(lr) inside svi.update (vae.py:165,212): b1, b2, eps as given;
 * step t counts from 1.  In place on params, m, v.                                                */
int gpvae_adam(void *stream, int dtype, int64_t count, void *params, const void *grads, void *m,
               void *v, int64_t t, double lr, double b1, double b2, double eps);

/* ---- a11: decoder forward / VJP for MCMC chains ----------------------------------------------------
 * Replaces decoder(decoder_params, z) of VAEInference.regression (inference.py:246) batched over
 * `chains`, and the reverse-mode product NUTS needs.  dec_params: the decoder tail of the flat
 * layout above: V1 c1 V2 c2 V3 c3.  Z: [chains, z], F: [chains, n].
 * vjp: given dF [chains, n] returns dZ [chains, z] (recomputes activations; no saved state).      */
int64_t gpvae_decoder_workspace_bytes(int dtype, int64_t chains, int64_t n, int h0, int h1, int z);
int gpvae_decoder_fwd(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1, int z,
                      const void *dec_params, const void *Z, void *F, void *work);
int gpvae_decoder_vjp(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1, int z,
                      const void *dec_params, const void *Z, const void *dF, void *dZ, void *work);

/* ---- a11 in one launch: potential energy of VAEInference.regression and its gradient ------------------
 * Replaces what NUTS evaluates per leapfrog step for src/model/inference.py:243-254 (sites noise ~
 * HalfNormal(noise_scale), z ~ N(0, I), y[obs] ~ Normal(decoder(z)[obs], noise)) in NumPyro's
 * unconstrained space, for a batch of chains: q [chains, ldq] = (z [z_dim], log noise),
 *   U[c] = potential energy,  dU [chains, lddu] = dU/dq.
 * The likelihood reads the decoder's output only at obs_idx (int64 [n_obs], device), so the output
 * layer is evaluated on the gathered columns: n_obs*h0 work per chain, everything in LDS, one launch.
 * For decoders whose head fits LDS (gpvae_vae_potential_supported; the reference's 10-30-35-300 and
 * 15-30-50-625 do); otherwise compose gpvae_decoder_fwd / gpvae_decoder_vjp.                          */
int gpvae_vae_potential_supported(int dtype, int h0, int h1, int z, int64_t n_obs);
int gpvae_vae_potential(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1, int z,
                        const void *dec_params, const void *q, int64_t ldq, const void *y_obs,
                        const int64_t *obs_idx, int64_t n_obs, double noise_scale, void *U, void *dU,
                        int64_t lddu);

/* ---- f4: the batched NUTS transition around a potential ---------------------------------------------
 * Replaces the per-leapfrog arithmetic of numpyro.infer.NUTS as the reference drives it
 * (src/model/inference.py:84-95, 477-487; NumPyro itself is third party): multinomial NUTS with the
 * iterative tree builder, all chains in lock step, one wave per chain.  The potential-and-gradient
 * evaluation stays the caller's (gpvae_vae_potential / gpvae_gp_potential or any other); this entry point
 * is everything between two such evaluations, so that a leapfrog step costs two small launches instead
 * of some forty elementwise ones.  All arrays are device memory of `dtype`, [C, D] row-major or [C]
 * unless noted; `op` selects the phase (gp_vae_amd/nuts.py `nuts_device` is the driver):
 *   GPVAE_NUTS_INIT    momentum p0 = z / sqrt(minv), H0, one-node tree at (q, p0, g)
 *   GPVAE_NUTS_BEGIN   doubling `depth`: direction from unif[:, depth], empty sub-tree at that edge
 *   GPVAE_NUTS_PRE     half kick and drift from the edge: ph, qn   (then the caller evaluates Un, gn at qn)
 *   GPVAE_NUTS_POST    second half kick, energy error, divergence (> 1000), multinomial choice inside the
 *                      sub-tree (unif[:, 2 max_depth + 2^depth - 1 + leaf]), momentum sum, U-turn checks
 *                      against the checkpoints [ck_lo, ck_hi] of leaf `leaf`; flags[0] |= a chain still builds
 *   GPVAE_NUTS_POST_PRE  POST of leaf `leaf` followed by PRE of the next leaf in one launch (two launches per
 *                      leapfrog step instead of three); harmless when the caller then leaves the sub-tree
 *   GPVAE_NUTS_MERGE   biased progressive choice between the tree and the new sub-tree
 *                      (unif[:, max_depth + depth]), edges, weights, U-turn of the whole tree;
 *                      flags[1] |= a chain is still active
 *   GPVAE_NUTS_ACCEPT  q, U, g = the proposal; accept[c] = mean acceptance statistic of the tree
 * BEGIN clears flags.  A chain whose tree has stopped rides along untouched, so its numbers do not depend
 * on the other chains of the batch.                                                                   */
enum { GPVAE_NUTS_INIT = 0, GPVAE_NUTS_BEGIN = 1, GPVAE_NUTS_PRE = 2, GPVAE_NUTS_POST = 3, GPVAE_NUTS_MERGE = 4,
       GPVAE_NUTS_ACCEPT = 5, GPVAE_NUTS_POST_PRE = 6 };
typedef struct gpvae_nuts_buffers {
  void *q, *U, *g;                /* chain state: position, potential, gradient */
  const void *minv, *eps;         /* inverse mass diagonal [C, D], step size [C] */
  const void *z, *unif;           /* the iteration's normals [C, D] and uniforms [C, ldu] */
  int64_t ldu;
  void *ql, *pl, *gl, *qr, *pr, *gr;              /* tree edges */
  void *q_prop, *U_prop, *g_prop, *logW, *p_sum;  /* proposal, log weight, momentum sum of the tree */
  void *H0, *sum_acc, *n_prop, *accept;           /* [C] */
  int32_t *active, *diverged;                     /* [C] */
  void *qe, *pe, *ge, *es;                        /* moving edge of the sub-tree; signed step [C] */
  void *sub_logW, *sub_psum, *sq, *sU, *sg;       /* sub-tree weight, momentum sum, proposal */
  int32_t *sub_stop, *building, *right;           /* [C] */
  void *ck_p, *ck_s;                              /* checkpoints [max_depth + 1][C][D] */
  void *ph, *qn;                                  /* PRE's outputs */
  const void *Un, *gn;                            /* the caller's potential [C] and gradient [C, ldg] at qn */
  int64_t ldg;
  int32_t *flags;                                 /* [2] */
} gpvae_nuts_buffers;
int gpvae_nuts_op(void *stream, int dtype, int op, int64_t C, int64_t D, int max_depth, int depth, int leaf,
                  int ck_lo, int ck_hi, const gpvae_nuts_buffers *b);

/* ---- f4 + a11 fused: one whole NUTS iteration of VAEInference.regression per launch ------------------
 * Replaces mcmc.run's inner loop for the decoder-prior model (src/model/inference.py:84-95 driving
 * inference.py:235-254) — what gpvae_nuts_op (INIT, then per doubling BEGIN, per leapfrog step PRE /
 * potential / POST, MERGE; ACCEPT) and gpvae_vae_potential do in ~2 launches per leapfrog step — as ONE
 * launch per iteration: four chains per workgroup, one wave each, the decoder's head and the output layer
 * gathered at obs_idx in LDS; every chain runs to the end of its own tree (no lock step with the deepest
 * tree of the batch).  Same phases, uniforms (unif[:, ...] indexed by depth and leaf) and arithmetic per
 * chain as the lock-step driver.  D = z + 1: q = (z, log noise).  b->Un [C] and b->gn [C, ldg] are the
 * kernel's scratch here; b->flags is not used.  Decoders whose head does not fit 64 KB of LDS:
 * gpvae_nuts_vae_supported() == 0, drive gpvae_nuts_op around gpvae_vae_potential instead.             */
int gpvae_nuts_vae_supported(int dtype, int h0, int h1, int z, int64_t n_obs);
int gpvae_nuts_vae_iteration(void *stream, int dtype, int64_t C, int max_depth, const gpvae_nuts_buffers *b,
                             int64_t n, int h0, int h1, int z, const void *dec_params, const void *y_obs,
                             const int64_t *obs_idx, int64_t n_obs, double noise_scale);

/* ---- f4 + a11: a whole MCMC run of VAEInference.regression in ONE launch -----------------------------
 * Replaces mcmc.run(rng_key, y, obs_idx) for the decoder-prior model (src/model/inference.py:84-95 with
 * the model of inference.py:235-254): num_warmup + num_samples NUTS iterations of every chain — chains are
 * independent, so each wave takes ITS chain through the whole run: transitions as in
 * gpvae_nuts_vae_iteration, the iteration's normals and uniforms drawn in place from the Philox stream
 * (seed, stream_base | iteration) at element chain_ids[c] * row_len + k (the numbers gpvae_randn writes
 * there; uniform = Phi(normal)), dual-averaging step size and a diagonal mass matrix over Stan's warm-up
 * windows per chain.  The launch lasts as long as the chain with the most leapfrog steps over the whole
 * run, not the deepest tree of every iteration.  Outputs: samples [ceil(num_samples / thinning)][C][D]
 * (D = z + 1: z, log noise), per chain the mean acceptance statistic and leapfrog steps per sampling
 * iteration, the adapted step size, inverse mass diagonal [C, D] and the number of divergent transitions. */
typedef struct gpvae_nuts_run {
  int32_t num_warmup, num_samples, thinning, adapt_mass, max_depth;
  int32_t win_start, n_ends, ends[24]; /* mass-matrix windows: first adapted iteration, last iteration of each window */
  uint64_t seed, stream_base;
  int64_t row_len;                     /* >= D + 2 max_depth + 2^max_depth - 1 */
  const int64_t *chain_ids;            /* [C] global chain ids (device), or NULL: 0 .. C-1 */
  double step_size, target_accept;
  const void *q0;                      /* [C, D] start points */
  void *samples, *accept_prob, *leapfrogs, *step_size_out, *inverse_mass;
  int32_t *divergent;
} gpvae_nuts_run;
int gpvae_nuts_vae_run(void *stream, int dtype, int64_t C, const gpvae_nuts_run *r, int64_t n, int h0, int h1,
                       int z, const void *dec_params, const void *y_obs, const int64_t *obs_idx, int64_t n_obs,
                       double noise_scale);

/* ---- the same VJP with no workspace argument and no allocation ------------------------------------
 * No counterpart in the reference.  For callers that cannot allocate or carry a workspace: code
 * under HIP graph capture, tight MCMC / serving loops, harnesses that forbid hipMalloc.  Same kernels
 * and results as gpvae_decoder_vjp; the scratch (gpvae_decoder_workspace_bytes: chains * (h1 + (2 + s) h0)
 * elements + 256 bytes, s <= 32 the number of k-slices of its one n-sized product) lives in
 * a static device array of GPVAE_STATIC_SCRATCH_BYTES = 4 MiB per GPU, held until the process ends.  Larger problems are rejected with -3 and a message.  Calls on one GPU share that
 * array: calls that overlap on one GPU (two streams, two threads) race — serialise them.  The first
 * call on each device looks the array's address up (hipGetSymbolAddress): make it outside a capture. */
#define GPVAE_STATIC_SCRATCH_BYTES (4 << 20)
int gpvae_decoder_vjp_static(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1,
                             int z, const void *dec_params, const void *Z, const void *dF, void *dZ);

#ifdef __cplusplus
}
#endif
#endif /* GPVAE_H */
