// trsm.hip — stand-alone triangular solve against an existing Cholesky factor (SURVEY.md §8b's export
// list; rows a9 / a10).
//
// The hot paths never need it: every solve of the kriging / MVN paths rides the factorisation's
// panel sweep as extra rows of gpvae_potrf (one pass over L).  This entry point is for callers that
// KEEP a factor and meet new right-hand sides later — the reference's posterior code re-uses
// `jnp.linalg.inv(k_xx + sigma I)` (src/model/inference.py:450-455) for `k_px` and `y` alike.
//
//   trans = 1:  X <- X L^-T   (every row x of X becomes the y with L y = x:   forward substitution)
//   trans = 0:  X <- X L^-1   (every row x of X becomes the y with L^T y = x: backward substitution)
// so X L^-T L^-1 = X A^-1 for A = L L^T.  Blocked by 64 columns, right-looking: each row finishes the 64
// unknowns of a block against the diagonal block held in LDS, then the block just solved updates the columns
// still to come through the generic tile product, X_rest -= X_block L_rest,block^T — inside its outer block
// of 256 columns at once (K = 64), everything beyond when the outer block is done (K = 256).  No split over
// K: every element of X is written by one lane per launch — no atomics, the same bits on every call.
// The update is "subtract the product from C" block by block — fine for a solve against a given factor (the
// ordering that matters on nearly singular matrices is the factorisation's, potrf.hip).
#include "gemm_tile.hpp"

namespace gpvae {

constexpr int TSB = 64;   // column block of a diagonal solve
constexpr int TOB = 256;  // outer block: columns solved between two updates of everything still to come

// rows [row0, row0+256) of X, columns [c0, c0+w): finish the solve against the diagonal block of L.
// One row per thread, its 64 entries in registers.  (Measured and dropped: the rows through an LDS image
// with row-by-row global accesses — 4.4 instead of 3.95 ms per 4096 x 4096 call; 16-byte instead of 4-byte
// broadcast reads of the multipliers and four partial sums per step are kept but made no difference:
// a launch takes 35 us for ~10 k cycles of work per wave on 16 workgroups.)
template <typename T, bool TRANS>
__global__ __launch_bounds__(NTHREADS) void trsm_diag_kernel(int64_t m, int w, const T *__restrict__ Ld,
                                                             int64_t ldl, T *X, int64_t ldx) {
  constexpr int VEC = VecT<T>::N;
  using V = typename VecT<T>::type;
  __shared__ __align__(16) T Lc[TSB][TSB + VEC];
  __shared__ T rd[TSB];  // reciprocals of the diagonal
  // Both variants run the SAME sweep, y_c = (x_c - sum_{k>c} y_k U[c][k]) / U[c][c], c descending:
  // backward substitution has U = L^T; forward substitution L y = x is that sweep on the index-reversed
  // problem (U = P L P, P = reversal).  (The ascending loop written directly made the compiler keep
  // every partial sum and every L entry live: 512 registers + 6.5 KB of scratch, 240 us a launch.)
  // Image: Lc[c][k] = U[c][k] — the multipliers of step c lie side by side, one 16-byte broadcast read
  // feeds VEC products; columns / rows >= w are an identity block.
  constexpr int R = TSB - 1;
  for (int idx = threadIdx.x; idx < TSB * TSB; idx += NTHREADS) {
    const int i = idx / TSB, j = idx % TSB;  // L[i][j]
    const T v = (i < w && j <= i) ? Ld[(int64_t)i * ldl + j] : (i == j ? T(1) : T(0));
    if (TRANS) Lc[R - i][R - j] = v;
    else Lc[j][i] = v;
  }
  if (threadIdx.x < TSB) {
    const int d = TRANS ? R - (int)threadIdx.x : (int)threadIdx.x;
    rd[threadIdx.x] = d < w ? T(1) / Ld[(int64_t)d * ldl + d] : T(1);
  }
  __syncthreads();
  const int64_t r = (int64_t)blockIdx.x * NTHREADS + threadIdx.x;
  if (r >= m) return;
  T *xr = X + r * ldx;
  T x[TSB];
#pragma unroll
  for (int c = 0; c < TSB; ++c) {
    const int s = TRANS ? R - c : c;
    x[c] = s < w ? xr[s] : T(0);
  }
#pragma unroll
  for (int c = TSB - 1; c >= 0; --c) {
    // VEC independent partial sums: the chain of dependent FMAs of a step is (63 - c) / VEC long
    T acc[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) acc[e] = T(0);
    acc[0] = x[c];
#pragma unroll
    for (int kb = ((c + 1) / VEC) * VEC; kb < TSB; kb += VEC) {
      const V t = *reinterpret_cast<const V *>(&Lc[c][kb]);
      const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (kb + e > c) acc[e] -= x[kb + e] * tp[e];
    }
    T a = acc[0];
#pragma unroll
    for (int e = 1; e < VEC; ++e) a += acc[e];
    x[c] = a * rd[c];
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int c = 0; c < TSB; ++c) {
    const int s = TRANS ? R - c : c;
    if (s < w) xr[s] = x[c];
  }
}

template <typename T>
static int run_trsm(hipStream_t st, int trans, int64_t n, int64_t m, const void *L_v, int64_t ldl,
                    void *X_v, int64_t ldx) {
  const T *L = (const T *)L_v;
  T *X = (T *)X_v;
  const unsigned dgrid = (unsigned)cdiv(m, NTHREADS);
  // two levels: inside an outer block of 256 columns a solved 64-block updates the rest of the outer block
  // (K = 64, few tiles); a finished outer block updates all columns still to come in one K = 256 product
  const int64_t nob = cdiv(n, TOB);
  auto update = [&](int64_t a0, int64_t k, int64_t cb, int64_t cn) {
    // X[:, cb:cb+cn] -= X[:, a0:a0+k] * (trans ? L[cb:cb+cn, a0:a0+k]^T : L[a0:a0+k, cb:cb+cn])
    if (cn <= 0 || k <= 0) return;
    GemmArgs<T> g{};
    g.A = X + a0; g.lda = ldx; g.K = k; g.M = m; g.C = X + cb; g.ldc = ldx; g.N = cn;
    g.addc = 1; g.negate = 1;  // C - A B, one k-slice: plain read-modify-write by the owning lane
    g.a_ones_row = -1; g.b_ones_row = -1;
    g.vec_a = kc_vec_ok<T>(g.A, ldx);
    if (trans) {  // both operands K-contiguous
      g.B = L + cb * ldl + a0; g.ldb = ldl; g.vec_b = kc_vec_ok<T>(g.B, ldl);
      launch_gemm<T, true, true, EPI_FUSED>(st, g, 1);
    } else {      // B is N-contiguous
      g.B = L + a0 * ldl + cb; g.ldb = ldl; g.vec_b = mc_vec_ok<T>(g.B, ldl);
      launch_gemm<T, true, false, EPI_FUSED>(st, g, 1);
    }
  };
  for (int64_t ob = 0; ob < nob; ++ob) {
    const int64_t o = trans ? ob : nob - 1 - ob;  // forward / backward over the outer blocks
    const int64_t o0 = o * TOB, oend = (o0 + TOB < n) ? o0 + TOB : n;
    const int64_t nib = cdiv(oend - o0, (int64_t)TSB);
    for (int64_t ib = 0; ib < nib; ++ib) {
      const int64_t c0 = o0 + (trans ? ib : nib - 1 - ib) * TSB;
      const int w = (int)((c0 + TSB <= oend) ? TSB : oend - c0);
      if (trans)
        hipLaunchKernelGGL((trsm_diag_kernel<T, true>), dim3(dgrid), dim3(NTHREADS), 0, st, m, w,
                           L + c0 * ldl + c0, ldl, X + c0, ldx);
      else
        hipLaunchKernelGGL((trsm_diag_kernel<T, false>), dim3(dgrid), dim3(NTHREADS), 0, st, m, w,
                           L + c0 * ldl + c0, ldl, X + c0, ldx);
      if (trans) update(c0, w, c0 + w, oend - (c0 + w));
      else update(c0, w, o0, c0 - o0);
    }
    if (trans) update(o0, oend - o0, oend, n - oend);
    else update(o0, oend - o0, 0, o0);
  }
  GPVAE_CHECK_LAUNCH("gpvae_trsm");
  return 0;
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int gpvae_trsm(void *stream, int dtype, int trans, int64_t n, int64_t m, const void *L,
                          int64_t ldl, void *X, int64_t ldx) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_trsm: bad dtype %d", dtype);
  GPVAE_REQUIRE(trans == 0 || trans == 1, 3, "gpvae_trsm: trans must be 0 or 1");
  GPVAE_REQUIRE(n >= 0, 4, "gpvae_trsm: n < 0");
  GPVAE_REQUIRE(m >= 0, 5, "gpvae_trsm: m < 0");
  if (n == 0 || m == 0) return 0;
  GPVAE_REQUIRE(L, 6, "gpvae_trsm: L is NULL");
  GPVAE_REQUIRE(ldl >= n, 7, "gpvae_trsm: ldl < n");
  GPVAE_REQUIRE(X, 8, "gpvae_trsm: X is NULL");
  GPVAE_REQUIRE(ldx >= n, 9, "gpvae_trsm: ldx < n");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32 ? run_trsm<float>(st, trans, n, m, L, ldl, X, ldx)
                            : run_trsm<double>(st, trans, n, m, L, ldl, X, ldx);
}
