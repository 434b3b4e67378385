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
#include "gemm_core.hpp"

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

// ---- f2: one workgroup per SMALL problem, everything in LDS ----------------------------------------------
// The reference calls gp_prediction under vmap over thousands of posterior draws with n_obs = 10..100
// (inference.py:511-521, toy_examples.py:199,639-642): thousands of tiny independent problems.  Each workgroup
// takes one: builds A = k_xx + jitter I + sigma I in LDS, factors it there (column sweeps, the pivot scaling of
// the big factorisation), inverts the factor in place (U = L^-1, row by row), and then streams the test
// points through in tiles of TN: the tile's k_px rows are generated into LDS and X = U k_px^T is one MFMA
// product against the resident U (U is lower triangular: row tile t needs k < (t + 1) TM only), reduced on
// the fly to  mean_p = X_p . (U y)  and  v_p = (var + jitter) - |X_p|^2.  Nothing but x, y, the three
// hyper-parameters and the outputs touches HBM.  KP = n_obs rounded up to the MFMA tile (compile time: it
// fixes the LDS row stride); rows / columns between n_obs and KP act as an identity block.
// k_px entries of the streamed test tiles: 1.5 million exponentials per 8000-draw call are a third of a tile's
// time with the Gram builder's compensated exp.  float32 here: exp2 of -(log2 e / 2) s / ls^2 in one rounding of
// the argument (relative error |x| 1e-7: below 2e-6 wherever the entry is above 1e-9 of var); float64 as the builder.
template <typename T> struct KrigExp {
  __device__ static __forceinline__ T expq(T s, T inv_ls) { return GramMath<T>::expq(s, inv_ls); }
};
template <> struct KrigExp<float> {
  __device__ static __forceinline__ float expq(float s, float inv_ls) {
    return __builtin_amdgcn_exp2f(-0.72134752044448170368f * (s * (inv_ls * inv_ls)));
  }
};

template <typename T> struct KrigSmall {
  static constexpr int TM = Mfma<T>::TM;            // 32 (f32) / 16 (f64)
  static constexpr int BK = TileK<T>::BK;           // 32 / 16
  static constexpr int PAD = RmPad<T>::V;           // 4 / 2
  static constexpr int TN = 32;                     // test points per tile
  static constexpr int FN = TN / TM;
  // rows of the resident image: n_obs rounded up to 4 (row tiles of the MFMA product read past them into the
  // test tile's image — finite or not, those rows are masked out of the reduction)
  static int64_t lds_bytes(int kp, int64_t n_obs) {
    const int ld = kp + PAD;
    return (int64_t)sizeof(T) * (round_up(n_obs, 4) * ld + (int64_t)TN * ld + 2 * kp + 4 * TN * 2 + 2 * kp) + 64;
  }
};

// 32 x 32 block products on row-major LDS images of stride LD (MFMA tiles TM x TM, FB x FB of them per block)
template <typename T, int LD> struct Blk32 {
  using M = Mfma<T>;
  static constexpr int TM = M::TM, FB = 32 / TM, BK = TileK<T>::BK;
  using acc_t = typename M::acc_t;
  // acc += A B^T (BRM: B row-major [j][k]) or A B (BRM == false: B's storage [k][j] read as a k-major image)
  template <bool BRM>
  __device__ static __forceinline__ void mma(acc_t (&acc)[FB][FB], const T *Ap, const T *Bp, int lane) {
#pragma unroll
    for (int k0 = 0; k0 < 32; k0 += BK)
      mma_ktile<T, FB, FB, BK, true, LD, BRM, LD>(Ap + k0, BRM ? Bp + k0 : Bp + k0 * LD, 0, 0, lane, acc);
  }
  __device__ static __forceinline__ void load_neg(acc_t (&acc)[FB][FB], const T *Cp, int lane) {
#pragma unroll
    for (int fm = 0; fm < FB; ++fm)
#pragma unroll
      for (int fn = 0; fn < FB; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) acc[fm][fn][r] = -Cp[(fm * TM + M::row(lane, r)) * LD + fn * TM + M::col(lane)];
  }
  // rows < rmax and columns < cmax of the block only (the last block row / column of a ragged matrix)
  __device__ static __forceinline__ void store(T *Cp, const acc_t (&acc)[FB][FB], T sign, int lane, int rmax, int cmax) {
#pragma unroll
    for (int fm = 0; fm < FB; ++fm)
#pragma unroll
      for (int fn = 0; fn < FB; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) {
          const int i = fm * TM + M::row(lane, r), j = fn * TM + M::col(lane);
          if (i < rmax && j < cmax) Cp[i * LD + j] = sign * acc[fm][fn][r];
        }
  }
};

// One wave: Cholesky factor of the 32 x 32 diagonal block at Dp (lane r < 32 holds row r in registers, row c of the
// factor reaches the other lanes by v_readlane; the pivot scaling of the big factorisation) and, from the same
// registers, the factor's inverse V = L^-1 (lane c solves column c by forward substitution), which REPLACES the
// block: the panel below it is then P = A V^T and the factor's inverse is assembled from the V's — products on
// the matrix pipe instead of substitution sweeps.  Rows >= nvalid act as identity rows.  Returns the first
// failing column (global index) or `fail` unchanged.
template <typename T, int LD>
__device__ __forceinline__ int diag_factor_invert(T *Dp, int nvalid, int lane, int col0, int fail) {
  // lanes 0..31: row r of the block (factor); lanes 32..63: column r of the inverse.  Step c needs row c of the
  // factor in every lane — ONE v_readlane per entry serves both halves: the factor's update s -= L[r][k] L[c][k]
  // and the inverse's forward substitution a -= L[c][k] V[k][r] are the same instruction on the shared array v.
  const int r = lane & 31;
  const bool lo = lane < 32;
  T v[32];
#pragma unroll
  for (int k = 0; k < 32; ++k) v[k] = lo ? ((r < nvalid) ? Dp[r * LD + k] : (k == r ? T(1) : T(0))) : T(0);
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    T t = lo ? v[c] : (r == c ? T(1) : T(0));
#pragma unroll
    for (int k = 0; k < c; ++k) t -= v[k] * lane_bcast<T>(v[k], c);  // C - sum, in column order
    const T piv = lane_bcast<T>(t, c);
    if (!(piv > T(0)) && fail < 0) fail = col0 + c;
    T ljj, rd;
    pivot_scale<T>(piv, ljj, rd);
    // factor: column c scaled below the diagonal; inverse: V[c][r] (zero above the diagonal falls out: r > c gives
    // t = 0 - sum over zeros)
    v[c] = lo ? ((r > c) ? div_scaled<T>(t, ljj, rd) : (r == c ? ljj : T(0))) : t * rd;
  }
  if (!lo) {
#pragma unroll
    for (int i = 0; i < 32; ++i)
      if (i < nvalid) Dp[i * LD + r] = v[i];
  }
  return fail;
}

template <typename T, int DIM, int KP>
__global__ __launch_bounds__(NTHREADS, 2) void krig_small_kernel(int n, int64_t n_test, const T *__restrict__ x_obs,
                                                              const T *__restrict__ x_test,
                                                              const T *__restrict__ y_obs, const T *var_p,
                                                              const T *ls_p, const T *sigma_p, T jitter,
                                                              const T *xi, T *mean, T *sample, T *var_out,
                                                              int32_t *info) {
  using M = Mfma<T>;
  using KS = KrigSmall<T>;
  constexpr int TM = KS::TM, BK = KS::BK, LD = KP + KS::PAD, TN = KS::TN, FN = KS::FN;
  constexpr int MT = KP / TM;            // row tiles of U
  constexpr int FMW = (MT + 3) / 4;      // row tiles per wave (tile t belongs to wave t % 4)
  extern __shared__ __align__(16) unsigned char krig_smem[];
  const int nr = (n + 3) & ~3;               // rows of the resident image
  T *Lm = reinterpret_cast<T *>(krig_smem);  // [nr][LD]: A, then L, then U = L^-1 (lower; zeros above)
  T *Bt = Lm + nr * LD;                      // [TN][LD]: k_px rows of the current tile (K-contiguous)
  T *wv = Bt + TN * LD;                      // [KP]: U y
  T *rowb = wv + KP;                         // [KP]: a row of L while it is being replaced by its row of U
  T *red = rowb + KP;                        // [4 waves][16][2] (sized [4][TN][2])
  T *xs = red + 4 * TN * 2;                  // [n][DIM]: the observed points
  __shared__ int fail_s;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.x;
  const T var = var_p[b], inv_ls = T(1) / ls_p[b], sigma = sigma_p[b];
  if (tid == 0) fail_s = -1;
  for (int e = tid; e < n * DIM; e += NTHREADS) xs[e] = x_obs[e];
  __syncthreads();

  // ---- A (lower triangle; identity beyond n; zeros above the diagonal) --------------------------------------
  for (int e = tid; e < nr * KP; e += NTHREADS) {
    const int i = e / KP, k = e - i * KP;
    T v = T(0);
    if (i < n && k <= i) {
      T s = T(0);
#pragma unroll
      for (int d = 0; d < DIM; ++d) {
        const T dl = xs[i * DIM + d] - xs[k * DIM + d];
        s += dl * dl;
      }
      v = var * GramMath<T>::expq(s, inv_ls);
      if (i == k) v = (v + jitter) + sigma;  // k_xx (+ jitter I: square) + eye * sigma   inference.py:449-450
    } else if (i >= n && i == k) {
      v = T(1);
    }
    Lm[i * LD + k] = v;
  }
  for (int i = tid; i < KP; i += NTHREADS) wv[i] = i < n ? y_obs[i] : T(0);
  // block rows of the ragged last block reach past the image into the test tile's buffer: zeros there (they meet
  // zero columns of V in the products below; 0 x whatever an uninitialised LDS word holds could be NaN)
  for (int e = tid; e < TN * LD; e += NTHREADS) Bt[e] = T(0);
  __syncthreads();

  // ---- factor and invert, blocked by 32: per block column one wave factors the diagonal block and leaves its
  // inverse V in place; the panel is P = A V^T, the trailing update C -= P P^T (accumulators start from -C), then
  // U = L^-1 is assembled block row by block row: M_IK = V_I L_IK, U_IJ = -sum_{J <= K < I} M_IK U_KJ (U_JJ = V_J).
  // All on the matrix pipe but the 32 x 32 diagonal blocks; 3 barriers per block column / block row.
  {
    using B32 = Blk32<T, LD>;
    constexpr int NBMAX = KP / 32;
    static_assert(NBMAX <= 5, "one accumulator block per wave and block row");
    const int NBL = (n + 31) / 32;
    int fail = -1;
    for (int J = 0; J < NBL; ++J) {
      T *DJ = Lm + (32 * J) * LD + 32 * J;
      if (wave == 0) fail = diag_factor_invert<T, LD>(DJ, nr - 32 * J < 32 ? nr - 32 * J : 32, lane, 32 * J, fail);
      __syncthreads();
      for (int I = J + 1 + wave; I < NBL; I += 4) {
        typename B32::acc_t acc[B32::FB][B32::FB];
        zero_acc<T, B32::FB, B32::FB>(acc);
        T *AIJ = Lm + (32 * I) * LD + 32 * J;
        B32::template mma<true>(acc, AIJ, DJ, lane);
        B32::store(AIJ, acc, T(1), lane, nr - 32 * I, 32);
      }
      __syncthreads();
      int t = 0;
      for (int I = J + 1; I < NBL; ++I)
        for (int K = J + 1; K <= I; ++K, ++t) {
          if ((t & 3) != wave) continue;
          typename B32::acc_t acc[B32::FB][B32::FB];
          T *AIK = Lm + (32 * I) * LD + 32 * K;
          B32::load_neg(acc, AIK, lane);
          B32::template mma<true>(acc, Lm + (32 * I) * LD + 32 * J, Lm + (32 * K) * LD + 32 * J, lane);
          B32::store(AIK, acc, T(-1), lane, nr - 32 * I, n - 32 * K);
        }
      __syncthreads();
    }
    if (wave == 0 && lane == 0) fail_s = fail;
    for (int I = 1; I < NBL; ++I) {
      const T *VI = Lm + (32 * I) * LD + 32 * I;
      for (int K = wave; K < I; K += 4) {
        typename B32::acc_t acc[B32::FB][B32::FB];
        zero_acc<T, B32::FB, B32::FB>(acc);
        T *LIK = Lm + (32 * I) * LD + 32 * K;
        B32::template mma<false>(acc, VI, LIK, lane);
        B32::store(LIK, acc, T(1), lane, nr - 32 * I, 32);
      }
      __syncthreads();
      typename B32::acc_t accu[B32::FB][B32::FB];
      zero_acc<T, B32::FB, B32::FB>(accu);
      const int Jw = wave;  // block (I, Jw): I <= 4, so one block per wave
      if (Jw < I)
        for (int K = Jw; K < I; ++K)
          B32::template mma<false>(accu, Lm + (32 * I) * LD + 32 * K, Lm + (32 * K) * LD + 32 * Jw, lane);
      __syncthreads();  // every wave has read row I's M blocks
      if (Jw < I) B32::store(Lm + (32 * I) * LD + 32 * Jw, accu, T(-1), lane, nr - 32 * I, 32);
      __syncthreads();
    }
  }
  // ---- w = U y (row i by thread i)
  {
    T a = T(0);
    if (tid < n)
#pragma unroll 4
      for (int k = 0; k <= tid; ++k) a += Lm[tid * LD + k] * wv[k];
    __syncthreads();
    if (tid < KP) wv[tid] = tid < n ? a : T(0);
    // (KP <= 256 = NTHREADS)
  }
  if (tid == 0 && info != nullptr) info[b] = fail_s < 0 ? 0 : fail_s + 1;
  __syncthreads();

  // ---- test points, TN = 32 at a time --------------------------------------------------------------------------
  // X = U k_px^T on 16 x 16 MFMA tiles: wave w takes the test points 16 (w & 1) .. + 15 and one of two sets of
  // 16-row tiles of U chosen so that both sets hold the same number of k chunks (row tile t of the lower
  // triangular U needs k < 16 (t + 1): tiles t = 0, 3 (mod 4) against t = 1, 2 (mod 4)) — every wave, i.e. every
  // SIMD's matrix pipe, gets the same work.  k values are dealt to the four lane groups in runs of a 16-byte
  // load, the same way for both operands.
  const bool sq_rule = n_test == (int64_t)n;  // the reference's kernel adds jitter to ANY square matrix (gp.py:79-80)
  using M16 = Mfma16<T>;
  using V = typename VecT<T>::type;
  constexpr int KV = VecT<T>::N, CK = 4 * KV, NC = KP / CK, MT16 = KP / 16;
  // (half, mset) from a scalar wave index: `mine` below gates MFMAs, which do not honour EXEC — a real branch
  const int wsc = __builtin_amdgcn_readfirstlane(wave);
  const int lr = lane & 15, g = lane >> 4, half = wsc & 1, mset = wsc >> 1;
  T wreg[MT16][4];
  bool rok[MT16][4];
#pragma unroll
  for (int t = 0; t < MT16; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = 16 * t + M16::row(lane, r);
      const bool mine = ((((t & 3) == 0) || ((t & 3) == 3)) ? 0 : 1) == mset;
      rok[t][r] = mine && row < n;
      wreg[t][r] = rok[t][r] ? wv[row] : T(0);
    }
  for (int64_t p0 = 0; p0 < n_test; p0 += TN) {
    {  // the tile's k_px rows: thread -> (test point tid / 8, columns tid % 8 + 8 j)
      const int p = tid >> 3;
      const bool pok = p0 + p < n_test;
      T xt[DIM];
#pragma unroll
      for (int d = 0; d < DIM; ++d) xt[d] = pok ? x_test[(p0 + p) * DIM + d] : T(0);
      for (int k = tid & 7; k < KP; k += 8) {
        T v = T(0);
        if (pok && k < n) {
          T s = T(0);
#pragma unroll
          for (int d = 0; d < DIM; ++d) {
            const T dl = xt[d] - xs[k * DIM + d];
            s += dl * dl;
          }
          v = var * KrigExp<T>::expq(s, inv_ls);
          if (sq_rule && p0 + p == k) v += jitter;
        }
        Bt[p * LD + k] = v;
      }
    }
    __syncthreads();
    typename M16::acc_t acc[MT16];
#pragma unroll
    for (int t = 0; t < MT16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) acc[t][r] = T(0);
    const T *bp = Bt + (16 * half + lr) * LD + g * KV;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      const V bv = *reinterpret_cast<const V *>(bp + c * CK);
      const T *bq = reinterpret_cast<const T *>(&bv);
#pragma unroll
      for (int t = 0; t < MT16; ++t) {
        if (16 * (t + 1) <= c * CK) continue;  // U[i][k] = 0 for k > i: nothing of this chunk in tile t
        const bool mine = ((((t & 3) == 0) || ((t & 3) == 3)) ? 0 : 1) == mset;  // wave-uniform
        if (mine) {
          const V av = *reinterpret_cast<const V *>(Lm + (16 * t + lr) * LD + c * CK + g * KV);
          const T *aq = reinterpret_cast<const T *>(&av);
#pragma unroll
          for (int j = 0; j < KV; ++j) acc[t] = M16::mma(aq[j], bq[j], acc[t]);
        }
      }
    }
    T dotp = T(0), sqp = T(0);
#pragma unroll
    for (int t = 0; t < MT16; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const T xv = rok[t][r] ? acc[t][r] : T(0);  // rows past n_obs (and the other set's tiles) do not count
        dotp += xv * wreg[t][r];
        sqp += xv * xv;
      }
    // a test point's rows sit in the four lane groups: fold them, then the two row sets through LDS
#pragma unroll
    for (int off = 16; off < 64; off <<= 1) {
      dotp += __shfl_xor(dotp, off);
      sqp += __shfl_xor(sqp, off);
    }
    if (lane < 16) {
      red[(wave * 16 + lane) * 2] = dotp;
      red[(wave * 16 + lane) * 2 + 1] = sqp;
    }
    __syncthreads();
    if (tid < TN && p0 + tid < n_test) {
      const int h = tid >> 4, l = tid & 15;
      const T dot = red[(h * 16 + l) * 2] + red[((h + 2) * 16 + l) * 2];
      const T sq = red[(h * 16 + l) * 2 + 1] + red[((h + 2) * 16 + l) * 2 + 1];
      const int64_t p = b * n_test + p0 + tid;
      const T v = (var + jitter) - sq;          // diag(k_pp - k_px A^-1 k_px^T)
      const T vc = v > T(0) ? v : T(0);         // jnp.clip(..., a_min=0.0); NaN stays NaN
      const T vv = (v != v) ? v : vc;
      if (mean) mean[p] = dot;
      if (var_out) var_out[p] = vv;
      if (sample) sample[p] = xi ? dot + dsqrt<T>(vv) * xi[p] : dot;
    }
    // (the next tile's rows are written right away: their readers, the MFMAs above, are done — every wave has
    // passed the barrier above; `red` is rewritten only after the next tile's first barrier)
  }
}

// largest n_obs the LDS-resident path takes: KP * (KP + pad) + TN * (KP + pad) elements must fit 160 KB
template <typename T> static inline int krig_small_kp(int64_t n_obs) {
  constexpr int TM = KrigSmall<T>::TM;
  const int kp = (int)round_up(n_obs, TM < 32 ? 32 : TM);  // K tiles of 32 (f32) / 16 (f64); kept a multiple of 32
  if (kp > 160) return 0;
  return KrigSmall<T>::lds_bytes(kp, n_obs) <= 150 * 1024 ? kp : 0;
}

template <typename T, int DIM>
static int launch_krig_small(hipStream_t st, int kp, int64_t n_obs, int64_t n_test, int batch, const void *x_obs,
                             const void *x_test, const void *y_obs, const void *var, const void *ls,
                             const void *sigma, double jitter, const void *xi, void *mean, void *sample,
                             void *var_out, int32_t *info) {
  const size_t shm = (size_t)KrigSmall<T>::lds_bytes(kp, n_obs);
#define GPVAE_KS_LAUNCH(KPV)                                                                                          \
  do {                                                                                                                \
    auto kern = krig_small_kernel<T, DIM, KPV>;                                                                       \
    if (shm > 64 * 1024 &&                                                                                            \
        hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm) != hipSuccess) { \
      set_error("gpvae_krig_predict_batched: cannot reserve %zu bytes of LDS", shm);                                  \
      return -1;                                                                                                      \
    }                                                                                                                 \
    hipLaunchKernelGGL(kern, dim3((unsigned)batch), dim3(NTHREADS), shm, st, (int)n_obs, n_test, (const T *)x_obs,    \
                       (const T *)x_test, (const T *)y_obs, (const T *)var, (const T *)ls, (const T *)sigma,          \
                       (T)jitter, (const T *)xi, (T *)mean, (T *)sample, (T *)var_out, info);                         \
  } while (0)
  switch (kp) {
    case 32: GPVAE_KS_LAUNCH(32); break;
    case 64: GPVAE_KS_LAUNCH(64); break;
    case 96: GPVAE_KS_LAUNCH(96); break;
    case 128: GPVAE_KS_LAUNCH(128); break;
    case 160: GPVAE_KS_LAUNCH(160); break;
    default: set_error("gpvae_krig_predict_batched: internal: KP %d", kp); return -1;
  }
#undef GPVAE_KS_LAUNCH
  GPVAE_CHECK_LAUNCH("gpvae_krig_predict_batched (LDS-resident path)");
  return 0;
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

// batches below this go through the big kernels (a single problem has 255 CUs to spare for its panel sweeps)
#define GPVAE_KRIG_SMALL_MIN_BATCH 8
static int g_krig_force_big = 0;  // tests and benchmarks compare the two paths
extern "C" void gpvae_krig_set_path(int force_big) { g_krig_force_big = force_big ? 1 : 0; }

extern "C" int64_t gpvae_krig_batched_workspace_bytes(int dtype, int64_t n_obs, int64_t n_test,
                                                      int batch) {
  const KrigWork w = krig_layout(n_obs, n_test);
  if (batch < 1) batch = 1;
  // the LDS-resident route (one workgroup per small problem) needs no workspace
  if (batch >= GPVAE_KRIG_SMALL_MIN_BATCH && n_test > 0 && !g_krig_force_big &&
      (dtype == GPVAE_F32 ? krig_small_kp<float>(n_obs) : krig_small_kp<double>(n_obs)) > 0)
    return 256;
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
  // small problems in batches (the reference's vmap over posterior draws): one workgroup per problem, all in LDS
  if (batch >= GPVAE_KRIG_SMALL_MIN_BATCH && n_test > 0 && !g_krig_force_big) {
    const int kp = dtype == GPVAE_F32 ? krig_small_kp<float>(n_obs) : krig_small_kp<double>(n_obs);
    if (kp > 0) {
      if (dtype == GPVAE_F32)
        return d == 1 ? launch_krig_small<float, 1>(st, kp, n_obs, n_test, batch, x_obs, x_test, y_obs, var, ls, sigma, jitter, xi, mean, sample, var_out, info)
                      : launch_krig_small<float, 2>(st, kp, n_obs, n_test, batch, x_obs, x_test, y_obs, var, ls, sigma, jitter, xi, mean, sample, var_out, info);
      return d == 1 ? launch_krig_small<double, 1>(st, kp, n_obs, n_test, batch, x_obs, x_test, y_obs, var, ls, sigma, jitter, xi, mean, sample, var_out, info)
                    : launch_krig_small<double, 2>(st, kp, n_obs, n_test, batch, x_obs, x_test, y_obs, var, ls, sigma, jitter, xi, mean, sample, var_out, info);
    }
  }
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
