// draw.hip — MVN draws from a Cholesky factor, and counter-based normals.
//
// gpvae_draw replaces  f = L eps ; y = f + sigma eps'  of GP.sample under Predictive
// (/root/reference/src/model/gp.py:190-194, vae.py:160-162,223-226) for nb draws that share L:
//   F[r][i] = sum_{j<=i} E[r][j] L[i][j]          (a triangular matrix product, n^2*nb flop)
// as a 128x128-tiled f32/f64 MFMA product whose k loop stops at the diagonal of L; the strict upper
// part of L is masked in the loader, so it may hold anything.  Column tiles are issued heaviest
// first (their k range grows with the column index).  Bound: MFMA.
#include "gemm_core.hpp"
#include "philox.hpp"

namespace gpvae {

constexpr int DBM = 128;

// VECM: L and E rows are 16-byte aligned (base and leading dimension) -> 16-byte loads.
template <typename T, bool VECM>
__global__ __launch_bounds__(NTHREADS, 2) void draw_kernel(
    int64_t n, int64_t nb, const T *L, int64_t ldl, const T *E, int64_t lde, const T *E2,
    int64_t lde2, const T *sigma_p, T *F, int64_t ldf, T *Y, int64_t ldy, int nrt, int nct) {
  using M = Mfma<T>;
  constexpr int DBK = TileK<T>::BK;
  constexpr int FR = 64 / M::TM;
  constexpr int LD = DBK + RmPad<T>::V;  // row-major images [128][LD]
  using LA = LoaderKC<T, DBM, DBK, VECM>;
  using LB = LoaderKC<T, DBM, DBK, VECM, true>;
  using Pipe = PipeKC<T, FR, FR, DBK, DBM, DBM, LA, LB>;
  static_assert(Pipe::LD == LD, "image stride");
  __shared__ __align__(16) T lds[Pipe::LDS_ELEMS];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // Plain round-robin placement already puts blocks b, b+8, ... (same column tile of L, row tiles
  // 8 apart) on one XCD; a contiguous-run remap (guide T1) was measured 1.8x SLOWER here because it
  // hands whole heavy column tiles to one XCD (the k range grows with the column index).
  const int ct = nct - 1 - (int)(blockIdx.x / nrt);  // heaviest column tiles first
  const int rt = (int)(blockIdx.x % nrt);
  const int64_t r0 = (int64_t)rt * DBM, c0 = (int64_t)ct * DBM;
  const int64_t kend = (c0 + DBM < n) ? c0 + DBM : n;

  LA la;
  LB lb;
  la.init([&](int s, int64_t &rg) -> const T * { rg = r0 + s; return (r0 + s < nb) ? E + (r0 + s) * lde : nullptr; }, E, tid);
  lb.init([&](int s, int64_t &rg) -> const T * { rg = c0 + s; return (c0 + s < n) ? L + (c0 + s) * ldl : nullptr; }, L, tid);

  typename M::acc_t acc[FR][FR];
  zero_acc<T, FR, FR>(acc);
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  Pipe pipe;
  pipe.run(la, lb, lds, 0, kend, wm, wn, lane, tid, acc);
  const T sigma = (E2 != nullptr) ? sigma_p[0] : T(0);
  // whole tile inside the matrices: buffer ops with scalar element offsets, no bounds checks
  const bool interior = r0 + DBM <= nb && c0 + DBM <= n && TileIO<T>::usable(ldf, 64) &&
                        TileIO<T>::usable(ldy, 64) && TileIO<T>::usable(lde2, 64);
  if (interior) {
    TileIO<T> iof, ioy, ioe;
    const int64_t rw = r0 + wm, cw = c0 + wn;
    if (F != nullptr) iof.init(F + rw * ldf + cw, ldf, lane);
    if (Y != nullptr) ioy.init(Y + rw * ldy + cw, ldy, lane);
    if (E2 != nullptr) ioe.init(E2 + rw * lde2 + cw, lde2, lane);
#pragma unroll
    for (int fm = 0; fm < FR; ++fm) {
      T e2[FR][M::NACC];
      if (E2 != nullptr) {
#pragma unroll
        for (int fn = 0; fn < FR; ++fn)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) e2[fn][r] = ioe.load(fm, fn, r);
      }
#pragma unroll
      for (int fn = 0; fn < FR; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) {
          const T f = acc[fm][fn][r];
          if (F != nullptr) iof.store(fm, fn, r, f);
          if (Y != nullptr) ioy.store(fm, fn, r, (E2 != nullptr) ? f + sigma * e2[fn][r] : f);
        }
    }
    return;
  }
#pragma unroll
  for (int fm = 0; fm < FR; ++fm) {
    T e2[M::NACC][FR];
    if (E2 != nullptr) {  // all loads of this half first (clamped address), then the stores
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int64_t row = r0 + wm + fm * M::TM + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < FR; ++fn) {
          const int64_t c = c0 + wn + fn * M::TM + M::col(lane);
          e2[r][fn] = E2[(row < nb && c < n) ? row * lde2 + c : 0];
        }
      }
    }
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) {
      const int64_t row = r0 + wm + fm * M::TM + M::row(lane, r);
#pragma unroll
      for (int fn = 0; fn < FR; ++fn) {
        const int64_t c = c0 + wn + fn * M::TM + M::col(lane);
        if (row < nb && c < n) {
          const T f = acc[fm][fn][r];
          if (F != nullptr) F[row * ldf + c] = f;
          if (Y != nullptr) Y[row * ldy + c] = (E2 != nullptr) ? f + sigma * e2[r][fn] : f;
        }
      }
    }
  }
}


// ---- one draw per factor: the Predictive(gp.sample) call with hyper-parameters left None ------------
// (vae.py:226, toy_examples.py:88, gp.py:171-176: every sample has its own K, L, sigma.)
// f_b = L_b e_b is a triangular matrix-vector product per sample: n^2/2 * s bytes of L read once
// => HBM-bound.  One workgroup per (64-row block, sample): e_b staged in LDS, one wave per row at a
// time, lanes along the row (coalesced 256-byte reads of L), wave reduction.
template <typename T>
__global__ __launch_bounds__(NTHREADS) void draw_batched_kernel(
    int64_t n, const T *L, int64_t ldl, int64_t strideL, const T *E, int64_t lde, const T *E2,
    int64_t lde2, const T *sigma, T *F, int64_t ldf, T *Y, int64_t ldy) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *e = reinterpret_cast<T *>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t b = blockIdx.y;
  const int64_t i0 = (int64_t)blockIdx.x * 64;
  const int64_t iend = (i0 + 64 < n) ? i0 + 64 : n;
  const T *Lb = L + b * strideL;
  const T *Eb = E + b * lde;
  for (int64_t j = tid; j < iend; j += NTHREADS) e[j] = Eb[j];
  __syncthreads();
  const T sg = (E2 != nullptr) ? sigma[b] : T(0);
  for (int64_t i = i0 + wave; i < iend; i += NTHREADS / 64) {
    const T *row = Lb + i * ldl;
    T a[4] = {T(0), T(0), T(0), T(0)};
    int u = 0;
    for (int64_t j = lane; j <= i; j += 64, u = (u + 1) & 3) a[u] += row[j] * e[j];
    const T f = wave_sum<T>((a[0] + a[1]) + (a[2] + a[3]));
    if (lane == 0) {
      if (F != nullptr) F[b * ldf + i] = f;
      if (Y != nullptr) Y[b * ldy + i] = (E2 != nullptr) ? f + sg * E2[b * lde2 + i] : f;
    }
  }
}


// ---- f3: hyper-prior samplers on the device ---------------------------------------------------------
// (ls, var, sigma) of GP.sample / epoch_train (gp.py:171-176, vae.py:157-159) without a host round
// trip: var ~ LogNormal(0, s_var), sigma ~ HalfNormal(s_sig), ls ~ InverseGamma(a, b) = b / Gamma(a)
// or Beta(a, b) = Ga / (Ga + Gb) (the README's prior).  Gamma by Marsaglia & Tsang (2000) with the
// a < 1 boost Gamma(a) = Gamma(a + 1) U^(1/a); every random word is Philox4x32-10 of a counter that
// names (sample index, purpose, attempt), so sample i does not depend on how calls are batched.
// All arithmetic in double (results are rounded to T at the end): the oracle restates it to ~1 ulp.
struct HyperRng {
  uint64_t seed, stream_id;
  uint32_t idx;
  // words of counter (idx, purpose << 24 | attempt, stream)
  __device__ void words(uint32_t purpose, uint32_t attempt, uint32_t (&c)[4]) const {
    c[0] = idx; c[1] = (purpose << 24) | attempt; c[2] = (uint32_t)stream_id; c[3] = (uint32_t)(stream_id >> 32);
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
  }
  __device__ static double unif(uint32_t w) { return ((double)w + 0.5) * (1.0 / 4294967296.0); }
  // (normal, normal) from words 0, 1 (Box-Muller, as philox_normals4<double>)
  __device__ static void normals(const uint32_t (&c)[4], double &z0, double &z1) {
    const double rad = sqrt(-2.0 * log(unif(c[0])));
    double sn, cs;
    sincospi(2.0 * unif(c[1]), &sn, &cs);
    z0 = rad * cs;
    z1 = rad * sn;
  }
  __device__ double gamma(double a, uint32_t purpose) const {
    const double a1 = a < 1.0 ? a + 1.0 : a;
    const double d = a1 - 1.0 / 3.0, cc = 1.0 / sqrt(9.0 * d);
    double g = d;  // (64 rejections in a row have probability < 1e-80: the mean stands in)
    for (uint32_t k = 0; k < 64; ++k) {
      uint32_t c[4];
      words(purpose, k, c);
      double x, unused;
      normals(c, x, unused);
      const double t = 1.0 + cc * x, v = t * t * t;
      if (v > 0.0 && log(unif(c[2])) < 0.5 * x * x + d - d * v + d * log(v)) { g = d * v; break; }
    }
    if (a < 1.0) {
      uint32_t c[4];
      words(purpose + 1, 0, c);
      g *= pow(unif(c[0]), 1.0 / a);
    }
    return g;
  }
};

template <typename T>
__global__ void hyper_sample_kernel(int64_t count, uint64_t seed, uint64_t stream_id, uint32_t first,
                                    int ls_prior, double ls_a, double ls_b, double s_var, double s_sig,
                                    T *ls, T *var, T *sigma) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  HyperRng r{seed, stream_id, first + (uint32_t)i};
  uint32_t c[4];
  r.words(0, 0, c);
  double z0, z1;
  HyperRng::normals(c, z0, z1);
  if (var) var[i] = (T)exp(s_var * z0);
  if (sigma) sigma[i] = (T)fabs(s_sig * z1);
  if (ls) {
    double v;
    if (ls_prior == GPVAE_PRIOR_INVGAMMA) {
      v = ls_b / r.gamma(ls_a, 1);
    } else {
      const double ga = r.gamma(ls_a, 1), gb = r.gamma(ls_b, 3);
      v = ga / (ga + gb);
    }
    ls[i] = (T)v;
  }
}

// ---- f4: the LGCP observation of PoiGP.sample (gp.py:249-250): rate = exp(f), y ~ Poisson(rate) -------------
// One Philox counter per (element, attempt), as above.  rate < 10: inversion by sequential search on one uniform;
// rate >= 10: Hoermann's transformed rejection (PTRS, 1993 — the method NumPy and JAX use), two uniforms per
// attempt.  Arithmetic in double on the rate rounded to T.  y is int32 (JAX's default integer): -1 for a rate
// that is NaN or negative, 2^31 - 1 for rates beyond 2^31 - 1 (exp(f) overflowing included).
__device__ inline int32_t poisson_draw(double lam, const HyperRng &r) {
  if (!(lam >= 0.0)) return -1;
  if (lam == 0.0) return 0;
  if (lam > 2147483647.0) return 2147483647;
  uint32_t c[4];
  if (lam < 10.0) {
    r.words(7, 0, c);
    const double u = HyperRng::unif(c[0]);
    double p = exp(-lam), s = p;
    int k = 0;
    while (u > s && k < 200) {
      ++k;
      p *= lam / (double)k;
      s += p;
    }
    return k;
  }
  const double slam = sqrt(lam), loglam = log(lam);
  const double b = 0.931 + 2.53 * slam, a = -0.059 + 0.02483 * b;
  const double inv_alpha = 1.1239 + 1.1328 / (b - 3.4), vr = 0.9277 - 3.6224 / (b - 2.0);
  for (uint32_t att = 0; att < 64; ++att) {
    r.words(7, att, c);
    const double U = HyperRng::unif(c[0]) - 0.5, V = HyperRng::unif(c[1]);
    const double us = 0.5 - fabs(U);
    const double kf = floor((2.0 * a / us + b) * U + lam + 0.43);
    if (us >= 0.07 && V <= vr) return (int32_t)fmin(kf, 2147483647.0);
    if (kf < 0.0 || (us < 0.013 && V > us)) continue;
    if (log(V) + log(inv_alpha) - log(a / (us * us) + b) <= -lam + kf * loglam - lgamma(kf + 1.0))
      return (int32_t)fmin(kf, 2147483647.0);
  }
  return (int32_t)fmin(floor(lam), 2147483647.0);  // 64 rejections in a row: probability < 1e-30
}

template <typename T>
__global__ void poisson_exp_kernel(int64_t count, uint64_t seed, uint64_t stream_id, uint32_t first, const T *f,
                                   T *rate, int32_t *y) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const T rt = (T)exp((double)f[i]);  // correctly rounded to T but for double's own last bit
  if (rate) rate[i] = rt;
  if (y) y[i] = poisson_draw((double)rt, HyperRng{seed, stream_id, first + (uint32_t)i});
}

template <typename T>
__global__ void randn_kernel(int64_t count, uint64_t seed, uint64_t stream_id, T *out) {
  const int64_t nq = (count + 3) / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
       q += (int64_t)gridDim.x * blockDim.x) {
    T z[4];
    philox_normals4<T>(q, seed, stream_id, z);
    const int64_t i = q * 4;
    if (i + 3 < count && ((reinterpret_cast<uintptr_t>(out + i) % (4 * sizeof(T))) == 0)) {
      typename Vec4<T>::type v;
      v.x = z[0]; v.y = z[1]; v.z = z[2]; v.w = z[3];
      *reinterpret_cast<typename Vec4<T>::type *>(out + i) = v;
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (i + e < count) out[i + e] = z[e];
    }
  }
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int gpvae_draw(void *stream, int dtype, int64_t n, int64_t nb, const void *L,
                          int64_t ldl, const void *E, int64_t lde, const void *E2, int64_t lde2,
                          const void *sigma, void *F, int64_t ldf, void *Y, int64_t ldy) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_draw: bad dtype %d", dtype);
  GPVAE_REQUIRE(n >= 0, 3, "gpvae_draw: n < 0");
  GPVAE_REQUIRE(nb >= 0, 4, "gpvae_draw: nb < 0");
  if (n == 0 || nb == 0) return 0;
  GPVAE_REQUIRE(L, 5, "gpvae_draw: L is NULL");
  GPVAE_REQUIRE(ldl >= n, 6, "gpvae_draw: ldl < n");
  GPVAE_REQUIRE(E, 7, "gpvae_draw: E is NULL");
  GPVAE_REQUIRE(lde >= n, 8, "gpvae_draw: lde < n");
  if (E2) {
    GPVAE_REQUIRE(lde2 >= n, 10, "gpvae_draw: lde2 < n");
    GPVAE_REQUIRE(sigma, 11, "gpvae_draw: sigma is NULL with E2 given");
  }
  GPVAE_REQUIRE(F || Y, 14, "gpvae_draw: both F and Y are NULL");
  if (F) GPVAE_REQUIRE(ldf >= n, 13, "gpvae_draw: ldf < n");
  if (Y) GPVAE_REQUIRE(ldy >= n, 15, "gpvae_draw: ldy < n");
  const int64_t nrt = cdiv(nb, DBM), nct = cdiv(n, DBM);
  GPVAE_REQUIRE(nrt * nct < (int64_t)1 << 31, 4, "gpvae_draw: too many tiles");
  hipStream_t st = (hipStream_t)stream;
  auto launch = [&](auto tag, auto vecm) {
    using T = decltype(tag);
    hipLaunchKernelGGL((draw_kernel<T, decltype(vecm)::value>), dim3((unsigned)(nrt * nct)),
                       dim3(NTHREADS), 0, st, n, nb, (const T *)L, ldl, (const T *)E, lde,
                       (const T *)E2, lde2, (const T *)sigma, (T *)F, ldf, (T *)Y, ldy, (int)nrt,
                       (int)nct);
  };
  if (dtype == GPVAE_F32) {
    if (kc_vec_ok<float>(E, lde) && kc_vec_ok<float>(L, ldl)) launch(float{}, std::true_type{});
    else launch(float{}, std::false_type{});
  } else {
    if (kc_vec_ok<double>(E, lde) && kc_vec_ok<double>(L, ldl)) launch(double{}, std::true_type{});
    else launch(double{}, std::false_type{});
  }
  GPVAE_CHECK_LAUNCH("gpvae_draw");
  return 0;
}

extern "C" int gpvae_draw_batched(void *stream, int dtype, int64_t n, int64_t nb, const void *L,
                                  int64_t ldl, int64_t strideL, const void *E, int64_t lde,
                                  const void *E2, int64_t lde2, const void *sigma, void *F,
                                  int64_t ldf, void *Y, int64_t ldy) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_draw_batched: bad dtype %d", dtype);
  GPVAE_REQUIRE(n >= 0, 3, "gpvae_draw_batched: n < 0");
  GPVAE_REQUIRE(nb >= 0 && nb <= 65535, 4, "gpvae_draw_batched: nb out of range [0,65535]");
  if (n == 0 || nb == 0) return 0;
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  GPVAE_REQUIRE(n * es <= 64 * 1024, 3, "gpvae_draw_batched: n too large (one row of normals must fit 64 KB of LDS)");
  GPVAE_REQUIRE(L, 5, "gpvae_draw_batched: L is NULL");
  GPVAE_REQUIRE(ldl >= n, 6, "gpvae_draw_batched: ldl < n");
  GPVAE_REQUIRE(nb == 1 || strideL >= n * ldl, 7, "gpvae_draw_batched: strideL < n*ldl");
  GPVAE_REQUIRE(E, 8, "gpvae_draw_batched: E is NULL");
  GPVAE_REQUIRE(lde >= n, 9, "gpvae_draw_batched: lde < n");
  if (E2) {
    GPVAE_REQUIRE(lde2 >= n, 11, "gpvae_draw_batched: lde2 < n");
    GPVAE_REQUIRE(sigma, 12, "gpvae_draw_batched: sigma is NULL with E2 given");
  }
  GPVAE_REQUIRE(F || Y, 15, "gpvae_draw_batched: both F and Y are NULL");
  if (F) GPVAE_REQUIRE(ldf >= n, 14, "gpvae_draw_batched: ldf < n");
  if (Y) GPVAE_REQUIRE(ldy >= n, 16, "gpvae_draw_batched: ldy < n");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(n, 64), (unsigned)nb);
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((draw_batched_kernel<float>), grid, dim3(NTHREADS), (size_t)(n * es), st, n,
                       (const float *)L, ldl, strideL, (const float *)E, lde, (const float *)E2, lde2,
                       (const float *)sigma, (float *)F, ldf, (float *)Y, ldy);
  else
    hipLaunchKernelGGL((draw_batched_kernel<double>), grid, dim3(NTHREADS), (size_t)(n * es), st, n,
                       (const double *)L, ldl, strideL, (const double *)E, lde, (const double *)E2, lde2,
                       (const double *)sigma, (double *)F, ldf, (double *)Y, ldy);
  GPVAE_CHECK_LAUNCH("gpvae_draw_batched");
  return 0;
}

extern "C" int gpvae_randn(void *stream, int dtype, int64_t count, uint64_t seed,
                           uint64_t stream_id, void *out) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_randn: bad dtype %d", dtype);
  GPVAE_REQUIRE(count >= 0, 3, "gpvae_randn: count < 0");
  if (count == 0) return 0;
  GPVAE_REQUIRE(out, 6, "gpvae_randn: out is NULL");
  const int64_t nq = cdiv(count, 4);
  int64_t blocks = cdiv(nq, 256);
  if (blocks > 4096) blocks = 4096;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((randn_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, count, seed,
                       stream_id, (float *)out);
  else
    hipLaunchKernelGGL((randn_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, count,
                       seed, stream_id, (double *)out);
  GPVAE_CHECK_LAUNCH("gpvae_randn");
  return 0;
}

extern "C" int gpvae_poisson_exp(void *stream, int dtype, int64_t count, uint64_t seed, uint64_t stream_id,
                                 int64_t first_index, const void *f, void *rate, int32_t *y) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_poisson_exp: bad dtype %d", dtype);
  GPVAE_REQUIRE(count >= 0, 3, "gpvae_poisson_exp: count < 0");
  GPVAE_REQUIRE(first_index >= 0 && first_index + count <= 0xffffffffLL, 6, "gpvae_poisson_exp: element indices must fit 32 bits");
  if (count == 0) return 0;
  GPVAE_REQUIRE(f, 7, "gpvae_poisson_exp: f is NULL");
  GPVAE_REQUIRE(rate || y, 9, "gpvae_poisson_exp: both rate and y are NULL");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(count, 256));
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((poisson_exp_kernel<float>), grid, dim3(256), 0, st, count, seed, stream_id, (uint32_t)first_index,
                       (const float *)f, (float *)rate, y);
  else
    hipLaunchKernelGGL((poisson_exp_kernel<double>), grid, dim3(256), 0, st, count, seed, stream_id, (uint32_t)first_index,
                       (const double *)f, (double *)rate, y);
  GPVAE_CHECK_LAUNCH("gpvae_poisson_exp");
  return 0;
}

extern "C" int gpvae_hyper_sample(void *stream, int dtype, int64_t count, uint64_t seed, uint64_t stream_id,
                                  int64_t first_index, int ls_prior, double ls_a, double ls_b,
                                  double var_scale, double sigma_scale, void *ls, void *var, void *sigma) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_hyper_sample: bad dtype %d", dtype);
  GPVAE_REQUIRE(count >= 0, 3, "gpvae_hyper_sample: count < 0");
  GPVAE_REQUIRE(first_index >= 0 && first_index + count <= 0xffffffffLL, 6, "gpvae_hyper_sample: sample indices must fit 32 bits");
  GPVAE_REQUIRE(ls_prior == GPVAE_PRIOR_INVGAMMA || ls_prior == GPVAE_PRIOR_BETA, 7, "gpvae_hyper_sample: unknown lengthscale prior %d", ls_prior);
  GPVAE_REQUIRE(ls_a > 0 && ls_b > 0, 8, "gpvae_hyper_sample: prior parameters must be > 0");
  GPVAE_REQUIRE(var_scale >= 0 && sigma_scale >= 0, 10, "gpvae_hyper_sample: scales must be >= 0");
  if (count == 0) return 0;
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)cdiv(count, 64));
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((hyper_sample_kernel<float>), grid, dim3(64), 0, st, count, seed, stream_id, (uint32_t)first_index,
                       ls_prior, ls_a, ls_b, var_scale, sigma_scale, (float *)ls, (float *)var, (float *)sigma);
  else
    hipLaunchKernelGGL((hyper_sample_kernel<double>), grid, dim3(64), 0, st, count, seed, stream_id, (uint32_t)first_index,
                       ls_prior, ls_a, ls_b, var_scale, sigma_scale, (double *)ls, (double *)var, (double *)sigma);
  GPVAE_CHECK_LAUNCH("gpvae_hyper_sample");
  return 0;
}
