// gram.hip — kernel-matrix builders.
//
// a1 gram_se: K = var * exp(-0.5 * (||x_i - z_j|| / ls)^2) (+ jitter*I iff square), following
//   /root/reference/src/model/gp.py:74-80 operation by operation (norm -> divide -> square -> exp),
//   fused into ONE pass whose only HBM traffic is the coalesced 16-byte-per-lane store of K.
//   Roofline: HBM write bandwidth; algorithmic bytes = n_x*n_z*sizeof(T) per matrix.
// a2 gram_agg: cell-averaged kernel of gp.py:119-127 (m^2 exp evaluations per entry) — bound by
//   the transcendental issue rate, not by memory.
#include "common.hpp"

namespace gpvae {

constexpr int GRAM_TX = 64;    // lanes along columns, 4 columns each => 256 columns per block
constexpr int GRAM_TY = 4;     // waves per block, one row each per iteration
constexpr int GRAM_ROWS = 32;  // rows per block

// GramMath<T>::expq (common.hpp): exp(-0.5 (sqrt(s) / ls)^2) in the reference's order of operations

template <typename T, int D>
__global__ __launch_bounds__(GRAM_TX *GRAM_TY) void gram_se_kernel(
    int64_t n_x, int64_t n_z, const T *__restrict__ x, const T *__restrict__ z,
    const T *__restrict__ var_p, const T *__restrict__ ls_p, T jitter, int add_jitter,
    T *__restrict__ K, int64_t ldk, int64_t strideK, int vec_ok) {
  const int b = blockIdx.z;
  const T var = var_p[b];
  const T inv_ls = T(1) / ls_p[b];
  T *Kb = K + (int64_t)b * strideK;
  // columns of this thread's four entries.  f32: four consecutive ones (one 16-byte store per lane, a wave writes
  // 1 KB contiguously).  f64: two PAIRS, 128 columns apart, so that each of its two 16-byte stores is contiguous
  // across the wave too — with four consecutive doubles per lane either store left every second 16 bytes of the
  // cache lines it touched to the other one, and the builder ran at 3.0 TB/s whatever its arithmetic cost.
  const int64_t jb = (int64_t)blockIdx.x * GRAM_TX * 4;
  auto colof = [&](int c) -> int64_t {
    if constexpr (sizeof(T) == 4) return jb + 4 * threadIdx.x + c;
    else return jb + (c >> 1) * (2 * GRAM_TX) + 2 * threadIdx.x + (c & 1);
  };
  const int64_t i0 = (int64_t)blockIdx.y * GRAM_ROWS;
  if (colof(0) >= n_z) return;

  T zc[4][D];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int64_t j = (colof(c) < n_z) ? colof(c) : n_z - 1;
#pragma unroll
    for (int e = 0; e < D; ++e) zc[c][e] = z[j * D + e];
  }
  const bool full0 = vec_ok && (sizeof(T) == 4 ? colof(3) < n_z : colof(1) < n_z);
  const bool full1 = vec_ok && colof(3) < n_z;  // f64: the second pair

  for (int r = threadIdx.y; r < GRAM_ROWS; r += GRAM_TY) {
    const int64_t i = i0 + r;
    if (i >= n_x) break;
    T xi[D];
#pragma unroll
    for (int e = 0; e < D; ++e) xi[e] = x[i * D + e];
    T out[4];
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      T s = T(0);
#pragma unroll
      for (int e = 0; e < D; ++e) {
        const T dlt = xi[e] - zc[c][e];
        s += dlt * dlt;
      }
      T k = var * GramMath<T>::expq(s, inv_ls);  // norm -> /ls -> square -> exp   gp.py:74-77
      if (add_jitter && (i == colof(c))) k += jitter;  // k += jitter*eye     gp.py:79-80
      out[c] = k;
    }
    T *row = Kb + i * ldk;
    if constexpr (sizeof(T) == 4) {
      if (full0) {
        // streaming store: the matrix is far larger than L2 and is read next by another launch (4.4 -> 5.9 TB/s)
        typedef float vec4_t __attribute__((ext_vector_type(4)));
        vec4_t v = {out[0], out[1], out[2], out[3]};
        __builtin_nontemporal_store(v, reinterpret_cast<vec4_t *>(row + colof(0)));
      } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
          if (colof(c) < n_z) row[colof(c)] = out[c];
      }
    } else {
      typedef double vec2_t __attribute__((ext_vector_type(2)));
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        if (h == 0 ? full0 : full1) {
          vec2_t v = {out[2 * h], out[2 * h + 1]};
          __builtin_nontemporal_store(v, reinterpret_cast<vec2_t *>(row + colof(2 * h)));
        } else {
#pragma unroll
          for (int c = 2 * h; c < 2 * h + 2; ++c)
            if (colof(c) < n_z) row[colof(c)] = out[c];
        }
      }
    }
  }
}

template <typename T>
static int launch_gram_se(hipStream_t st, int d, int64_t n_x, int64_t n_z, const void *x,
                          const void *z, int batch, const void *var, const void *ls, double jitter,
                          void *K, int64_t ldk, int64_t strideK) {
  dim3 block(GRAM_TX, GRAM_TY);
  dim3 grid((unsigned)cdiv(n_z, GRAM_TX * 4), (unsigned)cdiv(n_x, GRAM_ROWS), (unsigned)batch);
  const int vec_ok = (ldk % 4 == 0) && (strideK % 4 == 0) &&
                     ((reinterpret_cast<uintptr_t>(K) % (4 * sizeof(T))) == 0);
  const int add_jitter = (n_x == n_z);
  if (d == 1)
    hipLaunchKernelGGL((gram_se_kernel<T, 1>), grid, block, 0, st, n_x, n_z, (const T *)x,
                       (const T *)z, (const T *)var, (const T *)ls, (T)jitter, add_jitter, (T *)K,
                       ldk, strideK, vec_ok);
  else
    hipLaunchKernelGGL((gram_se_kernel<T, 2>), grid, block, 0, st, n_x, n_z, (const T *)x,
                       (const T *)z, (const T *)var, (const T *)ls, (T)jitter, add_jitter, (T *)K,
                       ldk, strideK, vec_ok);
  GPVAE_CHECK_LAUNCH("gpvae_gram_se");
  return 0;
}

// ---- a2: aggregated kernel ------------------------------------------------------------------------
// One block per 16x16 tile of cells; the 16*m sample points of the row cells and of the column
// cells are staged in LDS; each thread owns one (i,j) entry and runs the m*m double sum.
constexpr int AGG_T = 16;

template <typename T, int D>
__global__ __launch_bounds__(AGG_T *AGG_T) void gram_agg_kernel(
    int64_t n_cells, int m, const T *__restrict__ pts, const T *__restrict__ var_p,
    const T *__restrict__ ls_p, T jitter, T *__restrict__ K, int64_t ldk) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *prow = reinterpret_cast<T *>(smem_raw);      // [AGG_T][m][D]
  T *pcol = prow + (size_t)AGG_T * m * D;         // [AGG_T][m][D]
  const T var = var_p[0], inv_ls = T(1) / ls_p[0];
  const int64_t i0 = (int64_t)blockIdx.y * AGG_T, j0 = (int64_t)blockIdx.x * AGG_T;
  const int tid = threadIdx.y * AGG_T + threadIdx.x;
  const int per = AGG_T * m * D;
  for (int t = tid; t < per; t += AGG_T * AGG_T) {
    const int64_t ci = i0 + t / (m * D), cj = j0 + t / (m * D);
    const int rem = t % (m * D);
    prow[t] = (ci < n_cells) ? pts[ci * m * D + rem] : T(0);
    pcol[t] = (cj < n_cells) ? pts[cj * m * D + rem] : T(0);
  }
  __syncthreads();
  const int64_t i = i0 + threadIdx.y, j = j0 + threadIdx.x;
  if (i >= n_cells || j >= n_cells) return;
  const T *pi = prow + (size_t)threadIdx.y * m * D;
  const T *pj = pcol + (size_t)threadIdx.x * m * D;
  T acc = T(0);
  for (int a = 0; a < m; ++a) {
    T xa[D];
#pragma unroll
    for (int e = 0; e < D; ++e) xa[e] = pi[a * D + e];
    for (int bb = 0; bb < m; ++bb) {
      T s = T(0);
#pragma unroll
      for (int e = 0; e < D; ++e) {
        const T dlt = xa[e] - pj[bb * D + e];
        s += dlt * dlt;
      }
      acc += var * GramMath<T>::expq(s, inv_ls);  // the arithmetic of gram_se (a1), entry by entry
    }
  }
  // every m x m block is square, so the literal code adds jitter to its m diagonal entries:
  // sum gains m*jitter, and /m^2 leaves jitter/m on EVERY aggregated entry (gp.py:119-127).
  acc += (T)m * jitter;
  K[i * ldk + j] = acc / ((T)m * (T)m);
}

template <typename T>
static int launch_gram_agg(hipStream_t st, int d, int64_t n_cells, int m, const void *pts,
                           const void *var, const void *ls, double jitter, void *K, int64_t ldk) {
  dim3 block(AGG_T, AGG_T);
  dim3 grid((unsigned)cdiv(n_cells, AGG_T), (unsigned)cdiv(n_cells, AGG_T));
  const size_t shm = 2 * (size_t)AGG_T * m * d * sizeof(T);
  if (d == 1)
    hipLaunchKernelGGL((gram_agg_kernel<T, 1>), grid, block, shm, st, n_cells, m, (const T *)pts,
                       (const T *)var, (const T *)ls, (T)jitter, (T *)K, ldk);
  else
    hipLaunchKernelGGL((gram_agg_kernel<T, 2>), grid, block, shm, st, n_cells, m, (const T *)pts,
                       (const T *)var, (const T *)ls, (T)jitter, (T *)K, ldk);
  GPVAE_CHECK_LAUNCH("gpvae_gram_agg");
  return 0;
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int gpvae_gram_se(void *stream, int dtype, int d, int64_t n_x, int64_t n_z,
                             const void *x, const void *z, int batch, const void *var,
                             const void *ls, double jitter, void *K, int64_t ldk,
                             int64_t strideK) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_gram_se: bad dtype %d", dtype);
  GPVAE_REQUIRE(d == 1 || d == 2, 3, "gpvae_gram_se: d must be 1 or 2 (got %d)", d);
  GPVAE_REQUIRE(n_x >= 0, 4, "gpvae_gram_se: n_x < 0");
  GPVAE_REQUIRE(n_z >= 0, 5, "gpvae_gram_se: n_z < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 8, "gpvae_gram_se: batch out of range");
  if (n_x == 0 || n_z == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(x, 6, "gpvae_gram_se: x is NULL");
  GPVAE_REQUIRE(z, 7, "gpvae_gram_se: z is NULL");
  GPVAE_REQUIRE(var, 9, "gpvae_gram_se: var is NULL");
  GPVAE_REQUIRE(ls, 10, "gpvae_gram_se: ls is NULL");
  GPVAE_REQUIRE(K, 12, "gpvae_gram_se: K is NULL");
  GPVAE_REQUIRE(ldk >= n_z, 13, "gpvae_gram_se: ldk < n_z");
  GPVAE_REQUIRE(batch == 1 || strideK >= n_x * ldk, 14, "gpvae_gram_se: strideK too small");
  GPVAE_REQUIRE(cdiv(n_x, GRAM_ROWS) <= 65535, 4, "gpvae_gram_se: n_x too large");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? launch_gram_se<float>(st, d, n_x, n_z, x, z, batch, var, ls, jitter, K, ldk, strideK)
             : launch_gram_se<double>(st, d, n_x, n_z, x, z, batch, var, ls, jitter, K, ldk,
                                      strideK);
}

extern "C" int gpvae_gram_agg(void *stream, int dtype, int d, int64_t n_cells, int m,
                              const void *pts, const void *var, const void *ls, double jitter,
                              void *K, int64_t ldk) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_gram_agg: bad dtype");
  GPVAE_REQUIRE(d == 1 || d == 2, 3, "gpvae_gram_agg: d must be 1 or 2");
  GPVAE_REQUIRE(n_cells >= 0, 4, "gpvae_gram_agg: n_cells < 0");
  GPVAE_REQUIRE(m >= 1, 5, "gpvae_gram_agg: m < 1");
  if (n_cells == 0) return 0;
  const size_t es = dtype == GPVAE_F32 ? 4 : 8;
  GPVAE_REQUIRE(2 * (size_t)AGG_T * m * d * es <= 64 * 1024, 5,
                "gpvae_gram_agg: m too large for the LDS staging (m*d <= %d)",
                (int)(64 * 1024 / (2 * AGG_T * es)));
  GPVAE_REQUIRE(pts && var && ls && K, 6, "gpvae_gram_agg: NULL pointer");
  GPVAE_REQUIRE(ldk >= n_cells, 11, "gpvae_gram_agg: ldk < n_cells");
  GPVAE_REQUIRE(cdiv(n_cells, AGG_T) <= 65535, 4, "gpvae_gram_agg: n_cells too large");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? launch_gram_agg<float>(st, d, n_cells, m, pts, var, ls, jitter, K, ldk)
             : launch_gram_agg<double>(st, d, n_cells, m, pts, var, ls, jitter, K, ldk);
}
