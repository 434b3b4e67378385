// gemm_tile.hpp — the generic tile product behind the VAE's layers (vae.hip) and the stand-alone
// triangular solve (trsm.hip): C (M x N) = epilogue(A (M x K) B (K x N)), 128 x 64 or 64 x 64 tiles,
// optional split over K, operands K-contiguous (LoaderKC) or M-contiguous (LoaderMC).
#pragma once
#include "gemm_core.hpp"

namespace gpvae {

// ---- generic tile product C (M x N) = A (M x K) B (K x N), 128 x 64 tiles, split K -------------------
enum { EPI_STORE = 0, EPI_ATOMIC = 1, EPI_BIAS = 2, EPI_DF = 3, EPI_FUSED = 4 };

template <typename T> struct GemmArgs {
  const T *A;
  int64_t lda;
  const T *B;
  int64_t ldb;
  T *C;
  int64_t ldc;
  int64_t M, N, K;
  int64_t kchunk;       // k range per blockIdx.z
  int64_t a_ones_row = -1;  // AKC == false: A row index that reads as ones (-1: none)
  int64_t b_ones_row = -1;  // BKC == false: B column index that reads as ones (-1: none)
  const T *bias;        // EPI_BIAS / EPI_DF: per column
  const T *Yref;        // EPI_DF: data, ld = ldy
  int64_t ldy;
  T inv_var;            // EPI_DF: 1/obs_sigma^2
  T *loss;              // (unused: the loss parts go to loss_part)
  int vec_a, vec_b;
  int trans_c;          // store C transposed: C[col * ldc + row] (EPI_STORE / EPI_ATOMIC)
  // EPI_FUSED (the dense layers of the wide-MLP path, the solve's updates):
  // C = mask(relu([C +/-] A B [+ bias])); mask may alias C (each element is read before it is
  // written, by the same lane)
  int addc, relu;
  int negate;           // C - A B (EPI_FUSED with addc), -A B (EPI_ATOMIC)
  const T *mask;        // keep where mask > 0, else 0 (nullptr: keep all)
  int64_t ldm;
  // EPI_ATOMIC (the name is historical: no atomics are left): k-slice z STORES its partial tile at
  // C[z * slab + index]; the caller sums the slices in a fixed order (slab_sum_kernel, or the consumer
  // itself) — the same bits on every run.  slab must be set.
  int64_t slab = 0;
  T *loss_part = nullptr;  // EPI_DF: workgroup w stores its part of the loss at loss_part[w]; must be set
};

constexpr int GBM = 128, GBN = 64;
template <typename T> struct GemmK { static constexpr int BK = TileK<T>::BK; };

// KVEC: the K-contiguous operands (AKC / BKC) are 16-byte aligned row by row
// BM: rows per workgroup tile, 128 (4 waves stacked, each 32 x 64) or 64 (2 x 2 waves, each 32 x 32) —
// the latter for outputs of at most 64 rows ([dV3; dc3] is 51 x n: on 128-row tiles 60 % of its MFMA
// time multiplied padding, and that time was what bound the launch).
template <typename T, bool AKC, bool BKC, int EPI, bool KVEC, int BM = 128>
__global__ __launch_bounds__(NTHREADS) void gemm_kernel(GemmArgs<T> g) {
  using M = Mfma<T>;
  constexpr int GBK = GemmK<T>::BK;
  constexpr int GBM = BM;             // shadows the namespace constant below this line
  constexpr int WN = BM == 128 ? 1 : 2;  // waves along N
  constexpr int FM = 32 / M::TM;         // wave tile 32 x (64 / WN)
  constexpr int FN = 64 / WN / M::TM;
  // row-major image [rows][GBK + pad] for a K-contiguous operand, k-major [GBK][rows + pad] otherwise
  constexpr int LDA = AKC ? GBK + RmPad<T>::V : GBM + LdsPad<T>::V;
  constexpr int LDB = BKC ? GBK + RmPad<T>::V : GBN + LdsPad<T>::V;
  __shared__ __align__(16) T As[AKC ? GBM * LDA : GBK * LDA];
  __shared__ __align__(16) T Bs[BKC ? GBN * LDB : GBK * LDB];
  __shared__ T red[4];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = (wave / WN) * 32, wn = (wave % WN) * (64 / WN);  // the wave's corner inside the tile
  const int64_t m0 = (int64_t)blockIdx.x * GBM, n0 = (int64_t)blockIdx.y * GBN;
  const int64_t kbeg = (int64_t)blockIdx.z * g.kchunk;
  const int64_t kend = (kbeg + g.kchunk < g.K) ? kbeg + g.kchunk : g.K;
  const int64_t Mrows = g.M;  // includes the ones row when present
  const int64_t a_rend = (g.a_ones_row >= 0) ? g.a_ones_row : Mrows;

  LoaderKC<T, GBM, GBK, KVEC> lak;
  LoaderMC<T, GBM, GBK, true> lam;
  LoaderKC<T, GBN, GBK, KVEC> lbk;
  LoaderMC<T, GBN, GBK, true> lbm;
  const int64_t b_rend = (g.b_ones_row >= 0) ? g.b_ones_row : g.N;
  int64_t kfast = INT64_MAX;
  if constexpr (!AKC) {
    lam.init_fast(g.A, g.lda, m0, a_rend, g.K, tid, g.a_ones_row);
    if (lam.fast_end(kend) < kfast) kfast = lam.fast_end(kend);
  }
  if constexpr (!BKC) {
    lbm.init_fast(g.B, g.ldb, n0, b_rend, g.K, tid, g.b_ones_row);
    if (lbm.fast_end(kend) < kfast) kfast = lbm.fast_end(kend);
  }
  if constexpr (AKC) {
    // ragged last k-tile on the fast path: B (k-rows past K read as zero through its buffer range
    // check) supplies the zeros, A's rows are contiguous in one array (finite values behind a row)
    const bool kr = !BKC && lbm.fast_ok && kend == g.K && g.lda == g.K;
    lak.init([&](int s, int64_t &rg) -> const T * { rg = m0 + s; return (m0 + s < Mrows) ? g.A + (m0 + s) * g.lda : nullptr; }, g.A, tid,
             kr ? (Mrows - m0) * g.lda : 0);
    if (lak.fast_end(kend) < kfast) kfast = lak.fast_end(kend);
  }
  if constexpr (BKC) {
    lbk.init([&](int s, int64_t &rg) -> const T * { rg = n0 + s; return (n0 + s < g.N) ? g.B + (n0 + s) * g.ldb : nullptr; }, g.B, tid);
    if (lbk.fast_end(kend) < kfast) kfast = lbk.fast_end(kend);
  }
  if (kfast > kend + GBK) kfast = kend;

  typename M::acc_t acc[FM][FN];
  zero_acc<T, FM, FN>(acc);
  if (kbeg < kend) {
    kloop_kc<GBK>(
        kbeg, kend, kfast,
        [&](int64_t k0) {
          if constexpr (AKC) lak.load(k0, kend, tid);
          else lam.load(g.A, g.lda, m0, a_rend, k0, kbeg, kend, g.vec_a, tid, g.a_ones_row);
          if constexpr (BKC) lbk.load(k0, kend, tid);
          else lbm.load(g.B, g.ldb, n0, b_rend, k0, kbeg, kend, g.vec_b, tid, g.b_ones_row);
        },
        [&]() {
          if constexpr (AKC) lak.template store<LDA>(As, tid);
          else lam.template store<LDA>(As, tid);
          if constexpr (BKC) lbk.template store<LDB>(Bs, tid);
          else lbm.template store<LDB>(Bs, tid);
        },
        [&](int64_t k0) {
          if constexpr (AKC) lak.load_fast(k0, kend);
          else lam.load_fast(k0);
          if constexpr (BKC) lbk.load_fast(k0, kend);
          else lbm.load_fast(k0);
        },
        [&]() {
          if constexpr (AKC) lak.template store_fast<LDA>(As, tid);
          else lam.template store_fast<LDA>(As, tid);
          if constexpr (BKC) lbk.template store_fast<LDB>(Bs, tid);
          else lbm.template store_fast<LDB>(Bs, tid);
        },
        [&]() { mma_ktile<T, FM, FN, GBK, AKC, LDA, BKC, LDB>(As, Bs, wm, wn, lane, acc); });
  }

  T lsum = T(0);
  // whole tile inside C: element addresses split into a per-lane and a scalar part (TileIO)
  const bool interior = EPI != EPI_ATOMIC && EPI != EPI_FUSED && !g.trans_c && m0 + GBM <= Mrows && n0 + GBN <= g.N &&
                        TileIO<T>::usable(g.ldc, 32) && (EPI != EPI_DF || TileIO<T>::usable(g.ldy, 32));
  if (interior) {
    TileIO<T> ioc, ioy;
    const int64_t rw = m0 + wm;
    ioc.init(g.C + rw * g.ldc + n0 + wn, g.ldc, lane);
    if constexpr (EPI == EPI_DF) ioy.init(g.Yref + rw * g.ldy + n0 + wn, g.ldy, lane);
    T bias[FN];
    if constexpr (EPI == EPI_BIAS || EPI == EPI_DF) {
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) bias[fn] = g.bias[n0 + wn + fn * M::TM + M::col(lane)];
    }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      T y[FN][M::NACC];
      if constexpr (EPI == EPI_DF) {
#pragma unroll
        for (int fn = 0; fn < FN; ++fn)
#pragma unroll
          for (int r = 0; r < M::NACC; ++r) y[fn][r] = ioy.load(fm, fn, r);
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) {
          const T v = acc[fm][fn][r];
          if constexpr (EPI == EPI_STORE) {
            ioc.store(fm, fn, r, v);
          } else if constexpr (EPI == EPI_BIAS) {
            ioc.store(fm, fn, r, v + bias[fn]);
          } else if constexpr (EPI == EPI_DF) {
            const T diff = v + bias[fn] - y[fn][r];
            lsum += diff * diff;
            ioc.store(fm, fn, r, diff * g.inv_var);
          }
        }
    }
  } else {
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) {
      const int64_t row = m0 + wm + fm * M::TM + M::row(lane, r);
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const int64_t col = n0 + wn + fn * M::TM + M::col(lane);
        const bool in = row < Mrows && col < g.N;
        const T v = acc[fm][fn][r];
        const int64_t ci = g.trans_c ? col * g.ldc + row : row * g.ldc + col;
        if constexpr (EPI == EPI_STORE) {
          if (in) g.C[ci] = v;
        } else if constexpr (EPI == EPI_ATOMIC) {
          if (in) g.C[(int64_t)blockIdx.z * g.slab + ci] = g.negate ? -v : v;
        } else if constexpr (EPI == EPI_BIAS) {
          if (in) g.C[row * g.ldc + col] = v + g.bias[col];
        } else if constexpr (EPI == EPI_FUSED) {
          if (in) {
            T o = g.negate ? -v : v;
            if (g.addc) o += g.C[ci];
            if (g.bias != nullptr) o += g.bias[col];
            if (g.relu) o = o > T(0) ? o : T(0);
            if (g.mask != nullptr) o = g.mask[row * g.ldm + col] > T(0) ? o : T(0);
            g.C[ci] = o;
          }
        } else {
          const T y = g.Yref[in ? row * g.ldy + col : 0];
          const T b = g.bias[in ? col : 0];
          const T diff = in ? (v + b - y) : T(0);
          lsum += diff * diff;
          if (in) g.C[row * g.ldc + col] = diff * g.inv_var;
        }
      }
    }
  }
  if constexpr (EPI == EPI_DF) {
    lsum = wave_sum<T>(lsum);
    if (lane == 0) red[wave] = lsum;
    __syncthreads();
    if (tid == 0) {
      const T part = T(0.5) * g.inv_var * (red[0] + red[1] + red[2] + red[3]);
      g.loss_part[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = part;
    }
  }
}

// number of k-slices launch_gemm makes of a K-long product asked to split `splitk` ways
template <typename T> static inline int gemm_slices(int64_t K, int splitk) {
  constexpr int GBK = GemmK<T>::BK;
  if (splitk < 1) splitk = 1;
  int64_t kchunk = round_up(cdiv(K, (int64_t)splitk), (int64_t)GBK);
  if (kchunk < GBK) kchunk = GBK;
  return (int)(K > 0 ? cdiv(K, kchunk) : 1);
}

// sum over s < ns of p[s * slab + e], s ascending; eight loads in flight at a time (a plain loop waits for
// every load before it issues the next: 16 slabs were 16 L2 round trips)
template <typename T>
__device__ __forceinline__ T slab_sum_at(const T *__restrict__ p, int ns, int64_t slab, int64_t e) {
  T v = T(0);
  int s = 0;
  for (; s + 8 <= ns; s += 8) {
    T a[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) a[u] = p[(int64_t)(s + u) * slab + e];
#pragma unroll
    for (int u = 0; u < 8; ++u) v += a[u];
  }
  for (; s < ns; ++s) v += p[(int64_t)s * slab + e];
  return v;
}

// dst[i] = sum over s < ns of src[s * slab + i], s ascending (the fixed-order partner of slab stores)
template <typename T>
__global__ void slab_sum_kernel(T *dst, const T *__restrict__ src, int64_t count, int ns, int64_t slab) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count; i += (int64_t)gridDim.x * blockDim.x)
    dst[i] = slab_sum_at<T>(src, ns, slab, i);
}
template <typename T>
static void launch_slab_sum(hipStream_t st, T *dst, const T *src, int64_t count, int ns, int64_t slab) {
  if (count <= 0) return;
  int64_t blocks = cdiv(count, (int64_t)256 * 2);
  if (blocks > 2048) blocks = 2048;
  hipLaunchKernelGGL((slab_sum_kernel<T>), dim3((unsigned)blocks), dim3(256), 0, st, dst, src, count, ns, slab);
}

// returns the number of k-slices launched (= slabs written when g.slab > 0)
template <typename T, bool AKC, bool BKC, int EPI, int BM = 128>
static int launch_gemm(hipStream_t st, GemmArgs<T> g, int splitk) {
  constexpr int GBM = BM;
  if (g.M <= 0 || g.N <= 0) return 0;
  if (splitk < 1) splitk = 1;
  constexpr int GBK = GemmK<T>::BK;
  int64_t kchunk = round_up(cdiv(g.K, splitk), GBK);
  if (kchunk < GBK) kchunk = GBK;
  g.kchunk = kchunk;
  const int64_t nz = g.K > 0 ? cdiv(g.K, kchunk) : 1;
  dim3 grid((unsigned)cdiv(g.M, GBM), (unsigned)cdiv(g.N, GBN), (unsigned)nz);
  const bool kvec = (!AKC || g.vec_a) && (!BKC || g.vec_b);
  if (kvec || (!AKC && !BKC))
    hipLaunchKernelGGL((gemm_kernel<T, AKC, BKC, EPI, true, BM>), grid, dim3(NTHREADS), 0, st, g);
  else
    hipLaunchKernelGGL((gemm_kernel<T, AKC, BKC, EPI, false, BM>), grid, dim3(NTHREADS), 0, st, g);
  return (int)nz;
}

template <typename T> static inline bool mc_vec_ok(const void *base, int64_t ld) {
  return aligned16(base) && (ld % VecT<T>::N) == 0;
}

}  // namespace gpvae
