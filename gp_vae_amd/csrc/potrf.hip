// potrf.hip — batched blocked Cholesky with trailing rows (kriging a9).
//
// Replaces the Cholesky that dist.MultivariateNormal(covariance_matrix=k) runs
// (/root/reference/src/model/gp.py:190-193, inference.py:378-383,417-422) and, through the
// trailing-row solve, the explicit inverse of gp_prediction (inference.py:450-455).
//
// Two-level blocking, two kernels, no inter-workgroup hand-off inside a launch:
//   outer block NBO = 256 columns, right-looking:  potrf_syrk_kernel  C -= P P^T   (MFMA, K = 256;
//       >= 89 % of the flops at n = 4096; 128x128 tiles => 128 flop/byte of C traffic in f32).  On tall
//       matrices the update is delayed over groups of outer blocks (b3_group / mfma_group below): inside a
//       group only the next 256 columns are brought up to date, after it everything right of it with K = 256 g
//   inner block NB = 64 columns, left-looking inside the outer block: potrf_panel_kernel, ONE launch
//       per 64 columns.  Every workgroup (a) applies the outer block's earlier columns to the 64x64
//       diagonal block and to its own 64-row chunks with MFMA, (b) factors the diagonal block in LDS
//       (one wave, lane = row, row of L broadcast from LDS, 64 unrolled column steps), (c) solves
//       its chunks X L11^T = C with the same lane = row sweep, one chunk per wave.
//       The diagonal block is factored redundantly by every workgroup: that costs no wall time and
//       removes the launch boundary a separate "diagonal" kernel would need.
//       With GPVAE_POTRF_SHARE_CHIP (float32 planes route, calls that run beside other kernels) those launches cover
//       the outer block's own rows only and potrf_rows_kernel takes the rows below it through the four inner
//       blocks in one launch (half the chip time held, a longer chain).
// "Virtual rows": rows >= n come from a second array X (m_extra rows of length n); they ride along
// as extra chunks and come out as X L^-T — the right-hand solve kriging needs — for free.
// Rows/cols between n and the next multiple of 64 behave as an identity block.
#include "b3.hpp"

namespace gpvae {

constexpr int NB = 64;     // inner (panel) block
constexpr int NBO = 256;   // outer block
// outer blocks per group of the delayed trailing update (bf16 x 3 path; potrf_syrk_b3_kernel).  Measured at batch
// 16, f32 (ms, groups of 1 / 2 / 4 / 8, with the 128 x 128 tiles for badly filled launches): n = 1024: 0.473 / 0.455 /
// 0.469 / 0.476, 2048: 1.153 / 1.112 / 1.146 / 1.172, 4096: 4.44 / 4.04 / 3.90 / 3.95 (batch 32: 8.00 / 7.11 /
// 6.82 / 6.85) — the narrow launches of a group fill the chip only when the matrix is tall
// the same on the f32 / f64 MFMA trailing update (potrf_syrk_kernel; groups of 1 / 2 / 4): f64 16 x 4096 10.25 / 9.68 /
// 9.88 ms, f64 16 x 2048 2.24 / 2.22 / 2.40, f32 16 x 4000 5.61 / 5.30 / 5.39, f64 2 x 8192 12.1 / 11.9 / 12.4
static inline int mfma_group(int64_t n) { return n >= 15 * NBO ? 2 : 1; }
static inline int b3_group(int64_t n) { return n >= 16 * NBO ? 4 : (n >= 4 * NBO ? 2 : 1); }

// developer probe (make DEV=1 builds only): when set (gpvae_dev_set_stamps), block (0,0) of the panel
// launch of column block c0/64 writes s_memtime at its phase boundaries to stamps[8*(c0/64) + 0..7].
// The product build has no such global: the kernel's stamp pointer is a constant nullptr.
#ifdef GPVAE_DEV
static unsigned long long *g_stamps = nullptr;
#else
static constexpr unsigned long long *g_stamps = nullptr;
#endif
#define STAMP(i)                                                                   \
  do {                                                                             \
    if (stamps != nullptr && blockIdx.x == 0 && blockIdx.y == 0 && tid == 0)       \
      stamps[8 * (c0 / NB) + i] = __builtin_amdgcn_s_memtime();                                    \
  } while (0)

template <typename T> struct VRows {
  T *A;
  int64_t lda, n;
  T *X;
  int64_t ldx, npad, mx;
  __device__ __forceinline__ T *ptr(int64_t vr) const {
    if (vr < n) return A + vr * lda;
    const int64_t e = vr - npad;
    return (e >= 0 && e < mx) ? X + e * ldx : nullptr;
  }
};

// Where the bf16 planes of the factor's columns go (float32 problems on the bf16 x 3 path): piece pl of
// element (vrow, col) of matrix b at  p[b*stride + pl*plane + vrow*ld + col - coloff].
//   workspace mode (gpvae_potrf): only the current GROUP of outer panels (b3_group(n) of them), [row][256 D],
//     coloff = col0 = first column of the group;
//   factor mode (gpvae_potrf_b3): the caller's whole L3 [n][ld3], coloff = 0, every row.
// np = 2 (gpvae_potrf_h2): two fp16 planes of the factor TIMES scale[b] (a power of two chosen from the largest
// diagonal entry of matrix b: |L_ij| <= sqrt(max K_ii)), same addressing (two-byte elements either way).
struct B3Out {
  __bf16 *p;
  int64_t ld, plane, stride;
  int whole;           // factor mode
  int np;              // planes per operand: 3 (bf16) or 2 (fp16)
  const float *scale;  // np == 2: per matrix
  int64_t col0 = 0;    // matrix column held in plane column 0 (workspace mode: first column of the block pair)
};

template <typename T> struct PanelCfg;
template <> struct PanelCfg<float> {
  static constexpr int BK = 32;  // k-tile of the staged update GEMM
  static constexpr int BK3 = 32; // (three-chunk form: float64 only)
  static constexpr int NCH = 2;  // 64-row chunks per workgroup (one per wave in the solve)
  static constexpr int S = 68;   // LDS row stride of the 64-wide blocks (16-B aligned rows,
                                 // ds_read_b128 of one row per lane is conflict free: 17 slots/row)
};
template <> struct PanelCfg<double> {
  static constexpr int BK = 16;
  // Three chunks per workgroup (NCHV = 3 below): the float64 images of two chunks take 129 KB of LDS, one workgroup
  // per CU, so at batch 16 x 4096 the launches of the first eight outer blocks need two rounds of workgroups
  // (54 + 75 + 91 + 107 us per outer block against 26 + 35 + 43 + 51 in one round).  With k-tiles of 8 instead of
  // 16 in the staged update a third chunk fits (156 KB), and a launch that then needs fewer rounds takes that form
  // (the products are added in the same k order: the factor is the same bits).
  static constexpr int BK3 = 8;
  static constexpr int NCH = 2;
  static constexpr int S = 66;
};

// lane_bcast: common.hpp
// fast_rsqrt / pivot_scale / div_scaled: common.hpp (shared with the small-problem kriging kernel)

// ---- blocked (16-column windows) lane = row sweeps ---------------------------------------------------
// A 64-step sweep costs 2016 dependent FMAs per lane.  Split into four 16-column windows it costs
// 4 x 120, and the contribution of a finished window to the columns right of it is applied with
// 16x16x4 MFMAs on the LDS images (at most 16 per step, independent accumulators).  W is the first
// column of the window (run-time: the windows share one copy of the code), x[16] the lane's entries.
//
// DIAG (factor): row c+1 of L inside the window is fetched from LDS (broadcast read) while step c
// computes; only its newest entry travels by v_readlane (a readlane per entry was measured 45 %
// slower than the LDS broadcast).
// Solve: row c of L11 is the same for every lane and is read from LDS as a broadcast (zoff: a
// per-lane zero the compiler cannot see through keeps those reads in VGPRs, a provably uniform
// address makes hipcc add a v_readlane per value).
template <typename T, int S, bool DIAG>
__device__ __forceinline__ int sweep16(T (&x)[16], T *Lrow, T *rdiag, int W, int lane, int zoff,
                                       int fail) {
  constexpr int VEC = VecT<T>::N;
  using V = typename VecT<T>::type;
  if constexpr (DIAG) {
    T cur[16], nxt[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = W + j;
      if (j + 1 < 16) {
#pragma unroll
        for (int k = 0; k < j; k += VEC) {
          const V t = *reinterpret_cast<const V *>(&Lrow[(c + 1) * S + W + k + zoff]);
          const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
          for (int e = 0; e < VEC; ++e) nxt[k + e] = tp[e];
        }
      }
      T acc[4] = {x[j], T(0), T(0), T(0)};
#pragma unroll
      for (int k = 0; k < j; ++k) acc[k & 3] -= x[k] * cur[k];
      const T s = (acc[0] + acc[1]) + (acc[2] + acc[3]);
      const T piv = lane_bcast<T>(s, c);
      if (!(piv > T(0)) && fail < 0) fail = c;
      T ljj, rd;
      pivot_scale<T>(piv, ljj, rd);
      T v = (lane > c) ? div_scaled<T>(s, ljj, rd) : T(0);
      if (lane == c) v = ljj;
      x[j] = v;
      Lrow[lane * S + c] = v;
      if (lane == 0) rdiag[c] = rd;
      if (j + 1 < 16) {
        nxt[j] = lane_bcast<T>(v, c + 1);
#pragma unroll
        for (int k = 0; k <= j; ++k) cur[k] = nxt[k];
      }
    }
  } else {
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const int c = W + j;
      T acc[4] = {x[j], T(0), T(0), T(0)};
#pragma unroll
      for (int k = 0; k < j; k += VEC) {
        const V t = *reinterpret_cast<const V *>(&Lrow[c * S + W + k + zoff]);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if (k + e < j) acc[(k + e) & 3] -= x[k + e] * tp[e];
      }
      x[j] = div_scaled<T>((acc[0] + acc[1]) + (acc[2] + acc[3]), Lrow[c * S + c + zoff], rdiag[c + zoff]);
    }
  }
  return fail;
}

// Columns right of a finished window:  C[I, J] -= A[I, window] * L[J, window]^T  for the 16-row
// tiles I = 0..3 of the 64 rows in Cimg and the column blocks J > W/16 (DIAG: Cimg = Limg = the
// block being factored, lower tiles I >= J only).  Row-major LDS images, stride S, one wave.
// k values are dealt to the four lane groups in runs of VEC (one 16-byte read per operand and
// chunk), the same way for both operands.
template <typename T, int S, bool DIAG>
__device__ __forceinline__ void upd_window(T *Cimg, const T *Limg, int W, int lane) {
  using M = Mfma16<T>;
  using V = typename VecT<T>::type;
  constexpr int KV = VecT<T>::N, CK = 4 * KV, CH = 16 / CK;
  const int lr = lane & 15, g = lane >> 4;
  T a[4][CH][KV];
#pragma unroll
  for (int I = 0; I < 4; ++I)
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const V t = *reinterpret_cast<const V *>(&Cimg[(I * 16 + lr) * S + W + ch * CK + g * KV]);
      const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
      for (int e = 0; e < KV; ++e) a[I][ch][e] = -tp[e];
    }
#pragma unroll
  for (int J = W / 16 + 1; J < 4; ++J) {
    T bq[CH][KV];
#pragma unroll
    for (int ch = 0; ch < CH; ++ch) {
      const V t = *reinterpret_cast<const V *>(&Limg[(J * 16 + lr) * S + W + ch * CK + g * KV]);
      const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
      for (int e = 0; e < KV; ++e) bq[ch][e] = tp[e];
    }
    typename M::acc_t c[4];
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (!DIAG || I >= J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) c[I][r] = Cimg[(I * 16 + M::row(lane, r)) * S + J * 16 + lr];
      }
#pragma unroll
    for (int ch = 0; ch < CH; ++ch)
#pragma unroll
      for (int e = 0; e < KV; ++e)
#pragma unroll
        for (int I = 0; I < 4; ++I)
          if (!DIAG || I >= J) c[I] = M::mma(a[I][ch][e], bq[ch][e], c[I]);
#pragma unroll
    for (int I = 0; I < 4; ++I)
      if (!DIAG || I >= J) {
#pragma unroll
        for (int r = 0; r < 4; ++r) Cimg[(I * 16 + M::row(lane, r)) * S + J * 16 + lr] = c[I][r];
      }
  }
}

// one window of a 64-row image: load the lane's 16 entries, sweep, store, update to the right
template <typename T, int S, bool DIAG>
__device__ __forceinline__ int window16(T *img, T *Dm, T *rdiag, int W, int lane, int zoff, int fail) {
  constexpr int VEC = VecT<T>::N;
  using V = typename VecT<T>::type;
  T x[16];
  T *row = &img[lane * S + W];
#pragma unroll
  for (int k = 0; k < 16; k += VEC) {
    const V t = *reinterpret_cast<const V *>(&row[k]);
    const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[k + e] = tp[e];
  }
  fail = sweep16<T, S, DIAG>(x, Dm, rdiag, W, lane, zoff, fail);
#pragma unroll
  for (int k = 0; k < 16; k += VEC) {
    V t;
    T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
    for (int e = 0; e < VEC; ++e) tp[e] = x[k + e];
    *reinterpret_cast<V *>(&row[k]) = t;
  }
  if (W + 16 < NB) upd_window<T, S, DIAG>(img, Dm, W, lane);
  return fail;
}

// ---- the panel's left-looking update on the bf16 matrix pipe (float32 problems whose rows exist as planes) ----
// [D; C] -= [Pd; Pc] Pd^T over the columns [cJ, c0) the outer block has factored so far, read as the planes the
// earlier panel launches of this block wrote (b3.hpp: three bf16 pieces, six piece products; NP = 2: two fp16
// pieces of L * scale, three products).  The f32 MFMA version of this update was 0.8 of the 2.1 ms the 64 panel
// launches of a 16 x 4096 factorisation take (10 / 17 / 24 us of the second / third / fourth launch of a block).
// Per block of 64 columns: the diagonal rows' planes (the B operand of every product, and both operands of the
// wave's quadrant of D) are staged in LDS once per workgroup — [plane][8-k chunk][row][16 B], the conflict-free
// fragment image of PipeB3 — while every wave loads the fragments of its own 32 chunk rows straight from global
// memory, 16 bytes per lane = the 8 k values of a 32x32x16 operand row, four k16 steps ahead in a ring of
// registers.  (All nine fragments per step straight from global memory was tried first: a CU issues one 16-byte
// wave load per ~29 cycles whatever its pattern — tools/ub_vmem.hip — and 72 per step and CU took twice the
// MFMA time: 4.86 against 4.46 ms per 16 x 4096.)
template <int NP> struct PanelB3 {
  static constexpr int CP = 64 * 16 + 32;              // bytes per (plane, chunk) image: 64 rows x 16 B + pad
  static constexpr int LDS_BYTES = NP * 8 * CP;        // one block of 64 columns: 25.3 KB (NP = 3)
  static constexpr int NST = NP * 64 * 8 / NTHREADS;   // 16-byte pieces per thread and block: 6 (4)
  i32x4_t ringA[4][NP];   // [k16 step of the block][plane]: the wave's own rows
  i32x4_t st[NST];        // staging registers: piece i = plane i / 2, diagonal row tid / 8 + 32 (i % 2), chunk tid % 8
  i32x4_t fb[2][NP][2];   // fragments of the two 32-row tiles of the diagonal rows: this k16 step's and the next one's
  const char *pa, *pd;    // the lane's own row / the thread's staging row and chunk, at column cJ
  int64_t plane2, ld32;   // bytes between planes / between 32 rows
  int wr, rd;             // LDS byte offsets: staging write, fragment read

  __device__ __forceinline__ void init(const B3Out &o3, int b, int64_t row_a, int64_t row_d, int64_t cJ, int tid) {
    const int lane = tid & 63;
    const __bf16 *base = o3.p + (int64_t)b * o3.stride + (cJ - o3.col0);
    pa = reinterpret_cast<const char *>(base + (row_a + (lane & 31)) * o3.ld + (lane >> 5) * 8);
    pd = reinterpret_cast<const char *>(base + (row_d + (tid >> 3)) * o3.ld + (tid & 7) * 8);
    plane2 = o3.plane * 2;
    ld32 = o3.ld * 64;
    wr = (tid & 7) * CP + (tid >> 3) * 16;
    rd = (lane >> 5) * CP + (lane & 31) * 16;
  }
  template <int STEP> __device__ __forceinline__ void load_a(int blk) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
      ringA[STEP][p] = *reinterpret_cast<const i32x4_t *>(pa + p * plane2 + blk * 128 + STEP * 32);
  }
  __device__ __forceinline__ void stage_load(int blk) {
#pragma unroll
    for (int i = 0; i < NST; ++i)
      st[i] = *reinterpret_cast<const i32x4_t *>(pd + (i >> 1) * plane2 + (i & 1) * ld32 + blk * 128);
  }
  __device__ __forceinline__ void stage_store(unsigned char *lds) {
#pragma unroll
    for (int i = 0; i < NST; ++i)
      *reinterpret_cast<i32x4_t *>(lds + wr + (i >> 1) * 8 * CP + (i & 1) * 512) = st[i];
  }
  template <int STEP> __device__ __forceinline__ void read_b(const unsigned char *lds) {
#pragma unroll
    for (int p = 0; p < NP; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        fb[STEP & 1][p][t] = *reinterpret_cast<const i32x4_t *>(lds + rd + (p * 8 + 2 * STEP) * CP + t * 512);
  }
  __device__ __forceinline__ static f32x16 mfma(const i32x4_t &x, const i32x4_t &y, const f32x16 &c) {
    if constexpr (NP == 3)
      return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, x), __builtin_bit_cast(bf16x8_t, y), c, 0, 0, 0);
    else
      return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, x), __builtin_bit_cast(f16x8_t, y), c, 0, 0, 0);
  }
  // accC[fn] += Pc Pd(32 fn ..)^T ; accD += Pd(32 di ..) Pd(32 dj ..)^T, smallest piece products first.
  // di, dj MUST be scalar values (readfirstlane): an MFMA does not honour EXEC, so a quadrant choice the compiler
  // takes for divergent runs all four products into accD.
  template <int STEP, bool WD = true>
  __device__ __forceinline__ void mma(f32x16 (&accC)[2], f32x16 &accD, int di, int dj) {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};  // NP == 3
    constexpr int QA[3] = {0, 1, 0}, QB[3] = {1, 0, 0};                    // NP == 2
    constexpr int NPROD = NP == 3 ? 6 : 3;
#pragma unroll
    for (int q = 0; q < NPROD; ++q) {
      const int a = NP == 3 ? PA[q] : QA[q], bq = NP == 3 ? PB[q] : QB[q];
      constexpr int F = STEP & 1;
      accC[0] = mfma(ringA[STEP][a], fb[F][bq][0], accC[0]);
      accC[1] = mfma(ringA[STEP][a], fb[F][bq][1], accC[1]);
      if constexpr (!WD) continue;  // potrf_rows_kernel: the diagonal block was factored by an earlier launch
      if (di == 0 && dj == 0) accD = mfma(fb[F][a][0], fb[F][bq][0], accD);
      else if (di == 0) accD = mfma(fb[F][a][0], fb[F][bq][1], accD);
      else if (dj == 0) accD = mfma(fb[F][a][1], fb[F][bq][0], accD);
      else accD = mfma(fb[F][a][1], fb[F][bq][1], accD);
    }
  }
  // first requests, ahead of the barrier that ends the copy-in
  __device__ __forceinline__ void prologue() {
    stage_load(0);
    load_a<0>(0); load_a<1>(0); load_a<2>(0); load_a<3>(0);
  }
  // nblk blocks of four k16 steps (64 columns each); two barriers per block
  template <bool WD = true>
  __device__ __forceinline__ void run(unsigned char *lds, int nblk, f32x16 (&accC)[2], f32x16 &accD, int di, int dj) {
    for (int blk = 0; blk < nblk; ++blk) {
      const bool more = blk + 1 < nblk;
      stage_store(lds);
      __syncthreads();
      if (more) stage_load(blk + 1);
      read_b<0>(lds);
      read_b<1>(lds); mma<0, WD>(accC, accD, di, dj); if (more) load_a<0>(blk + 1);
      read_b<2>(lds); mma<1, WD>(accC, accD, di, dj); if (more) load_a<1>(blk + 1);
      read_b<3>(lds); mma<2, WD>(accC, accD, di, dj); if (more) load_a<2>(blk + 1);
      mma<3, WD>(accC, accD, di, dj); if (more) load_a<3>(blk + 1);
      __syncthreads();  // every wave has read this block's image
    }
  }
};

// NPL = 3 / 2: the rows exist as bf16 / fp16 planes (B3Out) and the left-looking update runs on them (float32
// only); NPL = 0: the update on the f32 / f64 MFMA from A itself
template <typename T, bool VECM, int NPL, int NCHV = 0>
__global__ __launch_bounds__(NTHREADS, NPL ? 2 : 1) void potrf_panel_kernel(VRows<T> vr, int64_t strideA,
                                                               int64_t strideX, int64_t c0,
                                                               int64_t cJ, int32_t *info, T *dwork,
                                                               B3Out o3,
                                                               unsigned long long *stamps, int64_t rlim) {
  using M = Mfma<T>;
  using Cfg = PanelCfg<T>;
  constexpr int NCH = NCHV ? NCHV : Cfg::NCH, S = Cfg::S, PBK = NCHV == 3 ? Cfg::BK3 : Cfg::BK;
  static_assert(NCHV == 0 || (NCHV == 3 && sizeof(T) == 8 && NPL == 0), "three chunks: the float64 form");
  constexpr int SROWS = NB + NB * NCH;              // stacked rows: diagonal block + chunks
  constexpr int LDT = PBK + RmPad<T>::V;            // row-major staging image [SROWS][LDT]
  constexpr int RPW = 16 * NCH;                     // chunk rows per wave in the GEMM phase
  constexpr int FM = RPW / M::TM;                   // = 2 for both types
  constexpr int FN = NB / M::TM;                    // 2 (f32) / 4 (f64)
  constexpr int DN = (M::TM == 32) ? 1 : 4;         // diagonal-block tiles per wave
  constexpr int VEC = VecT<T>::N;
  constexpr int VPR = NB / VEC;                     // vectors per 64-wide row
  using V = typename VecT<T>::type;
  static_assert(FM >= 1 && RPW % M::TM == 0, "chunk rows per wave");
  static_assert((SROWS * VPR) % NTHREADS == 0, "copy-in tiling");

  __shared__ __align__(16) T Blk[SROWS * S];   // [diag block (64 rows); chunks] row-major, stride S
  __shared__ __align__(16) T stage[SROWS * LDT];
  __shared__ T rdiag[NB];
  T *Dm = Blk, *Cw = Blk + NB * S;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  vr.A += (int64_t)b * strideA;
  if (vr.X) vr.X += (int64_t)b * strideX;
  const int64_t n = vr.n;
  const int64_t rbase = c0 + NB + (int64_t)blockIdx.x * NB * NCH;  // first chunk row (virtual)
  // rlim: virtual rows from rlim on are left to another launch (potrf_rows_kernel): they read as absent here
  const int64_t Mv = (vr.npad + vr.mx < rlim) ? vr.npad + vr.mx : rlim;
  auto rowp = [&](int64_t vrow) -> T * { return vrow < rlim ? vr.ptr(vrow) : nullptr; };

  STAMP(0);
  int zoff;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
  const int dm0 = (M::TM == 32) ? (wave >> 1) * 32 : wave * 16;
  const int dn0 = (M::TM == 32) ? (wave & 1) * 32 : 0;
  const bool vcols = VECM && (c0 + NB <= n);  // the 64 columns are whole and 16-byte aligned

  // planes route: the first block of plane fragments is requested ahead of the copy-in (both latencies together)
  constexpr bool PL = NPL != 0;
  PanelB3<PL ? NPL : 3> pl;
  if constexpr (PL) {
    if (c0 > cJ) {
      const int64_t row_a = rbase + wave * 32;  // rows past n (second chunk of the last workgroup) are never stored:
      pl.init(o3, b, row_a + 31 < n ? row_a : n - 32, c0, cJ, tid);  // read real rows instead
      pl.prologue();
    }
  }

  // ---- copy [D; C] (columns c0..c0+63 of the stacked rows) global -> LDS, 16 bytes per lane ------
  // branch-free (clamped pointer + select) so every load is in flight at once
  {
    constexpr int NVC = SROWS * VPR / NTHREADS;
    T tmp[NVC][VEC];
#pragma unroll
    for (int p = 0; p < NVC; ++p) {
      const int v = tid + p * NTHREADS;
      const int s = v / VPR, j = (v % VPR) * VEC;
      const int64_t vrow = (s < NB) ? c0 + s : rbase + (s - NB);
      const T *rp = rowp(vrow);
      const bool rok = rp != nullptr;
      const T *src = (rok ? rp : vr.A) + c0 + j;
      if (vcols) {
        const V t = *reinterpret_cast<const V *>(src);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) tmp[p][e] = rok ? tp[e] : T(0);
      } else {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const bool in = rok && c0 + j + e < n;
          tmp[p][e] = in ? src[in ? e : -(c0 + j)] : T(0);
        }
      }
      if (s < NB) {  // rows/cols past n act as an identity block
#pragma unroll
        for (int e = 0; e < VEC; ++e)
          if ((vrow >= n || c0 + j + e >= n)) tmp[p][e] = (s == j + e) ? T(1) : T(0);
      }
    }
    STAMP(6);
#pragma unroll
    for (int p = 0; p < NVC; ++p) {
      const int v = tid + p * NTHREADS;
      const int s = v / VPR, j = (v % VPR) * VEC;
      V t;
      T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) tp[e] = tmp[p][e];
      *reinterpret_cast<V *>(&Blk[s * S + j]) = t;
    }
    STAMP(7);
  }

  // ---- (a) apply columns [cJ, c0) of the outer block: [D; C] -= [Pd; Pc] Pd^T ------------------
  if constexpr (PL) {
    static_assert(sizeof(T) == 4 && VECM, "planes: float32 with 16-byte aligned rows");
    if (c0 > cJ) {  // the bf16 x 3 / fp16 x 2 route: these rows exist as planes (PanelB3 above)
      const int nblk = (int)((c0 - cJ) / NB);
      // scalar values: the quadrant branches of PanelB3::mma must be real branches (an MFMA does not honour EXEC)
      const int wu = __builtin_amdgcn_readfirstlane(wave);
      const int di = wu >> 1, dj = wu & 1;
      __shared__ __align__(16) unsigned char pst[PanelB3<PL ? NPL : 3>::LDS_BYTES];
      float sc2 = 1.0f, isc2 = 1.0f;  // fp16 x 2: the planes hold L * scale
      if constexpr (NPL == 2) {
        const float sc = o3.scale[b], is = 1.0f / sc;
        sc2 = sc * sc;
        isc2 = is * is;
      }
      __syncthreads();  // the copy-in above is complete
      f32x16 accC[2], accD;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accD[r] = -Dm[(di * 32 + M::row(lane, r)) * S + dj * 32 + M::col(lane)] * sc2;
        const int ci = wave * 32 + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) accC[fn][r] = -Cw[ci * S + fn * 32 + M::col(lane)] * sc2;
      }
      pl.run(pst, nblk, accC, accD, di, dj);
      STAMP(1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {  // every element belongs to exactly one lane
        Dm[(di * 32 + M::row(lane, r)) * S + dj * 32 + M::col(lane)] = -accD[r] * isc2;
        const int ci = wave * 32 + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) Cw[ci * S + fn * 32 + M::col(lane)] = -accC[fn][r] * isc2;
      }
    }
  }
  if (!PL && c0 > cJ) {
    // The accumulators START from the (negated) block and the products are added to them:
    // acc = -C + sum_k P P^T, stored back negated.  Summing the products first and subtracting the
    // sum from C afterwards loses the small terms against the large first one on nearly rank-one
    // matrices (long lengthscales): measured 30x LAPACK's backward error and spurious "not positive
    // definite" reports at n = 300, jitter 1e-4.
    typename M::acc_t accC[FM][FN];
    typename M::acc_t accD[1][DN];
    __syncthreads();  // the copy-in above is complete
#pragma unroll
    for (int fn = 0; fn < DN; ++fn)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r)
        accD[0][fn][r] = -Dm[(dm0 + M::row(lane, r)) * S + dn0 + fn * M::TM + M::col(lane)];
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int ci = wave * RPW + fm * M::TM + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) accC[fm][fn][r] = -Cw[ci * S + fn * M::TM + M::col(lane)];
      }
    LoaderKC<T, SROWS, PBK, VECM> ld;
    ld.init(
        [&](int s, int64_t &rg) -> const T * {
          const int64_t vrow = (s < NB) ? c0 + s : rbase + (s - NB);
          rg = vrow;
          return rowp(vrow);
        },
        vr.A, tid);
    kloop_kc<PBK>(
        cJ, c0, ld.fast_end(c0),
        [&](int64_t k) { ld.load(k, c0, tid); },
        [&]() { ld.template store<LDT>(stage, tid); },
        [&](int64_t k) { ld.load_fast(k, c0); },
        [&]() { ld.template store_fast<LDT>(stage, tid); },
        [&]() {
          mma_ktile<T, FM, FN, PBK, true, LDT, true, LDT>(stage, stage, NB + wave * RPW, 0, lane, accC);
          mma_ktile<T, 1, DN, PBK, true, LDT, true, LDT>(stage, stage, dm0, dn0, lane, accD);
        });
    STAMP(1);
    // back into the LDS image; every element belongs to exactly one lane
#pragma unroll
    for (int fn = 0; fn < DN; ++fn)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int i = dm0 + M::row(lane, r), j = dn0 + fn * M::TM + M::col(lane);
        Dm[i * S + j] = -accD[0][fn][r];
      }
#pragma unroll
    for (int fm = 0; fm < FM; ++fm)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int ci = wave * RPW + fm * M::TM + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) Cw[ci * S + fn * M::TM + M::col(lane)] = -accC[fm][fn][r];
      }
  }
  __syncthreads();
  STAMP(2);

  // ---- (b)+(c) factor the diagonal block (wave 3) and solve the chunks (waves 0..NCH-1, one
  // 64-row chunk each), both lane = row over 16-column windows, pipelined: while wave 3 factors
  // window w, the solving waves work on window w-1 — a solve window reads columns
  // [16(w-1), 16w) of L11 (final since the previous phase), the factoring wave touches columns
  // >= 16w only.  One barrier per phase.
  static_assert(NCH <= 3, "wave 3 factors the diagonal block");
  {
    int fail = -1;
    const bool solver = wave < NCH && rbase + wave * NB < Mv;
    T *chunk = &Cw[(wave < NCH ? wave : 0) * NB * S];
#pragma unroll
    for (int ph = 0; ph <= NB / 16; ++ph) {
      if (wave == 3 && ph < NB / 16) fail = window16<T, S, true>(Dm, Dm, rdiag, 16 * ph, lane, zoff, fail);
      if (solver && ph >= 1) window16<T, S, false>(chunk, Dm, rdiag, 16 * (ph - 1), lane, zoff, -1);
      __syncthreads();
    }
    if (wave == 3 && blockIdx.x == 0 && lane == 0 && fail >= 0 && c0 + fail < n) {
      if (info[b] == 0) info[b] = (int32_t)(c0 + fail + 1);
    }
  }
  STAMP(3);
  STAMP(4);

  // ---- coalesced stores ---------------------------------------------------------------------------
  // The factored diagonal block is NOT written over D here: a workgroup of this launch that starts
  // late (busy CUs, more workgroups than CUs) must still find the unfactored D in A.  L11 goes to
  // the caller's scratch (dense 64x64 per diagonal block) and potrf_finalize_kernel puts it in place.
  const int nst = (NCH + (blockIdx.x == 0 ? 1 : 0)) * NB * VPR;
  T *l11 = dwork + ((int64_t)b * (vr.npad / NB) + c0 / NB) * (NB * NB);
  if constexpr (PL) {
    // planes route (whole 64-column blocks of real rows): 8 columns per lane, so that every plane goes out as
    // 16-byte stores — 5 store instructions per 8 elements instead of 8 (the phase is bound by their issue rate)
    float sc = 1.0f;
    if constexpr (NPL == 2) sc = o3.scale[b];
#pragma unroll 2
    for (int idx = tid; idx < NCH * NB * 8; idx += NTHREADS) {
      const int s = idx >> 3, j = (idx & 7) * 8;
      T *rp = rowp(rbase + s);
      if (rp == nullptr) continue;
      const V t0 = *reinterpret_cast<const V *>(&Cw[s * S + j]), t1 = *reinterpret_cast<const V *>(&Cw[s * S + j + 4]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = reinterpret_cast<const float *>(&t0)[e];
        v[4 + e] = reinterpret_cast<const float *>(&t1)[e];
      }
      __bf16 *q = o3.p + (int64_t)b * o3.stride + (rbase + s) * o3.ld + (c0 - o3.col0) + j;
      if constexpr (NPL == 2) {
        _Float16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2_f16(v[e] * sc, hh[e], ll[e]);
        *reinterpret_cast<i32x4_t *>(q) = *reinterpret_cast<const i32x4_t *>(hh);
        *reinterpret_cast<i32x4_t *>(q + o3.plane) = *reinterpret_cast<const i32x4_t *>(ll);
      } else {
        __bf16 hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3_bf16(v[e], hh[e], mm[e], ll[e]);
        *reinterpret_cast<i32x4_t *>(q) = *reinterpret_cast<const i32x4_t *>(hh);
        *reinterpret_cast<i32x4_t *>(q + o3.plane) = *reinterpret_cast<const i32x4_t *>(mm);
        *reinterpret_cast<i32x4_t *>(q + 2 * o3.plane) = *reinterpret_cast<const i32x4_t *>(ll);
      }
      *reinterpret_cast<V *>(rp + c0 + j) = t0;
      *reinterpret_cast<V *>(rp + c0 + j + 4) = t1;
    }
  }
#pragma unroll 4
  for (int idx = PL ? NCH * NB * VPR + tid : tid; idx < nst; idx += NTHREADS) {
    const int s = idx / VPR, j = (idx % VPR) * VEC;  // s < NCH*NB: chunk rows; beyond: L11 (block 0)
    const bool is_l11 = s >= NCH * NB;
    const int li = s - NCH * NB;
    V t = *reinterpret_cast<const V *>(is_l11 ? &Dm[li * S + j] : &Cw[s * S + j]);
    T *tp = reinterpret_cast<T *>(&t);
    if (is_l11) {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (j + e > li) tp[e] = T(0);  // clean upper triangle inside the diagonal block
      *reinterpret_cast<V *>(&l11[li * NB + j]) = t;
      continue;
    }
    T *rp = rowp(rbase + s);
    if (rp == nullptr) continue;
    T *dst = rp + c0 + j;
    if constexpr (sizeof(T) == 4) {
      // the rows also go out as three bf16 planes: the operand of the bf16 x 3 trailing update
      // (potrf_syrk_b3_kernel) and of the next launches' left-looking update, pieces summing to the float exactly
      if (o3.p != nullptr) {  // (the rows inside the outer block too: the next launches' left-looking update reads them)
        __bf16 *q = o3.p + (int64_t)b * o3.stride + (rbase + s) * o3.ld + (c0 - o3.col0) + j;
        if (o3.np == 2) {
          const float sc = o3.scale[b];
          _Float16 hh[4], ll[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split2_f16(tp[e] * sc, hh[e], ll[e]);
          *reinterpret_cast<uint64_t *>(q) = *reinterpret_cast<const uint64_t *>(hh);
          *reinterpret_cast<uint64_t *>(q + o3.plane) = *reinterpret_cast<const uint64_t *>(ll);
        } else {
          __bf16 hh[4], mm[4], ll[4];
#pragma unroll
          for (int e = 0; e < 4; ++e) split3_bf16(tp[e], hh[e], mm[e], ll[e]);
          *reinterpret_cast<uint64_t *>(q) = *reinterpret_cast<const uint64_t *>(hh);
          *reinterpret_cast<uint64_t *>(q + o3.plane) = *reinterpret_cast<const uint64_t *>(mm);
          *reinterpret_cast<uint64_t *>(q + 2 * o3.plane) = *reinterpret_cast<const uint64_t *>(ll);
        }
      }
    }
    if (vcols) {
      *reinterpret_cast<V *>(dst) = t;
    } else {
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        if (c0 + j + e < n) dst[e] = tp[e];
    }
  }
  STAMP(5);
}

// ---- the rows below an outer block, all four inner blocks in ONE launch (planes route) --------------------------
// A panel workgroup holds half a CU's LDS for its whole life, and a draw workgroup (b3.hpp: 145 KB) the whole of it:
// while the factor stream's panel launches run beside the training step, every CU they sit on is lost to the draw,
// although the panel work is latency-bound and keeps the matrix pipe ~10 % busy.  Four launches per outer block, each
// with all rows below it, held 0.9 ms of chip time per 16 x 4096 factorisation.  Here the four launches of an outer
// block cover the rows INSIDE the block only (potrf_panel_kernel with rlim = the block's end: at most two
// workgroups per matrix; they factor the four 64 x 64 diagonal blocks and park them in dwork), and the rows below the
// block take ONE launch of this kernel: a workgroup keeps its 128 rows and walks the four inner blocks — copy-in,
// left-looking update on the planes (its own rows' planes of the earlier inner blocks are its own stores of a moment
// ago), triangular solve against the parked diagonal block by lane = row sweeps (no factorisation to wait for: the
// four windows run back to back), stores — without the three launch boundaries and without re-factoring the diagonal
// block in every workgroup.
template <int NPL>
__global__ __launch_bounds__(NTHREADS, 2) void potrf_rows_kernel(float *A, int64_t lda, int64_t strideA, int64_t cJ,
                                                                 const float *dwork, int64_t npad, B3Out o3) {
  using T = float;
  using M = Mfma<float>;
  constexpr int NCH = 2, S = PanelCfg<float>::S, VPR = NB / 4;
  using V = float4;
  __shared__ __align__(16) T Dm[NB * S];
  __shared__ __align__(16) T Cw[NCH * NB * S];
  __shared__ T rdiag[NB];
  __shared__ __align__(16) unsigned char pst[PanelB3<NPL>::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.y;
  A += (int64_t)b * strideA;
  const int64_t rbase = cJ + NBO + (int64_t)blockIdx.x * NB * NCH;  // whole workgroups of real rows (n % 256 == 0)
  int zoff;
  asm volatile("v_mov_b32 %0, 0" : "=v"(zoff));
  const int wu = __builtin_amdgcn_readfirstlane(wave);
  float sc = 1.0f, sc2 = 1.0f, isc2 = 1.0f;  // fp16 x 2: the planes hold L * scale
  if constexpr (NPL == 2) {
    sc = o3.scale[b];
    const float is = 1.0f / sc;
    sc2 = sc * sc;
    isc2 = is * is;
  }
  for (int ib = 0; ib < NBO / NB; ++ib) {
    const int64_t c0 = cJ + (int64_t)ib * NB;
    PanelB3<NPL> pl;
    if (ib > 0) {  // plane fragments of the first block of earlier columns, ahead of the copy-in
      pl.init(o3, b, rbase + wave * 32, c0, cJ, tid);
      pl.prologue();
    }
    {  // the parked diagonal block and this workgroup's rows of the inner block -> LDS
      // (requesting the next inner block's copy-in ahead of the solve was measured: the launch took 110-125 us
      // instead of 75-88)
      const T *l11 = dwork + ((int64_t)b * (npad / NB) + c0 / NB) * (NB * NB);
      V td[NB * VPR / NTHREADS], tc[NCH * NB * VPR / NTHREADS];
#pragma unroll
      for (int p = 0; p < NB * VPR / NTHREADS; ++p) {
        const int v = tid + p * NTHREADS;
        td[p] = *reinterpret_cast<const V *>(l11 + (v / VPR) * NB + (v % VPR) * 4);
      }
#pragma unroll
      for (int p = 0; p < NCH * NB * VPR / NTHREADS; ++p) {
        const int v = tid + p * NTHREADS;
        tc[p] = *reinterpret_cast<const V *>(A + (rbase + v / VPR) * lda + c0 + (v % VPR) * 4);
      }
#pragma unroll
      for (int p = 0; p < NB * VPR / NTHREADS; ++p) {
        const int v = tid + p * NTHREADS;
        *reinterpret_cast<V *>(&Dm[(v / VPR) * S + (v % VPR) * 4]) = td[p];
      }
#pragma unroll
      for (int p = 0; p < NCH * NB * VPR / NTHREADS; ++p) {
        const int v = tid + p * NTHREADS;
        *reinterpret_cast<V *>(&Cw[(v / VPR) * S + (v % VPR) * 4]) = tc[p];
      }
    }
    __syncthreads();
    if (tid < NB) {  // 1 / L_jj, refined (the solve divides as the factorisation does: div_scaled)
      const T ljj = Dm[tid * S + tid];
      T y = __builtin_amdgcn_rcpf(ljj);
      y = fma(fma(-ljj, y, T(1)), y, y);
      rdiag[tid] = fma(fma(-ljj, y, T(1)), y, y);
    }
    if (ib > 0) {  // C -= Pc Pd^T over the block's earlier columns (accumulators start from -C, as everywhere)
      f32x16 accC[2], accD;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        accD[r] = 0.0f;
        const int ci = wave * 32 + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) accC[fn][r] = -Cw[ci * S + fn * 32 + M::col(lane)] * sc2;
      }
      pl.template run<false>(pst, ib, accC, accD, wu >> 1, wu & 1);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = wave * 32 + M::row(lane, r);
#pragma unroll
        for (int fn = 0; fn < 2; ++fn) Cw[ci * S + fn * 32 + M::col(lane)] = -accC[fn][r] * isc2;
      }
    }
    __syncthreads();
    if (wave < NCH) {  // X L11^T = C: one 64-row chunk per wave, four 16-column windows back to back
      T *chunk = &Cw[wave * NB * S];
#pragma unroll
      for (int ph = 0; ph < NB / 16; ++ph) window16<T, S, false>(chunk, Dm, rdiag, 16 * ph, lane, zoff, -1);
    }
    __syncthreads();
    // stores: float32 rows and their planes, 8 columns per lane (as potrf_panel_kernel's planes route)
#pragma unroll 2
    for (int idx = tid; idx < NCH * NB * 8; idx += NTHREADS) {
      const int s2 = idx >> 3, j = (idx & 7) * 8;
      const V t0 = *reinterpret_cast<const V *>(&Cw[s2 * S + j]), t1 = *reinterpret_cast<const V *>(&Cw[s2 * S + j + 4]);
      float v[8];
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        v[e] = reinterpret_cast<const float *>(&t0)[e];
        v[4 + e] = reinterpret_cast<const float *>(&t1)[e];
      }
      __bf16 *q = o3.p + (int64_t)b * o3.stride + (rbase + s2) * o3.ld + (c0 - o3.col0) + j;
      if constexpr (NPL == 2) {
        _Float16 hh[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split2_f16(v[e] * sc, hh[e], ll[e]);
        *reinterpret_cast<i32x4_t *>(q) = *reinterpret_cast<const i32x4_t *>(hh);
        *reinterpret_cast<i32x4_t *>(q + o3.plane) = *reinterpret_cast<const i32x4_t *>(ll);
      } else {
        __bf16 hh[8], mm[8], ll[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) split3_bf16(v[e], hh[e], mm[e], ll[e]);
        *reinterpret_cast<i32x4_t *>(q) = *reinterpret_cast<const i32x4_t *>(hh);
        *reinterpret_cast<i32x4_t *>(q + o3.plane) = *reinterpret_cast<const i32x4_t *>(mm);
        *reinterpret_cast<i32x4_t *>(q + 2 * o3.plane) = *reinterpret_cast<const i32x4_t *>(ll);
      }
      T *rp = A + (rbase + s2) * lda + c0 + j;
      *reinterpret_cast<V *>(rp) = t0;
      *reinterpret_cast<V *>(rp + 4) = t1;
    }
    // the next inner block's left-looking update reads these planes back: the stores are complete (and the LDS
    // images free) once every wave has passed the barrier
    __syncthreads();
  }
}

// ---- outer trailing update: C[r][c] -= sum_{k in [cJ, cJ+kb)} P[r][k] P[c][k] ----------------------
// 128x128 tiles, 4 waves as 2x2, each 64x64.  One workgroup per tile on or below the diagonal
// (ct (ct+1)/2 of them) plus the (rt - ct) x ct rectangle of the extra rows, in a 1-D grid.
// The hardware deals consecutive workgroup ids to the 8 XCDs in turn.  A (row tiles, col tiles) grid
// whose row-tile count is a multiple of 8 therefore gives XCD x the tile rows = x (mod 8), and row
// r holds r + 1 tiles: at 16 tile rows XCD 7 had 2.4x the work of XCD 0 (62 instead of ~95 TFLOP/s
// for that launch).  Here id -> (xcd = id % 8, k = id / 8); XCD x owns the matrices b = x (mod 8) and
// walks their tiles in row-major order, so its L2 holds the panels of one matrix at a time and every
// XCD has the same number of tiles.  With fewer than 8 matrices the tiles are dealt round-robin.
constexpr int SBM = 128;

struct SyrkMap {
  int rt, ct, batch;
  int64_t per;  // tiles per matrix; narrow (the first two tile columns only): 2 rt - 1
  int narrow;
  __host__ __device__ static int64_t tiles(int rt, int ct) {
    return (int64_t)ct * (ct + 1) / 2 + (int64_t)(rt - ct) * ct;
  }
  __host__ int64_t grid() const {
    return batch >= 8 ? 8 * (int64_t)((batch + 7) / 8) * per : (int64_t)batch * per;
  }
  // false: this workgroup has nothing to do
  __device__ __forceinline__ bool decode(int64_t id, int &b, int &tr, int &tc) const {
    int64_t p;
    if (batch >= 8) {
      const int xcd = (int)(id & 7);
      const int64_t k = id >> 3;
      const int64_t mb = k / per;
      b = xcd + 8 * (int)mb;
      if (b >= batch) return false;
      p = k - mb * per;
    } else {
      b = (int)(id / per);
      p = id - (int64_t)b * per;
    }
    if (narrow) {
      tr = (int)((p + 1) >> 1);  // (0,0), (1,0), (1,1), (2,0), (2,1), ...
      tc = p == 0 ? 0 : (int)((p + 1) & 1);
      return true;
    }
    const int64_t ntri = (int64_t)ct * (ct + 1) / 2;
    if (p < ntri) {
      int r = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
      while ((int64_t)r * (r + 1) / 2 > p) --r;
      while ((int64_t)(r + 1) * (r + 2) / 2 <= p) ++r;
      tr = r;
      tc = (int)(p - (int64_t)r * (r + 1) / 2);
    } else {
      const int64_t q = p - ntri;
      tr = ct + (int)(q / ct);
      tc = (int)(q % ct);
    }
    return true;
  }
};

template <typename T, bool VECM>
__global__ __launch_bounds__(NTHREADS, 2) void potrf_syrk_kernel(
    VRows<T> vr, int64_t strideA, int64_t strideX, int64_t cP, int64_t kend, SyrkMap map) {
  using M = Mfma<T>;
  constexpr int SBK = TileK<T>::BK;
  constexpr int F = 64 / M::TM;  // MFMA tiles per wave edge
  constexpr int LD = SBK + RmPad<T>::V;  // row-major images [128][LD]
  int b, tr, tc;
  if (!map.decode(blockIdx.x, b, tr, tc)) return;
  using LA = LoaderKC<T, SBM, SBK, VECM>;
  using Pipe = PipeKC<T, F, F, SBK, SBM, SBM, LA, LA>;
  static_assert(Pipe::LD == LD, "image stride");
  __shared__ __align__(16) T lds[Pipe::LDS_ELEMS];
  // wave as a scalar: `wanted` below gates MFMAs, and an MFMA does not honour EXEC — the gate must be a real branch
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  vr.A += (int64_t)b * strideA;
  if (vr.X) vr.X += (int64_t)b * strideX;
  const int64_t n = vr.n;
  const int64_t r0 = kend + (int64_t)tr * SBM;  // first virtual row of the tile
  const int64_t q0 = kend + (int64_t)tc * SBM;  // first column of the tile (< n)
  const bool diag_tile = tr == tc;
  const int wm = (wave >> 1) * 64, wn = (wave & 1) * 64;

  LA la, lb;
  la.init([&](int s, int64_t &rg) -> const T * { rg = r0 + s; return vr.ptr(r0 + s); }, vr.A, tid);
  lb.init([&](int s, int64_t &rg) -> const T * { rg = q0 + s; return (q0 + s < n) ? vr.A + (q0 + s) * vr.lda : nullptr; }, vr.A, tid);
  // the accumulators start from -C (its loads are in flight together with the first operand
  // tile):  acc = -C + P_r P_c^T, stored back negated.  A tile whose 128 rows are all rows of A
  // (no padding rows, no rows of X) and whose columns are all < n goes through TileIO.
  typename M::acc_t acc[F][F];
  const bool interior = r0 + SBM <= n && q0 + SBM <= n && TileIO<T>::usable(vr.lda, 64);
  TileIO<T> io;
  if (interior) {
    io.init(vr.A + (r0 + wm) * vr.lda + q0 + wn, vr.lda, lane);
#pragma unroll
    for (int fm = 0; fm < F; ++fm)
#pragma unroll
      for (int fn = 0; fn < F; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) acc[fm][fn][r] = io.load(fm, fn, r);
  } else {
#pragma unroll
    for (int fm = 0; fm < F; ++fm)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const T *rp = vr.ptr(r0 + wm + fm * M::TM + M::row(lane, r));
#pragma unroll
        for (int fn = 0; fn < F; ++fn) {
          const int64_t c = q0 + wn + fn * M::TM + M::col(lane);
          acc[fm][fn][r] = (rp != nullptr && c < n) ? rp[c] : vr.A[0];
        }
      }
  }
#pragma unroll
  for (int fm = 0; fm < F; ++fm)
#pragma unroll
    for (int fn = 0; fn < F; ++fn)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) acc[fm][fn][r] = -acc[fm][fn][r];
  Pipe pipe;
  // in a diagonal tile the wave at (0,1) only holds entries above the diagonal: it stages operands
  // and keeps the barriers but issues no MFMA
  const bool wanted = !(diag_tile && wm == 0 && wn == 64);
  pipe.run(la, lb, lds, cP, kend, wm, wn, lane, tid, acc, wanted);
  if (!wanted) return;
  if (interior) {
#pragma unroll
    for (int fm = 0; fm < F; ++fm)
#pragma unroll
      for (int fn = 0; fn < F; ++fn)
#pragma unroll
        for (int r = 0; r < M::NACC; ++r) io.store(fm, fn, r, -acc[fm][fn][r]);
    return;
  }
#pragma unroll
  for (int fm = 0; fm < F; ++fm)
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) {
      T *rp = vr.ptr(r0 + wm + fm * M::TM + M::row(lane, r));
#pragma unroll
      for (int fn = 0; fn < F; ++fn) {
        const int64_t c = q0 + wn + fn * M::TM + M::col(lane);
        if (rp != nullptr && c < n) rp[c] = -acc[fm][fn][r];
      }
    }
}


// ---- the same trailing update on the bf16 matrix pipe (float32, whole outer blocks, no extra rows) ----
// C[r][c] -= sum_k P[r][k] P[c][k] with P read as the three bf16 planes the panel launches wrote
// (pieces sum to the float exactly; six piece products per float product, f32 accumulation: b3.hpp),
// 256 x 128 tiles, one workgroup per CU.  The accumulators start from -C as above.  Tiles that cross
// the diagonal (tc >= 2 tr) compute their upper part too and mask it in the store.
// Tile p of a matrix -> (tr, tc), tc <= 2 tr + 1: row tr holds 2 tr + 2 tiles.
// Delayed update: the outer blocks go in groups of D = b3_group(n).  After a block inside a group only the NEXT
// 256 columns are updated (narrow map), with every column of the group so far (K = 256, 512, ...); after the
// group's last block everything right of it with K = 256 D — the same products accumulated in the same order
// (f32 accumulators stored and reloaded between launches lose nothing: the factor is bit-identical to D = 1),
// but a tile's C load, pipeline fill and store (~6 us) are paid per 8 D k-tiles instead of per 8.
struct SyrkB3Map {
  int R, batch;   // tile rows of the trailing matrix (256 rows each; 128 with half)
  int64_t per;    // tiles per matrix: R (R + 1) (half: R (R + 1) / 2); narrow, the first 256 columns only: 2 R (2 R - 1)
  int narrow;
  int half;       // 128 x 128 tiles (PipeB3T<2>): twice the workgroups for launches whose 256-row tiles fill their
                  // last round of 256 CUs badly
  __host__ static int64_t tiles(int R, bool narrow, bool half) {
    if (half) return narrow ? 2 * (int64_t)R - 1 : (int64_t)R * (R + 1) / 2;
    return narrow ? 2 * (int64_t)R : (int64_t)R * (R + 1);
  }
  __host__ int64_t grid() const {
    return batch >= 8 ? 8 * (int64_t)((batch + 7) / 8) * per : (int64_t)batch * per;
  }
  __device__ __forceinline__ bool decode(int64_t id, int &b, int &tr, int &tc) const {
    int64_t p;
    if (batch >= 8) {  // XCD x owns the matrices b = x (mod 8), as in SyrkMap
      const int xcd = (int)(id & 7);
      const int64_t k = id >> 3;
      const int64_t mb = k / per;
      b = xcd + 8 * (int)mb;
      if (b >= batch) return false;
      p = k - mb * per;
    } else {
      b = (int)(id / per);
      p = id - (int64_t)b * per;
    }
    if (half) {  // square tiles: tc <= tr
      if (narrow) {  // (0,0), (1,0), (1,1), (2,0), (2,1), ...
        tr = (int)((p + 1) >> 1);
        tc = p == 0 ? 0 : (int)((p + 1) & 1);
        return true;
      }
      int r = (int)((sqrtf(8.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
      while ((int64_t)r * (r + 1) / 2 > p) --r;
      while ((int64_t)(r + 1) * (r + 2) / 2 <= p) ++r;
      tr = r;
      tc = (int)(p - (int64_t)r * (r + 1) / 2);
      return true;
    }
    if (narrow) {
      tr = (int)(p >> 1);
      tc = (int)(p & 1);
      return true;
    }
    int r = (int)((sqrtf(4.0f * (float)p + 1.0f) - 1.0f) * 0.5f);
    while ((int64_t)r * (r + 1) > p) --r;
    while ((int64_t)(r + 1) * (r + 2) <= p) ++r;
    tr = r;
    tc = (int)(p - (int64_t)r * (r + 1));
    return true;
  }
};

// NP = 2: the planes hold L * s (fp16 x 2, b3.hpp); the accumulators start from -C s^2 and are multiplied by
// 1 / s^2 before the store (powers of two: exact).
template <int NP, int WFM, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void potrf_syrk_b3_kernel(float *A, int64_t lda,
                                                                    int64_t strideA, B3Out o3,
                                                                    int64_t cP, int K, int64_t kend,
                                                                    SyrkB3Map map) {
  using M = Mfma<float>;
  using PipeB3 = PipeB3T<WFM, NP, NW>;
  constexpr int TM = PipeB3::TM;  // 256 (WFM = 4, or WFM = 2 on eight waves) or 128 (WFM = 2 on four: map.half)
  __shared__ __align__(16) unsigned char lds[PipeB3::LDS_BYTES];
  int b, tr, tc;
  if (!map.decode(blockIdx.x, b, tr, tc)) return;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  float *Ab = A + (int64_t)b * strideA;
  const __bf16 *Pb = o3.p + (int64_t)b * o3.stride + (cP - o3.col0);  // row 0, first column of the K panel columns
  const int64_t r0 = kend + (int64_t)tr * TM, q0 = kend + (int64_t)tc * B3_TN;
  const int wm = (wave >> 1) * (32 * WFM), wn = (wave & 1) * 64;
  PipeB3 P;
  P.init(Pb + r0 * o3.ld, o3.ld, o3.plane, Pb + q0 * o3.ld, o3.ld, o3.plane, tid, wm, wn);
  TileIO<float> io;
  io.init(Ab + (r0 + wm) * lda + q0 + wn, lda, lane);
  // (tried: the C tile through LDS with coalesced 16-byte loads and the first operand tile requested
  // behind them — 8.7 instead of 8.3 ms per 32 x 4096 batch: the tile's time is its k loop, 8 k-tiles at
  // ~3 us, not the C traffic)
#pragma unroll
  for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
    for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) P.acc[fm][fn][r] = io.load(fm, fn, r);
#pragma unroll
  for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
    for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) P.acc[fm][fn][r] = -P.acc[fm][fn][r];
  if constexpr (NP == 2) {
    const float sc = o3.scale[b], s2 = sc * sc;
#pragma unroll
    for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
      for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) P.acc[fm][fn][r] *= s2;
  }
  P.run(lds, 0, K);
  if constexpr (NP == 2) {
    const float is = 1.0f / o3.scale[b], is2 = is * is;
#pragma unroll
    for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
      for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) P.acc[fm][fn][r] *= is2;
  }
  if (TM == 256 ? tc < 2 * tr : tc < tr) {  // wholly below the diagonal
#pragma unroll
    for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
      for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) io.store(fm, fn, r, -P.acc[fm][fn][r]);
    return;
  }
#pragma unroll
  for (int fm = 0; fm < PipeB3::WFM; ++fm)
#pragma unroll
    for (int fn = 0; fn < PipeB3::WFN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int64_t row = r0 + wm + fm * 32 + M::row(lane, r), col = q0 + wn + fn * 32 + M::col(lane);
        if (col <= row) Ab[row * lda + col] = -P.acc[fm][fn][r];
      }
}

// ---- finalize: zero the strict upper triangle; NaN-fill failed matrices -------------------------------
template <typename T>
__global__ void potrf_finalize_kernel(T *A, int64_t lda, int64_t strideA, int64_t n, int64_t npad,
                                      const T *dwork, const int32_t *info, int flags, B3Out o3) {
  const int b = blockIdx.z;
  T *Ab = A + (int64_t)b * strideA;
  const bool failed = (flags & GPVAE_POTRF_NAN_ON_FAIL) && info[b] != 0;
  const bool zero_up = flags & GPVAE_POTRF_ZERO_UPPER;
  const int64_t i0 = (int64_t)blockIdx.y * NB;
  // without flags only the diagonal blocks are touched: one 64 x 4-thread block per diagonal block
  // (threadIdx.y strides the rows; with flags blockDim.y == 1)
  const int64_t j = flags ? (int64_t)blockIdx.x * blockDim.x + threadIdx.x : i0 + threadIdx.x;
  const int ry = threadIdx.y, rs = blockDim.y;
  if (j >= n) return;
  const int64_t i1 = (i0 + NB < n) ? i0 + NB : n;
  const bool in_diag = j >= i0 && j < i0 + NB;  // this column crosses the strip's diagonal block
  if (!failed && !in_diag && (!zero_up || j < i0)) return;  // nothing to do for this column strip
  const T *l11 = dwork + ((int64_t)b * (npad / NB) + blockIdx.y) * (NB * NB);
  for (int64_t i = i0 + ry; i < i1; i += rs) {
    if (failed) Ab[i * lda + j] = dnan<T>();
    else if (in_diag) Ab[i * lda + j] = l11[(i - i0) * NB + (j - i0)];  // L11 (upper part is zero)
    else if (j > i) Ab[i * lda + j] = T(0);
  }
  if constexpr (sizeof(T) == 4) {
    // factor mode of the bf16 planes: the diagonal blocks (parked in dwork) go out as planes too, and
    // the 64 x 64 block right of every EVEN diagonal block is cleared — the draw's k loop ends at the
    // 128-column tile that holds the diagonal, so that block is all it ever reads above the diagonal
    if (o3.p != nullptr && o3.whole && in_diag) {
      __bf16 *q = o3.p + (int64_t)b * o3.stride;
      const bool even = (blockIdx.y & 1) == 0 && j + NB < n;
      if (o3.np == 2) {
        _Float16 *q2 = reinterpret_cast<_Float16 *>(q);
        const float sc = o3.scale[b];
        for (int64_t i = i0 + ry; i < i1; i += rs) {
          _Float16 hh, ll;
          split2_f16(failed ? dnan<float>() : l11[(i - i0) * NB + (j - i0)] * sc, hh, ll);
          q2[i * o3.ld + j] = hh;
          q2[o3.plane + i * o3.ld + j] = ll;
          if (even) {
            q2[i * o3.ld + j + NB] = (_Float16)0.f;
            q2[o3.plane + i * o3.ld + j + NB] = (_Float16)0.f;
          }
        }
        return;
      }
      for (int64_t i = i0 + ry; i < i1; i += rs) {
        __bf16 hh, mm, ll;
        split3_bf16(failed ? dnan<float>() : l11[(i - i0) * NB + (j - i0)], hh, mm, ll);
        q[i * o3.ld + j] = hh;
        q[o3.plane + i * o3.ld + j] = mm;
        q[2 * o3.plane + i * o3.ld + j] = ll;
        if (even) {
          q[i * o3.ld + j + NB] = (__bf16)0.f;
          q[o3.plane + i * o3.ld + j + NB] = (__bf16)0.f;
          q[2 * o3.plane + i * o3.ld + j + NB] = (__bf16)0.f;
        }
      }
    }
  }
}

// fp16 x 2 factor: scale[b] = 2^e with sqrt(max_i K_ii) 2^e in (2^13, 2^14] — every |L_ij| <= sqrt(K_ii) — and its
// inverse for the consumer of the planes.  A matrix without a positive finite diagonal gets scale 1 (it fails).
__global__ __launch_bounds__(256) void potrf_h2_scale_kernel(const float *A, int64_t lda, int64_t strideA, int64_t n,
                                                             float *scale, float *inv_scale) {
  __shared__ float red[256];
  const int b = blockIdx.x;
  const float *Ab = A + (int64_t)b * strideA;
  float m = 0.f;
  bool bad = false;
  for (int64_t i = threadIdx.x; i < n; i += 256) {
    const float d = Ab[i * lda + i];
    bad = bad || !(d == d) || d > 3.0e38f;
    m = fmaxf(m, d);
  }
  red[threadIdx.x] = bad ? -1.f : m;
  __syncthreads();
  if (threadIdx.x == 0) {
    float mm = 0.f;
    bool any_bad = false;
    for (int t = 0; t < 256; ++t) {
      any_bad = any_bad || red[t] < 0.f;
      mm = fmaxf(mm, red[t]);
    }
    const float sc = (any_bad || !(mm > 0.f)) ? 1.0f : h2_scale(sqrtf(mm));
    scale[b] = sc;
    inv_scale[b] = 1.0f / sc;
  }
}

// compute units of the current device (rounds of one-per-CU workgroups are counted against it)
static inline int64_t cu_count() {
  static int cached[64] = {0};
  int dev = 0, v = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] > 0) return cached[dev];
  if (hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || v <= 0) return 256;
  cached[dev] = v;
  return v;
}

// workgroups of rows below an outer block from which potrf_rows_kernel takes them (GPVAE_POTRF_ROWS_MIN in the
// environment overrides: measurements)
static inline int64_t rows_split_min() {
  static const int64_t v = [] { const char *e = getenv("GPVAE_POTRF_ROWS_MIN"); return e ? atoll(e) : (int64_t)128; }();
  return v;
}

// workspace: [parked diagonal blocks | bf16 planes of the current outer panel (b3 path only)]
static inline int64_t l11_bytes(int64_t es, int64_t n, int batch) {
  return round_up((int64_t)batch * round_up(n, NB) * NB * es, 256);
}
static inline bool potrf_b3_ok(bool f32, int64_t n, int64_t mx) {
  return f32 && mx == 0 && n >= 2 * NBO && n % NBO == 0;
}

template <typename T, bool VECM>
static int run_potrf_v(hipStream_t st, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                     int64_t mx, void *X, int64_t ldx, int64_t strideX, int32_t *info, int flags,
                     void *work, B3Out ext = B3Out{nullptr, 0, 0, 0, 0, 3, nullptr}) {
  constexpr int NCH = PanelCfg<T>::NCH;
  VRows<T> vr{(T *)A, lda, n, (T *)X, ldx, round_up(n, NB), mx};
  const int64_t Mv = vr.npad + mx;
  // float32, whole 256-column outer blocks, no extra rows: trailing updates on the bf16 matrix pipe;
  // the panel launches then also write their rows as bf16 planes behind the parked diagonal blocks
  const bool b3 = potrf_b3_ok(sizeof(T) == 4, n, mx) && VECM && 128 * lda * 4 < ((int64_t)1 << 31);
  B3Out o3{nullptr, 0, 0, 0, 0};
  if (b3) {
    if (ext.p != nullptr) o3 = ext;  // the caller's whole-factor planes double as the panel operand
    else o3 = B3Out{(__bf16 *)((char *)work + l11_bytes(sizeof(T), n, batch)), b3_group(n) * NBO, vr.npad * b3_group(n) * NBO, 3 * vr.npad * b3_group(n) * NBO, 0, 3, nullptr};
  } else if (ext.p != nullptr) {
    set_error("gpvae_potrf_b3: needs 16-byte aligned rows (lda %% 4 == 0) and 128*lda*4 < 2^31");
    return -5;
  }
  if (hipMemsetAsync(info, 0, sizeof(int32_t) * batch, st) != hipSuccess) {
    set_error("gpvae_potrf: hipMemsetAsync(info) failed");
    return -12;
  }
  const int D = b3 ? b3_group(n) : mfma_group(n);
  const int64_t ncu = cu_count();
  for (int64_t cJ = 0; cJ < n; cJ += NBO) {
    const int64_t kend = (cJ + NBO < n) ? cJ + NBO : n;
    const int jj = (int)((cJ / NBO) % D);          // position in the group of D outer blocks (b3 path)
    const int64_t cP = cJ - (int64_t)jj * NBO;     // first column of the group
    if (b3 && !o3.whole) o3.col0 = cP;
    // planes route: with enough rows below the outer block to hold a good part of the chip, the four launches cover
    // the block's own rows only and ONE potrf_rows_kernel launch the rows below it
    const int64_t rows_wgs = (n - kend) / (NB * NCH) * batch;
    const bool split = b3 && (flags & GPVAE_POTRF_SHARE_CHIP) && kend - cJ == NBO && rows_wgs >= rows_split_min();
    const int64_t rlim = split ? kend : INT64_MAX;
    for (int64_t c0 = cJ; c0 < kend; c0 += NB) {
      const int64_t mvl = split ? kend : Mv;
      const int64_t below = mvl - (c0 + NB);
      const int64_t nwg = below > 0 ? cdiv(below, (int64_t)NB * NCH) : 1;
      if constexpr (sizeof(T) == 4 && VECM) {
        if (b3) {
          if (o3.np == 2)
            hipLaunchKernelGGL((potrf_panel_kernel<T, VECM, 2>), dim3((unsigned)nwg, (unsigned)batch),
                               dim3(NTHREADS), 0, st, vr, strideA, strideX, c0, cJ, info, (T *)work, o3, g_stamps, rlim);
          else
            hipLaunchKernelGGL((potrf_panel_kernel<T, VECM, 3>), dim3((unsigned)nwg, (unsigned)batch),
                               dim3(NTHREADS), 0, st, vr, strideA, strideX, c0, cJ, info, (T *)work, o3, g_stamps, rlim);
          continue;
        }
      }
      if constexpr (sizeof(T) == 8) {
        // three chunks per workgroup where that saves a round of workgroups (one per CU): PanelCfg<double>
        const int64_t nwg3 = below > 0 ? cdiv(below, (int64_t)NB * 3) : 1;
        if (cdiv(nwg3 * batch, ncu) < cdiv(nwg * batch, ncu)) {
          hipLaunchKernelGGL((potrf_panel_kernel<T, VECM, 0, 3>), dim3((unsigned)nwg3, (unsigned)batch),
                             dim3(NTHREADS), 0, st, vr, strideA, strideX, c0, cJ, info, (T *)work, o3, g_stamps, rlim);
          continue;
        }
      }
      hipLaunchKernelGGL((potrf_panel_kernel<T, VECM, 0>), dim3((unsigned)nwg, (unsigned)batch),
                         dim3(NTHREADS), 0, st, vr, strideA, strideX, c0, cJ, info, (T *)work,
                         o3, g_stamps, rlim);
    }
    if constexpr (sizeof(T) == 4) {
      if (split) {
        const dim3 grid((unsigned)((n - kend) / (NB * NCH)), (unsigned)batch);
        if (o3.np == 2)
          hipLaunchKernelGGL((potrf_rows_kernel<2>), grid, dim3(NTHREADS), 0, st, (float *)A, lda, strideA, cJ,
                             (const float *)work, vr.npad, o3);
        else
          hipLaunchKernelGGL((potrf_rows_kernel<3>), grid, dim3(NTHREADS), 0, st, (float *)A, lda, strideA, cJ,
                             (const float *)work, vr.npad, o3);
      }
    }
    if constexpr (sizeof(T) == 4) {
      if (b3 && kend < n) {
        const int R = (int)((n - kend) / B3_TM);
        // inside a group: the next 256 columns only, with every column of the group so far (with one block
        // left that is the whole update); after its last block: everything right of the group
        const bool narrow = jj + 1 < D && R > 1;
        // 256 x 128 or 128 x 128 tiles: whichever needs less time in rounds of 256 workgroups (one per CU); a
        // 128-row tile takes 0.53 of a 256-row one (half the MFMAs per k-tile against two thirds of the operand bytes)
        SyrkB3Map map{R, batch, SyrkB3Map::tiles(R, narrow, false), narrow ? 1 : 0, 0};
        const SyrkB3Map hmap{2 * R, batch, SyrkB3Map::tiles(2 * R, narrow, true), narrow ? 1 : 0, 1};
        if (0.53 * (double)cdiv(hmap.grid(), (int64_t)256) < (double)cdiv(map.grid(), (int64_t)256)) map = hmap;
        const int K = (jj + 1) * NBO;
        const dim3 grid((unsigned)map.grid());
#define GPVAE_SYRK_B3(NPV, WFMV, NWV)                                                                         \
  hipLaunchKernelGGL((potrf_syrk_b3_kernel<NPV, WFMV, NWV>), grid, dim3(64 * NWV), 0, st, (float *)A, lda, strideA, o3, \
                     cP, K, kend, map)
        if (o3.np == 2) {
          if (map.half) GPVAE_SYRK_B3(2, 2, 4); else GPVAE_SYRK_B3(2, 2, 8);
        } else {
          if (map.half) GPVAE_SYRK_B3(3, 2, 4); else GPVAE_SYRK_B3(3, 2, 8);
        }
#undef GPVAE_SYRK_B3
        continue;
      }
    }
    if (kend < n) {
      const int rt = (int)cdiv(Mv - kend, SBM), ct = (int)cdiv(n - kend, SBM);
      const bool narrow = jj + 1 < D && ct > 2;  // as on the bf16 x 3 path: the next 256 columns only
      const SyrkMap map{rt, ct, batch, narrow ? 2 * (int64_t)rt - 1 : SyrkMap::tiles(rt, ct), narrow ? 1 : 0};
      hipLaunchKernelGGL((potrf_syrk_kernel<T, VECM>), dim3((unsigned)map.grid()), dim3(NTHREADS), 0, st,
                         vr, strideA, strideX, cP, kend, map);
    }
  }
  const int fflags = flags & (GPVAE_POTRF_ZERO_UPPER | GPVAE_POTRF_NAN_ON_FAIL);  // what the finalize pass acts on
  hipLaunchKernelGGL((potrf_finalize_kernel<T>),
                     dim3(fflags ? (unsigned)cdiv(n, 256) : 1u, (unsigned)cdiv(n, NB), (unsigned)batch),
                     fflags ? dim3(256) : dim3(NB, 4), 0, st, (T *)A, lda, strideA, n, vr.npad, (const T *)work, info,
                     fflags, o3);
  GPVAE_CHECK_LAUNCH("gpvae_potrf");
  return 0;
}

// 16-byte loads need every row of A (and X) 16-byte aligned; otherwise the scalar-load build runs
template <typename T>
static int run_potrf(hipStream_t st, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                     int64_t mx, void *X, int64_t ldx, int64_t strideX, int32_t *info, int flags,
                     void *work, B3Out ext = B3Out{nullptr, 0, 0, 0, 0, 3, nullptr}) {
  constexpr int VEC = VecT<T>::N;
  const bool vec = kc_vec_ok<T>(A, lda) && (strideA % VEC == 0) &&
                   (mx == 0 || (kc_vec_ok<T>(X, ldx) && (strideX % VEC == 0)));
  return vec ? run_potrf_v<T, true>(st, n, batch, A, lda, strideA, mx, X, ldx, strideX, info, flags,
                                    work, ext)
             : run_potrf_v<T, false>(st, n, batch, A, lda, strideA, mx, X, ldx, strideX, info,
                                     flags, work, ext);
}

}  // namespace gpvae

using namespace gpvae;

#ifdef GPVAE_DEV
extern "C" void gpvae_dev_set_stamps(void *p) { gpvae::g_stamps = (unsigned long long *)p; }
#endif

extern "C" int64_t gpvae_potrf_workspace_bytes(int dtype, int64_t n, int64_t m_extra, int batch) {
  // one dense 64x64 block per diagonal block and matrix: the factored diagonal blocks wait here
  // until the last launch copies them over A (see potrf_panel_kernel's store phase)
  // ... and, for float32 problems that take the bf16 x 3 trailing update (n a multiple of 256, no extra
  // rows), the current group of outer panels as three bf16 planes: batch * 3 * n * 256 b3_group(n) * 2 bytes
  if (n <= 0 || batch <= 0) return 0;
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  int64_t bytes = l11_bytes(es, n, batch);
  if (potrf_b3_ok(dtype == GPVAE_F32, n, m_extra)) bytes += (int64_t)batch * 3 * round_up(n, NB) * b3_group(n) * NBO * 2;
  return bytes;
}

extern "C" int gpvae_potrf(void *stream, int dtype, int64_t n, int batch, void *A, int64_t lda,
                           int64_t strideA, int64_t m_extra, void *X, int64_t ldx,
                           int64_t strideX, int32_t *info, int flags, void *work) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_potrf: bad dtype %d", dtype);
  GPVAE_REQUIRE(n >= 0, 3, "gpvae_potrf: n < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 4, "gpvae_potrf: batch out of range [0,65535]");
  GPVAE_REQUIRE(m_extra >= 0, 8, "gpvae_potrf: m_extra < 0");
  if (n == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(A, 5, "gpvae_potrf: A is NULL");
  GPVAE_REQUIRE(lda >= n, 6, "gpvae_potrf: lda < n");
  GPVAE_REQUIRE(batch == 1 || strideA >= n * lda, 7, "gpvae_potrf: strideA < n*lda");
  if (m_extra > 0) {
    GPVAE_REQUIRE(X, 9, "gpvae_potrf: X is NULL with m_extra > 0");
    GPVAE_REQUIRE(ldx >= n, 10, "gpvae_potrf: ldx < n");
    GPVAE_REQUIRE(batch == 1 || strideX >= m_extra * ldx, 11, "gpvae_potrf: strideX too small");
  }
  GPVAE_REQUIRE(info, 12, "gpvae_potrf: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 14,
                "gpvae_potrf: work is NULL or not 16-byte aligned (gpvae_potrf_workspace_bytes)");
  {  // the trailing update is one 1-D launch over all tiles of all matrices
    const int rt = (int)cdiv(round_up(n, NB) + m_extra, SBM), ct = (int)cdiv(n, SBM);
    const SyrkMap map{rt, ct, batch, SyrkMap::tiles(rt, ct), 0};
    GPVAE_REQUIRE(map.grid() <= 0x7fffffffLL, 4, "gpvae_potrf: batch x tiles exceeds one launch; split the batch");
    GPVAE_REQUIRE((int64_t)batch * 3 * round_up(n, NB) * b3_group(n) * NBO < ((int64_t)1 << 40), 4, "gpvae_potrf: batch too large");
  }
  hipStream_t st = (hipStream_t)stream;
  if (m_extra == 0) X = nullptr;
  return dtype == GPVAE_F32
             ? run_potrf<float>(st, n, batch, A, lda, strideA, m_extra, X, ldx, strideX, info, flags,
                                work)
             : run_potrf<double>(st, n, batch, A, lda, strideA, m_extra, X, ldx, strideX, info,
                                 flags, work);
}

// float32 factorisation that also leaves the factor as bf16 x 3 planes (the operand gpvae_draw_b3 wants):
// the planes are written by the panel launches as they go and serve as the operand of the bf16 x 3
// trailing updates, so no separate split pass over L is needed.
extern "C" int gpvae_potrf_b3_supported(int64_t n) { return potrf_b3_ok(true, n, 0) ? 1 : 0; }

extern "C" int64_t gpvae_potrf_b3_workspace_bytes(int64_t n, int batch) {
  if (n <= 0 || batch <= 0) return 0;
  return l11_bytes(4, n, batch);
}

extern "C" int gpvae_potrf_b3(void *stream, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                              int32_t *info, int flags, void *work, void *L3, int64_t ld3,
                              int64_t plane, int64_t stride3) {
  GPVAE_REQUIRE(n >= 0, 2, "gpvae_potrf_b3: n < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 3, "gpvae_potrf_b3: batch out of range [0,65535]");
  if (n == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(potrf_b3_ok(true, n, 0), 2, "gpvae_potrf_b3: n must be a multiple of 256 and >= 512 (gpvae_potrf_b3_supported)");
  GPVAE_REQUIRE(A && aligned16(A), 4, "gpvae_potrf_b3: A is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(lda >= n && lda % 4 == 0, 5, "gpvae_potrf_b3: lda < n or not a multiple of 4");
  GPVAE_REQUIRE(batch == 1 || (strideA >= n * lda && strideA % 4 == 0), 6, "gpvae_potrf_b3: strideA < n*lda or not a multiple of 4");
  GPVAE_REQUIRE(info, 7, "gpvae_potrf_b3: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 9, "gpvae_potrf_b3: work is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(L3 && aligned16(L3), 10, "gpvae_potrf_b3: L3 is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(ld3 >= n && ld3 % 8 == 0, 11, "gpvae_potrf_b3: ld3 < n or not a multiple of 8");
  GPVAE_REQUIRE(plane >= n * ld3 && plane % 8 == 0, 12, "gpvae_potrf_b3: plane < n*ld3 or not a multiple of 8");
  GPVAE_REQUIRE(batch == 1 || (stride3 >= 3 * plane && stride3 % 8 == 0), 13, "gpvae_potrf_b3: stride3 < 3*plane or not a multiple of 8");
  GPVAE_REQUIRE(PipeB3::usable(ld3, ld3), 11, "gpvae_potrf_b3: ld3 too large");
  {
    const int rt = (int)cdiv(round_up(n, NB), SBM), ct = (int)cdiv(n, SBM);
    const SyrkMap map{rt, ct, batch, SyrkMap::tiles(rt, ct), 0};
    GPVAE_REQUIRE(map.grid() <= 0x7fffffffLL, 3, "gpvae_potrf_b3: batch x tiles exceeds one launch; split the batch");
  }
  return run_potrf<float>((hipStream_t)stream, n, batch, A, lda, strideA, 0, nullptr, 0, 0, info, flags, work,
                          B3Out{(__bf16 *)L3, ld3, plane, stride3, 1, 3, nullptr});
}

// The same with the factor left as fp16 x 2 planes of L * scale[b] (the operand gpvae_draw_h2 wants) and the
// trailing updates on those planes: half the matrix work of bf16 x 3, operands rounded to 22 bits.
extern "C" int64_t gpvae_potrf_h2_workspace_bytes(int64_t n, int batch) {
  if (n <= 0 || batch <= 0) return 0;
  return l11_bytes(4, n, batch) + round_up((int64_t)batch * 4, 256);
}

extern "C" int gpvae_potrf_h2(void *stream, int64_t n, int batch, void *A, int64_t lda, int64_t strideA,
                              int32_t *info, int flags, void *work, void *L2, int64_t ld2,
                              int64_t plane, int64_t stride2, float *inv_scale) {
  GPVAE_REQUIRE(n >= 0, 2, "gpvae_potrf_h2: n < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 3, "gpvae_potrf_h2: batch out of range [0,65535]");
  if (n == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(potrf_b3_ok(true, n, 0), 2, "gpvae_potrf_h2: n must be a multiple of 256 and >= 512 (gpvae_potrf_b3_supported)");
  GPVAE_REQUIRE(A && aligned16(A), 4, "gpvae_potrf_h2: A is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(lda >= n && lda % 4 == 0, 5, "gpvae_potrf_h2: lda < n or not a multiple of 4");
  GPVAE_REQUIRE(batch == 1 || (strideA >= n * lda && strideA % 4 == 0), 6, "gpvae_potrf_h2: strideA < n*lda or not a multiple of 4");
  GPVAE_REQUIRE(info, 7, "gpvae_potrf_h2: info is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 9, "gpvae_potrf_h2: work is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(L2 && aligned16(L2), 10, "gpvae_potrf_h2: L2 is NULL or not 16-byte aligned");
  GPVAE_REQUIRE(ld2 >= n && ld2 % 8 == 0, 11, "gpvae_potrf_h2: ld2 < n or not a multiple of 8");
  GPVAE_REQUIRE(plane >= n * ld2 && plane % 8 == 0, 12, "gpvae_potrf_h2: plane < n*ld2 or not a multiple of 8");
  GPVAE_REQUIRE(batch == 1 || (stride2 >= 2 * plane && stride2 % 8 == 0), 13, "gpvae_potrf_h2: stride2 < 2*plane or not a multiple of 8");
  GPVAE_REQUIRE(inv_scale, 14, "gpvae_potrf_h2: inv_scale is NULL");
  GPVAE_REQUIRE((PipeB3T<4, 2>::usable(ld2, ld2)), 11, "gpvae_potrf_h2: ld2 too large");
  {
    const int rt = (int)cdiv(round_up(n, NB), SBM), ct = (int)cdiv(n, SBM);
    const SyrkMap map{rt, ct, batch, SyrkMap::tiles(rt, ct), 0};
    GPVAE_REQUIRE(map.grid() <= 0x7fffffffLL, 3, "gpvae_potrf_h2: batch x tiles exceeds one launch; split the batch");
  }
  float *scale = (float *)((char *)work + l11_bytes(4, n, batch));
  hipLaunchKernelGGL(potrf_h2_scale_kernel, dim3((unsigned)batch), dim3(256), 0, (hipStream_t)stream, (const float *)A,
                     lda, strideA, n, scale, inv_scale);
  return run_potrf<float>((hipStream_t)stream, n, batch, A, lda, strideA, 0, nullptr, 0, 0, info, flags, work,
                          B3Out{(__bf16 *)L2, ld2, plane, stride2, 1, 2, scale});
}
