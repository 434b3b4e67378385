// gemm_core.hpp — building blocks of the f32/f64 MFMA tile GEMMs used by potrf, draw, krig, vae.
//
// Shape of every tile product: 256 threads (4 waves); operand tiles are staged global -> registers
// -> LDS in a k-major image  As[k][row]  (row contiguous), so a wave's MFMA fragment read
// (lane l needs A[row = l % TM][k = l / TM]) is one conflict-free scalar LDS read with consecutive
// lanes on consecutive addresses.  f32 MFMA (v_mfma_f32_32x32x2_f32) runs at 64 cycles per
// instruction, so one LDS read per operand per MFMA is far from the LDS limit
// (cdna_hip_programming.md §3 "FP32-input MFMA").  The next k-tile's global loads are issued before
// the current tile's MFMAs (register-staged double buffering, guide §6 G15).
#pragma once
#include "common.hpp"

namespace gpvae {

constexpr int NTHREADS = 256;

template <typename T> struct VecT;
template <> struct VecT<float> {
  static constexpr int N = 4;
  using type = float4;
};
template <> struct VecT<double> {
  static constexpr int N = 2;
  using type = double2;
};

// LDS row padding (elements) of an image filled by LoaderMC: rows stay 16-byte aligned for its
// vector stores.
template <typename T> struct LdsPad { static constexpr int V = 16 / sizeof(T); };
// Row padding of an image filled ONLY by LoaderKC (transposing scalar stores): an odd stride makes
// the 32 lanes of a store group (8 k-chunks x 4 rows) hit 32 different banks; fragment reads are
// consecutive-lane, conflict free for any stride.
constexpr int KC_PAD = 1;

// ---- loader: source contiguous along K:  element(r, k) = rowptr[r][k] ------------------------------
// ROWS x BK tile; per-thread row pointers are fixed over the k loop.  A row that does not exist
// reads zero: its pointer is replaced by `safe` (any row that can be read over the same k range)
// and the value is discarded with a select, so the common path has no divergent branch.
// `vec` (wave-uniform): every row pointer and k0 are 16-byte aligned => one 16-byte load per
// vector.  TRI: additionally zero elements with k > rowg (global row index) — the draw kernel's
// lower-triangle mask, only evaluated on k-tiles that reach the diagonal.
template <typename T, int ROWS, int BK, bool TRI = false>
struct LoaderKC {
  static constexpr int VEC = VecT<T>::N;
  static constexpr int VPR = BK / VEC;  // vectors per row
  static constexpr int NV = (ROWS * VPR + NTHREADS - 1) / NTHREADS;
  static constexpr bool EXACT = (ROWS * VPR) % NTHREADS == 0;
  using V = typename VecT<T>::type;
  const T *rp[NV];
  bool ok[NV];
  int64_t rowg[NV];
  int64_t rowg_min;  // smallest global row of the tile (TRI)
  T regs[NV][VEC];

  template <class RowPtr>
  __device__ __forceinline__ void init(const RowPtr &rowptr, const T *safe, int tid) {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int r = v / VPR;
      rowg[p] = 0;
      const T *q = (EXACT || r < ROWS) ? rowptr(r, rowg[p]) : nullptr;
      ok[p] = q != nullptr;
      rp[p] = ok[p] ? q : safe;
    }
    int64_t dummy;
    (void)rowptr(0, dummy);
    rowg_min = dummy;
  }
  // loads columns [k0, k0+BK) clipped to [kbeg, kend).  The fast path only ISSUES the loads: the
  // registers are not touched again until store(), so the wait for them lands after the MFMAs of
  // the tile in flight (masking of absent rows / the triangle happens in store()).
  int64_t k0_;
  bool fast_;
  __device__ __forceinline__ void load(int64_t k0, int64_t kbeg, int64_t kend, bool vec, int tid) {
    const bool interior = k0 >= kbeg && k0 + BK <= kend;  // wave-uniform
    k0_ = k0;
    fast_ = vec && interior;
    if (fast_) {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int v = tid + p * NTHREADS;
        const int64_t k = k0 + (v % VPR) * VEC;
        const V t = *reinterpret_cast<const V *>(rp[p] + k);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) regs[p][e] = tp[e];
      }
    } else {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int v = tid + p * NTHREADS;
        const int64_t k = k0 + (v % VPR) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const bool in = ok[p] && k + e >= kbeg && k + e < kend;
          regs[p][e] = in ? rp[p][in ? k + e : kbeg] : T(0);
        }
      }
    }
  }
  // lds: [BK][LD] k-major image;  NEG stores the negated tile (C - A B^T as an MFMA accumulate)
  template <int LD, bool NEG = false> __device__ __forceinline__ void store(T *lds, int tid) const {
    const bool tri = TRI && (k0_ + BK - 1 > rowg_min);  // wave-uniform: tile reaches the diagonal
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int r = v / VPR, kc = (v % VPR) * VEC;
      if (EXACT || r < ROWS) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          T val = regs[p][e];
          if (!ok[p]) val = T(0);
          if (TRI && tri && (k0_ + kc + e > rowg[p])) val = T(0);
          lds[(kc + e) * LD + r] = NEG ? -val : val;
        }
      }
    }
  }
};

// k-tile depth of the 128x128 products: 32 for f32 (a full 128-byte line per row), 16 for f64
template <typename T> struct TileK { static constexpr int BK = sizeof(T) == 4 ? 32 : 16; };

// All row pointers of a K-contiguous operand are 16-byte aligned at every multiple-of-VEC column.
template <typename T> static inline bool kc_vec_ok(const void *base, int64_t ld) {
  return base == nullptr || (aligned16(base) && (ld % VecT<T>::N) == 0);
}

// ---- loader: source contiguous along the tile's row index:  element(r, k) = base[k*ld + r] --------
// (the B operand of a row-major [K][N] matrix, or the A operand of a transposed product).
// ONES: row index `ones_row` reads 1 for every k — appends a bias/column-sum row to a product.
template <typename T, int ROWS, int BK, bool ONES = false>
struct LoaderMC {
  static constexpr int VEC = VecT<T>::N;
  static constexpr int VPK = ROWS / VEC;  // vectors per k-row
  static constexpr int NV = (BK * VPK + NTHREADS - 1) / NTHREADS;
  static constexpr bool EXACT = (BK * VPK) % NTHREADS == 0;
  using V = typename VecT<T>::type;
  T regs[NV][VEC];

  // rows [r0, r0+ROWS) clipped to < rend; k in [k0, k0+BK) clipped to [kbeg, kend).
  // vec (uniform): base and ld are 16-byte aligned, so a whole in-range vector can be one load.
  __device__ __forceinline__ void load(const T *base, int64_t ld, int64_t r0, int64_t rend,
                                       int64_t k0, int64_t kbeg, int64_t kend, bool vec, int tid,
                                       int64_t ones_row = -1) {
    const bool interior = vec && k0 >= kbeg && k0 + BK <= kend && r0 + ROWS <= rend &&
                          (r0 % VEC) == 0;  // wave-uniform
    if (interior) {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int v = tid + p * NTHREADS;
        const int kk = EXACT ? v / VPK : (v / VPK < BK ? v / VPK : BK - 1);
        const V t = *reinterpret_cast<const V *>(base + (k0 + kk) * ld + r0 + (v % VPK) * VEC);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) regs[p][e] = tp[e];
      }
    } else {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int v = tid + p * NTHREADS;
        const int kk = v / VPK;
        const int64_t r = r0 + (v % VPK) * VEC;
        const int64_t k = k0 + kk;
        const bool kin = (EXACT || kk < BK) && k >= kbeg && k < kend;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const bool in = kin && r + e < rend;
          regs[p][e] = in ? base[in ? k * ld + r + e : 0] : T(0);
          if (ONES && kin && r + e == ones_row) regs[p][e] = T(1);
        }
      }
    }
  }
  template <int LD> __device__ __forceinline__ void store(T *lds, int tid) const {
    static_assert((LD * sizeof(T)) % 16 == 0, "LDS rows must stay 16-byte aligned");
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int kk = v / VPK, rc = (v % VPK) * VEC;
      if (EXACT || kk < BK) {
        V t;
        T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) tp[e] = regs[p][e];
        *reinterpret_cast<V *>(lds + kk * LD + rc) = t;
      }
    }
  }
};

// ---- MFMA over one staged k-tile -------------------------------------------------------------------
// acc[fm][fn] += A[am0 + fm*TM ..][k] * B[k][bn0 + fn*TM ..], k over the BK rows of the LDS images.
// Fragments are read in groups of up to 8 k-steps BEFORE the group's MFMAs, so the LDS latency is
// paid once per group instead of once per k-step (hipcc otherwise waits lgkmcnt(0) per step).
template <typename T, int FM, int FN, int BK, int LDA, int LDB>
__device__ __forceinline__ void mma_ktile(const T *As, const T *Bs, int am0, int bn0, int lane,
                                          typename Mfma<T>::acc_t (&acc)[FM][FN]) {
  using M = Mfma<T>;
  constexpr int KS = BK / M::KI;
  constexpr int G = KS < 8 ? KS : 8;
  const int lr = lane % M::TM, kg = lane / M::TM;
  const T *ap = As + kg * LDA + am0 + lr;
  const T *bp = Bs + kg * LDB + bn0 + lr;
#pragma unroll
  for (int g0 = 0; g0 < KS; g0 += G) {
    T a[G][FM], b[G][FN];
#pragma unroll
    for (int g = 0; g < G; ++g) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) a[g][fm] = ap[(g0 + g) * M::KI * LDA + fm * M::TM];
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) b[g][fn] = bp[(g0 + g) * M::KI * LDB + fn * M::TM];
    }
    __builtin_amdgcn_sched_barrier(0);  // keep the group's reads ahead of its MFMAs
#pragma unroll
    for (int g = 0; g < G; ++g)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = M::mma(a[g][fm], b[g][fn], acc[fm][fn]);
    __builtin_amdgcn_sched_barrier(0);
  }
}

template <typename T, int FM, int FN>
__device__ __forceinline__ void zero_acc(typename Mfma<T>::acc_t (&acc)[FM][FN]) {
#pragma unroll
  for (int fm = 0; fm < FM; ++fm)
#pragma unroll
    for (int fn = 0; fn < FN; ++fn)
#pragma unroll
      for (int r = 0; r < Mfma<T>::NACC; ++r) acc[fm][fn][r] = T(0);
}

}  // namespace gpvae
