// gemm_core.hpp — building blocks of the f32/f64 MFMA tile GEMMs used by potrf, draw, krig, vae.
//
// Shape of every tile product: 256 threads (4 waves); operand tiles are staged global -> registers
// -> LDS, the next k-tile's global loads issued before the current tile's MFMAs (register-staged
// double buffering, guide §6 G15).
//
// What bounds these kernels (measured, tools/ub_mfma_valu.hip): on gfx950 an f32 MFMA occupies the
// SIMD's vector ALU — v_mfma_f32_32x32x2_f32 takes 64 cycles and NO other VALU instruction of
// either resident wave issues meanwhile (one wave streaming MFMAs + one wave streaming v_add: the
// times add up exactly; same inside one wave).  So time = 64 * #MFMA + ~4..8 * #VALU per SIMD and
// the only thing to optimise is the number of non-MFMA vector instructions per MFMA.  Hence two
// LDS images:
//   RM (row-major)  img[row][BK + RM_PAD]: filled by LoaderKC (source contiguous along k) with
//       16-byte ds_writes, no transpose; a fragment read is ONE 16-byte ds_read per 4 (f32) /
//       2 (f64) k-steps with a compile-time offset from one base register — zero address VALU.
//       The k values of a chunk of 8 are dealt to the MFMA lane groups as k = 8c + KV*g + j
//       (g = lane / TM, j = k-step inside the chunk); any assignment is valid as long as both
//       operands use the same one.  Row stride 36 dwords: the 16 lanes of a 16-byte access phase
//       land on 16 distinct 4-bank groups.
//   KM (k-major)  img[k][rows + pad]: filled by LoaderMC (source contiguous along rows);
//       scalar fragment reads, consecutive lanes on consecutive addresses.
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
// Row padding (elements) of a row-major image: stride BK + RM_PAD = 36 dwords for BK = 32 f32 /
// 16 f64.
template <typename T> struct RmPad { static constexpr int V = 16 / sizeof(T); };

// 16-byte buffer load: scalar resource (base address) + per-lane 32-bit byte offset + SCALAR byte
// offset -> the k advance of a loop costs one s_add and no vector instruction.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ i32x4_t raw_buffer_load_b128(i32x4_t rsrc, int voffset, int soffset,
                                        int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");
// a value the program knows to be wave-uniform, moved to scalar registers (the compiler treats the
// results of block-wide votes and anything derived from them as divergent)
__device__ __forceinline__ int64_t uniform64(int64_t v) {
  const uint32_t lo = __builtin_amdgcn_readfirstlane((int)(uint32_t)v);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((int)(uint32_t)((uint64_t)v >> 32));
  return (int64_t)(((uint64_t)hi << 32) | lo);
}
__device__ __forceinline__ i32x4_t make_buffer_rsrc(const void *p) {
  const uint64_t a = (uint64_t)uniform64((int64_t)reinterpret_cast<uint64_t>(p));
  i32x4_t r;
  r[0] = (int)(uint32_t)a;
  r[1] = (int)(uint32_t)(a >> 32);  // stride 0: raw buffer
  r[2] = -1;                        // num_records: no range check wanted
  r[3] = 0x00020000;                // gfx9 raw-buffer dword 3 (32-bit data format)
  return r;
}

// ---- accumulator-tile access without per-element vector arithmetic ---------------------------------
// A wave's FM x FN MFMA tiles live at  origin[(fm*TM + row(lane, r)) * ld + fn*TM + col(lane)].
// row(lane, r) = row(lane, 0) + row(0, r), so the address splits into a per-lane byte offset
// (one register, loop invariant) and a per-element SCALAR offset: element loads/stores are
// buffer ops with zero VALU instructions — the epilogue of one block then no longer crawls behind
// the MFMA stream of the block that shares its SIMDs.  Only for tiles that need no bounds checks
// and whose 64-row extent stays below 2 GiB (usable()).
__device__ float raw_buffer_load_f32(i32x4_t rsrc, int voffset, int soffset,
                                     int aux) __asm("llvm.amdgcn.raw.buffer.load.f32");
__device__ void raw_buffer_store_f32(float v, i32x4_t rsrc, int voffset, int soffset,
                                     int aux) __asm("llvm.amdgcn.raw.buffer.store.f32");
typedef int i32x2_t __attribute__((ext_vector_type(2)));
__device__ i32x2_t raw_buffer_load_b64(i32x4_t rsrc, int voffset, int soffset,
                                       int aux) __asm("llvm.amdgcn.raw.buffer.load.v2i32");
__device__ void raw_buffer_store_b64(i32x2_t v, i32x4_t rsrc, int voffset, int soffset,
                                     int aux) __asm("llvm.amdgcn.raw.buffer.store.v2i32");

template <typename T> struct TileIO {
  using M = Mfma<T>;
  i32x4_t rsrc;
  int voff;
  int ldb;  // leading dimension in bytes (scalar)
  __device__ static __forceinline__ bool usable(int64_t ld, int rows) {
    return (int64_t)rows * ld * (int64_t)sizeof(T) < ((int64_t)1 << 31);
  }
  // origin: wave-uniform pointer to the wave's first element
  __device__ __forceinline__ void init(const T *origin, int64_t ld, int lane) {
    rsrc = make_buffer_rsrc(origin);
    ldb = __builtin_amdgcn_readfirstlane((int)(ld * (int64_t)sizeof(T)));
    voff = M::row(lane, 0) * (int)(ld * (int64_t)sizeof(T)) + M::col(lane) * (int)sizeof(T);
  }
  __device__ __forceinline__ int soff(int fm, int fn, int r) const {
    return (fm * M::TM + M::row(0, r)) * ldb + fn * M::TM * (int)sizeof(T);
  }
  __device__ __forceinline__ T load(int fm, int fn, int r) const {
    if constexpr (sizeof(T) == 4) {
      return raw_buffer_load_f32(rsrc, voff, soff(fm, fn, r), 0);
    } else {
      const i32x2_t t = raw_buffer_load_b64(rsrc, voff, soff(fm, fn, r), 0);
      return __builtin_bit_cast(double, t);
    }
  }
  __device__ __forceinline__ void store(int fm, int fn, int r, T v) const {
    if constexpr (sizeof(T) == 4) {
      raw_buffer_store_f32(v, rsrc, voff, soff(fm, fn, r), 0);
    } else {
      raw_buffer_store_b64(__builtin_bit_cast(i32x2_t, v), rsrc, voff, soff(fm, fn, r), 0);
    }
  }
};

// ---- loader: source contiguous along K:  element(r, k) = rowptr[r][k] ------------------------------
// ROWS x BK tile; per-thread row pointers are fixed over the k loop.  Two paths:
//   generic  load()/store(): any row set (a row that does not exist reads through `safe` and is
//       zeroed), ragged end of the k range (clamped address, zeroed), TRI mask (zero k > global
//       row, the draw kernel's lower-triangle mask).  Branch-free loads: nothing touches the
//       registers until store(), so the wait lands after the MFMAs of the tile in flight.
//   fast  load_fast()/store_fast(): NO vector-ALU instruction at all (see the header: a VALU
//       instruction in this phase would wait for the partner wave's whole MFMA phase).  Loads are
//       buffer_load_dwordx4 with a scalar resource (tile origin), a loop-invariant 32-bit per-lane
//       byte offset and the k offset in a scalar register; stores are bare ds_write_b128.  Usable
//       for a k-tile when fast_ok (every row present and within 2 GiB of the first row; rows need
//       only 4-byte alignment, buffer loads run in unaligned mode) and the tile needs no masking:
//       k0 + BK <= fast_end(kend).
//   VECM (template): all row pointers and k0 are 16-byte aligned (lda % VEC == 0, aligned base);
//       otherwise VEC scalar loads.  NOTE: in VECM mode the last vector of a row may read up to
//       round_up(kend, VEC) <= lda: the buffer must span rows * lda elements.
template <typename T, int ROWS, int BK, bool VECM, bool TRI = false>
struct LoaderKC {
  static constexpr int VEC = VecT<T>::N;
  static constexpr int VPR = BK / VEC;  // vectors per row
  static constexpr int NV = (ROWS * VPR + NTHREADS - 1) / NTHREADS;
  static constexpr bool EXACT = (ROWS * VPR) % NTHREADS == 0;
  using V = typename VecT<T>::type;
  const T *rp[NV];   // row pointer + this thread's k offset inside a tile
  int off[NV];       // the same as a byte offset from the first row of the tile (fast path)
  i32x4_t rsrc;      // wave-uniform buffer resource: first row of the tile
  bool ok[NV];
  int64_t rowg[NV];
  int64_t rowg_min;  // smallest global row of the tile (TRI)
  bool all_ok;       // wave-uniform: every row of the tile exists
  bool fast_ok;      // wave-uniform: the fast path may be used
  bool kragged;      // wave-uniform: the fast path may take the ragged last k-tile
  int64_t k0_, kend_;
  T regs[NV][VEC];

  // kragged_ext > 0 (elements from the first row of the tile to the end of the array): the fast
  // path may also take the ragged last k-tile — the caller guarantees that the OTHER operand reads
  // as zero for k >= kend and that whatever lies behind a row's end is finite (the next row of the
  // same array); loads past the array return 0 through the buffer range check.
  template <class RowPtr>
  __device__ __forceinline__ void init(const RowPtr &rowptr, const T *safe, int tid,
                                       int64_t kragged_ext = 0) {
    kragged = kragged_ext > 0 && kragged_ext * (int64_t)sizeof(T) < ((int64_t)1 << 31);
    int64_t rg0;
    const T *base = rowptr(0, rg0);
    rowg_min = rg0;
    rsrc = make_buffer_rsrc(base);
    if (kragged) rsrc[2] = __builtin_amdgcn_readfirstlane((int)(kragged_ext * (int64_t)sizeof(T)));
    int mine = 1, fits = 1;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int r = v / VPR;
      rowg[p] = 0;
      const T *q = (EXACT || r < ROWS) ? rowptr(r, rowg[p]) : nullptr;
      ok[p] = q != nullptr;
      rp[p] = (ok[p] ? q : safe) + (v % VPR) * VEC;
      const int64_t d = reinterpret_cast<const char *>(rp[p]) - reinterpret_cast<const char *>(base);
      off[p] = (int)d;
      if (EXACT || r < ROWS) {
        if (!ok[p]) mine = 0;
        if (d < 0 || d >= ((int64_t)1 << 31)) fits = 0;
      }
    }
    // every thread of the block calls init
    all_ok = __syncthreads_and(mine) != 0;
    fast_ok = __syncthreads_and(fits) != 0 && all_ok && EXACT && base != nullptr;
  }
  // k-tiles [k0, k0+BK) with k0 + BK <= fast_end(kend) may use the fast path
  __device__ __forceinline__ int64_t fast_end(int64_t kend) const {
    if (!fast_ok || kend * (int64_t)sizeof(T) >= ((int64_t)1 << 31)) return INT64_MIN;
    if (TRI && rowg_min + 1 < kend) return rowg_min + 1;  // tiles that stay at or below the diagonal
    return kragged ? (kend + BK - 1) / BK * BK : kend;
  }
  __device__ __forceinline__ void load_fast(int64_t k0, int64_t kend) {
    k0_ = k0;  // scalars: a fast-loaded tile may also go through the generic store
    kend_ = kend;
    const int soff = (int)(k0 * (int64_t)sizeof(T));  // scalar
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const i32x4_t t = raw_buffer_load_b128(rsrc, off[p], soff, 0);
      const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) regs[p][e] = tp[e];
    }
  }
  template <int LD> __device__ __forceinline__ void store_fast(T *lds, int tid) const {
    static_assert(LD % VEC == 0, "row-major image rows must stay 16-byte aligned");
    T *dst = lds + (tid / VPR) * LD + (tid % VPR) * VEC;
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      V t;
      T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) tp[e] = regs[p][e];
      *reinterpret_cast<V *>(dst + p * (NTHREADS / VPR) * LD) = t;
    }
  }
  // issues the loads of columns [k0, k0+BK); columns >= kend are zeroed later (k0 < kend)
  __device__ __forceinline__ void load(int64_t k0, int64_t kend, int tid) {
    k0_ = k0;
    kend_ = kend;
    if constexpr (VECM) {
      const int64_t klast = (kend - 1) & ~(int64_t)(VEC - 1);  // last vector that starts < kend
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int kc = ((tid + p * NTHREADS) % VPR) * VEC;
        int64_t k = k0;
        if (k0 + BK > kend) k = (k0 + kc <= klast) ? k0 : klast - kc;  // uniform test, rare
        const V t = *reinterpret_cast<const V *>(rp[p] + k);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) regs[p][e] = tp[e];
      }
    } else {
#pragma unroll
      for (int p = 0; p < NV; ++p) {
        const int kc = ((tid + p * NTHREADS) % VPR) * VEC;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          int64_t k = k0 + e;
          if (k0 + BK > kend) k = (k0 + kc + e < kend) ? k0 + e : kend - 1 - kc;
          regs[p][e] = rp[p][k];
        }
      }
    }
  }
  // lds: [ROWS][LD] row-major image (LD = BK + RmPad), masked as described above
  template <int LD> __device__ __forceinline__ void store(T *lds, int tid) const {
    static_assert(LD % VEC == 0, "row-major image rows must stay 16-byte aligned");
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int r = v / VPR, kc = (v % VPR) * VEC;
      if (EXACT || r < ROWS) {
        V t;
        T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
          const int64_t k = k0_ + kc + e;
          T val = regs[p][e];
          if (!ok[p] || k >= kend_ || (TRI && k > rowg[p])) val = T(0);
          tp[e] = val;
        }
        *reinterpret_cast<V *>(lds + r * LD + kc) = t;
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
  i32x4_t rsrc;   // fast path: buffer resource at base + r0
  int voff[NV];   // fast path: per-lane byte offset of (kk, rc) inside a k-tile
  int ldb;        // ld in bytes (scalar)
  bool fast_ok;   // wave-uniform

  // Fast path (no VALU in the k loop, see LoaderKC) for whole k-tiles.  Rows >= rend are NOT
  // masked: in a product they only feed output rows/columns the caller discards, so whatever the
  // 16-byte loads find there (the next k-row's data, or 0 past the end of the array — the buffer
  // resource is sized to the array, out-of-range loads return 0 and never fault) is harmless.
  // The ones row is patched in store_fast().  Source rows need only 4-byte alignment (buffer
  // loads in unaligned mode).  nk: number of k-rows of the array.
  int64_t r0_, ones_row_;
  bool ones_here;
  __device__ __forceinline__ void init_fast(const T *base, int64_t ld, int64_t r0, int64_t rend,
                                            int64_t nk, int tid, int64_t ones_row = -1) {
    r0_ = r0;
    ones_row_ = ones_row;
    ones_here = ONES && ones_row >= r0 && ones_row < r0 + ROWS;
    const int64_t extent = ((nk - 1) * ld + (rend - r0)) * (int64_t)sizeof(T);  // bytes from base + r0
    fast_ok = EXACT && rend > r0 && nk > 0 && extent < ((int64_t)1 << 31) &&
              (nk + BK) * ld * (int64_t)sizeof(T) < ((int64_t)1 << 31);
    rsrc = make_buffer_rsrc(base + r0);
    rsrc[2] = __builtin_amdgcn_readfirstlane((int)extent);
    ldb = __builtin_amdgcn_readfirstlane((int)(ld * (int64_t)sizeof(T)));
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      voff[p] = (v / VPK) * (int)(ld * (int64_t)sizeof(T)) + (v % VPK) * VEC * (int)sizeof(T);
    }
  }
  __device__ __forceinline__ int64_t fast_end(int64_t kend) const { return fast_ok ? kend : INT64_MIN; }
  __device__ __forceinline__ void load_fast(int64_t k0) {
    const int soff = (int)k0 * ldb;  // scalar
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const i32x4_t t = raw_buffer_load_b128(rsrc, voff[p], soff, 0);
      const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e) regs[p][e] = tp[e];
    }
  }
  template <int LD> __device__ __forceinline__ void store_fast(T *lds, int tid) const {
#pragma unroll
    for (int p = 0; p < NV; ++p) {
      const int v = tid + p * NTHREADS;
      const int kk = v / VPK, rc = (v % VPK) * VEC;
      V t;
      T *tp = reinterpret_cast<T *>(&t);
#pragma unroll
      for (int e = 0; e < VEC; ++e)
        tp[e] = (ONES && ones_here && r0_ + rc + e == ones_row_) ? T(1) : regs[p][e];
      *reinterpret_cast<V *>(lds + kk * LD + rc) = t;
    }
  }

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
// acc[fm][fn] += A[am0 + fm*TM ..][k] * B[k][bn0 + fn*TM ..], k over the BK columns of the LDS images.
// ARM / BRM: the operand's image is row-major (LoaderKC) rather than k-major (LoaderMC); LDA / LDB
// are the image strides.  k-steps come in chunks of 8 k values: lane group g = lane / TM holds
// k = 8c + KV*g + j at step j of chunk c (KV = 4 f32, 2 f64), the same for both operands.
// All fragment reads of the tile are issued before its first MFMA (the counters release the MFMAs
// chunk by chunk), so LDS latency is paid once per tile.
template <typename T, int FM, int FN, int BK, bool ARM, int LDA, bool BRM, int LDB>
__device__ __forceinline__ void mma_ktile(const T *As, const T *Bs, int am0, int bn0, int lane,
                                          typename Mfma<T>::acc_t (&acc)[FM][FN]) {
  using M = Mfma<T>;
  using V = typename VecT<T>::type;
  constexpr int KV = VecT<T>::N;
  constexpr int NG = 64 / M::TM;       // lane groups = k values per MFMA step (M::KI)
  constexpr int CK = NG * KV;          // k values per chunk (8)
  constexpr int CH = BK / CK;
  static_assert(NG == M::KI && BK % CK == 0, "chunked k mapping");
  const int lr = lane % M::TM, g = lane / M::TM;
  const T *ap = ARM ? As + (am0 + lr) * LDA + g * KV : As + (g * KV) * LDA + am0 + lr;
  const T *bp = BRM ? Bs + (bn0 + lr) * LDB + g * KV : Bs + (g * KV) * LDB + bn0 + lr;
  T a[CH][FM][KV], b[CH][FN][KV];
#pragma unroll
  for (int c = 0; c < CH; ++c) {
#pragma unroll
    for (int fm = 0; fm < FM; ++fm) {
      if constexpr (ARM) {
        const V t = *reinterpret_cast<const V *>(ap + fm * M::TM * LDA + c * CK);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int j = 0; j < KV; ++j) a[c][fm][j] = tp[j];
      } else {
#pragma unroll
        for (int j = 0; j < KV; ++j) a[c][fm][j] = ap[(c * CK + j) * LDA + fm * M::TM];
      }
    }
#pragma unroll
    for (int fn = 0; fn < FN; ++fn) {
      if constexpr (BRM) {
        const V t = *reinterpret_cast<const V *>(bp + fn * M::TM * LDB + c * CK);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int j = 0; j < KV; ++j) b[c][fn][j] = tp[j];
      } else {
#pragma unroll
        for (int j = 0; j < KV; ++j) b[c][fn][j] = bp[(c * CK + j) * LDB + fn * M::TM];
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);  // reads ahead of the MFMAs
#pragma unroll
  for (int c = 0; c < CH; ++c)
#pragma unroll
    for (int j = 0; j < KV; ++j)
#pragma unroll
      for (int fm = 0; fm < FM; ++fm)
#pragma unroll
        for (int fn = 0; fn < FN; ++fn) acc[fm][fn] = M::mma(a[c][fm][j], b[c][fn][j], acc[fm][fn]);
  __builtin_amdgcn_sched_barrier(0);
}

// ---- the k loop of a product whose operands both come through LoaderKC ------------------------------
// Tiles k = kbeg, kbeg + BK, ... < kend (kbeg < kend).  `stage` stores the loaded tile(s) into LDS
// (generic or fast path), `mma` runs the MFMAs of the staged tile.  Leading tiles that every loader
// can take on its fast path run in a loop of their own that holds no VALU instruction besides the
// MFMAs; the remaining (masked / ragged) tiles run through the generic path.
template <int BK, class LoadG, class StoreG, class LoadF, class StoreF, class MmaF>
__device__ __forceinline__ void kloop_kc(int64_t kbeg, int64_t kend, int64_t kfast, const LoadG &load_g,
                                         const StoreG &store_g, const LoadF &load_f,
                                         const StoreF &store_f, const MmaF &mma) {
  kbeg = uniform64(kbeg);  // loop control on the scalar ALU
  kend = uniform64(kend);
  kfast = uniform64(kfast);
  int64_t k = kbeg;
  if (k + BK <= kfast) {
    load_f(k);
    store_f();
  } else {
    load_g(k);
    store_g();
  }
  __syncthreads();
  // tile k staged, tile k + BK fast.  A 32-bit trip count: gfx9 has no scalar 64-bit ordered
  // compare, a 64-bit loop test would put a VALU compare into the loop.
  const int nfast = (kfast - kbeg >= 2 * BK) ? (int)((kfast - kbeg) / BK) - 1 : 0;
  for (int i = 0; i < nfast; ++i, k += BK) {
    load_f(k + BK);
    mma();
    __syncthreads();
    store_f();
    __syncthreads();
  }
  for (; k < kend; k += BK) {
    const bool more = k + BK < kend;
    if (more) load_g(k + BK);
    mma();
    __syncthreads();
    if (more) {
      store_g();
      __syncthreads();
    }
  }
}

// ---- software-pipelined k loop: both operands through LoaderKC, row-major images, LDS double buffer --
// A wave that alternates "MFMA phase / load-store phase" leaves the MFMA pipe idle during its own
// load-store phase unless the partner wave happens to be in its MFMA phase — and the SIMD arbiter
// gives the older wave strict priority, so the two waves do not settle into anti-phase (measured:
// the older block runs as if alone at ~79 % pipe use, the younger one crawls in its gaps, and the
// tail of the launch runs single waves).  Here every wave keeps its OWN MFMA stream dense instead:
// per k-tile t (LDS buffer cur = t & 1, fragment register halves H0 / H1):
//     a   wait for the global loads of tile t+1 (issued one whole tile ago), ds_write them into
//         buffer cur^1, issue the global loads of tile t+2
//     a2  ds_read the second-half fragments of tile t (H1) from buffer cur
//     b   MFMAs of the first half (H0, read during the previous tile)
//     c   barrier: every wave has written tile t+1 and finished reading... (see hazards below)
//     d   ds_read the first-half fragments of tile t+1 (H0) from buffer cur^1
//     e   MFMAs of the second half (H1)
// All memory instructions are issued between MFMAs of the same wave and their latencies are a
// half tile or more ahead of their consumers; the only exposed time is barrier skew.
// Hazards: buffer cur^1 written in (a) of tile t was last read in (d) of tile t-2 / (a2) of tile
// t-1, both before the barrier of tile t-1, which the writer has passed.  Reads in (d) of tile t
// follow the barrier of tile t, after every wave's (a) writes (lgkmcnt(0) before the barrier).
// Leading tiles whose t+1 and t+2 are plain (LoaderKC::fast_end) run in an unrolled-by-two loop
// with no VALU instruction besides the MFMAs; the rest takes the generic (masked) loader path.
template <typename T, int FM, int FN, int BK, int RA, int RB, class LA, class LB>
struct PipeKC {
  using M = Mfma<T>;
  using V = typename VecT<T>::type;
  static constexpr int LD = BK + RmPad<T>::V;
  static constexpr int KV = VecT<T>::N;
  static constexpr int CK = (64 / M::TM) * KV;  // k values per chunk (8)
  static constexpr int CH = BK / CK;
  static constexpr int HC = CH / 2;              // chunks per fragment half
  static constexpr int SZ = (RA + RB) * LD;      // elements per LDS buffer
  static constexpr int LDS_ELEMS = 2 * SZ;
  static_assert(CH % 2 == 0, "two fragment halves");

  T ha[2][HC][FM][KV], hb[2][HC][FN][KV];

  template <int BUF, int HALF>
  __device__ __forceinline__ void read_half(const T *lds, int aoff, int boff) {
    const T *Ab = lds + BUF * SZ + aoff, *Bb = lds + BUF * SZ + RA * LD + boff;
#pragma unroll
    for (int c = 0; c < HC; ++c) {
#pragma unroll
      for (int fm = 0; fm < FM; ++fm) {
        const V t = *reinterpret_cast<const V *>(Ab + fm * M::TM * LD + (HALF * HC + c) * CK);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int j = 0; j < KV; ++j) ha[HALF][c][fm][j] = tp[j];
      }
#pragma unroll
      for (int fn = 0; fn < FN; ++fn) {
        const V t = *reinterpret_cast<const V *>(Bb + fn * M::TM * LD + (HALF * HC + c) * CK);
        const T *tp = reinterpret_cast<const T *>(&t);
#pragma unroll
        for (int j = 0; j < KV; ++j) hb[HALF][c][fn][j] = tp[j];
      }
    }
  }
  template <int HALF>
  __device__ __forceinline__ void mma_half(typename M::acc_t (&acc)[FM][FN]) {
#pragma unroll
    for (int c = 0; c < HC; ++c)
#pragma unroll
      for (int j = 0; j < KV; ++j)
#pragma unroll
        for (int fm = 0; fm < FM; ++fm)
#pragma unroll
          for (int fn = 0; fn < FN; ++fn)
            acc[fm][fn] = M::mma(ha[HALF][c][fm][j], hb[HALF][c][fn][j], acc[fm][fn]);
  }

  // one k-tile; CUR compile-time so every LDS offset is an immediate.  FAST: tiles t+1 and t+2
  // exist and are plain (no masking).  Otherwise they may not exist; `plain` (uniform) says that
  // what exists needs no masking and keeps the tile free of vector-ALU work all the same.
  template <int CUR, bool FAST>
  __device__ __forceinline__ void tile(LA &la, LB &lb, T *lds, int aoff, int boff, int64_t k,
                                       int64_t kend, int tid, bool compute,
                                       typename M::acc_t (&acc)[FM][FN], bool plain = false) {
    constexpr int NXT = CUR ^ 1;
    const bool has1 = FAST || k + BK < kend, has2 = FAST || k + 2 * BK < kend;
    if (has1) {
      if (FAST || plain) {
        la.template store_fast<LD>(lds + NXT * SZ, tid);
        lb.template store_fast<LD>(lds + NXT * SZ + RA * LD, tid);
      } else {
        la.template store<LD>(lds + NXT * SZ, tid);
        lb.template store<LD>(lds + NXT * SZ + RA * LD, tid);
      }
    }
    if (has2) {
      if (FAST || plain) {
        la.load_fast(k + 2 * BK, kend);
        lb.load_fast(k + 2 * BK, kend);
      } else {
        la.load(k + 2 * BK, kend, tid);
        lb.load(k + 2 * BK, kend, tid);
      }
    }
    if (compute) {
      read_half<CUR, 1>(lds, aoff, boff);
      __builtin_amdgcn_sched_barrier(0);
      mma_half<0>(acc);
      __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    if (compute) {
      if (has1) read_half<NXT, 0>(lds, aoff, boff);
      __builtin_amdgcn_sched_barrier(0);
      mma_half<1>(acc);
      __builtin_amdgcn_sched_barrier(0);
    }
  }

  // acc += A[am0.., k] B[bn0.., k]^T over k in [kbeg, kend), kbeg < kend; lds holds LDS_ELEMS.
  // compute (wave-uniform) = false: the wave only stages operands and keeps the barriers (its
  // output tile is not wanted) — its share of the MFMA pipe goes to the other block on the CU.
  __device__ __forceinline__ void run(LA &la, LB &lb, T *lds, int64_t kbeg, int64_t kend, int am0,
                                      int bn0, int lane, int tid,
                                      typename M::acc_t (&acc)[FM][FN], bool compute = true) {
    kbeg = uniform64(kbeg);
    kend = uniform64(kend);
    int64_t kfast = la.fast_end(kend);
    if (lb.fast_end(kend) < kfast) kfast = lb.fast_end(kend);
    kfast = uniform64(kfast);
    const int lr = lane % M::TM, g = lane / M::TM;
    const int aoff = (am0 + lr) * LD + g * KV, boff = (bn0 + lr) * LD + g * KV;
    // leading plain tiles; tile t runs in the fast loop when tiles t+1 and t+2 are plain
    const int nplain = (kfast > kbeg) ? (int)((kfast - kbeg) / BK) : 0;
    const int npairs = (nplain >= 4) ? (nplain - 2) / 2 : 0;

    // tiles 0 and 1: through the fast path too when they are plain (keeps the prologue, which may
    // run next to another block's MFMA stream, almost free of vector-ALU work)
    if (nplain >= 2) {
      // both leading tiles requested at once (tile 1 into a second register set that the
      // allocator renames into the loaders' own): one memory round trip before the first MFMA
      // instead of two — with K = 256 (the factorisation's trailing update) that is 3 % of a tile
      la.load_fast(kbeg, kend);
      lb.load_fast(kbeg, kend);
      LA la1 = la;
      LB lb1 = lb;
      la1.load_fast(kbeg + BK, kend);
      lb1.load_fast(kbeg + BK, kend);
      la.template store_fast<LD>(lds, tid);
      lb.template store_fast<LD>(lds + RA * LD, tid);
      la = la1;
      lb = lb1;
    } else {
      if (nplain >= 1) {
        la.load_fast(kbeg, kend);
        lb.load_fast(kbeg, kend);
        la.template store_fast<LD>(lds, tid);
        lb.template store_fast<LD>(lds + RA * LD, tid);
      } else {
        la.load(kbeg, kend, tid);
        lb.load(kbeg, kend, tid);
        la.template store<LD>(lds, tid);
        lb.template store<LD>(lds + RA * LD, tid);
      }
      if (kbeg + BK < kend) {
        la.load(kbeg + BK, kend, tid);
        lb.load(kbeg + BK, kend, tid);
      }
    }
    __syncthreads();
    if (compute) read_half<0, 0>(lds, aoff, boff);
    int64_t k = kbeg;
    for (int i = 0; i < npairs; ++i, k += 2 * BK) {
      tile<0, true>(la, lb, lds, aoff, boff, k, kend, tid, compute, acc);
      tile<1, true>(la, lb, lds, aoff, boff, k + BK, kend, tid, compute, acc);
    }
    const int nrest = (int)((kend - k + BK - 1) / BK);
    // the whole k range plain: the last tiles keep off the masked (vector-ALU) store path too
    const bool plain = nplain >= 2 && kbeg + (int64_t)nplain * BK >= kend;
    for (int i = 0; i < nrest; i += 2, k += 2 * BK) {
      tile<0, false>(la, lb, lds, aoff, boff, k, kend, tid, compute, acc, plain);
      if (i + 1 < nrest) tile<1, false>(la, lb, lds, aoff, boff, k + BK, kend, tid, compute, acc, plain);
    }
  }
};

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
