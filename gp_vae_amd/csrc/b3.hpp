// b3.hpp — "bf16 x 3": float32-accurate tile products on the bf16 matrix pipe (gfx950).
//
// Every f32 operand is split EXACTLY into three bf16 pieces, x = x0 + x1 + x2 (8 + 8 + 8 mantissa
// bits; the split is exact while no piece underflows), and a product sum_k a_k b_k is accumulated in
// f32 from the six piece products of weight >= 2^-16:
//     a0 b0 + (a0 b1 + a1 b0) + (a0 b2 + a1 b1 + a2 b0)          on v_mfma_f32_32x32x16_bf16.
// The three dropped products are below 2^-24 |a||b| each; measured max |err| / sum|a b| = 7e-8 at
// K = 4096 (tools/ub_bf16x3.hip) against 1.4e-7 for v_mfma_f32_32x32x2_f32 — float32 results at
// 2.5x the f32 MFMA rate.  Operands must be PRE-SPLIT in memory as three planes of bf16 (splitting
// in the k loop costs ~100 vector-ALU instructions per k-tile and thread, and vector work adds to
// MFMA time on this chip: gemm_core.hpp).
//
// PipeB3T<WFM, NP, NW>: C ((32 WFM NW / 2) x 128 per workgroup of NW waves, NW / 2 along M x 2 along N, each wave
// (32 WFM) x 64) += A B^T with both operands K-contiguous (rows of length >= K), one workgroup per CU.  The 256 x 128
// tile of the draw and the trailing update runs on EIGHT waves of 64 x 64 (WFM = 2, NW = 8: 145.5 KB of LDS, ~210
// registers per lane, two waves per SIMD) since late round 3: with four waves of 128 x 64 (WFM = 4: 512 registers, one
// wave per SIMD) the matrix pipe was busy 0.65 of a wave's time and nothing ran while the wave waited at the k-tile's
// barrier or for LDS (profiles/r03_draw_b3_issue_pmc.txt); two waves per SIMD cover each other's waits at a third more
// fragment reads per MFMA: draw 0.354 -> 0.331 ms, trailing updates -6 %, same bits.
// Software pipeline inside each wave (measured levers in tools/proto_draw_b3p.hip):
//   * LDS double buffer of k-tiles of 32; image [plane][8-k chunk][row][16 B] (+32 B per chunk
//     plane): fragment reads (4 x 16 lanes, 64 banks) and tile writes (8 x 8 lanes over a 128-byte
//     window) are conflict-free without row padding;
//   * per k-tile: first half = MFMAs of k16 step 0 with the reads of step 1's fragments, then the
//     ds_writes of tile t+1, each followed by the global load of the same register for tile t+2 (a
//     whole tile of latency cover), barrier, second half = MFMAs of step 1 with the reads of tile
//     t+1's step-0 fragments; __builtin_amdgcn_sched_group_barrier fixes the interleave (the
//     compiler's own schedule was 19 % slower).
// The k range must be a multiple of 64 and at least 128 (an even number >= 4 of k-tiles).
#pragma once
#include "gemm_core.hpp"

namespace gpvae {

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

constexpr int B3_TM = 256, B3_TN = 128, B3_BK = 32;

// exact three-way split of a float into bf16 pieces (round-to-nearest each time; the remainders are
// exactly representable, so hi + mid + lo == v bit for bit unless a piece underflows)
__device__ __forceinline__ void split3_bf16(float v, __bf16 &h, __bf16 &m, __bf16 &l) {
  // v must be the ROUNDED float: without this the compiler contracts the producer's last multiply
  // into the subtraction below (fma) and the pieces sum to the unrounded product instead
  asm volatile("" : "+v"(v));
  h = (__bf16)v;
  const float r1 = v - (float)h;
  m = (__bf16)r1;
  l = (__bf16)(r1 - (float)m);
}

// fp16 x 2: x * 2^e = h + l + (error <= 2^-22 |x| 2^e): h = fp16(x 2^e), l = fp16(x 2^e - h).  e is chosen per
// matrix from a bound on |x| so that bound * 2^e lies in (2^13, 2^14] (fp16 overflows at 2^16); entries below
// 2^-17 of the bound keep fewer bits in l (fp16 subnormals), an absolute error of 2^-39 of the bound.
__device__ __forceinline__ void split2_f16(float v, _Float16 &h, _Float16 &l) {
  asm volatile("" : "+v"(v));  // as in split3_bf16: v must be the rounded float
  h = (_Float16)v;
  l = (_Float16)(v - (float)h);
}
__device__ __forceinline__ float h2_scale(float bound) {  // 2^e with bound * 2^e in (2^13, 2^14]
  int ex;
  (void)frexpf(bound, &ex);  // bound = m 2^ex, m in [0.5, 1)
  int e = 14 - ex;           // |e| <= 62: the trailing update squares the scale, and 2^(2e) must stay a float
  e = e > 62 ? 62 : (e < -62 ? -62 : e);
  return ldexpf(1.0f, e);
}

// WFM_ = 2 on four waves: a 128 x 128 tile (99.8 KB of LDS) for problems with fewer 256-row tiles than CUs — half
// the MFMAs per k-tile against two thirds of the operand bytes, but twice the workgroups.
// NP_ = 3: three bf16 planes per operand, six piece products (above).  NP_ = 2: "fp16 x 2" — two fp16 planes
// (11 + 11 significand bits of the operand, which the caller has scaled into fp16's range) and the three
// products a0 b1, a1 b0, a0 b0 on v_mfma_f32_32x32x16_f16: half the matrix work; the operands are rounded to
// 22 bits, and the measured error of a K = 4096 product is the same 8e-8 of sum|a b| as bf16 x 3's, because the
// f32 accumulation dominates both (tools/ub_bf16x3.hip).  Planes are 2-byte elements either way.
template <int WFM_, int NP_ = 3, int NW_ = 4> struct PipeB3T {
  static constexpr int WFM = WFM_, WFN = 2;  // 32 x 32 MFMA tiles per wave: (32 WFM) x 64
  static constexpr int NP = NP_;
  static constexpr int NW = NW_;             // waves per workgroup: 2 along N (64 columns each) x NW / 2 along M
  static constexpr int NTH = 64 * NW;
  static constexpr int NPROD = NP == 3 ? 6 : 3;
  static constexpr int TM = 32 * WFM * (NW / 2);  // rows of the A operand per workgroup
  static constexpr int RPP = NTH / 4;             // rows one pass of tile loads covers (4 threads per row)
  static constexpr int PA = TM / RPP, PB = B3_TN / RPP;  // passes per plane
  static constexpr int PAD = 32;
  static constexpr int CPA = TM * 16 + PAD, CPB = B3_TN * 16 + PAD;  // bytes per (plane, chunk)
  static constexpr int A_BYTES = NP * 4 * CPA, STAGE = NP * 4 * (CPA + CPB);
  static constexpr int LDS_BYTES = 2 * STAGE;
  static constexpr int NQA = NP * PA, NQB = NP * PB;  // 16-byte vectors per thread and k-tile
  static constexpr int NRD = NP * (WFM + WFN);        // fragment reads per k16 step
  static constexpr int NMF = NPROD * WFM * WFN;       // MFMAs per k16 step
  static constexpr int LEAD = (WFM == 4 && NP == 3) ? 6 : 0;
  static_assert(NW == 4 || NW == 8, "four or eight waves");
  static_assert(TM % RPP == 0 && B3_TN % RPP == 0, "whole load passes");

  i32x4_t pre[NQA + NQB];
  i32x4_t fa[2][NP][WFM], fb[2][NP][WFN];  // fragments: 8 two-byte elements each
  f32x16 acc[WFM][WFN];
  i32x4_t rsA[NP], rsB[NP];  // per plane: resource at the tile's first row
  int voffA, voffB;        // per-lane byte offset of the loads (row tid/4, chunk tid%4)
  int lda2, ldb2;          // row strides in bytes (scalar)
  int wr_a, wr_b;          // per-lane LDS byte offsets of the tile writes
  int rd_a, rd_b;          // per-lane LDS byte offsets of the fragment reads

  // A3 / B3: plane 0 of the operand at the tile's first row and k = 0; planes psA / psB ELEMENTS apart
  template <typename E>
  __device__ __forceinline__ void init(const E *A3, int64_t lda, int64_t psA, const E *B3,
                                       int64_t ldb, int64_t psB, int tid, int wm, int wn) {
    static_assert(sizeof(E) == 2, "planes of two-byte elements");
    const int lane = tid & 63, r = lane & 31, h = lane >> 5;
#pragma unroll
    for (int p = 0; p < NP; ++p) {
      rsA[p] = make_buffer_rsrc(A3 + p * psA);
      rsB[p] = make_buffer_rsrc(B3 + p * psB);
    }
    lda2 = __builtin_amdgcn_readfirstlane((int)(lda * 2));
    ldb2 = __builtin_amdgcn_readfirstlane((int)(ldb * 2));
    voffA = (tid >> 2) * (int)(lda * 2) + (tid & 3) * 16;
    voffB = (tid >> 2) * (int)(ldb * 2) + (tid & 3) * 16;
    wr_a = (tid & 3) * CPA + (tid >> 2) * 16;
    wr_b = A_BYTES + (tid & 3) * CPB + (tid >> 2) * 16;
    rd_a = h * CPA + (wm + r) * 16;
    rd_b = A_BYTES + h * CPB + (wn + r) * 16;
  }
  // operand tiles need rows within 2 GiB of the tile's first row
  __host__ __device__ static bool usable(int64_t lda, int64_t ldb) {
    return (int64_t)TM * lda * 2 < ((int64_t)1 << 31) && (int64_t)B3_TN * ldb * 2 < ((int64_t)1 << 31);
  }

  __device__ __forceinline__ void gload(int k0) {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
      pre[q] = raw_buffer_load_b128(rsA[q / PA], voffA, (q % PA) * RPP * lda2 + k0 * 2, 0);
#pragma unroll
    for (int q = 0; q < NQB; ++q)
      pre[NQA + q] = raw_buffer_load_b128(rsB[q / PB], voffB, (q % PB) * RPP * ldb2 + k0 * 2, 0);
  }
  template <int BUF> __device__ __forceinline__ void lstore(unsigned char *lds) {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
      *reinterpret_cast<i32x4_t *>(lds + BUF * STAGE + wr_a + (q / PA) * 4 * CPA + (q % PA) * RPP * 16) = pre[q];
#pragma unroll
    for (int q = 0; q < NQB; ++q)
      *reinterpret_cast<i32x4_t *>(lds + BUF * STAGE + wr_b + (q / PB) * 4 * CPB + (q % PB) * RPP * 16) = pre[NQA + q];
  }
  template <int BUF, int S> __device__ __forceinline__ void read_sub(const unsigned char *lds) {
#pragma unroll
    for (int p = 0; p < NP; ++p) {
#pragma unroll
      for (int f = 0; f < WFM; ++f)
        fa[S][p][f] = *reinterpret_cast<const i32x4_t *>(lds + BUF * STAGE + rd_a + (p * 4 + 2 * S) * CPA + f * 512);
#pragma unroll
      for (int f = 0; f < WFN; ++f)
        fb[S][p][f] = *reinterpret_cast<const i32x4_t *>(lds + BUF * STAGE + rd_b + (p * 4 + 2 * S) * CPB + f * 512);
    }
  }
  template <int S> __device__ __forceinline__ void mma_sub() {
    // piece products smallest first; consecutive MFMAs go to DIFFERENT accumulators (a dependent
    // back-to-back pair on one accumulator would wait for the first one's last pass)
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};  // NP == 3
    constexpr int QA[3] = {0, 1, 0}, QB[3] = {1, 0, 0};                    // NP == 2
#pragma unroll
    for (int q = 0; q < NPROD; ++q)
#pragma unroll
      for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
        for (int fn = 0; fn < WFN; ++fn) {
          if constexpr (NP == 3)
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, fa[S][PA[q]][fm]),
                                                                  __builtin_bit_cast(bf16x8_t, fb[S][PB[q]][fn]),
                                                                  acc[fm][fn], 0, 0, 0);
          else
            acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8_t, fa[S][QA[q]][fm]),
                                                                 __builtin_bit_cast(f16x8_t, fb[S][QB[q]][fn]),
                                                                 acc[fm][fn], 0, 0, 0);
        }
  }
#define GPVAE_SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
  // k-tile in buffer CUR.  HAS1 / HAS2: tiles t+1 / t+2 exist.
  template <int CUR, bool HAS1, bool HAS2>
  __device__ __forceinline__ void tile(unsigned char *lds, int k0) {
    constexpr int NXT = CUR ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    read_sub<CUR, 1>(lds);
    if (HAS1) lstore<NXT>(lds);
    if (HAS2) gload(k0 + 2 * B3_BK);
    mma_sub<0>();
#pragma unroll
    for (int i = 0; i < NRD; ++i) { GPVAE_SGB(0x008, 1); GPVAE_SGB(0x100, 1); }  // MFMA, DS read
    if (LEAD > 0) GPVAE_SGB(0x008, LEAD);
    if (HAS1) {
#pragma unroll
      for (int i = 0; i < NQA + NQB; ++i) {
        GPVAE_SGB(0x008, 1);
        GPVAE_SGB(0x200, 1);            // DS write
        if (HAS2) GPVAE_SGB(0x020, 1);  // VMEM read
      }
    }
    GPVAE_SGB(0x008, NMF);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    if (HAS1) read_sub<NXT, 0>(lds);
    mma_sub<1>();
    if (HAS1) {
#pragma unroll
      for (int i = 0; i < NRD; ++i) { GPVAE_SGB(0x008, 1); GPVAE_SGB(0x100, 1); }
    }
    GPVAE_SGB(0x008, NMF);
    __builtin_amdgcn_sched_barrier(0);
  }
#undef GPVAE_SGB

  // acc += A[:, kbeg:kend) B[:, kbeg:kend)^T; (kend - kbeg) a multiple of 64 and >= 128
  __device__ __forceinline__ void run(unsigned char *lds, int kbeg, int kend) {
    const int nt = (kend - kbeg) / B3_BK;  // even, >= 4
    gload(kbeg);
    lstore<0>(lds);
    gload(kbeg + B3_BK);
    __syncthreads();
    read_sub<0, 0>(lds);
    int k0 = kbeg;
    for (int i = 0; i < (nt - 2) / 2; ++i, k0 += 2 * B3_BK) {
      tile<0, true, true>(lds, k0);
      tile<1, true, true>(lds, k0 + B3_BK);
    }
    tile<0, true, false>(lds, k0);
    tile<1, false, false>(lds, k0 + B3_BK);
  }
};
using PipeB3 = PipeB3T<4>;

}  // namespace gpvae
