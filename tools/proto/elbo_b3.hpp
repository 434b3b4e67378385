// PROTOTYPE (round 3), NOT part of the library: measured and not adopted — see the result at the end of this comment.
// elbo_b3.hpp — the ELBO's n-sized float32 products on the bf16 matrix pipe with operands split ON THE FLY.
//
// The products of vae.hip that stream Y (B x n) — A1 = Y W1 (vae.py:75-76, the encoder's first Dense) and
// [dW1; db1] = [Y | 1]^T dA1 — ran on the f32 MFMA (v_mfma_f32_32x32x2_f32, 1/16 of the bf16 rate) at 0.3-0.4 of
// that pipe.  b3.hpp's exact three-way bf16 split (x = x0 + x1 + x2 bit for bit, six piece products, f32
// accumulation: float32-accurate) needs pre-split operands where the split would sit in a k loop that is MFMA-bound
// (the draw, the Cholesky trailing update).  These products are HBM-bound instead: per 16 k values a wave issues 12
// MFMAs (384 cycles) and has ~44 vector instructions of splitting to do for the 8 floats a lane loaded — they fit
// in the MFMAs' shadow (a bf16 MFMA holds the vector issue for 8 of its 32 cycles) — so Y is read as float32
// straight from HBM into registers, split there and fed to the matrix pipe with no LDS pass at all; only the small
// operand (a slice of W1 / dA1) is staged in LDS, split once per workgroup.
//
// k mapping: v_mfma_f32_32x32x16_bf16 takes A[row l%32][k = 8 (l/32) + j] and B[k = 8 (l/32) + j][col l%32],
// j = 0..7.  Any permutation of k serves as long as both operands use it; here a block of 32 consecutive k is two
// steps s = 0, 1 with lane half h = l/32 holding k = 16 h + 8 s + j, so that a lane's A data of a block is 64
// contiguous bytes of its row of Y (four 16-byte loads).
//
// Result (MI355X, C3 shape, kernel times under rocprofv3, profiles/r03_elbo_b3_experiment.txt): A1 28.8 -> 22.1 us
// (the stream alone, no split and no MFMA: 15.9 us), the fused output layer with its three products on this scheme
// (fragments read from the same f32 LDS images, split in registers; the fragments of [D2 | 1] split once) 79.8 ->
// 73.4 us — both kernels are bound by their one-wave-per-SIMD latency chains (LDS round trips, epilogue, barrier),
// not by the matrix pipe, so moving the products to the faster pipe buys little; the training step did not move
// (0.7926 / 0.7928 ms with, 0.7958 / 0.7926 ms without, alternating runs on one box).  Also in here: a compiler trap —
// __builtin_bit_cast(float, t[e]) on the <4 x i32> result of llvm.amdgcn.raw.buffer.load shrinks the load to one
// dword and splats it; read the elements through a float pointer.
#pragma once
#include "b3.hpp"

namespace gpvae {

// 8 floats -> the lane's fragment in each of the three bf16 planes
__device__ __forceinline__ void split8_b3(const float (&v)[8], i32x4_t &p0, i32x4_t &p1, i32x4_t &p2) {
  __bf16 h[8], m[8], l[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) split3_bf16(v[e], h[e], m[e], l[e]);
  p0 = *reinterpret_cast<const i32x4_t *>(h);
  p1 = *reinterpret_cast<const i32x4_t *>(m);
  p2 = *reinterpret_cast<const i32x4_t *>(l);
}

__device__ __forceinline__ f32x16 mfma_b3(const i32x4_t &a, const i32x4_t &b, const f32x16 &c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8_t, a), __builtin_bit_cast(bf16x8_t, b), c, 0, 0, 0);
}

// acc[t] += a b[t] over the six piece products, smallest first (b3.hpp)
template <int NT>
__device__ __forceinline__ void mma6_b3(const i32x4_t (&a)[3], const i32x4_t (&b)[3][NT], f32x16 (&acc)[NT]) {
  constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
  for (int q = 0; q < 6; ++q)
#pragma unroll
    for (int t = 0; t < NT; ++t) acc[t] = mfma_b3(a[PA[q]], b[PB[q]][t], acc[t]);
}

// ---- fragments of an f32 LDS image, split on the fly ------------------------------------------------------------
// The lane's 8 k values of one k16 step (k = kb + 8 (l/32) + j) of image row / column `idx`:
//   row-major image [idx][k] (stride ld floats, rows 16-byte aligned, kb % 8 == 0): two 16-byte reads;
//   k-major image [k][idx]: eight scalar reads.
template <bool RM>
__device__ __forceinline__ void frag_from_f32(const float *img, int ld, int idx, int kb, int lane, i32x4_t (&f)[3]) {
  const int k = kb + 8 * (lane >> 5);
  float v[8];
  if constexpr (RM) {
    const float4 t0 = *reinterpret_cast<const float4 *>(img + idx * ld + k), t1 = *reinterpret_cast<const float4 *>(img + idx * ld + k + 4);
    v[0] = t0.x; v[1] = t0.y; v[2] = t0.z; v[3] = t0.w;
    v[4] = t1.x; v[5] = t1.y; v[6] = t1.z; v[7] = t1.w;
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = img[(k + j) * ld + idx];
  }
  split8_b3(v, f[0], f[1], f[2]);
}

// ---- A1 slabs: slab s = Y[:, s KS .. (s+1) KS) W1[s KS .. (s+1) KS), :]   (B x h0, h0 <= 64) -------------------
// grid (B / (32 NW rbw), n / KS); a workgroup of NW waves stages its slice of W1 as bf16 planes in LDS — fragment
// image [plane][8-k chunk][col 64][16 B], conflict-free for ds_read_b128 — and streams rbw blocks of 32 NW rows of Y
// through registers: wave w owns rows 32 w .. 32 w + 31 of a block; all loads of a block's slice are in flight at
// once and the next block's are requested as registers free up.  NW = 8 puts two waves on every SIMD: one wave's
// split and LDS waits run under the other's MFMAs.
template <int KS, int NW>
__global__ __launch_bounds__(64 * NW, KS == 128 ? 2 : 1) void a1_b3_kernel(const float *__restrict__ Y, int64_t ldy,
                                                                           const float *__restrict__ W1, int h0,
                                                                           float *slabs, int64_t slab, int rbw) {
  using M = Mfma<float>;
  constexpr int NQ = KS / 8, NB = KS / 32, NT = 64 * NW, RBK = 32 * NW;  // RBK: rows per block (32 per wave)
  static_assert((NQ * 16) % NT == 0, "staging: whole 8 x 4 groups per thread");
  __shared__ __align__(16) unsigned char lds[3 * NQ * 1024];
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r = lane & 31, h = lane >> 5;
  const int64_t k0 = (int64_t)blockIdx.y * KS;
  const int64_t blk0 = (int64_t)blockIdx.x * rbw;

  // the slice of W1 is requested FIRST (loads return in order: behind the Y loads it would hold the staging back
  // until the whole first block has arrived): a thread takes 8 k values x 4 columns, eight 16-byte loads
  constexpr int NG = NQ * 16 / NT;
  i32x4_t wrs = make_buffer_rsrc(W1 + k0 * h0);
  wrs[2] = __builtin_amdgcn_readfirstlane((int)((int64_t)KS * h0 * 4));  // reads past the slice return 0
  i32x4_t wv[NG][8];
#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int u = tid + gi * NT, g = u & 15, q = u >> 4;
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[gi][j] = raw_buffer_load_b128(wrs, ((q * 8 + j) * h0 + 4 * g) * 4, 0, 0);
  }
  i32x4_t ya[NB][4];
  const int voff = r * (int)(ldy * 4) + h * 64;
  auto rsrc_of = [&](int64_t blk) { return make_buffer_rsrc(Y + (blk * RBK + wave * 32) * ldy + k0); };
  i32x4_t rs = rsrc_of(blk0);
#pragma unroll
  for (int kb = 0; kb < NB; ++kb)
#pragma unroll
    for (int i = 0; i < 4; ++i) ya[kb][i] = raw_buffer_load_b128(rs, voff, kb * 128 + i * 16, 0);

#pragma unroll
  for (int gi = 0; gi < NG; ++gi) {
    const int u = tid + gi * NT, g = u & 15, q = u >> 4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const int c = 4 * g + e;
      float v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float w = reinterpret_cast<const float *>(&wv[gi][j])[e];
        v[j] = c < h0 ? w : 0.0f;
      }
      i32x4_t p0, p1, p2;
      split8_b3(v, p0, p1, p2);
      *reinterpret_cast<i32x4_t *>(lds + ((0 * NQ + q) * 64 + c) * 16) = p0;
      *reinterpret_cast<i32x4_t *>(lds + ((1 * NQ + q) * 64 + c) * 16) = p1;
      *reinterpret_cast<i32x4_t *>(lds + ((2 * NQ + q) * 64 + c) * 16) = p2;
    }
  }
  __syncthreads();

  const unsigned char *bp = lds + (2 * h * 64 + r) * 16;
  // One k16 step = 12 MFMAs (384 cycles of the matrix pipe).  Software pipeline inside the wave: while the MFMAs of
  // step i run, the B fragments of step i + 1 are read from LDS and its A fragment is split (about 60 vector
  // instructions, five per MFMA shadow); sched_group_barrier pins that interleave — the compiler's own order put
  // the split in front of the MFMAs and waited for every pair of LDS reads (~900 cycles per step).
  auto frag_b = [&](int st, i32x4_t (&b)[3][2]) {  // st = 2 kb + s
#pragma unroll
    for (int p = 0; p < 3; ++p)
#pragma unroll
      for (int t = 0; t < 2; ++t)
        b[p][t] = *reinterpret_cast<const i32x4_t *>(bp + ((p * NQ + 4 * (st >> 1) + (st & 1)) * 64 + 32 * t) * 16);
  };
  auto frag_a = [&](int st, i32x4_t (&a)[3]) {
    const int kb = st >> 1, s2 = st & 1;
    // (elements through a float pointer: __builtin_bit_cast(float, t[e]) on the <4 x i32> result of the raw
    // buffer load lets the compiler shrink the load to one dword and splat it)
    const float *y0 = reinterpret_cast<const float *>(&ya[kb][2 * s2]), *y1 = reinterpret_cast<const float *>(&ya[kb][2 * s2 + 1]);
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      v[e] = y0[e];
      v[4 + e] = y1[e];
    }
    split8_b3(v, a[0], a[1], a[2]);
  };
  for (int rb = 0; rb < rbw; ++rb) {
    const bool more = rb + 1 < rbw;
    if (more) rs = rsrc_of(blk0 + rb + 1);
    f32x16 acc[2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
    i32x4_t a[2][3], b[2][3][2];
    frag_b(0, b[0]);
    frag_a(0, a[0]);
#pragma unroll
    for (int st = 0; st < 2 * NB; ++st) {
      __builtin_amdgcn_sched_barrier(0);
      if (st + 1 < 2 * NB) {
        frag_b(st + 1, b[(st + 1) & 1]);
        frag_a(st + 1, a[(st + 1) & 1]);
      }
      mma6_b3<2>(a[st & 1], b[st & 1], acc);
      if (st + 1 < 2 * NB) {
#pragma unroll
        for (int i = 0; i < 12; ++i) {
          if (i < 6) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);  // DS read
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);             // MFMA
          __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);             // VALU
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      if (more && (st & 1)) {  // this block's registers are free: request the next row block's
#pragma unroll
        for (int i = 0; i < 4; ++i) ya[st >> 1][i] = raw_buffer_load_b128(rs, voff, (st >> 1) * 128 + i * 16, 0);
      }
    }
    float *out = slabs + (int64_t)blockIdx.y * slab + ((blk0 + rb) * RBK + wave * 32) * h0;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int col = 32 * t + M::col(lane);
        if (col < h0) out[(int64_t)M::row(lane, e) * h0 + col] = acc[t][e];
      }
  }
}

}  // namespace gpvae
