// b3.hip — the bf16 x 3 producers and the draw on them.
//
//   gpvae_split_b3   f32 matrix (optionally its lower triangle only) -> three bf16 planes
//   gpvae_randn_b3   the normals of gpvae_randn written straight as three bf16 planes (the f32 array
//                    never exists: E of the training step, /root/reference/src/model/vae.py:160-162)
//   gpvae_draw_b3    Y = E L^T + sigma E'  (gp.py:190-194) on PipeB3: float32-accurate results at
//                    2.5x the f32 MFMA rate (b3.hpp).  L3 holds zeros above the diagonal, so the k
//                    loop needs no mask; it stops at the column tile's diagonal block.
#include <algorithm>
#include "b3.hpp"
#include "philox.hpp"

namespace gpvae {

// grid (ceil(cols/1024), rows, batch); 4 elements per thread
__global__ __launch_bounds__(256) void split_b3_kernel(int64_t rows, int64_t cols, const float *X,
                                                       int64_t ldx, int64_t strideX, int lower,
                                                       __bf16 *out, int64_t ld3, int64_t plane,
                                                       int64_t stride3) {
  const int64_t b = blockIdx.z, i = blockIdx.y;
  const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  if (j0 >= cols) return;
  const float *src = X + b * strideX + i * ldx;
  __bf16 *dst = out + b * stride3 + i * ld3;
  __bf16 h[4], m[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t j = j0 + e;
    float v = (j < cols) ? src[j < cols ? j : 0] : 0.f;
    if (lower && j > i) v = 0.f;
    split3_bf16(v, h[e], m[e], l[e]);
  }
  if (j0 + 3 < cols && (ld3 & 3) == 0 && (plane & 3) == 0 && (stride3 & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    *reinterpret_cast<uint64_t *>(dst + j0) = *reinterpret_cast<const uint64_t *>(h);
    *reinterpret_cast<uint64_t *>(dst + plane + j0) = *reinterpret_cast<const uint64_t *>(m);
    *reinterpret_cast<uint64_t *>(dst + 2 * plane + j0) = *reinterpret_cast<const uint64_t *>(l);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j0 + e < cols) {
        dst[j0 + e] = h[e];
        dst[plane + j0 + e] = m[e];
        dst[2 * plane + j0 + e] = l[e];
      }
  }
}

// element i of the stream -> out[p * plane + i], p = 0..2 (count and plane multiples of 4, 8-byte
// aligned: checked by the entry point)
__global__ __launch_bounds__(256) void randn_b3_kernel(int64_t count, uint64_t seed, uint64_t stream_id,
                                                       __bf16 *out, int64_t plane) {
  const int64_t nq = count / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
       q += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    philox_normals4<float>(q, seed, stream_id, z);
    __bf16 h[4], m[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split3_bf16(z[e], h[e], m[e], l[e]);
    *reinterpret_cast<uint64_t *>(out + q * 4) = *reinterpret_cast<const uint64_t *>(h);
    *reinterpret_cast<uint64_t *>(out + plane + q * 4) = *reinterpret_cast<const uint64_t *>(m);
    *reinterpret_cast<uint64_t *>(out + 2 * plane + q * 4) = *reinterpret_cast<const uint64_t *>(l);
  }
}

// ---- fp16 x 2: two fp16 planes of a SCALED operand (b3.hpp, NP = 2) -----------------------------------------
// split2_f16 / h2_scale: b3.hpp
constexpr float H2_E_SCALE = 4096.0f;  // standard normals: |e| < 7.2 from 32-bit uniforms, 7.2 * 2^12 < 2^15

// grid (ceil(cols/1024), rows, batch); 4 elements per thread
__global__ __launch_bounds__(256) void split_h2_kernel(int64_t rows, int64_t cols, const float *X, int64_t ldx,
                                                       int64_t strideX, int lower, const float *bound,
                                                       _Float16 *out, int64_t ld2, int64_t plane, int64_t stride2,
                                                       float *inv_scale) {
  const int64_t b = blockIdx.z, i = blockIdx.y;
  const int64_t j0 = ((int64_t)blockIdx.x * 256 + threadIdx.x) * 4;
  const float sc = h2_scale(bound[b]);
  if (i == 0 && blockIdx.x == 0 && threadIdx.x == 0) inv_scale[b] = 1.0f / sc;
  if (j0 >= cols) return;
  const float *src = X + b * strideX + i * ldx;
  _Float16 *dst = out + b * stride2 + i * ld2;
  _Float16 h[4], l[4];
#pragma unroll
  for (int e = 0; e < 4; ++e) {
    const int64_t j = j0 + e;
    float v = (j < cols) ? src[j < cols ? j : 0] : 0.f;
    if (lower && j > i) v = 0.f;
    split2_f16(v * sc, h[e], l[e]);
  }
  if (j0 + 3 < cols && (ld2 & 3) == 0 && (plane & 3) == 0 && (stride2 & 3) == 0 &&
      (reinterpret_cast<uintptr_t>(out) & 7) == 0) {
    *reinterpret_cast<uint64_t *>(dst + j0) = *reinterpret_cast<const uint64_t *>(h);
    *reinterpret_cast<uint64_t *>(dst + plane + j0) = *reinterpret_cast<const uint64_t *>(l);
  } else {
#pragma unroll
    for (int e = 0; e < 4; ++e)
      if (j0 + e < cols) {
        dst[j0 + e] = h[e];
        dst[plane + j0 + e] = l[e];
      }
  }
}

// the normals of gpvae_randn(GPVAE_F32), times 2^12, as two fp16 planes
__global__ __launch_bounds__(256) void randn_h2_kernel(int64_t count, uint64_t seed, uint64_t stream_id,
                                                       _Float16 *out, int64_t plane) {
  const int64_t nq = count / 4;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nq;
       q += (int64_t)gridDim.x * blockDim.x) {
    float z[4];
    philox_normals4<float>(q, seed, stream_id, z);
    _Float16 h[4], l[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) split2_f16(z[e] * H2_E_SCALE, h[e], l[e]);
    *reinterpret_cast<uint64_t *>(out + q * 4) = *reinterpret_cast<const uint64_t *>(h);
    *reinterpret_cast<uint64_t *>(out + plane + q * 4) = *reinterpret_cast<const uint64_t *>(l);
  }
}

// One workgroup per 256 x 128 (eight waves of 64 x 64) or 128 x 128 (four waves) tile of Y, heaviest column tiles first
// (their k range grows with the column index), plain round-robin placement over the XCDs as in draw_kernel.
// NP = 2 (fp16 x 2): the planes hold scaled operands; the accumulators are multiplied by scale_p[0] * escale
// (powers of two: exact) before the epilogue.
template <int WFM, int NP, int NW = 4>
__global__ __launch_bounds__(64 * NW, 1) void draw_b3_kernel(
    int n, int nb, const uint16_t *L3, int64_t ldl, int64_t planeL, const uint16_t *E3, int64_t lde,
    int64_t planeE, const float *E2, int64_t lde2, const float *sigma_p, float *F, int64_t ldf,
    float *Y, int64_t ldy, const float *scale_p, float escale) {
  using Pipe = PipeB3T<WFM, NP, NW>;
  __shared__ __align__(16) unsigned char lds[Pipe::LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrt = nb / Pipe::TM, nct = n / B3_TN;
  const int ct = nct - 1 - (int)(blockIdx.x / nrt), rt = (int)(blockIdx.x % nrt);
  const int64_t m0 = (int64_t)rt * Pipe::TM, c0 = (int64_t)ct * B3_TN;
  const int kend = (ct + 1) * B3_TN;
  const int wm = (wave >> 1) * (32 * WFM), wn = (wave & 1) * 64;

  Pipe P;
  P.init(E3 + m0 * lde, lde, planeE, L3 + c0 * ldl, ldl, planeL, tid, wm, wn);
#pragma unroll
  for (int i = 0; i < Pipe::WFM; ++i)
#pragma unroll
    for (int j = 0; j < Pipe::WFN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) P.acc[i][j][e] = 0.f;
  P.run(lds, 0, kend);

  const float sigma = (E2 != nullptr) ? sigma_p[0] : 0.f;
  if constexpr (NP == 2) {
    const float sc = scale_p[0] * escale;
#pragma unroll
    for (int i = 0; i < Pipe::WFM; ++i)
#pragma unroll
      for (int j = 0; j < Pipe::WFN; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) P.acc[i][j][e] *= sc;
  }
  TileIO<float> iof, ioy, ioe;
  const int64_t rw = m0 + wm, cw = c0 + wn;
  if (F != nullptr) iof.init(F + rw * ldf + cw, ldf, lane);
  if (Y != nullptr) ioy.init(Y + rw * ldy + cw, ldy, lane);
  if (E2 != nullptr) ioe.init(E2 + rw * lde2 + cw, lde2, lane);
#pragma unroll
  for (int fm = 0; fm < Pipe::WFM; ++fm) {
    float e2[Pipe::WFN][16];
    if (E2 != nullptr) {
#pragma unroll
      for (int fn = 0; fn < Pipe::WFN; ++fn)
#pragma unroll
        for (int r = 0; r < 16; ++r) e2[fn][r] = ioe.load(fm, fn, r);
    }
#pragma unroll
    for (int fn = 0; fn < Pipe::WFN; ++fn)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float f = P.acc[fm][fn][r];
        if (F != nullptr) iof.store(fm, fn, r, f);
        if (Y != nullptr) ioy.store(fm, fn, r, (E2 != nullptr) ? f + sigma * e2[fn][r] : f);
      }
  }
}

}  // namespace gpvae

using namespace gpvae;

extern "C" int gpvae_split_b3(void *stream, int64_t rows, int64_t cols, int batch, const void *X,
                              int64_t ldx, int64_t strideX, int lower, void *out3, int64_t ld3,
                              int64_t plane, int64_t stride3) {
  GPVAE_REQUIRE(rows >= 0, 2, "gpvae_split_b3: rows < 0");
  GPVAE_REQUIRE(cols >= 0, 3, "gpvae_split_b3: cols < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 4, "gpvae_split_b3: batch out of range [0,65535]");
  if (rows == 0 || cols == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(rows <= 65535, 2, "gpvae_split_b3: more than 65535 rows; split the call");
  GPVAE_REQUIRE(X, 5, "gpvae_split_b3: X is NULL");
  GPVAE_REQUIRE(ldx >= cols, 6, "gpvae_split_b3: ldx < cols");
  GPVAE_REQUIRE(out3, 9, "gpvae_split_b3: out3 is NULL");
  GPVAE_REQUIRE(ld3 >= cols, 10, "gpvae_split_b3: ld3 < cols");
  GPVAE_REQUIRE(plane >= rows * ld3, 11, "gpvae_split_b3: plane stride < rows*ld3");
  GPVAE_REQUIRE(batch == 1 || stride3 >= 3 * plane, 12, "gpvae_split_b3: stride3 < 3*plane");
  hipLaunchKernelGGL(split_b3_kernel, dim3((unsigned)cdiv(cols, 1024), (unsigned)rows, (unsigned)batch),
                     dim3(256), 0, (hipStream_t)stream, rows, cols, (const float *)X, ldx, strideX, lower,
                     (__bf16 *)out3, ld3, plane, stride3);
  GPVAE_CHECK_LAUNCH("gpvae_split_b3");
  return 0;
}

extern "C" int gpvae_randn_b3(void *stream, int64_t count, uint64_t seed, uint64_t stream_id,
                              void *out3, int64_t plane) {
  GPVAE_REQUIRE(count >= 0 && count % 4 == 0, 2, "gpvae_randn_b3: count must be a non-negative multiple of 4");
  if (count == 0) return 0;
  GPVAE_REQUIRE(out3 && (reinterpret_cast<uintptr_t>(out3) & 7) == 0, 5, "gpvae_randn_b3: out3 NULL or not 8-byte aligned");
  GPVAE_REQUIRE(plane >= count && plane % 4 == 0, 6, "gpvae_randn_b3: plane stride < count or not a multiple of 4");
  int64_t blocks = cdiv(count / 4, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(randn_b3_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, count, seed,
                     stream_id, (__bf16 *)out3, plane);
  GPVAE_CHECK_LAUNCH("gpvae_randn_b3");
  return 0;
}

extern "C" int gpvae_draw_b3_supported(int64_t n, int64_t nb) {
  return n >= B3_TN && n % B3_TN == 0 && nb >= 128 && nb % 128 == 0 && n <= (1 << 20) && nb <= (1 << 20);
}

extern "C" int gpvae_draw_b3(void *stream, int64_t n, int64_t nb, const void *L3, int64_t ldl,
                             int64_t planeL, const void *E3, int64_t lde, int64_t planeE,
                             const void *E2, int64_t lde2, const void *sigma, void *F, int64_t ldf,
                             void *Y, int64_t ldy) {
  GPVAE_REQUIRE(n >= 0, 2, "gpvae_draw_b3: n < 0");
  GPVAE_REQUIRE(nb >= 0, 3, "gpvae_draw_b3: nb < 0");
  if (n == 0 || nb == 0) return 0;
  GPVAE_REQUIRE(gpvae_draw_b3_supported(n, nb), 2,
                "gpvae_draw_b3: needs n %% 128 == 0 and nb %% 128 == 0 (use gpvae_draw otherwise)");
  GPVAE_REQUIRE(L3 && aligned16(L3), 4, "gpvae_draw_b3: L3 NULL or not 16-byte aligned");
  GPVAE_REQUIRE(ldl >= n && ldl % 8 == 0, 5, "gpvae_draw_b3: ldl < n or not a multiple of 8");
  GPVAE_REQUIRE(planeL >= n * ldl && planeL % 8 == 0, 6, "gpvae_draw_b3: planeL < n*ldl or not a multiple of 8");
  GPVAE_REQUIRE(E3 && aligned16(E3), 7, "gpvae_draw_b3: E3 NULL or not 16-byte aligned");
  GPVAE_REQUIRE(lde >= n && lde % 8 == 0, 8, "gpvae_draw_b3: lde < n or not a multiple of 8");
  GPVAE_REQUIRE(planeE >= nb * lde && planeE % 8 == 0, 9, "gpvae_draw_b3: planeE < nb*lde or not a multiple of 8");
  GPVAE_REQUIRE(PipeB3::usable(lde, ldl), 8, "gpvae_draw_b3: leading dimensions too large");
  // 128-row tiles when the 256-row tiling would leave CUs without a workgroup (or nb is not a multiple of 256)
  const bool small = nb % B3_TM != 0 || (nb / B3_TM) * (n / B3_TN) < 256;
  if (E2) {
    GPVAE_REQUIRE(lde2 >= n, 11, "gpvae_draw_b3: lde2 < n");
    GPVAE_REQUIRE(sigma, 12, "gpvae_draw_b3: sigma is NULL with E2 given");
  }
  GPVAE_REQUIRE(F || Y, 15, "gpvae_draw_b3: both F and Y are NULL");
  if (F) GPVAE_REQUIRE(ldf >= n, 14, "gpvae_draw_b3: ldf < n");
  if (Y) GPVAE_REQUIRE(ldy >= n, 16, "gpvae_draw_b3: ldy < n");
  const int64_t ldmax = std::max(std::max(F ? ldf : 0, Y ? ldy : 0), E2 ? lde2 : 0);
  GPVAE_REQUIRE(128 * ldmax * 4 < ((int64_t)1 << 31), 14, "gpvae_draw_b3: output leading dimension too large");
  const int64_t grid = (nb / (small ? 128 : B3_TM)) * (n / B3_TN);
  GPVAE_REQUIRE(grid < ((int64_t)1 << 31), 3, "gpvae_draw_b3: too many tiles");
  if (small)
    hipLaunchKernelGGL((draw_b3_kernel<2, 3>), dim3((unsigned)grid), dim3(NTHREADS), 0, (hipStream_t)stream, (int)n,
                       (int)nb, (const uint16_t *)L3, ldl, planeL, (const uint16_t *)E3, lde, planeE,
                       (const float *)E2, lde2, (const float *)sigma, (float *)F, ldf, (float *)Y, ldy,
                       (const float *)nullptr, 1.0f);
  else  // 256 x 128 tiles on eight waves
    hipLaunchKernelGGL((draw_b3_kernel<2, 3, 8>), dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, (int)n,
                       (int)nb, (const uint16_t *)L3, ldl, planeL, (const uint16_t *)E3, lde, planeE,
                       (const float *)E2, lde2, (const float *)sigma, (float *)F, ldf, (float *)Y, ldy,
                       (const float *)nullptr, 1.0f);
  GPVAE_CHECK_LAUNCH("gpvae_draw_b3");
  return 0;
}

// ---- fp16 x 2 entry points ------------------------------------------------------------------------------
extern "C" int gpvae_split_h2(void *stream, int64_t rows, int64_t cols, int batch, const void *X, int64_t ldx,
                              int64_t strideX, int lower, const void *bound, void *out2, int64_t ld2,
                              int64_t plane, int64_t stride2, void *inv_scale) {
  GPVAE_REQUIRE(rows >= 0, 2, "gpvae_split_h2: rows < 0");
  GPVAE_REQUIRE(cols >= 0, 3, "gpvae_split_h2: cols < 0");
  GPVAE_REQUIRE(batch >= 0 && batch <= 65535, 4, "gpvae_split_h2: batch out of range [0,65535]");
  if (rows == 0 || cols == 0 || batch == 0) return 0;
  GPVAE_REQUIRE(rows <= 65535, 2, "gpvae_split_h2: more than 65535 rows; split the call");
  GPVAE_REQUIRE(X, 5, "gpvae_split_h2: X is NULL");
  GPVAE_REQUIRE(ldx >= cols, 6, "gpvae_split_h2: ldx < cols");
  GPVAE_REQUIRE(bound, 9, "gpvae_split_h2: bound is NULL");
  GPVAE_REQUIRE(out2, 10, "gpvae_split_h2: out2 is NULL");
  GPVAE_REQUIRE(ld2 >= cols, 11, "gpvae_split_h2: ld2 < cols");
  GPVAE_REQUIRE(plane >= rows * ld2, 12, "gpvae_split_h2: plane stride < rows*ld2");
  GPVAE_REQUIRE(batch == 1 || stride2 >= 2 * plane, 13, "gpvae_split_h2: stride2 < 2*plane");
  GPVAE_REQUIRE(inv_scale, 14, "gpvae_split_h2: inv_scale is NULL");
  hipLaunchKernelGGL(split_h2_kernel, dim3((unsigned)cdiv(cols, 1024), (unsigned)rows, (unsigned)batch), dim3(256), 0,
                     (hipStream_t)stream, rows, cols, (const float *)X, ldx, strideX, lower, (const float *)bound,
                     (_Float16 *)out2, ld2, plane, stride2, (float *)inv_scale);
  GPVAE_CHECK_LAUNCH("gpvae_split_h2");
  return 0;
}

extern "C" int gpvae_randn_h2(void *stream, int64_t count, uint64_t seed, uint64_t stream_id, void *out2,
                              int64_t plane) {
  GPVAE_REQUIRE(count >= 0 && count % 4 == 0, 2, "gpvae_randn_h2: count must be a non-negative multiple of 4");
  if (count == 0) return 0;
  GPVAE_REQUIRE(out2 && (reinterpret_cast<uintptr_t>(out2) & 7) == 0, 5, "gpvae_randn_h2: out2 NULL or not 8-byte aligned");
  GPVAE_REQUIRE(plane >= count && plane % 4 == 0, 6, "gpvae_randn_h2: plane stride < count or not a multiple of 4");
  int64_t blocks = cdiv(count / 4, 256);
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(randn_h2_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, count, seed,
                     stream_id, (_Float16 *)out2, plane);
  GPVAE_CHECK_LAUNCH("gpvae_randn_h2");
  return 0;
}

extern "C" int gpvae_draw_h2(void *stream, int64_t n, int64_t nb, const void *L2, int64_t ldl, int64_t planeL,
                             const void *inv_scale_L, const void *E2h, int64_t lde, int64_t planeE,
                             const void *E2, int64_t lde2, const void *sigma, void *F, int64_t ldf, void *Y,
                             int64_t ldy) {
  GPVAE_REQUIRE(n >= 0, 2, "gpvae_draw_h2: n < 0");
  GPVAE_REQUIRE(nb >= 0, 3, "gpvae_draw_h2: nb < 0");
  if (n == 0 || nb == 0) return 0;
  GPVAE_REQUIRE(gpvae_draw_b3_supported(n, nb), 2,
                "gpvae_draw_h2: needs n %% 128 == 0 and nb %% 128 == 0 (use gpvae_draw otherwise)");
  GPVAE_REQUIRE(L2 && aligned16(L2), 4, "gpvae_draw_h2: L2 NULL or not 16-byte aligned");
  GPVAE_REQUIRE(ldl >= n && ldl % 8 == 0, 5, "gpvae_draw_h2: ldl < n or not a multiple of 8");
  GPVAE_REQUIRE(planeL >= n * ldl && planeL % 8 == 0, 6, "gpvae_draw_h2: planeL < n*ldl or not a multiple of 8");
  GPVAE_REQUIRE(inv_scale_L, 7, "gpvae_draw_h2: inv_scale_L is NULL");
  GPVAE_REQUIRE(E2h && aligned16(E2h), 8, "gpvae_draw_h2: E2h NULL or not 16-byte aligned");
  GPVAE_REQUIRE(lde >= n && lde % 8 == 0, 9, "gpvae_draw_h2: lde < n or not a multiple of 8");
  GPVAE_REQUIRE(planeE >= nb * lde && planeE % 8 == 0, 10, "gpvae_draw_h2: planeE < nb*lde or not a multiple of 8");
  GPVAE_REQUIRE(PipeB3::usable(lde, ldl), 9, "gpvae_draw_h2: leading dimensions too large");
  if (E2) {
    GPVAE_REQUIRE(lde2 >= n, 12, "gpvae_draw_h2: lde2 < n");
    GPVAE_REQUIRE(sigma, 13, "gpvae_draw_h2: sigma is NULL with E2 given");
  }
  GPVAE_REQUIRE(F || Y, 16, "gpvae_draw_h2: both F and Y are NULL");
  if (F) GPVAE_REQUIRE(ldf >= n, 15, "gpvae_draw_h2: ldf < n");
  if (Y) GPVAE_REQUIRE(ldy >= n, 17, "gpvae_draw_h2: ldy < n");
  const int64_t ldmax = std::max(std::max(F ? ldf : 0, Y ? ldy : 0), E2 ? lde2 : 0);
  GPVAE_REQUIRE(128 * ldmax * 4 < ((int64_t)1 << 31), 15, "gpvae_draw_h2: output leading dimension too large");
  const bool small = nb % B3_TM != 0 || (nb / B3_TM) * (n / B3_TN) < 256;
  const int64_t grid = (nb / (small ? 128 : B3_TM)) * (n / B3_TN);
  GPVAE_REQUIRE(grid < ((int64_t)1 << 31), 3, "gpvae_draw_h2: too many tiles");
  if (small)
    hipLaunchKernelGGL((draw_b3_kernel<2, 2>), dim3((unsigned)grid), dim3(NTHREADS), 0, (hipStream_t)stream, (int)n,
                       (int)nb, (const uint16_t *)L2, ldl, planeL, (const uint16_t *)E2h, lde, planeE,
                       (const float *)E2, lde2, (const float *)sigma, (float *)F, ldf, (float *)Y, ldy,
                       (const float *)inv_scale_L, 1.0f / H2_E_SCALE);
  else
    hipLaunchKernelGGL((draw_b3_kernel<2, 2, 8>), dim3((unsigned)grid), dim3(512), 0, (hipStream_t)stream, (int)n,
                       (int)nb, (const uint16_t *)L2, ldl, planeL, (const uint16_t *)E2h, lde, planeE,
                       (const float *)E2, lde2, (const float *)sigma, (float *)F, ldf, (float *)Y, ldy,
                       (const float *)inv_scale_L, 1.0f / H2_E_SCALE);
  GPVAE_CHECK_LAUNCH("gpvae_draw_h2");
  return 0;
}
