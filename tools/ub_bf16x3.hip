// Micro-benchmark (dev tool, not part of the library): an f32 product on the bf16 matrix pipe.
// gfx950 has no f32-input MFMA faster than v_mfma_f32_32x32x2_f32 (1/16 of the bf16 rate).  Split
// every f32 operand exactly into three bf16 pieces, x = x1 + x2 + x3 (8 + 8 + 8 mantissa bits), and
// sum the six products of weight >= 2^-16 with f32 accumulation:
//   a b ~ a1 b1 + (a1 b2 + a2 b1) + (a1 b3 + a2 b2 + a3 b1)          (dropped terms <= 2^-24 |a b|)
// This program measures (1) the error of that against an f64 reference next to the f32 MFMA's own,
// on C = A B^T with K = 4096, and (2) the MFMA-pipe time of both per 16 k values (8 f32 MFMAs vs
// 6 bf16 MFMAs of 32x32x16).  Result quoted in DESIGN.md §7.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ub_bf16x3.hip -o tools/ub_bf16x3
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ __bf16 to_bf16(float x) { return (__bf16)x; }  // round to nearest even

// accuracy: one wave per block computes a 32 x 32 tile of C = A Bt^T, A, Bt: [32 x K] row-major
__global__ void acc_kernel(int K, const float *A, const float *Bt, float *Cf32, float *Cb3, float *Ch2 = nullptr) {
  const int lane = threadIdx.x, r = lane & 31, h = lane >> 5;
  f32x16 c32, c3, ch;
  for (int i = 0; i < 16; ++i) c32[i] = c3[i] = ch[i] = 0.f;
  for (int k = 0; k < K; k += 2) c32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[r * K + k + h], Bt[r * K + k + h], c32, 0, 0, 0);
  for (int k = 0; k < K; k += 16) {
    bf16x8 a[3], b[3];
    for (int j = 0; j < 8; ++j) {
      float x = A[r * K + k + 8 * h + j], y = Bt[r * K + k + 8 * h + j];
      for (int p = 0; p < 3; ++p) {
        a[p][j] = to_bf16(x); x -= (float)a[p][j];
        b[p][j] = to_bf16(y); y -= (float)b[p][j];
      }
    }
    // smallest terms first
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[2], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[1], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2], b[0], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[1], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1], b[0], c3, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0], b[0], c3, 0, 0, 0);
  }
  // two fp16 pieces per operand (11 + 11 significand bits), three products: a0 b0 + a0 b1 + a1 b0
  for (int k = 0; k < K; k += 16) {
    f16x8 a[2], b[2];
    for (int j = 0; j < 8; ++j) {
      float x = A[r * K + k + 8 * h + j], y = Bt[r * K + k + 8 * h + j];
      a[0][j] = (_Float16)x; a[1][j] = (_Float16)(x - (float)a[0][j]);
      b[0][j] = (_Float16)y; b[1][j] = (_Float16)(y - (float)b[0][j]);
    }
    ch = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[1], ch, 0, 0, 0);
    ch = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[1], b[0], ch, 0, 0, 0);
    ch = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[0], b[0], ch, 0, 0, 0);
  }
  for (int i = 0; i < 16; ++i) {
    const int row = (i & 3) + 8 * (i >> 2) + 4 * h;
    Cf32[row * 32 + r] = c32[i];
    Cb3[row * 32 + r] = c3[i];
    if (Ch2) Ch2[row * 32 + r] = ch[i];
  }
}

// pipe time: 256 threads per block, every wave streams MFMAs on 4 independent accumulators
template <int MODE>
__global__ __launch_bounds__(256) void rate_kernel(int iters, float *sink) {
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) acc[j][i] = 0.f;
  const float a = 1.0f + threadIdx.x, b = 0.5f;
  bf16x8 ha, hb;
  for (int j = 0; j < 8; ++j) { ha[j] = (__bf16)(a + j); hb[j] = (__bf16)b; }
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      if (MODE == 0) {  // 16 k values in f32: 8 MFMAs
#pragma unroll
        for (int s = 0; s < 8; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
      } else {  // 16 k values as six bf16 products
#pragma unroll
        for (int s = 0; s < 6; ++s) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ha, hb, acc[j], 0, 0, 0);
      }
    }
  }
  float s = 0.f;
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 16; ++i) s += acc[j][i];
  if (s == 12345.678f) sink[0] = s;
}

int main() {
  const int K = 4096;
  std::vector<float> A(32 * K), Bt(32 * K);
  srand(1);
  auto rnd = []() { return (float)rand() / RAND_MAX * 2.f - 1.f; };
  // a Cholesky-like worst case as well: rows nearly parallel (long-lengthscale Gram factors)
  for (int i = 0; i < 32 * K; ++i) { A[i] = 1.0f + 1e-3f * rnd(); Bt[i] = rnd(); }
  float *dA, *dB, *dC1, *dC2, *dC3;
  hipMalloc(&dA, 32 * K * 4); hipMalloc(&dB, 32 * K * 4); hipMalloc(&dC1, 4096); hipMalloc(&dC2, 4096); hipMalloc(&dC3, 4096);
  hipMemcpy(dA, A.data(), 32 * K * 4, hipMemcpyHostToDevice);
  hipMemcpy(dB, Bt.data(), 32 * K * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(acc_kernel, dim3(1), dim3(64), 0, 0, K, dA, dB, dC1, dC2, dC3);
  std::vector<float> C1(1024), C2(1024), C3(1024);
  hipMemcpy(C1.data(), dC1, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(C2.data(), dC2, 4096, hipMemcpyDeviceToHost);
  hipMemcpy(C3.data(), dC3, 4096, hipMemcpyDeviceToHost);
  double e1 = 0, e2 = 0, e3 = 0, nrm = 0;
  for (int i = 0; i < 32; ++i)
    for (int j = 0; j < 32; ++j) {
      double ref = 0, mag = 0;
      for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * Bt[j * K + k]; mag += fabs((double)A[i * K + k] * Bt[j * K + k]); }
      e1 = fmax(e1, fabs(C1[i * 32 + j] - ref) / mag);
      e2 = fmax(e2, fabs(C2[i * 32 + j] - ref) / mag);
      e3 = fmax(e3, fabs(C3[i * 32 + j] - ref) / mag);
      nrm = fmax(nrm, mag);
    }
  printf("K = %d: max |err| / sum|a b|   f32 MFMA %.3e   bf16 x 3 (6 products) %.3e   fp16 x 2 (3 products, operands in [-1, 1]) %.3e   (f32 eps = 6e-8)\n", K, e1, e2, e3);

  float *sink; hipMalloc(&sink, 4);
  const int iters = 2000, blocks = 1024;
  for (int mode = 0; mode < 2; ++mode) {
    hipEvent_t e0, e1v; hipEventCreate(&e0); hipEventCreate(&e1v);
    for (int rep = 0; rep < 2; ++rep) {
      hipEventRecord(e0);
      if (mode == 0) hipLaunchKernelGGL(rate_kernel<0>, dim3(blocks), dim3(256), 0, 0, iters, sink);
      else hipLaunchKernelGGL(rate_kernel<1>, dim3(blocks), dim3(256), 0, 0, iters, sink);
      hipEventRecord(e1v); hipEventSynchronize(e1v);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1v);
    // f32-equivalent flop: 2 * 32 * 32 * 16 per (accumulator, iteration) per wave
    const double flop = 2.0 * 32 * 32 * 16 * 4.0 * iters * 4.0 * blocks;
    printf("%s: %.3f ms, %.1f f32-equivalent TFLOP/s\n", mode == 0 ? "f32 MFMA 32x32x2 x 8     " : "bf16 MFMA 32x32x16 x 6   ", ms, flop / ms / 1e9);
  }
  return 0;
}
