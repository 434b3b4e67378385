// Micro-benchmark (dev tool, not part of the library): the bf16 MFMA rate the chip SUSTAINS with the draw's
// instruction mix of matrix work alone — one 4-wave workgroup per CU, 8 accumulator tiles per wave, 48 MFMAs per
// k16 step cycling over 12 A and 6 B fragment registers — with operand bits that are (0) all zero, (1) one
// constant, (2) random per lane and register.  No memory traffic in the loop: what is left below the 2.5 PFLOP/s
// peak is clock / power management, the ceiling any bf16 x 3 kernel on this chip lives under.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ub_mfma_power.hip -o tools/ub_mfma_power
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256, 1) void mfma_kernel(const i32x4 *data, int iters, float *sink) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x;
  bf16x8 fa[3][4], fb[3][2];
  for (int p = 0; p < 3; ++p) {
    for (int f = 0; f < 4; ++f) { i32x4 v = data[(p * 6 + f) * 256 + tid]; fa[p][f] = *reinterpret_cast<bf16x8 *>(&v); }
    for (int f = 0; f < 2; ++f) { i32x4 v = data[(p * 6 + 4 + f) * 256 + tid]; fb[p][f] = *reinterpret_cast<bf16x8 *>(&v); }
  }
  f32x16 acc[4][2];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
  constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int fm = 0; fm < 4; ++fm)
#pragma unroll
        for (int fn = 0; fn < 2; ++fn)
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[PA[q]][fm], fb[PB[q]][fn], acc[fm][fn], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 2; ++j)
      for (int e = 0; e < 16; ++e) s += acc[i][j][e];
  if (s == 12345.678f) sink[0] = s + lds[tid];
}


typedef float f32x4 __attribute__((ext_vector_type(4)));
// the same work per wave (128 x 64 output, six piece products) on v_mfma_f32_16x16x32_bf16: 8 x 4 tiles of 16 x 16, one
// k32 step = 192 MFMAs of 16 cycles (two k16 steps of the kernel above)
__global__ __launch_bounds__(256, 1) void mfma16_kernel(const i32x4 *data, int iters, float *sink) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x;
  bf16x8 fa[3][8], fb[3][4];
  for (int p = 0; p < 3; ++p) {
    for (int f = 0; f < 8; ++f) { i32x4 v = data[((p * 6 + f) % 18) * 256 + tid]; v[0] ^= f; fa[p][f] = *reinterpret_cast<bf16x8 *>(&v); }
    for (int f = 0; f < 4; ++f) { i32x4 v = data[((p * 6 + 4 + f) % 18) * 256 + tid]; v[1] ^= f; fb[p][f] = *reinterpret_cast<bf16x8 *>(&v); }
  }
  f32x4 acc[8][4];
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
  constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
  for (int it = 0; it < iters; it += 2) {
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int fm = 0; fm < 8; ++fm)
#pragma unroll
        for (int fn = 0; fn < 4; ++fn)
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fa[PA[q]][fm], fb[PB[q]][fn], acc[fm][fn], 0, 0, 0);
  }
  float s = 0.f;
  for (int i = 0; i < 8; ++i)
    for (int j = 0; j < 4; ++j)
      for (int e = 0; e < 4; ++e) s += acc[i][j][e];
  if (s == 12345.678f) sink[0] = s + lds[tid];
}

int main() {
  const int n = 18 * 256;
  std::vector<i32x4> h(n);
  i32x4 *d; float *sink;
  hipMalloc(&d, n * sizeof(i32x4)); hipMalloc(&sink, 4);
  hipFuncSetAttribute((const void *)mfma_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  hipFuncSetAttribute((const void *)mfma16_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 140 * 1024);
  const int iters = 6000;  // 6000 k16 steps x 48 MFMAs x 32 cycles = 9.2 M cycles: ~4-5 ms per launch
  const double flop = 2.0 * 32 * 32 * 16 * 48.0 * iters * 4.0 * 256;
  srand(3);
  for (int mode = 0; mode < 3; ++mode) {
    for (int i = 0; i < n; ++i)
      for (int e = 0; e < 4; ++e) {
        unsigned lo, hi;
        if (mode == 0) lo = hi = 0;
        else if (mode == 1) lo = hi = 0x3F80;  // 1.0
        else {  // random bf16 in (-2, 2): sign, exponent 0x7E..0x7F, random mantissa
          lo = ((rand() & 1) << 15) | ((0x7E + (rand() & 1)) << 7) | (rand() & 0x7F);
          hi = ((rand() & 1) << 15) | ((0x7E + (rand() & 1)) << 7) | (rand() & 0x7F);
        }
        h[i][e] = (int)(lo | (hi << 16));
      }
    hipMemcpy(d, h.data(), n * sizeof(i32x4), hipMemcpyHostToDevice);
    for (int shape = 0; shape < 2; ++shape) {
      auto launch = [&](int it) {
        if (shape == 0) hipLaunchKernelGGL(mfma_kernel, dim3(256), dim3(256), 140 * 1024, 0, d, it, sink);
        else hipLaunchKernelGGL(mfma16_kernel, dim3(256), dim3(256), 140 * 1024, 0, d, it, sink);
      };
      launch(100);
      hipDeviceSynchronize();
      printf("%s %s:", shape == 0 ? "32x32x16" : "16x16x32", mode == 0 ? "zero operands  " : mode == 1 ? "constant 1.0    " : "random operands ");
      hipEvent_t ev[9];
      for (int i = 0; i < 9; ++i) hipEventCreate(&ev[i]);
      hipEventRecord(ev[0]);
      for (int r = 0; r < 8; ++r) {
        launch(iters);
        hipEventRecord(ev[r + 1]);
      }
      hipEventSynchronize(ev[8]);
      for (int r = 0; r < 8; ++r) {
        float ms; hipEventElapsedTime(&ms, ev[r], ev[r + 1]);
        printf(" %.0f", flop / ms / 1e9);
      }
      printf("  TFLOP/s per launch (8 launches of ~4 ms back to back)\n");
    }
  }
  return 0;
}
