// Micro-benchmark (dev tool, not part of the library): how do f32 MFMA and other instruction classes
// share a gfx950 SIMD — inside one wave and between the two waves resident on a SIMD?
//   build: hipcc -O3 --offload-arch=gfx950 tools/ub_mfma_valu.hip -o tools/ub_mfma_valu
// Each block has 512 threads = 8 waves = 2 per SIMD (wave w runs on SIMD w % 4).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int ITERS = 2000;

// role 0: MFMA stream (4 independent accumulators), role 1: VALU stream, role 2: LDS read stream,
// role 3: idle, role 4: MFMA with NV VALU ops after each MFMA (same wave), role 5: MFMA with an
// LDS read after each MFMA
template <int NV>
__device__ __forceinline__ void mfma_mix(f32x16 (&acc)[4], float a, float b, int &x, int y) {
  for (int i = 0; i < ITERS; ++i) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
#pragma unroll
      for (int v = 0; v < NV; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
  }
}

__global__ __launch_bounds__(512) void ub(int roleA, int roleB, int nv, long long *cycles, float *sink,
                                          int *isink, const float *gsrc) {
  __shared__ __align__(16) float lds[4096];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int role = (wave < 4) ? roleA : roleB;
  for (int i = threadIdx.x; i < 4096; i += 512) lds[i] = (float)i;
  __syncthreads();
  f32x16 acc[4];
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  float a = 1.0f + lane, b = 0.5f;
  int x = lane, y = 3;
  float facc = 0.f;
  const long long w0 = wall_clock64();
  const long long t0 = clock64();
  if (role == 0) {
    mfma_mix<0>(acc, a, b, x, y);
  } else if (role == 1) {
    for (int i = 0; i < ITERS * 4; ++i) {
#pragma unroll
      for (int v = 0; v < 16; ++v) asm volatile("v_add_u32 %0, %0, %1" : "+v"(x) : "v"(y));
    }
  } else if (role == 2) {
    const volatile float *p = lds + lane;
    for (int i = 0; i < ITERS * 4; ++i) {
#pragma unroll
      for (int v = 0; v < 8; ++v) facc += p[v * 65];
    }
  } else if (role == 6) {  // pure LDS reads, no VALU: 16 ds_read_b128 per slot-group
    const unsigned addr = (unsigned)(lane * 144);
    for (int i = 0; i < ITERS; ++i) {
      f32x4 r0, r1, r2, r3;
      asm volatile(
          "ds_read_b128 %0, %4\n ds_read_b128 %1, %4 offset:32\n ds_read_b128 %2, %4 offset:64\n"
          "ds_read_b128 %3, %4 offset:96\n s_waitcnt lgkmcnt(0)"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
          : "v"(addr));
    }
  } else if (role == 7) {  // pure global loads (saddr + fixed voffset), no VALU
    const unsigned voff = (unsigned)(threadIdx.x * 16);
    const float *base = gsrc + (size_t)blockIdx.x * 65536;
    for (int i = 0; i < ITERS; ++i) {
      f32x4 r0, r1, r2, r3;
      const float *b = base + (i & 15) * 4096;  // scalar address arithmetic
      asm volatile(
          "global_load_dwordx4 %0, %4, %5\n global_load_dwordx4 %1, %4, %5 offset:1024\n"
          "global_load_dwordx4 %2, %4, %5 offset:2048\n global_load_dwordx4 %3, %4, %5 offset:3072\n"
          "s_waitcnt vmcnt(0)"
          : "=v"(r0), "=v"(r1), "=v"(r2), "=v"(r3)
          : "v"(voff), "s"(b));
    }
  } else if (role == 8) {  // pure LDS writes
    const unsigned addr = (unsigned)(lane * 144);
    f32x4 r0 = {1.f, 2.f, 3.f, 4.f};
    for (int i = 0; i < ITERS; ++i) {
      asm volatile(
          "ds_write_b128 %0, %1\n ds_write_b128 %0, %1 offset:32\n ds_write_b128 %0, %1 offset:64\n"
          "ds_write_b128 %0, %1 offset:96\n s_waitcnt lgkmcnt(0)"
          :
          : "v"(addr), "v"(r0));
    }
  } else if (role == 4) {
    switch (nv) {
      case 1: mfma_mix<1>(acc, a, b, x, y); break;
      case 2: mfma_mix<2>(acc, a, b, x, y); break;
      case 4: mfma_mix<4>(acc, a, b, x, y); break;
      case 8: mfma_mix<8>(acc, a, b, x, y); break;
      case 12: mfma_mix<12>(acc, a, b, x, y); break;
      case 16: mfma_mix<16>(acc, a, b, x, y); break;
      default: mfma_mix<0>(acc, a, b, x, y); break;
    }
  } else if (role == 5) {
    const volatile float *p = lds + lane;
    for (int i = 0; i < ITERS; ++i) {
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
        facc += p[j * 65];
        facc += p[j * 65 + 300];
      }
    }
  }
  const long long t1 = clock64();
  const long long w1 = wall_clock64();
  if (lane == 0) cycles[blockIdx.x * 8 + wave] = t1 - t0;
  if (lane == 0 && blockIdx.x == 0 && wave == 0) cycles[gridDim.x * 8] = w1 - w0;
  float s = facc;
  for (int j = 0; j < 4; ++j)
    for (int r = 0; r < 16; ++r) s += acc[j][r];
  if (s == 12345.678f) sink[0] = s;
  if (x == 123456789) isink[0] = x;
}

int main() {
  long long *d;
  float *sink;
  int *isink;
  const int nblk = 256;
  hipMalloc(&d, (nblk * 8 + 1) * sizeof(long long));
  hipMalloc(&sink, 4);
  hipMalloc(&isink, 4);
  float *gsrc;
  hipMalloc(&gsrc, (size_t)nblk * 65536 * 4 + (1 << 20));
  hipMemset(gsrc, 0, (size_t)nblk * 65536 * 4 + (1 << 20));
  std::vector<long long> h(nblk * 8 + 1);
  struct Cfg { int a, b, nv; const char *what; };
  const Cfg cfgs[] = {
      {0, 3, 0, "A: mfma alone (1 wave/SIMD)"},
      {0, 0, 0, "A,B: mfma + mfma"},
      {1, 3, 0, "A: valu alone"},
      {0, 1, 0, "A: mfma, B: valu (other wave)"},
      {2, 3, 0, "A: lds read alone"},
      {0, 2, 0, "A: mfma, B: lds read (other wave)"},
      {4, 3, 1, "A: mfma + 1 valu each (same wave)"},
      {4, 3, 2, "A: mfma + 2 valu each"},
      {4, 3, 4, "A: mfma + 4 valu each"},
      {4, 3, 8, "A: mfma + 8 valu each"},
      {4, 3, 12, "A: mfma + 12 valu each"},
      {4, 3, 16, "A: mfma + 16 valu each"},
      {4, 4, 4, "A,B: mfma + 4 valu each, both waves"},
      {4, 4, 8, "A,B: mfma + 8 valu each, both waves"},
      {6, 3, 0, "A: pure ds_read_b128 x4 (no valu) alone"},
      {0, 6, 0, "A: mfma, B: pure ds_read_b128"},
      {7, 3, 0, "A: pure global_load x4 (no valu) alone"},
      {0, 7, 0, "A: mfma, B: pure global_load"},
      {8, 3, 0, "A: pure ds_write_b128 x4 alone"},
      {0, 8, 0, "A: mfma, B: pure ds_write_b128"},
      {5, 3, 0, "A: mfma + 2 lds reads each"},
      {5, 5, 0, "A,B: mfma + 2 lds reads each"},
  };
  for (const Cfg &c : cfgs) {
    for (int rep = 0; rep < 2; ++rep) {
      hipLaunchKernelGGL(ub, dim3(nblk), dim3(512), 0, 0, c.a, c.b, c.nv, d, sink, isink, gsrc);
      hipDeviceSynchronize();
    }
    hipMemcpy(h.data(), d, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
    double sa = 0, sb = 0;
    for (int b = 0; b < nblk; ++b)
      for (int w = 0; w < 8; ++w) (w < 4 ? sa : sb) += (double)h[b * 8 + w];
    sa /= nblk * 4.0;
    sb /= nblk * 4.0;
    // clock64 = s_memtime: 100 MHz constant on gfx9; report raw ticks and per-MFMA ticks
    printf("%-44s A %10.0f ticks (%.2f / mfma-slot)   B %10.0f ticks   wall(100MHz) %lld\n", c.what,
           sa, sa / (ITERS * 4.0), sb, h[nblk * 8]);
  }
  return 0;
}
