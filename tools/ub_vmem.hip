// Micro-benchmark (dev tool, not part of the library): what a 16-byte-per-lane buffer load costs a CU
// when a wave-level instruction touches 16 half cache lines (64 B per row: PipeB3's k-tile of 32 bf16)
// against 8 full lines (128 B per row), with the operand rows 8 KB apart as in the C3 draw.
// One 256-thread workgroup per CU (100 KB of LDS requested), each wave keeps up to 24 loads in flight.
//   build: hipcc -O3 --offload-arch=gfx950 tools/ub_vmem.hip -o tools/ub_vmem
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef int i32x4 __attribute__((ext_vector_type(4)));

// MODE 0: lane -> (row = lane / 4, 16 B chunk = lane % 4): 16 rows x 64 B per instruction
// MODE 1: lane -> (row = lane / 8, chunk = lane % 8):      8 rows x 128 B per instruction
// rows_per_wg rows of `ld` bytes; every instruction group covers the workgroup's rows once for a 64-byte
// (MODE 0) or 128-byte (MODE 1) k-slice, then the slice moves on; `span` bytes of each row are cycled through.
template <int MODE>
__global__ __launch_bounds__(256) void vmem_kernel(const char *base, long ld, int rows_per_wg, int nblocks_data,
                                                   int span, int iters, int *sink) {
  extern __shared__ char lds[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const char *wg = base + (long)(blockIdx.x % nblocks_data) * rows_per_wg * ld;
  constexpr int RPI = MODE == 0 ? 16 : 8;        // rows per instruction
  constexpr int BPR = MODE == 0 ? 64 : 128;      // bytes per row per instruction
  const int r0 = MODE == 0 ? lane / 4 : lane / 8, c0 = (MODE == 0 ? lane % 4 : lane % 8) * 16;
  const int ipw = rows_per_wg / RPI / 4;         // instructions per wave and k-slice
  i32x4 acc = {0, 0, 0, 0};
  int k = 0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll 4
    for (int i = 0; i < ipw; ++i) {
      const int row = (wave * ipw + i) * RPI + r0;
      const i32x4 v = *reinterpret_cast<const i32x4 *>(wg + (long)row * ld + k + c0);
      acc ^= v;
    }
    k += BPR;
    if (k >= span) k = 0;
  }
  if (acc[0] == 0x12345678 && acc[1] == 7) sink[tid] = acc[2] + acc[3] + lds[tid];
}

template <int MODE>
static double run(const char *d, long ld, int rows, int nb, int span, int iters, int *sink) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  hipLaunchKernelGGL((vmem_kernel<MODE>), dim3(256), dim3(256), 100 * 1024, 0, d, ld, rows, nb, span, iters / 4, sink);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL((vmem_kernel<MODE>), dim3(256), dim3(256), 100 * 1024, 0, d, ld, rows, nb, span, iters, sink);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  return ms;
}

int main() {
  const long ld = 8192;  // bytes between rows (4096 bf16)
  const int rows = 256;
  char *d; int *sink;
  const long bytes = 64L * rows * ld;  // 64 row blocks: 128 MiB
  hipMalloc(&d, bytes); hipMemset(d, 1, bytes); hipMalloc(&sink, 4096);
  hipFuncSetAttribute((const void *)vmem_kernel<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  hipFuncSetAttribute((const void *)vmem_kernel<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
  // per "k-slice of 64 B per row": MODE 0 issues rows/16 instructions per workgroup, MODE 1 rows/8 per 128 B
  for (int nb : {1, 16, 64}) {
    for (int span : {256, 8192}) {
      const int it0 = 4096, it1 = 2048;  // same bytes: it0 * 64 B = it1 * 128 B per row
      const double t0 = run<0>(d, ld, rows, nb, span, it0, sink), t1 = run<1>(d, ld, rows, nb, span, it1, sink);
      const double instr0 = (double)it0 * rows / 16, instr1 = (double)it1 * rows / 8;  // per workgroup
      const double gb = (double)it0 * 64 * rows * 256 / 1e9;
      printf("row blocks %2d span %4d B: half lines %.3f ms (%.0f ns per wave-instruction and CU, %.0f GB/s) | full lines %.3f ms (%.0f ns, %.0f GB/s)\n",
             nb, span, t0, t0 * 1e6 / instr0, gb / (t0 * 1e-3), t1, t1 * 1e6 / instr1, gb / (t1 * 1e-3));
    }
  }
  return 0;
}
