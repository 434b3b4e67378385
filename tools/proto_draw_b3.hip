// Prototype (dev tool, NOT part of the library): the MVN draw Y = E L^T with f32 operands pre-split
// into three bf16 planes each and six bf16 MFMA products per k16 step (tools/ub_bf16x3.hip has the
// error analysis).  Measures what a first, un-pipelined kernel of that kind reaches at the C3 shape
// (n = B = 4096) next to the library's f32-MFMA draw (0.55 ms); result quoted in DESIGN.md §7.
//   build: hipcc -O3 --offload-arch=gfx950 [-DPBK=64 | -DPDB | -DPBIG] tools/proto_draw_b3.hip -o tools/proto_draw_b3
// Measured (MI355X, n = nb = 4096; error 1e-7 of sum|e l| in every variant):
//   default (k-tile 32, one LDS buffer, 2 blocks per CU)            0.50-0.52 ms
//   -DPBK=64 (k-tile 64, 123 KB of LDS, one block per CU)           0.51 ms
//   -DPDB    (LDS double buffer, one barrier per k-tile, 1 block)   0.54 ms
//   -DPBIG   (256 x 256 tile, 512 threads, waves of 128 x 64)       0.58 ms
// All four sit at the same ~0.5 ms although the MFMA pipe needs 0.18 ms: neither operand traffic (PBIG
// halves it) nor the barrier count (PDB halves it) is the limit; the fragment reads of one k16 step
// (12 ds_read_b128) are issued and waited for in front of its 24 MFMAs by every wave at once.  The
// library kernel needs the in-wave software pipeline of gemm_core.hpp's PipeKC (next k16 step's
// fragments in flight under this step's MFMAs) - next round's first item.
// Simple structure on purpose: 128 x 128 tile per 256-thread block (4 waves, 64 x 64 each), k-tile 32,
// one LDS buffer, global loads one tile ahead in registers; occupancy (2 blocks per CU) hides the rest.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

#ifndef PBK
#define PBK 32
#endif
#ifdef PBIG  // 256 x 256 tile, 512 threads: 8 waves as 2 x 4, each 128 x 64 (halves the operand traffic)
constexpr int TS = 256, NT = 512, WFM = 4, WFN = 2, WAVES_N = 4;
#else
constexpr int TS = 128, NT = 256, WFM = 2, WFN = 2, WAVES_N = 2;
#endif
constexpr int BK = PBK, ROWB = BK * 2 + 16;  // k-tile, LDS bytes per image row (+ pad)
constexpr int VPR = BK / 8;                  // 16-byte vectors per image row
constexpr int NQ = 6 * TS * VPR / NT;        // vectors per thread and k-tile

// x -> three bf16 planes (plane stride `ps` elements); lower: zero strictly above the diagonal
__global__ void split3(const float *x, __bf16 *out, int64_t rows, int64_t cols, int64_t ps, int lower) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= rows * cols) return;
  float v = x[i];
  if (lower && (i % cols) > (i / cols)) v = 0.f;
  for (int p = 0; p < 3; ++p) {
    const __bf16 h = (__bf16)v;
    out[p * ps + i] = h;
    v -= (float)h;
  }
}

#ifdef PDB
constexpr int NBUF = 2;  // LDS double buffer, one barrier per k-tile, one block per CU
#else
constexpr int NBUF = 1;
#endif
__global__ __launch_bounds__(NT, (PBK == 32 && NBUF == 1 && TS == 128) ? 2 : 1) void draw_b3(int n, int nb, const __bf16 *E3, const __bf16 *L3, float *Y) {
  __shared__ __align__(16) unsigned char lds_all[NBUF * 6 * TS * ROWB];  // A planes 0..2, B planes 0..2
  unsigned char *lds = lds_all;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrt = nb / TS, nct = n / TS;
  const int ct = nct - 1 - (int)(blockIdx.x / nrt), rt = (int)(blockIdx.x % nrt);  // heaviest first
  const int64_t m0 = (int64_t)rt * TS, c0 = (int64_t)ct * TS;
  const int kend = (ct + 1) * TS;
  const int64_t psE = (int64_t)nb * n, psL = (int64_t)n * n;
  const int wm = (wave / WAVES_N) * 32 * WFM, wn = (wave % WAVES_N) * 32 * WFN;
  const int r = lane & 31, h = lane >> 5;

  // loader: 6 images x 128 rows x BK/8 vectors (16 B) per k-tile, NQ per thread (12 at BK = 32)
  i32x4 pre[NQ];
  auto gload = [&](int k0) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int v = tid + q * NT;
      const int img = v / (TS * VPR), row = (v % (TS * VPR)) / VPR, c = v % VPR;
      const __bf16 *src = img < 3 ? E3 + img * psE + (m0 + row) * n + k0 + c * 8
                                  : L3 + (img - 3) * psL + (c0 + row) * n + k0 + c * 8;
      pre[q] = *reinterpret_cast<const i32x4 *>(src);
    }
  };
  auto lstore = [&](unsigned char *dst) {
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
      const int v = tid + q * NT;
      const int img = v / (TS * VPR), row = (v % (TS * VPR)) / VPR, c = v % VPR;
      *reinterpret_cast<i32x4 *>(dst + (img * TS + row) * ROWB + c * 16) = pre[q];
    }
  };

  f32x16 acc[WFM][WFN];
  for (int i = 0; i < WFM; ++i)
    for (int j = 0; j < WFN; ++j)
      for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

  auto compute = [&](const unsigned char *cur) {
#pragma unroll
    for (int s = 0; s < BK / 16; ++s) {
      bf16x8 a[3][WFM], b[3][WFN];
#pragma unroll
      for (int p = 0; p < 3; ++p) {
#pragma unroll
        for (int f = 0; f < WFM; ++f)
          a[p][f] = *reinterpret_cast<const bf16x8 *>(cur + (p * TS + wm + 32 * f + r) * ROWB + s * 32 + h * 16);
#pragma unroll
        for (int f = 0; f < WFN; ++f)
          b[p][f] = *reinterpret_cast<const bf16x8 *>(cur + ((3 + p) * TS + wn + 32 * f + r) * ROWB + s * 32 + h * 16);
      }
#pragma unroll
      for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
        for (int fn = 0; fn < WFN; ++fn) {
          f32x16 c = acc[fm][fn];
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][fm], b[2][fn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][fm], b[1][fn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[2][fm], b[0][fn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][fm], b[1][fn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[1][fm], b[0][fn], c, 0, 0, 0);
          c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[0][fm], b[0][fn], c, 0, 0, 0);
          acc[fm][fn] = c;
        }
    }
  };
  if (NBUF == 1) {
    gload(0);
    for (int k0 = 0; k0 < kend; k0 += BK) {
      lstore(lds);
      __syncthreads();
      if (k0 + BK < kend) gload(k0 + BK);
      compute(lds);
      __syncthreads();
    }
  } else {
    // tile t in buffer t & 1; per tile: write tile t+1 (loaded during tile t-1) into the other buffer,
    // request tile t+2, MFMAs of tile t, one barrier
    constexpr int BUFB = 6 * TS * ROWB;
    gload(0);
    lstore(lds);
    if (BK < kend) gload(BK);
    __syncthreads();
    int t = 0;
    for (int k0 = 0; k0 < kend; k0 += BK, ++t) {
      if (k0 + BK < kend) lstore(lds + ((t + 1) & 1) * BUFB);
      if (k0 + 2 * BK < kend) gload(k0 + 2 * BK);
      compute(lds + (t & 1) * BUFB);
      __syncthreads();
    }
  }
#pragma unroll
  for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
    for (int fn = 0; fn < WFN; ++fn)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm + 32 * fm + (e & 3) + 8 * (e >> 2) + 4 * h, col = wn + 32 * fn + r;
        Y[(m0 + row) * n + c0 + col] = acc[fm][fn][e];
      }
}

int main() {
  const int n = 4096, nb = 4096;
  const size_t cnt = (size_t)n * n;
  std::vector<float> E(cnt), L(cnt);
  srand(3);
  for (size_t i = 0; i < cnt; ++i) { E[i] = (float)rand() / RAND_MAX * 2.f - 1.f; L[i] = ((float)rand() / RAND_MAX * 2.f - 1.f) * 0.05f; }
  float *dE, *dL, *dY; __bf16 *dE3, *dL3;
  hipMalloc(&dE, cnt * 4); hipMalloc(&dL, cnt * 4); hipMalloc(&dY, cnt * 4);
  hipMalloc(&dE3, cnt * 6); hipMalloc(&dL3, cnt * 6);
  hipMemcpy(dE, E.data(), cnt * 4, hipMemcpyHostToDevice);
  hipMemcpy(dL, L.data(), cnt * 4, hipMemcpyHostToDevice);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms;
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(split3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, dE, dE3, (int64_t)nb, (int64_t)n, (int64_t)cnt, 0);
    hipLaunchKernelGGL(split3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, dL, dL3, (int64_t)n, (int64_t)n, (int64_t)cnt, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
  }
  hipEventElapsedTime(&ms, e0, e1);
  printf("split of E and L into 3 bf16 planes each: %.3f ms (both)\n", ms);
  const unsigned grid = (nb / TS) * (n / TS);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(draw_b3, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  hipEventRecord(e0);
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(draw_b3, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  hipEventRecord(e1); hipEventSynchronize(e1);
  hipEventElapsedTime(&ms, e0, e1);
  ms /= reps;
  printf("draw_b3 n = nb = 4096: %.3f ms  = %.1f f32-equivalent TFLOP/s (n^2 nb flop; library f32 draw: 0.55 ms, 125)\n",
         ms, (double)n * n * nb / ms / 1e9);
  std::vector<float> Yh(cnt);
  hipMemcpy(Yh.data(), dY, cnt * 4, hipMemcpyDeviceToHost);
  double emax = 0;
  for (int t = 0; t < 200; ++t) {
    const int i = rand() % nb, j = rand() % n;
    double ref = 0, mag = 0;
    for (int k = 0; k <= j; ++k) { ref += (double)E[(size_t)i * n + k] * L[(size_t)j * n + k]; mag += fabs((double)E[(size_t)i * n + k] * L[(size_t)j * n + k]); }
    emax = fmax(emax, fabs(Yh[(size_t)i * n + j] - ref) / (mag > 0 ? mag : 1));
  }
  printf("max |err| / sum|e l| over 200 sampled entries: %.3e\n", emax);
  return 0;
}
