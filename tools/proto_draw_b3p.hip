// Prototype 2 (dev tool, NOT part of the library): the bf16 x 3 draw of tools/proto_draw_b3.hip with
// the in-wave software pipeline of gemm_core.hpp's PipeKC and one wave per SIMD:
//   * 256 x 128 tile per 256-thread block (4 waves as 2 x 2, each 128 x 64: 18 fragment reads per 48
//     MFMAs instead of 12 per 24), one block per CU, 512 registers per lane;
//   * LDS double buffer of k-tiles of 32; image [plane][8-k chunk][row][16 B] (+32 B per chunk
//     plane): fragment reads (4 x 16 lanes, 64 banks) and tile writes (8 x 8 lanes over a 128-byte
//     window) are both conflict-free without row padding (SQ_LDS_BANK_CONFLICT 0; +64 B: 12 %);
//   * per k-tile: first half = MFMAs of k16 step 0 with the reads of step 1's fragments, then the
//     ds_writes of tile t+1 each followed by the global load of the same register for tile t+2
//     (a whole tile of latency cover), barrier, second half = MFMAs of step 1 with the reads of
//     tile t+1's step-0 fragments (__builtin_amdgcn_sched_group_barrier fixes the interleave).
// Measured (MI355X, n = nb = 4096, error 9e-8 of sum|e l|):
//   this file                                        0.395-0.41 ms  (library f32 draw 0.55; un-pipelined
//                                                                    prototype 0.50)
//   -DSCHED=1 (writes early, loads late: 78 instead of 96 MFMAs of load lead)   0.44 ms
//   -DNOSGB   (no sched_group_barrier: the compiler's own interleave)           0.47 ms
//   -DEXP_NOLOAD  (no global loads in the loop; results wrong)                  0.34 ms
//   -DEXP_NOSTAGE (no loads, no LDS writes, half the fragment reads)            0.29 ms
//   MFMA pipe alone: 405 k cycles = 0.24 ms at the 1.7 GHz the chip holds under this load
// PMC: 606 k cycles per launch, MFMA pipe 67 % busy, no bank conflicts.  So of the 0.16 ms above the
// MFMA time, ~0.05 is barriers / prologue / epilogue, ~0.05 the LDS writes (ds_write_b128 holds the
// wave's issue for 13-26 cycles: the 128 x 64 wave tile needs 18 of them per 96 MFMAs) and ~0.06
// the global loads (their latency under load exceeds 78 MFMA times).  Levers for the library
// version: 256 x 256 block tiles (0.75x the staged bytes per MFMA), the B operand straight from
// global memory into fragment registers, a deeper load lead.
//   build: hipcc -O3 --offload-arch=gfx950 tools/proto_draw_b3p.hip -o tools/proto_draw_b3p
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

__device__ i32x4 raw_buffer_load_b128(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.v4i32");

#ifndef PAD
#define PAD 32  // ds_write_b128 banks in groups of 8 lanes over a 128-byte window: chunk planes 32 B apart
#endif
constexpr int TM = 256, TN = 128, NT = 256, BK = 32, WFM = 4, WFN = 2;
constexpr int CPA = TM * 16 + PAD, CPB = TN * 16 + PAD;  // bytes per (plane, chunk) of the A / B image
constexpr int A_BYTES = 12 * CPA, STAGE = 12 * (CPA + CPB);
constexpr int NQA = 12, NQB = 6;                       // 16-byte vectors per thread and k-tile

#ifndef SCHED
#define SCHED 0
#endif
#ifndef LEAD
#define LEAD 6
#endif
#ifndef NOSGB
#define SGB(mask, n) __builtin_amdgcn_sched_group_barrier(mask, n, 0)
#else
#define SGB(mask, n)
#endif

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

__device__ __forceinline__ i32x4 make_rsrc(const void *p) {
  const uint64_t a = reinterpret_cast<uint64_t>(p);
  i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(uint32_t)a);
  r[1] = __builtin_amdgcn_readfirstlane((int)(uint32_t)(a >> 32));
  r[2] = -1;
  r[3] = 0x00020000;
  return r;
}

struct Pipe {
  i32x4 pre[NQA + NQB];
  bf16x8 fa[2][3][WFM], fb[2][3][WFN];
  f32x16 acc[WFM][WFN];
  i32x4 rsE, rsL;
  int voff;          // per-lane byte offset of the loads (row tid/4, chunk tid%4)
  int sA, sB;        // scalar byte offsets of the tile origins
  int psE2, psL2, n2;  // plane strides and row stride in bytes
  int wr_a, wr_b;    // per-lane LDS byte offsets of the tile writes
  int rd_a, rd_b;    // per-lane LDS byte offsets of the fragment reads

  __device__ __forceinline__ void gload(int k0) {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
      pre[q] = raw_buffer_load_b128(rsE, voff, sA + (q >> 2) * psE2 + (q & 3) * 64 * n2 + k0 * 2, 0);
#pragma unroll
    for (int q = 0; q < NQB; ++q)
      pre[NQA + q] = raw_buffer_load_b128(rsL, voff, sB + (q >> 1) * psL2 + (q & 1) * 64 * n2 + k0 * 2, 0);
  }
  template <int BUF>
  __device__ __forceinline__ void lstore(unsigned char *lds) {
#pragma unroll
    for (int q = 0; q < NQA; ++q)
      *reinterpret_cast<i32x4 *>(lds + BUF * STAGE + wr_a + (q >> 2) * 4 * CPA + (q & 3) * 1024) = pre[q];
#pragma unroll
    for (int q = 0; q < NQB; ++q)
      *reinterpret_cast<i32x4 *>(lds + BUF * STAGE + wr_b + (q >> 1) * 4 * CPB + (q & 1) * 1024) = pre[NQA + q];
  }
  template <int BUF, int S>
  __device__ __forceinline__ void read_sub(const unsigned char *lds) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int f = 0; f < WFM; ++f)
        fa[S][p][f] = *reinterpret_cast<const bf16x8 *>(lds + BUF * STAGE + rd_a + (p * 4 + 2 * S) * CPA + f * 512);
#pragma unroll
      for (int f = 0; f < WFN; ++f)
        fb[S][p][f] = *reinterpret_cast<const bf16x8 *>(lds + BUF * STAGE + rd_b + (p * 4 + 2 * S) * CPB + f * 512);
    }
  }
  template <int S>
  __device__ __forceinline__ void mma_sub() {
#pragma unroll
    for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
      for (int fn = 0; fn < WFN; ++fn) {
        f32x16 c = acc[fm][fn];
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][0][fm], fb[S][2][fn], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][1][fm], fb[S][1][fn], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][2][fm], fb[S][0][fn], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][0][fm], fb[S][1][fn], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][1][fm], fb[S][0][fn], c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][0][fm], fb[S][0][fn], c, 0, 0, 0);
        acc[fm][fn] = c;
      }
  }

  // k-tile in buffer CUR.  HAS1 / HAS2: tiles t+1 / t+2 exist.
  template <int CUR, bool HAS1_, bool HAS2_>
  __device__ __forceinline__ void tile(unsigned char *lds, int k0) {
    constexpr int NXT = CUR ^ 1;
#if defined(EXP_NOLOAD)   // timing experiments only (results wrong): no global loads in the loop
    constexpr bool HAS1 = HAS1_, HAS2 = false;
#elif defined(EXP_NOSTAGE)  // ... and no LDS writes either
    constexpr bool HAS1 = false, HAS2 = false;
#else
    constexpr bool HAS1 = HAS1_, HAS2 = HAS2_;
#endif
    // ---- first half: MFMAs of step 0 | reads of step 1's fragments, then tile t+1 written to the
    // other buffer and, register by register, tile t+2 requested (a whole tile of latency cover)
    __builtin_amdgcn_sched_barrier(0);
    read_sub<CUR, 1>(lds);
    if (HAS1) lstore<NXT>(lds);
    if (HAS2) gload(k0 + 2 * BK);
    mma_sub<0>();
#if SCHED == 1  // one memory instruction per MFMA gap: 9 x 2 reads, 18 writes, 18 loads
#pragma unroll
    for (int i = 0; i < 9; ++i) { SGB(0x008, 1); SGB(0x100, 2); }
    if (HAS1) {
#pragma unroll
      for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x200, 1); }
    }
    if (HAS2) {
#pragma unroll
      for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x020, 1); }
    }
#else
#pragma unroll
    for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x100, 1); }   // MFMA, DS read
    SGB(0x008, LEAD);
    if (HAS1) {
#pragma unroll
      for (int i = 0; i < 18; ++i) {
        SGB(0x008, 1);
        SGB(0x200, 1);            // DS write
        if (HAS2) SGB(0x020, 1);  // VMEM read
      }
    }
#endif
    SGB(0x008, 48);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
    // ---- second half: MFMAs of step 1 | reads of tile t+1's step-0 fragments
    if (HAS1) read_sub<NXT, 0>(lds);
    mma_sub<1>();
    if (HAS1) {
#if SCHED == 1
#pragma unroll
      for (int i = 0; i < 9; ++i) { SGB(0x008, 1); SGB(0x100, 2); }
#else
#pragma unroll
      for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x100, 1); }
#endif
    }
    SGB(0x008, 48);
    __builtin_amdgcn_sched_barrier(0);
  }
};

__global__ __launch_bounds__(NT, 1) void draw_b3p(int n, int nb, const __bf16 *E3, const __bf16 *L3, float *Y) {
  __shared__ __align__(16) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrt = nb / TM, nct = n / TN;
  const int ct = nct - 1 - (int)(blockIdx.x / nrt), rt = (int)(blockIdx.x % nrt);  // heaviest first
  const int m0 = rt * TM, c0 = ct * TN;
  const int kend = (ct + 1) * TN;
  const int wm = (wave >> 1) * 32 * WFM, wn = (wave & 1) * 32 * WFN;
  const int r = lane & 31, h = lane >> 5;

  Pipe P;
  P.rsE = make_rsrc(E3);
  P.rsL = make_rsrc(L3);
  P.n2 = n * 2;
  P.psE2 = nb * n * 2;
  P.psL2 = n * n * 2;
  P.voff = (tid >> 2) * P.n2 + (tid & 3) * 16;
  P.sA = m0 * P.n2;
  P.sB = c0 * P.n2;
  P.wr_a = (tid & 3) * CPA + (tid >> 2) * 16;
  P.wr_b = A_BYTES + (tid & 3) * CPB + (tid >> 2) * 16;
  P.rd_a = h * CPA + (wm + r) * 16;
  P.rd_b = A_BYTES + h * CPB + (wn + r) * 16;
#pragma unroll
  for (int i = 0; i < WFM; ++i)
#pragma unroll
    for (int j = 0; j < WFN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) P.acc[i][j][e] = 0.f;

  const int nt = kend / BK;  // even, >= 4
  P.gload(0);
  P.lstore<0>(lds);
  P.gload(BK);
  __syncthreads();
  P.read_sub<0, 0>(lds);
  int k0 = 0;
  for (int i = 0; i < (nt - 2) / 2; ++i, k0 += 2 * BK) {
    P.tile<0, true, true>(lds, k0);
    P.tile<1, true, true>(lds, k0 + BK);
  }
  P.tile<0, true, false>(lds, k0);
  P.tile<1, false, false>(lds, k0 + BK);

#pragma unroll
  for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
    for (int fn = 0; fn < WFN; ++fn)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int row = wm + 32 * fm + (e & 3) + 8 * (e >> 2) + 4 * h, col = wn + 32 * fn + r;
        Y[(int64_t)(m0 + row) * n + c0 + col] = P.acc[fm][fn][e];
      }
}

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at line %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)

int main() {
  const int n = 4096, nb = 4096;
  const size_t cnt = (size_t)n * n;
  std::vector<float> E(cnt), L(cnt);
  srand(3);
  for (size_t i = 0; i < cnt; ++i) { E[i] = (float)(rand() / (double)RAND_MAX * 2. - 1.); L[i] = (float)((rand() / (double)RAND_MAX * 2. - 1.) * 0.05); }
  float *dE, *dL, *dY; __bf16 *dE3, *dL3;
  CK(hipMalloc(&dE, cnt * 4)); CK(hipMalloc(&dL, cnt * 4)); CK(hipMalloc(&dY, cnt * 4));
  CK(hipMalloc(&dE3, cnt * 6)); CK(hipMalloc(&dL3, cnt * 6));
  CK(hipMemcpy(dE, E.data(), cnt * 4, hipMemcpyHostToDevice));
  CK(hipMemcpy(dL, L.data(), cnt * 4, hipMemcpyHostToDevice));
  CK(hipMemset(dY, 0, cnt * 4));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  float ms;
  hipLaunchKernelGGL(split3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, dE, dE3, (int64_t)nb, (int64_t)n, (int64_t)cnt, 0);
  hipLaunchKernelGGL(split3, dim3((unsigned)((cnt + 255) / 256)), dim3(256), 0, 0, dL, dL3, (int64_t)n, (int64_t)n, (int64_t)cnt, 1);
  const unsigned grid = (nb / TM) * (n / TN);
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(draw_b3p, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(draw_b3p, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("draw_b3p n = nb = 4096: %.3f ms  = %.1f f32-equivalent TFLOP/s (n^2 nb flop; library f32 draw: 0.55 ms, 125)\n",
         ms, (double)n * n * nb / ms / 1e9);
  std::vector<float> Yh(cnt);
  CK(hipMemcpy(Yh.data(), dY, cnt * 4, hipMemcpyDeviceToHost));
  double emax = 0;
  for (int t = 0; t < 400; ++t) {
    const int i = rand() % nb, j = rand() % n;
    double ref = 0, mag = 0;
    for (int k = 0; k <= j; ++k) { ref += (double)E[(size_t)i * n + k] * L[(size_t)j * n + k]; mag += fabs((double)E[(size_t)i * n + k] * L[(size_t)j * n + k]); }
    emax = fmax(emax, fabs(Yh[(size_t)i * n + j] - ref) / (mag > 0 ? mag : 1));
  }
  printf("max |err| / sum|e l| over 400 sampled entries: %.3e\n", emax);
  return 0;
}
