// Prototype 3 (dev tool, NOT part of the library): proto_draw_b3p.hip with the operand tiles brought in by
// LDS-DMA (buffer_load_dwordx4 ... lds) instead of registers + ds_write_b128:
//   * LDS image per plane row-major [row][64 B] (one k-tile of 32 bf16), 16-byte slots XOR-swizzled with
//     (row >> 2) & 3 — applied on the per-lane SOURCE address (the DMA's destination is lane-linear: one
//     wave instruction fills 1 KiB = 16 rows), and on the fragment reads;
//   * two buffers; the DMA of tile t+2 is issued right after the mid-tile barrier of tile t (all reads of
//     that buffer are done) and waited for (vmcnt(0)) before the mid-tile barrier of tile t+1: one k-tile
//     of lead, no staging registers, no ds_write.
//   build: hipcc -O3 --offload-arch=gfx950 tools/proto_draw_b3d.hip -o tools/proto_draw_b3d
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_void;

// LDS-DMA piece: 64 lanes x 16 B -> 1 KiB at LDS byte address lds_dst (wave-uniform).  Inline asm, not the
// llvm.amdgcn.raw.buffer.load.lds intrinsic: hipcc treats the intrinsic as an LDS store and puts an
// s_waitcnt vmcnt(0) in front of every later ds_read (the pipeline then runs 5x slower); the wait for
// these loads is the explicit vmcnt(0) before the mid-tile barrier.  M0 is written in the same statement.
__device__ __forceinline__ void dma_piece(i32x4 rsrc, int lds_dst, int voffset, int soffset) {
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds"
               :: "s"(lds_dst), "v"(voffset), "s"(rsrc), "s"(soffset) : "memory");
}

constexpr int TM = 256, TN = 128, NT = 256, BK = 32, WFM = 4, WFN = 2;
constexpr int PLA = TM * 64, PLB = TN * 64;           // bytes per plane of the A / B image
constexpr int A_BYTES = 3 * PLA, STAGE = 3 * (PLA + PLB);

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
  bf16x8 fa[2][3][WFM], fb[2][3][WFN];
  f32x16 acc[WFM][WFN];
  i32x4 rsE[3], rsL[3];
  int voff;           // per-lane byte offset of a DMA piece: row lane/4, swizzled chunk
  int n2;             // row stride in bytes
  int wave;
  int lds0;           // LDS byte address of the image (scalar)
  int rd_a[2], rd_b[2];  // per-lane LDS byte offsets of the fragment reads, per k16 step

  // every wave brings in 4 of the 16 pieces of each A plane and 2 of the 8 pieces of each B plane
  template <int BUF> __device__ __forceinline__ void dma(unsigned char *lds, int k0) {
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int row0 = (wave * 4 + i) * 16;
        dma_piece(rsE[pl], lds0 + BUF * STAGE + pl * PLA + row0 * 64, voff, row0 * n2 + k0 * 2);
      }
#pragma unroll
    for (int pl = 0; pl < 3; ++pl)
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int row0 = (wave * 2 + i) * 16;
        dma_piece(rsL[pl], lds0 + BUF * STAGE + A_BYTES + pl * PLB + row0 * 64, voff, row0 * n2 + k0 * 2);
      }
  }
  template <int BUF, int S> __device__ __forceinline__ void read_sub(const unsigned char *lds) {
#pragma unroll
    for (int p = 0; p < 3; ++p) {
#pragma unroll
      for (int f = 0; f < WFM; ++f)
        fa[S][p][f] = *reinterpret_cast<const bf16x8 *>(lds + BUF * STAGE + rd_a[S] + p * PLA + f * 2048);
#pragma unroll
      for (int f = 0; f < WFN; ++f)
        fb[S][p][f] = *reinterpret_cast<const bf16x8 *>(lds + BUF * STAGE + rd_b[S] + p * PLB + f * 2048);
    }
  }
  template <int S> __device__ __forceinline__ void mma_sub() {
    constexpr int PA[6] = {0, 1, 2, 0, 1, 0}, PB[6] = {2, 1, 0, 1, 0, 0};
#pragma unroll
    for (int q = 0; q < 6; ++q)
#pragma unroll
      for (int fm = 0; fm < WFM; ++fm)
#pragma unroll
        for (int fn = 0; fn < WFN; ++fn)
          acc[fm][fn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[S][PA[q]][fm], fb[S][PB[q]][fn], acc[fm][fn], 0, 0, 0);
  }

  // k-tile t in buffer CUR.  HAS1 / HAS2: tiles t+1 / t+2 exist.
  template <int CUR, bool HAS1, bool HAS2> __device__ __forceinline__ void tile(unsigned char *lds, int k0) {
    constexpr int NXT = CUR ^ 1;
    __builtin_amdgcn_sched_barrier(0);
    read_sub<CUR, 1>(lds);
    mma_sub<0>();
#pragma unroll
    for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x100, 1); }
    SGB(0x008, 48);
    __builtin_amdgcn_sched_barrier(0);
    if (HAS1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // tile t+1 has landed (this wave's pieces)
    __builtin_amdgcn_s_barrier();                               // ... and everyone's; buffer CUR is free
    __builtin_amdgcn_sched_barrier(0);
    if (HAS2) dma<CUR>(lds, k0 + 2 * BK);
    if (HAS1) read_sub<NXT, 0>(lds);
    mma_sub<1>();
    if (HAS1) {
#pragma unroll
      for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x100, 1); }
    }
    if (HAS2) {
#pragma unroll
      for (int i = 0; i < 18; ++i) { SGB(0x008, 1); SGB(0x020, 1); }
    }
    SGB(0x008, 48);
    __builtin_amdgcn_sched_barrier(0);
  }
};

__global__ __launch_bounds__(NT, 1) void draw_b3d(int n, int nb, const __bf16 *E3, const __bf16 *L3, float *Y) {
  __shared__ __align__(16) unsigned char lds[2 * STAGE];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int nrt = nb / TM, nct = n / TN;
  const int ct = nct - 1 - (int)(blockIdx.x / nrt), rt = (int)(blockIdx.x % nrt);  // heaviest first
  const int m0 = rt * TM, c0 = ct * TN;
  const int kend = (ct + 1) * TN;
  const int wm = (wave >> 1) * 32 * WFM, wn = (wave & 1) * 32 * WFN;
  const int r = lane & 31, h = lane >> 5;

  Pipe P;
  P.n2 = n * 2;
  for (int p = 0; p < 3; ++p) {
    P.rsE[p] = make_rsrc(E3 + (int64_t)p * nb * n + (int64_t)m0 * n);
    P.rsL[p] = make_rsrc(L3 + (int64_t)p * n * n + (int64_t)c0 * n);
  }
  P.wave = __builtin_amdgcn_readfirstlane(wave);
  P.lds0 = __builtin_amdgcn_readfirstlane((int)(uintptr_t)(lds_void *)lds);
  P.voff = (lane >> 2) * P.n2 + (((lane & 3) ^ ((lane >> 4) & 3)) * 16);
  {
    const int x = (r >> 2) & 3;
    for (int S = 0; S < 2; ++S) {
      P.rd_a[S] = (wm + r) * 64 + (((2 * S + h) ^ x) * 16);
      P.rd_b[S] = A_BYTES + (wn + r) * 64 + (((2 * S + h) ^ x) * 16);
    }
  }
#pragma unroll
  for (int i = 0; i < WFM; ++i)
#pragma unroll
    for (int j = 0; j < WFN; ++j)
#pragma unroll
      for (int e = 0; e < 16; ++e) P.acc[i][j][e] = 0.f;

  const int nt = kend / BK;  // even, >= 4
  P.dma<0>(lds, 0);
  P.dma<1>(lds, BK);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
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
  for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(draw_b3d, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0));
  const int reps = 10;
  for (int rep = 0; rep < reps; ++rep) hipLaunchKernelGGL(draw_b3d, dim3(grid), dim3(NT), 0, 0, n, nb, dE3, dL3, dY);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  CK(hipEventElapsedTime(&ms, e0, e1));
  ms /= reps;
  printf("draw_b3d n = nb = 4096: %.3f ms  = %.1f f32-equivalent TFLOP/s (n^2 nb flop; library f32 draw: 0.55 ms, 125)\n",
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
