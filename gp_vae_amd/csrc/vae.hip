// vae.hip — VAE encoder/decoder, one-particle ELBO forward + backward, Adam, chain decoder
// (SURVEY.md §8 rows a5, a6, a7, a11).
//
// Replaces, operation for operation:
//   stax encoder/decoder       /root/reference/src/model/vae.py:72-96
//   vae_model / vae_guide      vae.py:98-137 under Trace_ELBO() (vae.py:213-218): loss = -ELBO,
//                              summed over batch and dims, obs noise 0.1 (vae.py:117)
//   optim.Adam                 vae.py:212 (numpyro/jax Adam: b1 .9, b2 .999, eps 1e-8)
//   decoder(params, z)         src/model/inference.py:246
//
// Structure of one ELBO evaluation (B samples of length n):
//   [n-sized products on MFMA (gemm_tile.hpp), split over K where the output is small, partial sums
//    by float atomics]
//     A1  += Y  W1                 (B x h0, K = n)          gemm<KC,MC>   reads Y once
//     DF   = (D2 V3 + c3 - Y)/s^2  (B x n,  K = h0)         gemm<KC,MC>   + the data term of the loss
//     dV3|dc3 += [D2|1]^T DF       ((h0+1) x n, K = B)      gemm<MC,MC>
//     dD2 += DF V3^T               (B x h0, K = n)          gemm<KC,KC>
//     dW1|db1 += [Y|1]^T dA1       ((n+1) x h0, K = B)      gemm<MC,MC>
//    f32 training calls with whole tiles run DF / dV3|dc3 / dD2 as ONE kernel (vae_out_fused_kernel):
//    DF then never goes to HBM and Y is read twice (A1, output layer) + once for dW1.
//   [everything between]
//     small hidden layers (the reference's 35-30-10 / 50-30-15; weights fit LDS): per block of 16
//     samples in LDS on the vector ALU —
//       vae_mid_fwd: bias+relu, h0->h1->(mu, log sd), reparametrise, z->h1->h0; latent loss terms
//       vae_mid_bwd: the same chain backwards; weight gradients into 16 replicas (+ reduce kernel)
//     wide hidden layers (up to 1024): every layer a tile product with a fused bias / relu / mask
//     epilogue (EPI_FUSED) + four elementwise kernels.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <mutex>
#include "gemm_tile.hpp"

namespace gpvae {

// ---- per-sample-block dense layers on the vector ALU ----------------------------------------------
// RB = samples per workgroup in the mid kernels (a template parameter: 16, 8 or 4 by the batch size while
// the activation buffers fit the 64 KB dynamic-LDS default, 4 or 2 for wide hidden layers; pick_rb below)

// out[r][j] = bias[j] + sum_k in[r][k] W[k*N + j]   (r < RB); in/out are LDS, W/bias global
template <typename T, int RB>
__device__ __forceinline__ void dense_fwd(const T *in, int K, const T *__restrict__ W,
                                          const T *__restrict__ bias, int N, T *out, int tid) {
  for (int idx = tid; idx < RB * N; idx += NTHREADS) {
    const int r = idx / N, j = idx % N;
    T a = bias[j];
#pragma unroll 8
    for (int k = 0; k < K; ++k) a += in[r * K + k] * W[(int64_t)k * N + j];
    out[r * N + j] = a;
  }
}
// out[r][k] = sum_j d[r][j] W[k*N + j]   (input gradient of a dense layer)
template <typename T, int RB>
__device__ __forceinline__ void dense_bwd_in(const T *d, int N, const T *__restrict__ W, int K,
                                             T *out, int tid, bool accumulate = false) {
  for (int idx = tid; idx < RB * K; idx += NTHREADS) {
    const int r = idx / K, k = idx % K;
    T a = accumulate ? out[r * K + k] : T(0);
#pragma unroll 8
    for (int j = 0; j < N; ++j) a += d[r * N + j] * W[(int64_t)k * N + j];
    out[r * K + k] = a;
  }
}
// dW[k*N + j] = sum_r in[r][k] d[r][j];  db[j] = sum_r d[r][j]   over this workgroup's samples, STORED in
// the workgroup's own copy of the mid-layer gradients (elbo_finish_kernel sums the copies in order)
template <typename T, int RB>
__device__ __forceinline__ void dense_bwd_w(const T *in, int K, const T *d, int N, T *dW, T *db,
                                            int rows, int tid) {
  for (int idx = tid; idx < (K + 1) * N; idx += NTHREADS) {
    const int k = idx / N, j = idx % N;
    T a = T(0);
    if (k < K) {
#pragma unroll 8
      for (int r = 0; r < rows; ++r) a += in[r * K + k] * d[r * N + j];
      dW[(int64_t)k * N + j] = a;
    } else {
      for (int r = 0; r < rows; ++r) a += d[r * N + j];
      db[j] = a;
    }
  }
}

template <typename T> struct VaeDims {
  int64_t B, n;
  int h0, h1, z;
};

// flat parameter layout (include/gpvae.h)
template <typename T> struct VaeParams {
  const T *W1, *b1, *W2, *b2, *Wmu, *bmu, *Wsd, *bsd, *V1, *c1, *V2, *c2, *V3, *c3;
  __host__ __device__ VaeParams(const T *p, int64_t n, int h0, int h1, int z) {
    W1 = p;            b1 = W1 + n * h0;
    W2 = b1 + h0;      b2 = W2 + (int64_t)h0 * h1;
    Wmu = b2 + h1;     bmu = Wmu + (int64_t)h1 * z;
    Wsd = bmu + z;     bsd = Wsd + (int64_t)h1 * z;
    V1 = bsd + z;      c1 = V1 + (int64_t)z * h1;
    V2 = c1 + h1;      c2 = V2 + (int64_t)h1 * h0;
    V3 = c2 + h0;      c3 = V3 + (int64_t)h0 * n;
  }
  // the contiguous run b1 .. c2 (everything but the two n-sized layers) relocated to `mid`
  // (an LDS copy): elements [n*h0, n*h0 + mid_count) of the flat vector
  __host__ __device__ static int64_t mid_count(int h0, int h1, int z) {
    return h0 + (int64_t)h0 * h1 + h1 + 2 * ((int64_t)h1 * z + z) + (int64_t)z * h1 + h1 +
           (int64_t)h1 * h0 + h0;
  }
  __device__ void relocate_mid(const T *mid, int h0, int h1, int z) {
    b1 = mid;
    W2 = b1 + h0;      b2 = W2 + h0 * h1;
    Wmu = b2 + h1;     bmu = Wmu + h1 * z;
    Wsd = bmu + z;     bsd = Wsd + h1 * z;
    V1 = bsd + z;      c1 = V1 + z * h1;
    V2 = c1 + h1;      c2 = V2 + h1 * h0;
  }
};
// Copies the mid-layer weights (b1 .. c2, a few thousand elements) into LDS once per workgroup: the
// per-sample dense layers then read weights at LDS latency instead of one L2 round trip per k.
template <typename T>
__device__ __forceinline__ void stage_mid(VaeParams<T> &P, const T *params, T *wl, int64_t n, int h0,
                                          int h1, int z, int tid) {
  const int64_t cnt = VaeParams<T>::mid_count(h0, h1, z);
  const T *src = params + n * h0;
  for (int64_t i = tid; i < cnt; i += NTHREADS) wl[i] = src[i];
  P.relocate_mid(wl, h0, h1, z);
  __syncthreads();
}
static inline int64_t vae_param_count(int64_t n, int h0, int h1, int z) {
  return n * h0 + h0 + (int64_t)h0 * h1 + h1 + 2 * ((int64_t)h1 * z + z) + (int64_t)z * h1 + h1 +
         (int64_t)h1 * h0 + h0 + (int64_t)h0 * n + n;
}

// encoder tail + reparametrisation + decoder head, RB samples per workgroup.
// A1s holds Y W1 (no bias) as nsA k-slices slabA elements apart, summed here in slice order; H1 =
// relu(Y W1 + b1) on exit.  The workgroup's latent loss terms go to lp[blockIdx.x].
template <typename T, int RB>
__global__ __launch_bounds__(NTHREADS) void vae_mid_fwd_kernel(VaeDims<T> d, const T *params,
                                                               const T *eps, const T *A1s, int nsA,
                                                               int64_t slabA, T *H1, T *H2, T *SD,
                                                               T *Z, T *D1, T *D2, T *lp) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *bufA = reinterpret_cast<T *>(smem_raw);  // [RB][max dim]
  const int mx = max(d.h0, max(d.h1, d.z));
  T *bufB = bufA + RB * mx;
  T *bufC = bufB + RB * mx;
  __shared__ T red[4];
  VaeParams<T> P(params, d.n, d.h0, d.h1, d.z);
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * RB;
  const int rows = (int)((d.B - s0 < RB) ? d.B - s0 : RB);
  const int h0 = d.h0, h1 = d.h1, z = d.z;
  stage_mid<T>(P, params, bufC + RB * mx, d.n, h0, h1, z, tid);

  // H1 = relu(A1 + b1)
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const int r = idx / h0, j = idx % h0;
    T v = T(0);
    if (r < rows) {
      const int64_t e = (s0 + r) * h0 + j;
      v = slab_sum_at<T>(A1s, nsA, slabA, e) + P.b1[j];
      v = v > T(0) ? v : T(0);
      H1[(s0 + r) * h0 + j] = v;
    }
    bufA[idx] = v;
  }
  __syncthreads();
  dense_fwd<T, RB>(bufA, h0, P.W2, P.b2, h1, bufB, tid);  // A2
  __syncthreads();
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) {
    const int r = idx / h1;
    const T v = bufB[idx] > T(0) ? bufB[idx] : T(0);
    bufB[idx] = v;
    if (r < rows) H2[(s0 + r) * h1 + idx % h1] = v;
  }
  __syncthreads();
  dense_fwd<T, RB>(bufB, h1, P.Wmu, P.bmu, z, bufA, tid);  // mu
  dense_fwd<T, RB>(bufB, h1, P.Wsd, P.bsd, z, bufC, tid);  // log sd
  __syncthreads();
  T lsum = T(0);
  for (int idx = tid; idx < RB * z; idx += NTHREADS) {
    const int r = idx / z, j = idx % z;
    T zz = T(0);
    if (r < rows) {
      const T e = eps[(s0 + r) * z + j];
      const T ls = bufC[idx];
      const T sd = dexp<T>(ls);  // stax.Exp head = std (vae.py:83)
      zz = bufA[idx] + sd * e;   // z ~ Normal(z_loc, z_std)  (vae.py:135)
      SD[(s0 + r) * z + j] = sd;
      Z[(s0 + r) * z + j] = zz;
      // -(log p(z) - log q(z)) = 0.5 z^2 - 0.5 eps^2 - log sd   (the 0.5 log 2pi terms cancel)
      lsum += T(0.5) * zz * zz - T(0.5) * e * e - ls;
    }
    bufB[idx] = zz;  // reuse bufB as Z (h1 >= ... not assumed: RB*z <= RB*mx)
  }
  __syncthreads();
  dense_fwd<T, RB>(bufB, z, P.V1, P.c1, h1, bufA, tid);  // G1
  __syncthreads();
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) {
    const int r = idx / h1;
    const T v = bufA[idx] > T(0) ? bufA[idx] : T(0);
    bufA[idx] = v;
    if (r < rows) D1[(s0 + r) * h1 + idx % h1] = v;
  }
  __syncthreads();
  dense_fwd<T, RB>(bufA, h1, P.V2, P.c2, h0, bufC, tid);  // G2
  __syncthreads();
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const int r = idx / h0;
    const T v = bufC[idx] > T(0) ? bufC[idx] : T(0);
    if (r < rows) D2[(s0 + r) * h0 + idx % h0] = v;
  }
  if (lp != nullptr) {
    lsum = wave_sum<T>(lsum);
    if ((tid & 63) == 0) red[tid >> 6] = lsum;
    __syncthreads();
    if (tid == 0) lp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
  }
}

// the same chain backwards.  dD2s holds DF V3^T as nsD partial sums slabD elements apart (summed here in
// order); dA1 (gradient at Y W1 + b1) on exit; the mid-layer weight gradients of this workgroup's samples go
// to its own copy rep[blockIdx.x].
template <typename T, int RB>
__global__ __launch_bounds__(NTHREADS) void vae_mid_bwd_kernel(VaeDims<T> d, const T *params,
                                                               T *grads, const T *eps,
                                                               const T *H1, const T *H2,
                                                               const T *SD, const T *Z,
                                                               const T *D1, const T *D2,
                                                               const T *dD2s, int nsD, int64_t slabD,
                                                               T *dA1, T *rep) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int mx = max(d.h0, max(d.h1, d.z));
  T *act = reinterpret_cast<T *>(smem_raw);  // activations of the layer input
  T *gA = act + RB * mx;                     // gradient buffers
  T *gB = gA + RB * mx;
  T *gC = gB + RB * mx;
  VaeParams<T> P(params, d.n, d.h0, d.h1, d.z);
  VaeParams<T> G(grads, d.n, d.h0, d.h1, d.z);  // same layout; const-cast below
  G.relocate_mid(rep + (int64_t)blockIdx.x * VaeParams<T>::mid_count(d.h0, d.h1, d.z), d.h0, d.h1, d.z);
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * RB;
  const int rows = (int)((d.B - s0 < RB) ? d.B - s0 : RB);
  const int h0 = d.h0, h1 = d.h1, z = d.z;
  stage_mid<T>(P, params, gC + RB * mx, d.n, h0, h1, z, tid);
  auto ld = [&](const T *src, int64_t r, int w, int j) { return r < rows ? src[(s0 + r) * w + j] : T(0); };

  // dG2 = dD2 * (D2 > 0);  act = D1
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const int r = idx / h0, j = idx % h0;
    T g2 = T(0);
    if (r < rows) {
      g2 = slab_sum_at<T>(dD2s, nsD, slabD, (s0 + r) * h0 + j);
    }
    gA[idx] = (ld(D2, r, h0, j) > T(0)) ? g2 : T(0);
  }
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) act[idx] = ld(D1, idx / h1, h1, idx % h1);
  __syncthreads();
  dense_bwd_w<T, RB>(act, h1, gA, h0, (T *)G.V2, (T *)G.c2, rows, tid);
  dense_bwd_in<T, RB>(gA, h0, P.V2, h1, gB, tid);  // dD1
  __syncthreads();
  // dG1 = dD1 * (D1 > 0);  act = Z
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) gB[idx] = (act[idx] > T(0)) ? gB[idx] : T(0);
  __syncthreads();
  for (int idx = tid; idx < RB * z; idx += NTHREADS) act[idx] = ld(Z, idx / z, z, idx % z);
  __syncthreads();
  dense_bwd_w<T, RB>(act, z, gB, h1, (T *)G.V1, (T *)G.c1, rows, tid);
  dense_bwd_in<T, RB>(gB, h1, P.V1, z, gA, tid);  // dZ (decoder part)
  __syncthreads();
  // dmu = dZ + z ; dls = dZ * eps * sd - 1   -> gA = dmu, gC = dls ;  act = H2
  for (int idx = tid; idx < RB * z; idx += NTHREADS) {
    const int r = idx / z, j = idx % z;
    const T dz = (r < rows) ? gA[idx] + act[idx] : T(0);
    gA[idx] = dz;
    gC[idx] = (r < rows) ? dz * ld(eps, r, z, j) * ld(SD, r, z, j) - T(1) : T(0);
  }
  __syncthreads();
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) act[idx] = ld(H2, idx / h1, h1, idx % h1);
  __syncthreads();
  dense_bwd_w<T, RB>(act, h1, gA, z, (T *)G.Wmu, (T *)G.bmu, rows, tid);
  dense_bwd_w<T, RB>(act, h1, gC, z, (T *)G.Wsd, (T *)G.bsd, rows, tid);
  dense_bwd_in<T, RB>(gA, z, P.Wmu, h1, gB, tid);
  __syncthreads();
  dense_bwd_in<T, RB>(gC, z, P.Wsd, h1, gB, tid, true);  // dH2
  __syncthreads();
  // dA2 = dH2 * (H2 > 0);  act = H1
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) gB[idx] = (act[idx] > T(0)) ? gB[idx] : T(0);
  __syncthreads();
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) act[idx] = ld(H1, idx / h0, h0, idx % h0);
  __syncthreads();
  dense_bwd_w<T, RB>(act, h0, gB, h1, (T *)G.W2, (T *)G.b2, rows, tid);
  dense_bwd_in<T, RB>(gB, h1, P.W2, h0, gA, tid);  // dH1
  __syncthreads();
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const int r = idx / h0, j = idx % h0;
    if (r < rows) dA1[(s0 + r) * h0 + j] = (act[idx] > T(0)) ? gA[idx] : T(0);
  }
}

// Last launch of an ELBO evaluation: every gradient that was left as partial sums is summed here in a
// fixed order, and so is the loss — no floating-point atomics anywhere on the path, the same inputs give
// the same bits.  The flat gradient vector is [W1 | b1] (nW1 elements, nsW k-slices of the dW1 product),
// the rest of the mid run (nmid elements, nrep per-workgroup copies of mc elements; nrep == 0: written
// elsewhere, skipped) and [V3 | c3] (nV3 elements, nsV partial sums).  Block 0 also sums the nlp loss parts.
template <typename T>
__global__ __launch_bounds__(256) void elbo_finish_kernel(T *grads, int64_t nW1, const T *__restrict__ SW1, int nsW,
                                                          int64_t slabW, int h0, int64_t mc,
                                                          const T *__restrict__ REP, int64_t nrep, int64_t nV3,
                                                          const T *__restrict__ SV3, int nsV, int64_t slabV,
                                                          T *loss, const T *__restrict__ LP, int64_t nlp,
                                                          T loss_const, int blocks_a) {
  __shared__ T red[256];
  const int tid = threadIdx.x;
  if ((int)blockIdx.x < blocks_a) {
    // [W1 | b1] and [V3 | c3]: one output per thread and pass, a few tens of slabs each
    if (grads != nullptr) {
      const int64_t nmid = mc - h0;
      for (int64_t i = (int64_t)blockIdx.x * 256 + tid; i < nW1 + nV3; i += (int64_t)blocks_a * 256) {
        if (i < nW1) grads[i] = slab_sum_at<T>(SW1, nsW, slabW, i);
        else grads[i + nmid] = slab_sum_at<T>(SV3, nsV, slabV, i - nW1);
      }
    }
    if (blockIdx.x == 0 && loss != nullptr) {
      T a = T(0);
      for (int64_t i = tid; i < nlp; i += 256) a += LP[i];
      red[tid] = a;
      __syncthreads();
      for (int w = 128; w > 0; w >>= 1) {
        if (tid < w) red[tid] += red[tid + w];
        __syncthreads();
      }
      if (tid == 0) loss[0] = red[0] + loss_const;
    }
    return;
  }
  // mid run: few outputs, one copy per workgroup of vae_mid_bwd_kernel (hundreds): 16 outputs per block,
  // 16 threads per output, each adding its contiguous share of the copies in order, then the 16 shares in order
  const int64_t j = h0 + ((int64_t)blockIdx.x - blocks_a) * 16 + (tid & 15);
  const int c = tid >> 4;
  const int64_t per = (nrep + 15) / 16;
  const int64_t r0 = c * per, r1 = (r0 + per < nrep) ? r0 + per : nrep;
  T a = T(0);
  if (j < mc && r0 < r1) a = slab_sum_at<T>(REP + r0 * mc, (int)(r1 - r0), mc, j);
  red[tid] = a;
  __syncthreads();
  if (c == 0 && j < mc) {
    T v = red[tid];
#pragma unroll
    for (int k = 1; k < 16; ++k) v += red[k * 16 + tid];
    grads[nW1 + (j - h0)] = v;
  }
}

// decoder head only (chains): Z -> D1 -> D2 (a11 forward), and its reverse (a11 VJP)
template <typename T, int RB>
__global__ __launch_bounds__(NTHREADS) void dec_mid_fwd_kernel(VaeDims<T> d, const T *dec_params,
                                                               const T *Zin, T *D1, T *D2) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int mx = max(d.h0, max(d.h1, d.z));
  T *bufA = reinterpret_cast<T *>(smem_raw);
  T *bufB = bufA + RB * mx;
  const int h0 = d.h0, h1 = d.h1, z = d.z;
  const T *V1 = dec_params, *c1 = V1 + (int64_t)z * h1, *V2 = c1 + h1, *c2 = V2 + (int64_t)h1 * h0;
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * RB;
  const int rows = (int)((d.B - s0 < RB) ? d.B - s0 : RB);
  for (int idx = tid; idx < RB * z; idx += NTHREADS)
    bufA[idx] = (idx / z < rows) ? Zin[(s0 + idx / z) * z + idx % z] : T(0);
  __syncthreads();
  dense_fwd<T, RB>(bufA, z, V1, c1, h1, bufB, tid);
  __syncthreads();
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) {
    const T v = bufB[idx] > T(0) ? bufB[idx] : T(0);
    bufB[idx] = v;
    if (idx / h1 < rows) D1[(s0 + idx / h1) * h1 + idx % h1] = v;
  }
  __syncthreads();
  dense_fwd<T, RB>(bufB, h1, V2, c2, h0, bufA, tid);
  __syncthreads();
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const T v = bufA[idx] > T(0) ? bufA[idx] : T(0);
    if (idx / h0 < rows) D2[(s0 + idx / h0) * h0 + idx % h0] = v;
  }
}

template <typename T, int RB>
__global__ __launch_bounds__(NTHREADS) void dec_mid_bwd_kernel(VaeDims<T> d, const T *dec_params,
                                                               const T *D1, const T *D2,
                                                               const T *dD2s, int nsD, int64_t slabD,
                                                               T *dZ) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  const int mx = max(d.h0, max(d.h1, d.z));
  T *gA = reinterpret_cast<T *>(smem_raw);
  T *gB = gA + RB * mx;
  const int h0 = d.h0, h1 = d.h1, z = d.z;
  const T *V1 = dec_params, *c1 = V1 + (int64_t)z * h1, *V2 = c1 + h1;
  const int tid = threadIdx.x;
  const int64_t s0 = (int64_t)blockIdx.x * RB;
  const int rows = (int)((d.B - s0 < RB) ? d.B - s0 : RB);
  for (int idx = tid; idx < RB * h0; idx += NTHREADS) {
    const int r = idx / h0, j = idx % h0;
    gA[idx] = (r < rows && D2[(s0 + r) * h0 + j] > T(0)) ? slab_sum_at<T>(dD2s, nsD, slabD, (s0 + r) * h0 + j) : T(0);
  }
  __syncthreads();
  dense_bwd_in<T, RB>(gA, h0, V2, h1, gB, tid);
  __syncthreads();
  for (int idx = tid; idx < RB * h1; idx += NTHREADS) {
    const int r = idx / h1, j = idx % h1;
    gB[idx] = (r < rows && D1[(s0 + r) * h1 + j] > T(0)) ? gB[idx] : T(0);
  }
  __syncthreads();
  dense_bwd_in<T, RB>(gB, h1, V1, z, gA, tid);
  __syncthreads();
  for (int idx = tid; idx < RB * z; idx += NTHREADS)
    if (idx / z < rows) dZ[(s0 + idx / z) * z + idx % z] = gA[idx];
}

// ---- wide hidden layers: the same chain as MFMA tile products + elementwise kernels ------------------
// Used when the mid-layer weights do not fit in LDS beside the activation buffers (e.g. the scaled
// 512-256-64 MLP of SURVEY.md §8d): per-sample dense layers on the vector ALU then cost more than
// the n-sized products around them.
template <typename T>
__global__ void bias_relu_kernel(int64_t count, int w, T *H, const T *__restrict__ bias) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const T v = H[i] + bias[i % w];
    H[i] = v > T(0) ? v : T(0);
  }
}
template <typename T>
__global__ void relu_mask_kernel(int64_t count, T *G, const T *__restrict__ act) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x)
    G[i] = act[i] > T(0) ? G[i] : T(0);
}
// z = mu + exp(ls) eps (vae.py:83,135) and the latent terms of the loss; MU, LS are B x z
template <typename T>
__global__ __launch_bounds__(NTHREADS) void reparam_kernel(int64_t count, const T *__restrict__ MU,
                                                           const T *__restrict__ LS,
                                                           const T *__restrict__ eps, T *SD, T *Z,
                                                           T *lp) {
  __shared__ T red[4];
  T lsum = T(0);
  for (int64_t i = (int64_t)blockIdx.x * NTHREADS + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * NTHREADS) {
    const T e = eps[i], ls = LS[i];
    const T sd = dexp<T>(ls);
    const T zz = MU[i] + sd * e;
    SD[i] = sd;
    Z[i] = zz;
    lsum += T(0.5) * zz * zz - T(0.5) * e * e - ls;
  }
  lsum = wave_sum<T>(lsum);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lsum;
  __syncthreads();
  if (threadIdx.x == 0) lp[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}
// on entry MU holds dZ (decoder part); on exit MU = dmu = dZ + z, LS = dls = dmu eps sd - 1
template <typename T>
__global__ void latent_grad_kernel(int64_t count, T *MU, T *LS, const T *__restrict__ eps,
                                   const T *__restrict__ SD, const T *__restrict__ Z) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const T dmu = MU[i] + Z[i];
    MU[i] = dmu;
    LS[i] = dmu * eps[i] * SD[i] - T(1);
  }
}
static inline unsigned ew_blocks(int64_t count) {
  const int64_t b = cdiv(count, (int64_t)NTHREADS * 4);
  return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

// ---- decoder output layer, forward and backward in one pass (f32, h0 + 1 <= 64) ---------------------
// The three n-sized products of the decoder's last layer share their operands:
//   F   = [D2 | 1] [V3; c3]                     DF = (F - Y) / sigma^2 ,  loss += 0.5 sum (F - Y)^2 / sigma^2
//   dD2 += DF V3^T                              (per 64-column tile: K = the tile's columns)
//   [dV3; dc3] += [D2 | 1]^T DF                 (per 128-row block: K = the block's rows)
// Sums across workgroups are left as partial sums in slabs (no float atomics): row block bx stores its
// [dV3; dc3] tiles in slab bx of dV3s, column group by its dD2 rows in slab by of dD2s and its share of the
// loss in lp[by * gridDim.x + bx]; vae_mid_bwd_kernel / elbo_finish_kernel add the slabs in order.
// As separate launches DF (B x n) is written once and read twice (3 x 67 MB at C3).  Here a workgroup
// keeps its 128 rows of [D2 | 1] in LDS and walks `nt` column tiles: product 1 from LDS, the
// epilogue leaves the DF tile in LDS, and that tile is the A operand of product 2 (row-major image)
// and the B operand of product 3 (k-major image) as it lies; the V3 tile is product 1's k-major B
// image and, read the other way, product 2's row-major B image.  DF never goes to HBM; dD2 stays in
// registers over the walk.  Y is read once, V3 once per row block (L2).
constexpr int FS = 68;  // LDS row stride of the 64-wide images (16-byte aligned rows)

// Eight waves (two per SIMD) since round 3: the kernel is a chain of short MFMA bursts separated by LDS round trips, an
// epilogue and a barrier, and with one wave per SIMD nothing ran while a wave waited (the products on the bf16 pipe
// instead, 2.5x cheaper, gained 8 %: profiles/r03_elbo_b3_experiment.txt).  Product 1 is split over all eight waves
// (32 x 32 tiles of F); after the barrier waves 0-3 run product 3 (one 32 x 32 tile of [dV3; dc3] each, K = 128) and
// waves 4-7 product 2 (32 rows of dD2 each): 96 MFMAs per wave and tile either way.  Wave roles are scalar values
// (readfirstlane): an MFMA does not honour EXEC.  Same sums in the same order as the four-wave form.
constexpr int FUSED_THREADS = 512;
__global__ __launch_bounds__(FUSED_THREADS) void vae_out_fused_kernel(
    int64_t n, int h0, const float *__restrict__ D2, const float *__restrict__ V3c,
    const float *__restrict__ Y, int64_t ldy, float inv_var, float *lp, float *dD2s, int64_t slabD,
    float *dV3s, int64_t slabV, int nt) {
  using M = Mfma<float>;
  constexpr int NT = FUSED_THREADS;
  // 153 KB of LDS, one workgroup per CU.  The V3 image is triple- and the DF image double-buffered, so
  // that ONE barrier per tile (DF tile complete / next V3 tile staged) orders everything and product 1
  // of tile t+1 follows product 2 / 3 of tile t in each wave's MFMA stream:
  //   V3s[(t+1) % 3] is written in tile t; its last readers (product 2 of tile t-2) finished before
  //   the barrier of tile t-1, which every wave has passed; DFs[t & 1] likewise (readers: tile t-2).
  __shared__ __align__(16) float lds[128 * FS + 3 * 64 * FS + 2 * 128 * FS];
  float *D2s = lds;               // [row][k]: D2, a ones column at k = h0, zeros beyond
  float *V3b = D2s + 128 * FS;    // 3 x [k][col]: V3 rows, c3 at k = h0, zeros beyond
  float *DFb = V3b + 3 * 64 * FS; // 2 x [row][col]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wu = __builtin_amdgcn_readfirstlane(tid >> 6);  // 0..7, scalar
  const int64_t row0 = (int64_t)blockIdx.x * 128;
  const int64_t tile0 = (int64_t)blockIdx.y * nt;
  const int64_t ntiles = n / 64;
  const int ntl = (int)((tile0 + nt <= ntiles) ? nt : ntiles - tile0);  // tiles of this workgroup (>= 1)

  for (int idx = tid; idx < 128 * 64; idx += NT) {
    const int r = idx >> 6, k = idx & 63;
    D2s[r * FS + k] = k < h0 ? D2[(row0 + r) * h0 + k] : (k == h0 ? 1.0f : 0.0f);
  }
  // V3 tile: 64 k-rows x 16 vectors, 2 per thread; rows past h0 lie outside the buffer and read 0
  i32x4_t vrs = make_buffer_rsrc(V3c);
  vrs[2] = __builtin_amdgcn_readfirstlane((int)((int64_t)(h0 + 1) * n * 4));
  int vvo[2];
#pragma unroll
  for (int p = 0; p < 2; ++p) {
    const int v = tid + p * NT;
    vvo[p] = (v >> 4) * (int)(n * 4) + (v & 15) * 16;
  }
  i32x4_t vreg[2];
  auto load_v3 = [&](int64_t tile) {
    const int so = (int)(tile * 64 * 4);
#pragma unroll
    for (int p = 0; p < 2; ++p) vreg[p] = raw_buffer_load_b128(vrs, vvo[p], so, 0);
  };
  auto stage_v3 = [&](float *dst) {
#pragma unroll
    for (int p = 0; p < 2; ++p) {
      const int v = tid + p * NT;
      *reinterpret_cast<i32x4_t *>(&dst[(v >> 4) * FS + (v & 15) * 4]) = vreg[p];
    }
  };
  // product 1: wave -> rows 32 (wu >> 1) .., columns 32 (wu & 1) .. of the tile; its data tile is requested a tile
  // ahead and BEFORE the previous tile's gradient stores
  const int rb1 = (wu >> 1) * 32, cb1 = (wu & 1) * 32;
  float yv[M::NACC];
  auto load_y = [&](int64_t tile) {
    TileIO<float> ioy;
    ioy.init(Y + (row0 + rb1) * ldy + tile * 64 + cb1, ldy, lane);
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) yv[r] = ioy.load(0, 0, r);
  };
  load_v3(tile0);
  stage_v3(V3b);
  load_y(tile0);
  if (ntl > 1) load_v3(tile0 + 1);
  __syncthreads();  // D2s and the first V3 tile staged

  typename M::acc_t acc2[1][2];  // waves 4-7: dD2 rows 32 (wu - 4) .. x (h0 padded to 64), over the whole walk
  zero_acc<float, 1, 2>(acc2);
  float lsum = 0.0f;
  const int wm3 = ((wu & 3) >> 1) * 32, wn3 = (wu & 1) * 32;  // waves 0-3, product 3: (h0 index, column) quadrant

  for (int t = 0; t < ntl; ++t) {
    const int64_t col0 = (tile0 + t) * 64;
    const float *V3s = V3b + (t % 3) * 64 * FS;
    float *DFs = DFb + (t & 1) * 128 * FS;
    typename M::acc_t acc1[1][1];
    zero_acc<float, 1, 1>(acc1);
    mma_ktile<float, 1, 1, 64, true, FS, false, FS>(D2s, V3s, rb1, cb1, lane, acc1);
#pragma unroll
    for (int r = 0; r < M::NACC; ++r) {
      const float diff = acc1[0][0][r] - yv[r];
      lsum += diff * diff;
      DFs[(rb1 + M::row(lane, r)) * FS + cb1 + M::col(lane)] = diff * inv_var;
    }
    if (t + 1 < ntl) {  // next tile's operands, ahead of this tile's gradient stores
      stage_v3(V3b + ((t + 1) % 3) * 64 * FS);
      load_y(tile0 + t + 1);
      if (t + 2 < ntl) load_v3(tile0 + t + 2);
    }
    __syncthreads();  // DF tile complete, next V3 tile staged
    if (wu >= 4) {
      mma_ktile<float, 1, 2, 64, true, FS, true, FS>(DFs, V3s, (wu - 4) * 32, 0, lane, acc2);  // dD2 += DF V3^T
    } else {
      typename M::acc_t acc3[1][1];  // [dV3; dc3] tile: K = the 128 rows
      zero_acc<float, 1, 1>(acc3);
      // (two halves of K: all 128 k values' fragments at once do not fit beside the other role's registers)
      mma_ktile<float, 1, 1, 64, false, FS, false, FS>(D2s, DFs, wm3, wn3, lane, acc3);
      mma_ktile<float, 1, 1, 64, false, FS, false, FS>(D2s + 64 * FS, DFs + 64 * FS, wm3, wn3, lane, acc3);
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int hi = wm3 + M::row(lane, r);
        if (hi <= h0) dV3s[(int64_t)blockIdx.x * slabV + (int64_t)hi * n + col0 + wn3 + M::col(lane)] = acc3[0][0][r];
      }
    }
  }
  if (wu >= 4) {
#pragma unroll
    for (int fn = 0; fn < 2; ++fn)
#pragma unroll
      for (int r = 0; r < M::NACC; ++r) {
        const int hi = fn * 32 + M::col(lane);
        if (hi < h0) dD2s[(int64_t)blockIdx.y * slabD + (row0 + (wu - 4) * 32 + M::row(lane, r)) * h0 + hi] = acc2[0][fn][r];
      }
  }
  __shared__ float red[8];
  lsum = wave_sum<float>(lsum);
  if (lane == 0) red[wu] = lsum;
  __syncthreads();
  // (row block, column half) pairs summed first: the four-wave form's per-wave partial sums, in its order
  if (tid == 0)
    lp[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] =
        0.5f * inv_var * (((red[0] + red[1]) + (red[2] + red[3])) + ((red[4] + red[5]) + (red[6] + red[7])));
}

// ---- Adam (jax.experimental.optimizers.adam, as numpyro.optim.Adam wraps it) ------------------------
//   m = (1-b1) g + b1 m ; v = (1-b2) g^2 + b2 v ; p -= lr * (m/(1-b1^t)) / (sqrt(v/(1-b2^t)) + eps)
template <typename T>
__global__ void adam_kernel(int64_t count, T *p, const T *g, T *m, T *v, T lr, T b1, T b2, T eps,
                            T c1, T c2) {
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < count;
       i += (int64_t)gridDim.x * blockDim.x) {
    const T gi = g[i];
    const T mi = (T(1) - b1) * gi + b1 * m[i];
    const T vi = (T(1) - b2) * gi * gi + b2 * v[i];
    m[i] = mi;
    v[i] = vi;
    const T mhat = mi / c1, vhat = vi / c2;
    p[i] = p[i] - lr * mhat / (dsqrt<T>(vhat) + eps);
  }
}

// split K so that a launch has about 512 workgroups (2 per CU): measured best at the C3 shape —
// more, shorter slices pay more prologue and atomics than they gain in memory parallelism.  With more
// than 512 tiles' worth of work per slice the count is rounded to fill whole rounds of 512: 320 tiles
// (the scaled MLP's dV3) would otherwise leave a quarter of the CUs with twice the work of the rest.
static inline int pick_splitk(int64_t tiles, int64_t K, int64_t round = 512) {
  if (tiles < 1) tiles = 1;
  int64_t maxs = cdiv(K, 4 * 32);  // at least 4 (f32) k-tiles per slice
  if (maxs < 1) maxs = 1;
  if (maxs > 32) maxs = 32;
  int best = 1;
  double best_eff = 0;
  for (int s = 1; s <= maxs; ++s) {
    const int64_t wgs = tiles * s;
    const double eff = (double)wgs / (double)(cdiv(wgs, round) * round);
    if (eff > best_eff + 0.05) { best_eff = eff; best = s; }  // the smallest s within 5 % of the best
    if (wgs >= 4 * round) break;
  }
  return best;
}

// C (M x N) = mask(relu([C +] A (M x K) W (K x N) [+ bias]))       (a dense layer forwards)
template <typename T>
static void dense_gemm(hipStream_t st, const T *A, int64_t M, int K, const T *W, int N, const T *bias,
                       int relu, T *C) {
  GemmArgs<T> g{};
  g.A = A; g.lda = K; g.B = W; g.ldb = N; g.C = C; g.ldc = N;
  g.M = M; g.N = N; g.K = K; g.bias = bias; g.relu = relu;
  g.vec_a = kc_vec_ok<T>(A, K); g.vec_b = mc_vec_ok<T>(W, N);
  launch_gemm<T, true, false, EPI_FUSED>(st, g, 1);
}
// C (M x K) = mask([C +] G (M x N) W^T)        (input gradient of a dense layer; W is K x N)
template <typename T>
static void dense_gemm_t(hipStream_t st, const T *G, int64_t M, int N, const T *W, int K, T *C,
                         int addc, const T *mask) {
  GemmArgs<T> g{};
  g.A = G; g.lda = N; g.B = W; g.ldb = N; g.C = C; g.ldc = K;
  g.M = M; g.N = K; g.K = N; g.addc = addc; g.mask = mask; g.ldm = K;
  g.vec_a = kc_vec_ok<T>(G, N); g.vec_b = kc_vec_ok<T>(W, N);
  launch_gemm<T, true, true, EPI_FUSED>(st, g, 1);
}
// [dW; db] = [A | 1]^T G      ((K+1) x N, summed over the M samples; W and b adjacent in the layout):
// k-slices into `slabs`, then summed in slice order into dW
static inline int gemm_w_splits(int K, int N, int64_t M) { return pick_splitk(cdiv(K + 1, GBM) * cdiv(N, GBN), M); }
template <typename T>
static void dense_gemm_w(hipStream_t st, const T *A, int64_t M, int K, const T *G, int N, T *dW, T *slabs) {
  GemmArgs<T> g{};
  g.A = A; g.lda = K; g.B = G; g.ldb = N; g.C = slabs; g.ldc = N;
  g.M = K + 1; g.N = N; g.K = M; g.a_ones_row = K;
  g.slab = (int64_t)(K + 1) * N;
  g.vec_a = mc_vec_ok<T>(A, K); g.vec_b = mc_vec_ok<T>(G, N);
  const int ns = launch_gemm<T, false, false, EPI_ATOMIC>(st, g, gemm_w_splits(K, N, M));
  launch_slab_sum<T>(st, dW, slabs, g.slab, ns, g.slab);
}

// samples per workgroup of the mid kernels: the backward kernel keeps four [RB][mx] buffers in LDS and a
// launch may ask for 64 KB of it; h0, h1, z <= 1024 (check_dims) always leaves RB >= 2
// Among the sizes that fit, the largest that still gives two workgroups per CU (the kernels are latency-bound:
// at C3, B = 4096, the ELBO takes 200 / 192 / 195 us with 16 / 8 / 4 samples per workgroup), else 4.
template <typename T> static inline int pick_rb(int mx, int nbuf, int64_t rows) {
  int fit = 0;
  for (int rb : {16, 8, 4, 2})
    if ((size_t)nbuf * rb * mx * sizeof(T) <= 64 * 1024) {
      if (fit == 0) fit = rb;
      if (rb <= 4 || cdiv(rows, (int64_t)rb) >= 512) return rb;
    }
  return fit;
}
#define GPVAE_RB_SWITCH(rb, ...)                                   \
  switch (rb) {                                                    \
    case 16: { constexpr int RB = 16; __VA_ARGS__; } break;        \
    case 8:  { constexpr int RB = 8;  __VA_ARGS__; } break;        \
    case 4:  { constexpr int RB = 4;  __VA_ARGS__; } break;        \
    default: { constexpr int RB = 2;  __VA_ARGS__; } break;        \
  }

// Shapes of one ELBO evaluation: which path runs, how every cross-workgroup sum is split, and how much slab
// space that takes.  gpvae_vae_workspace_bytes and run_vae_elbo both read it, so they cannot disagree.
template <typename T> struct ElboPlan {
  int rb;
  bool wide;
  int64_t nblk, mc;
  bool fused_dims;   // the shapes allow vae_out_fused_kernel (it also needs gradients and a 2 GiB-safe ldy)
  int nt;            // column tiles per workgroup of the fused kernel
  int64_t fgx, fgy;  // its grid: row blocks x column groups
  int sA1, sW1, sV3, sD2;  // k-slices of the four n-sized split products (sV3, sD2: the unfused path)
  int64_t nsV3, nsD2;      // slabs reserved = the larger of the fused and the unfused need
  int64_t nlp_mid, nlp;    // loss parts: mid / reparam kernel, then the output layer's workgroups
  int64_t tmp_elems;       // wide path: slabs of one mid-layer weight-gradient product at a time
  int64_t rep_elems;       // small path: one copy of the mid-layer gradients per workgroup
  ElboPlan(int64_t B, int64_t n, int h0, int h1, int z) {
    const int mx = std::max(h0, std::max(h1, z));
    mc = VaeParams<T>::mid_count(h0, h1, z);
    const size_t mid_bytes = (size_t)mc * sizeof(T);
    rb = pick_rb<T>(mx, 4, B);
    wide = rb == 0 || mid_bytes + 4 * (size_t)rb * mx * sizeof(T) > 60 * 1024;
    nblk = cdiv(B, rb > 0 ? rb : 1);
    fused_dims = sizeof(T) == 4 && h0 + 1 <= 64 && B % 128 == 0 && n % 64 == 0 &&
                 (int64_t)(h0 + 1) * n * 4 < ((int64_t)1 << 31);
    const int64_t rb128 = B / 128, ntiles = n / 64;
    nt = (int)(rb128 * ntiles / 256);  // about 256 workgroups: one per CU (153 KB of LDS each) in one round
    if (nt < 1) nt = 1;
    fgx = rb128;
    fgy = cdiv(ntiles, nt);
    sA1 = gemm_slices<T>(n, pick_splitk(cdiv(B, 64) * cdiv(h0, GBN), n, 1024));
    sW1 = gemm_slices<T>(B, pick_splitk(cdiv(n + 1, 64) * cdiv(h0, GBN), B, 1024));
    sV3 = gemm_slices<T>(B, h0 + 1 <= 64 ? pick_splitk(cdiv(n, GBN), B)
                                          : pick_splitk(cdiv(h0 + 1, GBM) * cdiv(n, GBN), B));
    sD2 = gemm_slices<T>(n, pick_splitk(cdiv(B, 64) * cdiv(h0, GBN), n, 1024));
    nsV3 = std::max<int64_t>(sV3, fused_dims ? fgx : 0);
    nsD2 = std::max<int64_t>(sD2, fused_dims ? fgy : 0);
    nlp_mid = wide ? 2048 : nblk;
    nlp = nlp_mid + std::max<int64_t>(cdiv(B, GBM) * cdiv(n, GBN), fused_dims ? fgx * fgy : 0);
    tmp_elems = 0;
    if (wide) {
      const int kk[5] = {h1, z, h1, h1, h0}, nn[5] = {h0, h1, z, z, h1};
      for (int i = 0; i < 5; ++i)
        tmp_elems = std::max<int64_t>(tmp_elems, (int64_t)gemm_slices<T>(B, gemm_w_splits(kk[i], nn[i], B)) * (kk[i] + 1) * nn[i]);
    }
    rep_elems = wide ? 0 : nblk * mc;
  }
  int64_t slab_elems(int64_t B, int64_t n, int h0) const {
    return sA1 * B * h0 + nsD2 * B * h0 + nsV3 * (int64_t)(h0 + 1) * n + sW1 * (n + 1) * h0 + rep_elems + tmp_elems +
           nlp + 64;
  }
};

// workspace carve-up (elements of T)
template <typename T> struct VaeWork {
  T *H1, *H2, *SD, *Z, *D1, *D2, *dD2, *dA1, *MU, *LS, *DF;
  T *SA1, *SD2, *SV3, *SW1, *REP, *TMP, *LP;  // partial sums (ElboPlan): nothing here needs clearing
  static int64_t elems(int64_t B, int64_t n, int h0, int h1, int z) {
    const ElboPlan<T> p(B, n, h0, h1, z);
    return B * (4 * (int64_t)h0 + 2 * (int64_t)h1 + 4 * (int64_t)z) + p.slab_elems(B, n, h0) + B * n + 64;
  }
  __host__ VaeWork(T *w, const ElboPlan<T> &p, int64_t B, int64_t n, int h0, int h1, int z) {
    H1 = w;            dD2 = H1 + B * h0;
    H2 = dD2 + B * h0; SD = H2 + B * h1;   Z = SD + B * z;
    D1 = Z + B * z;    D2 = D1 + B * h1;   dA1 = D2 + B * h0;
    MU = dA1 + B * h0; LS = MU + B * z;  // wide-layer path only
    SA1 = LS + B * z;  SD2 = SA1 + p.sA1 * B * h0;  SV3 = SD2 + p.nsD2 * B * h0;
    SW1 = SV3 + p.nsV3 * (int64_t)(h0 + 1) * n;     REP = SW1 + p.sW1 * (n + 1) * h0;
    TMP = REP + p.rep_elems;  LP = TMP + p.tmp_elems;
    DF = LP + p.nlp;
    DF = (T *)(((uintptr_t)DF + 15) & ~(uintptr_t)15);
  }
};

template <typename T>
static int run_vae_elbo(hipStream_t st, int64_t B, int64_t n, int h0, int h1, int z,
                        const void *params_v, const void *Y_v, int64_t ldy, const void *eps_v,
                        double obs_sigma, void *loss_v, void *grads_v, void *work_v) {
  const T *params = (const T *)params_v, *Y = (const T *)Y_v, *eps = (const T *)eps_v;
  T *loss = (T *)loss_v, *grads = (T *)grads_v;
  const ElboPlan<T> pl(B, n, h0, h1, z);
  VaeWork<T> W((T *)work_v, pl, B, n, h0, h1, z);
  VaeParams<T> P(params, n, h0, h1, z);
  VaeDims<T> dims{B, n, h0, h1, z};
  const int mx = std::max(h0, std::max(h1, z));
  // Small hidden layers (the reference's 35-30-10 / 50-30-15): the whole chain between the two n-sized
  // layers runs per block of samples with its weights staged in LDS.  When weights + activation
  // buffers do not fit the 64 KB a launch may ask for, every layer is a tile product instead.
  // Every sum across workgroups (split K, per-sample-block weight gradients, the loss) is left as partial
  // sums in slabs of the workspace and added in a fixed order by its consumer or by elbo_finish_kernel:
  // no floating-point atomics, the same inputs give the same bits, and nothing needs clearing first.
  const int rb = pl.rb;
  const bool wide = pl.wide;
  const size_t mid_bytes = (size_t)pl.mc * sizeof(T);
  const size_t shm_f = 3 * (size_t)rb * mx * sizeof(T) + mid_bytes;
  const size_t shm_b = 4 * (size_t)rb * mx * sizeof(T) + mid_bytes;
  const unsigned nblk = (unsigned)pl.nblk;
  const int64_t Bz = B * z;
  const T inv_var = (T)(1.0 / (obs_sigma * obs_sigma));
  // constant of the data term: B*n*(log sigma + 0.5 log 2pi)
  const T loss_const = (T)((double)B * (double)n * (log(obs_sigma) + 0.9189385332046727));
  const int64_t slabH = B * h0;  // slab of a (B x h0) partial sum
  int64_t nlp = 0;               // loss parts written so far

  int nsA;
  {  // A1 = Y W1, as k-slices
    GemmArgs<T> g{};
    g.A = Y; g.lda = ldy; g.B = P.W1; g.ldb = h0; g.C = W.SA1; g.ldc = h0; g.slab = slabH;
    g.M = B; g.N = h0; g.K = n; g.a_ones_row = -1;
    g.vec_a = kc_vec_ok<T>(Y, ldy); g.vec_b = mc_vec_ok<T>(P.W1, h0);
    nsA = launch_gemm<T, true, false, EPI_ATOMIC, 64>(st, g, pick_splitk(cdiv(B, 64) * cdiv(h0, GBN), n, 1024));
  }
  if (!wide) {
    GPVAE_RB_SWITCH(rb, hipLaunchKernelGGL((vae_mid_fwd_kernel<T, RB>), dim3(nblk), dim3(NTHREADS), shm_f,
                                           st, dims, params, eps, (const T *)W.SA1, nsA, slabH, W.H1, W.H2,
                                           W.SD, W.Z, W.D1, W.D2, W.LP));
    nlp = nblk;
  } else {
    launch_slab_sum<T>(st, W.H1, W.SA1, slabH, nsA, slabH);
    hipLaunchKernelGGL((bias_relu_kernel<T>), dim3(ew_blocks(B * h0)), dim3(NTHREADS), 0, st, B * h0, h0,
                       W.H1, P.b1);
    dense_gemm<T>(st, W.H1, B, h0, P.W2, h1, P.b2, 1, W.H2);
    dense_gemm<T>(st, W.H2, B, h1, P.Wmu, z, P.bmu, 0, W.MU);
    dense_gemm<T>(st, W.H2, B, h1, P.Wsd, z, P.bsd, 0, W.LS);
    const unsigned rblocks = ew_blocks(Bz);
    hipLaunchKernelGGL((reparam_kernel<T>), dim3(rblocks), dim3(NTHREADS), 0, st, Bz, W.MU, W.LS, eps,
                       W.SD, W.Z, W.LP);
    nlp = rblocks;
    dense_gemm<T>(st, W.Z, B, z, P.V1, h1, P.c1, 1, W.D1);
    dense_gemm<T>(st, W.D1, B, h1, P.V2, h0, P.c2, 1, W.D2);
  }
  // f32 with gradients, whole tiles: the output layer's three n-sized products in one pass
  const int64_t slabV = (int64_t)(h0 + 1) * n;
  int nsV = 0, nsD = 0;
  bool fused = false;
  if constexpr (sizeof(T) == 4) fused = grads != nullptr && pl.fused_dims && 32 * ldy * 4 < ((int64_t)1 << 31);
  if (fused) {
    if constexpr (sizeof(T) == 4) {
      hipLaunchKernelGGL(vae_out_fused_kernel, dim3((unsigned)pl.fgx, (unsigned)pl.fgy), dim3(FUSED_THREADS), 0, st, n,
                         h0, W.D2, P.V3, Y, ldy, inv_var, W.LP + nlp, W.SD2, slabH, W.SV3, slabV, pl.nt);
      nlp += pl.fgx * pl.fgy;
      nsV = (int)pl.fgx;
      nsD = (int)pl.fgy;
    }
  } else {  // DF = (D2 V3 + c3 - Y)/sigma^2 ; loss parts 0.5 sum diff^2 / sigma^2
    GemmArgs<T> g{};
    g.A = W.D2; g.lda = h0; g.B = P.V3; g.ldb = n; g.C = W.DF; g.ldc = n;
    g.M = B; g.N = n; g.K = h0; g.a_ones_row = -1;
    g.bias = P.c3; g.Yref = Y; g.ldy = ldy; g.inv_var = inv_var; g.loss = loss; g.loss_part = W.LP + nlp;
    g.vec_a = kc_vec_ok<T>(W.D2, h0); g.vec_b = mc_vec_ok<T>(P.V3, n);
    launch_gemm<T, true, false, EPI_DF>(st, g, 1);
    nlp += cdiv(B, GBM) * cdiv(n, GBN);
  }
  int nsW = 0;
  if (grads != nullptr) {
    VaeParams<T> G(grads, n, h0, h1, z);
    if (!fused) {
      {  // [dV3; dc3] = [D2 | 1]^T DF        ((h0+1) x n; V3 and c3 are adjacent in the layout)
        GemmArgs<T> g{};
        g.A = W.D2; g.lda = h0; g.B = W.DF; g.ldb = n; g.C = W.SV3; g.ldc = n; g.slab = slabV;
        g.M = h0 + 1; g.N = n; g.K = B; g.a_ones_row = h0;
        g.vec_a = mc_vec_ok<T>(W.D2, h0); g.vec_b = mc_vec_ok<T>(W.DF, n);
        if (h0 + 1 <= 64)
          nsV = launch_gemm<T, false, false, EPI_ATOMIC, 64>(st, g, pick_splitk(cdiv(n, GBN), B));
        else
          nsV = launch_gemm<T, false, false, EPI_ATOMIC>(st, g, pick_splitk(cdiv(h0 + 1, GBM) * cdiv(n, GBN), B));
      }
      {  // dD2 = DF V3^T
        GemmArgs<T> g{};
        g.A = W.DF; g.lda = n; g.B = P.V3; g.ldb = n; g.C = W.SD2; g.ldc = h0; g.slab = slabH;
        g.M = B; g.N = h0; g.K = n; g.a_ones_row = -1;
        g.vec_a = kc_vec_ok<T>(W.DF, n); g.vec_b = kc_vec_ok<T>(P.V3, n);
        nsD = launch_gemm<T, true, true, EPI_ATOMIC, 64>(st, g, pick_splitk(cdiv(B, 64) * cdiv(h0, GBN), n, 1024));
      }
    }
    if (nsV > pl.nsV3 || nsD > pl.nsD2 || nsA > pl.sA1 || nlp > pl.nlp) {
      set_error("gpvae_vae_elbo: internal: slab plan exceeded");
      return -14;
    }
    if (!wide) {
      GPVAE_RB_SWITCH(rb, hipLaunchKernelGGL((vae_mid_bwd_kernel<T, RB>), dim3(nblk), dim3(NTHREADS),
                                             shm_b, st, dims, params, grads, eps, W.H1, W.H2, W.SD, W.Z,
                                             W.D1, W.D2, (const T *)W.SD2, nsD, slabH, W.dA1, W.REP));
    } else {
      // gradient buffers reuse the activations they mask: dG2 in dD2, dG1 then dA2 in D1, dmu | dls in MU | LS
      launch_slab_sum<T>(st, W.dD2, W.SD2, slabH, nsD, slabH);
      hipLaunchKernelGGL((relu_mask_kernel<T>), dim3(ew_blocks(B * h0)), dim3(NTHREADS), 0, st, B * h0,
                         W.dD2, W.D2);                                      // dG2
      dense_gemm_w<T>(st, W.D1, B, h1, W.dD2, h0, (T *)G.V2, W.TMP);       // dV2 | dc2
      dense_gemm_t<T>(st, W.dD2, B, h0, P.V2, h1, W.D1, 0, W.D1);          // dG1 = (dG2 V2^T) (D1 > 0)
      dense_gemm_w<T>(st, W.Z, B, z, W.D1, h1, (T *)G.V1, W.TMP);          // dV1 | dc1
      dense_gemm_t<T>(st, W.D1, B, h1, P.V1, z, W.MU, 0, (const T *)nullptr);  // dZ (decoder part)
      hipLaunchKernelGGL((latent_grad_kernel<T>), dim3(ew_blocks(Bz)), dim3(NTHREADS), 0, st, Bz, W.MU, W.LS,
                         eps, W.SD, W.Z);
      dense_gemm_w<T>(st, W.H2, B, h1, W.MU, z, (T *)G.Wmu, W.TMP);        // dWmu | dbmu
      dense_gemm_w<T>(st, W.H2, B, h1, W.LS, z, (T *)G.Wsd, W.TMP);        // dWsd | dbsd
      dense_gemm_t<T>(st, W.MU, B, z, P.Wmu, h1, W.D1, 0, (const T *)nullptr);
      dense_gemm_t<T>(st, W.LS, B, z, P.Wsd, h1, W.D1, 1, W.H2);           // dA2 = (dmu Wmu^T + dls Wsd^T) (H2 > 0)
      dense_gemm_w<T>(st, W.H1, B, h0, W.D1, h1, (T *)G.W2, W.TMP);        // dW2 | db2
      dense_gemm_t<T>(st, W.D1, B, h1, P.W2, h0, W.dA1, 0, W.H1);          // dA1 = (dA2 W2^T) (H1 > 0)
    }
    {  // [dW1; db1] = [Y | 1]^T dA1        ((n+1) x h0; W1 and b1 are adjacent in the layout)
      GemmArgs<T> g{};
      g.A = Y; g.lda = ldy; g.B = W.dA1; g.ldb = h0; g.C = W.SW1; g.ldc = h0; g.slab = (n + 1) * h0;
      g.M = n + 1; g.N = h0; g.K = B; g.a_ones_row = n;
      g.vec_a = mc_vec_ok<T>(Y, ldy); g.vec_b = mc_vec_ok<T>(W.dA1, h0);
      nsW = launch_gemm<T, false, false, EPI_ATOMIC, 64>(st, g, pick_splitk(cdiv(n + 1, 64) * cdiv(h0, GBN), B, 1024));
      if (nsW > pl.sW1) {
        set_error("gpvae_vae_elbo: internal: slab plan exceeded");
        return -14;
      }
    }
  }
  {  // gradients and loss from their parts, in order
    const int64_t nW1 = (n + 1) * h0;
    int64_t blocks_a = grads != nullptr ? cdiv(nW1 + slabV, 256 * 2) : 1;
    if (blocks_a < 1) blocks_a = 1;
    if (blocks_a > 4096) blocks_a = 4096;
    const int64_t nrep = (grads != nullptr && !wide) ? (int64_t)nblk : 0;
    const int64_t blocks_b = nrep > 0 ? cdiv(pl.mc - h0, 16) : 0;
    hipLaunchKernelGGL((elbo_finish_kernel<T>), dim3((unsigned)(blocks_a + blocks_b)), dim3(256), 0, st, grads, nW1,
                       (const T *)W.SW1, nsW, nW1, h0, pl.mc, (const T *)W.REP, nrep, slabV,
                       (const T *)W.SV3, nsV, slabV, loss, (const T *)W.LP, nlp, loss_const, (int)blocks_a);
  }
  GPVAE_CHECK_LAUNCH("gpvae_vae_elbo");
  return 0;
}

static inline int decoder_vjp_splits(int64_t C, int64_t n, int h0) { return pick_splitk(cdiv(C, GBM) * cdiv(h0, GBN), n); }

template <typename T>
static int run_decoder(hipStream_t st, bool vjp, int64_t C, int64_t n, int h0, int h1, int z,
                       const void *dec_v, const void *Z_v, const void *dF_v, void *out_v,
                       void *work_v) {
  const T *dec = (const T *)dec_v;
  const T *V3 = dec + (int64_t)z * h1 + h1 + (int64_t)h1 * h0 + h0, *c3 = V3 + (int64_t)h0 * n;
  T *D1 = (T *)work_v, *D2 = D1 + C * h1, *dD2 = D2 + C * h0, *SD = dD2 + C * h0;  // SD: k-slices of dF V3^T
  VaeDims<T> dims{C, n, h0, h1, z};
  const int mx = std::max(h0, std::max(h1, z));
  // as in run_vae_elbo: per-sample-block kernels for the reference's small heads, tile products when wide
  const int rb = pick_rb<T>(mx, 2, C);
  const bool wide = rb == 0 || (int64_t)z * h1 + (int64_t)h1 * h0 > 8192;
  const size_t shm = 2 * (size_t)rb * mx * sizeof(T);
  const unsigned nblk = (unsigned)cdiv(C, rb > 0 ? rb : 1);
  const T *V1 = dec, *c1 = V1 + (int64_t)z * h1, *V2 = c1 + h1, *c2 = V2 + (int64_t)h1 * h0;
  if (!wide) {
    GPVAE_RB_SWITCH(rb, hipLaunchKernelGGL((dec_mid_fwd_kernel<T, RB>), dim3(nblk), dim3(NTHREADS), shm, st,
                                           dims, dec, (const T *)Z_v, D1, D2));
  } else {
    dense_gemm<T>(st, (const T *)Z_v, C, z, V1, h1, c1, 1, D1);
    dense_gemm<T>(st, D1, C, h1, V2, h0, c2, 1, D2);
  }
  if (!vjp) {  // F = D2 V3 + c3
    GemmArgs<T> g{};
    g.A = D2; g.lda = h0; g.B = V3; g.ldb = n; g.C = (T *)out_v; g.ldc = n;
    g.M = C; g.N = n; g.K = h0; g.a_ones_row = -1; g.bias = c3;
    g.vec_a = kc_vec_ok<T>(D2, h0); g.vec_b = mc_vec_ok<T>(V3, n);
    launch_gemm<T, true, false, EPI_BIAS>(st, g, 1);
  } else {  // dD2 = dF V3^T, then back through the head
    GemmArgs<T> g{};
    // k-slices stored side by side and added in slice order by the consumer: no float atomics, no clearing
    g.A = (const T *)dF_v; g.lda = n; g.B = V3; g.ldb = n; g.C = SD; g.ldc = h0; g.slab = C * h0;
    g.M = C; g.N = h0; g.K = n; g.a_ones_row = -1;
    g.vec_a = kc_vec_ok<T>(dF_v, n); g.vec_b = kc_vec_ok<T>(V3, n);
    const int ns = launch_gemm<T, true, true, EPI_ATOMIC>(st, g, decoder_vjp_splits(C, n, h0));
    if (!wide) {
      GPVAE_RB_SWITCH(rb, hipLaunchKernelGGL((dec_mid_bwd_kernel<T, RB>), dim3(nblk), dim3(NTHREADS), shm,
                                             st, dims, dec, D1, D2, (const T *)SD, ns, C * h0, (T *)out_v));
    } else {
      launch_slab_sum<T>(st, dD2, SD, C * h0, ns, C * h0);
      hipLaunchKernelGGL((relu_mask_kernel<T>), dim3(ew_blocks(C * h0)), dim3(NTHREADS), 0, st, C * h0, dD2, D2);
      dense_gemm_t<T>(st, dD2, C, h0, V2, h1, D1, 0, D1);  // (dG2 V2^T) (D1 > 0), over D1
      dense_gemm_t<T>(st, D1, C, h1, V1, z, (T *)out_v, 0, (const T *)nullptr);
    }
  }
  GPVAE_CHECK_LAUNCH(vjp ? "gpvae_decoder_vjp" : "gpvae_decoder_fwd");
  return 0;
}


// ---- a11 in one launch: potential of VAEInference.regression and its gradient ---------------------------
// (inference.py:243-254 in NumPyro's unconstrained space: q = (z [zd], u = log noise))
//   U = -[ log N(y_obs; f[obs], sigma) + log N(z; 0, I) + log HalfNormal(sigma; s0) + u ],  sigma = e^u
// The likelihood touches the decoder's output only at the n_obs observed locations, so the output layer
// is evaluated (and differentiated) on the GATHERED columns V3[:, obs], c3[obs] — n_obs * h0 work per
// chain instead of n * h0 — and everything of a block of RB chains stays in LDS: one launch instead of
// decoder forward + ~12 elementwise / reduction / scatter launches + decoder VJP.
template <typename T, int RB>
__global__ __launch_bounds__(NTHREADS) void vae_potential_kernel(
    int64_t chains, int64_t n, int h0, int h1, int z, int n_obs, const T *dec_params, const T *q,
    int64_t ldq, const T *y_obs, const int64_t *obs_idx, T s0p, T *U, T *dU, int64_t lddu) {
  extern __shared__ __align__(16) unsigned char smem_raw[];
  T *w = reinterpret_cast<T *>(smem_raw);
  const int nmid = z * h1 + h1 + h1 * h0 + h0;
  T *V1 = w, *c1 = V1 + z * h1, *V2 = c1 + h1, *c2 = V2 + h1 * h0;
  T *V3g = w + nmid;                 // [h0][n_obs]
  T *c3g = V3g + h0 * n_obs;         // [n_obs]
  T *yo = c3g + n_obs;               // [n_obs]
  T *Zb = yo + n_obs;                // [RB][z]
  T *D1 = Zb + RB * z;               // [RB][h1]
  T *D2 = D1 + RB * h1;              // [RB][h0]
  T *R = D2 + RB * h0;               // [RB][n_obs]   residual, then dF at the observed columns
  T *G2 = R + RB * n_obs;            // [RB][h0]
  T *G1 = G2 + RB * h0;              // [RB][h1]
  T *dz = G1 + RB * h1;              // [RB][z]
  T *su = dz + RB * z;               // [RB] u, then 1/sigma^2
  const int tid = threadIdx.x;
  const int64_t c0 = (int64_t)blockIdx.x * RB;
  const int rows = (int)((chains - c0 < RB) ? chains - c0 : RB);
  const T *V3 = dec_params + nmid, *c3 = V3 + (int64_t)h0 * n;
  for (int i = tid; i < nmid; i += NTHREADS) w[i] = dec_params[i];
  // gather of the observed columns (a column named twice is gathered twice: f[obs_idx] counts it twice, as the
  // reference's does).  An index outside [0, n) is clamped — memory safety only, the host wrappers reject it.
  auto col = [&](int j) -> int64_t {
    const int64_t c = obs_idx[j];
    return c < 0 ? 0 : (c >= n ? (int64_t)n - 1 : c);
  };
  for (int i = tid; i < h0 * n_obs; i += NTHREADS) V3g[i] = V3[(int64_t)(i / n_obs) * n + col(i % n_obs)];
  for (int i = tid; i < n_obs; i += NTHREADS) {
    c3g[i] = c3[col(i)];
    yo[i] = y_obs[i];
  }
  for (int i = tid; i < RB * z; i += NTHREADS) Zb[i] = (i / z < rows) ? q[(c0 + i / z) * ldq + i % z] : T(0);
  if (tid < RB) su[tid] = (tid < rows) ? q[(c0 + tid) * ldq + z] : T(0);
  __syncthreads();
  dense_fwd<T, RB>(Zb, z, V1, c1, h1, D1, tid);
  __syncthreads();
  for (int i = tid; i < RB * h1; i += NTHREADS) D1[i] = D1[i] > T(0) ? D1[i] : T(0);
  __syncthreads();
  dense_fwd<T, RB>(D1, h1, V2, c2, h0, D2, tid);
  __syncthreads();
  for (int i = tid; i < RB * h0; i += NTHREADS) D2[i] = D2[i] > T(0) ? D2[i] : T(0);
  __syncthreads();
  dense_fwd<T, RB>(D2, h0, V3g, c3g, n_obs, R, tid);  // f at the observed locations
  __syncthreads();
  for (int i = tid; i < RB * n_obs; i += NTHREADS) R[i] -= yo[i % n_obs];
  __syncthreads();
  if (tid < rows) {  // one lane per chain: the scalar part of the potential
    const int r = tid;
    const T u = su[r];
    T sq = T(0), zz = T(0);
    for (int j = 0; j < n_obs; ++j) sq += R[r * n_obs + j] * R[r * n_obs + j];
    for (int j = 0; j < z; ++j) zz += Zb[r * z + j] * Zb[r * z + j];
    T du, iv;
    U[c0 + r] = vae_potential_scalar<T>(sq, zz, u, s0p, n_obs, z, du, iv);
    dU[(c0 + r) * lddu + z] = du;
    su[r] = iv;
  }
  __syncthreads();
  for (int i = tid; i < RB * n_obs; i += NTHREADS) R[i] *= su[i / n_obs];  // dF[obs] = r / sigma^2
  __syncthreads();
  dense_bwd_in<T, RB>(R, n_obs, V3g, h0, G2, tid);
  __syncthreads();
  for (int i = tid; i < RB * h0; i += NTHREADS) G2[i] = D2[i] > T(0) ? G2[i] : T(0);
  __syncthreads();
  dense_bwd_in<T, RB>(G2, h0, V2, h1, G1, tid);
  __syncthreads();
  for (int i = tid; i < RB * h1; i += NTHREADS) G1[i] = D1[i] > T(0) ? G1[i] : T(0);
  __syncthreads();
  dense_bwd_in<T, RB>(G1, h1, V1, z, dz, tid);
  __syncthreads();
  for (int i = tid; i < RB * z; i += NTHREADS)
    if (i / z < rows) dU[(c0 + i / z) * lddu + i % z] = Zb[i] + dz[i];
}

static inline int64_t vae_potential_lds(int64_t es, int rb, int h0, int h1, int z, int64_t n_obs) {
  return es * ((int64_t)z * h1 + h1 + (int64_t)h1 * h0 + h0 + (int64_t)h0 * n_obs + 2 * n_obs +
               (int64_t)rb * (2 * z + 2 * h1 + 2 * h0 + n_obs + 1));
}

// sd = exp(log sd) in place; Z = MU + sd * eps when eps and Z are given (the guide's sample, vae.py:135)
template <typename T>
__global__ void exp_reparam_kernel(int64_t count, const T *__restrict__ MU, T *SD,
                                   const T *__restrict__ eps, T *Z) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count) return;
  const T sd = dexp<T>(SD[i]);
  SD[i] = sd;
  if (Z != nullptr) Z[i] = MU[i] + sd * eps[i];
}

// encoder alone (vae_encoder's apply, vae.py:72-85; the guide's z, vae.py:134-135): four dense layers
// as tile products with fused bias / relu epilogues + one elementwise kernel
template <typename T>
static int run_encoder(hipStream_t st, int64_t B, int64_t n, int h0, int h1, int z,
                       const void *params_v, const void *Y_v, int64_t ldy, const void *eps_v,
                       void *MU_v, void *SD_v, void *Z_v, void *work_v) {
  VaeParams<T> P((const T *)params_v, n, h0, h1, z);
  T *H1 = (T *)work_v, *H2 = H1 + B * h0;
  {
    GemmArgs<T> g{};
    g.A = (const T *)Y_v; g.lda = ldy; g.B = P.W1; g.ldb = h0; g.C = H1; g.ldc = h0;
    g.M = B; g.N = h0; g.K = n; g.bias = P.b1; g.relu = 1;
    g.vec_a = kc_vec_ok<T>(Y_v, ldy); g.vec_b = mc_vec_ok<T>(P.W1, h0);
    launch_gemm<T, true, false, EPI_FUSED>(st, g, 1);
  }
  dense_gemm<T>(st, H1, B, h0, P.W2, h1, P.b2, 1, H2);
  dense_gemm<T>(st, H2, B, h1, P.Wmu, z, P.bmu, 0, (T *)MU_v);
  dense_gemm<T>(st, H2, B, h1, P.Wsd, z, P.bsd, 0, (T *)SD_v);
  const int64_t Bz = B * z;
  hipLaunchKernelGGL((exp_reparam_kernel<T>), dim3((unsigned)cdiv(Bz, 256)), dim3(256), 0, st, Bz,
                     (const T *)MU_v, (T *)SD_v, (const T *)eps_v, (T *)Z_v);
  GPVAE_CHECK_LAUNCH("gpvae_encoder_fwd");
  return 0;
}

}  // namespace gpvae

using namespace gpvae;

static int check_dims(const char *fn, int dtype, int64_t B, int64_t n, int h0, int h1, int z) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "%s: bad dtype %d", fn, dtype);
  GPVAE_REQUIRE(B >= 0, 3, "%s: batch < 0", fn);
  GPVAE_REQUIRE(n >= 1, 4, "%s: n < 1", fn);
  GPVAE_REQUIRE(h0 >= 1 && h0 <= 1024, 5, "%s: h0 out of range [1,1024]", fn);
  GPVAE_REQUIRE(h1 >= 1 && h1 <= 1024, 6, "%s: h1 out of range [1,1024]", fn);
  GPVAE_REQUIRE(z >= 1 && z <= 1024, 7, "%s: z out of range [1,1024]", fn);
  return 0;
}

extern "C" int64_t gpvae_vae_param_count(int64_t n, int h0, int h1, int z) {
  return vae_param_count(n, h0, h1, z);
}

extern "C" int64_t gpvae_vae_workspace_bytes(int dtype, int64_t B, int64_t n, int h0, int h1,
                                             int z) {
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  return (dtype == GPVAE_F64 ? VaeWork<double>::elems(B, n, h0, h1, z) : VaeWork<float>::elems(B, n, h0, h1, z)) * es + 256;
}

extern "C" int gpvae_vae_elbo(void *stream, int dtype, int64_t B, int64_t n, int h0, int h1, int z,
                              const void *params, const void *Y, int64_t ldy, const void *eps_z,
                              double obs_sigma, void *loss, void *grads, void *work) {
  if (int rc = check_dims("gpvae_vae_elbo", dtype, B, n, h0, h1, z)) return rc;
  GPVAE_REQUIRE(params, 8, "gpvae_vae_elbo: params is NULL");
  GPVAE_REQUIRE(loss, 13, "gpvae_vae_elbo: loss is NULL");
  GPVAE_REQUIRE(obs_sigma > 0, 12, "gpvae_vae_elbo: obs_sigma must be > 0");
  hipStream_t st = (hipStream_t)stream;
  if (B == 0) {
    (void)hipMemsetAsync(loss, 0, dtype == GPVAE_F64 ? 8 : 4, st);
    if (grads) (void)hipMemsetAsync(grads, 0, (dtype == GPVAE_F64 ? 8 : 4) * vae_param_count(n, h0, h1, z), st);
    return 0;
  }
  GPVAE_REQUIRE(Y, 9, "gpvae_vae_elbo: Y is NULL");
  GPVAE_REQUIRE(ldy >= n, 10, "gpvae_vae_elbo: ldy < n");
  GPVAE_REQUIRE(eps_z, 11, "gpvae_vae_elbo: eps_z is NULL");
  GPVAE_REQUIRE(work, 15, "gpvae_vae_elbo: work is NULL");
  GPVAE_REQUIRE(aligned16(work), 15, "gpvae_vae_elbo: work must be 16-byte aligned");
  return dtype == GPVAE_F32
             ? run_vae_elbo<float>(st, B, n, h0, h1, z, params, Y, ldy, eps_z, obs_sigma, loss, grads, work)
             : run_vae_elbo<double>(st, B, n, h0, h1, z, params, Y, ldy, eps_z, obs_sigma, loss, grads, work);
}

extern "C" int gpvae_adam(void *stream, int dtype, int64_t count, void *params, const void *grads,
                          void *m, void *v, int64_t t, double lr, double b1, double b2,
                          double eps) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_adam: bad dtype %d", dtype);
  GPVAE_REQUIRE(count >= 0, 3, "gpvae_adam: count < 0");
  GPVAE_REQUIRE(t >= 1, 8, "gpvae_adam: step t counts from 1");
  if (count == 0) return 0;
  GPVAE_REQUIRE(params && grads && m && v, 4, "gpvae_adam: NULL pointer");
  const double c1 = 1.0 - pow(b1, (double)t), c2 = 1.0 - pow(b2, (double)t);
  int64_t blocks = cdiv(count, 256);
  if (blocks > 2048) blocks = 2048;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((adam_kernel<float>), dim3((unsigned)blocks), dim3(256), 0, st, count,
                       (float *)params, (const float *)grads, (float *)m, (float *)v, (float)lr,
                       (float)b1, (float)b2, (float)eps, (float)c1, (float)c2);
  else
    hipLaunchKernelGGL((adam_kernel<double>), dim3((unsigned)blocks), dim3(256), 0, st, count,
                       (double *)params, (const double *)grads, (double *)m, (double *)v, lr, b1,
                       b2, eps, c1, c2);
  GPVAE_CHECK_LAUNCH("gpvae_adam");
  return 0;
}

extern "C" int64_t gpvae_encoder_workspace_bytes(int dtype, int64_t B, int64_t n, int h0, int h1,
                                                 int z) {
  (void)n; (void)z;
  return B * ((int64_t)h0 + (int64_t)h1) * (dtype == GPVAE_F64 ? 8 : 4) + 256;
}

extern "C" int gpvae_encoder_fwd(void *stream, int dtype, int64_t B, int64_t n, int h0, int h1, int z,
                                 const void *enc_params, const void *Y, int64_t ldy,
                                 const void *eps_z, void *mu, void *sd, void *Z, void *work) {
  if (int rc = check_dims("gpvae_encoder_fwd", dtype, B, n, h0, h1, z)) return rc;
  if (B == 0) return 0;
  GPVAE_REQUIRE(enc_params, 8, "gpvae_encoder_fwd: enc_params is NULL");
  GPVAE_REQUIRE(Y, 9, "gpvae_encoder_fwd: Y is NULL");
  GPVAE_REQUIRE(ldy >= n, 10, "gpvae_encoder_fwd: ldy < n");
  GPVAE_REQUIRE(mu && sd, 12, "gpvae_encoder_fwd: mu / sd is NULL");
  GPVAE_REQUIRE((Z == nullptr) || eps_z, 11, "gpvae_encoder_fwd: Z wanted but eps_z is NULL");
  GPVAE_REQUIRE(work && aligned16(work), 15, "gpvae_encoder_fwd: work is NULL or not 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? run_encoder<float>(st, B, n, h0, h1, z, enc_params, Y, ldy, eps_z, mu, sd, Z, work)
             : run_encoder<double>(st, B, n, h0, h1, z, enc_params, Y, ldy, eps_z, mu, sd, Z, work);
}

extern "C" int64_t gpvae_decoder_workspace_bytes(int dtype, int64_t chains, int64_t n, int h0,
                                                 int h1, int z) {
  (void)z;
  // D1 | D2 | dD2 | the k-slices of dF V3^T (the VJP's one n-sized product)
  const int64_t ns = chains > 0 && n > 0 ? (dtype == GPVAE_F64 ? gemm_slices<double>(n, decoder_vjp_splits(chains, n, h0))
                                                                : gemm_slices<float>(n, decoder_vjp_splits(chains, n, h0)))
                                         : 1;
  return chains * ((int64_t)h1 + (2 + ns) * (int64_t)h0) * (dtype == GPVAE_F64 ? 8 : 4) + 256;
}

extern "C" int gpvae_decoder_fwd(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1,
                                 int z, const void *dec_params, const void *Z, void *F,
                                 void *work) {
  if (int rc = check_dims("gpvae_decoder_fwd", dtype, chains, n, h0, h1, z)) return rc;
  if (chains == 0) return 0;
  GPVAE_REQUIRE(dec_params && Z && F && work, 8, "gpvae_decoder_fwd: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? run_decoder<float>(st, false, chains, n, h0, h1, z, dec_params, Z, nullptr, F, work)
             : run_decoder<double>(st, false, chains, n, h0, h1, z, dec_params, Z, nullptr, F, work);
}

extern "C" int gpvae_decoder_vjp(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1,
                                 int z, const void *dec_params, const void *Z, const void *dF,
                                 void *dZ, void *work) {
  if (int rc = check_dims("gpvae_decoder_vjp", dtype, chains, n, h0, h1, z)) return rc;
  if (chains == 0) return 0;
  GPVAE_REQUIRE(dec_params && Z && dF && dZ && work, 8, "gpvae_decoder_vjp: NULL pointer");
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? run_decoder<float>(st, true, chains, n, h0, h1, z, dec_params, Z, dF, dZ, work)
             : run_decoder<double>(st, true, chains, n, h0, h1, z, dec_params, Z, dF, dZ, work);
}

extern "C" int gpvae_vae_potential_supported(int dtype, int h0, int h1, int z, int64_t n_obs) {
  if (dtype != GPVAE_F32 && dtype != GPVAE_F64) return 0;
  if (h0 < 1 || h1 < 1 || z < 1 || n_obs < 1) return 0;
  return vae_potential_lds(dtype == GPVAE_F64 ? 8 : 4, 4, h0, h1, z, n_obs) <= 64 * 1024;
}

extern "C" int gpvae_vae_potential(void *stream, int dtype, int64_t chains, int64_t n, int h0, int h1, int z,
                                   const void *dec_params, const void *q, int64_t ldq, const void *y_obs,
                                   const int64_t *obs_idx, int64_t n_obs, double noise_scale, void *U,
                                   void *dU, int64_t lddu) {
  if (int rc = check_dims("gpvae_vae_potential", dtype, chains, n, h0, h1, z)) return rc;
  GPVAE_REQUIRE(n_obs >= 1 && n_obs <= n, 13, "gpvae_vae_potential: n_obs out of range [1, n]");
  GPVAE_REQUIRE(gpvae_vae_potential_supported(dtype, h0, h1, z, n_obs), 5,
                "gpvae_vae_potential: decoder head + gathered output columns exceed 64 KB of LDS "
                "(gpvae_vae_potential_supported); use gpvae_decoder_fwd / gpvae_decoder_vjp");
  if (chains == 0) return 0;
  GPVAE_REQUIRE(dec_params && q && y_obs && obs_idx && U && dU, 8, "gpvae_vae_potential: NULL pointer");
  GPVAE_REQUIRE(ldq >= z + 1, 10, "gpvae_vae_potential: ldq < z + 1");
  GPVAE_REQUIRE(lddu >= z + 1, 17, "gpvae_vae_potential: lddu < z + 1");
  GPVAE_REQUIRE(noise_scale > 0, 14, "gpvae_vae_potential: noise_scale must be > 0");
  hipStream_t st = (hipStream_t)stream;
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  int rb = 4;
  for (int c : {16, 8})
    if (vae_potential_lds(es, c, h0, h1, z, n_obs) <= 64 * 1024) { rb = c; break; }
  const size_t shm = (size_t)vae_potential_lds(es, rb, h0, h1, z, n_obs);
  const unsigned nblk = (unsigned)cdiv(chains, rb);
#define GPVAE_POT_LAUNCH(T, RBV)                                                                          \
  hipLaunchKernelGGL((vae_potential_kernel<T, RBV>), dim3(nblk), dim3(NTHREADS), shm, st, chains, n, h0, h1, \
                     z, (int)n_obs, (const T *)dec_params, (const T *)q, ldq, (const T *)y_obs, obs_idx,   \
                     (T)noise_scale, (T *)U, (T *)dU, lddu)
  if (dtype == GPVAE_F32) {
    if (rb == 16) GPVAE_POT_LAUNCH(float, 16); else if (rb == 8) GPVAE_POT_LAUNCH(float, 8); else GPVAE_POT_LAUNCH(float, 4);
  } else {
    if (rb == 16) GPVAE_POT_LAUNCH(double, 16); else if (rb == 8) GPVAE_POT_LAUNCH(double, 8); else GPVAE_POT_LAUNCH(double, 4);
  }
#undef GPVAE_POT_LAUNCH
  GPVAE_CHECK_LAUNCH("gpvae_vae_potential");
  return 0;
}

// ---- the one entry point that takes no workspace and allocates nothing ------------------------------
// For callers that cannot allocate or carry a workspace: code under HIP graph capture, tight MCMC /
// serving loops, harnesses that forbid hipMalloc.  Same kernels as gpvae_decoder_vjp; the scratch
// (D1 | D2 | dD2 | ns k-slice slabs of dD2: chains * (h1 + (2 + ns) h0) elements, exactly
// gpvae_decoder_workspace_bytes — run_decoder's layout) is a file-scope device array, one copy per GPU
// for the life of the process.  In host code the array's NAME is only a stand-in: its device address is
// looked up once per device with hipGetSymbolAddress and only that pointer is handed to kernels.
// Every call on a GPU shares that GPU's array (calls that overlap on one GPU race).  Nothing needs
// clearing: each k-slice of the n-sized product STORES its slab and the mid-layer kernel adds the slabs
// in slice order (no accumulator that must start from zero, no memset).
static_assert(GPVAE_STATIC_SCRATCH_BYTES == (4 << 20), "include/gpvae.h states 4 MiB per GPU");
static __device__ __align__(256) unsigned char scratch[GPVAE_STATIC_SCRATCH_BYTES];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mu;
void *g_scratch_ptr[kMaxDevices] = {};

// device address of `scratch` on the CURRENT device (looked up once per device)
int static_scratch(void **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    gpvae::set_error("gpvae_decoder_vjp_static: hipGetDevice failed or device index %d out of range: %s", dev,
                     hipGetErrorString(e));
    return -1;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mu);
  if (g_scratch_ptr[dev] == nullptr) {
    void *p = nullptr;
    e = hipGetSymbolAddress(&p, HIP_SYMBOL(scratch));
    if (e != hipSuccess || p == nullptr) {
      gpvae::set_error("gpvae_decoder_vjp_static: hipGetSymbolAddress(scratch) failed on device %d: %s", dev,
                       hipGetErrorString(e));
      return -1;
    }
    g_scratch_ptr[dev] = p;
  }
  *out = g_scratch_ptr[dev];
  return 0;
}
}  // namespace

extern "C" int gpvae_decoder_vjp_static(void *stream, int dtype, int64_t chains, int64_t n, int h0,
                                        int h1, int z, const void *dec_params, const void *Z,
                                        const void *dF, void *dZ) {
  if (int rc = check_dims("gpvae_decoder_vjp_static", dtype, chains, n, h0, h1, z)) return rc;
  if (chains == 0) return 0;
  GPVAE_REQUIRE(dec_params && Z && dF && dZ, 8, "gpvae_decoder_vjp_static: NULL pointer");
  const int64_t need = gpvae_decoder_workspace_bytes(dtype, chains, n, h0, h1, z);
  GPVAE_REQUIRE(need <= (int64_t)GPVAE_STATIC_SCRATCH_BYTES, 3,
                "gpvae_decoder_vjp_static: the scratch this call needs (%lld bytes) exceeds the static "
                "scratch of %lld bytes; use gpvae_decoder_vjp with a caller workspace",
                (long long)need, (long long)GPVAE_STATIC_SCRATCH_BYTES);
  void *work = nullptr;
  if (static_scratch(&work) != 0) return -1;
  hipStream_t st = (hipStream_t)stream;
  return dtype == GPVAE_F32
             ? run_decoder<float>(st, true, chains, n, h0, h1, z, dec_params, Z, dF, dZ, work)
             : run_decoder<double>(st, true, chains, n, h0, h1, z, dec_params, Z, dF, dZ, work);
}
