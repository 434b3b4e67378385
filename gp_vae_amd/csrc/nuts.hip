// nuts.hip — the batched No-U-Turn transition between two potential evaluations.
//
// The reference hands its models to numpyro.infer.NUTS (/root/reference/src/model/inference.py:84-95,
// 477-487); NumPyro is third-party and absent, gp_vae_amd/nuts.py restates the published algorithm
// (multinomial NUTS, iterative tree builder with momentum checkpoints, generalised U-turn criterion).
// As torch operations a leapfrog step is ~40 elementwise launches around a 30 us potential; here it is two:
// PRE (half kick + drift) and POST (half kick, energy, multinomial choice, momentum sums, U-turn checks).
// One wave per chain: lanes stride the D coordinates, dot products are wave sums, a chain's branch
// conditions are wave-uniform, and a chain whose tree has stopped is not touched (no dependence on the
// other chains of the batch).  Latency-bound by construction (C x D is a few thousand numbers).
#include "philox.hpp"
#include <cmath>
#include <limits>

namespace gpvae {

template <typename T> __device__ __forceinline__ T dlogn(T v);
template <> __device__ __forceinline__ float dlogn<float>(float v) { return logf(v); }
template <> __device__ __forceinline__ double dlogn<double>(double v) { return log(v); }

template <typename T> __device__ __forceinline__ T neg_inf() { return -std::numeric_limits<T>::infinity(); }
template <typename T> __device__ __forceinline__ T log_add_exp(T a, T b) {
  const T m = a > b ? a : b;
  if (m == neg_inf<T>()) return m;
  return m + dlogn<T>(dexp<T>(a - m) + dexp<T>(b - m));
}
template <typename T> __device__ __forceinline__ bool finite_(T v) { return v - v == T(0); }

// sum_j minv_j a_j (s_j - (a_j + b_j)/2) <= 0  or the same with b in front: the generalised U-turn test
template <typename T>
__device__ __forceinline__ bool turning(const T *minv, const T *pa, const T *pb, const T *sum, const T *sub,
                                        const T *add, int64_t D, int lane) {
  // momentum sum of the span = sum - sub + add (sub / add may be null: the plain sum)
  T d1 = T(0), d2 = T(0);
  for (int64_t j = lane; j < D; j += 64) {
    T s = sum[j];
    if (sub != nullptr) s = s - sub[j] + add[j];
    const T adj = s - T(0.5) * (pa[j] + pb[j]);
    d1 += minv[j] * pa[j] * adj;
    d2 += minv[j] * pb[j] * adj;
  }
  d1 = wave_sum<T>(d1);
  d2 = wave_sum<T>(d2);
  return d1 <= T(0) || d2 <= T(0);
}

// half kick and drift from the sub-tree's edge: ph, qn (lane j touches element j, j + 64, ... only)
template <typename T>
__device__ __forceinline__ void nuts_pre(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t D, int lane,
                                         const T *minv) {
  const T es = ((const T *)b.es)[c];
  const T *pe = (const T *)b.pe + o, *ge = (const T *)b.ge + o, *qe = (const T *)b.qe + o;
  T *ph = (T *)b.ph + o, *qn = (T *)b.qn + o;
  for (int64_t j = lane; j < D; j += 64) {
    const T h = pe[j] - T(0.5) * es * ge[j];
    ph[j] = h;
    qn[j] = qe[j] + es * minv[j] * h;
  }
}

// ---- the phases as wave-level device functions (one wave = one chain) ------------------------------------
// Shared by nuts_kernel (one phase per launch, the caller's potential in between) and nuts_vae_iter_kernel
// (a whole iteration per launch).  State lives in the caller's buffers; lane j touches elements j, j + 64, ...
// of a chain's vectors, scalars are written by lane 0.  Return values are wave-uniform.
#define V(name) ((T *)b.name + o)
#define S(name) (((T *)b.name)[c])

template <typename T>
__device__ __forceinline__ void nuts_init(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t D, int lane,
                                          const T *minv) {
  T k = T(0);
  for (int64_t j = lane; j < D; j += 64) {
    const T p0 = ((const T *)b.z)[o + j] / dsqrt<T>(minv[j]);
    const T qj = V(q)[j], gj = V(g)[j];
    V(pl)[j] = p0; V(pr)[j] = p0; V(p_sum)[j] = p0;
    V(ql)[j] = qj; V(qr)[j] = qj; V(q_prop)[j] = qj;
    V(gl)[j] = gj; V(gr)[j] = gj; V(g_prop)[j] = gj;
    k += minv[j] * p0 * p0;
  }
  k = wave_sum<T>(k);
  if (lane == 0) {
    S(H0) = S(U) + T(0.5) * k;
    S(U_prop) = S(U);
    S(logW) = T(0);
    S(sum_acc) = T(0);
    S(n_prop) = T(0);
    b.active[c] = 1;
    b.diverged[c] = 0;
  }
}

// doubling `depth`: direction, empty sub-tree at that edge.  active: the chain's tree is still growing.
template <typename T>
__device__ __forceinline__ void nuts_begin(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t D, int lane,
                                           T u_dir, int active, bool clear_flags) {
  const bool right = u_dir < T(0.5);
  for (int64_t j = lane; j < D; j += 64) {
    const T qj = right ? V(qr)[j] : V(ql)[j], gj = right ? V(gr)[j] : V(gl)[j];
    V(qe)[j] = qj;
    V(pe)[j] = right ? V(pr)[j] : V(pl)[j];
    V(ge)[j] = gj;
    V(sub_psum)[j] = T(0);
    V(sq)[j] = qj;
    V(sg)[j] = gj;
  }
  if (lane == 0) {
    const T e = ((const T *)b.eps)[c];
    S(es) = right ? e : -e;
    S(sub_logW) = neg_inf<T>();
    S(sU) = S(U);
    b.sub_stop[c] = 0;
    b.building[c] = active;
    b.right[c] = right ? 1 : 0;
    if (clear_flags && c == 0) {
      b.flags[0] = 0;
      b.flags[1] = 0;
    }
  }
}

// second half kick at the new point (Un, gn), energy error, multinomial choice, momentum sum, U-turn checks
// against the checkpoints [ck_lo, ck_hi].  Returns 1 while the sub-tree keeps building.
template <typename T>
__device__ __forceinline__ int nuts_post(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t C, int64_t D,
                                         int lane, const T *minv, T u, int leaf, int ck_lo, int ck_hi,
                                         bool set_flag) {
  const T es = S(es);
  const T *gn = (const T *)b.gn + c * b.ldg;
  const T Un = ((const T *)b.Un)[c];
  T k = T(0);
  for (int64_t j = lane; j < D; j += 64) {
    const T pn = V(ph)[j] - T(0.5) * es * gn[j];
    V(ph)[j] = pn;  // ph now holds the leaf's momentum
    k += minv[j] * pn * pn;
  }
  k = wave_sum<T>(k);
  T delta = Un + T(0.5) * k - S(H0);
  if (!finite_<T>(delta)) delta = -neg_inf<T>();
  const bool div = delta > T(1000);
  const T sub_logW = S(sub_logW);
  const T new_logW = log_add_exp<T>(sub_logW, -delta);
  const bool take = !div && new_logW != neg_inf<T>() && u < dexp<T>(-delta - new_logW);
  const T *pn = V(ph);
  for (int64_t j = lane; j < D; j += 64) {
    const T qn = V(qn)[j], gj = gn[j];
    if (take) { V(sq)[j] = qn; V(sg)[j] = gj; }
    V(sub_psum)[j] += pn[j];
    V(qe)[j] = qn; V(pe)[j] = pn[j]; V(ge)[j] = gj;
  }
  bool turn = false;
  const int64_t cs = C * D;  // checkpoint stride
  if ((leaf & 1) == 0) {
    for (int64_t j = lane; j < D; j += 64) {
      ((T *)b.ck_p)[ck_hi * cs + o + j] = pn[j];
      ((T *)b.ck_s)[ck_hi * cs + o + j] = V(sub_psum)[j];
    }
  } else {
    __builtin_amdgcn_s_waitcnt(0);  // this wave's own stores to sub_psum above are read back below
    for (int i = ck_hi; i >= ck_lo && !turn; --i)
      turn = turning<T>(minv, (const T *)b.ck_p + i * cs + o, pn, V(sub_psum), (const T *)b.ck_s + i * cs + o,
                        (const T *)b.ck_p + i * cs + o, D, lane);
  }
  if (lane == 0) {
    S(sum_acc) += dexp<T>(-(delta > T(0) ? delta : T(0)));
    S(n_prop) += T(1);
    S(sub_logW) = new_logW;
    if (take) S(sU) = Un;
    if (div) b.diverged[c] = 1;
    if (turn || div) {
      b.sub_stop[c] = 1;
      b.building[c] = 0;
    } else if (set_flag) {
      atomicOr(&b.flags[0], 1);
    }
  }
  return (turn || div) ? 0 : 1;
}

// biased progressive choice between the tree and the finished sub-tree; returns 1 while the tree keeps growing
// (sub_stop: the sub-tree turned or diverged — the tree ends without it)
template <typename T>
__device__ __forceinline__ int nuts_merge(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t D, int lane,
                                          const T *minv, T u, int sub_stop, bool set_flag) {
  if (sub_stop) {
    if (lane == 0) b.active[c] = 0;
    return 0;
  }
  const T logW = S(logW), sub_logW = S(sub_logW);
  const T dl = sub_logW - logW;
  const bool take = u < dexp<T>(dl < T(0) ? dl : T(0));
  const bool right = b.right[c] != 0;
  for (int64_t j = lane; j < D; j += 64) {
    if (take) { V(q_prop)[j] = V(sq)[j]; V(g_prop)[j] = V(sg)[j]; }
    V(p_sum)[j] += V(sub_psum)[j];
    if (right) { V(qr)[j] = V(qe)[j]; V(pr)[j] = V(pe)[j]; V(gr)[j] = V(ge)[j]; }
    else { V(ql)[j] = V(qe)[j]; V(pl)[j] = V(pe)[j]; V(gl)[j] = V(ge)[j]; }
  }
  __builtin_amdgcn_s_waitcnt(0);
  const bool turn = turning<T>(minv, V(pl), V(pr), V(p_sum), (const T *)nullptr, (const T *)nullptr, D, lane);
  if (lane == 0) {
    if (take) S(U_prop) = S(sU);
    S(logW) = log_add_exp<T>(logW, sub_logW);
    b.active[c] = turn ? 0 : 1;
    if (!turn && set_flag) atomicOr(&b.flags[1], 1);
  }
  return turn ? 0 : 1;
}

template <typename T>
__device__ __forceinline__ void nuts_accept(const gpvae_nuts_buffers &b, int64_t c, int64_t o, int64_t D, int lane) {
  for (int64_t j = lane; j < D; j += 64) {
    V(q)[j] = V(q_prop)[j];
    V(g)[j] = V(g_prop)[j];
  }
  if (lane == 0) {
    S(U) = S(U_prop);
    const T np = S(n_prop);
    S(accept) = S(sum_acc) / (np > T(1) ? np : T(1));
  }
}

template <typename T>
__global__ __launch_bounds__(64) void nuts_kernel(int op, int64_t C, int64_t D, int max_depth, int depth, int leaf,
                                                  int ck_lo, int ck_hi, gpvae_nuts_buffers b) {
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t o = c * D;
  const T *minv = (const T *)b.minv + o;
  if (op == GPVAE_NUTS_INIT) {
    nuts_init<T>(b, c, o, D, lane, minv);
  } else if (op == GPVAE_NUTS_BEGIN) {
    nuts_begin<T>(b, c, o, D, lane, ((const T *)b.unif)[c * b.ldu + depth], b.active[c], true);
  } else if (op == GPVAE_NUTS_PRE) {
    nuts_pre<T>(b, c, o, D, lane, minv);
  } else if (op == GPVAE_NUTS_POST || op == GPVAE_NUTS_POST_PRE) {
    // POST_PRE: the leaf's POST and the next leaf's PRE in one launch (PRE reads, lane by lane, what the same
    // lane's POST wrote: qe, pe, ge).  A chain that is not building skips POST and still runs PRE, as it would
    // under the separate ops.
    if (b.building[c])  // wave-uniform
      nuts_post<T>(b, c, o, C, D, lane, minv,
                   ((const T *)b.unif)[c * b.ldu + 2 * max_depth + ((int64_t)1 << depth) - 1 + leaf], leaf, ck_lo, ck_hi, true);
    if (op == GPVAE_NUTS_POST_PRE) nuts_pre<T>(b, c, o, D, lane, minv);
  } else if (op == GPVAE_NUTS_MERGE) {
    if (!b.active[c]) return;
    nuts_merge<T>(b, c, o, D, lane, minv, ((const T *)b.unif)[c * b.ldu + max_depth + depth], b.sub_stop[c], true);
  } else if (op == GPVAE_NUTS_ACCEPT) {
    nuts_accept<T>(b, c, o, D, lane);
  }
}

// ---- C5 in one launch per iteration: the transition above with VAEInference.regression's potential inside ---
// (src/model/inference.py:84-95 driving inference.py:235-254.)  Four chains per workgroup, one wave each; the
// decoder's head and its output layer gathered at the observed columns sit in LDS once per workgroup
// (gpvae_vae_potential's layout); every wave runs ITS chain's whole iteration — all doublings and leapfrog steps
// up to its own tree's end — so a chain neither waits for the deepest tree of the batch nor costs a launch per
// leapfrog step.  Same phases, same uniforms per (depth, leaf), same arithmetic per chain as the lock-step
// driver: the two give the same chains.  Phases exchange state through the caller's buffers (L2-resident);
// a workgroup-scope fence between phases orders a lane's loads behind the other lanes' stores.
template <typename T> struct VaePotLds {
  const T *V1, *c1, *V2, *c2, *V3g, *c3g, *yo;  // weights (shared by the workgroup)
  T *zq, *D1, *D2, *R, *G2, *G1;                // this wave's scratch
};

#define GPVAE_WAVE_FENCE() __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "workgroup")

// U and dU/dq at q = (z, log noise) for ONE chain by one wave; the summation orders are gpvae_vae_potential's
// (dense_fwd / dense_bwd_in in csrc/vae.hip: k ascending from the bias; the scalar sums serially)
template <typename T>
__device__ __forceinline__ void vae_pot_wave(const VaePotLds<T> &w, int h0, int h1, int z, int n_obs, T s0p,
                                             const T *q, T *Uout, T *gout, int lane) {
  for (int i = lane; i <= z; i += 64) w.zq[i] = q[i];
  GPVAE_WAVE_FENCE();
  for (int i = lane; i < h1; i += 64) {
    T a = w.c1[i];
    for (int k = 0; k < z; ++k) a += w.zq[k] * w.V1[k * h1 + i];
    w.D1[i] = a > T(0) ? a : T(0);
  }
  GPVAE_WAVE_FENCE();
  for (int i = lane; i < h0; i += 64) {
    T a = w.c2[i];
    for (int k = 0; k < h1; ++k) a += w.D1[k] * w.V2[k * h0 + i];
    w.D2[i] = a > T(0) ? a : T(0);
  }
  GPVAE_WAVE_FENCE();
  for (int j = lane; j < n_obs; j += 64) {
    T a = w.c3g[j];
    for (int k = 0; k < h0; ++k) a += w.D2[k] * w.V3g[k * n_obs + j];
    w.R[j] = a - w.yo[j];
  }
  GPVAE_WAVE_FENCE();
  // the scalar part, every lane redundantly (same order as the one-lane loop of vae_potential_kernel)
  const T u = w.zq[z];
  T sq = T(0), zz = T(0);
  for (int j = 0; j < n_obs; ++j) sq += w.R[j] * w.R[j];
  for (int j = 0; j < z; ++j) zz += w.zq[j] * w.zq[j];
  T du, iv;
  const T Uc = vae_potential_scalar<T>(sq, zz, u, s0p, n_obs, z, du, iv);
  if (lane == 0) {
    *Uout = Uc;
    gout[z] = du;
  }
  GPVAE_WAVE_FENCE();  // every lane has read R before it is scaled
  for (int j = lane; j < n_obs; j += 64) w.R[j] *= iv;  // dF at the observed columns
  GPVAE_WAVE_FENCE();
  for (int i = lane; i < h0; i += 64) {
    T a = T(0);
    for (int j = 0; j < n_obs; ++j) a += w.R[j] * w.V3g[i * n_obs + j];
    w.G2[i] = w.D2[i] > T(0) ? a : T(0);
  }
  GPVAE_WAVE_FENCE();
  for (int i = lane; i < h1; i += 64) {
    T a = T(0);
    for (int k = 0; k < h0; ++k) a += w.G2[k] * w.V2[i * h0 + k];
    w.G1[i] = w.D1[i] > T(0) ? a : T(0);
  }
  GPVAE_WAVE_FENCE();
  for (int i = lane; i < z; i += 64) {
    T a = T(0);
    for (int k = 0; k < h1; ++k) a += w.G1[k] * w.V1[i * h1 + k];
    gout[i] = w.zq[i] + a;
  }
}

constexpr int NUTS_WPB = 4;  // chains (waves) per workgroup of the fused kernel

// LDS: the decoder (shared by the workgroup), then per wave the potential's scratch and the chain's whole NUTS
// state (20 vectors of D = z + 1, 11 scalars, 5 flags, 2 (max_depth + 1) checkpoint vectors)
constexpr int NUTS_NVEC = 20, NUTS_NSCA = 11, NUTS_NINT = 5;
static inline int64_t nuts_vae_wave_elems(int h0, int h1, int z, int64_t n_obs, int max_depth) {
  const int64_t D = z + 1;
  return (z + 1) + 2 * h1 + 2 * h0 + n_obs + NUTS_NVEC * D + NUTS_NSCA + 2 * (int64_t)(max_depth + 1) * D;
}
static inline int64_t nuts_vae_lds_bytes(int64_t es, int h0, int h1, int z, int64_t n_obs, int max_depth,
                                         int64_t extra_elems = 0) {
  const int64_t shared = (int64_t)z * h1 + h1 + (int64_t)h1 * h0 + h0 + (int64_t)h0 * n_obs + 2 * n_obs;
  return round_up(es * (shared + NUTS_WPB * (nuts_vae_wave_elems(h0, h1, z, n_obs, max_depth) + extra_elems)), 16) +
         NUTS_WPB * NUTS_NINT * 4;
}

// the workgroup's decoder in LDS (all threads; ends with the only workgroup barrier) and this wave's blocks
template <typename T> struct NutsVaeCtx {
  VaePotLds<T> w;
  gpvae_nuts_buffers lb;  // state pointers -> this wave's LDS block, chain index 0
  T *extra;               // this wave's spare block (the run kernel's adaptation state), `extra_elems` long
};
template <typename T>
__device__ __forceinline__ void nuts_vae_setup(NutsVaeCtx<T> &x, unsigned char *smem, const gpvae_nuts_buffers &b,
                                               int64_t n, int h0, int h1, int z, int n_obs, int max_depth,
                                               const T *dec_params, const T *y_obs, const int64_t *obs_idx,
                                               int64_t int_off, int extra_elems, int tid) {
  T *sm = reinterpret_cast<T *>(smem);
  const int nmid = z * h1 + h1 + h1 * h0 + h0;
  T *V3g = sm + nmid, *c3g = V3g + h0 * n_obs, *yo = c3g + n_obs;
  const int wave = tid >> 6;
  const T *V3 = dec_params + nmid, *c3 = V3 + (int64_t)h0 * n;
  for (int i = tid; i < nmid; i += 64 * NUTS_WPB) sm[i] = dec_params[i];
  auto col = [&](int j) -> int64_t {  // out-of-range indices are clamped (memory safety; hosts reject them)
    const int64_t cc = obs_idx[j];
    return cc < 0 ? 0 : (cc >= n ? n - 1 : cc);
  };
  for (int i = tid; i < h0 * n_obs; i += 64 * NUTS_WPB) V3g[i] = V3[(int64_t)(i / n_obs) * n + col(i % n_obs)];
  for (int i = tid; i < n_obs; i += 64 * NUTS_WPB) {
    c3g[i] = c3[col(i)];
    yo[i] = y_obs[i];
  }
  __syncthreads();  // the only workgroup barrier: from here on every wave is on its own
  const int64_t D = z + 1;
  const int per_wave = (int)((z + 1) + 2 * h1 + 2 * h0 + n_obs + NUTS_NVEC * D + NUTS_NSCA +
                             2 * (int64_t)(max_depth + 1) * D) + extra_elems;
  T *ws = yo + n_obs + wave * per_wave;
  VaePotLds<T> &w = x.w;
  w.V1 = sm; w.c1 = w.V1 + z * h1; w.V2 = w.c1 + h1; w.c2 = w.V2 + h1 * h0;
  w.V3g = V3g; w.c3g = c3g; w.yo = yo;
  w.zq = ws; w.D1 = w.zq + (z + 1); w.D2 = w.D1 + h1; w.R = w.D2 + h0; w.G2 = w.R + n_obs; w.G1 = w.G2 + h0;
  // The chain's state in LDS: a copy of the buffer table whose state pointers name this wave's LDS block, so
  // that the phases run with chain index 0 (and C = 1: checkpoint stride D).  A global round trip per phase
  // (store, fence, load: ~2 us) was what a leapfrog step cost with the state in the caller's buffers.
  T *sv = w.G1 + h1;                       // NUTS_NVEC vectors
  T *ss = sv + NUTS_NVEC * D;              // NUTS_NSCA scalars
  T *sck = ss + NUTS_NSCA;                 // checkpoints: [max_depth + 1][D] twice
  x.extra = sck + 2 * (int64_t)(max_depth + 1) * D;
  int32_t *si = reinterpret_cast<int32_t *>(smem + int_off) + wave * NUTS_NINT;
  gpvae_nuts_buffers &lb = x.lb;
  lb = b;
  void **vecs[NUTS_NVEC] = {&lb.q, &lb.g, &lb.ql, &lb.pl, &lb.gl, &lb.qr, &lb.pr, &lb.gr, &lb.q_prop, &lb.g_prop,
                            &lb.p_sum, &lb.qe, &lb.pe, &lb.ge, &lb.sub_psum, &lb.sq, &lb.sg, &lb.ph, &lb.qn,
                            const_cast<void **>(&lb.gn)};
  for (int k = 0; k < NUTS_NVEC; ++k) *vecs[k] = sv + k * D;
  void **scas[NUTS_NSCA] = {&lb.U, &lb.U_prop, &lb.logW, &lb.H0, &lb.sum_acc, &lb.n_prop, &lb.accept, &lb.es,
                            &lb.sub_logW, &lb.sU, const_cast<void **>(&lb.Un)};
  for (int k = 0; k < NUTS_NSCA; ++k) *scas[k] = ss + k;
  lb.active = si; lb.diverged = si + 1; lb.sub_stop = si + 2; lb.building = si + 3; lb.right = si + 4;
  lb.ck_p = sck; lb.ck_s = sck + (int64_t)(max_depth + 1) * D;
  lb.ldg = D;
}

// one NUTS transition of the chain whose state sits in x.lb (q, U, g in; q, U, g, accept, n_prop, diverged out).
// unif(idx): uniform number idx of this chain's iteration (a direction and a transition draw per doubling, one
// multinomial draw per leaf: gp_vae_amd/nuts.py uniforms_per_iteration); lb.z: the iteration's normals; lb.eps.
template <typename T, typename Unif>
__device__ __forceinline__ void nuts_vae_transition(NutsVaeCtx<T> &x, int h0, int h1, int z, int n_obs, T s0p,
                                                    int max_depth, const T *minv, int lane, Unif unif) {
  const gpvae_nuts_buffers &lb = x.lb;
  const int64_t D = z + 1;
  nuts_init<T>(lb, 0, 0, D, lane, minv);
  int active = 1;
  for (int depth = 0; depth < max_depth && active; ++depth) {  // at most max_depth doublings: every wave leaves
    GPVAE_WAVE_FENCE();
    nuts_begin<T>(lb, 0, 0, D, lane, unif(depth), 1, false);
    GPVAE_WAVE_FENCE();
    int building = 1;
    const int leaves = 1 << depth;
    for (int leaf = 0; leaf < leaves && building; ++leaf) {  // at most 2^depth leaves
      nuts_pre<T>(lb, 0, 0, D, lane, minv);
      GPVAE_WAVE_FENCE();
      vae_pot_wave<T>(x.w, h0, h1, z, n_obs, s0p, (const T *)lb.qn, (T *)lb.Un, (T *)lb.gn, lane);
      GPVAE_WAVE_FENCE();
      const int idx_max = __popc((unsigned)(leaf >> 1));
      const int ones = __ffs(~leaf) - 1;  // trailing ones of leaf
      building = nuts_post<T>(lb, 0, 0, 1, D, lane, minv, unif(2 * max_depth + leaves - 1 + leaf), leaf,
                              idx_max - ones + 1, idx_max, false);
      GPVAE_WAVE_FENCE();
    }
    active = nuts_merge<T>(lb, 0, 0, D, lane, minv, unif(max_depth + depth), building ? 0 : 1, false);
  }
  GPVAE_WAVE_FENCE();
  nuts_accept<T>(lb, 0, 0, D, lane);
  GPVAE_WAVE_FENCE();
}

template <typename T>
__global__ __launch_bounds__(64 * NUTS_WPB) void nuts_vae_iter_kernel(int64_t C, int max_depth, gpvae_nuts_buffers b,
                                                                      int64_t n, int h0, int h1, int z, int n_obs,
                                                                      const T *dec_params, const T *y_obs,
                                                                      const int64_t *obs_idx, T s0p, int64_t int_off) {
  extern __shared__ __align__(16) unsigned char nuts_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  NutsVaeCtx<T> x;
  nuts_vae_setup<T>(x, nuts_smem, b, n, h0, h1, z, n_obs, max_depth, dec_params, y_obs, obs_idx, int_off, 0, tid);
  const int64_t c = (int64_t)blockIdx.x * NUTS_WPB + wave;
  if (c >= C) return;
  const int64_t D = z + 1, o = c * D;
  gpvae_nuts_buffers &lb = x.lb;
  lb.z = (const T *)b.z + o;
  lb.eps = (const T *)b.eps + c;
  const T *un = (const T *)b.unif + c * b.ldu;
  const T *minv = (const T *)b.minv + o;
  for (int64_t j = lane; j < D; j += 64) {
    ((T *)lb.q)[j] = ((const T *)b.q)[o + j];
    ((T *)lb.g)[j] = ((const T *)b.g)[o + j];
  }
  if (lane == 0) ((T *)lb.U)[0] = ((const T *)b.U)[c];
  GPVAE_WAVE_FENCE();
  nuts_vae_transition<T>(x, h0, h1, z, n_obs, s0p, max_depth, minv, lane, [&](int idx) -> T { return un[idx]; });
  // what the host reads between iterations: the new state, the acceptance statistic, leapfrog count, divergence
  for (int64_t j = lane; j < D; j += 64) {
    ((T *)b.q)[o + j] = ((const T *)lb.q)[j];
    ((T *)b.g)[o + j] = ((const T *)lb.g)[j];
  }
  if (lane == 0) {
    ((T *)b.U)[c] = ((const T *)lb.U)[0];
    ((T *)b.accept)[c] = ((const T *)lb.accept)[0];
    ((T *)b.n_prop)[c] = ((const T *)lb.n_prop)[0];
    ((T *)b.sum_acc)[c] = ((const T *)lb.sum_acc)[0];
    b.diverged[c] = lb.diverged[0];
    b.active[c] = lb.active[0];
  }
}

// ---- the whole run in one launch ------------------------------------------------------------------------------
// Chains are independent, so nothing forces them to meet after every iteration: each wave runs ITS chain through
// warm-up and sampling — transitions as above, the iteration's normals and uniforms drawn in place from the
// Philox stream the host driver's noise_fn reads (element (chain, k) of the [chains_total, D + K] matrix of
// stream stream_base | it: identical numbers, and only the uniforms a tree actually consumes are computed),
// dual-averaging step size and Stan-window diagonal mass matrix per chain (gp_vae_amd/nuts.py _Adapter).
// The kernel then lasts as long as the chain with the most leapfrog steps over the WHOLE run (a sum over
// iterations: within a few per cent of the mean) instead of the deepest tree of every single iteration.
template <typename T> struct NutsRunArgs {
  int num_warmup, num_samples, thinning, adapt_mass;
  int win_start, n_ends, ends[24];  // Stan's warm-up windows (nuts.py _Windows)
  uint64_t seed, stream_base;
  int64_t row_len;                  // D + uniforms_per_iteration(max_depth): row length of the noise matrix
  const int64_t *chain_ids;         // global chain id of local chain c (NULL: c)
  T step_size, target;
  const T *q0;                      // [C, D]
  T *samples;                       // [num_samples / thinning (rounded up)][C][D]
  T *accept_prob, *leapfrogs, *step_out, *minv_out;  // [C], [C], [C], [C, D]
  int32_t *divergent;               // [C]
};
constexpr int NUTS_RUN_EXTRA_VECS = 4;  // z, minv, Welford mean, Welford m2

// element e of the Philox normal stream (the number gpvae_randn writes at index e)
template <typename T> __device__ __forceinline__ T philox_normal_at(int64_t e, uint64_t seed, uint64_t stream_id) {
  T zz[4];
  philox_normals4<T>(e >> 2, seed, stream_id, zz);
  const int r = (int)(e & 3);
  return r == 0 ? zz[0] : (r == 1 ? zz[1] : (r == 2 ? zz[2] : zz[3]));
}
template <typename T> __device__ __forceinline__ T derf_(T v);
template <> __device__ __forceinline__ float derf_<float>(float v) { return erff(v); }
template <> __device__ __forceinline__ double derf_<double>(double v) { return erf(v); }
template <typename T> __device__ __forceinline__ T dpow_(T a, T b);
template <> __device__ __forceinline__ float dpow_<float>(float a, float b) { return powf(a, b); }
template <> __device__ __forceinline__ double dpow_<double>(double a, double b) { return pow(a, b); }

template <typename T>
__global__ __launch_bounds__(64 * NUTS_WPB) void nuts_vae_run_kernel(int64_t C, int max_depth, gpvae_nuts_buffers b,
                                                                     int64_t n, int h0, int h1, int z, int n_obs,
                                                                     const T *dec_params, const T *y_obs,
                                                                     const int64_t *obs_idx, T s0p, int64_t int_off,
                                                                     NutsRunArgs<T> a) {
  extern __shared__ __align__(16) unsigned char nuts_smem[];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int64_t D = z + 1;
  NutsVaeCtx<T> x;
  nuts_vae_setup<T>(x, nuts_smem, b, n, h0, h1, z, n_obs, max_depth, dec_params, y_obs, obs_idx, int_off,
                    (int)(NUTS_RUN_EXTRA_VECS * D) + 1, tid);
  const int64_t c = (int64_t)blockIdx.x * NUTS_WPB + wave;
  if (c >= C) return;
  const int64_t o = c * D;
  const int64_t gc = a.chain_ids ? a.chain_ids[c] : c;
  gpvae_nuts_buffers &lb = x.lb;
  T *zbuf = x.extra, *minv = zbuf + D, *wmean = minv + D, *wm2 = wmean + D, *epsb = wm2 + D;
  lb.z = zbuf;
  lb.eps = epsb;
  for (int64_t j = lane; j < D; j += 64) {
    ((T *)lb.q)[j] = a.q0[o + j];
    minv[j] = T(1);
    wmean[j] = T(0);
    wm2[j] = T(0);
  }
  GPVAE_WAVE_FENCE();
  vae_pot_wave<T>(x.w, h0, h1, z, n_obs, s0p, (const T *)lb.q, (T *)lb.U, (T *)lb.g, lane);
  GPVAE_WAVE_FENCE();
  // adaptation state (wave-uniform scalars, every lane carries a copy)
  T log_eps = dlogn<T>(a.step_size), log_eps_bar = log_eps, h_bar = T(0), mu = log_eps + T(2.302585092994046);
  int m = 0, w_n = 0, next_end = 0;
  T acc_sum = T(0), leap_sum = T(0);
  int n_div = 0;
  const int total = a.num_warmup + a.num_samples;
  for (int it = 0; it < total; ++it) {  // a fixed number of iterations: every wave leaves
    const bool warm = it < a.num_warmup;
    const uint64_t sid = a.stream_base | (uint64_t)it;
    const int64_t e0 = gc * a.row_len;
    if (lane == 0) epsb[0] = dexp<T>(warm ? log_eps : log_eps_bar);
    for (int64_t j = lane; j < D; j += 64) zbuf[j] = philox_normal_at<T>(e0 + j, a.seed, sid);
    GPVAE_WAVE_FENCE();
    nuts_vae_transition<T>(x, h0, h1, z, n_obs, s0p, max_depth, minv, lane, [&](int idx) -> T {
      const T g = philox_normal_at<T>(e0 + D + idx, a.seed, sid);
      return T(0.5) * (T(1) + derf_<T>(g * T(0.7071067811865476)));  // uniform = Phi(normal), as the host's noise_fn
    });
    const T acc = ((const T *)lb.accept)[0];
    if (warm) {
      // dual averaging (Hoffman & Gelman 2014, alg. 5: gamma 0.05, t0 10, kappa 0.75), then the mass window
      m += 1;
      const T mt = (T)m + T(10);
      h_bar = (T(1) - T(1) / mt) * h_bar + (a.target - acc) / mt;
      log_eps = mu - dsqrt<T>((T)m) / T(0.05) * h_bar;
      const T wgt = dpow_<T>((T)m, T(-0.75));
      log_eps_bar = wgt * log_eps + (T(1) - wgt) * log_eps_bar;
      if (a.adapt_mass && a.n_ends > 0 && it >= a.win_start) {
        w_n += 1;
        for (int64_t j = lane; j < D; j += 64) {
          const T qj = ((const T *)lb.q)[j];
          const T d1 = qj - wmean[j];
          wmean[j] += d1 / (T)w_n;
          wm2[j] += d1 * (qj - wmean[j]);
        }
        if (next_end < a.n_ends && it == a.ends[next_end]) {
          next_end += 1;
          if (w_n > 1) {
            const T nn = (T)w_n;
            for (int64_t j = lane; j < D; j += 64) {
              minv[j] = (nn / (nn + T(5))) * (wm2[j] / (nn - T(1))) + T(1.0e-3) * (T(5) / (nn + T(5)));
              wmean[j] = T(0);
              wm2[j] = T(0);
            }
            w_n = 0;
            log_eps = log_eps_bar;  // restart the step-size search around the last averaged value
            mu = log_eps + T(2.302585092994046);
            h_bar = T(0);
            m = 0;
          }
        }
        GPVAE_WAVE_FENCE();
      }
    } else {
      acc_sum += acc;
      leap_sum += ((const T *)lb.n_prop)[0];
      n_div += lb.diverged[0];
      const int s = it - a.num_warmup;
      if (s % a.thinning == 0) {
        T *dst = a.samples + ((int64_t)(s / a.thinning) * C + c) * D;
        for (int64_t j = lane; j < D; j += 64) dst[j] = ((const T *)lb.q)[j];
      }
    }
  }
  const T ns = (T)(a.num_samples > 1 ? a.num_samples : 1);
  for (int64_t j = lane; j < D; j += 64) a.minv_out[o + j] = minv[j];
  if (lane == 0) {
    a.accept_prob[c] = acc_sum / ns;
    a.leapfrogs[c] = leap_sum / ns;
    a.step_out[c] = dexp<T>(log_eps_bar);
    a.divergent[c] = n_div;
  }
}
#undef V
#undef S

}  // namespace gpvae

using namespace gpvae;

extern "C" int gpvae_nuts_op(void *stream, int dtype, int op, int64_t C, int64_t D, int max_depth, int depth, int leaf,
                             int ck_lo, int ck_hi, const gpvae_nuts_buffers *b) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_nuts_op: bad dtype %d", dtype);
  GPVAE_REQUIRE(op >= GPVAE_NUTS_INIT && op <= GPVAE_NUTS_POST_PRE, 3, "gpvae_nuts_op: unknown op %d", op);
  GPVAE_REQUIRE(C >= 0 && C <= 2147483647, 4, "gpvae_nuts_op: C out of range");
  GPVAE_REQUIRE(D >= 1, 5, "gpvae_nuts_op: D < 1");
  GPVAE_REQUIRE(max_depth >= 1 && max_depth <= 20, 6, "gpvae_nuts_op: max_depth out of range [1,20]");
  GPVAE_REQUIRE(depth >= 0 && depth < max_depth, 7, "gpvae_nuts_op: depth out of range");
  GPVAE_REQUIRE(leaf >= 0 && leaf < (1 << depth), 8, "gpvae_nuts_op: leaf out of range for this depth");
  GPVAE_REQUIRE(ck_hi >= 0 && ck_hi <= max_depth && ck_lo >= 0 && ck_lo <= ck_hi + 1, 9,
                "gpvae_nuts_op: checkpoint range out of bounds");
  if (C == 0) return 0;
  GPVAE_REQUIRE(b != nullptr, 11, "gpvae_nuts_op: buffers is NULL");
  GPVAE_REQUIRE(b->q && b->U && b->g && b->minv && b->eps && b->ql && b->pl && b->gl && b->qr && b->pr && b->gr &&
                    b->q_prop && b->U_prop && b->g_prop && b->logW && b->p_sum && b->H0 && b->sum_acc && b->n_prop &&
                    b->accept && b->active && b->diverged && b->qe && b->pe && b->ge && b->es && b->sub_logW &&
                    b->sub_psum && b->sq && b->sU && b->sg && b->sub_stop && b->building && b->right && b->ck_p &&
                    b->ck_s && b->ph && b->qn && b->flags,
                11, "gpvae_nuts_op: a state buffer is NULL");
  GPVAE_REQUIRE(op != GPVAE_NUTS_INIT || b->z, 11, "gpvae_nuts_op: z is NULL");
  GPVAE_REQUIRE((op != GPVAE_NUTS_BEGIN && op != GPVAE_NUTS_POST && op != GPVAE_NUTS_POST_PRE && op != GPVAE_NUTS_MERGE) ||
                    (b->unif && b->ldu >= 2 * max_depth + ((int64_t)1 << max_depth) - 1),
                11, "gpvae_nuts_op: unif NULL or ldu < 2 max_depth + 2^max_depth - 1");
  GPVAE_REQUIRE((op != GPVAE_NUTS_POST && op != GPVAE_NUTS_POST_PRE) || (b->Un && b->gn && b->ldg >= D), 11, "gpvae_nuts_op: Un / gn NULL or ldg < D");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((nuts_kernel<float>), dim3((unsigned)C), dim3(64), 0, st, op, C, D, max_depth, depth, leaf, ck_lo,
                       ck_hi, *b);
  else
    hipLaunchKernelGGL((nuts_kernel<double>), dim3((unsigned)C), dim3(64), 0, st, op, C, D, max_depth, depth, leaf, ck_lo,
                       ck_hi, *b);
  GPVAE_CHECK_LAUNCH("gpvae_nuts_op");
  return 0;
}

extern "C" int gpvae_nuts_vae_supported(int dtype, int h0, int h1, int z, int64_t n_obs) {
  if (h0 < 1 || h1 < 1 || z < 1 || n_obs < 1 || n_obs > 4096) return 0;
  // any max_depth <= 20, with the run kernel's extra block
  return nuts_vae_lds_bytes(dtype == GPVAE_F64 ? 8 : 4, h0, h1, z, n_obs, 20, 4 * (int64_t)(z + 1) + 1) <= 64 * 1024 ? 1 : 0;
}

extern "C" int gpvae_nuts_vae_iteration(void *stream, int dtype, int64_t C, int max_depth, const gpvae_nuts_buffers *b,
                                        int64_t n, int h0, int h1, int z, const void *dec_params, const void *y_obs,
                                        const int64_t *obs_idx, int64_t n_obs, double noise_scale) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_nuts_vae_iteration: bad dtype %d", dtype);
  GPVAE_REQUIRE(C >= 0 && C <= 2147483647, 3, "gpvae_nuts_vae_iteration: C out of range");
  GPVAE_REQUIRE(max_depth >= 1 && max_depth <= 20, 4, "gpvae_nuts_vae_iteration: max_depth out of range [1,20]");
  if (C == 0) return 0;
  GPVAE_REQUIRE(b != nullptr, 5, "gpvae_nuts_vae_iteration: buffers is NULL");
  GPVAE_REQUIRE(b->q && b->U && b->g && b->minv && b->eps && b->z && b->unif && b->ql && b->pl && b->gl && b->qr &&
                    b->pr && b->gr && b->q_prop && b->U_prop && b->g_prop && b->logW && b->p_sum && b->H0 &&
                    b->sum_acc && b->n_prop && b->accept && b->active && b->diverged && b->qe && b->pe && b->ge &&
                    b->es && b->sub_logW && b->sub_psum && b->sq && b->sU && b->sg && b->sub_stop && b->building &&
                    b->right && b->ck_p && b->ck_s && b->ph && b->qn && b->Un && b->gn,
                5, "gpvae_nuts_vae_iteration: a state buffer is NULL (Un / gn are the kernel's scratch here)");
  GPVAE_REQUIRE(b->ldu >= 2 * max_depth + ((int64_t)1 << max_depth) - 1, 5,
                "gpvae_nuts_vae_iteration: ldu < 2 max_depth + 2^max_depth - 1");
  GPVAE_REQUIRE(b->ldg >= z + 1, 5, "gpvae_nuts_vae_iteration: ldg < z + 1");
  GPVAE_REQUIRE(n >= 1 && h0 >= 1 && h1 >= 1 && z >= 1, 6, "gpvae_nuts_vae_iteration: decoder dims must be positive");
  GPVAE_REQUIRE(dec_params && y_obs && obs_idx, 10, "gpvae_nuts_vae_iteration: NULL decoder / data pointer");
  GPVAE_REQUIRE(gpvae_nuts_vae_supported(dtype, h0, h1, z, n_obs), 13,
                "gpvae_nuts_vae_iteration: decoder head + gathered output columns do not fit 64 KB of LDS "
                "(gpvae_nuts_vae_supported); drive gpvae_nuts_op around gpvae_vae_potential instead");
  GPVAE_REQUIRE(noise_scale > 0, 14, "gpvae_nuts_vae_iteration: noise_scale must be positive");
  hipStream_t st = (hipStream_t)stream;
  const unsigned nblk = (unsigned)cdiv(C, (int64_t)NUTS_WPB);
  const int64_t es = dtype == GPVAE_F64 ? 8 : 4;
  const size_t shm = (size_t)nuts_vae_lds_bytes(es, h0, h1, z, n_obs, max_depth);
  const int64_t int_off = (int64_t)shm - NUTS_WPB * NUTS_NINT * 4;  // the flags' block, behind the T-typed part
  if (dtype == GPVAE_F32)
    hipLaunchKernelGGL((nuts_vae_iter_kernel<float>), dim3(nblk), dim3(64 * NUTS_WPB), shm, st, C, max_depth, *b, n, h0,
                       h1, z, (int)n_obs, (const float *)dec_params, (const float *)y_obs, obs_idx, (float)noise_scale,
                       int_off);
  else
    hipLaunchKernelGGL((nuts_vae_iter_kernel<double>), dim3(nblk), dim3(64 * NUTS_WPB), shm, st, C, max_depth, *b, n, h0,
                       h1, z, (int)n_obs, (const double *)dec_params, (const double *)y_obs, obs_idx, (double)noise_scale,
                       int_off);
  GPVAE_CHECK_LAUNCH("gpvae_nuts_vae_iteration");
  return 0;
}

template <typename T>
static void launch_nuts_run(hipStream_t st, int64_t C, const gpvae_nuts_run *r, int64_t n, int h0, int h1, int z,
                            const void *dec_params, const void *y_obs, const int64_t *obs_idx, int64_t n_obs,
                            double noise_scale) {
  NutsRunArgs<T> a;
  a.num_warmup = r->num_warmup; a.num_samples = r->num_samples; a.thinning = r->thinning; a.adapt_mass = r->adapt_mass;
  a.win_start = r->win_start; a.n_ends = r->n_ends;
  for (int i = 0; i < 24; ++i) a.ends[i] = r->ends[i];
  a.seed = r->seed; a.stream_base = r->stream_base; a.row_len = r->row_len; a.chain_ids = r->chain_ids;
  a.step_size = (T)r->step_size; a.target = (T)r->target_accept;
  a.q0 = (const T *)r->q0; a.samples = (T *)r->samples; a.accept_prob = (T *)r->accept_prob;
  a.leapfrogs = (T *)r->leapfrogs; a.step_out = (T *)r->step_size_out; a.minv_out = (T *)r->inverse_mass;
  a.divergent = r->divergent;
  const int64_t extra = NUTS_RUN_EXTRA_VECS * (int64_t)(z + 1) + 1;
  const size_t shm = (size_t)nuts_vae_lds_bytes(sizeof(T), h0, h1, z, n_obs, r->max_depth, extra);
  const int64_t int_off = (int64_t)shm - NUTS_WPB * NUTS_NINT * 4;
  gpvae_nuts_buffers none{};
  hipLaunchKernelGGL((nuts_vae_run_kernel<T>), dim3((unsigned)cdiv(C, (int64_t)NUTS_WPB)), dim3(64 * NUTS_WPB), shm, st, C,
                     (int)r->max_depth, none, n, h0, h1, z, (int)n_obs, (const T *)dec_params, (const T *)y_obs, obs_idx,
                     (T)noise_scale, int_off, a);
}

extern "C" int gpvae_nuts_vae_run(void *stream, int dtype, int64_t C, const gpvae_nuts_run *r, int64_t n, int h0, int h1,
                                  int z, const void *dec_params, const void *y_obs, const int64_t *obs_idx,
                                  int64_t n_obs, double noise_scale) {
  GPVAE_REQUIRE(dtype == GPVAE_F32 || dtype == GPVAE_F64, 2, "gpvae_nuts_vae_run: bad dtype %d", dtype);
  GPVAE_REQUIRE(C >= 0 && C <= 2147483647, 3, "gpvae_nuts_vae_run: C out of range");
  if (C == 0) return 0;
  GPVAE_REQUIRE(r != nullptr, 4, "gpvae_nuts_vae_run: run description is NULL");
  GPVAE_REQUIRE(r->max_depth >= 1 && r->max_depth <= 20, 4, "gpvae_nuts_vae_run: max_depth out of range [1,20]");
  GPVAE_REQUIRE(r->num_warmup >= 0 && r->num_samples >= 0 && r->thinning >= 1, 4,
                "gpvae_nuts_vae_run: num_warmup / num_samples < 0 or thinning < 1");
  GPVAE_REQUIRE(r->n_ends >= 0 && r->n_ends <= 24, 4, "gpvae_nuts_vae_run: more than 24 mass-matrix windows");
  GPVAE_REQUIRE(r->row_len >= (int64_t)(z + 1) + 2 * r->max_depth + ((int64_t)1 << r->max_depth) - 1, 4,
                "gpvae_nuts_vae_run: row_len < D + 2 max_depth + 2^max_depth - 1");
  GPVAE_REQUIRE(r->step_size > 0 && r->target_accept > 0 && r->target_accept < 1, 4,
                "gpvae_nuts_vae_run: step_size <= 0 or target_accept outside (0, 1)");
  GPVAE_REQUIRE(r->q0 && r->accept_prob && r->leapfrogs && r->step_size_out && r->inverse_mass && r->divergent &&
                    (r->samples || r->num_samples == 0),
                4, "gpvae_nuts_vae_run: an input / output array is NULL");
  GPVAE_REQUIRE(n >= 1 && h0 >= 1 && h1 >= 1 && z >= 1, 5, "gpvae_nuts_vae_run: decoder dims must be positive");
  GPVAE_REQUIRE(dec_params && y_obs && obs_idx, 9, "gpvae_nuts_vae_run: NULL decoder / data pointer");
  GPVAE_REQUIRE(gpvae_nuts_vae_supported(dtype, h0, h1, z, n_obs), 12,
                "gpvae_nuts_vae_run: decoder head + gathered output columns do not fit 64 KB of LDS "
                "(gpvae_nuts_vae_supported)");
  GPVAE_REQUIRE(noise_scale > 0, 13, "gpvae_nuts_vae_run: noise_scale must be positive");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == GPVAE_F32) launch_nuts_run<float>(st, C, r, n, h0, h1, z, dec_params, y_obs, obs_idx, n_obs, noise_scale);
  else launch_nuts_run<double>(st, C, r, n, h0, h1, z, dec_params, y_obs, obs_idx, n_obs, noise_scale);
  GPVAE_CHECK_LAUNCH("gpvae_nuts_vae_run");
  return 0;
}
