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
#include "common.hpp"
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

template <typename T>
__global__ __launch_bounds__(64) void nuts_kernel(int op, int64_t C, int64_t D, int max_depth, int depth, int leaf,
                                                  int ck_lo, int ck_hi, gpvae_nuts_buffers b) {
  const int64_t c = blockIdx.x;
  const int lane = threadIdx.x;
  const int64_t o = c * D;
  const T *minv = (const T *)b.minv + o;
#define V(name) ((T *)b.name + o)
#define S(name) (((T *)b.name)[c])
  if (op == GPVAE_NUTS_INIT) {
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
  } else if (op == GPVAE_NUTS_BEGIN) {
    const bool right = ((const T *)b.unif)[c * b.ldu + depth] < T(0.5);
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
      b.building[c] = b.active[c];
      b.right[c] = right ? 1 : 0;
      if (c == 0) {
        b.flags[0] = 0;
        b.flags[1] = 0;
      }
    }
  } else if (op == GPVAE_NUTS_PRE) {
    nuts_pre<T>(b, c, o, D, lane, minv);
  } else if (op == GPVAE_NUTS_POST || op == GPVAE_NUTS_POST_PRE) {
    // POST_PRE: the leaf's POST and the next leaf's PRE in one launch (PRE reads, lane by lane, what the same
    // lane's POST wrote: qe, pe, ge).  A chain that is not building skips POST and still runs PRE, as it would
    // under the separate ops.
    if (!b.building[c]) {  // wave-uniform
      if (op == GPVAE_NUTS_POST_PRE) nuts_pre<T>(b, c, o, D, lane, minv);
      return;
    }
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
    const T u = ((const T *)b.unif)[c * b.ldu + 2 * max_depth + ((int64_t)1 << depth) - 1 + leaf];
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
      } else {
        atomicOr(&b.flags[0], 1);
      }
    }
    if (op == GPVAE_NUTS_POST_PRE) nuts_pre<T>(b, c, o, D, lane, minv);
  } else if (op == GPVAE_NUTS_MERGE) {
    if (!b.active[c]) return;
    const bool ok = !b.sub_stop[c];
    if (!ok) {
      if (lane == 0) b.active[c] = 0;
      return;
    }
    const T logW = S(logW), sub_logW = S(sub_logW);
    const T u = ((const T *)b.unif)[c * b.ldu + max_depth + depth];
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
      if (!turn) atomicOr(&b.flags[1], 1);
    }
  } else if (op == GPVAE_NUTS_ACCEPT) {
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
#undef V
#undef S
}

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
