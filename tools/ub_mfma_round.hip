// ub_mfma_round.hip — how do gfx950's MFMA instructions round their f32 / f64 accumulation?
// Every A entry of column k is a_k, every B entry of row k is b_k, every C entry is c: all outputs are
// c + sum_k a_k b_k as the matrix pipe computes it.  Printed against round-to-nearest-even and truncation.
//   hipcc --offload-arch=gfx950 -O2 -o tools/ub_mfma_round tools/ub_mfma_round.hip && tools/ub_mfma_round
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

// mode 0: v_mfma_f32_32x32x2_f32 (K = 2), 1: v_mfma_f32_16x16x4_f32 (K = 4), 2: v_mfma_f32_32x32x16_bf16 (K = 16),
// 3: v_mfma_f64_16x16x4_f64 (K = 4).  a, b: K values each (as float / double); reps: the instruction is applied
// reps times to the running accumulator.
__global__ void probe(int mode, const double *a, const double *b, double c, int reps, double *out) {
  const int lane = threadIdx.x;
  if (mode == 0) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = (float)c;
    const int k = lane >> 5;
    for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x2f32((float)a[k], (float)b[k], acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[5];
  } else if (mode == 1) {
    f32x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = (float)c;
    const int k = lane >> 4;
    for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f32_16x16x4f32((float)a[k], (float)b[k], acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[3];
  } else if (mode == 2) {
    f32x16 acc;
    for (int i = 0; i < 16; ++i) acc[i] = (float)c;
    bf16x8 va, vb;
    for (int e = 0; e < 8; ++e) {
      va[e] = (__bf16)(float)a[8 * (lane >> 5) + e];
      vb[e] = (__bf16)(float)b[8 * (lane >> 5) + e];
    }
    for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(va, vb, acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[5];
  } else {
    f64x4 acc;
    for (int i = 0; i < 4; ++i) acc[i] = c;
    const int k = lane >> 4;
    for (int r = 0; r < reps; ++r) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[k], b[k], acc, 0, 0, 0);
    if (lane == 0) out[0] = acc[0];
    if (lane == 37) out[1] = acc[3];
  }
}

static double bf16r(double v) {  // round to bf16 (RNE) on the host
  float f = (float)v; unsigned u; memcpy(&u, &f, 4);
  u += 0x7FFF + ((u >> 16) & 1); u &= 0xFFFF0000u; memcpy(&f, &u, 4); return f;
}
static double *da, *db, *dout;
static double run(int mode, const double *a, const double *b, int K, double c, int reps = 1) {
  double ha[16] = {0}, hb[16] = {0}, ho[2];
  for (int i = 0; i < K; ++i) { ha[i] = a[i]; hb[i] = b[i]; }
  hipMemcpy(da, ha, sizeof ha, hipMemcpyHostToDevice);
  hipMemcpy(db, hb, sizeof hb, hipMemcpyHostToDevice);
  probe<<<1, 64>>>(mode, da, db, c, reps, dout);
  hipMemcpy(ho, dout, sizeof ho, hipMemcpyDeviceToHost);
  if (ho[0] != ho[1]) printf("   (entries differ: %.17g %.17g)\n", ho[0], ho[1]);
  return ho[0];
}

int main() {
  hipMalloc(&da, 128); hipMalloc(&db, 128); hipMalloc(&dout, 16);
  const char *names[4] = {"f32 32x32x2", "f32 16x16x4", "bf16 32x32x16 (f32 acc)", "f64 16x16x4"};
  const int Ks[4] = {2, 4, 16, 4};
  for (int mode = 0; mode < 4; ++mode) {
    const int K = Ks[mode];
    const double u = mode == 3 ? ldexp(1.0, -52) : ldexp(1.0, -23);  // ulp of 1.0
    printf("== %s (ulp = %g)\n", names[mode], u);
    double a[16], b[16];
    // (1) one product, c = +-1, product = f * ulp: the result in ulps above c
    const double fr[] = {0.25, 0.5, 0.75, 1.0, 1.25, 1.5, 1.75};
    for (double sgn : {1.0, -1.0})
      for (double prodsgn : {1.0, -1.0}) {
        printf("  c = %+g, product %+g*f*ulp; result - c in ulp for f = .25 .5 .75 1 1.25 1.5 1.75:", sgn, prodsgn);
        for (double f : fr) {
          memset(a, 0, sizeof a); memset(b, 0, sizeof b);
          a[0] = prodsgn; b[0] = f * u;
          printf(" %+.2f", (run(mode, a, b, K, sgn) - sgn) / u);
        }
        printf("\n");
      }
    // (2) all K products f*ulp/K... each 0.3 ulp: fused sum = 0.3 K ulp
    for (double each : {0.3, 0.45}) {
      for (int i = 0; i < K; ++i) { a[i] = 1.0; b[i] = mode == 2 ? bf16r(each * u) : each * u; }
      printf("  c = 1, %d products of %.2f ulp (sum %.2f ulp): result - c = %+.2f ulp\n", K, each, each * K,
             (run(mode, a, b, K, 1.0) - 1.0) / u);
    }
    // (3) bias of a long chain: c = 1, reps instructions each adding K products of 0.37/K ulp... exact total known
    for (int reps : {100, 1000}) {
      for (int i = 0; i < K; ++i) { a[i] = 1.0; b[i] = 0.37 * u / K * (1 + 0.001 * i); if (mode == 2) b[i] = bf16r(b[i]); }
      double per = 0; for (int i = 0; i < K; ++i) per += b[i];
      double got = run(mode, a, b, K, 1.0, reps);
      printf("  chain: %d instructions adding %.4f ulp each to c = 1: got +%.1f ulp, exact +%.1f ulp\n", reps, per / u,
             (got - 1.0) / u, per * reps / u);
      for (int i = 0; i < K; ++i) b[i] = -b[i];
      got = run(mode, a, b, K, 1.0, reps);
      printf("  chain: %d instructions adding %.4f ulp each to c = 1: got %.1f ulp(1), exact %.1f ulp\n", reps, -per / u,
             (got - 1.0) / u, -per * reps / u);
    }
  }
  return 0;
}
