// api.hip — library-wide pieces of the C ABI (error string, version).
#include "common.hpp"

namespace gpvae {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
}  // namespace gpvae

extern "C" int gpvae_abi_version(void) { return GPVAE_ABI_VERSION; }
extern "C" const char *gpvae_last_error(void) { return gpvae::g_err; }

#ifdef GPVAE_DEV
// ---- developer probe (make DEV=1 only; not exported by the product build, not in include/gpvae.h):
// shader clock under the caller's load
// One wave spins on dependent FMAs for `iters` iterations and reports
// (delta s_memtime [shader cycles], delta s_memrealtime [100 MHz ticks]) into out[0..1].
__global__ void gpvae_dev_clock_kernel(int iters, unsigned long long *out) {
  float x = (float)threadIdx.x;
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
  for (int i = 0; i < iters; ++i) x = fmaf(x, 1.0000001f, 1e-7f);
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
  if (threadIdx.x == 0) {
    out[0] = t1 - t0;
    out[1] = r1 - r0;
  }
  if (x == 12345.678f) out[2] = 1;  // keep the loop alive
}
extern "C" int gpvae_dev_clock(void *stream, int iters, void *out3_u64) {
  hipLaunchKernelGGL(gpvae_dev_clock_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, iters,
                     (unsigned long long *)out3_u64);
  return hipGetLastError() == hipSuccess ? 0 : -1;
}
#endif  // GPVAE_DEV
