// dev tool: the on-the-fly bf16 x 3 prototype kernels of tools/proto/elbo_b3.hpp alone — check against a double product on
// the host and time back to back.   build: hipcc -O3 -std=c++17 --offload-arch=gfx950 -I tools/proto -I gp_vae_amd/csrc -I include tools/proto/chk_elbo_b3.hip -o tools/proto/chk_elbo_b3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
#include "elbo_b3.hpp"
using namespace gpvae;

static double urand() { return (rand() & 0xFFFFFF) / double(0x1000000) * 2 - 1; }

int main(int argc, char **argv) {
  const int64_t B = argc > 1 ? atoll(argv[1]) : 4096, n = argc > 2 ? atoll(argv[2]) : 4096;
  const int h0 = argc > 3 ? atoi(argv[3]) : 50;
  const int ks = (B / 128) * (n / 256) >= 256 ? 256 : 128;
  const int nw = getenv("NW") ? atoi(getenv("NW")) : ((ks == 256 && B % 256 == 0) ? 8 : 4);
  const int64_t wgs = (B / (32 * nw)) * (n / ks);
  const int rbw = getenv("RBW") ? atoi(getenv("RBW")) : ((wgs >= 512 && (B / (32 * nw)) % 2 == 0) ? 2 : 1);
  const int ns = (int)(n / ks);
  std::vector<float> Y(B * n), W(n * h0);
  srand(1);
  const bool probe = argc > 4;
  for (auto &v : Y) v = (float)urand();
  for (auto &v : W) v = (float)(urand() * 0.05);
  if (probe) {  // Y[row][k] = k + 1000 row, W picks k = 7 c + 3: out[row][c] = 7 c + 3 + 1000 row
    for (int64_t r = 0; r < B; ++r) for (int64_t k = 0; k < n; ++k) Y[r * n + k] = (float)(k + 1000 * (r % 8));
    for (int64_t k = 0; k < n; ++k) for (int c = 0; c < h0; ++c) W[k * h0 + c] = (k == 7 * c + 3) ? 1.f : 0.f;
  }
  float *dY, *dW, *dS;
  hipMalloc(&dY, Y.size() * 4); hipMalloc(&dW, W.size() * 4); hipMalloc(&dS, (size_t)ns * B * h0 * 4);
  hipMemcpy(dY, Y.data(), Y.size() * 4, hipMemcpyHostToDevice);
  hipMemcpy(dW, W.data(), W.size() * 4, hipMemcpyHostToDevice);
  hipMemset(dS, 0xFF, (size_t)ns * B * h0 * 4);
  const dim3 grid((unsigned)(B / (32 * nw * rbw)), (unsigned)ns);
  auto launch = [&]() {
    if (ks == 256 && nw == 8) hipLaunchKernelGGL((a1_b3_kernel<256, 8>), grid, dim3(512), 0, 0, dY, n, dW, h0, dS, B * h0, rbw);
    else if (ks == 256) hipLaunchKernelGGL((a1_b3_kernel<256, 4>), grid, dim3(256), 0, 0, dY, n, dW, h0, dS, B * h0, rbw);
    else hipLaunchKernelGGL((a1_b3_kernel<128, 4>), grid, dim3(256), 0, 0, dY, n, dW, h0, dS, B * h0, rbw);
  };
  launch();
  if (hipDeviceSynchronize() != hipSuccess) { printf("launch failed\n"); return 1; }
  std::vector<float> S((size_t)ns * B * h0);
  hipMemcpy(S.data(), dS, S.size() * 4, hipMemcpyDeviceToHost);
  if (probe) {
    for (int row = 0; row < 3; ++row) {
      printf("row %d:", row);
      for (int c = 0; c < 12; ++c) { double got = 0; for (int sl = 0; sl < ns; ++sl) got += S[(size_t)sl * B * h0 + row * h0 + c]; printf(" %g", got); }
      printf("\n");
    }
  }
  double maxerr = 0, maxref = 0;
  for (int64_t row = 0; row < B; row += (B > 512 ? 97 : 1))
    for (int c = 0; c < h0; ++c) {
      double ref = 0, got = 0;
      for (int64_t k = 0; k < n; ++k) ref += (double)Y[row * n + k] * W[k * h0 + c];
      for (int s = 0; s < ns; ++s) got += S[(size_t)s * B * h0 + row * h0 + c];
      maxerr = fmax(maxerr, fabs(ref - got)); maxref = fmax(maxref, fabs(ref));
    }
  printf("a1_b3 nw=%d B=%ld n=%ld h0=%d ks=%d rbw=%d: max err %.3e (max |ref| %.3e)\n", nw, (long)B, (long)n, h0, ks, rbw, maxerr, maxref);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int i = 0; i < 5; ++i) launch();
  hipEventRecord(e0);
  for (int i = 0; i < 50; ++i) launch();
  hipEventRecord(e1); hipEventSynchronize(e1);
  float ms; hipEventElapsedTime(&ms, e0, e1);
  printf("  %.2f us per launch, %.2f TB/s of Y\n", ms / 50 * 1e3, B * n * 4.0 / (ms / 50 * 1e-3) / 1e12);
  return 0;
}
