// philox.hpp — counter-based standard normals shared by gpvae_randn (draw.hip) and the bf16 x 3
// producers (b3.hip): the same (seed, stream_id, element index) gives the same float everywhere.
#pragma once
#include "common.hpp"

namespace gpvae {

// ---- Philox4x32-10 + Box-Muller -------------------------------------------------------------------
// counter = (i_lo, i_hi, stream_lo, stream_hi) with i = element index / 4, key = (seed_lo, seed_hi);
// the 4 output words give 4 normals: (u0,u1) -> (r cos, r sin), (u2,u3) -> (r cos, r sin),
// u = (word + 0.5) * 2^-32 in (0,1] (formed in T), r = sqrt(-2 ln u_even), angle = 2 pi u_odd.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
    const uint32_t n1 = (uint32_t)p1;
    const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    const uint32_t n3 = (uint32_t)p0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
}

// the four normals of counter q (elements 4q .. 4q+3 of the stream)
template <typename T>
__device__ __forceinline__ void philox_normals4(int64_t q, uint64_t seed, uint64_t stream_id, T (&z)[4]) {
  uint32_t c[4] = {(uint32_t)q, (uint32_t)((uint64_t)q >> 32), (uint32_t)stream_id,
                     (uint32_t)(stream_id >> 32)};
    philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      if constexpr (sizeof(T) == 4) {
        const float u1 = ((float)c[2 * h] + 0.5f) * 2.3283064365386963e-10f;
        const float u2 = ((float)c[2 * h + 1] + 0.5f) * 2.3283064365386963e-10f;
        // hardware transcendentals: v_log_f32 (log2), v_sqrt_f32, v_sin_f32 / v_cos_f32 (argument in
        // revolutions = u2 itself).  With libm's logf / sqrtf / sincospif the kernel was bound by
        // the vector ALU (45 instructions per normal); absolute error of a normal <= 1e-5.
        const float rad = __builtin_amdgcn_sqrtf(-1.3862943611198906f * __builtin_amdgcn_logf(u1));
        z[2 * h] = rad * __builtin_amdgcn_cosf(u2);
        z[2 * h + 1] = rad * __builtin_amdgcn_sinf(u2);
      } else {
        const double u1 = ((double)c[2 * h] + 0.5) * (1.0 / 4294967296.0);
        const double u2 = ((double)c[2 * h + 1] + 0.5) * (1.0 / 4294967296.0);
        const double rad = sqrt(-2.0 * log(u1));
        double sn, cs;
        sincospi(2.0 * u2, &sn, &cs);
        z[2 * h] = rad * cs;
        z[2 * h + 1] = rad * sn;
      }
    }
}

}  // namespace gpvae
