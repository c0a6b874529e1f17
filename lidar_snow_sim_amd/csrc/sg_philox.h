// sg_philox.h -- Philox4x32-10 (Salmon et al., SC'11): counter-based random numbers kept in registers, one independent
// stream per (key, counter) -- used by the on-device snowflake sampler (snowgpu_sampler.hip) and by the seeded RANSAC of the
// ground-plane estimator (snowgpu_plane.hip).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

__device__ __forceinline__ void philox_round(uint32_t (&c)[4], uint32_t (&k)[2])
{
    const uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u;
    const uint32_t hi0 = __umulhi(M0, c[0]), lo0 = M0 * c[0];
    const uint32_t hi1 = __umulhi(M1, c[2]), lo1 = M1 * c[2];
    const uint32_t n0 = hi1 ^ c[1] ^ k[0], n1 = lo1, n2 = hi0 ^ c[3] ^ k[1], n3 = lo0;
    c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
    k[0] += 0x9E3779B9u; k[1] += 0xBB67AE85u;
}

// the four 32-bit words of block (idx, group, tag) under key `seed`
__device__ __forceinline__ void philox_u32x4(uint64_t seed, uint64_t idx, uint32_t group, uint32_t tag, uint32_t (&out)[4])
{
    uint32_t c[4] = {(uint32_t)idx, (uint32_t)(idx >> 32), group, tag};
    uint32_t k[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
#pragma unroll
    for (int r = 0; r < 10; ++r) philox_round(c, k);
    out[0] = c[0]; out[1] = c[1]; out[2] = c[2]; out[3] = c[3];
}

// two uniform doubles in [0, 1) for (seed, candidate, draw group)
__device__ __forceinline__ void philox_u2(uint64_t seed, uint64_t idx, uint32_t group, double &u0, double &u1)
{
    uint32_t c[4];
    philox_u32x4(seed, idx, group, 0x534E4F57u /* "SNOW" */, c);
    const uint64_t a = ((uint64_t)c[0] << 32) | c[1], b = ((uint64_t)c[2] << 32) | c[3];
    u0 = (double)(a >> 11) * (1.0 / 9007199254740992.0);
    u1 = (double)(b >> 11) * (1.0 / 9007199254740992.0);
}
