// Counter-based random numbers shared by the kernels of libmarl_env.so.
#ifndef MARL_RNG_H
#define MARL_RNG_H
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace marl_rng {
// Philox4x32-10 (Salmon et al., SC'11).  Counter = (env index, draw index, stream id, episode/step),
// key = seed.  Same values whatever the launch geometry.
__device__ __forceinline__ void philox4x32(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3,
                                           uint32_t k0, uint32_t k1, uint32_t out[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}
// Philox2x32-10 (same paper): 64-bit counter, 32-bit key, two words out - half the multiplications of the 4x32
// form (32-bit integer multiplies run at a quarter of the vector rate), for consumers that need 64 bits per draw.
__device__ __forceinline__ void philox2x32(uint32_t c0, uint32_t c1, uint32_t k, uint32_t out[2]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const uint64_t p = (uint64_t)0xD256D193u * c0;
        c0 = (uint32_t)(p >> 32) ^ k ^ c1;
        c1 = (uint32_t)p;
        k += 0x9E3779B9u;
    }
    out[0] = c0; out[1] = c1;
}
__device__ __forceinline__ float u01f(uint32_t r) {                  // (0,1), 24 bits, float32
    return ((float)(r >> 8) + 0.5f) * (1.0f / 16777216.0f);
}
__device__ __forceinline__ double u01(uint32_t hi, uint32_t lo) {   // (0,1), 53 bits
    const uint64_t v = ((uint64_t)hi << 21) ^ (uint64_t)(lo >> 11);
    return ((double)(v & ((1ull << 53) - 1)) + 0.5) * (1.0 / 9007199254740992.0);
}
// Stream ids, mixed into the second counter word: one per consumer, so that no two draws share a counter.
enum { STREAM_NOISE = 1, STREAM_RESET_ENEMY = 2, STREAM_RESET_JAMMER = 3, STREAM_RESET_ENV = 4, STREAM_POLICY = 5 };
}  // namespace marl_rng
#endif
