// Counter-based rating sampler shared by host code and the HIP kernels.
//
// Replaces the reference's per-user cuRAND XORWOW state (initCurand + curand_uniform,
// sgd.cu:11-16,36): no state array, no init kernel, 0 bytes of RNG traffic per update.
// The draw for (user, iteration) is the first output word of the Philox4x32-10 block with
//   counter = {iteration_lo, iteration_hi, user_lo, user_hi},  key = {seed_lo, seed_hi}
// which is exactly rocRAND's device stream
//   rocrand_init(seed, /*subsequence=*/user, /*offset=*/4 * iteration, &st); rocrand(&st);
// (rocrand_philox4x32_10.h: seed -> key, subsequence -> counter.zw, offset/4 -> counter.xy).
// The map to a rating index is the reference's own (sgd.cu:36-37):
//   u = rocrand_uniform = 2^-32 + x * 2^-32 in (0,1];  y_i = ceil(u * n) - 1 + low.
// Scaling by 2^-32 is exact, so u is the same whether or not the compiler fuses the
// multiply-add, and host and device agree bit for bit.
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define CU2REC_HD __host__ __device__ __forceinline__
#else
#define CU2REC_HD inline
#endif

namespace cu2rec {

CU2REC_HD uint32_t mulhi32(uint32_t a, uint32_t b) {
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return static_cast<uint32_t>((static_cast<uint64_t>(a) * b) >> 32);
#endif
}

// First word of Philox4x32-10(counter, key). Only what feeds word 0 of the last round is kept.
CU2REC_HD uint32_t philox4x32_10_word0(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1) {
    constexpr uint32_t M0 = 0xD2511F53u, M1 = 0xCD9E8D57u, W0 = 0x9E3779B9u, W1 = 0xBB67AE85u;
#pragma unroll
    for (int round = 0; round < 9; ++round) {
        const uint32_t hi0 = mulhi32(M0, c0), lo0 = M0 * c0;
        const uint32_t hi1 = mulhi32(M1, c2), lo1 = M1 * c2;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += W0;
        k1 += W1;
    }
    return mulhi32(M1, c2) ^ c1 ^ k0;  // round 10, word 0 only
}

CU2REC_HD uint32_t sampler_draw(uint64_t seed, uint64_t user, uint64_t iteration) {
    return philox4x32_10_word0(static_cast<uint32_t>(iteration), static_cast<uint32_t>(iteration >> 32),
                               static_cast<uint32_t>(user), static_cast<uint32_t>(user >> 32),
                               static_cast<uint32_t>(seed), static_cast<uint32_t>(seed >> 32));
}

CU2REC_HD float sampler_uniform(uint32_t x) {
    const float inv = 2.3283064365386963e-10f;  // 2^-32
    return inv + static_cast<float>(x) * inv;
}

// sgd.cu:36-37 -- index in [low, high); requires high > low.
CU2REC_HD int sampler_index(uint64_t seed, uint64_t user, uint64_t iteration, int low, int high) {
    const float u = sampler_uniform(sampler_draw(seed, user, iteration));
#if defined(__HIP_DEVICE_COMPILE__)
    const float c = ceilf(u * static_cast<float>(high - low));
#else
    const float c = __builtin_ceilf(u * static_cast<float>(high - low));
#endif
    return static_cast<int>(c) - 1 + low;
}

}  // namespace cu2rec
