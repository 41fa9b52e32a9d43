// philox.h — the counter-based generator behind every random draw of the library (noise.hip; the Dropout keep-masks of the
// SmallRes tower, drawn by noise.hip's kernel or, inside a train step, by the step's first launch in smallres.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace alink {

// ---- Philox4x32-10 (Salmon et al., SC'11) -------------------------------------------------------
struct U4 { unsigned int x, y, z, w; };

__device__ __forceinline__ U4 philox4x32_10(U4 c, unsigned int k0, unsigned int k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        const unsigned long long p0 = (unsigned long long)0xD2511F53u * c.x;
        const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c.z;
        U4 n;
        n.x = (unsigned int)(p1 >> 32) ^ c.y ^ k0;
        n.y = (unsigned int)p1;
        n.z = (unsigned int)(p0 >> 32) ^ c.w ^ k1;
        n.w = (unsigned int)p0;
        c = n;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    return c;
}

// stream ids (counter word 3) keep the draws of different kernels disjoint under one seed
enum { ST_NORMAL = 0, ST_POISSON = 1, ST_SALTPEPPER = 2, ST_PERLIN = 3, ST_DROPOUT = 4 };

__device__ __forceinline__ U4 draw(unsigned long long seed, unsigned long long idx, unsigned int sub, unsigned int st) {
    U4 c{(unsigned int)idx, (unsigned int)(idx >> 32), sub, st};
    return philox4x32_10(c, (unsigned int)seed, (unsigned int)(seed >> 32));
}

// 24-bit uniform in the open interval (0, 1)
__device__ __forceinline__ float u01(unsigned int x) {
    return (float)(x >> 8) * 5.9604644775390625e-8f + 2.98023223876953125e-8f;
}

// Dropout keep-masks: mask[e] = 1 with probability `keep`: the element's 24-bit uniform (word e & 3 of Philox block e >> 2) < keep.
// One call = Philox block b (elements 4b .. 4b + 3), of which those in [first, first + count) are written to out[e - first].
__device__ __forceinline__ void keep_mask_block(unsigned char* __restrict__ out, long long count, float keep, unsigned long long seed,
                                                unsigned long long first, unsigned long long b) {
    const unsigned long long e0 = b * 4;
    if (e0 >= first + (unsigned long long)count) return;
    const U4 r = draw(seed, b, 0, ST_DROPOUT);
    const unsigned int w[4] = {r.x, r.y, r.z, r.w};
    for (int j = 0; j < 4; ++j) {
        const unsigned long long e = e0 + j;
        if (e >= first && e < first + (unsigned long long)count) out[e - first] = u01(w[j]) < keep ? 1 : 0;
    }
}

}  // namespace alink
