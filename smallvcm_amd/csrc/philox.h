// philox.h -- the counter-based RNG that replaces the reference's sequential
// `Rng` (src/rng.hxx:41-86), host+device.
//
// Philox4x32-10 (Salmon et al., SC'11).  Stream definition:
//   key     = (seed, localIteration)       localIteration = RunIteration calls
//                                          made so far on this renderer
//   counter = (pathIndex, kind, block, 0)  kind 0 = light sub-path, 1 = camera
//   float k of a path = word (k&3) of block (k>>2), (2*(word >> 9) + 1) * 2^-24 in (0,1)
// Draw order inside a path is the reference's (vertexcm.hxx:822-824, :576,
// :672-673, :944, :964).  One lane owns one path: 4 words are generated per
// 10-round call and kept in registers, so a path costs <= 18 Philox calls.
#ifndef SMALLVCM_AMD_PHILOX_H
#define SMALLVCM_AMD_PHILOX_H
#include "vcm_math.h"

namespace vcm {

VCM_HD uint32_t mulhi32(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
    return __umulhi(a, b);
#else
    return (uint32_t)(((uint64_t)a * (uint64_t)b) >> 32);
#endif
}

/* The stream is random-access (float k = word k&3 of block k>>2), so a path keeps only its position. */
struct PathRng {
    uint32_t key0, key1, path, kind;
    uint32_t k;               /* floats drawn so far */
};

VCM_HD void philox4x32_10(uint32_t c0, uint32_t c1, uint32_t c2, uint32_t c3, uint32_t k0, uint32_t k1,
                          uint32_t &o0, uint32_t &o1, uint32_t &o2, uint32_t &o3)
{
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int r = 0; r < 10; r++) {
        const uint32_t hi0 = mulhi32(0xD2511F53u, c0), lo0 = 0xD2511F53u * c0;
        const uint32_t hi1 = mulhi32(0xCD9E8D57u, c2), lo1 = 0xCD9E8D57u * c2;
        const uint32_t n0 = hi1 ^ c1 ^ k0;
        const uint32_t n2 = hi0 ^ c3 ^ k1;
        c0 = n0; c1 = lo1; c2 = n2; c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    o0 = c0; o1 = c1; o2 = c2; o3 = c3;
}

VCM_HD void rng_init(PathRng &r, uint32_t seed, uint32_t localIter, uint32_t path, uint32_t kind)
{
    r.key0 = seed; r.key1 = localIter; r.path = path; r.kind = kind; r.k = 0;
}

/* open interval (0,1): (2k+1) * 2^-24 with k = top 23 bits; exact 0 would make
   AreaLight::Emit return a zero pdf (lights.hxx:178-186) -> inf throughput -> NaN */
VCM_HD float rng_word_to_float(uint32_t w) { return (float)(((w >> 9) << 1) | 1u) * (1.0f / 16777216.0f); }

/* Floats k0 .. k0+n-1 of the path's stream, n <= 5, WITHOUT advancing it.  On the GPU what matters is where
 * Philox runs: a per-float "refill when the block is used up" puts a 100-instruction generator behind a
 * divergent branch at every draw (a wave step of the camera kernel executed it at 6 sites, each for a
 * quarter of its lanes).  Here the two blocks that can hold the n floats are generated unconditionally, by
 * all lanes, at one site per bounce step.  n floats from word offset o <= 3 reach word o+n-1 <= 7. */
VCM_HD void rng_peek(const PathRng &r, uint32_t k0, float *out, int n)
{
    uint32_t w[8];
    const uint32_t blk = k0 >> 2;
    philox4x32_10(r.path, r.kind, blk, 0u, r.key0, r.key1, w[0], w[1], w[2], w[3]);
    philox4x32_10(r.path, r.kind, blk + 1u, 0u, r.key0, r.key1, w[4], w[5], w[6], w[7]);
    const uint32_t o = k0 & 3u;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
#endif
    for (int j = 0; j < n; j++) {
        const uint32_t a = w[j], b = w[j + 1], c = w[j + 2], d = w[j + 3];   /* select, no dynamic indexing */
        out[j] = rng_word_to_float(o == 0u ? a : o == 1u ? b : o == 2u ? c : d);
    }
}
/* the same when the floats are known to lie in ONE block (k0 & 3) + n <= 4, e.g. the first draws of a path */
VCM_HD void rng_peek_block(const PathRng &r, uint32_t k0, float *out, int n)
{
    uint32_t w[4];
    philox4x32_10(r.path, r.kind, k0 >> 2, 0u, r.key0, r.key1, w[0], w[1], w[2], w[3]);
    const uint32_t o = k0 & 3u;
    for (int j = 0; j < n; j++) {
        const uint32_t i = o + (uint32_t)j;
        out[j] = rng_word_to_float(i == 0u ? w[0] : i == 1u ? w[1] : i == 2u ? w[2] : w[3]);
    }
}
/* one float, sequential interface (tests, spec kernels; the path code uses rng_peek) */
VCM_HD float rng_float(PathRng &r)
{
    float f;
    rng_peek_block(r, r.k, &f, 1);
    r.k++;
    return f;
}

} // namespace vcm
#endif
