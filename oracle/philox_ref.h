/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * philox_ref.h -- CPU statement of the counter-based RNG that replaces
 * SmallVCM's sequential `Rng` (reference: src/rng.hxx:41-86; north_star:
 * "a counter-based RNG replacing rng.hxx").
 *
 * Algorithm: Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy
 * as 1, 2, 3", SC'11).  Pinned by the Random123 known-answer vectors in
 * tests/test_rng_detmath.py.
 *
 * Stream definition (shared with smallvcm_amd/csrc/philox.h):
 *   key     = (seed, localIteration)          localIteration = number of
 *                                             RunIteration calls made so far
 *                                             on this renderer
 *   counter = (pathIndex, kind, block, 0)     kind 0 = light path, 1 = camera
 *   float k of a path = word (k & 3) of block (k >> 2), mapped to the OPEN
 *                       interval (0,1) as (2*(word >> 9) + 1) * 2^-24
 * The reference draws floats in a fixed order inside a path
 * (vertexcm.hxx:822-824, :576, :672-673, :944, :964); this keeps that order
 * and makes paths independent of each other.
 */
#ifndef ORACLE_PHILOX_REF_H
#define ORACLE_PHILOX_REF_H
#include <stdint.h>

static inline void philox4x32_10_ref(const uint32_t ctr[4], const uint32_t key[2], uint32_t out[4])
{
    uint32_t c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3];
    uint32_t k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; r++) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c0;
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c2;
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c1 ^ k0;
        const uint32_t n1 = (uint32_t)p1;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c3 ^ k1;
        const uint32_t n3 = (uint32_t)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

static inline float philox_u32_to_float_ref(uint32_t w)
{
    /* open interval: (2k+1) * 2^-24, k = top 23 bits -> [2^-24, 1-2^-24], every value
       exact in binary32.  Exact 0 or 1 must not occur: the reference's own generators
       never produce them (rng.hxx:141 is (0,1]; generate_canonical hits 0 with
       probability 2^-64), and u = 0 makes AreaLight::Emit return emissionPdfW = 0
       (lights.hxx:178-186), i.e. an infinite path throughput and NaN pixels. */
    return (float)(((w >> 9) << 1) | 1u) * (1.0f / 16777216.0f);
}

/* Sequential view of one path's stream. */
typedef struct PathRngRef {
    uint32_t key[2];
    uint32_t path, kind;
    uint32_t k;       /* floats drawn so far */
    uint32_t blk[4];
} PathRngRef;

static inline void path_rng_init_ref(PathRngRef *r, uint32_t seed, uint32_t localIter,
                                     uint32_t path, uint32_t kind)
{
    r->key[0] = seed; r->key[1] = localIter;
    r->path = path; r->kind = kind; r->k = 0;
}

static inline float path_rng_float_ref(PathRngRef *r)
{
    if ((r->k & 3u) == 0u) {
        const uint32_t ctr[4] = { r->path, r->kind, r->k >> 2, 0u };
        philox4x32_10_ref(ctr, r->key, r->blk);
    }
    const float f = philox_u32_to_float_ref(r->blk[r->k & 3u]);
    r->k++;
    return f;
}

#endif
