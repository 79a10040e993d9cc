/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * detmath_ref.h -- CPU statement of the "deterministic libm" subset the hot
 * path needs: sinf, cosf, powf.  The reference calls libm for these
 * (std::sin/std::cos/std::pow in src/utils.hxx:85-103, :119-160, :173-190,
 * :212-230 and src/bsdf.hxx:290-318, :414-446).  glibc's and ROCm's OCML
 * results differ in the last bit for a fraction of inputs, and one flipped
 * Russian-roulette / lobe / hit decision per few million paths already breaks
 * the RMSE < 1e-4 bar (SURVEY.md section 7 "Hard parts").  So both sides compute
 * these three functions with the SAME sequence of IEEE-754 operations (no FMA
 * contraction), which is bit-reproducible on x86-64 and gfx950.  oracle/_ref links
 * the unmodified reference against these definitions (symbol interposition of
 * sinf/cosf/sincosf/powf).
 *
 * Definition (round 2; smallvcm_amd/csrc/detmath.h states it for the device, tests compare the two
 * bit for bit):
 *   sinf, cosf   binary32: r = x - n pi/2 by a four-term Cody-Waite reduction, minimax polynomials of degree
 *                7 / 8 on [-pi/4, pi/4]; <= 1.6 ulp for |x| <= 8.
 *   powf(x, y)   |y| = n + f: x^n by binary exponentiation in binary64, x^f in binary32 as exp2(f log2 x);
 *                <= 1.9 ulp for 0 < y < 1, integer exponents correctly rounded; x <= 0 -> 0, y == 0 -> 1.
 * (tests/test_rng_detmath.py measures both bounds.)
 */
#ifndef ORACLE_DETMATH_REF_H
#define ORACLE_DETMATH_REF_H
#include <stdint.h>
#include <string.h>
#include <math.h>

/* Two other definitions of the same three functions, for ONE purpose: to state what the deterministic definition
 * costs against the arithmetic the reference itself is built with (oracle/libm_tolerance.py):
 *   -DORACLE_LIBM_GLIBC   the host's libm, i.e. what std::sin / std::cos / std::pow of a float resolve to in the
 *                         reference (utils.hxx:85-117, :173-199, bsdf.hxx:290-318, :414-446);
 *   -DORACLE_LIBM_CR      round 1's definition: evaluated in binary64 and rounded once (correctly rounded but for
 *                         double-rounding cases of probability ~2^-29).
 * liboracle.so (the checker) is always built WITHOUT either. */
#if defined(ORACLE_LIBM_GLIBC) || defined(ORACLE_LIBM_CR)
#if defined(ORACLE_LIBM_GLIBC)
static inline float dmr_sinf(float x) { return sinf(x); }
static inline float dmr_cosf(float x) { return cosf(x); }
static inline float dmr_powf(float x, float y) { return powf(x, y); }
#else
static inline float dmr_sinf(float x) { return (float)sin((double)x); }
static inline float dmr_cosf(float x) { return (float)cos((double)x); }
static inline float dmr_powf(float x, float y) { return (float)pow((double)x, (double)y); }
#endif
static inline void dmr_sincosf(float x, float *s, float *c) { *s = dmr_sinf(x); *c = dmr_cosf(x); }
#else

static inline float dmr_from_bits32(uint32_t b) { float f; memcpy(&f, &b, 4); return f; }
static inline uint32_t dmr_to_bits32(float f) { uint32_t b; memcpy(&b, &f, 4); return b; }

/* *s = sin x, *c = cos x */
static inline void dmr_sincosf(float x, float *s, float *c)
{
    const float q  = x * 0.636619747f;      /* 2/pi */
    const float nf = floorf(q + 0.5f);
    const int   n  = (int)nf;
    /* pi/2 = 1.5703125 + 4.83751297e-4 + 7.54979013e-8 - 1.71512451e-15; the first two products are exact */
    float r = x - nf * 1.5703125f;
    r = r - nf * 4.83751297e-4f;
    r = r - nf * 7.54953362e-8f;
    r = r - nf * 2.56334407e-12f;
    const float z = r * r;
    float ps = -1.95094646e-4f;
    ps = ps * z + 8.33211839e-3f;
    ps = ps * z + -1.66666538e-1f;
    const float sinr = r + r * (z * ps);
    float pc = 2.44285529e-5f;
    pc = pc * z + -1.38872792e-3f;
    pc = pc * z + 4.16666456e-2f;
    const float cosr = (1.0f - 0.5f * z) + (z * z) * pc;
    switch (n & 3) {
    case 0:  *s =  sinr; *c =  cosr; break;
    case 1:  *s =  cosr; *c = -sinr; break;
    case 2:  *s = -sinr; *c = -cosr; break;
    default: *s = -cosr; *c =  sinr; break;
    }
}
static inline float dmr_sinf(float x) { float s, c; dmr_sincosf(x, &s, &c); return s; }
static inline float dmr_cosf(float x) { float s, c; dmr_sincosf(x, &s, &c); return c; }

/* x^f, x > 0, 0 < f < 1, binary32 */
static inline float dmr_pow_frac(float x, float f)
{
    uint32_t bits = dmr_to_bits32(x);
    int e = -127;
    if (bits < 0x00800000u) { bits = dmr_to_bits32(x * 16777216.f); e = -127 - 24; }
    e += (int)(bits >> 23);
    float m = dmr_from_bits32((bits & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }
    /* ln(m) = 2 s (1 + z (L0 + L1 z + L2 z^2 + L3 z^3)), s = (m-1)/(m+1), z = s^2 */
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 1.17941231e-1f;
    p = p * z + 1.42684832e-1f;
    p = p * z + 2.00001702e-1f;
    p = p * z + 3.33333313e-1f;
    const float s2 = s + s;
    const float lg = (s2 + s2 * (z * p)) * 1.44269502f;
    /* t = f * (e + lg) with f * e exact: f = fh + fl, 12 bits each */
    const float ef = (float)e;
    const float fh = dmr_from_bits32(dmr_to_bits32(f) & 0xfffff000u), fl = f - fh;
    const float a = fh * ef;
    const float b = fl * ef + f * lg;
    const float kf = floorf((a + b) + 0.5f);
    const float w = (a - kf) + b;
    float q = 1.54673908e-4f;
    q = q * w + 1.34004594e-3f;
    q = q * w + 9.61803552e-3f;
    q = q * w + 5.55032715e-2f;
    q = q * w + 2.40226507e-1f;
    q = q * w + 6.93147182e-1f;
    const float r = 1.0f + w * q;
    const int k = (int)kf, k1 = k >> 1, k2 = k - k1;
    return (r * dmr_from_bits32((uint32_t)(k1 + 127) << 23)) * dmr_from_bits32((uint32_t)(k2 + 127) << 23);
}

/* x^y (the uses on the hot path: Phong lobe bsdf.hxx:317, :445, utils.hxx:91, :111; radius schedule
 * vertexcm.hxx:296).  x <= 0 -> 0, y == 0 -> 1. */
static inline float dmr_powf(float xf, float yf)
{
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    const float ya = fabsf(yf);
    const float nf = floorf(ya);
    const float f = ya - nf;
    double p = 1.0;
    if (nf >= 1.0f) {   /* integer part: binary exponentiation in binary64, least-significant bit first */
        unsigned n = (nf < 4294967040.f) ? (unsigned)nf : 4294967040u;
        double b = (double)xf;
        for (;;) {
            if (n & 1u) p = p * b;
            n >>= 1;
            if (n == 0u) break;
            b = b * b;
        }
        if (f != 0.0f) p = p * (double)dmr_pow_frac(xf, f);
    } else {
        p = (double)dmr_pow_frac(xf, f);
    }
    if (yf < 0.0f) p = 1.0 / p;
    return (float)p;
}

#endif /* the deterministic definition */
#endif
