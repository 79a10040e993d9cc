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
 * these three functions with the SAME sequence of IEEE-754 binary64
 * add/mul/div/floor operations (no FMA contraction), which is bit-reproducible
 * on x86-64 and gfx950.  oracle/_ref links the unmodified reference against
 * these definitions (symbol interposition of sinf/cosf/sincosf/powf).
 *
 * Accuracy (tests/test_rng_detmath.py): <= 1 ulp(float) versus glibc on the
 * domains used: sin/cos |x| <= 1e4, pow x >= 0, y > 0.
 *
 * The same functions are restated for the device in
 * smallvcm_amd/csrc/detmath.h; tests compare the two bit-for-bit.
 */
#ifndef ORACLE_DETMATH_REF_H
#define ORACLE_DETMATH_REF_H
#include <stdint.h>
#include <string.h>
#include <math.h>

static inline double dmr_from_bits(uint64_t b) { double d; memcpy(&d, &b, 8); return d; }
static inline uint64_t dmr_to_bits(double d) { uint64_t b; memcpy(&b, &d, 8); return b; }

/* r = x - n*pi/2 (two-term Cody-Waite), n = round(x*2/pi); |r| <= pi/4 */
static inline double dmr_reduce(double x, int *n)
{
    const double q  = x * 0.63661977236758138;
    const double nf = floor(q + 0.5);
    *n = (int)nf;
    return (x - nf * 1.5707963267948966) - nf * 6.123233995736766e-17;
}

static inline double dmr_sin_poly(double r)
{
    const double r2 = r * r;
    double p = -1.0 / 1307674368000.0;
    p = p * r2 + 1.0 / 6227020800.0;
    p = p * r2 + -1.0 / 39916800.0;
    p = p * r2 + 1.0 / 362880.0;
    p = p * r2 + -1.0 / 5040.0;
    p = p * r2 + 1.0 / 120.0;
    p = p * r2 + -1.0 / 6.0;
    return r + r * (r2 * p);
}

static inline double dmr_cos_poly(double r)
{
    const double r2 = r * r;
    double p = 1.0 / 20922789888000.0;
    p = p * r2 + -1.0 / 87178291200.0;
    p = p * r2 + 1.0 / 479001600.0;
    p = p * r2 + -1.0 / 3628800.0;
    p = p * r2 + 1.0 / 40320.0;
    p = p * r2 + -1.0 / 720.0;
    p = p * r2 + 1.0 / 24.0;
    p = p * r2 + -0.5;
    return 1.0 + r2 * p;
}

static inline float dmr_sinf(float xf)
{
    int n;
    const double r = dmr_reduce((double)xf, &n);
    double v;
    switch (n & 3) {
    case 0:  v =  dmr_sin_poly(r); break;
    case 1:  v =  dmr_cos_poly(r); break;
    case 2:  v = -dmr_sin_poly(r); break;
    default: v = -dmr_cos_poly(r); break;
    }
    return (float)v;
}

static inline float dmr_cosf(float xf)
{
    int n;
    const double r = dmr_reduce((double)xf, &n);
    double v;
    switch (n & 3) {
    case 0:  v =  dmr_cos_poly(r); break;
    case 1:  v = -dmr_sin_poly(r); break;
    case 2:  v = -dmr_cos_poly(r); break;
    default: v =  dmr_sin_poly(r); break;
    }
    return (float)v;
}

/* x^y for x >= 0, y > 0 (the only uses on the hot path: Phong lobe,
 * bsdf.hxx:317, :445, utils.hxx:91, :111; radius schedule vertexcm.hxx:296).
 * x <= 0 -> 0, y == 0 -> 1. */
static inline float dmr_powf(float xf, float yf)
{
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;

    const double x = (double)xf;
    /* small positive integer exponent (the Phong exponent, 90 in the built-in
       scenes): binary exponentiation in binary64, least-significant bit first */
    if (yf >= 1.0f && yf <= 256.0f && yf == floorf(yf)) {
        unsigned n = (unsigned)yf;
        double b = x, r = 1.0;
        for (;;) {
            if (n & 1u) r = r * b;
            n >>= 1;
            if (n == 0u) break;
            b = b * b;
        }
        return (float)r;
    }
    const uint64_t bits = dmr_to_bits(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = dmr_from_bits((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }

    /* ln(m) = 2 s (1 + s^2/3 + s^4/5 + ...), s = (m-1)/(m+1), |s| <= 0.1716 */
    const double s  = (m - 1.0) / (m + 1.0);
    const double s2 = s * s;
    double p = 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double lnm   = 2.0 * s * p;
    const double log2x = (double)e + lnm * 1.4426950408889634;

    const double t = (double)yf * log2x;
    if (t >= 128.0)  return INFINITY;
    if (t < -160.0)  return 0.0f;

    const double kf = floor(t + 0.5);
    const int    k  = (int)kf;
    const double z  = (t - kf) * 0.6931471805599453;   /* |z| <= 0.3466 */
    double q = 1.0 / 6227020800.0;
    q = q * z + 1.0 / 479001600.0;
    q = q * z + 1.0 / 39916800.0;
    q = q * z + 1.0 / 3628800.0;
    q = q * z + 1.0 / 362880.0;
    q = q * z + 1.0 / 40320.0;
    q = q * z + 1.0 / 5040.0;
    q = q * z + 1.0 / 720.0;
    q = q * z + 1.0 / 120.0;
    q = q * z + 1.0 / 24.0;
    q = q * z + 1.0 / 6.0;
    q = q * z + 0.5;
    q = q * z + 1.0;
    q = q * z + 1.0;
    const double scale = dmr_from_bits((uint64_t)(k + 1023) << 52);   /* 2^k, k in [-160,128] */
    return (float)(q * scale);
}

#endif
