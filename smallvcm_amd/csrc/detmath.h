// detmath.h -- bit-reproducible sinf / cosf / powf for the hot path,
// host+device.
//
// The reference calls libm (std::sin / std::cos / std::pow: src/utils.hxx:85-113,
// :119-160, :173-190, :212-230, src/bsdf.hxx:317, :445).  libm results are not
// portable bit-for-bit (glibc vs ROCm OCML differ in the last place for ~10 %
// of arguments), and a single flipped Russian-roulette / lobe / hit decision
// per few million paths already exceeds the per-pixel RMSE budget.  These
// three functions are therefore DEFINED as a fixed sequence of IEEE-754
// binary64 add / mul / div / floor operations (no FMA), which every conforming
// machine evaluates identically.  Results are correctly rounded to binary32
// for all but ~1e-9 of arguments (tests/test_rng_detmath.py).
// Domains: sin/cos |x| < ~1e4; pow x >= 0, y > 0 (x <= 0 -> 0, y == 0 -> 1).
// The CPU checker (oracle/detmath_ref.h) states the same definition.
#ifndef SMALLVCM_AMD_DETMATH_H
#define SMALLVCM_AMD_DETMATH_H
#include "vcm_math.h"

namespace vcm {

VCM_HD double bits2d(uint64_t b) { double d; __builtin_memcpy(&d, &b, 8); return d; }
VCM_HD uint64_t d2bits(double d) { uint64_t b; __builtin_memcpy(&b, &d, 8); return b; }

/* r = x - n*pi/2 (two-term Cody-Waite), n = round(x*2/pi) */
VCM_HD double dm_reduce(double x, int &n)
{
    const double q  = x * 0.63661977236758138;
    const double nf = floor(q + 0.5);
    n = (int)nf;
    return (x - nf * 1.5707963267948966) - nf * 6.123233995736766e-17;
}

VCM_HD double dm_sin_poly(double r)
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

VCM_HD double dm_cos_poly(double r)
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

/* sin and cos of the same argument share the reduction (every call site
   needs both: utils.hxx:97-101, :156-158, :180-183, :219-222) */
VCM_HD void dm_sincosf(float xf, float &s, float &c)
{
    int n;
    const double r = dm_reduce((double)xf, n);
    const double sp = dm_sin_poly(r);
    const double cp = dm_cos_poly(r);
    double sv, cv;
    switch (n & 3) {
    case 0:  sv =  sp; cv =  cp; break;
    case 1:  sv =  cp; cv = -sp; break;
    case 2:  sv = -sp; cv = -cp; break;
    default: sv = -cp; cv =  sp; break;
    }
    s = (float)sv;
    c = (float)cv;
}
VCM_HD float dm_sinf(float x) { float s, c; dm_sincosf(x, s, c); return s; }
VCM_HD float dm_cosf(float x) { float s, c; dm_sincosf(x, s, c); return c; }

VCM_HD float dm_powf(float xf, float yf)
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
    const uint64_t bits = d2bits(x);
    int e = (int)((bits >> 52) & 0x7ff) - 1023;
    double m = bits2d((bits & 0x000fffffffffffffull) | 0x3ff0000000000000ull);
    if (m > 1.4142135623730951) { m = m * 0.5; e = e + 1; }

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
    const double z  = (t - kf) * 0.6931471805599453;
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
    const double scale = bits2d((uint64_t)(k + 1023) << 52);
    return (float)(q * scale);
}

/* dm_powf for call sites where the exponent is a material constant (the Phong
 * exponent): if every active lane of the wave holds the same small integer
 * exponent, the binary exponentiation is driven by SCALAR control flow -- only
 * the ~log2(n)+popcount(n) binary64 multiplies remain as vector work, instead
 * of a per-lane loop with selects.  Same multiplication sequence as dm_powf,
 * hence the same bits; any other case falls back to dm_powf. */
VCM_HD float dm_powf_wave(float xf, float yf)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yf)));
    if (y0 >= 1.0f && y0 <= 256.0f && y0 == floorf(y0) && __all(yf == y0)) {   /* wave-uniform */
        unsigned n = (unsigned)y0;
        double b = (double)xf, r = 1.0;
        for (;;) {
            if (n & 1u) r = r * b;
            n >>= 1;
            if (n == 0u) break;
            b = b * b;
        }
        float res = (float)r;
        res = (xf == 1.0f) ? 1.0f : res;
        res = !(xf > 0.0f) ? 0.0f : res;
        return res;
    }
#endif
    return dm_powf(xf, yf);
}

} // namespace vcm
#endif
