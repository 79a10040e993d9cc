// detmath.h -- bit-reproducible sinf / cosf / powf for the hot path,
// host+device.
//
// The reference calls libm (std::sin / std::cos / std::pow: src/utils.hxx:85-113,
// :119-160, :173-190, :212-230, src/bsdf.hxx:317, :445).  libm results are not
// portable bit-for-bit (glibc vs ROCm OCML differ in the last place for ~10 %
// of arguments), and a single flipped Russian-roulette / lobe / hit decision
// per few million paths already exceeds the per-pixel RMSE budget.  These
// three functions are therefore DEFINED as a fixed sequence of IEEE-754
// operations (no FMA: the library is built with -ffp-contract=off), which every
// conforming machine evaluates identically; the CPU checker (oracle/detmath_ref.h)
// states the same definition and the unmodified reference is linked against it
// in the parity tests.
//
// Definition, round 2 (round 1 evaluated everything in binary64 with Taylor series to double accuracy: correctly
// rounded, and ~330 instructions per general powf, ~150 per sincosf, at every bounce of every path):
//   sinf/cosf  binary32 throughout: four-term Cody-Waite reduction by pi/2 (the first three products are exact),
//              degree-7 / degree-8 minimax polynomials on [-pi/4, pi/4].   <= 1.6 ulp for |x| <= 8 (the path
//              passes 2 pi u, u in (0,1), and the concentric-disc angle in (-pi/4, 7 pi/4)).
//   powf(x,y)  y = n + f, n = floor(y):  x^n by binary exponentiation in binary64 (one rounding at the end: the
//              Phong lobe, x^90, is correctly rounded as before), x^f in binary32 as exp2(f log2 x): log2 via the
//              atanh series in s = (m-1)/(m+1), the product f*e of the exponent part carried exactly (f split
//              12 + 12 bits), exp2 by a degree-6 minimax polynomial.   <= 1.9 ulp for 0 < y < 1 over the whole
//              binary32 range of x (tests/test_rng_detmath.py); x <= 0 -> 0, y == 0 -> 1, y < 0 -> 1 / x^|y|.
// ~45 instructions per sincosf, ~60 per fractional powf.
#ifndef SMALLVCM_AMD_DETMATH_H
#define SMALLVCM_AMD_DETMATH_H
#include "vcm_math.h"

namespace vcm {

/* sin and cos of the same argument share the reduction (every call site
   needs both: utils.hxx:97-101, :156-158, :180-183, :219-222) */
VCM_HD void dm_sincosf(float x, float &s, float &c)
{
    /* r = x - n pi/2, n = round(x 2/pi); pi/2 = P1 + P2 + P3 + P4, n P1, n P2 and n P3 exact for |n| < 2^12 */
    const float q  = x * 0.636619747f;                 /* 0x3f22f983 */
    const float nf = floorf(q + 0.5f);
    const int   n  = (int)nf;
    float r = x - nf * 1.5703125f;                      /* 0x3fc90000 */
    r = r - nf * 4.83751297e-4f;                        /* 0x39fda000 */
    r = r - nf * 7.54953362e-8f;                        /* 0x33a22000 */
    r = r - nf * 2.56334407e-12f;                      /* 0x2c34611a: keeps sin / cos near their zeros to an ulp */
    const float z = r * r;
    float ps = -1.95094646e-4f;                         /* 0xb94c9252 */
    ps = ps * z + 8.33211839e-3f;                       /* 0x3c088370 */
    ps = ps * z + -1.66666538e-1f;                      /* 0xbe2aaaa2 */
    const float sp = r + r * (z * ps);
    float pc = 2.44285529e-5f;                          /* 0x37ccebeb */
    pc = pc * z + -1.38872792e-3f;                      /* 0xbab605fa */
    pc = pc * z + 4.16666456e-2f;                       /* 0x3d2aaaa5 */
    const float cp = (1.0f - 0.5f * z) + (z * z) * pc;
    const int k = n & 3;
    const float sv = (k & 1) ? cp : sp;
    const float cv = (k & 1) ? sp : cp;
    s = (k & 2) ? -sv : sv;                             /* k: 0 sp, 1 cp, 2 -sp, 3 -cp */
    c = ((k + 1) & 2) ? -cv : cv;                       /* k: 0 cp, 1 -sp, 2 -cp, 3 sp */
}
VCM_HD float dm_sinf(float x) { float s, c; dm_sincosf(x, s, c); return s; }
VCM_HD float dm_cosf(float x) { float s, c; dm_sincosf(x, s, c); return c; }

/* x^n, n >= 1 an integer-valued float: binary exponentiation in binary64, least-significant bit first */
VCM_HD double dm_pow_int(float xf, float nf)
{
    unsigned n = (nf < 4294967040.f) ? (unsigned)nf : 4294967040u;
    double b = (double)xf, r = 1.0;
    for (;;) {
        if (n & 1u) r = r * b;
        n >>= 1;
        if (n == 0u) break;
        b = b * b;
    }
    return r;
}

/* x^f for x > 0 (finite), 0 < f < 1, binary32 throughout */
VCM_HD float dm_pow_frac(float x, float f)
{
    uint32_t bits = f2u(x);
    int e = -127;
    if (bits < 0x00800000u) { bits = f2u(x * 16777216.f); e = -127 - 24; }   /* subnormal x: scaled by 2^24 */
    e += (int)(bits >> 23);
    float m = u2f((bits & 0x007fffffu) | 0x3f800000u);                        /* [1, 2) */
    if (m > 1.41421354f) { m = m * 0.5f; e = e + 1; }                         /* [sqrt(1/2), sqrt 2) */
    /* ln m = 2 s + 2 s z (L0 + L1 z + L2 z^2 + L3 z^3), s = (m-1)/(m+1), z = s^2, |s| <= 0.1716 */
    const float s = (m - 1.0f) / (m + 1.0f);
    const float z = s * s;
    float p = 1.17941231e-1f;                           /* 0x3df18b2c */
    p = p * z + 1.42684832e-1f;                         /* 0x3e121bf9 */
    p = p * z + 2.00001702e-1f;                         /* 0x3e4ccd3f */
    p = p * z + 3.33333313e-1f;                         /* 0x3eaaaaaa */
    const float s2 = s + s;
    const float lg = (s2 + s2 * (z * p)) * 1.44269502f; /* log2 m; 0x3fb8aa3b */
    /* t = f (e + lg): f e is the large part and is carried exactly (f = fh + fl, 12 bits each; |e| < 2^8) */
    const float ef = (float)e;
    const float fh = u2f(f2u(f) & 0xfffff000u), fl = f - fh;
    const float a = fh * ef;
    const float b = fl * ef + f * lg;
    const float kf = floorf((a + b) + 0.5f);
    const float w = (a - kf) + b;                       /* [-0.5, 0.5] */
    float q = 1.54673908e-4f;                           /* 0x39222ff6 */
    q = q * w + 1.34004594e-3f;                         /* 0x3aafa47b */
    q = q * w + 9.61803552e-3f;                         /* 0x3c1d94f7 */
    q = q * w + 5.55032715e-2f;                         /* 0x3d635766 */
    q = q * w + 2.40226507e-1f;                         /* 0x3e75fdf0 */
    q = q * w + 6.93147182e-1f;                         /* 0x3f317218 */
    const float r = 1.0f + w * q;                       /* 2^w */
    /* times 2^k in two steps (k in [-150, 128]: either factor is a normal number) */
    const int k = (int)kf, k1 = k >> 1, k2 = k - k1;
    return (r * u2f((uint32_t)(k1 + 127) << 23)) * u2f((uint32_t)(k2 + 127) << 23);
}

VCM_HD float dm_powf(float xf, float yf)
{
    if (yf == 0.0f) return 1.0f;
    if (!(xf > 0.0f)) return 0.0f;
    if (xf == 1.0f) return 1.0f;
    const float ya = fabsf(yf);
    const float nf = floorf(ya);
    const float f = ya - nf;                            /* exact */
    /* ONE call site of each part (the function is inlined wherever a Phong lobe is evaluated); a factor 1.0 and a
       binary32 value widened to binary64 and back are exact */
    double p = (nf >= 1.0f) ? dm_pow_int(xf, nf) : 1.0;
    if (f != 0.0f) p = p * (double)dm_pow_frac(xf, f);
    if (yf < 0.0f) p = 1.0 / p;                         /* not on the path (the exponents there are positive) */
    return (float)p;
}

/* dm_powf for call sites where the exponent is a material constant (the Phong
 * exponent): if every active lane of the wave holds the same integer
 * exponent, the binary exponentiation is driven by SCALAR control flow -- only
 * the ~log2(n)+popcount(n) binary64 multiplies remain as vector work, instead
 * of a per-lane loop with selects.  Same multiplication sequence as dm_powf,
 * hence the same bits; any other case falls back to dm_powf. */
VCM_HD float dm_powf_wave(float xf, float yf)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yf)));
    if (y0 >= 1.0f && y0 <= 65536.0f && y0 == floorf(y0) && __builtin_amdgcn_ballot_w64(!(yf == y0)) == 0ull) {   /* wave-uniform */
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
