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
 * Definition (round 4; smallvcm_amd/csrc/detmath.h states it for the device, tests compare the two bit for bit):
 * the reference's OWN libm, restated -- glibc 2.35 (Ubuntu GLIBC 2.35-0ubuntu3.11, x86-64), the single-precision
 * sinf / cosf / sincosf / powf of sysdeps/ieee754/flt-32 (ARM Optimized Routines; MIT) in the FMA multiarch variants
 * the dynamic linker picks on this host: binary64 evaluation, every multiply-add fused, one rounding to binary32.
 * libm is a third-party dependency absent from /root/reference; the restatement is PINNED against the libm of this
 * image: oracle/libm_check.c, all 2^32 arguments of sinf / cosf / sincosf, 19 x 2^32 argument pairs of powf, no
 * difference (profiles/archive/r06_libm_check.txt).  One deviation (dmr_powf): integer exponents 1..65536 are the correctly
 * rounded power (binary exponentiation in binary64), which glibc's powf misses by one ulp for 0.17 % of the arguments.
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
static inline double dmr_from_bits64(uint64_t b) { double f; memcpy(&f, &b, 8); return f; }
static inline uint64_t dmr_to_bits64(double f) { uint64_t b; memcpy(&b, &f, 8); return b; }
#define DMR_FMA(a, b, c) __builtin_fma((a), (b), (c))   /* fused: one rounding */
/* (constants in decimal with 17 significant digits = the exact binary64 values: this header is also compiled as C++0x,
   which has no hexadecimal floating literals; smallvcm_amd/csrc/detmath.h shows them in hexadecimal, and
   tests/test_rng_detmath.py compares the two definitions bit for bit) */

/* 96 bits of 4/pi, indexed by the exponent of |y| (glibc __inv_pio4) */
static const uint32_t dmr_inv_pio4[24] = {
    0xa2, 0xa2f9, 0xa2f983, 0xa2f9836e, 0xf9836e4e, 0x836e4e44, 0x6e4e4415, 0x4e441529, 0x441529fc, 0x1529fc27,
    0x29fc2757, 0xfc2757d1, 0x2757d1f5, 0x57d1f534, 0xd1f534dd, 0xf534ddc0, 0x34ddc0db, 0xddc0db62, 0xc0db6295,
    0xdb629599, 0x6295993c, 0x95993c43, 0x993c4390, 0x3c439041 };

/* *s = sin y, *c = cos y: glibc 2.35 sincosf / sinf / cosf, FMA variants (s_sincosf.h) */
static inline void dmr_sincosf(float y, float *s, float *c)
{
    const uint32_t xi = dmr_to_bits32(y);
    const uint32_t top = (xi >> 20) & 0x7ffu;
    double x = (double)y;
    int n = 0, m = 0;
    if (top < 0x42fu) {            /* |y| < 120: reduce_fast (covers the |y| < pi/4 branch: n = 0, x = y) */
        const double r = x * 10680707.430881744;
        n = ((int32_t)r + 0x800000) >> 24;
        x = DMR_FMA(-(double)n, 1.5707963267948966, x);
        m = n;
    } else if (top < 0x7f8u) {     /* reduce_large */
        const uint32_t *arr = &dmr_inv_pio4[(xi >> 26) & 15];
        const int shift = (xi >> 23) & 7;
        uint32_t mm = (xi & 0xffffff) | 0x800000;
        mm <<= shift;
        uint64_t res0 = (uint32_t)(mm * arr[0]);
        const uint64_t res1 = (uint64_t)mm * arr[4];
        const uint64_t res2 = (uint64_t)mm * arr[8];
        res0 = (res2 >> 32) | (res0 << 32);
        res0 += res1;
        const uint64_t nn = (res0 + (1ULL << 61)) >> 62;
        res0 -= nn << 62;
        x = (double)(int64_t)res0 * 3.4061215800865545e-19;
        n = (int)nn;
        m = n + (int)(xi >> 31);
    } else { *s = *c = y - y; return; }
    const double x2 = x * x;
    const double x3 = x * x2, s1 = DMR_FMA(x2, -0.00019517298981385725, 0.0083321781461388536), x5 = x3 * x2;
    const double sa = DMR_FMA(x3, -0.16666654943701084, x);
    const float sinp = (float)DMR_FMA(x5, s1, sa);
    const double x4 = x2 * x2, c2 = DMR_FMA(x2, 2.4390450703564542e-05, -0.0013886763794376041);
    const double c1 = DMR_FMA(x2, -0.49999999725108224, 1.0);
    const double x6 = x4 * x2, ca = DMR_FMA(x4, 0.041666623324344516, c1);
    const float cosp = (float)DMR_FMA(x6, c2, ca);
    const float so = ((m + 1) & 2) ? -sinp : sinp;   /* sign[m & 3] = {1, -1, -1, 1} */
    const float ce = (m & 2) ? -cosp : cosp;         /* __sincosf_table[1] = the negated even polynomial */
    if (n & 1) { *s = ce; *c = so; } else { *s = so; *c = ce; }
    if (top < 0x398u) { *s = y; *c = 1.0f; }         /* |y| < 2^-12 */
}
static inline float dmr_sinf(float x) { float s, c; dmr_sincosf(x, &s, &c); return s; }
static inline float dmr_cosf(float x) { float s, c; dmr_sincosf(x, &s, &c); return c; }

/* glibc __powf_log2_data.tab {invc, logc} and __exp2f_data.tab */
static const double dmr_log2_tab[32] = {
    1.3989071621465281, -0.48430022186289673, 1.3403141896637998, -0.42257122959194704, 1.286432210124115,
    -0.36337543476735562, 1.2367150214269895, -0.30651309567405577, 1.1906977166711752, -0.25180720160537634,
    1.1479821020556429, -0.19910014943794563, 1.1082251448272158, -0.14825100623281615, 1.0711297413057381,
    -0.099133238073183916, 1.0364372789772831, -0.051632812977629436, 1.0, 0.0,
    0.9492859795739057, 0.075085319379430041, 0.89510494286090037, 0.15987125980713107, 0.84768216203511026,
    0.23840466643176811, 0.80503148516920009, 0.31288288605863257, 0.7664671008843108, 0.38370422656453185,
    0.73142860331632797, 0.45121104893581498 };
static const uint64_t dmr_exp2_tab[32] = {
    0x3ff0000000000000, 0x3fefd9b0d3158574, 0x3fefb5586cf9890f, 0x3fef9301d0125b51, 0x3fef72b83c7d517b,
    0x3fef54873168b9aa, 0x3fef387a6e756238, 0x3fef1e9df51fdee1, 0x3fef06fe0a31b715, 0x3feef1a7373aa9cb,
    0x3feedea64c123422, 0x3feece086061892d, 0x3feebfdad5362a27, 0x3feeb42b569d4f82, 0x3feeab07dd485429,
    0x3feea47eb03a5585, 0x3feea09e667f3bcd, 0x3fee9f75e8ec5f74, 0x3feea11473eb0187, 0x3feea589994cce13,
    0x3feeace5422aa0db, 0x3feeb737b0cdc5e5, 0x3feec49182a3f090, 0x3feed503b23e255d, 0x3feee89f995ad3ad,
    0x3feeff76f2fb5e47, 0x3fef199bdd85529c, 0x3fef3720dcef9069, 0x3fef5818dcfba487, 0x3fef7c97337b9b5f,
    0x3fefa4afa2a490da, 0x3fefd0765b6e4540 };

static inline int dmr_checkint(uint32_t iy)   /* 0: not an integer, 1: odd, 2: even */
{
    const int e = iy >> 23 & 0xff;
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
static inline int dmr_zeroinfnan(uint32_t ix) { return 2 * ix - 1 >= 2u * 0x7f800000 - 1; }

/* glibc 2.35 powf, FMA variant (e_powf.c), every case */
static inline float dmr_powf_glibc(float x, float y)
{
    uint32_t sign_bias = 0;
    uint32_t ix = dmr_to_bits32(x), iy = dmr_to_bits32(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || dmr_zeroinfnan(iy)) {
        if (dmr_zeroinfnan(iy)) {
            if (2 * iy == 0) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2 * ix > 2u * 0x7f800000u || 2 * iy > 2u * 0x7f800000u) return x + y;
            if (2 * ix == 2 * 0x3f800000u) return 1.0f;
            if ((2 * ix < 2 * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;
            return y * y;
        }
        if (dmr_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && dmr_checkint(iy) == 1) { x2 = -x2; sign_bias = 1; }
            if (2 * ix == 0 && (iy & 0x80000000u)) return sign_bias ? -INFINITY : INFINITY;
            return (iy & 0x80000000u) ? 1 / x2 : x2;
        }
        if (ix & 0x80000000u) {
            const int yint = dmr_checkint(iy);
            if (yint == 0) return dmr_from_bits32(0x7fc00000u);
            if (yint == 1) sign_bias = 0x10000u;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {
            ix = dmr_to_bits32(dmr_from_bits32(ix) * 8388608.0f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (tmp >> 19) & 15;
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    const double invc = dmr_log2_tab[2 * i], logc = dmr_log2_tab[2 * i + 1];
    const double z = (double)dmr_from_bits32(iz);
    const double r = DMR_FMA(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double a = DMR_FMA(0.28845758110921399, r, -0.36092606229713164);
    const double p = DMR_FMA(0.48089848147257702, r, -0.72134746750062906);
    const double r2 = r * r;
    double q = DMR_FMA(r, 1.4426950408774342, y0);
    const double r4 = r2 * r2;
    q = DMR_FMA(r2, p, q);
    const double logx = DMR_FMA(a, r4, q);
    const double ylogx = (double)y * logx;
    if ((dmr_to_bits64(ylogx) >> 47 & 0xffff) >= 0x80bf) {
        if (ylogx > 127.99999995700433) return sign_bias ? -INFINITY : INFINITY;
        if (ylogx <= -150.0) return sign_bias ? -0.0f : 0.0f;
        if (ylogx < -149.0) return sign_bias ? -dmr_from_bits32(1u) : dmr_from_bits32(1u);
    }
    double kd = ylogx + 211106232532992.0;
    const uint64_t ki = dmr_to_bits64(kd);
    kd -= 211106232532992.0;
    const double rr = ylogx - kd;
    uint64_t t = dmr_exp2_tab[ki & 31];
    t += (ki + sign_bias) << 47;
    const double sc = dmr_from_bits64(t);
    const double zz = DMR_FMA(0.055503615593415351, rr, 0.2402284522445722);
    const double rr2 = rr * rr;
    double yy = DMR_FMA(0.69314718069162029, rr, 1.0);
    yy = DMR_FMA(zz, rr2, yy);
    yy = yy * sc;
    return (float)yy;
}

/* x^y (the uses on the hot path: Phong lobe bsdf.hxx:317, :445, utils.hxx:91, :111; radius schedule
 * vertexcm.hxx:296): glibc's powf, EXCEPT that an integer exponent 1 <= y <= 65536 -- the Phong lobe's x^90 -- is the
 * correctly rounded power by binary exponentiation in binary64 (the product's one deviation, detmath.h). */
static inline float dmr_powf(float xf, float yf)
{
    if (yf >= 1.0f && yf <= 65536.0f && yf == floorf(yf)) {
        unsigned n = (unsigned)yf;
        double b = (double)xf, p = 1.0;
        for (;;) {
            if (n & 1u) p = p * b;
            n >>= 1;
            if (n == 0u) break;
            b = b * b;
        }
        return (float)p;
    }
    return dmr_powf_glibc(xf, yf);
}

#endif /* the deterministic definition */
#endif
