// detmath.h -- sinf / cosf / powf of the hot path as FIXED sequences of IEEE-754 operations, host+device.
//
// The reference calls the host's libm (std::sin / std::cos / std::pow of a float: src/utils.hxx:85-113, :119-160,
// :173-190, :212-230, src/bsdf.hxx:317, :445, src/vertexcm.hxx:296).  libm is a third-party dependency of the
// reference and its results are not portable bit for bit (glibc and ROCm's OCML differ in the last place for ~10 %
// of the arguments), while one flipped Russian-roulette / lobe / hit decision per few million paths already exceeds
// the per-pixel RMSE budget.  So the three functions are DEFINED here, and since round 4 the definition is the
// reference's own arithmetic:
//
//   glibc 2.35 (the libm of the image the reference is built and timed in: Ubuntu GLIBC 2.35-0ubuntu3.11, x86-64),
//   sysdeps/ieee754/flt-32/{s_sinf.c, s_cosf.c, s_sincosf.c, s_sincosf.h, e_powf.c}, i.e. the ARM Optimized Routines
//   single-precision functions (S. Nagy; MIT licence) in the multiarch variants the dynamic linker selects on every
//   x86-64 CPU with FMA (__sinf_fma, __cosf_fma, __sincosf_fma, __powf_fma): evaluation in binary64 with every
//   multiply-add FUSED, one rounding to binary32 at the end.
//
//   sinf / cosf   x = y - n pi/2 with n = round(y 2/pi) from ONE binary64 product scaled by 2^24 (|y| < 120; beyond
//                 that 96 bits of 4/pi in integer arithmetic), then a degree-7 odd / degree-8 even polynomial in
//                 binary64.  The published algorithm; constants = the published tables.
//   powf(x, y)    log2 x = k + log2 c_i + log2(z / c_i) from a 16-entry table {1/c, log2 c} and a degree-5 polynomial,
//                 t = y log2 x, 2^t from a 32-entry table of 2^(i/32) and a degree-3 polynomial; all of glibc's
//                 special cases (zeros, infinities, NaNs, negative x with integer y, overflow, underflow).
//
// PINNED against that libm itself: oracle/libm_check.c compares the restatement with the host's sinf, cosf, sincosf
// over ALL 2^32 arguments and with powf over 19 x 2^32 argument pairs (every x for the exponents the path uses and a
// dozen others, 2^32 random bit patterns, 2^32 pairs spanning overflow to underflow): no difference
// (profiles/archive/r06_libm_check.txt); tests/test_rng_detmath.py keeps a sampled version under test.
//
// ONE deliberate deviation, for the GPU's sake: a POSITIVE INTEGER exponent n <= 65536 (the Phong lobe: pow(x, 90),
// bsdf.hxx:317, :445, utils.hxx:111 -- evaluated once per accepted photon of the merge, 2 x 10^8 times per iteration)
// is computed by binary exponentiation in binary64, rounded once, instead of glibc's table walk: nine multiplies
// under scalar loop control (dm_powf_wave) against ~17 fused operations and three per-lane table reads.  The result is
// the correctly rounded power; glibc's own result differs from it in 0.17 % of the arguments by one unit in the last
// place (measured for x^90 over (1e-3, 1]), and only VALUES depend on it (a pdf and a BSDF value), no decision --
// oracle/libm_tolerance.py prices it: RMSE ~1e-8 against the reference built with its stock libm.
//
// Everything here is explicit: dm_fma is a fused multiply-add (v_fma_f64 on gfx950, fma() / vfmadd on the host), all
// other operations are plain IEEE operations (the library and the checker are built with -ffp-contract=off).
#ifndef SMALLVCM_AMD_DETMATH_H
#define SMALLVCM_AMD_DETMATH_H
#include "vcm_math.h"

namespace vcm {

VCM_HD double dm_fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
VCM_HD double u2d(unsigned long long u) { double d; __builtin_memcpy(&d, &u, 8); return d; }
VCM_HD unsigned long long d2u(double d) { unsigned long long u; __builtin_memcpy(&u, &d, 8); return u; }

/* ---- sinf, cosf (glibc s_sincosf.h: reduce_fast, reduce_large, sinf_poly; __sincosf_table) ---- */

/* |y| >= 120: x = |y| mod pi/2 from 96 bits of 4/pi (__inv_pio4, indexed by the exponent), in integer arithmetic */
VCM_HD double dm_reduce_large(uint32_t xi, int &n)
{
    static const uint32_t inv_pio4[24] = {
        0xa2u, 0xa2f9u, 0xa2f983u, 0xa2f9836eu, 0xf9836e4eu, 0x836e4e44u, 0x6e4e4415u, 0x4e441529u, 0x441529fcu,
        0x1529fc27u, 0x29fc2757u, 0xfc2757d1u, 0x2757d1f5u, 0x57d1f534u, 0xd1f534ddu, 0xf534ddc0u, 0x34ddc0dbu,
        0xddc0db62u, 0xc0db6295u, 0xdb629599u, 0x6295993cu, 0x95993c43u, 0x993c4390u, 0x3c439041u };
    const uint32_t *arr = &inv_pio4[(xi >> 26) & 15u];
    const int shift = (int)((xi >> 23) & 7u);
    uint32_t m = (xi & 0xffffffu) | 0x800000u;
    m <<= shift;
    unsigned long long res0 = (uint32_t)(m * arr[0]);
    const unsigned long long res1 = (unsigned long long)m * arr[4];
    const unsigned long long res2 = (unsigned long long)m * arr[8];
    res0 = (res2 >> 32) | (res0 << 32);
    res0 += res1;
    const unsigned long long nn = (res0 + (1ull << 61)) >> 62;
    res0 -= nn << 62;
    n = (int)nn;
    return (double)(long long)res0 * 0x1.921fb54442d18p-62;
}

/* sin and cos of the same argument share the reduction and x^2 (every call site needs both: utils.hxx:97-101,
   :156-158, :180-183, :219-222; glibc's sincosf returns the two values sinf and cosf return) */
VCM_HD void dm_sincosf(float y, float &s, float &c)
{
    const uint32_t xi = f2u(y);
    const uint32_t top = (xi >> 20) & 0x7ffu;   /* abstop12 */
    double x = (double)y;
    int n = 0;        /* which polynomial: sin(y) is the odd one for even n */
    int m = 0;        /* which signs: n plus the sign bit of y on the large path */
    if (top < 0x42fu) {
        /* |y| < 120 (reduce_fast): n = round(y 2/pi), the quotient scaled by 2^24 and truncated; below pi/4 this gives
           n = 0 and x = y, which is what glibc's first branch computes */
        const double r = x * 0x1.45f306dc9c883p+23;
        n = ((int32_t)r + 0x800000) >> 24;
        x = dm_fma(-(double)n, 0x1.921fb54442d18p+0, x);
        m = n;
    } else if (top < 0x7f8u) {
        x = dm_reduce_large(xi, n);
        m = n + (int)(xi >> 31);
    } else {          /* infinity, NaN */
        s = y - y; c = y - y;
        return;
    }
    const double x2 = x * x;
    /* sinf_poly, n even:  x + x^3 S1 + x^5 (S2 + x^2 S3) */
    const double x3 = x * x2;
    const double s1 = dm_fma(x2, -0x1.994eb3774cf24p-13, 0x1.1107605230bc4p-7);
    const double x5 = x3 * x2;
    const double sa = dm_fma(x3, -0x1.555545995a603p-3, x);
    const float sinp = (float)dm_fma(x5, s1, sa);
    /* sinf_poly, n odd:  (C0 + x^2 C1) + x^4 C2 + x^6 (C3 + x^2 C4) */
    const double x4 = x2 * x2;
    const double c2 = dm_fma(x2, 0x1.99343027bf8c3p-16, -0x1.6c087e89a359dp-10);
    const double c1 = dm_fma(x2, -0x1.ffffffd0c621cp-2, 1.0);
    const double x6 = x4 * x2;
    const double ca = dm_fma(x4, 0x1.55553e1068f19p-5, c1);
    const float cosp = (float)dm_fma(x6, c2, ca);
    /* sign[m & 3] = {1, -1, -1, 1} multiplies x in the odd polynomial, table m & 2 holds the negated even polynomial:
       negation commutes with every rounding, so the signs are applied to the results */
    const float so = ((m + 1) & 2) ? -sinp : sinp;
    const float ce = (m & 2) ? -cosp : cosp;
    s = (n & 1) ? ce : so;
    c = (n & 1) ? so : ce;
    if (top < 0x398u) { s = y; c = 1.0f; }   /* |y| < 2^-12 */
}
/* the same as a CALL on the device, for sites off the hot path (light sampling: once per path or per light sample of an
   environment light): the inlined binary64 polynomials would cost those kernels registers all the time */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VCM_DM_INLINE_SINCOS)
struct DmSinCos { float s, c; };
__device__ __attribute__((noinline)) DmSinCos dm_sincosf_call(float y) { DmSinCos r; dm_sincosf(y, r.s, r.c); return r; }
VCM_HD void dm_sincosf_cold(float y, float &s, float &c) { const DmSinCos r = dm_sincosf_call(y); s = r.s; c = r.c; }
#else
VCM_HD void dm_sincosf_cold(float y, float &s, float &c) { dm_sincosf(y, s, c); }
#endif
VCM_HD float dm_sinf(float x) { float s, c; dm_sincosf(x, s, c); return s; }
VCM_HD float dm_cosf(float x) { float s, c; dm_sincosf(x, s, c); return c; }

/* ---- powf (glibc e_powf.c: log2_inline, exp2_inline; __powf_log2_data, __exp2f_data) ---- */
#define VCM_DM_LOG2_WORDS 32   /* {invc, logc} x 16 */
#define VCM_DM_EXP2_WORDS 32
VCM_HD const double *dm_log2_table()
{
    static const double T[VCM_DM_LOG2_WORDS] = {
        0x1.661ec79f8f3bep+0, -0x1.efec65b963019p-2, 0x1.571ed4aaf883dp+0, -0x1.b0b6832d4fca4p-2,
        0x1.49539f0f010b0p+0, -0x1.7418b0a1fb77bp-2, 0x1.3c995b0b80385p+0, -0x1.39de91a6dcf7bp-2,
        0x1.30d190c8864a5p+0, -0x1.01d9bf3f2b631p-2, 0x1.25e227b0b8ea0p+0, -0x1.97c1d1b3b7af0p-3,
        0x1.1bb4a4a1a343fp+0, -0x1.2f9e393af3c9fp-3, 0x1.12358f08ae5bap+0, -0x1.960cbbf788d5cp-4,
        0x1.0953f419900a7p+0, -0x1.a6f9db6475fcep-5, 0x1.0000000000000p+0, 0x0.0p+0,
        0x1.e608cfd9a47acp-1, 0x1.338ca9f24f53dp-4, 0x1.ca4b31f026aa0p-1, 0x1.476a9543891bap-3,
        0x1.b2036576afce6p-1, 0x1.e840b4ac4e4d2p-3, 0x1.9c2d163a1aa2dp-1, 0x1.40645f0c6651cp-2,
        0x1.886e6037841edp-1, 0x1.88e9c2c1b9ff8p-2, 0x1.767dcf5534862p-1, 0x1.ce0a44eb17bccp-2 };
    return T;
}
VCM_HD const unsigned long long *dm_exp2_table()
{   /* asuint64(2^(i/32)) - (i << 47) */
    static const unsigned long long T[VCM_DM_EXP2_WORDS] = {
        0x3ff0000000000000ull, 0x3fefd9b0d3158574ull, 0x3fefb5586cf9890full, 0x3fef9301d0125b51ull,
        0x3fef72b83c7d517bull, 0x3fef54873168b9aaull, 0x3fef387a6e756238ull, 0x3fef1e9df51fdee1ull,
        0x3fef06fe0a31b715ull, 0x3feef1a7373aa9cbull, 0x3feedea64c123422ull, 0x3feece086061892dull,
        0x3feebfdad5362a27ull, 0x3feeb42b569d4f82ull, 0x3feeab07dd485429ull, 0x3feea47eb03a5585ull,
        0x3feea09e667f3bcdull, 0x3fee9f75e8ec5f74ull, 0x3feea11473eb0187ull, 0x3feea589994cce13ull,
        0x3feeace5422aa0dbull, 0x3feeb737b0cdc5e5ull, 0x3feec49182a3f090ull, 0x3feed503b23e255dull,
        0x3feee89f995ad3adull, 0x3feeff76f2fb5e47ull, 0x3fef199bdd85529cull, 0x3fef3720dcef9069ull,
        0x3fef5818dcfba487ull, 0x3fef7c97337b9b5full, 0x3fefa4afa2a490daull, 0x3fefd0765b6e4540ull };
    return T;
}

/* The two tables are per-lane GATHERS.  A kernel that samples Phong lobes (pow(u, 1 / (n + 1)): utils.hxx:91) copies
 * them into LDS next to the scene's small tables (stage_scene_tables -> dm_stage_tables: 512 bytes) and reads them
 * with ds_read; `lds = false` (the merge kernels, host code) reads the constant arrays. */
#if defined(__HIP_DEVICE_COMPILE__)
__shared__ double g_ldsDmLog2[VCM_DM_LOG2_WORDS];
__shared__ unsigned long long g_ldsDmExp2[VCM_DM_EXP2_WORDS];
#endif
VCM_HD void dm_stage_tables()   /* every thread of the block; the caller's barrier publishes the copy */
{
#if defined(__HIP_DEVICE_COMPILE__)
    const int t = (int)threadIdx.x;
    if (t < VCM_DM_LOG2_WORDS) g_ldsDmLog2[t] = dm_log2_table()[t];
    else if (t < VCM_DM_LOG2_WORDS + VCM_DM_EXP2_WORDS) g_ldsDmExp2[t - VCM_DM_LOG2_WORDS] = dm_exp2_table()[t - VCM_DM_LOG2_WORDS];
#endif
}

/* checkint: 0 = y is not an integer, 1 = odd, 2 = even */
VCM_HD int dm_checkint(uint32_t iy)
{
    const int e = (int)((iy >> 23) & 0xffu);
    if (e < 0x7f) return 0;
    if (e > 0x7f + 23) return 2;
    if (iy & ((1u << (0x7f + 23 - e)) - 1u)) return 0;
    if (iy & (1u << (0x7f + 23 - e))) return 1;
    return 2;
}
VCM_HD bool dm_zeroinfnan(uint32_t ix) { return 2u * ix - 1u >= 2u * 0x7f800000u - 1u; }

/* glibc's powf, every case */
#if defined(__HIP_DEVICE_COMPILE__) && !defined(VCM_DM_INLINE_POW)
#define VCM_DM_COLD __device__ __attribute__((noinline))
#else
#define VCM_DM_COLD VCM_HD
#endif
template <bool LDS>
VCM_DM_COLD float dm_powf_glibc_t(float x, float y)
{
    const bool lds = LDS; (void)lds;
    uint32_t signBias = 0u;
    uint32_t ix = f2u(x);
    const uint32_t iy = f2u(y);
    if (ix - 0x00800000u >= 0x7f800000u - 0x00800000u || dm_zeroinfnan(iy)) {
        /* x < 2^-126, infinite or NaN; or y zero, infinite or NaN */
        if (dm_zeroinfnan(iy)) {
            if (2u * iy == 0u) return 1.0f;
            if (ix == 0x3f800000u) return 1.0f;
            if (2u * ix > 2u * 0x7f800000u || 2u * iy > 2u * 0x7f800000u) return x + y;
            if (2u * ix == 2u * 0x3f800000u) return 1.0f;
            if ((2u * ix < 2u * 0x3f800000u) == !(iy & 0x80000000u)) return 0.0f;   /* |x| < 1 and y = inf, or |x| > 1 and y = -inf */
            return y * y;
        }
        if (dm_zeroinfnan(ix)) {
            float x2 = x * x;
            if ((ix & 0x80000000u) && dm_checkint(iy) == 1) { x2 = -x2; signBias = 1u; }
            if (2u * ix == 0u && (iy & 0x80000000u)) return signBias ? -u2f(0x7f800000u) : u2f(0x7f800000u);
            return (iy & 0x80000000u) ? 1.0f / x2 : x2;
        }
        if (ix & 0x80000000u) {   /* finite x < 0 */
            const int yint = dm_checkint(iy);
            if (yint == 0) return u2f(0x7fc00000u);
            if (yint == 1) signBias = 0x10000u;
            ix &= 0x7fffffffu;
        }
        if (ix < 0x00800000u) {   /* subnormal x: normalised, the exponent becomes negative */
            ix = f2u(u2f(ix) * 0x1p23f);
            ix &= 0x7fffffffu;
            ix -= 23u << 23;
        }
    }
    /* log2_inline: x = 2^k z, z in [0x1.66p-1, 0x1.66p0), c_i near the centre of z's sixteenth */
    const uint32_t tmp = ix - 0x3f330000u;
    const int i = (int)((tmp >> 19) & 15u);
    const uint32_t top = tmp & 0xff800000u;
    const uint32_t iz = ix - top;
    const int k = (int32_t)top >> 23;
    double invc, logc;
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds) { invc = g_ldsDmLog2[2 * i]; logc = g_ldsDmLog2[2 * i + 1]; }
    else
#endif
    { invc = dm_log2_table()[2 * i]; logc = dm_log2_table()[2 * i + 1]; }
    const double z = (double)u2f(iz);
    const double r = dm_fma(z, invc, -1.0);
    const double y0 = logc + (double)k;
    const double a = dm_fma(0x1.27616c9496e0bp-2, r, -0x1.71969a075c67ap-2);
    const double p = dm_fma(0x1.ec70a6ca7baddp-2, r, -0x1.7154748bef6c8p-1);
    const double r2 = r * r;
    double q = dm_fma(r, 0x1.71547652ab82bp+0, y0);
    const double r4 = r2 * r2;
    q = dm_fma(r2, p, q);
    const double logx = dm_fma(a, r4, q);
    const double ylogx = (double)y * logx;   /* cannot overflow: y is binary32 */
    if (((d2u(ylogx) >> 47) & 0xffffull) >= 0x80bfull) {   /* |y log2 x| >= 126 */
        if (ylogx > 0x1.fffffffd1d571p+6) return signBias ? -u2f(0x7f800000u) : u2f(0x7f800000u);   /* overflow */
        if (ylogx <= -150.0) return signBias ? -0.0f : 0.0f;                                            /* underflow */
        if (ylogx < -149.0) return signBias ? -u2f(0x00000001u) : u2f(0x00000001u);   /* __math_may_uflowf: 0x1.4p-75f squared */
    }
    /* exp2_inline: 32 t = kk + rr, 2^t = 2^(kk/32) (1 + rr ln2/32 ...) */
    double kd = ylogx + 0x1.8p+47;
    const unsigned long long ki = d2u(kd);
    kd -= 0x1.8p+47;
    const double rr = ylogx - kd;
    unsigned long long t;
#if defined(__HIP_DEVICE_COMPILE__)
    if (lds) t = g_ldsDmExp2[ki & 31ull];
    else
#endif
    t = dm_exp2_table()[ki & 31ull];
    t += (ki + signBias) << 47;
    const double sc = u2d(t);
    const double zz = dm_fma(0x1.c6af84b912394p-5, rr, 0x1.ebfce50fac4f3p-3);
    const double rr2 = rr * rr;
    double yy = dm_fma(0x1.62e42ff0c52d6p-1, rr, 1.0);
    yy = dm_fma(zz, rr2, yy);
    yy = yy * sc;
    return (float)yy;
}

VCM_HD float dm_powf_glibc(float x, float y, bool lds) { return lds ? dm_powf_glibc_t<true>(x, y) : dm_powf_glibc_t<false>(x, y); }

/* x^n, n >= 1 an integer-valued float <= 65536: binary exponentiation in binary64, least-significant bit first */
VCM_HD double dm_pow_int(float xf, float nf)
{
    unsigned n = (unsigned)nf;
    double b = (double)xf, r = 1.0;
    for (;;) {
        if (n & 1u) r = r * b;
        n >>= 1;
        if (n == 0u) break;
        b = b * b;
    }
    return r;
}
/* the one deviation from glibc (see the top of the file): which exponents take the binary exponentiation, for EVERY x
   (zeros, negative numbers, infinities and NaNs come out of the multiplications as glibc returns them, up to the NaN's
   payload) */
VCM_HD bool dm_pow_is_int_case(float yf) { return yf >= 1.0f && yf <= 65536.0f && yf == floorf(yf); }

VCM_HD float dm_powf(float xf, float yf, bool lds = false)
{
    if (dm_pow_is_int_case(yf)) return (float)dm_pow_int(xf, yf);
    return dm_powf_glibc(xf, yf, lds);
}

/* dm_powf for call sites where the exponent is a material constant (the Phong exponent): if every active lane of the
 * wave holds the same integer exponent, the binary exponentiation is driven by SCALAR control flow -- only the
 * ~log2(n) + popcount(n) binary64 multiplies remain as vector work, instead of a per-lane loop with selects.  Same
 * multiplication sequence as dm_pow_int, hence the same bits; any other case falls back to dm_powf. */
VCM_HD float dm_powf_wave(float xf, float yf, bool lds = false /* true only in kernels that call stage_scene_tables() */, bool intOnly = false)
{
#if defined(__HIP_DEVICE_COMPILE__)
    const float y0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, yf)));
    if (y0 >= 1.0f && y0 <= 65536.0f && y0 == floorf(y0) && __builtin_amdgcn_ballot_w64(!(yf == y0)) == 0ull) {   /* wave-uniform */
        unsigned n = (unsigned)y0;
        double b = (double)xf, r = 1.0;
        if (n == 90u) {
            /* the exponent of the reference's glossy floor (scene.hxx:173), straight-line: the products the loop below forms
               for n = 1011010b, least-significant bit first -- 1.0 * b^2 is b^2 exactly -- six squarings and three
               multiplies, no select, no branch */
            const double b2 = b * b, b4 = b2 * b2, b8 = b4 * b4, b16 = b8 * b8, b32 = b16 * b16, b64 = b32 * b32;
            r = b2 * b8;
            r = r * b16;
            r = r * b64;
            return (float)r;
        }
        for (;;) {
            if (n & 1u) r = r * b;
            n >>= 1;
            if (n == 0u) break;
            b = b * b;
        }
        return (float)r;
    }
    /* intOnly: the caller's kernel was chosen because every Phong exponent of the scene is an integer in [1, 65536]
       (vcm_core.h DScene::kIntPhong) -- lanes with DIFFERENT such exponents: the per-lane loop, and no call of the
       general function in this kernel */
    if (intOnly) return (float)dm_pow_int(xf, yf);
#else
    (void)intOnly;
#endif
    return dm_powf(xf, yf, lds);
}

} // namespace vcm
#endif
