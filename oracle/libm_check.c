/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * libm_check.c -- pins the restatement of the reference's libm (oracle/detmath_ref.h, smallvcm_amd/csrc/detmath.h)
 * against the libm the reference is actually linked with in this image (glibc 2.35, x86-64, FMA variants):
 *   sinf, cosf, sincosf   every one of the 2^32 binary32 arguments;
 *   powf                  every x for 17 exponents (the path's 90, 1/91, 0.125 and others, negative ones included),
 *                         2^32 random pairs of bit patterns (all special cases), 2^32 pairs with x in (0, 4),
 *                         y in (-130, 130) (overflow to underflow).
 * Prints one line per set with the number of arguments whose result differs in ANY bit (NaNs compare equal to NaNs).
 * `make -C oracle libm_check && oracle/libm_check` (about four minutes on 8 cores); --quick: 1/64 of every set.
 * The integer-exponent deviation of dmr_powf is reported separately (last lines).
 */
#define _GNU_SOURCE
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "detmath_ref.h"

static inline int same(float a, float b) { return dmr_to_bits32(a) == dmr_to_bits32(b) || (a != a && b != b); }
static inline uint64_t mix(uint64_t z)
{
    z += 0x9e3779b97f4a7c15ull; z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull; z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    const long long step = (argc > 1 && !strcmp(argv[1], "--quick")) ? 64 : 1;
    long total = 0;
    {
        long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
        for (long long i = 0; i < (1ll << 32); i += step) {
            const float x = dmr_from_bits32((uint32_t)i);
            float s, c, s2, c2;
            dmr_sincosf(x, &s, &c);
            volatile float xv = x;
            const float ls = sinf(xv), lc = cosf(xv);
            sincosf(xv, &s2, &c2);
            if (!same(s, ls) || !same(c, lc) || !same(s2, ls) || !same(c2, lc)) bad++;
        }
        printf("sinf / cosf / sincosf, all binary32 arguments (step %lld): %ld differ\n", step, bad);
        total += bad;
    }
    const float ys[] = { 90.f, 1.f / 91.f, 0.125f, 1.f / 2.2f, 0.5f, 2.f, 3.f, 17.f, 1000.f, 90.5f, -0.5f, -3.f, 0.99f, 1e-3f, 1.f / 11.f, 255.f, 1.f / 256.f };
    for (unsigned j = 0; j < sizeof(ys) / sizeof(ys[0]); j++) {
        long bad = 0;
        const float y = ys[j];
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
        for (long long i = 0; i < (1ll << 32); i += step) {
            const float x = dmr_from_bits32((uint32_t)i);
            volatile float xv = x, yv = y;
            if (!same(powf(xv, yv), dmr_powf_glibc(x, y))) bad++;
        }
        printf("powf(x, %a), every x (step %lld): %ld differ\n", y, step, bad);
        total += bad;
    }
    {
        long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
        for (long long i = 0; i < (1ll << 32); i += step) {
            const uint64_t h = mix((uint64_t)i);
            const float x = dmr_from_bits32((uint32_t)h), y = dmr_from_bits32((uint32_t)(h >> 32));
            volatile float xv = x, yv = y;
            if (!same(powf(xv, yv), dmr_powf_glibc(x, y))) bad++;
        }
        printf("powf, random bit patterns (step %lld): %ld differ\n", step, bad);
        total += bad;
    }
    {
        long bad = 0;
#pragma omp parallel for reduction(+ : bad) schedule(dynamic, 1 << 16)
        for (long long i = 0; i < (1ll << 32); i += step) {
            const uint64_t h = mix((uint64_t)i + (1ull << 40));
            const float x = (float)((h & 0xffffff) + 1) * (4.f / 16777216.f);
            const float y = ((float)((h >> 24) & 0xffffff) * (1.f / 16777216.f) - 0.5f) * 260.f;
            volatile float xv = x, yv = y;
            if (!same(powf(xv, yv), dmr_powf_glibc(x, y))) bad++;
        }
        printf("powf, x in (0, 4), y in (-130, 130) (step %lld): %ld differ\n", step, bad);
        total += bad;
    }
    printf("TOTAL restatement vs libm: %ld differ\n", total);
    /* the deviation: integer exponents by binary exponentiation (correctly rounded) against glibc's own powf */
    {
        long n = 0, bad = 0, worst = 0;
        for (uint32_t i = dmr_to_bits32(1e-3f); i <= dmr_to_bits32(1.0f); i += 1) {
            const float x = dmr_from_bits32(i);
            volatile float xv = x, yv = 90.f;
            const float a = dmr_powf(x, 90.f), b = powf(xv, yv);
            if (b > 0.f) {
                n++;
                if (!same(a, b)) { bad++; const long d = labs((long)dmr_to_bits32(a) - (long)dmr_to_bits32(b)); if (d > worst) worst = d; }
            }
        }
        printf("deviation: x^90 by binary exponentiation vs libm powf(x, 90), x in [1e-3, 1]: %ld of %ld differ (%.3f %%), at most %ld ulp\n",
               bad, n, 100.0 * bad / n, worst);
    }
    return total != 0;
}
