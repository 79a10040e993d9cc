// vcm_math.h -- fp32 vector maths of the hot path, host+device.
//
// Operand order is part of the contract: every function evaluates exactly the
// expression tree of the reference's src/math.hxx / src/frame.hxx (cited per
// function) with IEEE-754 binary32 operations and NO FMA contraction
// (the library is built with -ffp-contract=off), so that results are
// bit-identical to the x86-64 reference build.  Do not "simplify" (a/len is a
// division per component, not a multiply by the reciprocal; Dot starts from 0).
#ifndef SMALLVCM_AMD_VCM_MATH_H
#define SMALLVCM_AMD_VCM_MATH_H

#include <stdint.h>
#include <math.h>

#if defined(__HIPCC__)
#include <hip/hip_runtime.h>
#define VCM_HD __host__ __device__ inline __attribute__((always_inline))
#define VCM_D  __device__ inline __attribute__((always_inline))
#else
#define VCM_HD inline
#define VCM_D  inline
#endif

namespace vcm {

#define VCM_PI_F       3.14159265358979f          /* math.hxx:30 */
#define VCM_INV_PI_F   (1.f / VCM_PI_F)           /* math.hxx:31 */
#define VCM_EPS_COSINE 1e-6f                      /* utils.hxx:32 */
#define VCM_EPS_RAY    1e-3f                      /* utils.hxx:33 */
#define VCM_EPS_PHONG  1e-3f                      /* bsdf.hxx:59 */

VCM_HD float u2f(uint32_t u) { float f; __builtin_memcpy(&f, &u, 4); return f; }
VCM_HD uint32_t f2u(float f) { uint32_t u; __builtin_memcpy(&u, &f, 4); return u; }

/* std::max / std::min semantics (first argument wins ties, no fmaxf NaN rules) */
VCM_HD float smax(float a, float b) { return (a < b) ? b : a; }
VCM_HD float smin(float a, float b) { return (b < a) ? b : a; }
VCM_HD float sqr(float a) { return a * a; }

struct alignas(16) F4 { float x, y, z, w; };
VCM_HD F4 mk4(float x, float y, float z, float w) { F4 r; r.x = x; r.y = y; r.z = z; r.w = w; return r; }

/* Vec3f: math.hxx:87-152 */
struct V3 { float x, y, z; };
VCM_HD V3 mk3(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
VCM_HD V3 sp3(float a) { return mk3(a, a, a); }
VCM_HD V3 ld3(const float *p) { return mk3(p[0], p[1], p[2]); }
VCM_HD V3 operator+(V3 a, V3 b) { return mk3(a.x + b.x, a.y + b.y, a.z + b.z); }
VCM_HD V3 operator-(V3 a, V3 b) { return mk3(a.x - b.x, a.y - b.y, a.z - b.z); }
VCM_HD V3 operator*(V3 a, V3 b) { return mk3(a.x * b.x, a.y * b.y, a.z * b.z); }
VCM_HD V3 operator/(V3 a, V3 b) { return mk3(a.x / b.x, a.y / b.y, a.z / b.z); }
VCM_HD V3 operator*(V3 a, float s) { return mk3(a.x * s, a.y * s, a.z * s); }
VCM_HD V3 operator*(float s, V3 a) { return mk3(s * a.x, s * a.y, s * a.z); }
VCM_HD V3 operator/(V3 a, float s) { return mk3(a.x / s, a.y / s, a.z / s); }
VCM_HD V3 operator-(V3 a) { return mk3(-a.x, -a.y, -a.z); }
/* Dot: math.hxx:138-139 -- T res(0); res += a_i*b_i in order */
VCM_HD float dot(V3 a, V3 b) { float r = 0.f; r += a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
/* Dot(a,a): the leading "0 +" of Dot is a bitwise no-op here (a.x*a.x is never -0) */
VCM_HD float lensqr(V3 a) { float r = a.x * a.x; r += a.y * a.y; r += a.z * a.z; return r; }
VCM_HD bool iszero(V3 a) { return a.x == 0.f && a.y == 0.f && a.z == 0.f; }
VCM_HD float vmax3(V3 a) { float r = a.x; r = smax(r, a.y); r = smax(r, a.z); return r; }
/* Cross: math.hxx:154-162 */
VCM_HD V3 cross(V3 a, V3 b)
{
    return mk3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* Normalize: math.hxx:164-169 */
VCM_HD V3 normalize(V3 a)
{
    const float l2 = dot(a, a);
    const float l = sqrtf(l2);
    return a / l;
}

/* Frame: frame.hxx:32-78 */
struct Frame { V3 mX, mY, mZ; };
VCM_HD void frame_from_z(Frame &f, V3 z)
{   /* :53-59 */
    const V3 tmpZ = f.mZ = normalize(z);
    const V3 tmpX = (fabsf(tmpZ.x) > 0.99f) ? mk3(0.f, 1.f, 0.f) : mk3(1.f, 0.f, 0.f);
    f.mY = normalize(cross(tmpZ, tmpX));
    f.mX = cross(f.mY, tmpZ);
}
VCM_HD V3 to_world(const Frame &f, V3 a) { return f.mX * a.x + f.mY * a.y + f.mZ * a.z; }   /* :61-64 */
VCM_HD V3 to_local(const Frame &f, V3 a) { return mk3(dot(a, f.mX), dot(a, f.mY), dot(a, f.mZ)); }   /* :66-69 */

/* Mat4f::TransformPoint: math.hxx:202-223 (column-major storage :173) */
VCM_HD V3 transform_point(const float *m, V3 v)
{
    float w = m[15];
    w += m[3] * v.x;
    w += m[7] * v.y;
    w += m[11] * v.z;
    const float invW = 1.f / w;
    float rx = m[12]; rx += v.x * m[0]; rx += v.y * m[4]; rx += v.z * m[8];  rx *= invW;
    float ry = m[13]; ry += v.x * m[1]; ry += v.y * m[5]; ry += v.z * m[9];  ry *= invW;
    float rz = m[14]; rz += v.x * m[2]; rz += v.y * m[6]; rz += v.z * m[10]; rz *= invW;
    return mk3(rx, ry, rz);
}

} // namespace vcm
#endif
