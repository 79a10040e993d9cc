/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this.  The product (smallvcm_amd/) never includes or links it.
 *
 * vcm_oracle.cpp -- CPU restatement of SmallVCM's VertexCM::RunIteration
 * (reference: src/vertexcm.hxx:284-1006) and everything it calls:
 *   HashGrid            src/hashgrid.hxx:41-201
 *   BSDF<tFromLight>    src/bsdf.hxx:95-566
 *   lights              src/lights.hxx:112-514
 *   geometry            src/geometry.hxx:65-237, src/scene.hxx:53-102
 *   camera              src/camera.hxx:95-117
 *   samplers / pdfs     src/utils.hxx:36-259, src/frame.hxx:53-69
 *   framebuffer         src/framebuffer.hxx:43-57
 * Every expression keeps the reference's operand order so that, compiled for
 * x86-64 without FMA contraction, results are BIT-IDENTICAL to the reference
 * when both draw the same random numbers and use the same sinf/cosf/powf
 * (oracle/ref_driver.cpp replays the random numbers into the unmodified
 * reference; tests/test_oracle_vs_reference.py asserts bit equality).
 *
 * Deviations from the reference, all deliberate:
 *   - random numbers come from the counter-based stream of philox_ref.h
 *     (one independent stream per path) instead of one sequential mt19937_64;
 *   - sinf/cosf/powf come from detmath_ref.h;
 *   - the iteration is split into phases (light / grid / camera) over a local
 *     path range so that the multi-GPU host logic can be tested on CPU.
 * PARITY: pinned (bit-exact vs oracle/_ref built from /root/reference).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include <vector>
#include <algorithm>
#ifdef _OPENMP
#include <omp.h>
#endif

#include "../include/smallvcm_amd.h"
#include "philox_ref.h"
#include "detmath_ref.h"

namespace {

/* The scene as the oracle reads it: the members of vcm_scene_desc by the same names, the arrays owned here, so that
 * the built-in boxes (vcm_scene_desc) and arbitrary scenes (vcm_scene_desc2: any number of primitives / materials /
 * lights) go through the same restatement -- always the reference's brute-force walk over every primitive in list
 * order (scene.hxx:53-85); the oracle has no acceleration structure, like the reference. */
struct OScene {
    int nPrims, nMaterials, nLights, backgroundLight;
    std::vector<vcm_prim> prims;
    std::vector<vcm_material> materials;
    std::vector<int> mat2light;
    std::vector<vcm_light> lights;
    float sceneCenter[3], sceneRadius, invSceneRadiusSqr;
    vcm_camera camera;
};


/* ---- constants: src/math.hxx:30-31, src/utils.hxx:32-33, src/bsdf.hxx:59 */
#define O_PI_F     3.14159265358979f
#define O_INV_PI_F (1.f / O_PI_F)
#define O_EPS_COSINE 1e-6f
#define O_EPS_RAY    1e-3f
#define O_EPS_PHONG  1e-3f

/* std::max / std::min semantics (first argument wins ties) */
static inline float omax(float a, float b) { return (a < b) ? b : a; }
static inline float omin(float a, float b) { return (b < a) ? b : a; }
static inline float osqr(float a) { return a * a; }

/* ---- Vec3f: src/math.hxx:87-152 (component-wise ops, scalar promoted) */
struct V3 { float x, y, z; };
static inline V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
static inline V3 sp(float a) { return mk(a, a, a); }
static inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
static inline V3 operator*(V3 a, float s) { return a * sp(s); }
static inline V3 operator*(float s, V3 a) { return sp(s) * a; }
static inline V3 operator/(V3 a, float s) { return a / sp(s); }
static inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
static inline float dot(V3 a, V3 b) { float r = 0; r += a.x * b.x; r += a.y * b.y; r += a.z * b.z; return r; }
static inline float lensqr(V3 a) { return dot(a, a); }
static inline bool  iszero(V3 a) { return a.x == 0 && a.y == 0 && a.z == 0; }
static inline float vmax(V3 a) { float r = a.x; r = omax(r, a.y); r = omax(r, a.z); return r; }
static inline V3 cross(V3 a, V3 b)
{   /* src/math.hxx:154-162 */
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline V3 normalize(V3 a)
{   /* src/math.hxx:164-169 */
    const float l2 = dot(a, a);
    const float l = sqrtf(l2);
    return a / l;
}
static inline V3 ld3(const float *p) { return mk(p[0], p[1], p[2]); }

/* ---- Frame: src/frame.hxx:32-78 */
struct Frame { V3 mX, mY, mZ; };
static inline void frame_from_z(Frame &f, V3 z)
{   /* :53-59 */
    const V3 tmpZ = f.mZ = normalize(z);
    const V3 tmpX = (fabsf(tmpZ.x) > 0.99f) ? mk(0, 1, 0) : mk(1, 0, 0);
    f.mY = normalize(cross(tmpZ, tmpX));
    f.mX = cross(f.mY, tmpZ);
}
static inline V3 to_world(const Frame &f, V3 a) { return f.mX * a.x + f.mY * a.y + f.mZ * a.z; }  /* :61-64 */
static inline V3 to_local(const Frame &f, V3 a) { return mk(dot(a, f.mX), dot(a, f.mY), dot(a, f.mZ)); } /* :66-69 */

/* ---- Ray / Isect: src/ray.hxx:34-65 */
struct Ray { V3 org, dir; float tmin; };
struct Isect { float dist; int matID; int lightID; V3 normal; };

/* ---- utils: src/utils.hxx */
static inline float luminance(V3 c)
{   /* :36-41 */
    return 0.212671f * c.x + 0.715160f * c.y + 0.072169f * c.z;
}
static float fresnel_dielectric(float cosInc, float ior)
{   /* :43-74 */
    if (ior < 0) return 1.f;
    float eta;
    if (cosInc < 0.f) { cosInc = -cosInc; eta = ior; }
    else              { eta = 1.f / ior; }
    const float sinTrans2 = osqr(eta) * (1.f - osqr(cosInc));
    const float cosTrans = sqrtf(omax(0.f, 1.f - sinTrans2));
    const float term1 = eta * cosTrans;
    const float rParallel = (cosInc - term1) / (cosInc + term1);
    const float term2 = eta * cosInc;
    const float rPerp = (term2 - cosTrans) / (term2 + cosTrans);
    return 0.5f * (osqr(rParallel) + osqr(rPerp));
}
static inline V3 reflect_local(V3 v) { return mk(-v.x, -v.y, v.z); }  /* :77-80 */

static inline V3 sample_power_cos_hemisphere(float sx, float sy, float power)
{   /* :85-103 (oPdfW == NULL at the only call site bsdf.hxx:296) */
    const float term1 = 2.f * O_PI_F * sx;
    const float term2 = dmr_powf(sy, 1.f / (power + 1.f));
    const float term3 = sqrtf(1.f - term2 * term2);
    return mk(dmr_cosf(term1) * term3, dmr_sinf(term1) * term3, term2);
}
static inline float power_cos_hemisphere_pdf(V3 n, V3 d, float power)
{   /* :105-113 */
    const float cosTheta = omax(0.f, dot(n, d));
    return (power + 1.f) * dmr_powf(cosTheta, power) * (O_INV_PI_F * 0.5f);
}
static inline void sample_concentric_disc(float sx, float sy, float &ox, float &oy)
{   /* :119-160 */
    float phi, r;
    const float a = 2 * sx - 1;
    const float b = 2 * sy - 1;
    if (a > -b) {
        if (a > b) { r = a;  phi = (O_PI_F / 4.f) * (b / a); }
        else       { r = b;  phi = (O_PI_F / 4.f) * (2.f - (a / b)); }
    } else {
        if (a < b) { r = -a; phi = (O_PI_F / 4.f) * (4.f + (b / a)); }
        else {
            r = -b;
            if (b != 0) phi = (O_PI_F / 4.f) * (6.f - (a / b));
            else        phi = 0;
        }
    }
    ox = r * dmr_cosf(phi);
    oy = r * dmr_sinf(phi);
}
static inline float concentric_disc_pdf_a() { return O_INV_PI_F; }  /* :162-165 */
static inline V3 sample_cos_hemisphere(float sx, float sy, float *pdfW)
{   /* :173-190 */
    const float term1 = 2.f * O_PI_F * sx;
    const float term2 = sqrtf(1.f - sy);
    const V3 ret = mk(dmr_cosf(term1) * term2, dmr_sinf(term1) * term2, sqrtf(sy));
    if (pdfW) *pdfW = ret.z * O_INV_PI_F;
    return ret;
}
static inline float cos_hemisphere_pdf(V3 n, V3 d) { return omax(0.f, dot(n, d)) * O_INV_PI_F; } /* :192-197 */
static inline void sample_uniform_triangle(float sx, float sy, float &u, float &v)
{   /* :202-207 */
    const float term = sqrtf(sx);
    u = 1.f - term;
    v = sy * term;
}
static inline V3 sample_uniform_sphere(float sx, float sy, float *pdf)
{   /* :212-230 */
    const float term1 = 2.f * O_PI_F * sx;
    const float term2 = 2.f * sqrtf(sy - sy * sy);
    const V3 ret = mk(dmr_cosf(term1) * term2, dmr_sinf(term1) * term2, 1.f - 2.f * sy);
    if (pdf) *pdf = O_INV_PI_F * 0.25f;
    return ret;
}
static inline float uniform_sphere_pdf() { return O_INV_PI_F * 0.25f; }  /* :232-236 */
static inline float pdf_w_to_a(float pdfW, float dist, float cosThere)
{   /* :245-251 */
    return pdfW * fabsf(cosThere) / osqr(dist);
}

/* ---- geometry: src/geometry.hxx */
static bool tri_intersect(const vcm_prim &t, const Ray &ray, Isect &res)
{   /* :125-156 */
    const V3 ao = ld3(t.p0) - ray.org;
    const V3 bo = ld3(t.p1) - ray.org;
    const V3 co = ld3(t.p2) - ray.org;
    const V3 v0 = cross(co, bo);
    const V3 v1 = cross(bo, ao);
    const V3 v2 = cross(ao, co);
    const float v0d = dot(v0, ray.dir);
    const float v1d = dot(v1, ray.dir);
    const float v2d = dot(v2, ray.dir);
    if (((v0d < 0.f) && (v1d < 0.f) && (v2d < 0.f)) ||
        ((v0d >= 0.f) && (v1d >= 0.f) && (v2d >= 0.f))) {
        const V3 n = ld3(t.n);
        const float distance = dot(n, ao) / dot(n, ray.dir);
        if ((distance > ray.tmin) && (distance < res.dist)) {
            res.normal = n;
            res.matID = t.matID;
            res.dist = distance;
            return true;
        }
    }
    return false;
}
static bool sph_intersect(const vcm_prim &s, const Ray &ray, Isect &res)
{   /* :198-237; note disc is evaluated in float, then widened (:211) */
    const V3 center = ld3(s.p0);
    const float radius = s.p1[0];
    const V3 to = ray.org - center;
    const float A = dot(ray.dir, ray.dir);
    const float B = 2 * dot(ray.dir, to);
    const float C = dot(to, to) - (radius * radius);
    const float discF = B * B - 4 * A * C;
    const double disc = discF;
    if (disc < 0) return false;
    const double discSqrt = sqrt(disc);
    const double q = (B < 0) ? ((-B - discSqrt) / 2.f) : ((-B + discSqrt) / 2.f);
    double t0 = q / A;
    double t1 = C / q;
    if (t0 > t1) std::swap(t0, t1);
    float resT;
    if (t0 > ray.tmin && t0 < res.dist)      resT = float(t0);
    else if (t1 > ray.tmin && t1 < res.dist) resT = float(t1);
    else return false;
    res.dist = resT;
    res.matID = s.matID;
    res.normal = normalize(to + sp(resT) * ray.dir);
    return true;
}
static inline bool prim_intersect(const vcm_prim &p, const Ray &ray, Isect &res)
{
    return (p.type == VCM_PRIM_TRIANGLE) ? tri_intersect(p, ray, res) : sph_intersect(p, ray, res);
}

/* ---- BSDF: src/bsdf.hxx:61-576 */
enum { kDiffuse = 1, kPhong = 2, kReflect = 4, kRefract = 8, kSpecular = 12 };
struct Bsdf {
    int   matID;        /* <0 invalid */
    Frame frame;
    V3    localDirFix;
    bool  isDelta;
    float diffProb, phongProb, reflProb, refrProb;
    float contProb;
    float reflectCoeff;
    V3    isectNormal;  /* kept for debugging / device comparison */
};

struct LightVertex {    /* src/vertexcm.hxx:79-101 */
    V3 hitpoint, throughput;
    unsigned pathLength;
    Bsdf bsdf;
    float dVCM, dVC, dVM;
};
struct SubPathState {   /* src/vertexcm.hxx:64-76 */
    V3 origin, direction, throughput;
    unsigned pathLength;
    unsigned isFiniteLight;
    unsigned specularPath;
    float dVCM, dVC, dVM;
};
struct Splat { int pixel; V3 c; };

struct Oracle {
    OScene sc;
    bool useVM, useVC, lightTraceOnly, ppm;
    int renderer;               /* 0 VertexCM, 1 PathTracer (pathtracer.hxx), 2 EyeLight (eyelight.hxx) */
    int iteration;              /* aIteration of the current RunIteration */
    float baseRadius, radiusAlpha;
    int seed;
    int rank, world;
    int resX, resY, N;
    int p0, p1;                 /* local path range */
    int threads;
    int iterations;             /* mIterations == RNG localIteration */

    /* per-iteration (vertexcm.hxx:288-308) */
    unsigned minLen, maxLen;
    float radius, radiusSqr, vmNormalization, misVmWeightFactor, misVcWeightFactor;
    float lightSubPathCount;

    std::vector<LightVertex> lightVertices;   /* local, reference order */
    std::vector<int> pathEnds;                /* local */
    std::vector<float> records;               /* grid input: all ranks' merge records */
    long long nRecords;

    /* HashGrid state: src/hashgrid.hxx:203-213 */
    V3 bboxMin, bboxMax;
    std::vector<int> indices, cellEnds;
    float gRadiusSqr, cellSize, invCellSize;

    std::vector<float> fb;                    /* running sum, W*H*3 */
    std::vector<unsigned char> lightCounts, camCounts;
    std::vector<float> camColor;              /* per local camera path, this iteration */
    std::vector<int> camTarget;               /* pixel the colour is added to, -1 = dropped */
    vcm_stats st;
};

/* ---- scene: src/scene.hxx:53-102 */
static bool scene_intersect(const OScene &sc, const Ray &ray, Isect &res)
{   /* :53-70 + GeometryList::Intersect geometry.hxx:65-78 */
    bool any = false;
    for (int i = 0; i < sc.nPrims; i++) {
        const bool hit = prim_intersect(sc.prims[i], ray, res);
        if (hit) any = hit;
    }
    if (any) res.lightID = sc.mat2light[res.matID];
    return any;
}
static bool scene_occluded(const OScene &sc, V3 point, V3 dir, float tmax)
{   /* :72-85 + GeometryList::IntersectP geometry.hxx:80-91 */
    Ray ray;
    ray.org = point + dir * O_EPS_RAY;
    ray.dir = dir;
    ray.tmin = 0;
    Isect isect;
    isect.dist = tmax - 2 * O_EPS_RAY;
    for (int i = 0; i < sc.nPrims; i++)
        if (prim_intersect(sc.prims[i], ray, isect)) return true;
    return false;
}

/* ---- BSDF methods */
static void bsdf_component_probabilities(Bsdf &b, const vcm_material &m)
{   /* bsdf.hxx:528-566 */
    b.reflectCoeff = fresnel_dielectric(b.localDirFix.z, m.ior);
    const float albedoDiffuse = luminance(ld3(m.diffuse));
    const float albedoPhong   = luminance(ld3(m.phong));
    const float albedoReflect = b.reflectCoeff * luminance(ld3(m.mirror));
    const float albedoRefract = (1.f - b.reflectCoeff) * (m.ior > 0.f ? 1.f : 0.f);
    const float totalAlbedo = albedoDiffuse + albedoPhong + albedoReflect + albedoRefract;
    if (totalAlbedo < 1e-9f) {
        b.diffProb = b.phongProb = b.reflProb = b.refrProb = 0.f;
        b.contProb = 0.f;
    } else {
        b.diffProb  = albedoDiffuse / totalAlbedo;
        b.phongProb = albedoPhong / totalAlbedo;
        b.reflProb  = albedoReflect / totalAlbedo;
        b.refrProb  = albedoRefract / totalAlbedo;
        b.contProb = vmax(ld3(m.diffuse) + ld3(m.phong) + b.reflectCoeff * ld3(m.mirror)) +
                     (1.f - b.reflectCoeff);
        b.contProb = omin(1.f, omax(0.f, b.contProb));
    }
}
static void bsdf_setup(Bsdf &b, const Ray &ray, const Isect &isect, const OScene &sc)
{   /* bsdf.hxx:95-117 */
    b.matID = -1;
    b.isectNormal = isect.normal;
    frame_from_z(b.frame, isect.normal);
    b.localDirFix = to_local(b.frame, -ray.dir);
    if (fabsf(b.localDirFix.z) < O_EPS_COSINE) return;
    bsdf_component_probabilities(b, sc.materials[isect.matID]);
    b.isDelta = (b.diffProb == 0) && (b.phongProb == 0);
    b.matID = isect.matID;
}
static V3 bsdf_eval_diffuse(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf)
{   /* bsdf.hxx:393-412 */
    if (b.diffProb == 0) return sp(0);
    if (b.localDirFix.z < O_EPS_COSINE || gen.z < O_EPS_COSINE) return sp(0);
    if (dirPdf) *dirPdf += b.diffProb * omax(0.f, gen.z * O_INV_PI_F);
    if (revPdf) *revPdf += b.diffProb * omax(0.f, b.localDirFix.z * O_INV_PI_F);
    return ld3(m.diffuse) * O_INV_PI_F;
}
static V3 bsdf_eval_phong(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf)
{   /* bsdf.hxx:414-446 */
    if (b.phongProb == 0) return sp(0);
    if (b.localDirFix.z < O_EPS_COSINE || gen.z < O_EPS_COSINE) return sp(0);
    const V3 refl = reflect_local(b.localDirFix);
    const float dot_R_Wi = dot(refl, gen);
    if (dot_R_Wi <= O_EPS_PHONG) return sp(0.f);
    if (dirPdf || revPdf) {
        const float pdfW = b.phongProb * power_cos_hemisphere_pdf(refl, gen, m.phongExp);
        if (dirPdf) *dirPdf += pdfW;
        if (revPdf) *revPdf += pdfW;
    }
    const V3 rho = ld3(m.phong) * (m.phongExp + 2.f) * 0.5f * O_INV_PI_F;
    return rho * dmr_powf(dot_R_Wi, m.phongExp);
}
static void bsdf_pdf_diffuse(const Bsdf &b, V3 gen, float *dirPdf, float *revPdf)
{   /* bsdf.hxx:456-472 */
    if (b.diffProb == 0) return;
    if (dirPdf) *dirPdf += b.diffProb * omax(0.f, gen.z * O_INV_PI_F);
    if (revPdf) *revPdf += b.diffProb * omax(0.f, b.localDirFix.z * O_INV_PI_F);
}
static void bsdf_pdf_phong(const Bsdf &b, const vcm_material &m, V3 gen, float *dirPdf, float *revPdf)
{   /* bsdf.hxx:474-503 */
    if (b.phongProb == 0) return;
    const V3 refl = reflect_local(b.localDirFix);
    const float dot_R_Wi = dot(refl, gen);
    if (dot_R_Wi <= O_EPS_PHONG) return;
    if (dirPdf || revPdf) {
        const float pdfW = power_cos_hemisphere_pdf(refl, gen, m.phongExp) * b.phongProb;
        if (dirPdf) *dirPdf += pdfW;
        if (revPdf) *revPdf += pdfW;
    }
}
static V3 bsdf_evaluate(const Bsdf &b, const OScene &sc, V3 worldDirGen, float &cosThetaGen,
                        float *dirPdf, float *revPdf)
{   /* bsdf.hxx:128-153 */
    V3 result = sp(0);
    if (dirPdf) *dirPdf = 0;
    if (revPdf) *revPdf = 0;
    const V3 gen = to_local(b.frame, worldDirGen);
    if (gen.z * b.localDirFix.z < 0) return result;
    cosThetaGen = fabsf(gen.z);
    const vcm_material &m = sc.materials[b.matID];
    result = result + bsdf_eval_diffuse(b, m, gen, dirPdf, revPdf);
    result = result + bsdf_eval_phong(b, m, gen, dirPdf, revPdf);
    return result;
}
static float bsdf_pdf(const Bsdf &b, const OScene &sc, V3 worldDirGen, bool evalRev)
{   /* bsdf.hxx:161-180 */
    const V3 gen = to_local(b.frame, worldDirGen);
    if (gen.z * b.localDirFix.z < 0) return 0;
    const vcm_material &m = sc.materials[b.matID];
    float directPdfW = 0, reversePdfW = 0;
    bsdf_pdf_diffuse(b, gen, &directPdfW, &reversePdfW);
    bsdf_pdf_phong(b, m, gen, &directPdfW, &reversePdfW);
    return evalRev ? reversePdfW : directPdfW;
}
static V3 bsdf_sample(const Bsdf &b, const OScene &sc, bool fixIsLight, V3 rnd,
                      V3 &worldDirGen, float &pdfW, float &cosThetaGen, unsigned &sampledEvent)
{   /* bsdf.hxx:191-257 */
    if (rnd.z < b.diffProb) sampledEvent = kDiffuse;
    else if (rnd.z < b.diffProb + b.phongProb) sampledEvent = kPhong;
    else if (rnd.z < b.diffProb + b.phongProb + b.reflProb) sampledEvent = kReflect;
    else sampledEvent = kRefract;

    const vcm_material &m = sc.materials[b.matID];
    pdfW = 0;
    V3 result = sp(0);
    V3 gen = sp(0);

    if (sampledEvent == kDiffuse) {
        /* SampleDiffuse :274-288 */
        if (b.localDirFix.z < O_EPS_COSINE) return sp(0);
        float unweightedPdfW;
        gen = sample_cos_hemisphere(rnd.x, rnd.y, &unweightedPdfW);
        pdfW += unweightedPdfW * b.diffProb;
        result = result + ld3(m.diffuse) * O_INV_PI_F;
        if (iszero(result)) return sp(0);
        result = result + bsdf_eval_phong(b, m, gen, &pdfW, NULL);
    } else if (sampledEvent == kPhong) {
        /* SamplePhong :290-318 */
        gen = sample_power_cos_hemisphere(rnd.x, rnd.y, m.phongExp);
        const V3 refl = reflect_local(b.localDirFix);
        {
            Frame fr;
            frame_from_z(fr, refl);
            gen = to_world(fr, gen);
        }
        const float dot_R_Wi = dot(refl, gen);
        if (dot_R_Wi <= O_EPS_PHONG) return sp(0.f);
        bsdf_pdf_phong(b, m, gen, &pdfW, NULL);
        const V3 rho = ld3(m.phong) * (m.phongExp + 2.f) * 0.5f * O_INV_PI_F;
        result = result + rho * dmr_powf(dot_R_Wi, m.phongExp);
        if (iszero(result)) return sp(0);
        result = result + bsdf_eval_diffuse(b, m, gen, &pdfW, NULL);
    } else if (sampledEvent == kReflect) {
        /* SampleReflect :320-333 */
        gen = reflect_local(b.localDirFix);
        pdfW += b.reflProb;
        result = result + b.reflectCoeff * ld3(m.mirror) / fabsf(gen.z);
        if (iszero(result)) return sp(0);
    } else {
        /* SampleRefract :335-388 */
        if (m.ior < 0) return sp(0);
        float cosI = b.localDirFix.z;
        float cosT, eta;
        if (cosI < 0.f) { eta = m.ior; cosI = -cosI; cosT = 1.f; }
        else            { eta = 1.f / m.ior; cosT = -1.f; }
        const float sinI2 = 1.f - cosI * cosI;
        const float sinT2 = osqr(eta) * sinI2;
        if (sinT2 < 1.f) {
            cosT *= sqrtf(omax(0.f, 1.f - sinT2));
            gen = mk(-eta * b.localDirFix.x, -eta * b.localDirFix.y, cosT);
            pdfW += b.refrProb;
            const float refractCoeff = 1.f - b.reflectCoeff;
            if (!fixIsLight) result = result + sp(refractCoeff * osqr(eta) / fabsf(cosT));
            else             result = result + sp(refractCoeff / fabsf(cosT));
        } else {
            pdfW += 0.f;
            result = result + sp(0.f);
        }
        if (iszero(result)) return sp(0);
    }

    cosThetaGen = fabsf(gen.z);
    if (cosThetaGen < O_EPS_COSINE) return sp(0.f);
    worldDirGen = to_world(b.frame, gen);
    return result;
}

/* ---- lights: src/lights.hxx */
static inline bool light_is_finite(const vcm_light &l) { return l.type == VCM_LIGHT_AREA || l.type == VCM_LIGHT_POINT; }
static inline bool light_is_delta(const vcm_light &l) { return l.type == VCM_LIGHT_DIRECTIONAL || l.type == VCM_LIGHT_POINT; }
static inline Frame light_frame(const vcm_light &l) { Frame f; f.mX = ld3(l.frameX); f.mY = ld3(l.frameY); f.mZ = ld3(l.frameZ); return f; }

static V3 light_illuminate(const vcm_light &l, const OScene &sc, V3 recvPos, float rx, float ry,
                           V3 &dirToLight, float &distance, float &directPdfW,
                           float *emissionPdfW, float *cosAtLight)
{
    switch (l.type) {
    case VCM_LIGHT_AREA: {   /* :129-166 */
        float u, v;
        sample_uniform_triangle(rx, ry, u, v);
        const V3 lightPoint = ld3(l.p0) + ld3(l.e1) * u + ld3(l.e2) * v;
        dirToLight = lightPoint - recvPos;
        const float distSqr = lensqr(dirToLight);
        distance = sqrtf(distSqr);
        dirToLight = dirToLight / distance;
        const float cosNormalDir = dot(ld3(l.frameZ), -dirToLight);
        if (cosNormalDir < O_EPS_COSINE) return sp(0.f);
        directPdfW = l.invArea * distSqr / cosNormalDir;
        if (cosAtLight) *cosAtLight = cosNormalDir;
        if (emissionPdfW) *emissionPdfW = l.invArea * cosNormalDir * O_INV_PI_F;
        return ld3(l.intensity);
    }
    case VCM_LIGHT_DIRECTIONAL: {   /* :245-265 */
        dirToLight = -ld3(l.frameZ);
        distance = 1e36f;
        directPdfW = 1.f;
        if (cosAtLight) *cosAtLight = 1.f;
        if (emissionPdfW) *emissionPdfW = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        return ld3(l.intensity);
    }
    case VCM_LIGHT_POINT: {   /* :330-353 */
        dirToLight = ld3(l.p0) - recvPos;
        const float distSqr = lensqr(dirToLight);
        directPdfW = distSqr;
        distance = sqrtf(distSqr);
        dirToLight = dirToLight / distance;
        if (cosAtLight) *cosAtLight = 1.f;
        if (emissionPdfW) *emissionPdfW = uniform_sphere_pdf();
        return ld3(l.intensity);
    }
    default: {   /* background :410-437 */
        dirToLight = sample_uniform_sphere(rx, ry, &directPdfW);
        const V3 radiance = ld3(l.intensity) * l.scale;
        distance = 1e36f;
        if (emissionPdfW) *emissionPdfW = directPdfW * concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        if (cosAtLight) *cosAtLight = 1.f;
        return radiance;
    }
    }
}

static V3 light_emit(const vcm_light &l, const OScene &sc, float dx, float dy, float px, float py,
                     V3 &position, V3 &direction, float &emissionPdfW, float *directPdfA, float *cosThetaLight)
{
    switch (l.type) {
    case VCM_LIGHT_AREA: {   /* :168-198 */
        float u, v;
        sample_uniform_triangle(px, py, u, v);
        position = ld3(l.p0) + ld3(l.e1) * u + ld3(l.e2) * v;
        V3 localDirOut = sample_cos_hemisphere(dx, dy, &emissionPdfW);
        emissionPdfW *= l.invArea;
        localDirOut.z = omax(localDirOut.z, O_EPS_COSINE);
        direction = to_world(light_frame(l), localDirOut);
        if (directPdfA) *directPdfA = l.invArea;
        if (cosThetaLight) *cosThetaLight = localDirOut.z;
        return ld3(l.intensity) * localDirOut.z;
    }
    case VCM_LIGHT_DIRECTIONAL: {   /* :267-294 */
        float x, y;
        sample_concentric_disc(px, py, x, y);
        position = ld3(sc.sceneCenter) + sc.sceneRadius * (-ld3(l.frameZ) + ld3(l.frameX) * x + ld3(l.frameY) * y);
        direction = ld3(l.frameZ);
        emissionPdfW = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        if (directPdfA) *directPdfA = 1.f;
        if (cosThetaLight) *cosThetaLight = 1.f;
        return ld3(l.intensity);
    }
    case VCM_LIGHT_POINT: {   /* :355-376 */
        position = ld3(l.p0);
        direction = sample_uniform_sphere(dx, dy, &emissionPdfW);
        if (directPdfA) *directPdfA = 1.f;
        if (cosThetaLight) *cosThetaLight = 1.f;
        return ld3(l.intensity);
    }
    default: {   /* background :439-481 */
        float directPdf;
        direction = sample_uniform_sphere(dx, dy, &directPdf);
        const V3 radiance = ld3(l.intensity) * l.scale;
        float x, y;
        sample_concentric_disc(px, py, x, y);
        Frame frame;
        frame_from_z(frame, direction);
        position = ld3(sc.sceneCenter) + sc.sceneRadius * (-direction + frame.mX * x + frame.mY * y);
        emissionPdfW = directPdf * concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        if (directPdfA) *directPdfA = directPdf;
        if (cosThetaLight) *cosThetaLight = 1.f;
        return radiance;
    }
    }
}

static V3 light_get_radiance(const vcm_light &l, const OScene &sc, V3 rayDir, V3 /*hitPoint*/,
                             float *directPdfA, float *emissionPdfW)
{
    switch (l.type) {
    case VCM_LIGHT_AREA: {   /* :200-221 */
        const float cosOutL = omax(0.f, dot(ld3(l.frameZ), -rayDir));
        if (cosOutL == 0) return sp(0);
        if (directPdfA) *directPdfA = l.invArea;
        if (emissionPdfW) {
            *emissionPdfW = cos_hemisphere_pdf(ld3(l.frameZ), -rayDir);
            *emissionPdfW *= l.invArea;
        }
        return ld3(l.intensity);
    }
    case VCM_LIGHT_DIRECTIONAL:   /* :296-304 */
    case VCM_LIGHT_POINT:         /* :378-386 */
        return sp(0);
    default: {   /* background :483-504 */
        const float directPdf = uniform_sphere_pdf();
        const V3 radiance = ld3(l.intensity) * l.scale;
        const float positionPdf = concentric_disc_pdf_a() * sc.invSceneRadiusSqr;
        if (directPdfA) *directPdfA = directPdf;
        if (emissionPdfW) *emissionPdfW = directPdf * positionPdf;
        return radiance;
    }
    }
}

/* ---- camera: src/camera.hxx:95-117, src/math.hxx:202-223 */
static V3 transform_point(const float *m, V3 v)
{
    const float a[3] = { v.x, v.y, v.z };
    float w = m[3 + 3 * 4];
    for (int c = 0; c < 3; c++) w += m[3 + c * 4] * a[c];
    const float invW = 1.f / w;
    float res[3];
    for (int r = 0; r < 3; r++) {
        res[r] = m[r + 3 * 4];
        for (int c = 0; c < 3; c++) res[r] += a[c] * m[r + c * 4];
        res[r] *= invW;
    }
    return mk(res[0], res[1], res[2]);
}
static inline const vcm_light &get_light(const OScene &sc, int idx)
{   /* Scene::GetLightPtr scene.hxx:98-102 */
    idx = std::min<int>(idx, sc.nLights - 1);
    return sc.lights[idx];
}

/* ---- VertexCM pieces */
static inline float mis(float pdf) { return pdf; }   /* vertexcm.hxx:553-557 */

static void generate_light_sample(Oracle &o, PathRngRef &rng, SubPathState &st)
{   /* vertexcm.hxx:816-858 */
    const OScene &sc = o.sc;
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    const int lightID = int(path_rng_float_ref(&rng) * lightCount);
    const float dx = path_rng_float_ref(&rng);
    const float dy = path_rng_float_ref(&rng);
    const float px = path_rng_float_ref(&rng);
    const float py = path_rng_float_ref(&rng);
    const vcm_light &light = get_light(sc, lightID);
    float emissionPdfW, directPdfA, cosLight;
    st.throughput = light_emit(light, sc, dx, dy, px, py, st.origin, st.direction,
                               emissionPdfW, &directPdfA, &cosLight);
    emissionPdfW *= lightPickProb;
    directPdfA *= lightPickProb;
    st.throughput = st.throughput / emissionPdfW;
    st.pathLength = 1;
    st.isFiniteLight = light_is_finite(light) ? 1 : 0;
    st.specularPath = 0;
    st.dVCM = mis(directPdfA / emissionPdfW);
    if (!light_is_delta(light)) {
        const float usedCosLight = light_is_finite(light) ? cosLight : 1.f;
        st.dVC = mis(usedCosLight / emissionPdfW);
    } else {
        st.dVC = 0.f;
    }
    st.dVM = st.dVC * o.misVcWeightFactor;
}

static void connect_to_camera(Oracle &o, const SubPathState &st, V3 hitpoint, const Bsdf &bsdf,
                              std::vector<Splat> &splats, vcm_stats &stats)
{   /* vertexcm.hxx:862-933 */
    const OScene &sc = o.sc;
    const vcm_camera &cam = sc.camera;
    V3 directionToCamera = ld3(cam.position) - hitpoint;
    if (dot(ld3(cam.forward), -directionToCamera) <= 0.f) return;
    const V3 ip3 = transform_point(cam.worldToRaster, hitpoint);
    const float ipx = ip3.x, ipy = ip3.y;
    if (!(ipx >= 0 && ipy >= 0 && ipx < cam.resolution[0] && ipy < cam.resolution[1])) return;
    const float distEye2 = lensqr(directionToCamera);
    const float distance = sqrtf(distEye2);
    directionToCamera = directionToCamera / distance;
    float cosToCamera, bsdfDirPdfW, bsdfRevPdfW;
    const V3 bsdfFactor = bsdf_evaluate(bsdf, sc, directionToCamera, cosToCamera, &bsdfDirPdfW, &bsdfRevPdfW);
    if (iszero(bsdfFactor)) return;
    bsdfRevPdfW *= bsdf.contProb;
    const float cosAtCamera = dot(ld3(cam.forward), -directionToCamera);
    const float imagePointToCameraDist = cam.imagePlaneDist / cosAtCamera;
    const float imageToSolidAngleFactor = osqr(imagePointToCameraDist) / cosAtCamera;
    const float imageToSurfaceFactor = imageToSolidAngleFactor * fabsf(cosToCamera) / osqr(distance);
    const float cameraPdfA = imageToSurfaceFactor;
    const float wLight = mis(cameraPdfA / o.lightSubPathCount) *
                         (o.misVmWeightFactor + st.dVCM + st.dVC * mis(bsdfRevPdfW));
    const float misWeight = o.lightTraceOnly ? 1.f : (1.f / (wLight + 1.f));
    const float surfaceToImageFactor = 1.f / imageToSurfaceFactor;
    const V3 contrib = misWeight * st.throughput * bsdfFactor / (o.lightSubPathCount * surfaceToImageFactor);
    if (!iszero(contrib)) {
        stats.shadowRays++;
        if (scene_occluded(sc, hitpoint, directionToCamera, distance)) return;
        /* Framebuffer::AddColor framebuffer.hxx:43-57 (bounds already hold) */
        const int x = int(ipx), y = int(ipy);
        Splat s; s.pixel = x + y * o.resX; s.c = contrib;
        splats.push_back(s);
        stats.lightSplats++;
    }
}

template <bool tLightSample>
static bool sample_scattering(Oracle &o, PathRngRef &rng, const Bsdf &bsdf, V3 hitPoint, SubPathState &st)
{   /* vertexcm.hxx:938-1006 */
    const float r0 = path_rng_float_ref(&rng);
    const float r1 = path_rng_float_ref(&rng);
    const float r2 = path_rng_float_ref(&rng);
    float bsdfDirPdfW, cosThetaOut;
    unsigned sampledEvent;
    const V3 bsdfFactor = bsdf_sample(bsdf, o.sc, tLightSample, mk(r0, r1, r2), st.direction,
                                      bsdfDirPdfW, cosThetaOut, sampledEvent);
    if (iszero(bsdfFactor)) return false;
    float bsdfRevPdfW = bsdfDirPdfW;
    if ((sampledEvent & kSpecular) == 0)
        bsdfRevPdfW = bsdf_pdf(bsdf, o.sc, st.direction, true);
    const float contProb = bsdf.contProb;
    if (path_rng_float_ref(&rng) > contProb) return false;
    bsdfDirPdfW *= contProb;
    bsdfRevPdfW *= contProb;
    if (sampledEvent & kSpecular) {
        st.dVCM = 0.f;
        st.dVC *= mis(cosThetaOut);
        st.dVM *= mis(cosThetaOut);
        st.specularPath &= 1;
    } else {
        st.dVC = mis(cosThetaOut / bsdfDirPdfW) * (st.dVC * mis(bsdfRevPdfW) + st.dVCM + o.misVmWeightFactor);
        st.dVM = mis(cosThetaOut / bsdfDirPdfW) * (st.dVM * mis(bsdfRevPdfW) + st.dVCM * o.misVcWeightFactor + 1.f);
        st.dVCM = mis(1.f / bsdfDirPdfW);
        st.specularPath &= 0;
    }
    st.origin = hitPoint;
    st.throughput = st.throughput * (bsdfFactor * (cosThetaOut / bsdfDirPdfW));
    return true;
}

/* one light sub-path: vertexcm.hxx:323-395 */
static void trace_light_path(Oracle &o, int pathIdx, std::vector<LightVertex> &verts,
                             std::vector<Splat> &splats, unsigned char &rngCount, vcm_stats &stats)
{
    const OScene &sc = o.sc;
    PathRngRef rng;
    path_rng_init_ref(&rng, (uint32_t)o.seed, (uint32_t)o.iterations, (uint32_t)pathIdx, 0u);
    SubPathState st;
    generate_light_sample(o, rng, st);
    for (;; ++st.pathLength) {
        Ray ray; ray.org = st.origin + st.direction * O_EPS_RAY; ray.dir = st.direction; ray.tmin = 0;
        Isect isect; isect.dist = 1e36f;
        stats.lightRays++;
        if (!scene_intersect(sc, ray, isect)) break;
        const V3 hitPoint = ray.org + ray.dir * isect.dist;
        isect.dist += O_EPS_RAY;
        Bsdf bsdf;
        bsdf_setup(bsdf, ray, isect, sc);
        if (bsdf.matID < 0) break;
        {
            if (st.pathLength > 1 || st.isFiniteLight == 1) st.dVCM *= mis(osqr(isect.dist));
            st.dVCM /= mis(fabsf(bsdf.localDirFix.z));
            st.dVC  /= mis(fabsf(bsdf.localDirFix.z));
            st.dVM  /= mis(fabsf(bsdf.localDirFix.z));
        }
        if (!bsdf.isDelta && (o.useVC || o.useVM)) {
            LightVertex lv;
            lv.hitpoint = hitPoint;
            lv.throughput = st.throughput;
            lv.pathLength = st.pathLength;
            lv.bsdf = bsdf;
            lv.dVCM = st.dVCM; lv.dVC = st.dVC; lv.dVM = st.dVM;
            verts.push_back(lv);
        }
        if (!bsdf.isDelta && (o.useVC || o.lightTraceOnly)) {
            if (st.pathLength + 1 >= o.minLen) connect_to_camera(o, st, hitPoint, bsdf, splats, stats);
        }
        if (st.pathLength + 2 > o.maxLen) break;
        if (!sample_scattering<true>(o, rng, bsdf, hitPoint, st)) break;
    }
    rngCount = (unsigned char)rng.k;
}

/* merge record = the part of LightVertex that RangeQuery::Process reads */
static inline void make_record(const LightVertex &lv, float *r)
{
    r[0] = lv.hitpoint.x; r[1] = lv.hitpoint.y; r[2] = lv.hitpoint.z;
    const V3 wd = to_world(lv.bsdf.frame, lv.bsdf.localDirFix);   /* BSDF::WorldDirFix bsdf.hxx:264 */
    r[3] = wd.x; r[4] = wd.y; r[5] = wd.z;
    r[6] = lv.throughput.x; r[7] = lv.throughput.y; r[8] = lv.throughput.z;
    r[9] = lv.dVCM; r[10] = lv.dVM; r[11] = lv.bsdf.contProb;
    const uint32_t len = lv.pathLength;
    memcpy(&r[12], &len, 4);
}

/* ---- HashGrid: src/hashgrid.hxx */
static inline int grid_cell_index_i(const Oracle &o, int cx, int cy, int cz)
{   /* :179-187 */
    const unsigned x = unsigned(cx), y = unsigned(cy), z = unsigned(cz);
    return int(((x * 73856093) ^ (y * 19349663) ^ (z * 83492791)) % unsigned(o.cellEnds.size()));
}
static inline int grid_cell_index_p(const Oracle &o, V3 p)
{   /* :189-201 */
    const V3 distMin = p - o.bboxMin;
    const float fx = floorf(o.invCellSize * distMin.x);
    const float fy = floorf(o.invCellSize * distMin.y);
    const float fz = floorf(o.invCellSize * distMin.z);
    return grid_cell_index_i(o, int(fx), int(fy), int(fz));
}
static void grid_build(Oracle &o)
{   /* Reserve :35-38 (nCells = pathCount, vertexcm.hxx:406), Build :41-107 */
    o.cellEnds.assign((size_t)o.N, 0);
    const float radius = o.radius;
    o.gRadiusSqr = osqr(radius);
    o.cellSize = radius * 2.f;
    o.invCellSize = 1.f / o.cellSize;
    o.bboxMin = sp(1e36f);
    o.bboxMax = sp(-1e36f);
    const long long n = o.nRecords;
    const float *rec = o.records.data();
    for (long long i = 0; i < n; i++) {
        const float *p = rec + i * VCM_MERGE_RECORD_FLOATS;
        o.bboxMax.x = omax(o.bboxMax.x, p[0]); o.bboxMin.x = omin(o.bboxMin.x, p[0]);
        o.bboxMax.y = omax(o.bboxMax.y, p[1]); o.bboxMin.y = omin(o.bboxMin.y, p[1]);
        o.bboxMax.z = omax(o.bboxMax.z, p[2]); o.bboxMin.z = omin(o.bboxMin.z, p[2]);
    }
    o.indices.resize((size_t)n);
    for (long long i = 0; i < n; i++) {
        const float *p = rec + i * VCM_MERGE_RECORD_FLOATS;
        o.cellEnds[grid_cell_index_p(o, mk(p[0], p[1], p[2]))]++;
    }
    int sum = 0;
    for (size_t i = 0; i < o.cellEnds.size(); i++) {
        const int temp = o.cellEnds[i];
        o.cellEnds[i] = sum;
        sum += temp;
    }
    for (long long i = 0; i < n; i++) {
        const float *p = rec + i * VCM_MERGE_RECORD_FLOATS;
        const int targetIdx = o.cellEnds[grid_cell_index_p(o, mk(p[0], p[1], p[2]))]++;
        o.indices[targetIdx] = int(i);
    }
}

struct RangeQuery {   /* vertexcm.hxx:109-178 */
    const Oracle *o;
    V3 cameraPosition;
    const Bsdf *cameraBsdf;
    const SubPathState *cameraState;
    V3 contrib;
};
static inline void range_query_process(RangeQuery &q, const float *r, vcm_stats &stats)
{   /* :130-169 */
    const Oracle &o = *q.o;
    uint32_t lvLen; memcpy(&lvLen, &r[12], 4);
    stats.mergeAccepted++;
    if ((lvLen + q.cameraState->pathLength > o.maxLen) || (lvLen + q.cameraState->pathLength < o.minLen)) return;
    const V3 lightDirection = mk(r[3], r[4], r[5]);
    float cosCamera, cameraBsdfDirPdfW, cameraBsdfRevPdfW;
    const V3 cameraBsdfFactor = bsdf_evaluate(*q.cameraBsdf, o.sc, lightDirection, cosCamera,
                                              &cameraBsdfDirPdfW, &cameraBsdfRevPdfW);
    if (iszero(cameraBsdfFactor)) return;
    cameraBsdfDirPdfW *= q.cameraBsdf->contProb;
    cameraBsdfRevPdfW *= r[11];
    const float wLight = r[9] * o.misVcWeightFactor + r[10] * mis(cameraBsdfDirPdfW);
    const float wCamera = q.cameraState->dVCM * o.misVcWeightFactor + q.cameraState->dVM * mis(cameraBsdfRevPdfW);
    const float misWeight = o.ppm ? 1.f : 1.f / (wLight + 1.f + wCamera);
    q.contrib = q.contrib + misWeight * cameraBsdfFactor * mk(r[6], r[7], r[8]);
}
static void grid_process(const Oracle &o, RangeQuery &q, vcm_stats &stats)
{   /* hashgrid.hxx:110-169 */
    const V3 queryPos = q.cameraPosition;
    const V3 distMin = queryPos - o.bboxMin;
    const V3 distMax = o.bboxMax - queryPos;
    if (distMin.x < 0.f || distMax.x < 0.f) return;
    if (distMin.y < 0.f || distMax.y < 0.f) return;
    if (distMin.z < 0.f || distMax.z < 0.f) return;
    const V3 cellPt = o.invCellSize * distMin;
    const V3 coordF = mk(floorf(cellPt.x), floorf(cellPt.y), floorf(cellPt.z));
    const int px = int(coordF.x), py = int(coordF.y), pz = int(coordF.z);
    const V3 fractCoord = cellPt - coordF;
    const int pxo = px + (fractCoord.x < 0.5f ? -1 : +1);
    const int pyo = py + (fractCoord.y < 0.5f ? -1 : +1);
    const int pzo = pz + (fractCoord.z < 0.5f ? -1 : +1);
    for (int j = 0; j < 8; j++) {
        const int cx = (j & 4) ? pxo : px;
        const int cy = (j & 2) ? pyo : py;
        const int cz = (j & 1) ? pzo : pz;
        const int cell = grid_cell_index_i(o, cx, cy, cz);
        int lo = (cell == 0) ? 0 : o.cellEnds[cell - 1];   /* GetCellRange :173-177 */
        const int hi = o.cellEnds[cell];
        for (; lo < hi; lo++) {
            const int particleIndex = o.indices[lo];
            const float *r = o.records.data() + (size_t)particleIndex * VCM_MERGE_RECORD_FLOATS;
            const float distSqr = lensqr(queryPos - mk(r[0], r[1], r[2]));
            stats.mergeCandidates++;
            if (distSqr <= o.gRadiusSqr) range_query_process(q, r, stats);
        }
    }
}

/* ---- camera side: vertexcm.hxx:564-809 */
static V3 get_light_radiance(const Oracle &o, const vcm_light &light, const SubPathState &st, V3 hitpoint, V3 rayDir)
{   /* :617-658 */
    const int lightCount = o.sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    float directPdfA, emissionPdfW;
    const V3 radiance = light_get_radiance(light, o.sc, rayDir, hitpoint, &directPdfA, &emissionPdfW);
    if (iszero(radiance)) return sp(0);
    if (st.pathLength == 1) return radiance;
    if (o.useVM && !o.useVC) return st.specularPath ? radiance : sp(0);
    directPdfA *= lightPickProb;
    emissionPdfW *= lightPickProb;
    const float wCamera = mis(directPdfA) * st.dVCM + mis(emissionPdfW) * st.dVC;
    const float misWeight = 1.f / (1.f + wCamera);
    return misWeight * radiance;
}
static V3 direct_illumination(Oracle &o, PathRngRef &rng, const SubPathState &st, V3 hitpoint, const Bsdf &bsdf,
                              vcm_stats &stats)
{   /* :663-738 */
    const OScene &sc = o.sc;
    const int lightCount = sc.nLights;
    const float lightPickProb = 1.f / lightCount;
    const int lightID = int(path_rng_float_ref(&rng) * lightCount);
    const float rx = path_rng_float_ref(&rng);
    const float ry = path_rng_float_ref(&rng);
    const vcm_light &light = get_light(sc, lightID);
    V3 directionToLight;
    float distance, directPdfW, emissionPdfW, cosAtLight;
    const V3 radiance = light_illuminate(light, sc, hitpoint, rx, ry, directionToLight, distance, directPdfW,
                                         &emissionPdfW, &cosAtLight);
    if (iszero(radiance)) return sp(0);
    float bsdfDirPdfW, bsdfRevPdfW, cosToLight;
    const V3 bsdfFactor = bsdf_evaluate(bsdf, sc, directionToLight, cosToLight, &bsdfDirPdfW, &bsdfRevPdfW);
    if (iszero(bsdfFactor)) return sp(0);
    const float continuationProbability = bsdf.contProb;
    bsdfDirPdfW *= light_is_delta(light) ? 0.f : continuationProbability;
    bsdfRevPdfW *= continuationProbability;
    const float wLight = mis(bsdfDirPdfW / (lightPickProb * directPdfW));
    const float wCamera = mis(emissionPdfW * cosToLight / (directPdfW * cosAtLight)) *
                          (o.misVmWeightFactor + st.dVCM + st.dVC * mis(bsdfRevPdfW));
    const float misWeight = 1.f / (wLight + 1.f + wCamera);
    const V3 contrib = (misWeight * cosToLight / (lightPickProb * directPdfW)) * (radiance * bsdfFactor);
    if (iszero(contrib)) return sp(0);
    stats.shadowRays++;
    if (scene_occluded(sc, hitpoint, directionToLight, distance)) return sp(0);
    return contrib;
}
static V3 connect_vertices(Oracle &o, const LightVertex &lv, const Bsdf &cameraBsdf, V3 cameraHitpoint,
                           const SubPathState &st, vcm_stats &stats)
{   /* :743-809 */
    const OScene &sc = o.sc;
    stats.connections++;
    V3 direction = lv.hitpoint - cameraHitpoint;
    const float dist2 = lensqr(direction);
    const float distance = sqrtf(dist2);
    direction = direction / distance;
    float cosCamera, cameraBsdfDirPdfW, cameraBsdfRevPdfW;
    const V3 cameraBsdfFactor = bsdf_evaluate(cameraBsdf, sc, direction, cosCamera, &cameraBsdfDirPdfW, &cameraBsdfRevPdfW);
    if (iszero(cameraBsdfFactor)) return sp(0);
    const float cameraCont = cameraBsdf.contProb;
    cameraBsdfDirPdfW *= cameraCont;
    cameraBsdfRevPdfW *= cameraCont;
    float cosLight, lightBsdfDirPdfW, lightBsdfRevPdfW;
    const V3 lightBsdfFactor = bsdf_evaluate(lv.bsdf, sc, -direction, cosLight, &lightBsdfDirPdfW, &lightBsdfRevPdfW);
    if (iszero(lightBsdfFactor)) return sp(0);
    const float lightCont = lv.bsdf.contProb;
    lightBsdfDirPdfW *= lightCont;
    lightBsdfRevPdfW *= lightCont;
    const float geometryTerm = cosLight * cosCamera / dist2;
    if (geometryTerm < 0) return sp(0);
    const float cameraBsdfDirPdfA = pdf_w_to_a(cameraBsdfDirPdfW, distance, cosLight);
    const float lightBsdfDirPdfA = pdf_w_to_a(lightBsdfDirPdfW, distance, cosCamera);
    const float wLight = mis(cameraBsdfDirPdfA) * (o.misVmWeightFactor + lv.dVCM + lv.dVC * mis(lightBsdfRevPdfW));
    const float wCamera = mis(lightBsdfDirPdfA) * (o.misVmWeightFactor + st.dVCM + st.dVC * mis(cameraBsdfRevPdfW));
    const float misWeight = 1.f / (wLight + 1.f + wCamera);
    const V3 contrib = (misWeight * geometryTerm) * cameraBsdfFactor * lightBsdfFactor;
    if (iszero(contrib)) return sp(0);
    stats.shadowRays++;
    if (scene_occluded(sc, cameraHitpoint, direction, distance)) return sp(0);
    return contrib;
}

/* one camera sub-path: vertexcm.hxx:417-544 */
static void trace_camera_path(Oracle &o, int pathIdx, unsigned char &rngCount, vcm_stats &stats)
{
    const OScene &sc = o.sc;
    const vcm_camera &cam = sc.camera;
    PathRngRef rng;
    path_rng_init_ref(&rng, (uint32_t)o.seed, (uint32_t)o.iterations, (uint32_t)pathIdx, 1u);

    /* GenerateCameraSample :564-606 */
    SubPathState st;
    const int x = pathIdx % o.resX;
    const int y = pathIdx / o.resX;
    const float jx = path_rng_float_ref(&rng);
    const float jy = path_rng_float_ref(&rng);
    const float sx = float(x) + jx, sy = float(y) + jy;
    {
        const V3 worldRaster = transform_point(cam.rasterToWorld, mk(sx, sy, 0));   /* camera.hxx:108-117 */
        const V3 org = ld3(cam.position);
        const V3 dir = normalize(worldRaster - org);
        const float cosAtCamera = dot(ld3(cam.forward), dir);
        const float imagePointToCameraDist = cam.imagePlaneDist / cosAtCamera;
        const float imageToSolidAngleFactor = osqr(imagePointToCameraDist) / cosAtCamera;
        const float cameraPdfW = imageToSolidAngleFactor;
        st.origin = org;
        st.direction = dir;
        st.throughput = sp(1);
        st.pathLength = 1;
        st.specularPath = 1;
        st.isFiniteLight = 0;
        st.dVCM = mis(o.lightSubPathCount / cameraPdfW);
        st.dVC = 0;
        st.dVM = 0;
    }
    V3 color = sp(0);
    const int lp = pathIdx - o.p0;   /* local index */

    for (;; ++st.pathLength) {
        Ray ray; ray.org = st.origin + st.direction * O_EPS_RAY; ray.dir = st.direction; ray.tmin = 0;
        Isect isect; isect.dist = 1e36f;
        stats.cameraRays++;
        if (!scene_intersect(sc, ray, isect)) {
            if (sc.backgroundLight >= 0) {
                if (st.pathLength >= o.minLen)
                    color = color + st.throughput * get_light_radiance(o, sc.lights[sc.backgroundLight], st, sp(0), ray.dir);
            }
            break;
        }
        const V3 hitPoint = ray.org + ray.dir * isect.dist;
        isect.dist += O_EPS_RAY;
        Bsdf bsdf;
        bsdf_setup(bsdf, ray, isect, sc);
        if (bsdf.matID < 0) break;
        {
            st.dVCM *= mis(osqr(isect.dist));
            st.dVCM /= mis(fabsf(bsdf.localDirFix.z));
            st.dVC  /= mis(fabsf(bsdf.localDirFix.z));
            st.dVM  /= mis(fabsf(bsdf.localDirFix.z));
        }
        if (isect.lightID >= 0) {
            const vcm_light &light = get_light(sc, isect.lightID);
            if (st.pathLength >= o.minLen)
                color = color + st.throughput * get_light_radiance(o, light, st, hitPoint, ray.dir);
            break;
        }
        if (st.pathLength >= o.maxLen) break;

        if (!bsdf.isDelta && o.useVC) {
            if (st.pathLength + 1 >= o.minLen)
                color = color + st.throughput * direct_illumination(o, rng, st, hitPoint, bsdf, stats);
        }
        if (!bsdf.isDelta && o.useVC) {
            const int r0 = (lp == 0) ? 0 : o.pathEnds[lp - 1];
            const int r1 = o.pathEnds[lp];
            for (int i = r0; i < r1; i++) {
                const LightVertex &lv = o.lightVertices[i];
                if (lv.pathLength + 1 + st.pathLength < o.minLen) continue;
                if (lv.pathLength + 1 + st.pathLength > o.maxLen) break;
                color = color + st.throughput * lv.throughput * connect_vertices(o, lv, bsdf, hitPoint, st, stats);
            }
        }
        if (!bsdf.isDelta && o.useVM) {
            RangeQuery q; q.o = &o; q.cameraPosition = hitPoint; q.cameraBsdf = &bsdf; q.cameraState = &st; q.contrib = sp(0);
            stats.mergeQueries++;
            grid_process(o, q, stats);
            color = color + st.throughput * o.vmNormalization * q.contrib;
            if (o.ppm) break;
        }
        if (!sample_scattering<false>(o, rng, bsdf, hitPoint, st)) break;
    }
    /* Framebuffer::AddColor(screenSample, color) :544, framebuffer.hxx:43-57.
       The pixel is derived from the JITTERED sample: float(x)+jx can round up
       to x+1 (then the colour lands in the next pixel, or is dropped at the
       right/bottom edge).  Deferred: the caller adds in path order. */
    int target = -1;
    if (!(sx < 0 || sx >= cam.resolution[0]) && !(sy < 0 || sy >= cam.resolution[1]))
        target = int(sx) + int(sy) * o.resX;
    o.camTarget[lp] = target;
    o.camColor[(size_t)lp * 3 + 0] = color.x;
    o.camColor[(size_t)lp * 3 + 1] = color.y;
    o.camColor[(size_t)lp * 3 + 2] = color.z;
    rngCount = (unsigned char)rng.k;
}

/* ================= PathTracer::RunIteration, src/pathtracer.hxx:45-215 ================= */
static inline float mis2(float samplePdf, float otherPdf) { return mis(samplePdf) / (mis(samplePdf) + mis(otherPdf)); }  /* :226-231 */
static inline float pdf_a_to_w(float pdfA, float dist, float cosThere) { return pdfA * osqr(dist) / std::abs(cosThere); }  /* utils.hxx:253-259 */

static void store_path_colour(Oracle &o, int lp, float sx, float sy, V3 color, bool add)
{   /* Framebuffer::AddColor(sample, color), deferred: the caller adds in pixel-loop order */
    const vcm_camera &cam = o.sc.camera;
    int target = -1;
    if (add && !(sx < 0 || sx >= cam.resolution[0]) && !(sy < 0 || sy >= cam.resolution[1]))
        target = int(sx) + int(sy) * o.resX;
    o.camTarget[lp] = target;
    o.camColor[(size_t)lp * 3 + 0] = color.x;
    o.camColor[(size_t)lp * 3 + 1] = color.y;
    o.camColor[(size_t)lp * 3 + 2] = color.z;
}

static void trace_pt_path(Oracle &o, int pixID, unsigned char &rngCount, vcm_stats &stats)
{
    const OScene &sc = o.sc;
    const vcm_camera &cam = sc.camera;
    PathRngRef rng;
    path_rng_init_ref(&rng, (uint32_t)o.seed, (uint32_t)o.iterations, (uint32_t)pixID, 1u);
    const int lightCount = sc.nLights;                 /* :48-49 */
    const float lightPickProb = 1.f / lightCount;
    const int x = pixID % o.resX, y = pixID / o.resX;  /* :56-57 */
    const float jx = path_rng_float_ref(&rng);
    const float jy = path_rng_float_ref(&rng);
    const float sx = float(x) + jx, sy = float(y) + jy;   /* :59 */
    Ray ray;                                            /* Camera::GenerateRay camera.hxx:108-117 */
    ray.org = ld3(cam.position);
    ray.dir = normalize(transform_point(cam.rasterToWorld, mk(sx, sy, 0)) - ray.org);
    ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp(0);
    V3 pathWeight = sp(1), color = sp(0);
    unsigned pathLength = 1;
    bool lastSpecular = true;
    float lastPdfW = 1;
    for (;; ++pathLength) {
        stats.cameraRays++;
        if (!scene_intersect(sc, ray, isect)) {          /* :73-97 */
            if (pathLength < o.minLen) break;
            if (sc.backgroundLight < 0) break;
            float directPdfW;
            const V3 contrib = light_get_radiance(sc.lights[sc.backgroundLight], sc, ray.dir, sp(0), &directPdfW, NULL);
            if (iszero(contrib)) break;
            float misWeight = 1.f;
            if (pathLength > 1 && !lastSpecular) misWeight = mis2(lastPdfW, directPdfW * lightPickProb);
            color = color + pathWeight * misWeight * contrib;
            break;
        }
        const V3 hitPoint = ray.org + ray.dir * isect.dist;   /* :99-100 */
        isect.dist += O_EPS_RAY;
        Bsdf bsdf;
        bsdf_setup(bsdf, ray, isect, sc);
        if (bsdf.matID < 0) break;
        if (isect.lightID >= 0) {                        /* :107-129 */
            if (pathLength < o.minLen) break;
            const vcm_light &light = get_light(sc, isect.lightID);
            float directPdfA;
            const V3 contrib = light_get_radiance(light, sc, ray.dir, hitPoint, &directPdfA, NULL);
            if (iszero(contrib)) break;
            float misWeight = 1.f;
            if (pathLength > 1 && !lastSpecular) {
                const float directPdfW = pdf_a_to_w(directPdfA, isect.dist, bsdf.localDirFix.z);
                misWeight = mis2(lastPdfW, directPdfW * lightPickProb);
            }
            color = color + pathWeight * misWeight * contrib;
            break;
        }
        if (pathLength >= o.maxLen) break;               /* :131 */
        if (bsdf.contProb == 0) break;                   /* :134 */
        if (!bsdf.isDelta && pathLength + 1 >= o.minLen) {   /* :138-179 */
            const int lightID = int(path_rng_float_ref(&rng) * lightCount);
            const vcm_light &light = get_light(sc, lightID);
            const float rx = path_rng_float_ref(&rng);
            const float ry = path_rng_float_ref(&rng);
            V3 directionToLight;
            float distance, directPdfW;
            const V3 radiance = light_illuminate(light, sc, hitPoint, rx, ry, directionToLight, distance, directPdfW, NULL, NULL);
            if (!iszero(radiance)) {
                float bsdfPdfW, cosThetaOut;
                const V3 factor = bsdf_evaluate(bsdf, sc, directionToLight, cosThetaOut, &bsdfPdfW, NULL);
                if (!iszero(factor)) {
                    float weight = 1.f;
                    if (!light_is_delta(light)) {
                        const float contProb = bsdf.contProb;
                        bsdfPdfW *= contProb;
                        weight = mis2(directPdfW * lightPickProb, bsdfPdfW);
                    }
                    const V3 contrib = (weight * cosThetaOut / (lightPickProb * directPdfW)) * (radiance * factor);
                    stats.shadowRays++;
                    if (!scene_occluded(sc, hitPoint, directionToLight, distance)) color = color + pathWeight * contrib;
                }
            }
        }
        {   /* :182-212 */
            const float r0 = path_rng_float_ref(&rng);
            const float r1 = path_rng_float_ref(&rng);
            const float r2 = path_rng_float_ref(&rng);
            float pdf, cosThetaOut;
            unsigned sampledEvent;
            const V3 factor = bsdf_sample(bsdf, sc, false, mk(r0, r1, r2), ray.dir, pdf, cosThetaOut, sampledEvent);
            if (iszero(factor)) break;
            const float contProb = bsdf.contProb;
            lastSpecular = (sampledEvent & kSpecular) != 0;
            lastPdfW = pdf * contProb;
            if (contProb < 1.f) {
                if (path_rng_float_ref(&rng) > contProb) break;
                pdf *= contProb;
            }
            pathWeight = pathWeight * (factor * (cosThetaOut / pdf));
            ray.org = hitPoint + O_EPS_RAY * ray.dir;
            ray.tmin = 0.f;
            isect.dist = 1e36f;
        }
    }
    store_path_colour(o, pixID - o.p0, sx, sy, color, true);   /* :214 */
    rngCount = (unsigned char)rng.k;
}

/* ================= EyeLight::RunIteration, src/eyelight.hxx:46-77 ================= */
static void trace_eyelight_path(Oracle &o, int pixID, unsigned char &rngCount, vcm_stats &stats)
{
    const OScene &sc = o.sc;
    const vcm_camera &cam = sc.camera;
    PathRngRef rng;
    path_rng_init_ref(&rng, (uint32_t)o.seed, (uint32_t)o.iterations, (uint32_t)pixID, 1u);
    const int x = pixID % o.resX, y = pixID / o.resX;
    float jx = 0.5f, jy = 0.5f;
    if (o.iteration != 1) { jx = path_rng_float_ref(&rng); jy = path_rng_float_ref(&rng); }   /* :60-61 */
    const float sx = float(x) + jx, sy = float(y) + jy;
    Ray ray;
    ray.org = ld3(cam.position);
    ray.dir = normalize(transform_point(cam.rasterToWorld, mk(sx, sy, 0)) - ray.org);
    ray.tmin = 0;
    Isect isect; isect.dist = 1e36f; isect.matID = 0; isect.lightID = -1; isect.normal = sp(0);
    stats.cameraRays++;
    V3 color = sp(0);
    bool hit = false;
    if (scene_intersect(sc, ray, isect)) {               /* :67-76 */
        const float dotLN = dot(isect.normal, -ray.dir);
        color = (dotLN > 0) ? sp(dotLN) : mk(-dotLN, 0, 0);
        hit = true;
    }
    store_path_colour(o, pixID - o.p0, sx, sy, color, hit);
    rngCount = (unsigned char)rng.k;
}

static void add_stats(vcm_stats &a, const vcm_stats &b)
{
    a.lightRays += b.lightRays; a.cameraRays += b.cameraRays; a.shadowRays += b.shadowRays;
    a.mergeQueries += b.mergeQueries; a.mergeCandidates += b.mergeCandidates; a.mergeAccepted += b.mergeAccepted;
    a.connections += b.connections; a.lightSplats += b.lightSplats;
}

} // namespace

template <typename Desc> static void oscene_fill(OScene &s, const Desc &d)
{
    s.nPrims = d.nPrims; s.nMaterials = d.nMaterials; s.nLights = d.nLights; s.backgroundLight = d.backgroundLight;
    s.prims.assign(d.prims, d.prims + d.nPrims);
    s.materials.assign(d.materials, d.materials + d.nMaterials);
    s.mat2light.assign(d.mat2light, d.mat2light + d.nMaterials);
    s.lights.assign(d.lights, d.lights + d.nLights);
    for (int k = 0; k < 3; k++) s.sceneCenter[k] = d.sceneCenter[k];
    s.sceneRadius = d.sceneRadius; s.invSceneRadiusSqr = d.invSceneRadiusSqr;
    s.camera = d.camera;
}
/* ======================= C API (ctypes) ================================ */
extern "C" {

static void *oracle_finish_create(Oracle *o, int algorithm, float radiusFactor, float radiusAlpha, int seed, int rank, int world);
void *oracle_create(const vcm_scene_desc *scene, int algorithm, float radiusFactor, float radiusAlpha,
                    int seed, int rank, int world)
{
    Oracle *o = new Oracle();
    oscene_fill(o->sc, *scene);
    return oracle_finish_create(o, algorithm, radiusFactor, radiusAlpha, seed, rank, world);
}
void *oracle_create2(const vcm_scene_desc2 *scene, int algorithm, float radiusFactor, float radiusAlpha,
                     int seed, int rank, int world)
{
    Oracle *o = new Oracle();
    oscene_fill(o->sc, *scene);
    return oracle_finish_create(o, algorithm, radiusFactor, radiusAlpha, seed, rank, world);
}
static void *oracle_finish_create(Oracle *o, int algorithm, float radiusFactor, float radiusAlpha, int seed, int rank, int world)
{   /* VertexCM::VertexCM vertexcm.hxx:208-282 */
    const OScene *scene = &o->sc;
    o->useVM = o->useVC = o->lightTraceOnly = o->ppm = false;
    o->renderer = 0; o->iteration = 0;
    switch (algorithm) {
    case VCM_ALGO_LIGHT_TRACE: o->lightTraceOnly = true; break;
    case VCM_ALGO_PPM: o->ppm = true; o->useVM = true; break;
    case VCM_ALGO_BPM: o->useVM = true; break;
    case VCM_ALGO_BPT: o->useVC = true; break;
    case VCM_ALGO_VCM: o->useVC = true; o->useVM = true; break;
    case VCM_ALGO_PATH_TRACE: o->renderer = 1; break;   /* config.hxx:120-121 */
    case VCM_ALGO_EYE_LIGHT: o->renderer = 2; break;    /* config.hxx:118-119 */
    default: break;
    }
    if (o->ppm) {   /* :246-278 */
        for (int i = 0; i < scene->nMaterials; i++) {
            const vcm_material &m = scene->materials[i];
            const bool hasNonSpecular = (vmax(ld3(m.diffuse)) > 0) || (vmax(ld3(m.phong)) > 0);
            const bool hasSpecular = (vmax(ld3(m.mirror)) > 0) || (m.ior > 0);
            if (hasNonSpecular && hasSpecular) { o->ppm = false; break; }
        }
    }
    o->baseRadius = radiusFactor * scene->sceneRadius;   /* :280 */
    o->radiusAlpha = radiusAlpha;
    o->seed = seed;
    o->rank = rank; o->world = world;
    o->resX = int(scene->camera.resolution[0]);
    o->resY = int(scene->camera.resolution[1]);
    o->N = o->resX * o->resY;
    o->p0 = (int)((long long)o->N * rank / world);
    o->p1 = (int)((long long)o->N * (rank + 1) / world);
    o->threads = 1;
    o->iterations = 0;
    o->fb.assign((size_t)o->N * 3, 0.f);
    o->lightCounts.assign((size_t)(o->p1 - o->p0), 0);
    o->camCounts.assign((size_t)(o->p1 - o->p0), 0);
    o->nRecords = 0;
    memset(&o->st, 0, sizeof(o->st));
    return o;
}
void oracle_destroy(void *h) { delete (Oracle *)h; }
void oracle_set_threads(void *h, int n) { ((Oracle *)h)->threads = n < 1 ? 1 : n; }
int oracle_max_threads(void)
{
#ifdef _OPENMP
    return omp_get_num_procs();
#else
    return 1;
#endif
}

void oracle_begin_iteration(void *h, int iteration, unsigned minLen, unsigned maxLen)
{   /* vertexcm.hxx:288-316 */
    Oracle &o = *(Oracle *)h;
    o.minLen = minLen; o.maxLen = maxLen;
    o.iteration = iteration;
    o.lightSubPathCount = float(o.resX * o.resY);
    float radius = o.baseRadius;
    radius /= dmr_powf(float(iteration + 1), 0.5f * (1 - o.radiusAlpha));
    radius = omax(radius, 1e-7f);
    const float radiusSqr = osqr(radius);
    o.radius = radius; o.radiusSqr = radiusSqr;
    o.vmNormalization = 1.f / (radiusSqr * O_PI_F * o.lightSubPathCount);
    const float etaVCM = (O_PI_F * radiusSqr) * o.lightSubPathCount;
    o.misVmWeightFactor = o.useVM ? mis(etaVCM) : 0.f;
    o.misVcWeightFactor = o.useVC ? mis(1.f / etaVCM) : 0.f;
    o.lightVertices.clear();
    o.pathEnds.assign((size_t)(o.p1 - o.p0), 0);
    o.nRecords = 0;
    memset(&o.st, 0, sizeof(o.st));
    o.st.radius = radius;
}

void oracle_trace_light(void *h)
{   /* vertexcm.hxx:321-396 over the local path range; chunked so that the
       OpenMP variant produces exactly the serial result */
    Oracle &o = *(Oracle *)h;
    if (o.renderer) return;   /* PathTracer / EyeLight: no light pass */
    const int nLocal = o.p1 - o.p0;
    const int CH = 1024;
    const int nChunks = (nLocal + CH - 1) / CH;
    std::vector<std::vector<LightVertex> > cv((size_t)nChunks);
    std::vector<std::vector<Splat> > cs((size_t)nChunks);
    std::vector<vcm_stats> cst((size_t)nChunks);
    std::vector<int> counts((size_t)nLocal, 0);
#pragma omp parallel for schedule(dynamic, 1) num_threads(o.threads)
    for (int c = 0; c < nChunks; c++) {
        memset(&cst[c], 0, sizeof(vcm_stats));
        const int b = c * CH, e = std::min(nLocal, b + CH);
        for (int lp = b; lp < e; lp++) {
            const size_t before = cv[c].size();
            trace_light_path(o, o.p0 + lp, cv[c], cs[c], o.lightCounts[lp], cst[c]);
            counts[lp] = (int)(cv[c].size() - before);
        }
    }
    /* serial merge, reference order */
    int end = 0;
    for (int lp = 0; lp < nLocal; lp++) { end += counts[lp]; o.pathEnds[lp] = end; }   /* :395 */
    o.lightVertices.reserve((size_t)end);
    for (int c = 0; c < nChunks; c++) {
        o.lightVertices.insert(o.lightVertices.end(), cv[c].begin(), cv[c].end());
        for (size_t i = 0; i < cs[c].size(); i++) {
            float *px = &o.fb[(size_t)cs[c][i].pixel * 3];
            px[0] = px[0] + cs[c][i].c.x; px[1] = px[1] + cs[c][i].c.y; px[2] = px[2] + cs[c][i].c.z;
        }
        add_stats(o.st, cst[c]);
    }
    o.st.lightVertices = (long long)o.lightVertices.size();
    /* local merge records */
    o.nRecords = (long long)o.lightVertices.size();
    o.records.resize((size_t)o.nRecords * VCM_MERGE_RECORD_FLOATS);
    for (long long i = 0; i < o.nRecords; i++)
        make_record(o.lightVertices[(size_t)i], &o.records[(size_t)i * VCM_MERGE_RECORD_FLOATS]);
}

long long oracle_light_record_count(void *h) { return ((Oracle *)h)->nRecords; }
void oracle_export_light_records(void *h, float *out)
{
    Oracle &o = *(Oracle *)h;
    memcpy(out, o.records.data(), (size_t)o.nRecords * VCM_MERGE_RECORD_FLOATS * sizeof(float));
}
void oracle_import_light_records(void *h, const float *recs, long long total)
{
    Oracle &o = *(Oracle *)h;
    o.nRecords = total;
    o.records.assign(recs, recs + (size_t)total * VCM_MERGE_RECORD_FLOATS);
}

void oracle_build_grid(void *h)
{   /* vertexcm.hxx:403-408 */
    Oracle &o = *(Oracle *)h;
    o.st.gridVertices = o.nRecords;
    if (o.useVM) grid_build(o);
}

/* rowStride > 1 traces only pixel rows y % rowStride == 0 (bounded CPU
   baseline sample for bench.py); 1 = the full pass. */
/* rows y with (y mod rowStride) < rowWidth only: a bounded sample of the camera pass for the CPU baseline
 * (bench.py) and for full-size parity checks.  A pixel of row y collects the colours of paths of rows y-1
 * and y (jittered AddColor), so with rowWidth >= 2 the rows with (y mod rowStride) in [1, rowWidth) are complete. */
void oracle_trace_camera_window(void *h, int rowStride, int rowWidth)
{   /* vertexcm.hxx:415-545 */
    Oracle &o = *(Oracle *)h;
    if (o.lightTraceOnly) return;
    const int nLocal = o.p1 - o.p0;
    const int CH = 256;
    const int nChunks = (nLocal + CH - 1) / CH;
    std::vector<vcm_stats> cst((size_t)nChunks);
    o.camColor.assign((size_t)nLocal * 3, 0.f);
    o.camTarget.assign((size_t)nLocal, -1);
#pragma omp parallel for schedule(dynamic, 1) num_threads(o.threads)
    for (int c = 0; c < nChunks; c++) {
        memset(&cst[c], 0, sizeof(vcm_stats));
        const int b = c * CH, e = std::min(nLocal, b + CH);
        for (int lp = b; lp < e; lp++) {
            const int p = o.p0 + lp;
            if (rowStride > 1 && ((p / o.resX) % rowStride) >= rowWidth) continue;
            if (o.renderer == 1) trace_pt_path(o, p, o.camCounts[lp], cst[c]);
            else if (o.renderer == 2) trace_eyelight_path(o, p, o.camCounts[lp], cst[c]);
            else trace_camera_path(o, p, o.camCounts[lp], cst[c]);
        }
    }
    for (int c = 0; c < nChunks; c++) add_stats(o.st, cst[c]);
    /* the AddColor calls of :544, in path order */
    for (int lp = 0; lp < nLocal; lp++) {
        const int t = o.camTarget[lp];
        if (t < 0) continue;
        float *px = &o.fb[(size_t)t * 3];
        px[0] = px[0] + o.camColor[(size_t)lp * 3 + 0];
        px[1] = px[1] + o.camColor[(size_t)lp * 3 + 1];
        px[2] = px[2] + o.camColor[(size_t)lp * 3 + 2];
    }
}
void oracle_trace_camera_rows(void *h, int rowStride) { oracle_trace_camera_window(h, rowStride, 1); }
void oracle_trace_camera(void *h) { oracle_trace_camera_window(h, 1, 1); }

void oracle_end_iteration(void *h) { ((Oracle *)h)->iterations++; }   /* :547 */

void oracle_run_iteration(void *h, int iteration, unsigned minLen, unsigned maxLen)
{
    oracle_begin_iteration(h, iteration, minLen, maxLen);
    oracle_trace_light(h);
    oracle_build_grid(h);
    oracle_trace_camera(h);
    oracle_end_iteration(h);
}

void oracle_get_framebuffer(void *h, float *out)
{
    Oracle &o = *(Oracle *)h;
    memcpy(out, o.fb.data(), o.fb.size() * sizeof(float));
}
void oracle_add_framebuffer(void *h, const float *in)
{   /* used by the 2-rank CPU tests to emulate the framebuffer reduce */
    Oracle &o = *(Oracle *)h;
    for (size_t i = 0; i < o.fb.size(); i++) o.fb[i] += in[i];
}
void oracle_get_counts(void *h, unsigned char *light, unsigned char *cam)
{
    Oracle &o = *(Oracle *)h;
    memcpy(light, o.lightCounts.data(), o.lightCounts.size());
    memcpy(cam, o.camCounts.data(), o.camCounts.size());
}
void oracle_get_stats(void *h, vcm_stats *out) { *out = ((Oracle *)h)->st; }
int oracle_iterations(void *h) { return ((Oracle *)h)->iterations; }
void oracle_local_range(void *h, int *first, int *count)
{
    Oracle &o = *(Oracle *)h;
    *first = o.p0; *count = o.p1 - o.p0;
}
/* hash-grid internals for the grid-build parity test */
long long oracle_grid_cells(void *h) { return (long long)((Oracle *)h)->cellEnds.size(); }
void oracle_get_grid(void *h, int *cellEnds, int *indices, float *bbox6)
{
    Oracle &o = *(Oracle *)h;
    if (cellEnds) memcpy(cellEnds, o.cellEnds.data(), o.cellEnds.size() * sizeof(int));
    if (indices) memcpy(indices, o.indices.data(), o.indices.size() * sizeof(int));
    if (bbox6) { bbox6[0] = o.bboxMin.x; bbox6[1] = o.bboxMin.y; bbox6[2] = o.bboxMin.z;
                 bbox6[3] = o.bboxMax.x; bbox6[4] = o.bboxMax.y; bbox6[5] = o.bboxMax.z; }
}

void oracle_world_to_raster(const vcm_scene_desc *sc, int n, const float *pts, float *out)
{
    for (int i = 0; i < n; i++) {
        const V3 r = transform_point(sc->camera.worldToRaster, mk(pts[3*i], pts[3*i+1], pts[3*i+2]));
        out[2*i] = r.x; out[2*i+1] = r.y;
    }
}

/* numeric spec exports (tests compare the product's versions against these) */
float oracle_sinf(float x) { return dmr_sinf(x); }
float oracle_cosf(float x) { return dmr_cosf(x); }
float oracle_powf(float x, float y) { return dmr_powf(x, y); }
void oracle_philox(const uint32_t *ctr, const uint32_t *key, uint32_t *out) { philox4x32_10_ref(ctr, key, out); }
float oracle_path_float(uint32_t seed, uint32_t iter, uint32_t path, uint32_t kind, uint32_t k)
{
    PathRngRef r; path_rng_init_ref(&r, seed, iter, path, kind);
    r.k = k & ~3u;
    float f = 0;
    for (uint32_t i = r.k; i <= k; i++) f = path_rng_float_ref(&r);
    return f;
}

} // extern "C"
