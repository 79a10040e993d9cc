/*
 * TEST INFRASTRUCTURE (oracle) -- not part of the shipped product path.
 *
 * ref_driver.cpp -- harness around the UNMODIFIED reference sources.  It is
 * compiled against /root/reference/src by include path (nothing is copied;
 * see oracle/Makefile) into oracle/_ref/:
 *
 *   libsmallvcm_ref_tape.so   (-DREF_TAPE)
 *       the reference's VertexCM::RunIteration with
 *       (a) `class Rng` replaced WITHOUT editing any reference file: the
 *           include guard __RNG_HXX__ (src/rng.hxx:25-26) is pre-defined and
 *           a tape-replay Rng is supplied.  The tape is the per-path count of
 *           floats the implementation under test consumed; the floats
 *           themselves are the counter-based stream of philox_ref.h.  The
 *           reference therefore sees exactly the random numbers the
 *           implementation under test used, in its own serial order
 *           (light paths 0..N-1, then camera paths 0..N-1:
 *           src/vertexcm.hxx:321, :415);
 *       (b) sinf/cosf/sincosf/powf interposed with detmath_ref.h (linked
 *           -Bsymbolic), see that header for why.
 *       Output: the raw fp32 framebuffer SUM (renderer.hxx:68).
 *
 *   libsmallvcm_ref_tape_libm.so  (-DREF_TAPE -DREF_TAPE_LIBM)
 *       (a) without (b): the taped random numbers, and the image's OWN libm (glibc 2.35) for sinf / cosf / powf --
 *       the reference as shipped but for the random numbers.  Since round 4 detmath restates that libm (one
 *       deviation: integer exponents), so this build measures what is left between the product and the reference's
 *       own arithmetic (tests/test_oracle_vs_reference.py, oracle/libm_tolerance.py).
 *
 *   libsmallvcm_ref_stock.so  (-DREF_STOCK)
 *       the reference exactly as its Makefile builds it (mt19937_64 Rng,
 *       glibc libm), driven by a restatement of render()
 *       (src/smallvcm.cxx:52-151) that takes resolution / threads / seed as
 *       arguments (they are not on the reference CLI: src/config.hxx:233-237)
 *       and measures wall-clock time.  Used for statistical parity and as the
 *       "reference" CPU baseline.
 *
 * Both also export ref_flatten_scene(), which builds the reference's Cornell
 * scenes (src/scene.hxx:132) and flattens them with the product's
 * smallvcm_amd/dropin/flatten_scene.hxx -- the source of tests/golden/scene_*.
 */
#include <vector>
#include <cmath>
#include <map>
#include <set>
#include <string>
#include <sstream>
#include <fstream>
#include <random>
#include <cassert>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <string.h>
#include <time.h>
#include <algorithm>
#include <chrono>
#include <omp.h>
#include <stdint.h>

#include "philox_ref.h"
#include "detmath_ref.h"

#if defined(REF_TAPE) && defined(REF_TAPE_LIBM)
static long long g_detmath_calls = -1;   /* nothing is interposed: the reference calls the image's libm */
#elif defined(REF_TAPE)
/* ---- (b) libm interposition ------------------------------------------- */
static long long g_detmath_calls = 0;
extern "C" {
float sinf(float x) noexcept { g_detmath_calls++; return dmr_sinf(x); }
float cosf(float x) noexcept { g_detmath_calls++; return dmr_cosf(x); }
void  sincosf(float x, float *s, float *c) noexcept { g_detmath_calls++; *s = dmr_sinf(x); *c = dmr_cosf(x); }
float powf(float x, float y) noexcept { g_detmath_calls++; return dmr_powf(x, y); }
}
#endif

/* the harness reads AbstractRenderer::mFramebuffer / Framebuffer::mColor
   (renderer.hxx:68, framebuffer.hxx:255), which have no raw accessor */
#define private public
#define protected public

#include "math.hxx"

#ifdef REF_TAPE
/* ---- (a) tape-replay Rng ---------------------------------------------- */
#define __RNG_HXX__
struct RefTape {
    const unsigned char *counts[2];   /* [0] light, [1] camera; N entries each */
    int N;
    uint32_t key[2];
    int phase, path;
    uint32_t k;
    uint32_t blk[4];
    long long consumed;
    int overrun;
};
static RefTape g_tape;

static inline float tape_next()
{
    RefTape &t = g_tape;
    while (t.phase < 2 && t.k == (uint32_t)t.counts[t.phase][t.path]) {
        t.k = 0;
        t.path++;
        if (t.path == t.N) { t.path = 0; t.phase++; }
    }
    if (t.phase >= 2) { t.overrun++; return 0.5f; }
    if ((t.k & 3u) == 0u) {
        const uint32_t ctr[4] = { (uint32_t)t.path, (uint32_t)t.phase, t.k >> 2, 0u };
        philox4x32_10_ref(ctr, t.key, t.blk);
    }
    const float f = philox_u32_to_float_ref(t.blk[t.k & 3u]);
    t.k++;
    t.consumed++;
    return f;
}

class Rng
{
public:
    Rng(int /*aSeed*/ = 1234) {}
    int   GetInt()   { return int(tape_next() * 2147483647.f); }
    uint  GetUint()  { return uint(tape_next() * 4294967295.f); }
    float GetFloat() { return tape_next(); }
    Vec2f GetVec2f() { float a = GetFloat(); float b = GetFloat(); return Vec2f(a, b); }
    Vec3f GetVec3f() { float a = GetFloat(); float b = GetFloat(); float c = GetFloat(); return Vec3f(a, b, c); }
};
#endif

#include "ray.hxx"
#include "geometry.hxx"
#include "camera.hxx"
#include "framebuffer.hxx"
#include "scene.hxx"
#include "eyelight.hxx"
#include "pathtracer.hxx"
#include "bsdf.hxx"
#include "vertexcm.hxx"
#include "html_writer.hxx"
#include "config.hxx"

#undef private
#undef protected

#include "../smallvcm_amd/dropin/flatten_scene.hxx"
#include "../include/smallvcm_amd_debug.h"

static Scene *make_scene(unsigned boxMask, int resX, int resY)
{   /* as ParseCommandline does: src/config.hxx:366-370 */
    Scene *scene = new Scene;
    scene->LoadCornellBox(Vec2i(resX, resY), boxMask);
    scene->BuildSceneSphere();
    return scene;
}

static Config::Algorithm algo_from_vcm(int vcmAlgo)
{
    switch (vcmAlgo) {
    case VCM_ALGO_LIGHT_TRACE: return Config::kLightTracing;
    case VCM_ALGO_PPM: return Config::kProgressivePhotonMapping;
    case VCM_ALGO_BPM: return Config::kBidirectionalPhotonMapping;
    case VCM_ALGO_BPT: return Config::kBidirectionalPathTracing;
    default: return Config::kVertexConnectionMerging;
    }
}

/* Function-level known answers (T0): the records of include/smallvcm_amd_debug.h
 * (VCM_KAT_*) answered by the reference's OWN classes, one call per record.  Compared bit for bit with the
 * product's device functions on the host (tests/host_emul) and on the device (vcm_debug_kat). */
template <bool FixIsLight>
static void kat_bsdf(const Scene &scene, int op, const float *in, float *out)
{
    Ray ray(Vec3f(0), Vec3f(in[0], in[1], in[2]), 0);
    Isect isect(1e36f);
    isect.normal = Vec3f(in[3], in[4], in[5]);
    isect.matID = (int)in[6];
    isect.lightID = -1;
    BSDF<FixIsLight> bsdf(ray, isect, scene);
    if (!bsdf.IsValid()) return;
    out[0] = 1.f;
    if (op == VCM_KAT_BSDF_EVAL) {
        out[1] = bsdf.IsDelta() ? 1.f : 0.f; out[2] = bsdf.ContinuationProb();
        const Vec3f gen(in[7], in[8], in[9]);
        float cosGen = 0.f, dirPdf = 0.f, revPdf = 0.f;
        const Vec3f f = bsdf.Evaluate(scene, gen, cosGen, &dirPdf, &revPdf);
        out[3] = f.x; out[4] = f.y; out[5] = f.z;
        if (!f.IsZero()) out[6] = cosGen;
        out[7] = dirPdf; out[8] = revPdf;
        out[9] = bsdf.Pdf(scene, gen, false);
        out[10] = bsdf.Pdf(scene, gen, true);
        const Vec3f w = bsdf.WorldDirFix();
        out[11] = w.x; out[12] = w.y; out[13] = w.z; out[14] = bsdf.CosThetaFix();
    } else {
        Vec3f gen(0);
        float pdfW = 0.f, cosGen = 0.f;
        uint ev = 0;
        const Vec3f f = bsdf.Sample(scene, Vec3f(in[7], in[8], in[9]), gen, pdfW, cosGen, &ev);
        if (f.IsZero()) return;
        out[1] = f.x; out[2] = f.y; out[3] = f.z; out[4] = gen.x; out[5] = gen.y; out[6] = gen.z;
        out[7] = pdfW; out[8] = cosGen; out[9] = (float)ev;
    }
}

extern "C" {

unsigned ref_scene_config_mask(int sceneID) { return g_SceneConfigs[sceneID]; }

int ref_flatten_scene(unsigned boxMask, int resX, int resY, vcm_scene_desc *out)
{
    Scene *scene = make_scene(boxMask, resX, resY);
    const int rc = smallvcm_amd::FlattenScene(*scene, *out);
    delete scene;
    return rc;
}

/* function-level known-answer hooks (T0): the reference's own camera maths */
void ref_world_to_raster(unsigned boxMask, int resX, int resY, int n, const float *pts, float *out)
{
    Scene *scene = make_scene(boxMask, resX, resY);
    for (int i = 0; i < n; i++) {
        const Vec2f r = scene->mCamera.WorldToRaster(Vec3f(pts[3*i], pts[3*i+1], pts[3*i+2]));
        out[2*i] = r.x; out[2*i+1] = r.y;
    }
    delete scene;
}

int ref_kat(unsigned boxMask, int resX, int resY, int op, int n, const float *inAll, float *outAll)
{
    Scene *scene = make_scene(boxMask, resX, resY);
    for (int i = 0; i < n; i++) {
        const float *in = inAll + (size_t)i * VCM_KAT_FLOATS;
        float *out = outAll + (size_t)i * VCM_KAT_FLOATS;
        for (int k = 0; k < VCM_KAT_FLOATS; k++) out[k] = 0.f;
        switch (op) {
        case VCM_KAT_INTERSECT: {
            Ray ray(Vec3f(in[0], in[1], in[2]), Vec3f(in[3], in[4], in[5]), in[6]);
            Isect is(1e36f);
            if (scene->Intersect(ray, is)) {
                out[0] = 1.f; out[1] = is.dist; out[2] = (float)is.matID; out[3] = (float)is.lightID;
                out[4] = is.normal.x; out[5] = is.normal.y; out[6] = is.normal.z;
            }
        } break;
        case VCM_KAT_OCCLUDED:
            out[0] = scene->Occluded(Vec3f(in[0], in[1], in[2]), Vec3f(in[3], in[4], in[5]), in[6]) ? 1.f : 0.f;
            break;
        case VCM_KAT_BSDF_EVAL:
            kat_bsdf<false>(*scene, op, in, out);
            break;
        case VCM_KAT_BSDF_SAMPLE:
            if (in[10] != 0.f) kat_bsdf<true>(*scene, op, in, out); else kat_bsdf<false>(*scene, op, in, out);
            break;
        case VCM_KAT_LIGHT_EMIT: {
            const AbstractLight *l = scene->GetLightPtr((int)in[0]);
            Vec3f pos(0), dir(0);
            float emissionPdfW = 0.f, directPdfA = 0.f, cosLight = 0.f;
            const Vec3f e = l->Emit(scene->mSceneSphere, Vec2f(in[1], in[2]), Vec2f(in[3], in[4]), pos, dir, emissionPdfW,
                                    &directPdfA, &cosLight);
            out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = pos.x; out[4] = pos.y; out[5] = pos.z;
            out[6] = dir.x; out[7] = dir.y; out[8] = dir.z; out[9] = emissionPdfW; out[10] = directPdfA; out[11] = cosLight;
            out[12] = l->IsFinite() ? 1.f : 0.f; out[13] = l->IsDelta() ? 1.f : 0.f;
        } break;
        case VCM_KAT_LIGHT_ILLUMINATE: {
            const AbstractLight *l = scene->GetLightPtr((int)in[0]);
            Vec3f dir(0);
            float dist = 0.f, directPdfW = 0.f, emissionPdfW = 0.f, cosAtLight = 0.f;
            const Vec3f r = l->Illuminate(scene->mSceneSphere, Vec3f(in[1], in[2], in[3]), Vec2f(in[4], in[5]), dir, dist,
                                          directPdfW, &emissionPdfW, &cosAtLight);
            out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = dir.x; out[4] = dir.y; out[5] = dir.z; out[6] = dist;
            if (!r.IsZero()) { out[7] = directPdfW; out[8] = emissionPdfW; out[9] = cosAtLight; }
        } break;
        case VCM_KAT_LIGHT_RADIANCE: {
            const AbstractLight *l = scene->GetLightPtr((int)in[0]);
            float directPdfA = 0.f, emissionPdfW = 0.f;
            const Vec3f r = l->GetRadiance(scene->mSceneSphere, Vec3f(in[1], in[2], in[3]), Vec3f(in[4], in[5], in[6]),
                                           &directPdfA, &emissionPdfW);
            out[0] = r.x; out[1] = r.y; out[2] = r.z;
            if (!r.IsZero()) { out[3] = directPdfA; out[4] = emissionPdfW; }
        } break;
        case VCM_KAT_CAMERA: {
            const Ray ray = scene->mCamera.GenerateRay(Vec2f(in[0], in[1]));
            out[0] = ray.dir.x; out[1] = ray.dir.y; out[2] = ray.dir.z;
            const Vec2f ip = scene->mCamera.WorldToRaster(Vec3f(in[2], in[3], in[4]));
            out[3] = ip.x; out[4] = ip.y;
            out[5] = scene->mCamera.CheckRaster(ip) ? 1.f : 0.f;
        } break;
        default: delete scene; return -1;
        }
    }
    delete scene;
    return 0;
}

/* ---- version-2 scenes (include/smallvcm_amd.h vcm_scene_desc2): the reference has no loader for anything but its
 *      Cornell boxes, so the harness builds a reference `Scene` from the arrays WITH THE REFERENCE'S OWN CONSTRUCTORS
 *      (Triangle, Sphere, AreaLight, DirectionalLight, PointLight, BackgroundLight, Camera::Setup is not re-run: the
 *      camera members are public) and checks that every derived member the description carries -- triangle normals,
 *      light frames, 1 / area -- is what the constructor produced.  Returns NULL (and sets *why) otherwise. */
static bool same3(const Vec3f &a, const float *b) { return a.x == b[0] && a.y == b[1] && a.z == b[2]; }
static Scene *make_scene2(const vcm_scene_desc2 *d, int *why)
{
    *why = 0;
    Scene *scene = new Scene;
    GeometryList *list = new GeometryList;
    scene->mGeometry = list;
    for (int i = 0; i < d->nMaterials; i++) {
        const vcm_material &m = d->materials[i];
        Material mat;
        mat.Reset();
        mat.mDiffuseReflectance = Vec3f(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
        mat.mPhongReflectance = Vec3f(m.phong[0], m.phong[1], m.phong[2]);
        mat.mPhongExponent = m.phongExp;
        mat.mMirrorReflectance = Vec3f(m.mirror[0], m.mirror[1], m.mirror[2]);
        mat.mIOR = m.ior;
        scene->mMaterials.push_back(mat);
        if (d->mat2light[i] >= 0) scene->mMaterial2Light.insert(std::make_pair(i, d->mat2light[i]));
    }
    for (int i = 0; i < d->nPrims; i++) {
        const vcm_prim &p = d->prims[i];
        if (p.type == VCM_PRIM_TRIANGLE) {
            Triangle *t = new Triangle(Vec3f(p.p0[0], p.p0[1], p.p0[2]), Vec3f(p.p1[0], p.p1[1], p.p1[2]), Vec3f(p.p2[0], p.p2[1], p.p2[2]), p.matID);
            if (!same3(t->mNormal, p.n)) *why = 1;
            list->mGeometry.push_back(t);
        } else {
            list->mGeometry.push_back(new Sphere(Vec3f(p.p0[0], p.p0[1], p.p0[2]), p.p1[0], p.matID));
        }
    }
    for (int i = 0; i < d->nLights; i++) {
        const vcm_light &l = d->lights[i];
        const Vec3f inten(l.intensity[0], l.intensity[1], l.intensity[2]);
        if (l.type == VCM_LIGHT_AREA) {
            const Vec3f p0(l.p0[0], l.p0[1], l.p0[2]);
            AreaLight *a = new AreaLight(p0, p0 + Vec3f(l.e1[0], l.e1[1], l.e1[2]), p0 + Vec3f(l.e2[0], l.e2[1], l.e2[2]));
            /* p0 + e - p0 need not give e back bit for bit: the description's edges are the truth, the frame and the
               area must be what the constructor derives from THEM */
            a->e1 = Vec3f(l.e1[0], l.e1[1], l.e1[2]); a->e2 = Vec3f(l.e2[0], l.e2[1], l.e2[2]);
            const Vec3f normal = Cross(a->e1, a->e2);
            a->mInvArea = 2.f / normal.Length();
            a->mFrame.SetFromZ(normal);
            if (a->mInvArea != l.invArea || !same3(a->mFrame.mX, l.frameX) || !same3(a->mFrame.mY, l.frameY) || !same3(a->mFrame.mZ, l.frameZ)) *why = 2;
            a->mIntensity = inten;
            scene->mLights.push_back(a);
        } else if (l.type == VCM_LIGHT_DIRECTIONAL) {
            DirectionalLight *dl = new DirectionalLight(Vec3f(l.frameZ[0], l.frameZ[1], l.frameZ[2]));
            dl->mFrame.mX = Vec3f(l.frameX[0], l.frameX[1], l.frameX[2]); dl->mFrame.mY = Vec3f(l.frameY[0], l.frameY[1], l.frameY[2]);
            dl->mFrame.mZ = Vec3f(l.frameZ[0], l.frameZ[1], l.frameZ[2]);
            dl->mIntensity = inten;
            scene->mLights.push_back(dl);
        } else if (l.type == VCM_LIGHT_POINT) {
            PointLight *pl = new PointLight(Vec3f(l.p0[0], l.p0[1], l.p0[2]));
            pl->mIntensity = inten;
            scene->mLights.push_back(pl);
        } else {
            BackgroundLight *bl = new BackgroundLight;
            bl->mBackgroundColor = inten; bl->mScale = l.scale;
            scene->mLights.push_back(bl);
            if (i == d->backgroundLight) scene->mBackground = bl;
        }
    }
    scene->mSceneSphere.mSceneCenter = Vec3f(d->sceneCenter[0], d->sceneCenter[1], d->sceneCenter[2]);
    scene->mSceneSphere.mSceneRadius = d->sceneRadius;
    scene->mSceneSphere.mInvSceneRadiusSqr = d->invSceneRadiusSqr;
    {   /* the reference's own BuildSceneSphere must agree */
        Scene probe; probe.mGeometry = list;
        probe.BuildSceneSphere();
        if (!same3(probe.mSceneSphere.mSceneCenter, d->sceneCenter) || probe.mSceneSphere.mSceneRadius != d->sceneRadius ||
            probe.mSceneSphere.mInvSceneRadiusSqr != d->invSceneRadiusSqr) *why = 3;
        probe.mGeometry = NULL;
    }
    Camera &c = scene->mCamera;
    c.mPosition = Vec3f(d->camera.position[0], d->camera.position[1], d->camera.position[2]);
    c.mForward = Vec3f(d->camera.forward[0], d->camera.forward[1], d->camera.forward[2]);
    c.mResolution = Vec2f(d->camera.resolution[0], d->camera.resolution[1]);
    memcpy(&c.mRasterToWorld, d->camera.rasterToWorld, 16 * sizeof(float));
    memcpy(&c.mWorldToRaster, d->camera.worldToRaster, 16 * sizeof(float));
    c.mImagePlaneDist = d->camera.imagePlaneDist;
    if (*why) { delete scene; return NULL; }
    return scene;
}

/* the reference's constructors, one object each (tests compare the product's vcm_make_* with these bit for bit) */
void ref_make_triangle(const float *p0, const float *p1, const float *p2, int matID, vcm_prim *out)
{
    Triangle t(Vec3f(p0[0], p0[1], p0[2]), Vec3f(p1[0], p1[1], p1[2]), Vec3f(p2[0], p2[1], p2[2]), matID);
    memset(out, 0, sizeof(*out));
    out->type = VCM_PRIM_TRIANGLE; out->matID = t.matID;
    for (int k = 0; k < 3; k++) { out->p0[k] = t.p[0].Get(k); out->p1[k] = t.p[1].Get(k); out->p2[k] = t.p[2].Get(k); out->n[k] = t.mNormal.Get(k); }
}
void ref_make_area_light(const float *p0, const float *p1, const float *p2, const float *intensity, vcm_light *out)
{
    AreaLight a(Vec3f(p0[0], p0[1], p0[2]), Vec3f(p1[0], p1[1], p1[2]), Vec3f(p2[0], p2[1], p2[2]));
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_AREA;
    for (int k = 0; k < 3; k++) {
        out->p0[k] = a.p0.Get(k); out->e1[k] = a.e1.Get(k); out->e2[k] = a.e2.Get(k);
        out->frameX[k] = a.mFrame.mX.Get(k); out->frameY[k] = a.mFrame.mY.Get(k); out->frameZ[k] = a.mFrame.mZ.Get(k);
        out->intensity[k] = intensity[k];
    }
    out->invArea = a.mInvArea;
}
void ref_make_directional_light(const float *dir, const float *intensity, vcm_light *out)
{
    DirectionalLight d(Vec3f(dir[0], dir[1], dir[2]));
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_DIRECTIONAL;
    for (int k = 0; k < 3; k++) {
        out->frameX[k] = d.mFrame.mX.Get(k); out->frameY[k] = d.mFrame.mY.Get(k); out->frameZ[k] = d.mFrame.mZ.Get(k);
        out->intensity[k] = intensity[k];
    }
}
int ref_make_camera(const float *pos, const float *fwd, const float *up, float fov, int resX, int resY, vcm_camera *out)
{
    Camera c;
    c.Setup(Vec3f(pos[0], pos[1], pos[2]), Vec3f(fwd[0], fwd[1], fwd[2]), Vec3f(up[0], up[1], up[2]), Vec2f(float(resX), float(resY)), fov);
    memset(out, 0, sizeof(*out));
    for (int k = 0; k < 3; k++) { out->position[k] = c.mPosition.Get(k); out->forward[k] = c.mForward.Get(k); }
    out->resolution[0] = c.mResolution.x; out->resolution[1] = c.mResolution.y;
    memcpy(out->rasterToWorld, &c.mRasterToWorld, 16 * sizeof(float));
    memcpy(out->worldToRaster, &c.mWorldToRaster, 16 * sizeof(float));
    out->imagePlaneDist = c.mImagePlaneDist;
    return 0;
}
/* 0 = the description is what the reference's constructors build from the same input; 1 triangle normal, 2 area light,
   3 scene sphere differ */
int ref_check_scene2(const vcm_scene_desc2 *d)
{
    int why = 0;
    Scene *s = make_scene2(d, &why);
    delete s;
    return why;
}
/* function-level known answers over a version-2 scene (the same records as ref_kat; Scene::Intersect = the brute-force
   walk over every primitive) */
int ref_kat2(const vcm_scene_desc2 *d, int op, int n, const float *inAll, float *outAll)
{
    int why = 0;
    Scene *scene = make_scene2(d, &why);
    if (!scene) return -100 - why;
    int rc = 0;
    for (int i = 0; i < n && rc == 0; i++) {
        const float *in = inAll + (size_t)i * VCM_KAT_FLOATS;
        float *out = outAll + (size_t)i * VCM_KAT_FLOATS;
        for (int k = 0; k < VCM_KAT_FLOATS; k++) out[k] = 0.f;
        if (op == VCM_KAT_INTERSECT) {
            Ray ray(Vec3f(in[0], in[1], in[2]), Vec3f(in[3], in[4], in[5]), in[6]);
            Isect is(1e36f);
            if (scene->Intersect(ray, is)) {
                out[0] = 1.f; out[1] = is.dist; out[2] = (float)is.matID; out[3] = (float)is.lightID;
                out[4] = is.normal.x; out[5] = is.normal.y; out[6] = is.normal.z;
            }
        } else if (op == VCM_KAT_OCCLUDED) {
            out[0] = scene->Occluded(Vec3f(in[0], in[1], in[2]), Vec3f(in[3], in[4], in[5]), in[6]) ? 1.f : 0.f;
        } else rc = -1;
    }
    delete scene;
    return rc;
}

#ifdef REF_TAPE
long long ref_detmath_calls(void) { return g_detmath_calls; }

/* ref_run_tape for a version-2 scene */
int ref_run_tape2(const vcm_scene_desc2 *d, int vcmAlgo, float radiusFactor, float radiusAlpha, int seed,
                  int firstIteration, int nIter, unsigned minLen, unsigned maxLen,
                  const unsigned char *lightCounts, const unsigned char *camCounts, float *fbSumOut, long long *consumedOut)
{
    int why = 0;
    Scene *scene = make_scene2(d, &why);
    if (!scene) return -100 - why;
    const int N = int(d->camera.resolution[0]) * int(d->camera.resolution[1]);
    AbstractRenderer *r;
    bool lightTraceOnly = false;
    if (vcmAlgo == 5) r = new PathTracer(*scene, seed);
    else if (vcmAlgo == 6) r = new EyeLight(*scene, seed);
    else {
        VertexCM *v = new VertexCM(*scene, (VertexCM::AlgorithmType)vcmAlgo, radiusFactor, radiusAlpha, seed);
        lightTraceOnly = v->mLightTraceOnly;
        r = v;
    }
    r->mMaxPathLength = maxLen;
    r->mMinPathLength = minLen;
    int bad = 0;
    long long total = 0;
    for (int i = 0; i < nIter; i++) {
        memset(&g_tape, 0, sizeof(g_tape));
        g_tape.counts[0] = lightCounts + (size_t)i * N;
        g_tape.counts[1] = camCounts + (size_t)i * N;
        g_tape.N = N;
        g_tape.key[0] = (uint32_t)seed;
        g_tape.key[1] = (uint32_t)i;
        long long expect = 0;
        for (int p = 0; p < N; p++) expect += g_tape.counts[0][p];
        if (!lightTraceOnly) for (int p = 0; p < N; p++) expect += g_tape.counts[1][p];
        r->RunIteration(firstIteration + i);
        if (g_tape.overrun || g_tape.consumed != expect) bad = 1;
        total += g_tape.consumed;
    }
    memcpy(fbSumOut, &r->mFramebuffer.mColor[0], (size_t)N * 3 * sizeof(float));
    if (consumedOut) *consumedOut = total;
    delete r;
    delete scene;
    return bad;
}

/* Runs nIter iterations (global iteration index = firstIteration + i, RNG
 * local iteration = i) of the reference's VertexCM on one renderer.
 * lightCounts / camCounts: nIter*N bytes each.  Returns 0 if the reference
 * consumed exactly the taped number of floats, 1 otherwise. */
int ref_run_tape(unsigned boxMask, int resX, int resY, int vcmAlgo,
                 float radiusFactor, float radiusAlpha, int seed,
                 int firstIteration, int nIter, unsigned minLen, unsigned maxLen,
                 const unsigned char *lightCounts, const unsigned char *camCounts,
                 float *fbSumOut, long long *consumedOut)
{
    Scene *scene = make_scene(boxMask, resX, resY);
    const int N = resX * resY;
    /* vcmAlgo 0..4 = VertexCM::AlgorithmType; 5 = PathTracer, 6 = EyeLight (created as config.hxx:118-121 does) */
    AbstractRenderer *r;
    bool lightTraceOnly = false;
    if (vcmAlgo == 5) r = new PathTracer(*scene, seed);
    else if (vcmAlgo == 6) r = new EyeLight(*scene, seed);
    else {
        VertexCM *v = new VertexCM(*scene, (VertexCM::AlgorithmType)vcmAlgo, radiusFactor, radiusAlpha, seed);
        lightTraceOnly = v->mLightTraceOnly;
        r = v;
    }
    r->mMaxPathLength = maxLen;   /* src/smallvcm.cxx:70-71 */
    r->mMinPathLength = minLen;
    int bad = 0;
    long long total = 0;
    for (int i = 0; i < nIter; i++) {
        memset(&g_tape, 0, sizeof(g_tape));
        g_tape.counts[0] = lightCounts + (size_t)i * N;
        g_tape.counts[1] = camCounts + (size_t)i * N;
        g_tape.N = N;
        g_tape.key[0] = (uint32_t)seed;
        g_tape.key[1] = (uint32_t)i;
        long long expect = 0;
        for (int p = 0; p < N; p++) expect += g_tape.counts[0][p];
        if (!lightTraceOnly) for (int p = 0; p < N; p++) expect += g_tape.counts[1][p];
        r->RunIteration(firstIteration + i);
        if (g_tape.overrun || g_tape.consumed != expect) bad = 1;
        total += g_tape.consumed;
    }
    memcpy(fbSumOut, &r->mFramebuffer.mColor[0], (size_t)N * 3 * sizeof(float));
    if (consumedOut) *consumedOut = total;
    delete r;
    delete scene;
    return bad;
}
#endif

#ifdef REF_STOCK
/* render() of src/smallvcm.cxx:52-151, iteration-based branch (:96-109),
 * with resolution / threads / seed as parameters and wall-clock timing.
 * fbOut = averaged framebuffer exactly as render() leaves it (:116-142). */
/* iterIndex (optional): the value handed to RunIteration for loop index i -- it only sets the merge radius
 * (vertexcm.hxx:295-296), so a benchmark can make the reference render the SAME radius window the GPU is timed on
 * while keeping render()'s one-renderer-per-thread loop; NULL = the loop index itself, as in render(). */
int ref_render_stock_iters(unsigned boxMask, int resX, int resY, int configAlgo /* Config::Algorithm or -1 */,
                           int vcmAlgo, int iterations, const int *iterIndex, int numThreads, int baseSeed,
                           unsigned minLen, unsigned maxLen, float radiusFactor, float radiusAlpha,
                           float *fbOut, double *wallSeconds)
{
    Scene *scene = make_scene(boxMask, resX, resY);
    Config config;
    config.mScene = scene;
    config.mAlgorithm = (configAlgo >= 0) ? (Config::Algorithm)configAlgo : algo_from_vcm(vcmAlgo);
    config.mIterations = iterations;
    config.mMaxTime = -1.f;
    config.mRadiusFactor = radiusFactor;
    config.mRadiusAlpha = radiusAlpha;
    config.mNumThreads = numThreads > 0 ? numThreads : omp_get_num_procs();
    config.mBaseSeed = baseSeed;
    config.mMaxPathLength = maxLen;
    config.mMinPathLength = minLen;
    config.mResolution = Vec2i(resX, resY);
    config.mFullReport = false;
    Framebuffer fb;
    config.mFramebuffer = &fb;

    omp_set_num_threads(config.mNumThreads);
    std::vector<AbstractRenderer*> renderers((size_t)config.mNumThreads);
    for (int i = 0; i < config.mNumThreads; i++) {
        renderers[i] = CreateRenderer(config, config.mBaseSeed + i);
        renderers[i]->mMaxPathLength = config.mMaxPathLength;
        renderers[i]->mMinPathLength = config.mMinPathLength;
    }
    const auto t0 = std::chrono::steady_clock::now();
    int iter;
#pragma omp parallel for
    for (iter = 0; iter < config.mIterations; iter++) {
        const int threadId = omp_get_thread_num();
        renderers[threadId]->RunIteration(iterIndex ? iterIndex[iter] : iter);
    }
    const auto t1 = std::chrono::steady_clock::now();
    int usedRenderers = 0;
    for (int i = 0; i < config.mNumThreads; i++) {
        if (!renderers[i]->WasUsed()) continue;
        if (usedRenderers == 0) renderers[i]->GetFramebuffer(fb);
        else { Framebuffer tmp; renderers[i]->GetFramebuffer(tmp); fb.Add(tmp); }
        usedRenderers++;
    }
    fb.Scale(1.f / usedRenderers);
    for (int i = 0; i < config.mNumThreads; i++) delete renderers[i];
    if (fbOut) memcpy(fbOut, &fb.mColor[0], (size_t)resX * resY * 3 * sizeof(float));
    if (wallSeconds) *wallSeconds = std::chrono::duration<double>(t1 - t0).count();
    delete scene;
    return usedRenderers;
}

int ref_render_stock(unsigned boxMask, int resX, int resY, int configAlgo, int vcmAlgo, int iterations, int numThreads,
                     int baseSeed, unsigned minLen, unsigned maxLen, float radiusFactor, float radiusAlpha,
                     float *fbOut, double *wallSeconds)
{
    return ref_render_stock_iters(boxMask, resX, resY, configAlgo, vcmAlgo, iterations, NULL, numThreads, baseSeed, minLen,
                                  maxLen, radiusFactor, radiusAlpha, fbOut, wallSeconds);
}
#endif

} // extern "C"
