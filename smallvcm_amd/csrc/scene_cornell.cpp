// scene_cornell.cpp -- host-side builder of the reference's built-in Cornell
// box scenes as a `vcm_scene_desc`, for hosts that do not link the
// reference's scene code (the Python host, bench.py).  The C++ drop-in shim
// does NOT use this: it flattens the reference's own Scene object
// (smallvcm_amd/dropin/flatten_scene.hxx).
//
// Restates, with the reference's operand order (results are compared
// bit-for-bit against descs flattened from the reference, tests/golden/scene_*):
//   Scene::LoadCornellBox      src/scene.hxx:132-385
//   Scene::BuildSceneSphere    src/scene.hxx:387-398
//   Camera::Setup              src/camera.hxx:37-76
//   Mat4f product/Perspective/Invert  src/math.hxx:225-421
//   AreaLight / DirectionalLight ctors  src/lights.hxx:116-127, :239-243
//   Triangle ctor / GrowBBox   src/geometry.hxx:111-123, :158-170, :239-259
#include <string.h>
#include <math.h>
#include "../../include/smallvcm_amd.h"
#include "vcm_math.h"

using namespace vcm;

namespace {

struct M4 { float m[16]; };   /* column-major: Get(r,c) = m[r + 4c] (math.hxx:173) */
inline float &at(M4 &a, int r, int c) { return a.m[r + c * 4]; }
inline float at(const M4 &a, int r, int c) { return a.m[r + c * 4]; }
M4 m4_fill(float v) { M4 r; for (int i = 0; i < 16; i++) r.m[i] = v; return r; }
M4 m4_identity() { M4 r = m4_fill(0.f); for (int i = 0; i < 4; i++) at(r, i, i) = 1.f; return r; }
M4 m4_scale(V3 s)
{   /* math.hxx:232-238 */
    M4 r = m4_identity();
    at(r, 0, 0) = s.x; at(r, 1, 1) = s.y; at(r, 2, 2) = s.z; at(r, 3, 3) = 1;
    return r;
}
M4 m4_translate(V3 t)
{   /* math.hxx:240-246 */
    M4 r = m4_identity();
    at(r, 0, 3) = t.x; at(r, 1, 3) = t.y; at(r, 2, 3) = t.z; at(r, 3, 3) = 1;
    return r;
}
M4 m4_perspective(float fov, float nearP, float farP)
{   /* math.hxx:248-265 */
    const float f = 1.f / (tanf(fov * VCM_PI_F / 360.0f));
    const float d = 1.f / (nearP - farP);
    M4 r = m4_fill(0.f);
    at(r, 0, 0) = f;
    at(r, 1, 1) = -f;
    at(r, 2, 2) = (nearP + farP) * d; at(r, 2, 3) = 2.0f * nearP * farP * d;
    at(r, 3, 2) = -1.0f;
    return r;
}
M4 m4_mul(const M4 &l, const M4 &rr)
{   /* math.hxx:267-276 */
    M4 res = m4_fill(0.f);
    for (int row = 0; row < 4; row++)
        for (int col = 0; col < 4; col++)
            for (int i = 0; i < 4; i++)
                at(res, row, col) += at(l, row, i) * at(rr, i, col);
    return res;
}
/* Invert (math.hxx:280-419): adjugate / determinant.  Each adjugate entry is
 * s*(t0 - t1 - t2 + t3 + t4 - t5) with t = product of three entries,
 * accumulated left to right. */
struct AdjRow { int out; int sign; unsigned char t[6][3]; };
const AdjRow kAdj[16] = {
    { 0, +1, {{5,10,15},{5,11,14},{9,6,15},{9,7,14},{13,6,11},{13,7,10}}},
    { 4, -1, {{4,10,15},{4,11,14},{8,6,15},{8,7,14},{12,6,11},{12,7,10}}},
    { 8, +1, {{4,9,15},{4,11,13},{8,5,15},{8,7,13},{12,5,11},{12,7,9}}},
    {12, -1, {{4,9,14},{4,10,13},{8,5,14},{8,6,13},{12,5,10},{12,6,9}}},
    { 1, -1, {{1,10,15},{1,11,14},{9,2,15},{9,3,14},{13,2,11},{13,3,10}}},
    { 5, +1, {{0,10,15},{0,11,14},{8,2,15},{8,3,14},{12,2,11},{12,3,10}}},
    { 9, -1, {{0,9,15},{0,11,13},{8,1,15},{8,3,13},{12,1,11},{12,3,9}}},
    {13, +1, {{0,9,14},{0,10,13},{8,1,14},{8,2,13},{12,1,10},{12,2,9}}},
    { 2, +1, {{1,6,15},{1,7,14},{5,2,15},{5,3,14},{13,2,7},{13,3,6}}},
    { 6, -1, {{0,6,15},{0,7,14},{4,2,15},{4,3,14},{12,2,7},{12,3,6}}},
    {10, +1, {{0,5,15},{0,7,13},{4,1,15},{4,3,13},{12,1,7},{12,3,5}}},
    {14, -1, {{0,5,14},{0,6,13},{4,1,14},{4,2,13},{12,1,6},{12,2,5}}},
    { 3, -1, {{1,6,11},{1,7,10},{5,2,11},{5,3,10},{9,2,7},{9,3,6}}},
    { 7, +1, {{0,6,11},{0,7,10},{4,2,11},{4,3,10},{8,2,7},{8,3,6}}},
    {11, -1, {{0,5,11},{0,7,9},{4,1,11},{4,3,9},{8,1,7},{8,3,5}}},
    {15, +1, {{0,5,10},{0,6,9},{4,1,10},{4,2,9},{8,1,6},{8,2,5}}},
};
M4 m4_invert(const M4 &a)
{
    const float *m = a.m;
    float inv[16];
    static const int pat[6] = { +1, -1, -1, +1, +1, -1 };
    for (int k = 0; k < 16; k++) {
        const AdjRow &row = kAdj[k];
        float acc = 0.f;
        for (int j = 0; j < 6; j++) {
            const float t = m[row.t[j][0]] * m[row.t[j][1]] * m[row.t[j][2]];
            const int sg = row.sign * pat[j];
            if (j == 0) acc = (sg > 0) ? t : -t;
            else        acc = (sg > 0) ? acc + t : acc - t;
        }
        inv[row.out] = acc;
    }
    float det = m[0] * inv[0] + m[1] * inv[4] + m[2] * inv[8] + m[3] * inv[12];
    if (det == 0) return m4_identity();
    det = 1.f / det;
    M4 res;
    for (int i = 0; i < 16; i++) res.m[i] = inv[i] * det;
    return res;
}

void put3(float *d, V3 v) { d[0] = v.x; d[1] = v.y; d[2] = v.z; }

struct BBox { V3 mn, mx; };
void grow(BBox &b, V3 p)
{
    b.mn.x = smin(b.mn.x, p.x); b.mx.x = smax(b.mx.x, p.x);
    b.mn.y = smin(b.mn.y, p.y); b.mx.y = smax(b.mx.y, p.y);
    b.mn.z = smin(b.mn.z, p.z); b.mx.z = smax(b.mx.z, p.z);
}

void add_triangle(vcm_scene_desc &d, V3 p0, V3 p1, V3 p2, int matID)
{   /* Triangle ctor geometry.hxx:111-123 */
    vcm_prim &p = d.prims[d.nPrims++];
    p.type = VCM_PRIM_TRIANGLE; p.matID = matID;
    put3(p.p0, p0); put3(p.p1, p1); put3(p.p2, p2);
    put3(p.n, normalize(cross(p1 - p0, p2 - p0)));
}
void add_sphere(vcm_scene_desc &d, V3 center, float radius, int matID)
{
    vcm_prim &p = d.prims[d.nPrims++];
    p.type = VCM_PRIM_SPHERE; p.matID = matID;
    put3(p.p0, center); p.p1[0] = radius;
}
void add_area_light(vcm_scene_desc &d, V3 a0, V3 a1, V3 a2, float intensity, int matID)
{   /* AreaLight ctor lights.hxx:116-127 */
    const int idx = d.nLights++;
    vcm_light &l = d.lights[idx];
    l.type = VCM_LIGHT_AREA;
    const V3 e1 = a1 - a0, e2 = a2 - a0;
    put3(l.p0, a0); put3(l.e1, e1); put3(l.e2, e2);
    const V3 normal = cross(e1, e2);
    const float len = sqrtf(lensqr(normal));
    l.invArea = 2.f / len;
    Frame f; frame_from_z(f, normal);
    put3(l.frameX, f.mX); put3(l.frameY, f.mY); put3(l.frameZ, f.mZ);
    put3(l.intensity, sp3(intensity));
    d.mat2light[matID] = idx;
}
void camera_setup(V3 aPosition, V3 aForward, V3 aUp, float rx, float ry, float fov, vcm_camera &cam)
{   /* Camera::Setup camera.hxx:37-76 */
    const V3 forward = normalize(aForward);
    const V3 up = normalize(cross(aUp, -forward));
    const V3 left = cross(-forward, up);
    const V3 pos = mk3(dot(up, aPosition), dot(left, aPosition), dot(-forward, aPosition));
    M4 worldToCamera = m4_identity();
    const V3 nf = -forward;
    at(worldToCamera, 0, 0) = up.x;   at(worldToCamera, 0, 1) = up.y;   at(worldToCamera, 0, 2) = up.z;   at(worldToCamera, 0, 3) = -pos.x;
    at(worldToCamera, 1, 0) = left.x; at(worldToCamera, 1, 1) = left.y; at(worldToCamera, 1, 2) = left.z; at(worldToCamera, 1, 3) = -pos.y;
    at(worldToCamera, 2, 0) = nf.x;   at(worldToCamera, 2, 1) = nf.y;   at(worldToCamera, 2, 2) = nf.z;   at(worldToCamera, 2, 3) = -pos.z;
    const M4 perspective = m4_perspective(fov, 0.1f, 10000.f);
    const M4 worldToNScreen = m4_mul(perspective, worldToCamera);
    const M4 nscreenToWorld = m4_invert(worldToNScreen);
    const M4 worldToRaster = m4_mul(m4_mul(m4_scale(mk3(rx * 0.5f, ry * 0.5f, 0)), m4_translate(mk3(1.f, 1.f, 0))),
                                    worldToNScreen);
    const M4 rasterToWorld = m4_mul(m4_mul(nscreenToWorld, m4_translate(mk3(-1.f, -1.f, 0))),
                                    m4_scale(mk3(2.f / rx, 2.f / ry, 0)));
    const float tanHalfAngle = tanf(fov * VCM_PI_F / 360.f);
    put3(cam.position, aPosition);
    put3(cam.forward, forward);
    cam.resolution[0] = rx; cam.resolution[1] = ry;
    memcpy(cam.rasterToWorld, rasterToWorld.m, sizeof(float) * 16);
    memcpy(cam.worldToRaster, worldToRaster.m, sizeof(float) * 16);
    cam.imagePlaneDist = rx / (2.f * tanHalfAngle);
}
void scene_sphere(const vcm_prim *prims, int nPrims, float *center3, float *radius, float *invRadiusSqr)
{   /* BuildSceneSphere scene.hxx:387-398 (+ GrowBBox geometry.hxx:158-170, :239-259) */
    BBox bb; bb.mn = sp3(1e36f); bb.mx = sp3(-1e36f);
    for (int i = 0; i < nPrims; i++) {
        const vcm_prim &p = prims[i];
        if (p.type == VCM_PRIM_TRIANGLE) { grow(bb, ld3(p.p0)); grow(bb, ld3(p.p1)); grow(bb, ld3(p.p2)); }
        else {
            for (int k = 0; k < 8; k++) {
                V3 h = sp3(p.p1[0]);
                if (k & 1) h.x = -h.x;
                if (k & 2) h.y = -h.y;
                if (k & 4) h.z = -h.z;
                grow(bb, ld3(p.p0) + h);
            }
        }
    }
    const float radius2 = lensqr(bb.mx - bb.mn);
    put3(center3, (bb.mx + bb.mn) * 0.5f);
    *radius = sqrtf(radius2) * 0.5f;
    *invRadiusSqr = 1.f / sqr(*radius);
}
void reset_material(vcm_material &m)
{   /* Material::Reset materials.hxx:44-51 */
    memset(&m, 0, sizeof(m));
    m.phongExp = 1.f;
    m.ior = -1.f;
}

} // namespace

extern "C" unsigned vcm_scene_config_mask(int sceneID)
{   /* g_SceneConfigs src/config.hxx:146-151 */
    enum { kLightCeiling = 1, kLightSun = 2, kLightPoint = 4, kLightBackground = 8, kLargeMirrorSphere = 16,
           kSmallMirrorSphere = 64, kSmallGlassSphere = 128, kGlossyFloor = 256 };
    const unsigned both = kSmallMirrorSphere | kSmallGlassSphere;
    switch (sceneID) {
    case 0: return kGlossyFloor | both | kLightSun;
    case 1: return kGlossyFloor | kLargeMirrorSphere | kLightCeiling;
    case 2: return kGlossyFloor | both | kLightPoint;
    case 3: return kGlossyFloor | both | kLightBackground;
    default: return 0;
    }
}

extern "C" int vcm_scene_cornell(int resX, int resY, unsigned aBoxMask, vcm_scene_desc *out)
{
    if (!out || resX <= 0 || resY <= 0) return -1;
    vcm_scene_desc &d = *out;
    memset(&d, 0, sizeof(d));
    for (int i = 0; i < VCM_MAX_MATERIALS; i++) d.mat2light[i] = -1;
    d.backgroundLight = -1;

    enum { kLightCeiling = 1, kLightSun = 2, kLightPoint = 4, kLightBackground = 8, kLargeMirrorSphere = 16,
           kLargeGlassSphere = 32, kSmallMirrorSphere = 64, kSmallGlassSphere = 128, kGlossyFloor = 256 };
    const unsigned kBothLargeSpheres = kLargeMirrorSphere | kLargeGlassSphere;
    if ((aBoxMask & kBothLargeSpheres) == kBothLargeSpheres) aBoxMask &= ~(unsigned)kLargeGlassSphere;   /* :138-142 */

    const bool light_ceiling    = (aBoxMask & kLightCeiling) != 0;
    const bool light_sun        = (aBoxMask & kLightSun) != 0;
    const bool light_point      = (aBoxMask & kLightPoint) != 0;
    const bool light_background = (aBoxMask & kLightBackground) != 0;
    bool light_box = true;
    if (light_point) light_box = false;   /* :152-153 */

    /* ---- Camera::Setup camera.hxx:37-76 with the arguments of scene.hxx:156-160 */
    camera_setup(mk3(-0.0439815f, -4.12529f, 0.222539f), mk3(0.00688625f, 0.998505f, -0.0542161f),
                 mk3(3.73896e-4f, 0.0542148f, 0.998529f), float(resX), float(resY), 45, d.camera);

    /* ---- materials scene.hxx:162-209 */
    {
        vcm_material mat;
        reset_material(mat);
        d.materials[0] = mat;   /* 0) light1 */
        d.materials[1] = mat;   /* 1) light2 */
        reset_material(mat);    /* 2) glossy white floor */
        put3(mat.diffuse, sp3(0.1f)); put3(mat.phong, sp3(0.7f)); mat.phongExp = 90.f;
        d.materials[2] = mat;
        reset_material(mat);    /* 3) diffuse green left wall */
        put3(mat.diffuse, mk3(0.156863f, 0.803922f, 0.172549f));
        d.materials[3] = mat;
        reset_material(mat);    /* 4) diffuse red right wall */
        put3(mat.diffuse, mk3(0.803922f, 0.152941f, 0.152941f));
        d.materials[4] = mat;
        reset_material(mat);    /* 5) diffuse white back wall */
        put3(mat.diffuse, mk3(0.803922f, 0.803922f, 0.803922f));
        d.materials[5] = mat;
        reset_material(mat);    /* 6) mirror ball */
        put3(mat.mirror, sp3(1.f));
        d.materials[6] = mat;
        reset_material(mat);    /* 7) glass ball */
        put3(mat.mirror, sp3(1.f)); mat.ior = 1.6f;
        d.materials[7] = mat;
        reset_material(mat);    /* 8) diffuse blue wall */
        put3(mat.diffuse, mk3(0.156863f, 0.172549f, 0.803922f));
        d.materials[8] = mat;
        d.nMaterials = 9;
    }

    /* ---- geometry scene.hxx:213-329 */
    const V3 cb[8] = {
        mk3(-1.27029f,  1.30455f, -1.28002f), mk3( 1.28975f,  1.30455f, -1.28002f),
        mk3( 1.28975f,  1.30455f,  1.28002f), mk3(-1.27029f,  1.30455f,  1.28002f),
        mk3(-1.27029f, -1.25549f, -1.28002f), mk3( 1.28975f, -1.25549f, -1.28002f),
        mk3( 1.28975f, -1.25549f,  1.28002f), mk3(-1.27029f, -1.25549f,  1.28002f) };
    if ((aBoxMask & kGlossyFloor) != 0) {
        add_triangle(d, cb[0], cb[4], cb[5], 2); add_triangle(d, cb[5], cb[1], cb[0], 2);
        add_triangle(d, cb[0], cb[1], cb[2], 8); add_triangle(d, cb[2], cb[3], cb[0], 8);
    } else {
        add_triangle(d, cb[0], cb[4], cb[5], 5); add_triangle(d, cb[5], cb[1], cb[0], 5);
        add_triangle(d, cb[0], cb[1], cb[2], 5); add_triangle(d, cb[2], cb[3], cb[0], 5);
    }
    if (light_ceiling && !light_box) {
        add_triangle(d, cb[2], cb[6], cb[7], 0); add_triangle(d, cb[7], cb[3], cb[2], 1);
    } else {
        add_triangle(d, cb[2], cb[6], cb[7], 5); add_triangle(d, cb[7], cb[3], cb[2], 5);
    }
    add_triangle(d, cb[3], cb[7], cb[4], 3); add_triangle(d, cb[4], cb[0], cb[3], 3);
    add_triangle(d, cb[1], cb[5], cb[6], 4); add_triangle(d, cb[6], cb[2], cb[1], 4);

    const float largeRadius = 0.8f;
    const V3 center = (cb[0] + cb[1] + cb[4] + cb[5]) * (1.f / 4.f) + mk3(0, 0, largeRadius);
    if ((aBoxMask & kLargeMirrorSphere) != 0) add_sphere(d, center, largeRadius, 6);
    if ((aBoxMask & kLargeGlassSphere) != 0) add_sphere(d, center, largeRadius, 7);
    const float smallRadius = 0.5f;
    const V3 leftWallCenter  = (cb[0] + cb[4]) * (1.f / 2.f) + mk3(0, 0, smallRadius);
    const V3 rightWallCenter = (cb[1] + cb[5]) * (1.f / 2.f) + mk3(0, 0, smallRadius);
    const float xlen = rightWallCenter.x - leftWallCenter.x;
    const V3 leftBallCenter  = leftWallCenter  + mk3(2.f * xlen / 7.f, 0, 0);
    const V3 rightBallCenter = rightWallCenter - mk3(2.f * xlen / 7.f, 0, 0);
    if ((aBoxMask & kSmallMirrorSphere) != 0) add_sphere(d, leftBallCenter, smallRadius, 6);
    if ((aBoxMask & kSmallGlassSphere) != 0) add_sphere(d, rightBallCenter, smallRadius, 7);

    const V3 lb[8] = {
        mk3(-0.25f,  0.25f, 1.26002f), mk3( 0.25f,  0.25f, 1.26002f),
        mk3( 0.25f,  0.25f, 1.28002f), mk3(-0.25f,  0.25f, 1.28002f),
        mk3(-0.25f, -0.25f, 1.26002f), mk3( 0.25f, -0.25f, 1.26002f),
        mk3( 0.25f, -0.25f, 1.28002f), mk3(-0.25f, -0.25f, 1.28002f) };
    if (light_box) {
        add_triangle(d, lb[0], lb[2], lb[1], 5); add_triangle(d, lb[2], lb[0], lb[3], 5);
        add_triangle(d, lb[3], lb[4], lb[7], 5); add_triangle(d, lb[4], lb[3], lb[0], 5);
        add_triangle(d, lb[1], lb[6], lb[5], 5); add_triangle(d, lb[6], lb[1], lb[2], 5);
        add_triangle(d, lb[4], lb[5], lb[6], 5); add_triangle(d, lb[6], lb[7], lb[4], 5);
        if (light_ceiling) { add_triangle(d, lb[0], lb[5], lb[4], 0); add_triangle(d, lb[5], lb[0], lb[1], 1); }
        else               { add_triangle(d, lb[0], lb[5], lb[4], 5); add_triangle(d, lb[5], lb[0], lb[1], 5); }
    }

    /* ---- lights scene.hxx:333-384 */
    if (light_ceiling && !light_box) {
        add_area_light(d, cb[2], cb[6], cb[7], 0.95492965f, 0);
        add_area_light(d, cb[7], cb[3], cb[2], 0.95492965f, 1);
    } else if (light_ceiling && light_box) {
        add_area_light(d, lb[0], lb[5], lb[4], 25.03329895614464f, 0);
        add_area_light(d, lb[5], lb[0], lb[1], 25.03329895614464f, 1);
    }
    if (light_sun) {
        vcm_light &l = d.lights[d.nLights++];
        l.type = VCM_LIGHT_DIRECTIONAL;
        Frame f; frame_from_z(f, mk3(-1.f, 1.5f, -1.f));
        put3(l.frameX, f.mX); put3(l.frameY, f.mY); put3(l.frameZ, f.mZ);
        put3(l.intensity, mk3(0.5f, 0.2f, 0.f) * 20.f);
    }
    if (light_point) {
        vcm_light &l = d.lights[d.nLights++];
        l.type = VCM_LIGHT_POINT;
        put3(l.p0, mk3(0.0f, -0.5f, 1.0f));
        put3(l.intensity, sp3(70.f * (VCM_INV_PI_F * 0.25f)));
    }
    if (light_background) {
        d.backgroundLight = d.nLights;
        vcm_light &l = d.lights[d.nLights++];
        l.type = VCM_LIGHT_BACKGROUND;
        put3(l.intensity, mk3(135, 206, 250) / sp3(255.f));   /* lights.hxx:406 */
        l.scale = 1.f;
    }

    /* ---- BuildSceneSphere scene.hxx:387-398 */
    scene_sphere(d.prims, d.nPrims, d.sceneCenter, &d.sceneRadius, &d.invSceneRadiusSqr);
    return 0;
}

/* ---- constructors for version-2 scenes: one call per object, each doing what the reference's constructor does (the
 *      back end of a scene loader; include/smallvcm_amd.h).  Verified bit for bit against the reference's own classes
 *      (tests/test_scene2.py through oracle/ref_driver.cpp). */
extern "C" void vcm_make_triangle(const float *p0, const float *p1, const float *p2, int matID, vcm_prim *out)
{   /* Triangle::Triangle geometry.hxx:111-123 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_PRIM_TRIANGLE; out->matID = matID;
    put3(out->p0, ld3(p0)); put3(out->p1, ld3(p1)); put3(out->p2, ld3(p2));
    put3(out->n, normalize(cross(ld3(p1) - ld3(p0), ld3(p2) - ld3(p0))));
}
extern "C" void vcm_make_sphere(const float *center, float radius, int matID, vcm_prim *out)
{   /* Sphere::Sphere geometry.hxx:184-192 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_PRIM_SPHERE; out->matID = matID;
    put3(out->p0, ld3(center)); out->p1[0] = radius;
}
extern "C" void vcm_make_area_light(const float *p0, const float *p1, const float *p2, const float *intensity, vcm_light *out)
{   /* AreaLight::AreaLight lights.hxx:116-127 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_AREA;
    const V3 e1 = ld3(p1) - ld3(p0), e2 = ld3(p2) - ld3(p0);
    put3(out->p0, ld3(p0)); put3(out->e1, e1); put3(out->e2, e2);
    const V3 normal = cross(e1, e2);
    const float len = sqrtf(lensqr(normal));
    out->invArea = 2.f / len;
    Frame f; frame_from_z(f, normal);
    put3(out->frameX, f.mX); put3(out->frameY, f.mY); put3(out->frameZ, f.mZ);
    put3(out->intensity, ld3(intensity));
}
extern "C" void vcm_make_directional_light(const float *direction, const float *intensity, vcm_light *out)
{   /* DirectionalLight::DirectionalLight lights.hxx:239-243 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_DIRECTIONAL;
    Frame f; frame_from_z(f, ld3(direction));
    put3(out->frameX, f.mX); put3(out->frameY, f.mY); put3(out->frameZ, f.mZ);
    put3(out->intensity, ld3(intensity));
}
extern "C" void vcm_make_point_light(const float *position, const float *intensity, vcm_light *out)
{   /* PointLight::PointLight lights.hxx:324-328 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_POINT;
    put3(out->p0, ld3(position));
    put3(out->intensity, ld3(intensity));
}
extern "C" void vcm_make_background_light(float scale, vcm_light *out)
{   /* BackgroundLight::BackgroundLight lights.hxx:404-408 */
    memset(out, 0, sizeof(*out));
    out->type = VCM_LIGHT_BACKGROUND;
    put3(out->intensity, mk3(135, 206, 250) / sp3(255.f));
    out->scale = scale;
}
extern "C" void vcm_make_material(vcm_material *out)
{   /* Material::Reset materials.hxx:44-51 */
    reset_material(*out);
}
extern "C" int vcm_make_camera(const float *position, const float *forward, const float *up, float horizontalFovDeg, int resX,
                               int resY, vcm_camera *out)
{   /* Camera::Setup camera.hxx:37-76 */
    if (!out || resX <= 0 || resY <= 0) return -1;
    memset(out, 0, sizeof(*out));
    camera_setup(ld3(position), ld3(forward), ld3(up), float(resX), float(resY), horizontalFovDeg, *out);
    return 0;
}
extern "C" void vcm_make_scene_sphere(const vcm_prim *prims, int nPrims, float *center3, float *radius, float *invRadiusSqr)
{   /* Scene::BuildSceneSphere scene.hxx:387-398 */
    scene_sphere(prims, nPrims, center3, radius, invRadiusSqr);
}
