// flatten_scene.hxx -- turns the reference's `Scene` object into the POD
// `vcm_scene_desc` that crosses the C-ABI (include/smallvcm_amd.h).
//
// Compiled only inside a translation unit that has already included the
// reference's own headers (scene.hxx, geometry.hxx, lights.hxx, camera.hxx):
// the drop-in shim `vertexcm.hxx` next to this file, and the parity harness
// oracle/ref_driver.cpp.  Reads only public members:
//   Scene            src/scene.hxx:476-485
//   GeometryList     src/geometry.hxx:104
//   Triangle/Sphere  src/geometry.hxx:174-176, :263-265
//   Material         src/materials.hxx:54-65
//   lights           src/lights.hxx:229-232, :314-315, :395-396, :512-513
//   SceneSphere      src/lights.hxx:32-40
//   Camera           src/camera.hxx:121-126
#ifndef SMALLVCM_AMD_FLATTEN_SCENE_HXX
#define SMALLVCM_AMD_FLATTEN_SCENE_HXX

#include <string.h>
#include <vector>
#include "smallvcm_amd.h"

namespace smallvcm_amd {

inline void put3(float *dst, const Vec3f &v) { dst[0] = v.x; dst[1] = v.y; dst[2] = v.z; }

// returns 0 on success, otherwise a negative code (scene exceeds the fixed
// capacities of vcm_scene_desc or holds an unknown geometry/light class)
inline int FlattenScene(const Scene &aScene, vcm_scene_desc &oDesc)
{
    memset(&oDesc, 0, sizeof(oDesc));

    const GeometryList *list = dynamic_cast<const GeometryList*>(aScene.mGeometry);
    if(list == NULL) return -1;
    if(list->mGeometry.size() > VCM_MAX_PRIMS) return -2;
    oDesc.nPrims = int(list->mGeometry.size());
    for(int i = 0; i < oDesc.nPrims; i++)
    {
        vcm_prim &p = oDesc.prims[i];
        if(const Triangle *t = dynamic_cast<const Triangle*>(list->mGeometry[i]))
        {
            p.type  = VCM_PRIM_TRIANGLE;
            p.matID = t->matID;
            put3(p.p0, t->p[0]); put3(p.p1, t->p[1]); put3(p.p2, t->p[2]);
            put3(p.n, t->mNormal);
        }
        else if(const Sphere *s = dynamic_cast<const Sphere*>(list->mGeometry[i]))
        {
            p.type  = VCM_PRIM_SPHERE;
            p.matID = s->matID;
            put3(p.p0, s->center);
            p.p1[0] = s->radius;
        }
        else return -3;
    }

    if(aScene.mMaterials.size() > VCM_MAX_MATERIALS) return -4;
    oDesc.nMaterials = int(aScene.mMaterials.size());
    for(int i = 0; i < VCM_MAX_MATERIALS; i++) oDesc.mat2light[i] = -1;
    for(int i = 0; i < oDesc.nMaterials; i++)
    {
        const Material &m = aScene.mMaterials[i];
        vcm_material &d = oDesc.materials[i];
        put3(d.diffuse, m.mDiffuseReflectance);
        put3(d.phong, m.mPhongReflectance);
        d.phongExp = m.mPhongExponent;
        put3(d.mirror, m.mMirrorReflectance);
        d.ior = m.mIOR;
    }
    for(std::map<int, int>::const_iterator it = aScene.mMaterial2Light.begin();
        it != aScene.mMaterial2Light.end(); ++it)
    {
        if(it->first < 0 || it->first >= VCM_MAX_MATERIALS) return -5;
        oDesc.mat2light[it->first] = it->second;
    }

    if(aScene.mLights.size() > VCM_MAX_LIGHTS) return -6;
    oDesc.nLights = int(aScene.mLights.size());
    oDesc.backgroundLight = -1;
    for(int i = 0; i < oDesc.nLights; i++)
    {
        vcm_light &d = oDesc.lights[i];
        const AbstractLight *l = aScene.mLights[i];
        if(const AreaLight *a = dynamic_cast<const AreaLight*>(l))
        {
            d.type = VCM_LIGHT_AREA;
            put3(d.p0, a->p0); put3(d.e1, a->e1); put3(d.e2, a->e2);
            put3(d.frameX, a->mFrame.mX); put3(d.frameY, a->mFrame.mY); put3(d.frameZ, a->mFrame.mZ);
            put3(d.intensity, a->mIntensity);
            d.invArea = a->mInvArea;
        }
        else if(const DirectionalLight *dl = dynamic_cast<const DirectionalLight*>(l))
        {
            d.type = VCM_LIGHT_DIRECTIONAL;
            put3(d.frameX, dl->mFrame.mX); put3(d.frameY, dl->mFrame.mY); put3(d.frameZ, dl->mFrame.mZ);
            put3(d.intensity, dl->mIntensity);
        }
        else if(const PointLight *pl = dynamic_cast<const PointLight*>(l))
        {
            d.type = VCM_LIGHT_POINT;
            put3(d.p0, pl->mPosition);
            put3(d.intensity, pl->mIntensity);
        }
        else if(const BackgroundLight *bl = dynamic_cast<const BackgroundLight*>(l))
        {
            d.type = VCM_LIGHT_BACKGROUND;
            put3(d.intensity, bl->mBackgroundColor);
            d.scale = bl->mScale;
            if(bl == aScene.mBackground) oDesc.backgroundLight = i;
        }
        else return -7;
    }

    put3(oDesc.sceneCenter, aScene.mSceneSphere.mSceneCenter);
    oDesc.sceneRadius       = aScene.mSceneSphere.mSceneRadius;
    oDesc.invSceneRadiusSqr = aScene.mSceneSphere.mInvSceneRadiusSqr;

    const Camera &c = aScene.mCamera;
    put3(oDesc.camera.position, c.mPosition);
    put3(oDesc.camera.forward, c.mForward);
    oDesc.camera.resolution[0] = c.mResolution.x;
    oDesc.camera.resolution[1] = c.mResolution.y;
    memcpy(oDesc.camera.rasterToWorld, c.mRasterToWorld.GetPtr(), 16 * sizeof(float));
    memcpy(oDesc.camera.worldToRaster, c.mWorldToRaster.GetPtr(), 16 * sizeof(float));
    oDesc.camera.imagePlaneDist = c.mImagePlaneDist;
    return 0;
}

// The same into the version-2 description (pointer + count: any number of primitives, materials, lights), for a
// reference Scene that outgrows the fixed capacities -- e.g. one filled by a scene loader added to the reference.  The
// arrays live in oStorage, which must outlive the vcm_create2 call (the library copies them).
struct SceneArrays
{
    std::vector<vcm_prim>     prims;
    std::vector<vcm_material> materials;
    std::vector<int>          mat2light;
    std::vector<vcm_light>    lights;
};

inline int FlattenScene2(const Scene &aScene, SceneArrays &oStorage, vcm_scene_desc2 &oDesc)
{
    memset(&oDesc, 0, sizeof(oDesc));
    const GeometryList *list = dynamic_cast<const GeometryList*>(aScene.mGeometry);
    if(list == NULL) return -1;
    oStorage.prims.resize(list->mGeometry.size());
    for(size_t i = 0; i < list->mGeometry.size(); i++)
    {
        vcm_prim &p = oStorage.prims[i];
        memset(&p, 0, sizeof(p));
        if(const Triangle *t = dynamic_cast<const Triangle*>(list->mGeometry[i]))
        {
            p.type = VCM_PRIM_TRIANGLE; p.matID = t->matID;
            put3(p.p0, t->p[0]); put3(p.p1, t->p[1]); put3(p.p2, t->p[2]); put3(p.n, t->mNormal);
        }
        else if(const Sphere *s = dynamic_cast<const Sphere*>(list->mGeometry[i]))
        {
            p.type = VCM_PRIM_SPHERE; p.matID = s->matID;
            put3(p.p0, s->center); p.p1[0] = s->radius;
        }
        else return -3;
    }
    oStorage.materials.resize(aScene.mMaterials.size());
    oStorage.mat2light.assign(aScene.mMaterials.size(), -1);
    for(size_t i = 0; i < aScene.mMaterials.size(); i++)
    {
        const Material &m = aScene.mMaterials[i];
        vcm_material &d = oStorage.materials[i];
        put3(d.diffuse, m.mDiffuseReflectance); put3(d.phong, m.mPhongReflectance); d.phongExp = m.mPhongExponent;
        put3(d.mirror, m.mMirrorReflectance); d.ior = m.mIOR;
    }
    for(std::map<int, int>::const_iterator it = aScene.mMaterial2Light.begin(); it != aScene.mMaterial2Light.end(); ++it)
    {
        if(it->first < 0 || it->first >= int(oStorage.mat2light.size())) return -5;
        oStorage.mat2light[it->first] = it->second;
    }
    oStorage.lights.resize(aScene.mLights.size());
    oDesc.backgroundLight = -1;
    for(size_t i = 0; i < aScene.mLights.size(); i++)
    {
        vcm_light &d = oStorage.lights[i];
        memset(&d, 0, sizeof(d));
        const AbstractLight *l = aScene.mLights[i];
        if(const AreaLight *a = dynamic_cast<const AreaLight*>(l))
        {
            d.type = VCM_LIGHT_AREA;
            put3(d.p0, a->p0); put3(d.e1, a->e1); put3(d.e2, a->e2);
            put3(d.frameX, a->mFrame.mX); put3(d.frameY, a->mFrame.mY); put3(d.frameZ, a->mFrame.mZ);
            put3(d.intensity, a->mIntensity); d.invArea = a->mInvArea;
        }
        else if(const DirectionalLight *dl = dynamic_cast<const DirectionalLight*>(l))
        {
            d.type = VCM_LIGHT_DIRECTIONAL;
            put3(d.frameX, dl->mFrame.mX); put3(d.frameY, dl->mFrame.mY); put3(d.frameZ, dl->mFrame.mZ);
            put3(d.intensity, dl->mIntensity);
        }
        else if(const PointLight *pl = dynamic_cast<const PointLight*>(l))
        {
            d.type = VCM_LIGHT_POINT; put3(d.p0, pl->mPosition); put3(d.intensity, pl->mIntensity);
        }
        else if(const BackgroundLight *bl = dynamic_cast<const BackgroundLight*>(l))
        {
            d.type = VCM_LIGHT_BACKGROUND; put3(d.intensity, bl->mBackgroundColor); d.scale = bl->mScale;
            if(bl == aScene.mBackground) oDesc.backgroundLight = int(i);
        }
        else return -7;
    }
    oDesc.nPrims = int(oStorage.prims.size());         oDesc.prims = oStorage.prims.empty() ? NULL : &oStorage.prims[0];
    oDesc.nMaterials = int(oStorage.materials.size()); oDesc.materials = oStorage.materials.empty() ? NULL : &oStorage.materials[0];
    oDesc.mat2light = oStorage.mat2light.empty() ? NULL : &oStorage.mat2light[0];
    oDesc.nLights = int(oStorage.lights.size());       oDesc.lights = oStorage.lights.empty() ? NULL : &oStorage.lights[0];
    put3(oDesc.sceneCenter, aScene.mSceneSphere.mSceneCenter);
    oDesc.sceneRadius       = aScene.mSceneSphere.mSceneRadius;
    oDesc.invSceneRadiusSqr = aScene.mSceneSphere.mInvSceneRadiusSqr;
    const Camera &c = aScene.mCamera;
    put3(oDesc.camera.position, c.mPosition);
    put3(oDesc.camera.forward, c.mForward);
    oDesc.camera.resolution[0] = c.mResolution.x;
    oDesc.camera.resolution[1] = c.mResolution.y;
    memcpy(oDesc.camera.rasterToWorld, c.mRasterToWorld.GetPtr(), 16 * sizeof(float));
    memcpy(oDesc.camera.worldToRaster, c.mWorldToRaster.GetPtr(), 16 * sizeof(float));
    oDesc.camera.imagePlaneDist = c.mImagePlaneDist;
    return 0;
}

} // namespace smallvcm_amd

#endif
