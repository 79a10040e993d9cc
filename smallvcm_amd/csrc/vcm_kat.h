// vcm_kat.h -- function-level known-answer hooks (T0 of SURVEY.md section 8(c)): ONE call of a device function
// of vcm_core.h per record, host+device.  libsmallvcm_amd.so exposes it as vcm_debug_kat (a kernel, one lane per
// record; include/smallvcm_amd_debug.h), tests/host_emul runs the same function on the CPU, and oracle/ref_driver.cpp
// answers the same records with the reference's own classes -- so a parity break shows up at the function that
// caused it, not only as "the iteration differs".  Not used by the render path.
#ifndef SMALLVCM_AMD_VCM_KAT_H
#define SMALLVCM_AMD_VCM_KAT_H

#include "vcm_core.h"
#include "../../include/smallvcm_amd_debug.h"

namespace vcm {

template <class SC>
VCM_HD void kat_eval(const SC &sc, int op, const float *in, float *out)
{
    for (int i = 0; i < VCM_KAT_FLOATS; i++) out[i] = 0.f;
    switch (op) {
    case VCM_KAT_INTERSECT: {   /* Scene::Intersect scene.hxx:53-70 */
        Ray ray; ray.org = ld3(in); ray.dir = ld3(in + 3); ray.tmin = in[6];
        Isect is; is.dist = 1e36f; is.matID = 0; is.lightID = -1; is.normal = sp3(0.f);
        if (scene_intersect(sc, ray, is)) {
            out[0] = 1.f; out[1] = is.dist; out[2] = (float)is.matID; out[3] = (float)is.lightID;
            out[4] = is.normal.x; out[5] = is.normal.y; out[6] = is.normal.z;
        }
    } break;
    case VCM_KAT_OCCLUDED:   /* Scene::Occluded scene.hxx:72-85 */
        out[0] = scene_occluded(sc, ld3(in), ld3(in + 3), in[6]) ? 1.f : 0.f;
        break;
    case VCM_KAT_BSDF_EVAL: {   /* BSDF::Setup / Evaluate / Pdf / accessors, bsdf.hxx:95-180, :260-264 */
        Bsdf b;
        bsdf_setup(b, ld3(in), ld3(in + 3), (int)in[6], (int)in[11] - 1, sc);   /* in[11]: prim + 1, 0 = unknown (no table) */
        if (b.matID < 0) break;
        out[0] = 1.f; out[1] = b.isDelta ? 1.f : 0.f; out[2] = b.contProb;
        float cosGen = 0.f, dirPdf = 0.f, revPdf = 0.f;
        const V3 f = bsdf_evaluate(b, sc, ld3(in + 7), cosGen, &dirPdf, &revPdf);
        out[3] = f.x; out[4] = f.y; out[5] = f.z;
        if (!iszero(f)) out[6] = cosGen;   /* Evaluate leaves oCosThetaGen untouched on an early return */
        out[7] = dirPdf; out[8] = revPdf;
        out[9] = bsdf_pdf(b, sc, ld3(in + 7), false);
        out[10] = bsdf_pdf(b, sc, ld3(in + 7), true);
        const V3 w = to_world(b.frame, b.localDirFix);   /* WorldDirFix */
        out[11] = w.x; out[12] = w.y; out[13] = w.z; out[14] = b.localDirFix.z;   /* CosThetaFix */
    } break;
    case VCM_KAT_BSDF_SAMPLE: {   /* BSDF<FixIsLight>::Sample bsdf.hxx:191-257 */
        Bsdf b;
        bsdf_setup(b, ld3(in), ld3(in + 3), (int)in[6], (int)in[11] - 1, sc);   /* in[11]: prim + 1, 0 = unknown (no table) */
        if (b.matID < 0) break;
        out[0] = 1.f;
        V3 gen = sp3(0.f);
        float pdfW = 0.f, cosGen = 0.f;
        uint32_t ev = 0u;
        const V3 f = bsdf_sample(b, sc, in[10] != 0.f, in[7], in[8], in[9], gen, pdfW, cosGen, ev);
        if (iszero(f)) break;   /* the sample is discarded; the other outputs are unspecified */
        out[1] = f.x; out[2] = f.y; out[3] = f.z; out[4] = gen.x; out[5] = gen.y; out[6] = gen.z;
        out[7] = pdfW; out[8] = cosGen; out[9] = (float)ev;
    } break;
    case VCM_KAT_LIGHT_EMIT: {   /* AbstractLight::Emit lights.hxx:168, :267, :354, :438 */
        const vcm_light &l = get_light(sc, (int)in[0]);
        V3 pos = sp3(0.f), dir = sp3(0.f);
        float emissionPdfW = 0.f, directPdfA = 0.f, cosLight = 0.f;
        const V3 e = light_emit(l, sc, in[1], in[2], in[3], in[4], pos, dir, emissionPdfW, directPdfA, cosLight);
        out[0] = e.x; out[1] = e.y; out[2] = e.z; out[3] = pos.x; out[4] = pos.y; out[5] = pos.z;
        out[6] = dir.x; out[7] = dir.y; out[8] = dir.z; out[9] = emissionPdfW; out[10] = directPdfA; out[11] = cosLight;
        out[12] = light_is_finite(l) ? 1.f : 0.f; out[13] = light_is_delta(l) ? 1.f : 0.f;
    } break;
    case VCM_KAT_LIGHT_ILLUMINATE: {   /* AbstractLight::Illuminate lights.hxx:131, :244, :329, :410 */
        const vcm_light &l = get_light(sc, (int)in[0]);
        V3 dir = sp3(0.f);
        float dist = 0.f, directPdfW = 0.f, emissionPdfW = 0.f, cosAtLight = 0.f;
        const V3 r = light_illuminate(l, sc, ld3(in + 1), in[4], in[5], dir, dist, directPdfW, emissionPdfW, cosAtLight);
        out[0] = r.x; out[1] = r.y; out[2] = r.z; out[3] = dir.x; out[4] = dir.y; out[5] = dir.z; out[6] = dist;
        if (!iszero(r)) { out[7] = directPdfW; out[8] = emissionPdfW; out[9] = cosAtLight; }   /* unset on the early return */
    } break;
    case VCM_KAT_LIGHT_RADIANCE: {   /* AbstractLight::GetRadiance lights.hxx:198, :296, :377, :480 */
        const vcm_light &l = get_light(sc, (int)in[0]);
        float directPdfA = 0.f, emissionPdfW = 0.f;
        const V3 r = light_get_radiance(l, sc, ld3(in + 1), directPdfA, emissionPdfW);
        out[0] = r.x; out[1] = r.y; out[2] = r.z;
        if (!iszero(r)) { out[3] = directPdfA; out[4] = emissionPdfW; }
    } break;
    case VCM_KAT_CAMERA: {   /* Camera::GenerateRay / WorldToRaster / CheckRaster camera.hxx:95-117 */
        const vcm_camera &cam = sc.camera;
        const V3 worldRaster = transform_point(cam.rasterToWorld, mk3(in[0], in[1], 0.f));
        const V3 d = normalize(worldRaster - ld3(cam.position));
        out[0] = d.x; out[1] = d.y; out[2] = d.z;
        const V3 ip = transform_point(cam.worldToRaster, ld3(in + 2));
        out[3] = ip.x; out[4] = ip.y;
        out[5] = (ip.x >= 0 && ip.y >= 0 && ip.x < cam.resolution[0] && ip.y < cam.resolution[1]) ? 1.f : 0.f;
    } break;
    default: break;
    }
}

} // namespace vcm
#endif
