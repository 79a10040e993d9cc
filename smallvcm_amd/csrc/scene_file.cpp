// scene_file.cpp -- scene files for the version-2 scene description (include/smallvcm_amd.h: vcm_scene_load).
//
// The reference has no loader: "Scenes are hard-coded" (README:210), Scene::LoadCornellBox (src/scene.hxx:132-398) is
// the only way a scene comes into being.  This one reads the common denominator of what people have -- Wavefront OBJ
// geometry with an MTL material library -- plus a small text file for what OBJ cannot say (camera, analytic spheres,
// non-area lights), and produces a vcm_scene_desc2 with the library's own constructors (vcm_make_triangle /
// vcm_make_area_light / vcm_make_camera / ...: the reference's constructors restated, scene_cornell.cpp), so a loaded
// scene holds exactly what the reference's Scene would hold for the same numbers.
//
//   .vcmscene   one directive per line, processed IN ORDER (primitive order = list order = tie-break order of
//               Scene::Intersect, geometry.hxx:65-78); '#' starts a comment; paths are relative to the file
//       obj <file.obj>                                   triangles of a Wavefront OBJ (with its mtllib)
//       mtllib <file.mtl>                                materials for the directives below (spheres)
//       sphere cx cy cz radius <material>                Sphere (geometry.hxx:184-192)
//       camera px py pz  fx fy fz  ux uy uz  hfovDeg     Camera::Setup (camera.hxx:37-76)
//       light point px py pz  r g b                      PointLight (lights.hxx:324-328)
//       light directional dx dy dz  r g b                DirectionalLight (lights.hxx:239-243)
//       light background scale                           BackgroundLight (lights.hxx:404-408)
//   .obj        v, f (triangles; polygons are fanned around their first vertex; v, v/vt, v/vt/vn, v//vn; negative
//               = relative indices), usemtl, mtllib; everything else is skipped
//   .mtl        newmtl, Kd -> mDiffuseReflectance, Ks + Ns -> mPhongReflectance / mPhongExponent (materials.hxx:54-65),
//               illum 3 | 5 -> mirror: mMirrorReflectance = Ks;  illum 4 | 6 | 7 | 9 -> glass: mIOR = Ni and
//               mMirrorReflectance = Ks;  Ke != 0 -> every triangle of the material is an AreaLight of that intensity:
//               a primitive with a material of its own whose mat2light entry names the light, and no reflectance,
//               as the reference's light box (scene.hxx:333-361)
// A bare .obj loads too (default camera: in front of the bounding sphere, looking along +y, z up, 45 degrees).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <vector>

#include "smallvcm_amd.h"

struct vcm_scene_file {
    std::vector<vcm_prim> prims;
    std::vector<vcm_material> materials;
    std::vector<int> mat2light;
    std::vector<vcm_light> lights;
    vcm_scene_desc2 desc;
};

namespace {

thread_local std::string g_sceneError;

struct MtlEntry { vcm_material m; float ke[3]; bool emissive; int index; /* in materials, -1: not added yet */ };

struct Loader {
    vcm_scene_file *out;
    std::map<std::string, MtlEntry> mtl;
    int background;
    bool haveCamera;
    float camPos[3], camFwd[3], camUp[3], camFov;
    Loader() : out(NULL), background(-1), haveCamera(false), camFov(45.f) {}

    bool fail(const std::string &what) { g_sceneError = what; return false; }

    static std::string dir_of(const std::string &path)
    {
        const size_t k = path.find_last_of("/\\");
        return k == std::string::npos ? std::string() : path.substr(0, k + 1);
    }
    static bool ends_with(const std::string &s, const char *suffix)
    {
        const size_t n = strlen(suffix);
        return s.size() >= n && s.compare(s.size() - n, n, suffix) == 0;
    }
    static bool floats(const char *&p, float *v, int n)
    {
        for (int i = 0; i < n; i++) {
            char *e = NULL;
            v[i] = strtof(p, &e);
            if (e == p) return false;
            p = e;
        }
        return true;
    }
    static std::string word(const char *&p)
    {
        while (*p == ' ' || *p == '\t') p++;
        const char *b = p;
        while (*p && *p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') p++;
        return std::string(b, p);
    }

    int material_index(const std::string &name)
    {   // materials enter the scene when first used: unused library entries cost nothing
        std::map<std::string, MtlEntry>::iterator it = mtl.find(name);
        if (it == mtl.end()) return -1;
        if (it->second.index < 0) {
            it->second.index = (int)out->materials.size();
            out->materials.push_back(it->second.m);
            out->mat2light.push_back(-1);
        }
        return it->second.index;
    }

    /* one whole line of any length (a polygon with a thousand vertices is one `f` line: a fixed buffer would cut it in
       the middle of an index and read the tail as an unknown key) */
    static bool read_line(FILE *f, std::string &line)
    {
        /* character by character (getc is buffered): a NUL byte inside a line is data like any other -- fgets + strlen lost
           everything between it and the end of the chunk and then glued the next physical line onto this one */
        line.clear();
        bool any = false;
        for (int ch; (ch = getc(f)) != EOF;) {
            any = true;
            line += (char)ch;
            if (ch == '\n') break;
        }
        return any;
    }

    bool load_mtl(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return fail("cannot open " + path);
        std::string lineBuf;
        MtlEntry *cur = NULL;
        int illum = 2;
        float ks[3] = { 0, 0, 0 }, ni = 1.f;
        bool haveKs = false;
        auto finish = [&]() {
            if (!cur) return;
            const bool mirror = illum == 3 || illum == 5, glass = illum == 4 || illum == 6 || illum == 7 || illum == 9;
            if (mirror || glass) {
                for (int k = 0; k < 3; k++) { cur->m.mirror[k] = haveKs ? ks[k] : 1.f; cur->m.phong[k] = 0.f; }
                if (glass) cur->m.ior = ni;
            } else if (haveKs) {
                for (int k = 0; k < 3; k++) cur->m.phong[k] = ks[k];
            }
        };
        while (read_line(f, lineBuf)) {
            const char *p = lineBuf.c_str();
            const std::string key = word(p);
            if (key.empty() || key[0] == '#') continue;
            if (key == "newmtl") {
                finish();
                const std::string name = word(p);
                if (mtl.count(name)) { fclose(f); return fail(path + ": material '" + name + "' is defined twice"); }
                MtlEntry e;
                vcm_make_material(&e.m);
                e.ke[0] = e.ke[1] = e.ke[2] = 0.f; e.emissive = false; e.index = -1;
                cur = &(mtl[name] = e);
                illum = 2; haveKs = false; ni = 1.f;
            } else if (!cur) {
                continue;
            } else if (key == "Kd") { if (!floats(p, cur->m.diffuse, 3)) { fclose(f); return fail(path + ": bad Kd"); } }
            else if (key == "Ks") { if (!floats(p, ks, 3)) { fclose(f); return fail(path + ": bad Ks"); } haveKs = true; }
            else if (key == "Ns") { if (!floats(p, &cur->m.phongExp, 1)) { fclose(f); return fail(path + ": bad Ns"); } }
            else if (key == "Ni") { if (!floats(p, &ni, 1)) { fclose(f); return fail(path + ": bad Ni"); } }
            else if (key == "illum") { illum = atoi(p); }
            else if (key == "Ke") {
                if (!floats(p, cur->ke, 3)) { fclose(f); return fail(path + ": bad Ke"); }
                cur->emissive = cur->ke[0] != 0.f || cur->ke[1] != 0.f || cur->ke[2] != 0.f;
            }
        }
        finish();
        fclose(f);
        return true;
    }

    void add_triangle(const float *a, const float *b, const float *c, const std::string &mtlName, const MtlEntry *e)
    {
        int mat;
        if (e && e->emissive) {   // scene.hxx:333-361: its own material, mat2light -> the AreaLight over the same points
            vcm_material m;
            vcm_make_material(&m);
            mat = (int)out->materials.size();
            out->materials.push_back(m);
            vcm_light l;
            vcm_make_area_light(a, b, c, e->ke, &l);
            out->mat2light.push_back((int)out->lights.size());
            out->lights.push_back(l);
        } else {
            mat = material_index(mtlName);
        }
        vcm_prim p;
        vcm_make_triangle(a, b, c, mat, &p);
        out->prims.push_back(p);
    }

    bool load_obj(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return fail("cannot open " + path);
        std::vector<float> v;
        std::string current;
        const MtlEntry *cur = NULL;
        std::string lineBuf;
        long lineNo = 0;
        while (read_line(f, lineBuf)) {
            lineNo++;
            const char *p = lineBuf.c_str();
            const std::string key = word(p);
            if (key == "v") {
                float x[3];
                if (!floats(p, x, 3)) { fclose(f); return fail(path + ": bad vertex at line " + std::to_string(lineNo)); }
                v.insert(v.end(), x, x + 3);
            } else if (key == "mtllib") {
                if (!load_mtl(dir_of(path) + word(p))) { fclose(f); return false; }
            } else if (key == "usemtl") {
                current = word(p);
                std::map<std::string, MtlEntry>::const_iterator it = mtl.find(current);
                if (it == mtl.end()) { fclose(f); return fail(path + ": unknown material " + current); }
                cur = &it->second;
            } else if (key == "f") {
                std::vector<long> idx;
                for (;;) {
                    const std::string w = word(p);
                    if (w.empty()) break;
                    long i = strtol(w.c_str(), NULL, 10);   // the vertex index is what precedes the first '/'
                    const long n = (long)(v.size() / 3);
                    if (i < 0) i = n + i + 1;
                    if (i < 1 || i > n) { fclose(f); return fail(path + ": face index out of range at line " + std::to_string(lineNo)); }
                    idx.push_back(i - 1);
                }
                if (idx.size() < 3) { fclose(f); return fail(path + ": face with fewer than 3 vertices at line " + std::to_string(lineNo)); }
                if (!cur) { fclose(f); return fail(path + ": face before any usemtl at line " + std::to_string(lineNo)); }
                for (size_t k = 1; k + 1 < idx.size(); k++)
                    add_triangle(&v[3 * idx[0]], &v[3 * idx[k]], &v[3 * idx[k + 1]], current, cur);
            }
        }
        fclose(f);
        return true;
    }

    bool load_scene(const std::string &path)
    {
        FILE *f = fopen(path.c_str(), "r");
        if (!f) return fail("cannot open " + path);
        const std::string base = dir_of(path);
        std::string lineBuf;
        long lineNo = 0;
        bool ok = true;
        while (ok && read_line(f, lineBuf)) {
            lineNo++;
            const char *p = lineBuf.c_str();
            const std::string key = word(p);
            const std::string at = path + " line " + std::to_string(lineNo);
            if (key.empty() || key[0] == '#') continue;
            if (key == "obj") ok = load_obj(base + word(p));
            else if (key == "mtllib") ok = load_mtl(base + word(p));
            else if (key == "sphere") {
                float x[4];
                if (!floats(p, x, 4)) { ok = fail(at + ": sphere cx cy cz radius material"); break; }
                const int mat = material_index(word(p));
                if (mat < 0) { ok = fail(at + ": unknown material"); break; }
                vcm_prim s;
                vcm_make_sphere(x, x[3], mat, &s);
                out->prims.push_back(s);
            } else if (key == "camera") {
                float x[10];
                if (!floats(p, x, 10)) { ok = fail(at + ": camera px py pz fx fy fz ux uy uz fov"); break; }
                memcpy(camPos, x, 12); memcpy(camFwd, x + 3, 12); memcpy(camUp, x + 6, 12); camFov = x[9];
                haveCamera = true;
            } else if (key == "light") {
                const std::string kind = word(p);
                vcm_light l;
                float x[6];
                if (kind == "point" && floats(p, x, 6)) vcm_make_point_light(x, x + 3, &l);
                else if (kind == "directional" && floats(p, x, 6)) vcm_make_directional_light(x, x + 3, &l);
                else if (kind == "background" && floats(p, x, 1)) { vcm_make_background_light(x[0], &l); background = (int)out->lights.size(); }
                else { ok = fail(at + ": light point|directional x y z r g b, or light background scale"); break; }
                out->lights.push_back(l);
            } else ok = fail(at + ": unknown directive " + key);
        }
        fclose(f);
        return ok;
    }

    bool finish(int resX, int resY)
    {
        if (out->prims.empty()) return fail("the scene has no primitives");
        if (out->lights.empty()) return fail("the scene has no light (an MTL material with Ke, or a light directive)");
        vcm_scene_desc2 &d = out->desc;
        memset(&d, 0, sizeof(d));
        d.nPrims = (int)out->prims.size(); d.prims = out->prims.data();
        d.nMaterials = (int)out->materials.size(); d.materials = out->materials.data(); d.mat2light = out->mat2light.data();
        d.nLights = (int)out->lights.size(); d.lights = out->lights.data();
        d.backgroundLight = background;
        vcm_make_scene_sphere(d.prims, d.nPrims, d.sceneCenter, &d.sceneRadius, &d.invSceneRadiusSqr);
        if (!haveCamera) {
            camPos[0] = d.sceneCenter[0]; camPos[1] = d.sceneCenter[1] - 2.6f * d.sceneRadius; camPos[2] = d.sceneCenter[2];
            camFwd[0] = 0.f; camFwd[1] = 1.f; camFwd[2] = 0.f;
            camUp[0] = 0.f; camUp[1] = 0.f; camUp[2] = 1.f;
            camFov = 45.f;
        }
        if (vcm_make_camera(camPos, camFwd, camUp, camFov, resX, resY, &d.camera) != 0) return fail("bad camera");
        return true;
    }
};

} // namespace

extern "C" {

const char *vcm_scene_load_error(void) { return g_sceneError.c_str(); }

vcm_scene_file *vcm_scene_load(const char *path, int resX, int resY)
{
    g_sceneError.clear();
    if (!path || resX < 1 || resY < 1) { g_sceneError = "vcm_scene_load: bad argument"; return NULL; }
    vcm_scene_file *s = new (std::nothrow) vcm_scene_file();
    if (!s) { g_sceneError = "out of memory"; return NULL; }
    bool ok = false;
    try {   /* nothing may be thrown through the C-ABI (std::string / std::vector / std::map allocate) */
        Loader ld;
        ld.out = s;
        const std::string p(path);
        ok = (Loader::ends_with(p, ".obj") ? ld.load_obj(p) : ld.load_scene(p)) && ld.finish(resX, resY);
    } catch (const std::exception &e) {
        try { g_sceneError = std::string("vcm_scene_load: ") + e.what(); } catch (...) { try { g_sceneError.assign("vcm_scene_load: exception"); } catch (...) {} }
        ok = false;
    } catch (...) {
        try { g_sceneError = "vcm_scene_load: exception"; } catch (...) {}   /* never a stale message from an earlier call */
        ok = false;
    }
    if (!ok) { delete s; return NULL; }
    return s;
}

const vcm_scene_desc2 *vcm_scene_file_desc(const vcm_scene_file *s) { return s ? &s->desc : NULL; }

void vcm_scene_file_free(vcm_scene_file *s) { delete s; }

} // extern "C"
