// vcm_render.cpp -- a C++ host over the C-ABI alone (include/smallvcm_amd.h; no
// reference headers, no Python): the driver options the reference's CLI does not
// expose.  smallvcm.cxx hard-wires 512x512, seed 1234,
// path lengths 0..10, one renderer per host core and CPU-seconds timing
// (src/config.hxx:233-240, src/smallvcm.cxx:66, :150); the benchmark and parity
// configurations need other resolutions, a fixed renderer count, raw fp32
// output and wall-clock time.
//
//   vcm_render -s <scene 0..3> -a <el|pt|lt|ppm|bpm|bpt|vcm> -i <iterations>
//              [--res W H] [--seed S] [--minlen A] [--maxlen B]
//              [--renderers R] [--radius-factor F] [--radius-alpha A]
//              [--device D] [--strict] [--warmup W] [-o out.pfm] [--json] [--scene-file f.vcmscene|f.obj]
//              [--gpus N [--shards S] [--inflight K] [--devices 0,1,..] [--collectives rccl|threads] [--same-window]]
//
// --gpus N: the multi-GPU host (vcm_farm.hpp): N ranks = N host threads, one per GPU, cut into N / S groups; a
// renderer lives on a group (S path-index shards, RCCL all-gather of the light vertices every iteration), K
// renderers take turns on a group, one RCCL all-reduce of the framebuffers at read-out.  --shards 1 = one renderer per
// GPU (the reference's own iteration-parallel scheme).  --collectives threads replaces RCCL by an in-process
// stand-in so that several ranks can share one GPU (tests; RCCL refuses two ranks on one device).
//
// -s / -a / -i keep the meaning they have in the reference's CLI
// (src/config.hxx:246-395; scenes = g_SceneConfigs[0..3], :146-151).
// --renderers R reproduces render() (src/smallvcm.cxx:52-151) with R "threads":
// renderer g has seed S+g and runs the iterations OpenMP's static schedule gives
// thread g; the image is the mean of the used renderers' means.  -o picks the
// format by extension like the reference (src/smallvcm.cxx:300-309): .bmp
// (gamma 2.2, Framebuffer::SaveBMP src/framebuffer.hxx:170-214), .hdr (SaveHDR
// :219-251), anything else raw fp32 PFM (SavePFM :137-146: "PF", "W H", "-1",
// rows top to bottom).  With one renderer the 8-bit formats are encoded on the
// device (vcm_read_image); with several, from the averaged framebuffer here.
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <string>
#include <vector>

#include "smallvcm_amd.h"
#include "vcm_farm.hpp"

static int die(const char *what)
{
    fprintf(stderr, "vcm_render: %s: %s\n", what, vcm_last_error());
    return 2;   // the reference's convention for fatal errors (src/config.hxx:140-141)
}

static int algorithm_by_acronym(const std::string &a)
{   // Config::GetAcronym, src/config.hxx:86-91
    if (a == "el") return VCM_ALGO_EYE_LIGHT;
    if (a == "pt") return VCM_ALGO_PATH_TRACE;
    if (a == "lt") return VCM_ALGO_LIGHT_TRACE;
    if (a == "ppm") return VCM_ALGO_PPM;
    if (a == "bpm") return VCM_ALGO_BPM;
    if (a == "bpt") return VCM_ALGO_BPT;
    if (a == "vcm") return VCM_ALGO_VCM;
    return -1;
}

int main(int argc, char **argv)
{
    int sceneID = 0, algorithm = VCM_ALGO_VCM, iterations = 1, resX = 512, resY = 512, seed = 1234;   // config.hxx:233-240
    unsigned minLen = 0, maxLen = 10;
    int renderers = 1, device = 0, warmup = 0, strict = 0, json = 0;
    int gpus = 0, shards = 0, inflight = 0, rccl = 1, sameWindow = 0, rcclRanks = 0;
    std::vector<float> rankMs;
    std::vector<int> devices;
    float radiusFactor = 0.003f, radiusAlpha = 0.75f;
    std::string out, algoName = "vcm", sceneFile;
    for (int i = 1; i < argc; i++) {
        const std::string a(argv[i]);
        auto need = [&](int n) { if (i + n >= argc) { fprintf(stderr, "vcm_render: %s needs %d argument(s)\n", a.c_str(), n); exit(2); } };
        if (a == "-s") { need(1); sceneID = atoi(argv[++i]); }
        else if (a == "-a") { need(1); algoName = argv[++i]; algorithm = algorithm_by_acronym(algoName); }
        else if (a == "-i") { need(1); iterations = atoi(argv[++i]); }
        else if (a == "-o") { need(1); out = argv[++i]; }
        else if (a == "--res") { need(2); resX = atoi(argv[++i]); resY = atoi(argv[++i]); }
        else if (a == "--seed") { need(1); seed = atoi(argv[++i]); }
        else if (a == "--minlen") { need(1); minLen = (unsigned)atoi(argv[++i]); }
        else if (a == "--maxlen") { need(1); maxLen = (unsigned)atoi(argv[++i]); }
        else if (a == "--renderers") { need(1); renderers = atoi(argv[++i]); }
        else if (a == "--radius-factor") { need(1); radiusFactor = (float)atof(argv[++i]); }
        else if (a == "--radius-alpha") { need(1); radiusAlpha = (float)atof(argv[++i]); }
        else if (a == "--device") { need(1); device = atoi(argv[++i]); }
        else if (a == "--warmup") { need(1); warmup = atoi(argv[++i]); }
        else if (a == "--gpus") { need(1); gpus = atoi(argv[++i]); }
        else if (a == "--shards") { need(1); shards = atoi(argv[++i]); }
        else if (a == "--inflight") { need(1); inflight = atoi(argv[++i]); }
        else if (a == "--collectives") { need(1); rccl = std::string(argv[++i]) == "threads" ? 0 : 1; }
        else if (a == "--devices") { need(1); for (const char *p = argv[++i]; *p;) { char *e; devices.push_back((int)strtol(p, &e, 10)); p = (*e == ',') ? e + 1 : e; if (e == p && *p) break; } }
        else if (a == "--same-window") sameWindow = 1;   // benchmark schedule: every renderer runs the iteration indices warmup ..
        else if (a == "--scene-file") { need(1); sceneFile = argv[++i]; }   // instead of -s: OBJ + MTL / .vcmscene (vcm_scene_load)
        else if (a == "--strict") strict = 1;
        else if (a == "--json") json = 1;
        else { fprintf(stderr, "vcm_render: unknown option %s (see the header of vcm_render.cpp)\n", a.c_str()); return 2; }
    }
    if (algorithm < 0 || sceneID < 0 || sceneID > 3 || iterations < 1 || resX < 1 || resY < 1 || renderers < 1) {
        fprintf(stderr, "vcm_render: invalid argument\n");
        return 2;
    }

    vcm_scene_desc scene;
    if (vcm_scene_cornell(resX, resY, vcm_scene_config_mask(sceneID), &scene)) return die("vcm_scene_cornell");
    vcm_scene_file *loaded = NULL;   // --scene-file: a version-2 description (any number of primitives, BVH)
    if (!sceneFile.empty()) {
        loaded = vcm_scene_load(sceneFile.c_str(), resX, resY);
        if (!loaded) { fprintf(stderr, "vcm_render: %s\n", vcm_scene_load_error()); return 2; }
        if (gpus > 0) { fprintf(stderr, "vcm_render: --scene-file with --gpus is not supported (the farm takes the built-in scenes)\n"); return 2; }
    }

    const size_t n3 = (size_t)resX * resY * 3;
    std::vector<float> fb(n3, 0.f), tmp(n3);
    double wall = 0;
    vcm_stats st;
    memset(&st, 0, sizeof(st));
    std::vector<vcm_ctx *> r;
    if (gpus > 0) {   // multi-GPU host: ranks x shards x in-flight renderers (vcm_farm.hpp)
        FarmConfig fc;
        fc.scene = scene; fc.algorithm = algorithm; fc.radiusFactor = radiusFactor; fc.radiusAlpha = radiusAlpha;
        fc.baseSeed = seed; fc.minLen = minLen; fc.maxLen = maxLen; fc.iterations = iterations; fc.ranks = gpus;
        fc.shards = shards > 0 ? shards : gpus;        // default: north_star's decomposition, one renderer across all GPUs
        fc.inflight = inflight > 0 ? inflight : 1;
        fc.rccl = rccl != 0; fc.warmup = warmup;
        fc.firstRank = 0; fc.localRanks = gpus;   // every rank is a thread of this process
        fc.sameWindow = sameWindow != 0;
        const int visible = vcm_device_count();
        if (visible <= 0) { fprintf(stderr, "vcm_render: no HIP device available (this program has no CPU path)\n"); return 2; }
        for (int k = 0; k < gpus; k++) fc.devices.push_back(k < (int)devices.size() ? devices[(size_t)k] : (fc.rccl ? k : k % visible));
        for (int d : fc.devices) if (d < 0 || d >= visible) { fprintf(stderr, "vcm_render: device %d not visible (%d device(s))\n", d, visible); return 2; }
        const FarmResult fr = farm_render(fc);
        if (!fr.error.empty()) { fprintf(stderr, "vcm_render: %s\n", fr.error.c_str()); return 2; }
        fb = fr.image;
        wall = fr.wallSeconds;
        renderers = fr.renderers;
        rcclRanks = fr.rcclRanks;
        rankMs = fr.rankIterationMs;
        st = fr.meanStats;
    } else {
    // render(): one renderer per "thread", seed base + i (smallvcm.cxx:61-72)
    r.assign((size_t)renderers, (vcm_ctx *)NULL);
    for (int g = 0; g < renderers; g++) {
        r[g] = loaded ? vcm_create_sharded2(vcm_scene_file_desc(loaded), algorithm, radiusFactor, radiusAlpha, seed + g, device, 0, 1)
                      : vcm_create_sharded(&scene, algorithm, radiusFactor, radiusAlpha, seed + g, device, 0, 1);
        if (!r[g]) return die("vcm_create");
        if (strict && vcm_set_strict_order(r[g], 1)) return die("vcm_set_strict_order");
    }
    // untimed warm-up on a throw-away renderer: allocations, first-launch costs
    if (warmup > 0) {
        vcm_ctx *w = loaded ? vcm_create_sharded2(vcm_scene_file_desc(loaded), algorithm, radiusFactor, radiusAlpha, seed, device, 0, 1)
                            : vcm_create_sharded(&scene, algorithm, radiusFactor, radiusAlpha, seed, device, 0, 1);
        if (!w) return die("vcm_create");
        for (int it = 0; it < warmup; it++) if (vcm_run_iteration(w, it, minLen, maxLen)) return die("vcm_run_iteration");
        vcm_synchronize(w);
        vcm_destroy(w);
    }

    const auto t0 = std::chrono::steady_clock::now();
    // static schedule of `#pragma omp parallel for` (smallvcm.cxx:98-108): contiguous blocks, the first
    // iterations % renderers threads get one more
    const int q = iterations / renderers, rem = iterations % renderers;
    for (int g = 0; g < renderers; g++) {
        const int lo = g * q + (g < rem ? g : rem), n = q + (g < rem ? 1 : 0);
        for (int it = lo; it < lo + n; it++)
            if (vcm_run_iteration(r[g], it, minLen, maxLen)) return die("vcm_run_iteration");
    }
    for (int g = 0; g < renderers; g++) if (vcm_synchronize(r[g])) return die("vcm_synchronize");
    wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // accumulate the used renderers: mean of (running sum / own iterations), smallvcm.cxx:116-142
    int used = 0;
    for (int g = 0; g < renderers; g++) {
        const int its = vcm_iterations(r[g]);
        if (its == 0) continue;                       // WasUsed(), renderer.hxx:58
        if (vcm_read_framebuffer(r[g], tmp.data())) return die("vcm_read_framebuffer");
        const float s = 1.f / its;                    // renderer.hxx:53-54
        if (used == 0) for (size_t i = 0; i < n3; i++) fb[i] = tmp[i] * s;
        else for (size_t i = 0; i < n3; i++) fb[i] = fb[i] + tmp[i] * s;   // Framebuffer::Add, framebuffer.hxx:75-79
        used++;
    }
    const float su = 1.f / used;                      // Framebuffer::Scale, smallvcm.cxx:142
    for (size_t i = 0; i < n3; i++) fb[i] = fb[i] * su;
    vcm_get_stats(r[0], &st);
    }

    if (!out.empty()) {
        const std::string ext = out.size() >= 4 ? out.substr(out.size() - 4) : "";
        const bool bmp = ext == ".bmp", hdr = ext == ".hdr";
        std::vector<unsigned char> px;
        if (bmp || hdr) {
            px.resize((size_t)resX * resY * (bmp ? 3 : 4));
            if (renderers == 1 && !r.empty()) {   // encoded on the device
                if (vcm_read_image(r[0], bmp ? VCM_IMAGE_BGR8 : VCM_IMAGE_RGBE, 1.f / vcm_iterations(r[0]), 2.2f, px.data()))
                    return die("vcm_read_image");
            } else if (bmp) {       // Framebuffer::SaveBMP, framebuffer.hxx:194-214
                const float invGamma = 1.f / 2.2f;
                for (int y = 0; y < resY; y++) for (int x = 0; x < resX; x++) {
                    const float *c = &fb[((size_t)x + (size_t)(resY - y - 1) * resX) * 3];
                    unsigned char *o = &px[((size_t)y * resX + x) * 3];
                    for (int k = 0; k < 3; k++)
                        o[k] = (unsigned char)std::min(255.f, std::max(0.f, std::pow(c[2 - k], invGamma) * 255.f));
                }
            } else {                // Framebuffer::SaveHDR, framebuffer.hxx:229-247
                for (size_t p = 0; p < (size_t)resX * resY; p++) {
                    const float *c = &fb[p * 3];
                    unsigned char *o = &px[p * 4];
                    o[0] = o[1] = o[2] = o[3] = 0;
                    float v = std::max(c[0], std::max(c[1], c[2]));
                    if (v >= 1e-32f) {
                        int e;
                        v = float(frexp(v, &e) * 256.f / v);
                        o[0] = (unsigned char)(c[0] * v); o[1] = (unsigned char)(c[1] * v); o[2] = (unsigned char)(c[2] * v);
                        o[3] = (unsigned char)(e + 128);
                    }
                }
            }
        }
        FILE *f = fopen(out.c_str(), "wb");
        if (!f) { fprintf(stderr, "vcm_render: cannot write %s\n", out.c_str()); return 2; }
        if (bmp) {   // BmpHeader, framebuffer.hxx:150-168, :175-191
            const uint32_t img = (uint32_t)resX * resY * 3;
            uint32_t h[13] = { 54u + img, 0u, 54u, 40u, (uint32_t)resX, (uint32_t)resY, 1u | (24u << 16), 0u, img, 2953u, 2953u, 0u, 0u };
            fwrite("BM", 1, 2, f);
            fwrite(h, 4, 13, f);
            fwrite(px.data(), 1, px.size(), f);
        } else if (hdr) {
            fprintf(f, "#?RADIANCE\n# SmallVCM\nFORMAT=32-bit_rle_rgbe\n\n-Y %d +X %d\n", resY, resX);
            fwrite(px.data(), 1, px.size(), f);
        } else {     // Framebuffer::SavePFM, framebuffer.hxx:137-146
            fprintf(f, "PF\n%d %d\n-1\n", resX, resY);
            fwrite(fb.data(), sizeof(float), n3, f);
        }
        fclose(f);
    }
    for (size_t g = 0; g < r.size(); g++) vcm_destroy(r[g]);
    vcm_scene_file_free(loaded);
    double mean[3] = { 0, 0, 0 };
    for (size_t i = 0; i < n3; i++) mean[i % 3] += fb[i];
    const double paths = (algorithm == VCM_ALGO_PATH_TRACE || algorithm == VCM_ALGO_EYE_LIGHT ? 1.0 : 2.0) * resX * resY * iterations;
    if (json) {
        std::string ms = "[";
        for (size_t i = 0; i < rankMs.size(); i++) { char b[32]; snprintf(b, sizeof(b), "%s%.3f", i ? ", " : "", rankMs[i]); ms += b; }
        ms += "]";
        printf("{\"scene\": %d, \"algorithm\": \"%s\", \"res\": [%d, %d], \"iterations\": %d, \"renderers\": %d, \"seed\": %d, "
               "\"gpus\": %d, \"rccl_ranks\": %d, \"wall_s\": %.6f, \"Mpaths_s\": %.3f, \"image_mean\": [%.6f, %.6f, %.6f], "
               "\"last_iteration_ms\": %.3f, \"rank_iteration_ms\": %s, \"library\": \"%s\", "
               "\"last_iteration_kernel_ms\": {\"light\": %.3f, \"camera\": %.3f, \"connect_di\": %.3f, \"merge\": %.3f, \"grid_side\": %.3f, \"light_phase\": %.3f, \"camera_phase\": %.3f}, "
               "\"last_iteration_counters\": {\"lightVertices\": %lld, \"lightRays\": %lld, \"cameraRays\": %lld, \"shadowRays\": %lld, "
               "\"mergeQueries\": %lld, \"mergeCandidates\": %lld, \"mergeAccepted\": %lld, \"connections\": %lld, \"lightSplats\": %lld}}\n",
               sceneID, algoName.c_str(), resX, resY, iterations, renderers, seed, gpus > 0 ? gpus : 1, rcclRanks, wall, paths / wall / 1e6,
               mean[0] / (n3 / 3), mean[1] / (n3 / 3), mean[2] / (n3 / 3), st.msTotal, ms.c_str(), vcm_build_tag(),
               st.msLightKernel, st.msCameraKernel, st.msConnectKernels, st.msMergeKernel, st.msGrid, st.msLight, st.msCamera,
               st.lightVertices, st.lightRays, st.cameraRays, st.shadowRays, st.mergeQueries, st.mergeCandidates, st.mergeAccepted,
               st.connections, st.lightSplats);
    }
    else
        printf("scene %d, %s, %dx%d, %d iteration(s) on %d renderer(s): %.3f s wall clock, %.2f Mpaths/s\n", sceneID,
               algoName.c_str(), resX, resY, iterations, renderers, wall, paths / wall / 1e6);
    return 0;
}
