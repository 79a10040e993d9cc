// dropin_rate.cpp -- what the reference's OWN renderer interface costs at a given resolution: Mpaths/s of
//   (a) VertexCM::RunIteration through the drop-in (dropin/vertexcm.hxx -> gpu_renderer.hxx), which leaves the host
//       Framebuffer equal to the running sum after EVERY call (renderer.hxx:49-55 reads it without a hook), and
//   (b) the same iterations through the C-ABI alone (vcm_run_iteration, one synchronisation at the end),
// in one process on one GPU, same scene object, same seed, same iteration window -- and whether the two frames agree.
// The reference's CLI has no resolution switch (config.hxx:237), so the Config is built here the way render()'s caller
// would (config.hxx:112-142, smallvcm.cxx:52-72); nothing of the reference's integrator is compiled (vertexcm.hxx is
// the drop-in's).  Test-side harness: built by dropin/Makefile where /root/reference exists, the binary travels.
//
//   dropin_rate <res> [iterations=20] [warmup=5] [scene=1] [algorithm: vcm|bpm|bpt|ppm|lt]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "math.hxx"
#include "ray.hxx"
#include "geometry.hxx"
#include "camera.hxx"
#include "framebuffer.hxx"
#include "scene.hxx"
#include "vertexcm.hxx"   // the drop-in's (this directory), found before the reference's
#include "config.hxx"     // g_SceneConfigs (config.hxx:146-151)

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main(int argc, char **argv)
{
    const int res = argc > 1 ? atoi(argv[1]) : 2048;
    const int iters = argc > 2 ? atoi(argv[2]) : 20;
    const int warm = argc > 3 ? atoi(argv[3]) : 5;
    const int sceneId = argc > 4 ? atoi(argv[4]) : 1;
    const char *algoName = argc > 5 ? argv[5] : "vcm";
    VertexCM::AlgorithmType algo = VertexCM::kVcm;
    if (!strcmp(algoName, "bpm")) algo = VertexCM::kBpm; else if (!strcmp(algoName, "bpt")) algo = VertexCM::kBpt;
    else if (!strcmp(algoName, "ppm")) algo = VertexCM::kPpm; else if (!strcmp(algoName, "lt")) algo = VertexCM::kLightTrace;
    if (res < 1 || iters < 1 || sceneId < 0 || sceneId > 3) { fprintf(stderr, "usage: dropin_rate <res> [iterations] [warmup] [scene 0..3] [algorithm]\n"); return 2; }

    Scene scene;                                                          // config.hxx:296-305
    scene.LoadCornellBox(Vec2i(res, res), g_SceneConfigs[sceneId]);
    scene.BuildSceneSphere();
    const double paths = 2.0 * res * (double)res;

    // (a) the drop-in: the reference's interface, host framebuffer current after every call
    double tDrop;
    Framebuffer fbDrop;
    {
        VertexCM r(scene, algo, 0.003f, 0.75f, 1234);                    // config.hxx:239-240, :234
        r.mMaxPathLength = 10; r.mMinPathLength = 0;                      // smallvcm.cxx:70-71
        for (int i = 0; i < warm; i++) r.RunIteration(i);
        const double t0 = now();
        for (int i = 0; i < iters; i++) r.RunIteration(warm + i);
        tDrop = now() - t0;
        r.GetFramebuffer(fbDrop);                                         // renderer.hxx:49-55
    }
    // (b) the C-ABI alone on the same scene description
    double tAbi;
    std::vector<float> fbAbi((size_t)res * res * 3);
    {
        VertexCM r(scene, algo, 0.003f, 0.75f, 1234);
        vcm_ctx *c = r.Context();
        for (int i = 0; i < warm; i++) if (vcm_run_iteration(c, i, 0, 10)) { fprintf(stderr, "%s\n", vcm_last_error()); return 1; }
        if (vcm_synchronize(c)) return 1;
        const double t0 = now();
        for (int i = 0; i < iters; i++) if (vcm_run_iteration(c, warm + i, 0, 10)) { fprintf(stderr, "%s\n", vcm_last_error()); return 1; }
        if (vcm_synchronize(c)) return 1;
        tAbi = now() - t0;
        if (vcm_read_framebuffer(c, fbAbi.data())) return 1;
    }
    // the drop-in's frame = running sum / iterations (incl. warm-up), the C-ABI's = the running sum
    const float scale = 1.f / float(warm + iters);
    size_t differ = 0;
    Framebuffer probe; probe.Setup(Vec2f(float(res), float(res)));
    // Framebuffer has no pixel getter: compare through SavePFM-free means -- add the negated C-ABI frame and sum |.|
    for (int y = 0; y < res; y++)
        for (int x = 0; x < res; x++) {
            const float *p = &fbAbi[((size_t)y * res + x) * 3];
            probe.AddColor(Vec2f(x + 0.5f, y + 0.5f), Vec3f(-(p[0] * scale), -(p[1] * scale), -(p[2] * scale)));
        }
    probe.Add(fbDrop);
    const float residual = probe.TotalLuminance();   // framebuffer.hxx:89: exactly 0 when every pixel cancelled
    (void)differ;
    printf("{\"res\": %d, \"scene\": %d, \"algorithm\": \"%s\", \"iterations\": %d, \"warmup\": %d, "
           "\"dropin_Mpaths_s\": %.2f, \"dropin_ms_per_iteration\": %.3f, \"cabi_Mpaths_s\": %.2f, \"cabi_ms_per_iteration\": %.3f, "
           "\"dropin_over_cabi\": %.4f, \"frame_residual_luminance\": %g, \"refresh\": \"%s\"}\n",
           res, sceneId, algoName, iters, warm, paths * iters / tDrop / 1e6, tDrop / iters * 1e3, paths * iters / tAbi / 1e6, tAbi / iters * 1e3,
           tAbi / tDrop, (double)residual, smallvcm_amd::GpuRenderer::RefreshKind());
    return residual == 0.f ? 0 : 3;
}
