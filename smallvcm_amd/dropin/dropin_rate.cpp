// dropin_rate.cpp -- what the reference's OWN renderer interface costs at a given resolution: Mpaths/s of
//   (a) VertexCM::RunIteration through the drop-in (dropin/vertexcm.hxx -> gpu_renderer.hxx), which leaves the host
//       Framebuffer equal to the running sum after EVERY call (renderer.hxx:49-55 reads it without a hook), and
//   (b) the same iterations through the C-ABI alone (vcm_run_iteration, one synchronisation at the end),
// in one process on one GPU, same scene object, same seed, same iteration window -- and whether the two frames agree.
// The reference's CLI has no resolution switch (config.hxx:237), so the Config is built here the way render()'s caller
// would (config.hxx:112-142, smallvcm.cxx:52-72); nothing of the reference's integrator is compiled (vertexcm.hxx is
// the drop-in's).  Test-side harness: built by dropin/Makefile where /root/reference exists, the binary travels.
//
// With `renderers` > 1 the iterations are dealt to that many renderers on as many host threads, the way render() does it
// (smallvcm.cxx:61-72, :99-108: one renderer per host core, `#pragma omp parallel for` over the iterations): one renderer's
// read-out then crosses PCIe while another one's kernels run.
//
//   dropin_rate <res> [iterations=20] [warmup=5] [scene=1] [algorithm: vcm|bpm|bpt|ppm|lt] [renderers=1]
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <omp.h>

#include "math.hxx"
#include "ray.hxx"
#include "geometry.hxx"
#include "camera.hxx"
#include "framebuffer.hxx"
#include "scene.hxx"
#include "vertexcm.hxx"   // the drop-in's (this directory), found before the reference's
#include "config.hxx"     // g_SceneConfigs (config.hxx:146-151)

// the harness reads the Framebuffer's texels to compare frames bit for bit (Framebuffer has no pixel getter; the same
// explicit-instantiation door as dropin/gpu_renderer.hxx, under a tag of its own)
template <typename Tag, typename Tag::type Member> struct HarnessDoor { friend typename Tag::type harness_member(Tag) { return Member; } };
struct HarnessColors { typedef std::vector<Vec3f> Framebuffer::*type; friend type harness_member(HarnessColors); };
template struct HarnessDoor<HarnessColors, &Framebuffer::mColor>;

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

    const int R = argc > 6 ? atoi(argv[6]) : 1;
    if (R > 1) {   // render()'s scheme: R renderers (seeds 1234 + t), the timed iterations dealt out by schedule(static)
        double t[2];
        for (int pass = 0; pass < 2; pass++) {   // 0: through the drop-in, 1: the C-ABI alone
            std::vector<VertexCM *> rs;
            for (int k = 0; k < R; k++) { rs.push_back(new VertexCM(scene, algo, 0.003f, 0.75f, 1234 + k)); rs[k]->mMaxPathLength = 10; rs[k]->mMinPathLength = 0; }
            // warm-up on the renderers' own threads, at the same time: the device's pool of iteration-scratch arenas grows to
            // what R iterations in flight need (17 GB each at 2048^2) before the clock starts
#pragma omp parallel for num_threads(R) schedule(static, 1)
            for (int k = 0; k < R; k++)
                for (int i = 0; i < (warm > 2 ? warm : 2); i++) { if (pass == 0) rs[k]->RunIteration(i); else vcm_run_iteration(rs[k]->Context(), i, 0, 10); }
            for (int k = 0; k < R; k++) vcm_synchronize(rs[k]->Context());
            const double t0 = now();
#pragma omp parallel for num_threads(R) schedule(static)
            for (int i = 0; i < iters; i++) {
                VertexCM *r = rs[omp_get_thread_num()];
                if (pass == 0) r->RunIteration(warm + i); else vcm_run_iteration(r->Context(), warm + i, 0, 10);
            }
            for (int k = 0; k < R; k++) vcm_synchronize(rs[k]->Context());
            t[pass] = now() - t0;
            for (int k = 0; k < R; k++) delete rs[k];
        }
        printf("{\"res\": %d, \"scene\": %d, \"algorithm\": \"%s\", \"iterations\": %d, \"warmup\": %d, \"renderers\": %d, "
               "\"dropin_Mpaths_s\": %.2f, \"cabi_Mpaths_s\": %.2f, \"dropin_over_cabi\": %.4f, \"refresh\": \"%s\"}\n",
               res, sceneId, algoName, iters, warm, R, paths * iters / t[0] / 1e6, paths * iters / t[1] / 1e6, t[1] / t[0],
               smallvcm_amd::GpuRenderer::RefreshKind());
        return 0;
    }

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
    // the drop-in's frame = running sum * (1 / iterations) (renderer.hxx:53-54, warm-up included), the C-ABI's = the running sum
    const float scale = 1.f / float(warm + iters);
    const std::vector<Vec3f> &px = fbDrop.*harness_member(HarnessColors());
    size_t differ = 0;
    for (size_t i = 0; i < px.size(); i++) {
        const Vec3f want = Vec3f(fbAbi[3 * i], fbAbi[3 * i + 1], fbAbi[3 * i + 2]) * Vec3f(scale);   // framebuffer.hxx:81-85
        if (memcmp(&px[i], &want, sizeof(Vec3f)) != 0) differ++;
    }
    const double residual = (double)differ;
    printf("{\"res\": %d, \"scene\": %d, \"algorithm\": \"%s\", \"iterations\": %d, \"warmup\": %d, "
           "\"dropin_Mpaths_s\": %.2f, \"dropin_ms_per_iteration\": %.3f, \"cabi_Mpaths_s\": %.2f, \"cabi_ms_per_iteration\": %.3f, "
           "\"dropin_over_cabi\": %.4f, \"pixels_that_differ\": %g, \"refresh\": \"%s\"}\n",
           res, sceneId, algoName, iters, warm, paths * iters / tDrop / 1e6, tDrop / iters * 1e3, paths * iters / tAbi / 1e6, tAbi / iters * 1e3,
           tAbi / tDrop, (double)residual, smallvcm_amd::GpuRenderer::RefreshKind());
    return differ == 0 ? 0 : 3;
}
