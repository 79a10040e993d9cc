// gpu_renderer.hxx -- the part the three drop-in renderer classes share: an
// AbstractRenderer (src/renderer.hxx:33-70) whose RunIteration forwards to the
// MI355X library through the C-ABI (include/smallvcm_amd.h).
#ifndef SMALLVCM_AMD_DROPIN_GPU_RENDERER_HXX
#define SMALLVCM_AMD_DROPIN_GPU_RENDERER_HXX

#include <vector>
#include <map>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <string>    // scene.hxx uses std::string but relies on the unity-build include order
// reference headers this shim builds on (found via -I<SmallVCM>/src)
#include "math.hxx"
#include "frame.hxx"
#include "ray.hxx"
#include "utils.hxx"
#include "renderer.hxx"     // AbstractRenderer, Scene, Framebuffer
#include "smallvcm_amd.h"
#include "flatten_scene.hxx"

namespace smallvcm_amd {

class GpuRenderer : public AbstractRenderer
{
public:

    // aAlgorithm: VCM_ALGO_* (include/smallvcm_amd.h)
    GpuRenderer(
        const Scene&  aScene,
        int           aAlgorithm,
        const float   aRadiusFactor,
        const float   aRadiusAlpha,
        int           aSeed
    ) :
        AbstractRenderer(aScene),
        mCtx(NULL)
    {
        vcm_scene_desc desc;
        const int rc = FlattenScene(aScene, desc);
        if(rc == 0)
        {
            // render() builds one renderer per host core (smallvcm.cxx:61-72): dealt round-robin over the GPUs of the node
            mCtx = vcm_create_sharded(&desc, aAlgorithm, aRadiusFactor, aRadiusAlpha, aSeed, vcm_next_device(), 0, 1);
        }
        else
        {
            // more primitives / materials / lights than the reference's built-in boxes have: the version-2
            // description (traced through a BVH beyond 32 primitives)
            SceneArrays arrays;
            vcm_scene_desc2 desc2;
            const int rc2 = FlattenScene2(aScene, arrays, desc2);
            if(rc2 != 0)
            {
                // same convention as the reference's factory (src/config.hxx:140-141)
                fprintf(stderr, "smallvcm_amd: scene cannot be flattened (code %d)\n", rc2);
                exit(2);
            }
            mCtx = vcm_create_sharded2(&desc2, aAlgorithm, aRadiusFactor, aRadiusAlpha, aSeed, vcm_next_device(), 0, 1);
        }
        if(mCtx == NULL)
        {
            fprintf(stderr, "smallvcm_amd: %s\n", vcm_last_error());
            exit(2);
        }

        mResX = int(aScene.mCamera.mResolution.x);
        mResY = int(aScene.mCamera.mResolution.y);
    }

    virtual ~GpuRenderer()
    {
        vcm_destroy(mCtx);
    }

    // src/vertexcm.hxx:284-548, src/pathtracer.hxx:45-215, src/eyelight.hxx:46-77.
    // mMaxPathLength / mMinPathLength are assigned by the driver after
    // construction (src/smallvcm.cxx:70-71), so they are read here, per call.
    virtual void RunIteration(int aIteration)
    {
        if(vcm_run_iteration(mCtx, aIteration, mMinPathLength, mMaxPathLength) != 0)
        {
            fprintf(stderr, "smallvcm_amd: %s\n", vcm_last_error());
            exit(2);
        }

        // AbstractRenderer::GetFramebuffer (src/renderer.hxx:49-55) is not
        // virtual and reads the protected host mFramebuffer, which must hold
        // the running SUM over iterations.  Framebuffer has no bulk setter
        // (src/framebuffer.hxx:253-258): refresh it with Clear + one AddColor
        // per pixel (0 + c == c, so the copy is exact).
        mHost.resize(size_t(mResX) * mResY * 3);
        if(vcm_read_framebuffer(mCtx, &mHost[0]) != 0)
        {
            fprintf(stderr, "smallvcm_amd: %s\n", vcm_last_error());
            exit(2);
        }

        mFramebuffer.Clear();
        for(int y = 0; y < mResY; y++)
        {
            for(int x = 0; x < mResX; x++)
            {
                const float *c = &mHost[(size_t(y) * mResX + x) * 3];
                mFramebuffer.AddColor(Vec2f(x + 0.5f, y + 0.5f), Vec3f(c[0], c[1], c[2]));
            }
        }

        mIterations++;   // vertexcm.hxx:547, pathtracer.hxx:216, eyelight.hxx:79
    }

    vcm_ctx* Context() { return mCtx; }

private:

    vcm_ctx            *mCtx;
    int                 mResX, mResY;
    std::vector<float>  mHost;
};

} // namespace smallvcm_amd

#endif //SMALLVCM_AMD_DROPIN_GPU_RENDERER_HXX
