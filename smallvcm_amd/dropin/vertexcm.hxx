// vertexcm.hxx -- drop-in replacement for SmallVCM's src/vertexcm.hxx.
//
// Defines `class VertexCM : public AbstractRenderer` with the reference's
// AlgorithmType enum (src/vertexcm.hxx:182-204) and constructor signature
// (:208-214), so that the reference's `smallvcm.cxx` and `config.hxx` compile
// UNCHANGED against it (they name only VertexCM, VertexCM::k{LightTrace,Ppm,
// Bpm,Bpt,Vcm} and the 5-argument constructor: src/config.hxx:124-138).
// All work is forwarded to the MI355X library through the C-ABI
// (include/smallvcm_amd.h); nothing of the reference's integrator is compiled.
//
// Two ways to use it:
//   (1) replace src/vertexcm.hxx in a SmallVCM checkout by this file (plus
//       flatten_scene.hxx next to it) and link libsmallvcm_amd.so;
//   (2) leave the checkout untouched and compile its smallvcm.cxx with
//       -D__VERTEXCM_HXX__ -include <this file>   (the reference's own header
//       is then skipped by its include guard) -- what dropin/Makefile does.
#ifndef SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
#define SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
#ifndef __VERTEXCM_HXX__
#define __VERTEXCM_HXX__   /* the reference's guard (src/vertexcm.hxx:25-26) */
#endif

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

class VertexCM : public AbstractRenderer
{
public:

    enum AlgorithmType   // src/vertexcm.hxx:182-204
    {
        kLightTrace = 0,
        kPpm,
        kBpm,
        kBpt,
        kVcm
    };

    VertexCM(
        const Scene&  aScene,
        AlgorithmType aAlgorithm,
        const float   aRadiusFactor,
        const float   aRadiusAlpha,
        int           aSeed = 1234
    ) :
        AbstractRenderer(aScene),
        mCtx(NULL)
    {
        vcm_scene_desc desc;
        const int rc = smallvcm_amd::FlattenScene(aScene, desc);
        if(rc != 0)
        {
            // same convention as the reference's factory (src/config.hxx:140-141)
            fprintf(stderr, "smallvcm_amd: scene cannot be flattened (code %d)\n", rc);
            exit(2);
        }

        mCtx = vcm_create(&desc, int(aAlgorithm), aRadiusFactor, aRadiusAlpha, aSeed);
        if(mCtx == NULL)
        {
            fprintf(stderr, "smallvcm_amd: %s\n", vcm_last_error());
            exit(2);
        }

        mResX = int(aScene.mCamera.mResolution.x);
        mResY = int(aScene.mCamera.mResolution.y);
    }

    virtual ~VertexCM()
    {
        vcm_destroy(mCtx);
    }

    // src/vertexcm.hxx:284-548.  mMaxPathLength / mMinPathLength are assigned
    // by the driver after construction (src/smallvcm.cxx:70-71), so they are
    // read here, per call.
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

        mIterations++;   // :547
    }

    vcm_ctx* Context() { return mCtx; }

private:

    vcm_ctx            *mCtx;
    int                 mResX, mResY;
    std::vector<float>  mHost;
};

#endif //SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
