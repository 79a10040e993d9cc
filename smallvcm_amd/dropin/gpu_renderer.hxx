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

// Framebuffer keeps its pixels private and has no bulk setter (src/framebuffer.hxx:253-258); AbstractRenderer::GetFramebuffer
// is not virtual and copies the protected mFramebuffer (src/renderer.hxx:49-55).  So after EVERY RunIteration the host
// Framebuffer has to hold the running sum.  Round 2 did that through the public interface alone -- Clear() + one AddColor per
// pixel behind a 50 MB read into pageable memory: at 2048^2 that is 4.2 M calls and more host time than the GPU needs for
// the iteration itself.  The storage is a std::vector<Vec3f> of 12-byte texels in pixel order -- exactly the layout of the
// device framebuffer -- so the refresh below copies device -> mColor DIRECTLY, the vector's memory page-locked once
// (vcm_pin_host_memory).  Reaching the private member needs no edit of framebuffer.hxx: a pointer to member may be named in an
// EXPLICIT template instantiation whatever its access (C++ [temp.spec]: "the usual access checking rules do not apply
// to names used to specify explicit instantiations").  -DSMALLVCM_AMD_DROPIN_ADDCOLOR keeps the public-interface refresh.
#if !defined(SMALLVCM_AMD_DROPIN_ADDCOLOR)
template <typename Tag, typename Tag::type Member> struct MemberDoor { friend typename Tag::type member_of(Tag) { return Member; } };
struct FramebufferColors { typedef std::vector<Vec3f> Framebuffer::*type; friend type member_of(FramebufferColors); };
template struct MemberDoor<FramebufferColors, &Framebuffer::mColor>;
#endif

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
        mCtx(NULL), mPixels(NULL), mPixelBytes(0), mPinned(false)
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
#if !defined(SMALLVCM_AMD_DROPIN_ADDCOLOR)
        // AbstractRenderer's constructor has sized the pixels (renderer.hxx:41 -> framebuffer.hxx:62-69); nothing resizes them later
        static_assert(sizeof(Vec3f) == 3 * sizeof(float), "Vec3f is three packed floats (math.hxx:87-128)");
        std::vector<Vec3f> &pixels = mFramebuffer.*member_of(FramebufferColors());
        mPixels = pixels.empty() ? NULL : &pixels[0].x;
        mPixelBytes = pixels.size() * sizeof(Vec3f);
        mPinned = mPixels != NULL && pixels.size() == size_t(mResX) * size_t(mResY) && vcm_pin_host_memory(mPixels, mPixelBytes) == 0;
#endif
    }

    virtual ~GpuRenderer()
    {
#if !defined(SMALLVCM_AMD_DROPIN_ADDCOLOR)
        if(mPinned) vcm_unpin_host_memory(mPixels);
#endif
        vcm_destroy(mCtx);
    }

    static const char* RefreshKind()
    {
#if defined(SMALLVCM_AMD_DROPIN_ADDCOLOR)
        return "Clear + AddColor per pixel (public interface only)";
#else
        return "device -> Framebuffer::mColor, page-locked";
#endif
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

        // AbstractRenderer::GetFramebuffer (src/renderer.hxx:49-55) is not virtual and reads the protected host
        // mFramebuffer, which must hold the running SUM over iterations when it is called -- and nothing tells this
        // class when that is: the refresh happens here, after every iteration.
#if !defined(SMALLVCM_AMD_DROPIN_ADDCOLOR)
        if(mPixels != NULL && mPixelBytes == size_t(mResX) * size_t(mResY) * sizeof(Vec3f))
        {
            if(vcm_read_framebuffer(mCtx, mPixels) != 0)   // one DMA into the Framebuffer's own texels (12 bytes, pixel order)
            {
                fprintf(stderr, "smallvcm_amd: %s\n", vcm_last_error());
                exit(2);
            }
            mIterations++;   // vertexcm.hxx:547, pathtracer.hxx:216, eyelight.hxx:79
            return;
        }
#endif
        // public interface only: Framebuffer has no bulk setter (src/framebuffer.hxx:253-258): Clear + one AddColor
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
    float              *mPixels;      // the Framebuffer's own texels (NULL: public-interface refresh)
    size_t              mPixelBytes;
    bool                mPinned;
};

} // namespace smallvcm_amd

#endif //SMALLVCM_AMD_DROPIN_GPU_RENDERER_HXX
