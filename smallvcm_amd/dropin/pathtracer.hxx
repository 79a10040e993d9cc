// pathtracer.hxx -- drop-in replacement for SmallVCM's src/pathtracer.hxx:
// `class PathTracer : public AbstractRenderer` with the reference's constructor
// (src/pathtracer.hxx:37-43, created at src/config.hxx:120-121), running
// PathTracer::RunIteration (:45-215) on the MI355X (VCM_ALGO_PATH_TRACE).
// Use like vertexcm.hxx: replace the file, or compile the untouched checkout
// with -D__PATHTRACER_HXX__ -include <this file>.
#ifndef SMALLVCM_AMD_DROPIN_PATHTRACER_HXX
#define SMALLVCM_AMD_DROPIN_PATHTRACER_HXX
#ifndef __PATHTRACER_HXX__
#define __PATHTRACER_HXX__   /* the reference's guard (src/pathtracer.hxx:25-26) */
#endif

#include "gpu_renderer.hxx"

class PathTracer : public smallvcm_amd::GpuRenderer
{
public:

    PathTracer(
        const Scene& aScene,
        int aSeed = 1234
    ) :
        smallvcm_amd::GpuRenderer(aScene, VCM_ALGO_PATH_TRACE, 0.f, 0.f, aSeed)
    {}
};

#endif //SMALLVCM_AMD_DROPIN_PATHTRACER_HXX
