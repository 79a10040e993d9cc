// eyelight.hxx -- drop-in replacement for SmallVCM's src/eyelight.hxx:
// `class EyeLight : public AbstractRenderer` with the reference's constructor
// (src/eyelight.hxx:39-44, created at src/config.hxx:118-119), running
// EyeLight::RunIteration (:46-77) on the MI355X (VCM_ALGO_EYE_LIGHT).
// Use like vertexcm.hxx: replace the file, or compile the untouched checkout
// with -D__EYELIGHT_HXX__ -include <this file>.
#ifndef SMALLVCM_AMD_DROPIN_EYELIGHT_HXX
#define SMALLVCM_AMD_DROPIN_EYELIGHT_HXX
#ifndef __EYELIGHT_HXX__
#define __EYELIGHT_HXX__   /* the reference's guard (src/eyelight.hxx:25-26) */
#endif

#include "gpu_renderer.hxx"

class EyeLight : public smallvcm_amd::GpuRenderer
{
public:

    EyeLight(
        const Scene& aScene,
        int aSeed = 1234
    ) :
        smallvcm_amd::GpuRenderer(aScene, VCM_ALGO_EYE_LIGHT, 0.f, 0.f, aSeed)
    {}
};

#endif //SMALLVCM_AMD_DROPIN_EYELIGHT_HXX
