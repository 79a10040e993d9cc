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
// pathtracer.hxx and eyelight.hxx next to this file do the same for the
// reference's two other renderers.
#ifndef SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
#define SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
#ifndef __VERTEXCM_HXX__
#define __VERTEXCM_HXX__   /* the reference's guard (src/vertexcm.hxx:25-26) */
#endif

#include "gpu_renderer.hxx"

class VertexCM : public smallvcm_amd::GpuRenderer
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
        smallvcm_amd::GpuRenderer(aScene, int(aAlgorithm), aRadiusFactor, aRadiusAlpha, aSeed)
    {}
};

#endif //SMALLVCM_AMD_DROPIN_VERTEXCM_HXX
