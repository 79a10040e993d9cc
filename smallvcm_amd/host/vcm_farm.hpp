// vcm_farm.hpp -- the reference's render() (src/smallvcm.cxx:52-151) over the GPUs of ONE node, in C++, on top of the
// C-ABI (include/smallvcm_amd.h) and RCCL: one host thread per GPU ("rank"), the ranks cut into groups of `shards`.
//
//   * A renderer lives on a group: seed base + g, the iterations OpenMP's static schedule gives "thread" g
//     (smallvcm.cxx:98-108).  Its ranks split the paths of every iteration by index (vcm_create_sharded): light path p
//     and pixel p on the same rank, so vertex connection stays local (vertexcm.hxx:504-506); the merge needs ALL light
//     vertices (vertexcm.hxx:532-533), so per iteration the ranks of a group
//         1. exchange 7 numbers each -- vertex count and the bounding box of their own vertices (host side: the ranks
//            are threads of one process), which gives every rank the box HashGrid::Build takes over the whole array
//            (hashgrid.hxx:50-61) before the vertices have arrived,
//         2. ncclAllGather the 52-byte merge records on max-padded slabs, on a second stream, while the camera pass
//            (which needs only the local light vertices) runs on the first,
//         3. build the identical hash grid and merge their own pixels.
//   * `inflight` renderers take turns on a group (default 2 when shards > 1): every call only enqueues, so one
//     renderer's exchange crosses xGMI behind the other's kernels.  Each (group, slot) has its own communicator.
//   * Read-out: every rank scales its framebuffer by 1 / (own iterations * used renderers) and ONE ncclAllReduce over
//     all ranks gives the image render() leaves in the framebuffer (smallvcm.cxx:116-142: mean over the used renderers
//     of their running sums / iterations; the shards of a renderer hold partial sums of it).
//   * shards == 1 is the reference's own scheme (replicas, no exchange at all).
//
// Collectives: RCCL (ncclCommInitAll, one process) -- or, for tests on a single GPU, where RCCL refuses two ranks on
// one device, an in-process stand-in (device-to-device copies ordered by events, host barriers) behind the same
// interface, so the rank logic above runs unchanged with several ranks per device.
#ifndef SMALLVCM_AMD_VCM_FARM_HPP
#define SMALLVCM_AMD_VCM_FARM_HPP

#include <string>
#include <vector>

#include "smallvcm_amd.h"

struct FarmConfig {
    vcm_scene_desc scene;
    int algorithm;
    float radiusFactor, radiusAlpha;
    int baseSeed;
    unsigned minLen, maxLen;
    int iterations;
    int ranks;                 // host threads = ranks; rank r runs on devices[r]
    std::vector<int> devices;
    int shards;                // ranks that share one iteration
    int inflight;              // renderers taking turns on a group
    bool rccl;                 // false: in-process stand-in (several ranks per device allowed)
    int warmup;                // untimed iterations (indices 0..warmup-1 of every renderer, framebuffer cleared afterwards)
};

struct FarmResult {
    std::vector<float> image;  // W*H*3, the averaged framebuffer (valid on return)
    double wallSeconds;        // timed region: the iterations, barrier to barrier
    int renderers;
    std::string error;         // empty on success
};

FarmResult farm_render(const FarmConfig &cfg);

#endif
