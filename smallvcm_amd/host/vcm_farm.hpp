// vcm_farm.hpp -- the reference's render() (src/smallvcm.cxx:52-151) over the GPUs of ONE node, in C++, on top of the
// C-ABI (include/smallvcm_amd.h) and RCCL: one host thread per GPU ("rank"), the ranks cut into groups of `shards`.
//
//   * A renderer lives on a group: seed base + g, the iterations OpenMP's static schedule gives "thread" g
//     (smallvcm.cxx:98-108).  Its ranks split the paths of every iteration by index (vcm_create_sharded): light path p
//     and pixel p on the same rank, so vertex connection stays local (vertexcm.hxx:504-506); the merge needs ALL light
//     vertices (vertexcm.hxx:532-533), so per iteration the ranks of a group
//         1. exchange 7 numbers each -- vertex count and the bounding box of their own vertices -- which gives every
//            rank the box HashGrid::Build takes over the whole array (hashgrid.hxx:50-61) before the vertices have
//            arrived (host memory when the ranks are threads of one process, a 32-byte ncclAllGather otherwise),
//         2. ncclAllGather the 52-byte merge records on max-padded slabs, on the rank's communication stream, while the
//            camera pass (which needs only the local light vertices) runs on the renderer's stream,
//         3. build the identical hash grid and merge their own pixels.
//   * `inflight` renderers take turns on a group (default 2 when shards > 1): every call only enqueues, so one
//     renderer's exchange crosses xGMI behind the other's kernels.
//   * Read-out: every rank scales its framebuffer by 1 / (own iterations * used renderers) and ONE ncclAllReduce over
//     all ranks gives the image render() leaves in the framebuffer (smallvcm.cxx:116-142: mean over the used renderers
//     of their running sums / iterations; the shards of a renderer hold partial sums of it).
//   * shards == 1 is the reference's own scheme (replicas, no exchange at all).
//
// Collective order (what makes this safe on RCCL): a rank owns ONE communication stream and belongs to TWO
// communicators, its group's and the world's.  Every collective of a step is enqueued on that one stream in program
// order, and the program order is the same on every rank of a communicator: the small exchanges of all in-flight
// renderers first (slot 0, slot 1, ...), then their all-gathers (slot 0, slot 1, ...).  The world communicator is
// used only outside the iteration loop (barriers around the timed region, the framebuffer all-reduce), after the
// rank has drained its communication stream.  No two communicators ever have collectives in flight on one device at
// the same time, and no rank can enqueue two collectives in an order another rank does not.
//
// Hosts: farm_render() runs `localRanks` of the `ranks` as threads of the calling process.  All ranks in one process
// (vcm_render --gpus N, `python bench.py --gpus N`): the communicators come from one ncclGetUniqueId each, made here.
// One process per GPU (`python -m torch.distributed.run ... bench.py --gpus N`): world rank 0's process makes the ids
// (vcm_farm_unique_ids), the launcher ships them, every process passes them in `uniqueIds`.
//
// For tests on a single GPU, where RCCL refuses two ranks on one device, an in-process stand-in (device-to-device
// copies ordered by events, host barriers) sits behind the same interface, so the rank logic above runs unchanged
// with several ranks per device (single process only).
#ifndef SMALLVCM_AMD_VCM_FARM_HPP
#define SMALLVCM_AMD_VCM_FARM_HPP

#include <string>
#include <vector>

#include "smallvcm_amd.h"
#include "smallvcm_amd_farm.h"

struct FarmConfig {
    vcm_scene_desc scene;
    int algorithm;
    float radiusFactor, radiusAlpha;
    int baseSeed;
    unsigned minLen, maxLen;
    int iterations;            // timed iterations of the whole farm
    int ranks;                 // world size: one rank per GPU
    int firstRank, localRanks; // this process hosts ranks [firstRank, firstRank + localRanks) as threads
    std::vector<int> devices;  // HIP device of each LOCAL rank
    int shards;                // ranks that share one iteration
    int inflight;              // renderers taking turns on a group
    bool rccl;                 // false: in-process stand-in (several ranks per device allowed; single process only)
    int warmup;                // untimed iterations of every renderer (framebuffer cleared afterwards)
    // false: render()'s schedule -- renderer g gets the block of `iterations` OpenMP's static schedule gives thread g,
    //        warm-up = iteration indices 0..warmup-1.
    // true:  benchmark schedule -- every renderer runs iterations / renderers iterations with the SAME indices
    //        warmup .. warmup + n - 1 (the radius window bench.py times at one GPU), warm-up = indices 0..warmup-1.
    bool sameWindow;
    std::vector<char> uniqueIds;   // empty: all ranks are local; else (1 + groups) ids of vcm_farm_unique_id_bytes() each
};

struct FarmResult {
    std::vector<float> image;  // W*H*3, the averaged framebuffer (valid where firstRank == 0)
    double wallSeconds;        // timed region, barrier to barrier, maximum over all ranks
    int renderers;
    int rcclRanks;             // ranks that took part in RCCL collectives (0 with the stand-in)
    std::vector<float> rankIterationMs;   // per world rank: mean device time of an iteration of its first renderer
    vcm_stats meanStats;       // first renderer, mean over its timed iterations: kernel times of world rank 0; work counters summed over the renderer's shards hosted by this process
    std::string error;         // empty on success
};

FarmResult farm_render(const FarmConfig &cfg);

#endif
