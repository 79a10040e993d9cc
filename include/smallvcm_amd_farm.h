/* smallvcm_amd_farm.h -- C-ABI of the multi-GPU host (smallvcm_amd/host/libsmallvcm_amd_farm.so).
 *
 * Replaces the reference's render() (src/smallvcm.cxx:52-151: one renderer per host thread, seeds mBaseSeed + i
 * :61-72, iterations dealt out by OpenMP's static schedule :99-108, mean of the used renderers' framebuffers
 * :116-142) for the GPUs of one node.  C++ over include/smallvcm_amd.h, the HIP runtime and RCCL only
 * (smallvcm_amd/host/vcm_farm.hpp has the decomposition); plain C types here so that any host can bind it
 * (bench.py does, through ctypes; vcm_render links the same code).
 */
#ifndef SMALLVCM_AMD_FARM_H
#define SMALLVCM_AMD_FARM_H

#include "smallvcm_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

#define VCM_FARM_MAX_RANKS 64

typedef struct vcm_farm_config {
    vcm_scene_desc scene;
    int   algorithm;                 /* VCM_ALGO_* */
    float radiusFactor, radiusAlpha; /* config.hxx:239-240 */
    int   baseSeed;                  /* config.hxx:234; renderer g gets baseSeed + g (smallvcm.cxx:68) */
    unsigned minLen, maxLen;         /* smallvcm.cxx:70-71 */
    int   iterations;                /* timed iterations of the whole farm (config.hxx:233) */
    int   warmup;                    /* untimed iterations of every renderer before them */
    int   sameWindow;                /* 0: render()'s static schedule; 1: every renderer runs the indices warmup .. (benchmark) */
    int   ranks;                     /* world size, one rank per GPU */
    int   firstRank, localRanks;     /* the ranks this process hosts (threads): [firstRank, firstRank + localRanks) */
    int   devices[VCM_FARM_MAX_RANKS]; /* HIP device of each local rank */
    int   shards;                    /* ranks that share one iteration (path-index shards, all-gather of the light vertices) */
    int   inflight;                  /* renderers taking turns on a group of `shards` ranks */
    int   collectives;               /* 0: RCCL; 1: in-process stand-in (tests on one GPU; single process only) */
    const void *uniqueIds;           /* NULL when localRanks == ranks; else (1 + ranks / shards) ids from vcm_farm_unique_ids */
    int   nUniqueIds;
} vcm_farm_config;

typedef struct vcm_farm_result {
    double wallSeconds;              /* timed region, barrier to barrier, maximum over all ranks */
    int    renderers;                /* (ranks / shards) * inflight */
    int    rcclRanks;                /* ranks that took part in RCCL collectives (0 with the stand-in) */
    float  rankIterationMs[VCM_FARM_MAX_RANKS]; /* per world rank: mean device time of one iteration of its first renderer */
    vcm_stats meanStats;             /* first renderer, mean over its timed iterations: times = world rank 0's; work counters = the sum over its shards in this process */
} vcm_farm_result;

/* Runs the farm; blocks until every local rank is done.  imageOut (W*H*3 floats, may be NULL) receives the averaged
 * framebuffer in the process that hosts world rank 0.  Returns 0, or -1 with vcm_farm_last_error(). */
int vcm_farm_render(const vcm_farm_config *cfg, vcm_farm_result *out, float *imageOut);
const char *vcm_farm_last_error(void);

/* One process per GPU: world rank 0's process calls vcm_farm_unique_ids(buf, 1 + ranks / shards) -- the world
 * communicator's id and one per group -- and the launcher hands the bytes to every process (vcm_farm_config.uniqueIds). */
int vcm_farm_unique_id_bytes(void);
int vcm_farm_unique_ids(void *out, int n);

/* sizes of the two PODs above, for hosts that mirror them (smallvcm_amd/farm.py; tests/test_abi.py) */
unsigned vcm_farm_sizeof_config(void);
unsigned vcm_farm_sizeof_result(void);

#ifdef __cplusplus
}
#endif
#endif /* SMALLVCM_AMD_FARM_H */
