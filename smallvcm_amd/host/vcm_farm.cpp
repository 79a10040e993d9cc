// vcm_farm.cpp -- see vcm_farm.hpp.  Host code only: everything on the device goes through the C-ABI
// (libsmallvcm_amd.so), the HIP runtime (streams, events, copies) and RCCL.
#include "vcm_farm.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstring>
#include <mutex>
#include <thread>

namespace {

// ---- a reusable barrier that can be broken (a failing rank must not leave the others waiting) ----
class Barrier {
public:
    explicit Barrier(int n) : mN(n), mCount(0), mGen(0), mBroken(false) {}
    bool wait()
    {
        std::unique_lock<std::mutex> lk(mM);
        if (mBroken) return false;
        const unsigned long long gen = mGen;
        if (++mCount == mN) { mCount = 0; mGen++; mCv.notify_all(); return true; }
        mCv.wait(lk, [&] { return mGen != gen || mBroken; });
        return !mBroken;
    }
    void breakAll() { std::lock_guard<std::mutex> g(mM); mBroken = true; mCv.notify_all(); }
private:
    std::mutex mM; std::condition_variable mCv; int mN, mCount; unsigned long long mGen; bool mBroken;
};

struct Shared {   // one per farm
    std::mutex m;
    std::string error;
    std::atomic<bool> failed;
    std::vector<Barrier *> barriers;
    Shared() : failed(false) {}
    void fail(const std::string &what)
    {
        { std::lock_guard<std::mutex> g(m); if (error.empty()) error = what; }
        failed = true;
        for (Barrier *b : barriers) b->breakAll();
    }
};

#define HIPOK(expr) do { hipError_t e_ = (expr); if (e_ != hipSuccess) { sh.fail(std::string(#expr) + ": " + hipGetErrorString(e_)); return false; } } while (0)
#define NCCLOK(expr) do { ncclResult_t r_ = (expr); if (r_ != ncclSuccess) { sh.fail(std::string(#expr) + ": " + ncclGetErrorString(r_)); return false; } } while (0)
#define VCMOK(expr) do { if ((expr) != 0) { sh.fail(std::string(#expr) + ": " + vcm_last_error()); return false; } } while (0)

// ---- collectives of one communicator (the ranks of a group, or all ranks) ----
class Collectives {
public:
    virtual ~Collectives() {}
    // every rank: `send` (floatsPerRank floats, device) -> slot `rank` of everybody's `recv` (size ranks * floatsPerRank)
    virtual bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t floatsPerRank, hipStream_t s) = 0;
    virtual bool allReduceSum(Shared &sh, int rank, float *buf, size_t n, hipStream_t s) = 0;
    // before `rank` overwrites a buffer it has handed to allGather as `send`: wait (on s) until nobody reads it any more
    virtual bool sendBufferFree(Shared &sh, int rank, hipStream_t s) = 0;
    // 7 numbers per rank, host side (the ranks are threads of this process); all: ranks * 7
    bool exchange7(Shared &sh, int rank, const double *mine, double *all)
    {
        for (int i = 0; i < 7; i++) mSmall[(size_t)rank * 7 + i] = mine[i];
        if (!mBar.wait()) return false;
        for (size_t i = 0; i < mSmall.size(); i++) all[i] = mSmall[i];
        if (!mBar.wait()) return false;   // nobody overwrites mSmall before everybody has read it
        (void)sh;
        return true;
    }
    int size() const { return mRanks; }
protected:
    Collectives(Shared &sh, int ranks) : mRanks(ranks), mBar(ranks), mSmall((size_t)ranks * 7, 0.0) { sh.barriers.push_back(&mBar); }
    int mRanks;
    Barrier mBar;
    std::vector<double> mSmall;
};

class RcclCollectives : public Collectives {
public:
    RcclCollectives(Shared &sh, const std::vector<int> &devices) : Collectives(sh, (int)devices.size()), mComms(devices.size())
    {
        const ncclResult_t r = ncclCommInitAll(mComms.data(), (int)devices.size(), devices.data());
        if (r != ncclSuccess) { sh.fail(std::string("ncclCommInitAll: ") + ncclGetErrorString(r)); mComms.clear(); }
    }
    ~RcclCollectives() { for (ncclComm_t c : mComms) ncclCommDestroy(c); }
    bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t n, hipStream_t s) override
    {
        if (mComms.empty()) return false;
        NCCLOK(ncclAllGather(send, recv, n, ncclFloat, mComms[(size_t)rank], s));
        return true;
    }
    bool allReduceSum(Shared &sh, int rank, float *buf, size_t n, hipStream_t s) override
    {
        if (mComms.empty()) return false;
        NCCLOK(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, mComms[(size_t)rank], s));
        return true;
    }
    bool sendBufferFree(Shared &, int, hipStream_t) override { return true; }   // RCCL reads `send` in stream order
private:
    std::vector<ncclComm_t> mComms;
};

// Stand-in for tests on one GPU (RCCL refuses two ranks on one device): the ranks are threads of this process, data
// moves with device-to-device copies ordered by events.  Same interface, same call pattern as the RCCL class.
class ThreadCollectives : public Collectives {
public:
    ThreadCollectives(Shared &sh, int ranks) : Collectives(sh, ranks), mSend((size_t)ranks, NULL), mReady((size_t)ranks),
                                               mDone((size_t)ranks), mHave((size_t)ranks, 0), mHost((size_t)ranks)
    {
        for (int r = 0; r < ranks; r++) { mReady[(size_t)r] = NULL; mDone[(size_t)r] = NULL; }
    }
    ~ThreadCollectives()
    {
        for (hipEvent_t e : mReady) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : mDone) if (e) (void)hipEventDestroy(e);
    }
    bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t n, hipStream_t s) override
    {
        if (!events(sh, rank)) return false;
        mSend[(size_t)rank] = send;
        HIPOK(hipEventRecord(mReady[(size_t)rank], s));
        if (!mBar.wait()) return false;
        for (int r = 0; r < mRanks; r++) {
            HIPOK(hipStreamWaitEvent(s, mReady[(size_t)r], 0));
            HIPOK(hipMemcpyAsync(recv + (size_t)r * n, mSend[(size_t)r], n * sizeof(float), hipMemcpyDeviceToDevice, s));
        }
        HIPOK(hipEventRecord(mDone[(size_t)rank], s));
        mHave[(size_t)rank] = 1;
        return mBar.wait();
    }
    bool sendBufferFree(Shared &sh, int rank, hipStream_t s) override
    {
        (void)rank;
        for (int r = 0; r < mRanks; r++) if (mHave[(size_t)r]) HIPOK(hipStreamWaitEvent(s, mDone[(size_t)r], 0));
        return true;
    }
    bool allReduceSum(Shared &sh, int rank, float *buf, size_t n, hipStream_t s) override
    {
        std::vector<float> &mine = mHost[(size_t)rank];
        mine.resize(n);
        HIPOK(hipMemcpyAsync(mine.data(), buf, n * sizeof(float), hipMemcpyDeviceToHost, s));
        HIPOK(hipStreamSynchronize(s));
        if (!mBar.wait()) return false;
        std::vector<float> sum(mHost[0]);
        for (int r = 1; r < mRanks; r++) for (size_t i = 0; i < n; i++) sum[i] = sum[i] + mHost[(size_t)r][i];   // rank order: same bits everywhere
        if (!mBar.wait()) return false;
        HIPOK(hipMemcpyAsync(buf, sum.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
        HIPOK(hipStreamSynchronize(s));
        return true;
    }
private:
    bool events(Shared &sh, int rank)
    {
        if (!mReady[(size_t)rank]) {
            HIPOK(hipEventCreateWithFlags(&mReady[(size_t)rank], hipEventDisableTiming));
            HIPOK(hipEventCreateWithFlags(&mDone[(size_t)rank], hipEventDisableTiming));
        }
        return true;
    }
    std::vector<const float *> mSend;
    std::vector<hipEvent_t> mReady, mDone;
    std::vector<char> mHave;   // one byte per rank: the rank threads write their own element concurrently
    std::vector<std::vector<float>> mHost;
};

// the iterations OpenMP's default schedule(static) gives thread `tid` of `threads` (smallvcm.cxx:98-108)
void static_schedule(int iterations, int threads, int tid, int *first, int *count)
{
    const int q = iterations / threads, r = iterations % threads;
    *first = tid * q + std::min(tid, r);
    *count = q + (tid < r ? 1 : 0);
}

// one renderer as seen by ONE of its ranks
struct Slot {
    vcm_ctx *ctx;
    Collectives *group;          // NULL when shards == 1
    hipStream_t stream, commStream;
    hipEvent_t evRecords, evGathered;
    float *local, *gathered;     // slabs: stride records / shards * stride records
    size_t capRecords;
    int first, count;            // iterations of this renderer
    std::vector<long long> counts;
    long long stride;
    bool exchanging;
    Slot() : ctx(NULL), group(NULL), stream(NULL), commStream(NULL), evRecords(NULL), evGathered(NULL), local(NULL), gathered(NULL),
             capRecords(0), first(0), count(0), stride(0), exchanging(false) {}
};

struct RankArgs {
    const FarmConfig *cfg;
    Shared *sh;
    int rank, device, group, shard;
    std::vector<Collectives *> groupComms;   // per slot
    Collectives *world;
    Barrier *startLine;
    FarmResult *result;
    double *wall;
};

bool slot_start(Shared &sh, const FarmConfig &cfg, Slot &sl, int shard, int iteration)
{   // light pass, start of the exchange, the part of the camera pass that does not need the other ranks' vertices
    VCMOK(vcm_begin_iteration(sl.ctx, iteration, cfg.minLen, cfg.maxLen));
    VCMOK(vcm_trace_light(sl.ctx));
    sl.exchanging = false;
    if (!sl.group) return true;
    const int S = sl.group->size();
    float mn[3], mx[3];
    long long n = 0;
    VCMOK(vcm_local_light_bbox(sl.ctx, mn, mx, &n));   // synchronises the stream: the one host wait of an iteration
    double mine[7] = { (double)n, mn[0], mn[1], mn[2], mx[0], mx[1], mx[2] }, all[7 * 64];
    if (S > 64) { sh.fail("more than 64 shards"); return false; }
    if (!sl.group->exchange7(sh, shard, mine, all)) return false;
    sl.counts.assign((size_t)S, 0);
    sl.stride = 1;
    float gmn[3] = { 1e36f, 1e36f, 1e36f }, gmx[3] = { -1e36f, -1e36f, -1e36f };   // hashgrid.hxx:47-48
    for (int r = 0; r < S; r++) {
        sl.counts[(size_t)r] = (long long)all[r * 7];
        sl.stride = std::max(sl.stride, sl.counts[(size_t)r]);
        if (sl.counts[(size_t)r] > 0)
            for (int k = 0; k < 3; k++) { gmn[k] = std::min(gmn[k], (float)all[r * 7 + 1 + k]); gmx[k] = std::max(gmx[k], (float)all[r * 7 + 4 + k]); }
    }
    VCMOK(vcm_set_grid_bbox(sl.ctx, gmn, gmx));
    if ((size_t)sl.stride > sl.capRecords) {   // grow the slabs (rare: the counts vary by a fraction of a percent)
        HIPOK(hipStreamSynchronize(sl.stream));
        HIPOK(hipStreamSynchronize(sl.commStream));
        if (sl.local) (void)hipFree(sl.local);
        if (sl.gathered) (void)hipFree(sl.gathered);
        sl.capRecords = (size_t)sl.stride + (size_t)sl.stride / 16 + 1024;
        HIPOK(hipMalloc((void **)&sl.local, sl.capRecords * VCM_MERGE_RECORD_FLOATS * sizeof(float)));
        HIPOK(hipMalloc((void **)&sl.gathered, sl.capRecords * (size_t)S * VCM_MERGE_RECORD_FLOATS * sizeof(float)));
    }
    if (!sl.group->sendBufferFree(sh, shard, sl.stream)) return false;
    VCMOK(vcm_export_light_records(sl.ctx, sl.local, n));
    // the all-gather runs on the second stream, behind the export and next to the camera pass
    HIPOK(hipEventRecord(sl.evRecords, sl.stream));
    HIPOK(hipStreamWaitEvent(sl.commStream, sl.evRecords, 0));
    if (!sl.group->allGather(sh, shard, sl.local, sl.gathered, (size_t)sl.stride * VCM_MERGE_RECORD_FLOATS, sl.commStream)) return false;
    HIPOK(hipEventRecord(sl.evGathered, sl.commStream));
    sl.exchanging = true;
    if (vcm_is_wavefront(sl.ctx, cfg.maxLen)) VCMOK(vcm_trace_camera(sl.ctx));   // needs only the local light vertices
    return true;
}

bool slot_finish(Shared &sh, const FarmConfig &cfg, Slot &sl)
{   // wait for the exchange, grid build, (camera pass,) merge, resolve
    if (sl.exchanging) {
        HIPOK(hipStreamWaitEvent(sl.stream, sl.evGathered, 0));
        VCMOK(vcm_import_light_records(sl.ctx, sl.gathered, sl.counts.data(), (int)sl.counts.size(), sl.stride));
    }
    VCMOK(vcm_build_grid(sl.ctx));
    if (!(sl.exchanging && vcm_is_wavefront(sl.ctx, cfg.maxLen))) VCMOK(vcm_trace_camera(sl.ctx));
    VCMOK(vcm_merge(sl.ctx));
    VCMOK(vcm_end_iteration(sl.ctx));
    return true;
}

bool run_steps(Shared &sh, const FarmConfig &cfg, std::vector<Slot> &slots, int shard, int offset, int steps)
{   // renderers advance in lock-step; a step = one iteration half of every in-flight renderer, then the other half
    for (int t = 0; t < steps; t++) {
        for (Slot &sl : slots) if (t < sl.count || offset < 0) { if (!slot_start(sh, cfg, sl, shard, offset < 0 ? t : sl.first + t)) return false; }
        for (Slot &sl : slots) if (t < sl.count || offset < 0) { if (!slot_finish(sh, cfg, sl)) return false; }
    }
    return true;
}

bool rank_main(RankArgs &a)
{
    const FarmConfig &cfg = *a.cfg;
    Shared &sh = *a.sh;
    HIPOK(hipSetDevice(a.device));
    const int groups = cfg.ranks / cfg.shards, R = groups * cfg.inflight;
    std::vector<Slot> slots((size_t)cfg.inflight);
    int maxCount = 0;
    for (int k = 0; k < cfg.inflight; k++) {
        Slot &sl = slots[(size_t)k];
        const int rid = a.group * cfg.inflight + k;
        sl.ctx = vcm_create_sharded(&cfg.scene, cfg.algorithm, cfg.radiusFactor, cfg.radiusAlpha, cfg.baseSeed + rid, a.device,
                                    a.shard, cfg.shards);   // seed: smallvcm.cxx:68
        if (!sl.ctx) { sh.fail(std::string("vcm_create_sharded: ") + vcm_last_error()); return false; }
        HIPOK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        HIPOK(hipStreamCreateWithFlags(&sl.commStream, hipStreamNonBlocking));
        HIPOK(hipEventCreateWithFlags(&sl.evRecords, hipEventDisableTiming));
        HIPOK(hipEventCreateWithFlags(&sl.evGathered, hipEventDisableTiming));
        VCMOK(vcm_set_stream(sl.ctx, sl.stream));
        VCMOK(vcm_reserve(sl.ctx, cfg.maxLen));
        sl.group = cfg.shards > 1 ? a.groupComms[(size_t)k] : NULL;
        static_schedule(cfg.iterations, R, rid, &sl.first, &sl.count);
        maxCount = std::max(maxCount, sl.count);
    }
    if (cfg.warmup > 0) {   // untimed: iterations 0..warmup-1 of every renderer, then the framebuffers start over
        if (!run_steps(sh, cfg, slots, a.shard, -1, cfg.warmup)) return false;
        for (Slot &sl : slots) VCMOK(vcm_clear_framebuffer(sl.ctx));
    }
    for (Slot &sl : slots) VCMOK(vcm_synchronize(sl.ctx));
    if (!a.startLine->wait()) return false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!run_steps(sh, cfg, slots, a.shard, 0, maxCount)) return false;
    for (Slot &sl : slots) { VCMOK(vcm_synchronize(sl.ctx)); HIPOK(hipStreamSynchronize(sl.commStream)); }
    if (!a.startLine->wait()) return false;
    if (a.rank == 0) *a.wall = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

    // read-out (smallvcm.cxx:116-142): mean over the used renderers of (running sum / own iterations); a renderer's
    // shards hold partial sums of it, so ONE all-reduce over all ranks does both sums
    int used = 0;
    for (int rid = 0; rid < R; rid++) { int f, c; static_schedule(cfg.iterations, R, rid, &f, &c); if (c > 0) used++; }
    const size_t n3 = (size_t)((int)cfg.scene.camera.resolution[0]) * (size_t)((int)cfg.scene.camera.resolution[1]) * 3;
    float *acc = NULL, *tmp = NULL;
    HIPOK(hipMalloc((void **)&acc, n3 * sizeof(float)));
    HIPOK(hipMalloc((void **)&tmp, n3 * sizeof(float)));
    HIPOK(hipMemsetAsync(acc, 0, n3 * sizeof(float), slots[0].stream));
    std::vector<float> hostAcc(n3, 0.f), hostTmp(n3);
    bool any = false;
    for (Slot &sl : slots) {
        if (sl.count == 0) continue;   // WasUsed(), renderer.hxx:58
        const float scale = 1.f / ((float)sl.count * (float)used);
        VCMOK(vcm_export_framebuffer_scaled(sl.ctx, any ? tmp : acc, scale));
        VCMOK(vcm_synchronize(sl.ctx));
        if (any) {   // second and further renderers of this rank: summed on the host (read-out, not the timed path)
            HIPOK(hipMemcpy(hostAcc.data(), acc, n3 * sizeof(float), hipMemcpyDeviceToHost));
            HIPOK(hipMemcpy(hostTmp.data(), tmp, n3 * sizeof(float), hipMemcpyDeviceToHost));
            for (size_t i = 0; i < n3; i++) hostAcc[i] = hostAcc[i] + hostTmp[i];
            HIPOK(hipMemcpy(acc, hostAcc.data(), n3 * sizeof(float), hipMemcpyHostToDevice));
        }
        any = true;
    }
    HIPOK(hipStreamSynchronize(slots[0].stream));
    if (a.world->size() > 1) {
        if (!a.world->allReduceSum(sh, a.rank, acc, n3, slots[0].stream)) return false;
        HIPOK(hipStreamSynchronize(slots[0].stream));
    }
    if (a.rank == 0) {
        a.result->image.resize(n3);
        HIPOK(hipMemcpy(a.result->image.data(), acc, n3 * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(acc); (void)hipFree(tmp);
    if (!a.startLine->wait()) return false;   // nobody tears a communicator's peer down while a collective runs
    for (Slot &sl : slots) {
        vcm_destroy(sl.ctx);
        if (sl.local) (void)hipFree(sl.local);
        if (sl.gathered) (void)hipFree(sl.gathered);
        (void)hipEventDestroy(sl.evRecords); (void)hipEventDestroy(sl.evGathered);
        (void)hipStreamDestroy(sl.commStream); (void)hipStreamDestroy(sl.stream);
    }
    return true;
}

} // namespace

FarmResult farm_render(const FarmConfig &cfg)
{
    FarmResult res;
    res.wallSeconds = 0;
    res.renderers = 0;
    if (cfg.ranks < 1 || cfg.shards < 1 || cfg.ranks % cfg.shards || cfg.inflight < 1 || (int)cfg.devices.size() != cfg.ranks) {
        res.error = "ranks must be a multiple of shards, one device per rank, inflight >= 1";
        return res;
    }
    Shared sh;
    const int groups = cfg.ranks / cfg.shards;
    res.renderers = groups * cfg.inflight;
    Barrier startLine(cfg.ranks);
    sh.barriers.push_back(&startLine);
    // communicators: one per (group, in-flight slot) -- collectives of one communicator execute in order, and the
    // small count exchange of one renderer must not queue behind the large all-gather of the other -- plus the world
    std::vector<std::vector<Collectives *>> groupComms((size_t)groups);
    for (int g = 0; g < groups && cfg.shards > 1; g++) {
        std::vector<int> devs(cfg.devices.begin() + g * cfg.shards, cfg.devices.begin() + (g + 1) * cfg.shards);
        for (int k = 0; k < cfg.inflight; k++)
            groupComms[(size_t)g].push_back(cfg.rccl ? (Collectives *)new RcclCollectives(sh, devs) : (Collectives *)new ThreadCollectives(sh, cfg.shards));
    }
    Collectives *world = cfg.rccl ? (Collectives *)new RcclCollectives(sh, cfg.devices) : (Collectives *)new ThreadCollectives(sh, cfg.ranks);
    double wall = 0;
    std::vector<RankArgs> args((size_t)cfg.ranks);
    std::vector<std::thread> threads;
    if (!sh.failed) {
        for (int r = 0; r < cfg.ranks; r++) {
            RankArgs &a = args[(size_t)r];
            a.cfg = &cfg; a.sh = &sh; a.rank = r; a.device = cfg.devices[(size_t)r]; a.group = r / cfg.shards; a.shard = r % cfg.shards;
            a.groupComms = groupComms[(size_t)a.group]; a.world = world; a.startLine = &startLine; a.result = &res; a.wall = &wall;
            threads.emplace_back([&a, &sh] { if (!rank_main(a)) sh.fail("rank failed"); });
        }
        for (std::thread &t : threads) t.join();
    }
    for (auto &v : groupComms) for (Collectives *c : v) delete c;
    delete world;
    res.wallSeconds = wall;
    res.error = sh.error;
    return res;
}
