// vcm_farm.cpp -- see vcm_farm.hpp.  Host code only: everything on the device goes through the C-ABI
// (libsmallvcm_amd.so), the HIP runtime (streams, events, copies) and RCCL.
#include "vcm_farm.hpp"

#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
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

struct Shared {   // one per farm_render call
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

// MEASUREMENT MODE for boxes with ONE GPU (SMALLVCM_AMD_FARM_SERIALIZE=1, stand-in collectives only): the rank threads take
// turns on the device -- a rank enqueues a phase, waits until the device has finished it and only then lets the next rank
// enqueue -- so that the kernels of one rank never share the chip with another rank's.  A kernel trace of such a run holds
// the kernel times ONE rank of an S-GPU node would see (its shard of the paths, the merge over everybody's vertices); with
// the ranks running concurrently on one chip they would be each other's contention.  profiles/tools/scaling_model.py reads it.
struct GpuTurn {
    static std::mutex &token() { static std::mutex m; return m; }
    static bool enabled() { static const bool on = [] { const char *e = getenv("SMALLVCM_AMD_FARM_SERIALIZE"); return e && e[0] == '1'; }(); return on; }
    explicit GpuTurn(hipStream_t a, hipStream_t b) : mA(a), mB(b), mHeld(enabled()) { if (mHeld) token().lock(); }
    ~GpuTurn() { if (mHeld) { (void)hipStreamSynchronize(mA); if (mB) (void)hipStreamSynchronize(mB); (void)hipDeviceSynchronize(); token().unlock(); } }
    hipStream_t mA, mB; bool mHeld;
};

// How the slabs of light vertices travel.  SMALLVCM_AMD_FARM_EXCHANGE =
//   allgather  (default) ONE ncclAllGather per iteration: RCCL's ring / tree schedules over the xGMI links
//   direct     every rank SENDS its slab to each of its S - 1 peers and receives theirs, all 2 (S - 1) transfers in one RCCL
//              group: point-to-point over the seven links of a GPU at once instead of a ring's neighbour hops
// Same bytes in the same places either way; which is faster is a question for an 8-GPU node (bench.py --gpus N --selftest
// runs both for correctness wherever two GPUs exist).
static bool exchange_is_direct()
{
    static const bool on = [] { const char *e = getenv("SMALLVCM_AMD_FARM_EXCHANGE"); return e && !strcmp(e, "direct"); }();
    return on;
}

// what the ranks of a group tell each other before the vertices travel: 7 numbers (hashgrid.hxx:47-61)
struct Xchg { long long n; float mn[3], mx[3]; int hist[256]; };   // hist: the second exchange of the space-sharded merge (else unused)
#define XCHG_WORDS (8 + 256)

// How a sharded renderer MERGES.  SMALLVCM_AMD_FARM_MERGE =
//   index  (default) north_star's decomposition: every rank merges its pixels' camera vertices against ALL ranks' light vertices
//   space  round 6 prototype: ranks own slabs of grid cells; light vertices go to the owners of their cell (+ one cell of
//          halo), camera vertices to the owner of their base cell, the 16-byte merge terms back to the pixels' owner -- the grid build and
//          the merge shrink with the shard, three all-to-alls replace the all-gather.  Wavefront mode, merging algorithms, one renderer in
//          flight per group.
static bool merge_by_space()
{
    static const bool on = [] { const char *e = getenv("SMALLVCM_AMD_FARM_MERGE"); return e && !strcmp(e, "space"); }();
    return on;
}

// ---- collectives of one communicator (the ranks of a group, or all ranks); `rank` = rank INSIDE the communicator ----
class Collectives {
public:
    virtual ~Collectives() {}
    // every rank: `send` (floatsPerRank floats, device) -> slot `rank` of everybody's `recv` (size ranks * floatsPerRank)
    virtual bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t floatsPerRank, hipStream_t s) = 0;
    virtual bool allReduceSum(Shared &sh, int rank, float *buf, size_t n, hipStream_t s) = 0;
    // every rank: sendCounts[d] elements of `elemFloats` floats for rank d at send + d * strideElems * elemFloats -> what rank r sent to
    // `rank` lands at recv + r * strideElems * elemFloats, recvCounts[r] = its number of elements (host values, known when the call returns)
    virtual bool allToAllV(Shared &sh, int rank, const float *send, const long long *sendCounts, long long strideElems, int elemFloats,
                           float *recv, long long *recvCounts, hipStream_t s) = 0;
    // before `rank` overwrites a buffer it has handed to allGather as `send`: wait (on s) until nobody reads it any more
    virtual bool sendBufferFree(Shared &sh, int rank, hipStream_t s) = 0;
    // all[r] = what rank r passed as `mine`; returns when every rank's numbers are known to the caller (host side)
    virtual bool exchange(Shared &sh, int rank, const Xchg &mine, Xchg *all, hipStream_t s)
    {   // the members are threads of this process: host memory and two barrier crossings
        (void)s; (void)sh;
        mSmall[(size_t)rank] = mine;
        if (!mBar.wait()) return false;
        for (int r = 0; r < mRanks; r++) all[r] = mSmall[(size_t)r];
        return mBar.wait();   // nobody overwrites mSmall before everybody has read it
    }
    // every member has arrived (and, across processes, its stream s has drained up to here)
    virtual bool barrier(Shared &sh, int rank, hipStream_t s) { (void)sh; (void)rank; (void)s; return mBar.wait(); }
    virtual int rcclRanks() const { return 0; }
    int size() const { return mRanks; }
protected:
    Collectives(Shared &sh, int ranks, int localMembers) : mRanks(ranks), mBar(localMembers), mSmall((size_t)ranks)
    {
        sh.barriers.push_back(&mBar);
    }
    int mRanks;
    Barrier mBar;               // over the members hosted by this process
    std::vector<Xchg> mSmall;
};

// RCCL.  One ncclComm_t per member hosted here (ncclCommInitRank inside one group call); members [first, first + n)
// of the communicator are local.  All collectives of a member are enqueued by its own host thread on the stream the
// caller passes -- the rank's single communication stream (vcm_farm.hpp: "Collective order").
class RcclCollectives : public Collectives {
public:
    RcclCollectives(Shared &sh, int ranks, int firstLocal, const std::vector<int> &localDevices, const ncclUniqueId *id)
        : Collectives(sh, ranks, (int)localDevices.size()), mFirst(firstLocal), mAllLocal((int)localDevices.size() == ranks),
          mComms(localDevices.size(), (ncclComm_t)NULL), mDevices(localDevices), mScratch(localDevices.size(), (uint32_t *)NULL),
          mPinned(localDevices.size(), (uint32_t *)NULL)
    {
        ncclUniqueId own;
        if (!id) {
            if (!mAllLocal) { sh.fail("RcclCollectives: members in other processes need a shared ncclUniqueId"); mComms.clear(); return; }
            const ncclResult_t r = ncclGetUniqueId(&own);
            if (r != ncclSuccess) { sh.fail(std::string("ncclGetUniqueId: ") + ncclGetErrorString(r)); mComms.clear(); return; }
            id = &own;
        }
        ncclResult_t r = ncclGroupStart();
        for (size_t i = 0; i < localDevices.size() && r == ncclSuccess; i++) {
            if (hipSetDevice(localDevices[i]) != hipSuccess) { r = ncclUnhandledCudaError; break; }
            r = ncclCommInitRank(&mComms[i], ranks, *id, firstLocal + (int)i);
        }
        const ncclResult_t e = ncclGroupEnd();
        if (r == ncclSuccess) r = e;
        if (r != ncclSuccess) { sh.fail(std::string("ncclCommInitRank: ") + ncclGetErrorString(r)); mComms.clear(); return; }
        { const char *f = getenv("SMALLVCM_AMD_FARM_RCCL_EXCHANGE"); if (f && f[0] == '1') mAllLocal = false; }   // tests: the cross-process exchange with one process
        if (!mAllLocal) {
            // A second communicator over the same ranks for the 32-byte exchanges and barriers, with a stream of its own
            // per member: a rank's communication stream may still hold another in-flight renderer's all-gather (58 - 464 MB)
            // when the next step's 7 numbers are due, and the host WAITS for those.  Collectives of one communicator run in
            // issue order whatever stream they are given; two communicators progress independently.  Every rank issues the
            // small ones in the same order among themselves (pass 2 of a step) and the large ones likewise (pass 3).
            mSmallComms.assign(localDevices.size(), (ncclComm_t)NULL);
            mSmallStreams.assign(localDevices.size(), (hipStream_t)NULL);
            r = ncclGroupStart();
            for (size_t i = 0; i < localDevices.size() && r == ncclSuccess; i++)
                r = ncclCommSplit(mComms[i], 0, firstLocal + (int)i, &mSmallComms[i], NULL);
            const ncclResult_t e2 = ncclGroupEnd();
            if (r == ncclSuccess) r = e2;
            if (r != ncclSuccess) {   // not fatal: the small collectives then share the main communicator (round 3's behaviour)
                fprintf(stderr, "vcm_farm: ncclCommSplit: %s -- the exchanges share the main communicator\n", ncclGetErrorString(r));
                mSmallComms.clear();
            } else {
                for (size_t i = 0; i < localDevices.size(); i++) {
                    if (hipSetDevice(localDevices[i]) != hipSuccess || hipStreamCreateWithFlags(&mSmallStreams[i], hipStreamNonBlocking) != hipSuccess) {
                        sh.fail("RcclCollectives: cannot create the exchange stream"); mComms.clear(); return;
                    }
                }
            }
        }
    }
    ~RcclCollectives()
    {
        for (size_t i = 0; i < mSmallComms.size(); i++) if (mSmallComms[i]) ncclCommDestroy(mSmallComms[i]);
        for (size_t i = 0; i < mSmallStreams.size(); i++) if (mSmallStreams[i]) { (void)hipSetDevice(mDevices[i]); (void)hipStreamDestroy(mSmallStreams[i]); }
        for (size_t i = 0; i < mComms.size(); i++) if (mComms[i]) ncclCommDestroy(mComms[i]);
        for (size_t i = 0; i < mScratch.size(); i++) {
            if (mScratch[i]) { (void)hipSetDevice(mDevices[i]); (void)hipFree(mScratch[i]); }
            if (mPinned[i]) (void)hipHostFree(mPinned[i]);
        }
    }
    bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t n, hipStream_t s) override
    {
        if (mComms.empty()) return false;
        ncclComm_t comm = mComms[(size_t)(rank - mFirst)];
        if (exchange_is_direct() && mRanks > 1) {
            /* own slab in place; then every peer at once: RCCL runs the sends and receives of a group concurrently */
            if (recv + (size_t)rank * n != send) HIPOK(hipMemcpyAsync(recv + (size_t)rank * n, send, n * sizeof(float), hipMemcpyDeviceToDevice, s));
            NCCLOK(ncclGroupStart());
            for (int d = 1; d < mRanks; d++) {   /* staggered: rank r starts with r + 1, so no peer is everybody's first target */
                const int to = (rank + d) % mRanks, from = (rank - d + mRanks) % mRanks;
                NCCLOK(ncclSend(send, n, ncclFloat, to, comm, s));
                NCCLOK(ncclRecv(recv + (size_t)from * n, n, ncclFloat, from, comm, s));
            }
            NCCLOK(ncclGroupEnd());
            return true;
        }
        NCCLOK(ncclAllGather(send, recv, n, ncclFloat, comm, s));
        return true;
    }
    bool allReduceSum(Shared &sh, int rank, float *buf, size_t n, hipStream_t s) override
    {
        if (mComms.empty()) return false;
        NCCLOK(ncclAllReduce(buf, buf, n, ncclFloat, ncclSum, mComms[(size_t)(rank - mFirst)], s));
        return true;
    }
    bool allToAllV(Shared &sh, int rank, const float *send, const long long *sendCounts, long long strideElems, int elemFloats,
                   float *recv, long long *recvCounts, hipStream_t s) override
    {
        if (mComms.empty() || !scratch(sh, rank) || mRanks > 128) return false;
        const size_t i = (size_t)(rank - mFirst);
        // the counts first: every rank's row of the matrix (2 words per entry) through the small communicator, host-synchronous
        uint32_t *host = mPinned[i], *dev = mScratch[i];
        if (2 * mRanks > XCHG_WORDS) { sh.fail("allToAllV: more ranks than the count exchange holds"); return false; }
        for (int d = 0; d < mRanks; d++) { host[2 * d] = (uint32_t)((unsigned long long)sendCounts[d] & 0xffffffffu); host[2 * d + 1] = (uint32_t)((unsigned long long)sendCounts[d] >> 32); }
        ncclComm_t small = mComms[i]; hipStream_t ss = s;
        if (!mSmallComms.empty()) { small = mSmallComms[i]; ss = mSmallStreams[i]; HIPOK(hipStreamSynchronize(s)); }
        HIPOK(hipMemcpyAsync(dev, host, 8 * (size_t)mRanks, hipMemcpyHostToDevice, ss));
        NCCLOK(ncclAllGather(dev, dev + XCHG_WORDS, 2 * (size_t)mRanks, ncclUint32, small, ss));
        HIPOK(hipMemcpyAsync(host + XCHG_WORDS, dev + XCHG_WORDS, 8 * (size_t)mRanks * (size_t)mRanks, hipMemcpyDeviceToHost, ss));
        HIPOK(hipStreamSynchronize(ss));
        for (int r = 0; r < mRanks; r++) {
            const uint32_t *w = host + XCHG_WORDS + 2 * ((size_t)r * (size_t)mRanks + (size_t)rank);
            recvCounts[r] = (long long)((unsigned long long)w[0] | ((unsigned long long)w[1] << 32));
        }
        const size_t ef = (size_t)elemFloats, st = (size_t)strideElems * ef;
        if (recvCounts[rank] > 0) HIPOK(hipMemcpyAsync(recv + (size_t)rank * st, send + (size_t)rank * st, (size_t)recvCounts[rank] * ef * sizeof(float), hipMemcpyDeviceToDevice, s));
        NCCLOK(ncclGroupStart());
        for (int d = 1; d < mRanks; d++) {
            const int to = (rank + d) % mRanks, from = (rank - d + mRanks) % mRanks;
            if (sendCounts[to] > 0) NCCLOK(ncclSend(send + (size_t)to * st, (size_t)sendCounts[to] * ef, ncclFloat, to, mComms[i], s));
            if (recvCounts[from] > 0) NCCLOK(ncclRecv(recv + (size_t)from * st, (size_t)recvCounts[from] * ef, ncclFloat, from, mComms[i], s));
        }
        NCCLOK(ncclGroupEnd());
        return true;
    }
    bool sendBufferFree(Shared &, int, hipStream_t) override { return true; }   // RCCL reads `send` in stream order
    bool exchange(Shared &sh, int rank, const Xchg &mine, Xchg *all, hipStream_t s) override
    {
        if (mAllLocal) return Collectives::exchange(sh, rank, mine, all, s);
        // members in other processes: 8 words per rank through a tiny ncclAllGather (bit patterns, no conversion)
        if (mComms.empty() || !scratch(sh, rank)) return false;
        const size_t i = (size_t)(rank - mFirst);
        uint32_t *host = mPinned[i], *dev = mScratch[i];
        host[0] = (uint32_t)((unsigned long long)mine.n & 0xffffffffu); host[7] = (uint32_t)((unsigned long long)mine.n >> 32);
        memcpy(host + 1, mine.mn, 12); memcpy(host + 4, mine.mx, 12); memcpy(host + 8, mine.hist, 1024);
        ncclComm_t comm = mComms[i];
        if (!mSmallComms.empty()) { comm = mSmallComms[i]; s = mSmallStreams[i]; }   // never behind an all-gather of records
        HIPOK(hipMemcpyAsync(dev, host, 4 * XCHG_WORDS, hipMemcpyHostToDevice, s));
        NCCLOK(ncclAllGather(dev, dev + XCHG_WORDS, XCHG_WORDS, ncclUint32, comm, s));
        HIPOK(hipMemcpyAsync(host + XCHG_WORDS, dev + XCHG_WORDS, 4 * XCHG_WORDS * (size_t)mRanks, hipMemcpyDeviceToHost, s));
        HIPOK(hipStreamSynchronize(s));
        for (int r = 0; r < mRanks; r++) {
            const uint32_t *w = host + XCHG_WORDS + XCHG_WORDS * (size_t)r;
            all[r].n = (long long)((unsigned long long)w[0] | ((unsigned long long)w[7] << 32));
            memcpy(all[r].mn, w + 1, 12); memcpy(all[r].mx, w + 4, 12); memcpy(all[r].hist, w + 8, 1024);
        }
        return true;
    }
    bool barrier(Shared &sh, int rank, hipStream_t s) override
    {
        if (mAllLocal) return mBar.wait();
        if (mComms.empty() || !scratch(sh, rank)) return false;
        const size_t i = (size_t)(rank - mFirst);
        HIPOK(hipStreamSynchronize(s));   // "its stream s has drained up to here"
        ncclComm_t comm = mComms[i];
        if (!mSmallComms.empty()) { comm = mSmallComms[i]; s = mSmallStreams[i]; }
        NCCLOK(ncclAllReduce(mScratch[i], mScratch[i], 1, ncclUint32, ncclSum, comm, s));
        HIPOK(hipStreamSynchronize(s));
        return true;
    }
    int rcclRanks() const override { return mRanks; }
private:
    bool scratch(Shared &sh, int rank)
    {
        const size_t i = (size_t)(rank - mFirst);
        if (!mScratch[i]) {   // the calling rank thread has its device current
            HIPOK(hipMalloc((void **)&mScratch[i], 4 * XCHG_WORDS * (size_t)(mRanks + 1)));
            HIPOK(hipMemset(mScratch[i], 0, 4 * XCHG_WORDS * (size_t)(mRanks + 1)));
            HIPOK(hipHostMalloc((void **)&mPinned[i], 4 * XCHG_WORDS * (size_t)(mRanks + 1), hipHostMallocDefault));
        }
        return true;
    }
    int mFirst;
    bool mAllLocal;
    std::vector<ncclComm_t> mComms;
    std::vector<ncclComm_t> mSmallComms;      // empty: the small collectives use mComms (all members local, or the split failed)
    std::vector<hipStream_t> mSmallStreams;
    std::vector<int> mDevices;
    std::vector<uint32_t *> mScratch, mPinned;
};

// Stand-in for tests on one GPU (RCCL refuses two ranks on one device): the ranks are threads of this process, data
// moves with device-to-device copies ordered by events.  Same interface, same call pattern as the RCCL class.
class ThreadCollectives : public Collectives {
public:
    ThreadCollectives(Shared &sh, int ranks) : Collectives(sh, ranks, ranks), mSend((size_t)ranks, NULL), mRecv((size_t)ranks, NULL), mCounts((size_t)ranks), mFree((size_t)ranks, (hipEvent_t)NULL), mReady((size_t)ranks),
                                               mDone((size_t)ranks), mHave((size_t)ranks, 0), mHost((size_t)ranks)
    {
        for (int r = 0; r < ranks; r++) { mReady[(size_t)r] = NULL; mDone[(size_t)r] = NULL; }
    }
    ~ThreadCollectives()
    {
        for (hipEvent_t e : mReady) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : mDone) if (e) (void)hipEventDestroy(e);
        for (hipEvent_t e : mFree) if (e) (void)hipEventDestroy(e);
    }
    bool allGather(Shared &sh, int rank, const float *send, float *recv, size_t n, hipStream_t s) override
    {
        if (!events(sh, rank)) return false;
        if (exchange_is_direct()) {
            /* the direct exchange as the stand-in can show it: every rank WRITES its slab into each peer's receive buffer on
               its own stream (the peers published where), the receivers wait for all writers' events */
            mRecv[(size_t)rank] = recv;
            HIPOK(hipEventRecord(mFree[(size_t)rank], s));   // my stream is here: whatever read my receive buffer last has been ordered before it
            if (!mBar.wait()) return false;   // every receive buffer is known
            for (int d = 0; d < mRanks; d++) {
                const int to = (rank + d) % mRanks;
                HIPOK(hipStreamWaitEvent(s, mFree[(size_t)to], 0));   // (ncclRecv gives the real exchange the same guarantee: it is posted in the receiver's stream order)
                HIPOK(hipMemcpyAsync(mRecv[(size_t)to] + (size_t)rank * n, send, n * sizeof(float), hipMemcpyDeviceToDevice, s));
            }
            HIPOK(hipEventRecord(mReady[(size_t)rank], s));   // "my slab has landed everywhere"
            if (!mBar.wait()) return false;
            for (int r = 0; r < mRanks; r++) HIPOK(hipStreamWaitEvent(s, mReady[(size_t)r], 0));
            HIPOK(hipEventRecord(mDone[(size_t)rank], s));
            mHave[(size_t)rank] = 1;
            return mBar.wait();
        }
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
    bool allToAllV(Shared &sh, int rank, const float *send, const long long *sendCounts, long long strideElems, int elemFloats,
                   float *recv, long long *recvCounts, hipStream_t s) override
    {
        if (!events(sh, rank)) return false;
        mSend[(size_t)rank] = send;
        mCounts[(size_t)rank].assign(sendCounts, sendCounts + mRanks);
        HIPOK(hipEventRecord(mReady[(size_t)rank], s));
        if (!mBar.wait()) return false;
        const size_t ef = (size_t)elemFloats, st = (size_t)strideElems * ef;
        for (int r = 0; r < mRanks; r++) {   // pull what everybody has for me
            recvCounts[r] = mCounts[(size_t)r][(size_t)rank];
            HIPOK(hipStreamWaitEvent(s, mReady[(size_t)r], 0));
            if (recvCounts[r] > 0)
                HIPOK(hipMemcpyAsync(recv + (size_t)r * st, mSend[(size_t)r] + (size_t)rank * st, (size_t)recvCounts[r] * ef * sizeof(float), hipMemcpyDeviceToDevice, s));
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
            HIPOK(hipEventCreateWithFlags(&mFree[(size_t)rank], hipEventDisableTiming));
        }
        return true;
    }
    std::vector<const float *> mSend;
    std::vector<float *> mRecv;
    std::vector<std::vector<long long>> mCounts;
    std::vector<hipEvent_t> mFree;
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
// the iterations of renderer `rid` in the timed region
void renderer_schedule(const FarmConfig &cfg, int R, int rid, int *first, int *count)
{
    if (cfg.sameWindow) { *first = cfg.warmup; *count = cfg.iterations / R; }   // every renderer the same radius window
    else static_schedule(cfg.iterations, R, rid, first, count);
}

// one renderer as seen by ONE of its ranks
struct Slot {
    vcm_ctx *ctx;
    Collectives *group;          // NULL when shards == 1
    hipStream_t stream;          // the renderer's kernels
    hipStream_t commStream;      // the RANK's communication stream (shared by its slots)
    hipEvent_t evRecords, evGathered;
    float *local, *gathered;     // slabs: one / shards of them, `slabWords` 4-byte words each
    size_t capWords;             // words a slab can hold
    long long slabWords;         // this iteration's slab
    bool sorted;                 // this iteration uses the sorted exchange (every rank decides the same: same counts, same context shape)
    int first, count;            // iterations of this renderer
    std::vector<long long> counts;
    long long stride, nLocal, nLocalPaths;
    bool exchanging, live;
    Xchg mine;
    // the merge sharded by space: send / receive buffers of the three all-to-alls (light records; queries; merge terms), S segments each
    bool space;
    float *spLight, *spLightIn, *spQ, *spQIn, *spRes, *spResIn;
    long long spLightStride, spQStride;          // elements per segment the buffers hold
    std::vector<long long> spCounts, spCountsIn, spQCounts, spQCountsIn, spResCountsIn;
    Slot() : ctx(NULL), group(NULL), stream(NULL), commStream(NULL), evRecords(NULL), evGathered(NULL), local(NULL), gathered(NULL),
             capWords(0), slabWords(0), sorted(false), first(0), count(0), stride(0), nLocal(0), exchanging(false), live(false), space(false),
             spLight(NULL), spLightIn(NULL), spQ(NULL), spQIn(NULL), spRes(NULL), spResIn(NULL), spLightStride(0), spQStride(0) {}
};

struct RankArgs {
    const FarmConfig *cfg;
    Shared *sh;
    int rank, device, group, shard;
    Collectives *groupComm;      // NULL when shards == 1
    Collectives *world;
    Barrier *startLine;          // the ranks hosted by this process
    FarmResult *result;
    std::mutex *resultMutex;
};

// ---- one step of a rank: every in-flight renderer advances by one iteration.  Four passes over the slots, so that
// (a) every collective is enqueued in the same order on every rank of the group (vcm_farm.hpp), (b) the host waits of
// one renderer (the 7 numbers) find the other renderers' light passes already queued on the GPU, and (c) the small
// exchanges never queue behind a large all-gather of the same step. ----
bool step_light(Shared &sh, const FarmConfig &cfg, Slot &sl, int iteration)
{   // light pass (vertexcm.hxx:321-396)
    GpuTurn turn(sl.stream, NULL);
    VCMOK(vcm_begin_iteration(sl.ctx, iteration, cfg.minLen, cfg.maxLen));
    VCMOK(vcm_trace_light(sl.ctx));
    sl.exchanging = false;
    return true;
}
bool step_counts(Shared &sh, Slot &sl, int shard)
{   // this rank's 7 numbers (one stream synchronisation), then everybody's
    if (!sl.group) return true;
    const int S = sl.group->size();
    if (S > VCM_FARM_MAX_RANKS) { sh.fail("too many shards"); return false; }
    long long n = 0;
    { GpuTurn turn(sl.stream, NULL); VCMOK(vcm_local_light_bbox(sl.ctx, sl.mine.mn, sl.mine.mx, &n)); }
    sl.mine.n = n; sl.nLocal = n;
    Xchg all[VCM_FARM_MAX_RANKS];
    if (!sl.group->exchange(sh, shard, sl.mine, all, sl.commStream)) return false;
    sl.counts.assign((size_t)S, 0);
    sl.stride = 1;
    float gmn[3] = { 1e36f, 1e36f, 1e36f }, gmx[3] = { -1e36f, -1e36f, -1e36f };   // hashgrid.hxx:47-48
    for (int r = 0; r < S; r++) {
        sl.counts[(size_t)r] = all[r].n;
        sl.stride = std::max(sl.stride, all[r].n);
        if (all[r].n > 0)
            for (int k = 0; k < 3; k++) { gmn[k] = std::min(gmn[k], all[r].mn[k]); gmx[k] = std::max(gmx[k], all[r].mx[k]); }
    }
    { GpuTurn turn(sl.stream, NULL); VCMOK(vcm_set_grid_bbox(sl.ctx, gmn, gmx)); }
    sl.space = merge_by_space() && vcm_is_wavefront(sl.ctx, 0) && vcm_sorted_slab_words(sl.ctx, sl.stride) > 0;   // (a merging algorithm, <= 64 shards)
    if (sl.space) {
        // slabs of cells along the box's longest axis, split where the SUM of the ranks' histograms reaches s / S of the photons
        int axis = 0;
        for (int k = 1; k < 3; k++) if (gmx[k] - gmn[k] > gmx[axis] - gmn[axis]) axis = k;
        float lo = 0.f, bw = 1.f;
        Xchg h = sl.mine;
        { GpuTurn turn(sl.stream, NULL); VCMOK(vcm_space_histogram(sl.ctx, axis, &lo, &bw, h.hist)); }
        if (!sl.group->exchange(sh, shard, h, all, sl.commStream)) return false;
        // equal photon COUNTS per slab.  (Weighing a bin by photons^1.5 .. ^3 -- the merge costs ~ photons x queries of a region -- was
        // measured and is worse: a wall perpendicular to the axis puts a fifth of all photons into ONE cell layer, several split points
        // fall onto that layer and slabs come out empty; profiles/r18_space_k4_shards.txt.)
        double sum[256], total = 0;
        for (int b = 0; b < 256; b++) { double n = 0; for (int r = 0; r < S; r++) n += all[r].hist[b]; sum[b] = n; total += n; }
        float splits[VCM_FARM_MAX_RANKS + 1];
        double cum = 0; int b = 0;
        splits[0] = lo;
        for (int s2 = 1; s2 < S; s2++) {
            const double want = total * s2 / S;
            while (b < 256 && cum + sum[b] <= want) { cum += sum[b]; b++; }
            splits[s2] = lo + bw * (float)b;
        }
        { GpuTurn turn(sl.stream, NULL); VCMOK(vcm_space_set_slabs(sl.ctx, axis, splits, S)); }
    }
    return true;
}
bool step_exchange_camera(Shared &sh, const FarmConfig &cfg, Slot &sl, int shard)
{   // start of the all-gather; the part of the camera pass that does not need the other ranks' vertices
    if (!sl.group) return true;
    const int S = sl.group->size();
    // Sorted exchange (round 5, the default): this rank sorts its OWN vertices by hash cell, the slabs travel, every rank
    // merges them cell block by cell block -- the grid build is no longer replicated.  The unsorted exchange of rounds 1-4
    // (records in the reference's order, the whole build on every rank) stays behind SMALLVCM_AMD_SORTED_EXCHANGE=0 and
    // for shapes the sorted slabs do not cover (vcm_sorted_slab_words says so; every rank of the group gets the same answer).
    GpuTurn *turn = new GpuTurn(sl.stream, NULL);
    struct Release { GpuTurn *&t; ~Release() { delete t; t = NULL; } } release = { turn };
    if (sl.space) {
        // light records grouped by the slab that owns their cell (+ halo), all-to-all on the communication stream beside the camera pass
        if (sl.spLightStride < sl.stride) {
            HIPOK(hipStreamSynchronize(sl.stream)); HIPOK(hipStreamSynchronize(sl.commStream));
            if (sl.spLight) { (void)hipFree(sl.spLight); (void)hipFree(sl.spLightIn); }
            sl.spLightStride = sl.stride + sl.stride / 16 + 1024;
            const size_t bytes = (size_t)S * (size_t)sl.spLightStride * VCM_MERGE_RECORD_FLOATS * sizeof(float);
            HIPOK(hipMalloc((void **)&sl.spLight, bytes));
            HIPOK(hipMalloc((void **)&sl.spLightIn, bytes));
        }
        sl.spCounts.assign((size_t)S, 0); sl.spCountsIn.assign((size_t)S, 0);
        if (!sl.group->sendBufferFree(sh, shard, sl.stream)) return false;
        VCMOK(vcm_space_partition_light(sl.ctx, sl.spLight, sl.spLightStride, sl.spCounts.data()));
        HIPOK(hipEventRecord(sl.evRecords, sl.stream));
        HIPOK(hipStreamWaitEvent(sl.commStream, sl.evRecords, 0));
        delete turn; turn = NULL;
        if (!sl.group->allToAllV(sh, shard, sl.spLight, sl.spCounts.data(), sl.spLightStride, VCM_MERGE_RECORD_FLOATS, sl.spLightIn, sl.spCountsIn.data(), sl.commStream)) return false;
        HIPOK(hipEventRecord(sl.evGathered, sl.commStream));
        sl.exchanging = true;
        turn = new GpuTurn(sl.stream, sl.commStream);
        VCMOK(vcm_trace_camera(sl.ctx));
        return true;
    }
    static const bool allowSorted = [] { const char *e = getenv("SMALLVCM_AMD_SORTED_EXCHANGE"); return !(e && e[0] == '0'); }();
    const long long sortedWords = allowSorted ? vcm_sorted_slab_words(sl.ctx, sl.stride) : -1;
    sl.sorted = sortedWords > 0;
    sl.slabWords = sl.sorted ? sortedWords : sl.stride * VCM_MERGE_RECORD_FLOATS;
    if ((size_t)sl.slabWords > sl.capWords) {   // grow the slabs (rare: the counts vary by a fraction of a percent)
        HIPOK(hipStreamSynchronize(sl.stream));
        HIPOK(hipStreamSynchronize(sl.commStream));
        if (sl.local) (void)hipFree(sl.local);
        if (sl.gathered) (void)hipFree(sl.gathered);
        sl.capWords = (size_t)sl.slabWords + (size_t)sl.slabWords / 16 + 16384;
        HIPOK(hipMalloc((void **)&sl.local, sl.capWords * sizeof(float)));
        HIPOK(hipMalloc((void **)&sl.gathered, sl.capWords * (size_t)S * sizeof(float)));
    }
    if (!sl.group->sendBufferFree(sh, shard, sl.stream)) return false;
    if (sl.sorted) VCMOK(vcm_sort_light_records(sl.ctx, sl.local, sl.stride));
    else VCMOK(vcm_export_light_records(sl.ctx, sl.local, sl.nLocal));
    // the all-gather runs on the communication stream, behind the export and next to the camera pass
    HIPOK(hipEventRecord(sl.evRecords, sl.stream));
    HIPOK(hipStreamWaitEvent(sl.commStream, sl.evRecords, 0));
    delete turn; turn = NULL;   // (measurement mode: the collective below meets the other ranks -- nobody holds the device across it)
    if (!sl.group->allGather(sh, shard, sl.local, sl.gathered, (size_t)sl.slabWords, sl.commStream)) return false;
    HIPOK(hipEventRecord(sl.evGathered, sl.commStream));
    sl.exchanging = true;
    turn = new GpuTurn(sl.stream, sl.commStream);
    if (vcm_is_wavefront(sl.ctx, cfg.maxLen)) VCMOK(vcm_trace_camera(sl.ctx));   // needs only the local light vertices
    return true;
}
bool step_finish_space(Shared &sh, Slot &sl, int shard)
{   // the merge sharded by space: own photons' grid, queries to their owners, terms back, resolve
    const int S = sl.group->size();
    {
        GpuTurn turn(sl.stream, sl.commStream);
        HIPOK(hipStreamWaitEvent(sl.stream, sl.evGathered, 0));
        VCMOK(vcm_import_light_records(sl.ctx, sl.spLightIn, sl.spCountsIn.data(), S, sl.spLightStride));
        VCMOK(vcm_build_grid(sl.ctx));
        const long long want = 4 * std::max<long long>(sl.nLocalPaths, 1);
        if (sl.spQStride < want) {
            HIPOK(hipStreamSynchronize(sl.stream)); HIPOK(hipStreamSynchronize(sl.commStream));
            if (sl.spQ) { (void)hipFree(sl.spQ); (void)hipFree(sl.spQIn); (void)hipFree(sl.spRes); (void)hipFree(sl.spResIn); }
            sl.spQStride = want;
            const size_t q = (size_t)S * (size_t)sl.spQStride;
            HIPOK(hipMalloc((void **)&sl.spQ, q * 64)); HIPOK(hipMalloc((void **)&sl.spQIn, q * 64));
            HIPOK(hipMalloc((void **)&sl.spRes, q * 16)); HIPOK(hipMalloc((void **)&sl.spResIn, q * 16));
        }
        sl.spQCounts.assign((size_t)S, 0); sl.spQCountsIn.assign((size_t)S, 0); sl.spResCountsIn.assign((size_t)S, 0);
        if (!sl.group->sendBufferFree(sh, shard, sl.stream)) return false;
        VCMOK(vcm_space_partition_queries(sl.ctx, sl.spQ, sl.spQStride, sl.spQCounts.data()));
    }
    if (!sl.group->allToAllV(sh, shard, sl.spQ, sl.spQCounts.data(), sl.spQStride, 16, sl.spQIn, sl.spQCountsIn.data(), sl.stream)) return false;
    {
        GpuTurn turn(sl.stream, sl.commStream);
        if (!sl.group->sendBufferFree(sh, shard, sl.stream)) return false;
        VCMOK(vcm_space_merge(sl.ctx, sl.spQIn, sl.spQCountsIn.data(), S, sl.spQStride, sl.spRes));
    }
    // the terms travel the other way: what I evaluated for rank r goes to r; what comes back from d are the terms of the queries I sent to d
    if (!sl.group->allToAllV(sh, shard, sl.spRes, sl.spQCountsIn.data(), sl.spQStride, 4, sl.spResIn, sl.spResCountsIn.data(), sl.stream)) return false;
    {
        GpuTurn turn(sl.stream, sl.commStream);
        for (int d = 0; d < S; d++) if (sl.spResCountsIn[(size_t)d] != sl.spQCounts[(size_t)d]) { sh.fail("space merge: a rank returned another number of terms than it was sent queries"); return false; }
        VCMOK(vcm_space_import_results(sl.ctx, sl.spResIn, sl.spQStride));
        VCMOK(vcm_merge(sl.ctx));
        VCMOK(vcm_end_iteration(sl.ctx));
    }
    return true;
}
bool step_finish(Shared &sh, const FarmConfig &cfg, Slot &sl, int shard)
{   // wait for the exchange, grid build, (camera pass,) merge, resolve
    if (sl.space && sl.exchanging) return step_finish_space(sh, sl, shard);
    GpuTurn turn(sl.stream, sl.commStream);
    if (sl.exchanging) {
        HIPOK(hipStreamWaitEvent(sl.stream, sl.evGathered, 0));
        if (sl.sorted) VCMOK(vcm_import_sorted_light_records(sl.ctx, sl.gathered, sl.counts.data(), (int)sl.counts.size(), sl.stride));
        else VCMOK(vcm_import_light_records(sl.ctx, sl.gathered, sl.counts.data(), (int)sl.counts.size(), sl.stride));
    }
    VCMOK(vcm_build_grid(sl.ctx));
    if (!(sl.exchanging && vcm_is_wavefront(sl.ctx, cfg.maxLen))) VCMOK(vcm_trace_camera(sl.ctx));
    VCMOK(vcm_merge(sl.ctx));
    VCMOK(vcm_end_iteration(sl.ctx));
    return true;
}

bool run_steps(Shared &sh, const FarmConfig &cfg, std::vector<Slot> &slots, int shard, bool warm, int steps)
{   // renderers advance in lock-step; warm-up steps use the iteration indices 0 .. steps - 1
    for (int t = 0; t < steps; t++) {
        for (Slot &sl : slots) sl.live = warm || t < sl.count;
        for (Slot &sl : slots) if (sl.live && !step_light(sh, cfg, sl, warm ? t : sl.first + t)) return false;
        for (Slot &sl : slots) if (sl.live && !step_counts(sh, sl, shard)) return false;
        for (Slot &sl : slots) if (sl.live && !step_exchange_camera(sh, cfg, sl, shard)) return false;
        for (Slot &sl : slots) if (sl.live && !step_finish(sh, cfg, sl, shard)) return false;
    }
    return true;
}

void stats_accumulate(vcm_stats &acc, const vcm_stats &s)
{
    acc.lightVertices += s.lightVertices; acc.gridVertices += s.gridVertices; acc.lightRays += s.lightRays; acc.cameraRays += s.cameraRays;
    acc.shadowRays += s.shadowRays; acc.mergeQueries += s.mergeQueries; acc.mergeCandidates += s.mergeCandidates;
    acc.mergeAccepted += s.mergeAccepted; acc.connections += s.connections; acc.lightSplats += s.lightSplats;
    acc.msLight += s.msLight; acc.msGrid += s.msGrid; acc.msCamera += s.msCamera; acc.msTotal += s.msTotal;
    acc.msLightKernel += s.msLightKernel; acc.msCameraKernel += s.msCameraKernel; acc.msMergeKernel += s.msMergeKernel;
    acc.msQuerySort += s.msQuerySort; acc.msConnectKernels += s.msConnectKernels; acc.radius += s.radius;
}
void stats_scale(vcm_stats &a, int n)
{
    if (n <= 0) return;
    a.lightVertices /= n; a.gridVertices /= n; a.lightRays /= n; a.cameraRays /= n; a.shadowRays /= n; a.mergeQueries /= n;
    a.mergeCandidates /= n; a.mergeAccepted /= n; a.connections /= n; a.lightSplats /= n;
    const float f = 1.f / (float)n;
    a.msLight *= f; a.msGrid *= f; a.msCamera *= f; a.msTotal *= f; a.msLightKernel *= f; a.msCameraKernel *= f; a.msMergeKernel *= f;
    a.msQuerySort *= f; a.msConnectKernels *= f; a.radius *= f;
}

bool rank_main(RankArgs &a)
{
    const FarmConfig &cfg = *a.cfg;
    Shared &sh = *a.sh;
    HIPOK(hipSetDevice(a.device));
    const int groups = cfg.ranks / cfg.shards, R = groups * cfg.inflight;
    std::vector<Slot> slots((size_t)cfg.inflight);
    hipStream_t commStream = NULL;
    HIPOK(hipStreamCreateWithFlags(&commStream, hipStreamNonBlocking));
    int maxCount = 0;
    for (int k = 0; k < cfg.inflight; k++) {
        Slot &sl = slots[(size_t)k];
        const int rid = a.group * cfg.inflight + k;
        sl.ctx = vcm_create_sharded(&cfg.scene, cfg.algorithm, cfg.radiusFactor, cfg.radiusAlpha, cfg.baseSeed + rid, a.device,
                                    a.shard, cfg.shards);   // seed: smallvcm.cxx:68
        if (!sl.ctx) { sh.fail(std::string("vcm_create_sharded: ") + vcm_last_error()); return false; }
        HIPOK(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
        sl.commStream = commStream;
        HIPOK(hipEventCreateWithFlags(&sl.evRecords, hipEventDisableTiming));
        HIPOK(hipEventCreateWithFlags(&sl.evGathered, hipEventDisableTiming));
        VCMOK(vcm_set_stream(sl.ctx, sl.stream));
        VCMOK(vcm_reserve(sl.ctx, cfg.maxLen));
        sl.group = cfg.shards > 1 ? a.groupComm : NULL;
        {   // paths of this shard (the library's own split: contiguous index blocks)
            const long long N = (long long)cfg.scene.camera.resolution[0] * (long long)cfg.scene.camera.resolution[1];
            sl.nLocalPaths = (N + cfg.shards - 1) / cfg.shards;
        }
        renderer_schedule(cfg, R, rid, &sl.first, &sl.count);
        maxCount = std::max(maxCount, sl.count);
    }
    // everybody arrives (threads of this process, then the other processes), the rank's streams drained
    auto line = [&]() -> bool {
        for (Slot &sl : slots) VCMOK(vcm_synchronize(sl.ctx));
        HIPOK(hipStreamSynchronize(commStream));
        if (!a.startLine->wait()) return false;
        return a.world->barrier(sh, a.rank, commStream);
    };
    if (cfg.warmup > 0) {   // untimed: iterations 0..warmup-1 of every renderer, then the framebuffers start over
        if (!run_steps(sh, cfg, slots, a.shard, true, cfg.warmup)) return false;
        for (Slot &sl : slots) VCMOK(vcm_clear_framebuffer(sl.ctx));
    }
    if (!line()) return false;
    const auto t0 = std::chrono::steady_clock::now();
    if (!run_steps(sh, cfg, slots, a.shard, false, maxCount)) return false;
    for (Slot &sl : slots) VCMOK(vcm_synchronize(sl.ctx));
    HIPOK(hipStreamSynchronize(commStream));
    const double mine = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    if (!line()) return false;

    // per-rank figures: device time of an iteration of the first renderer (its own stamps), counters on rank 0
    float iterMs = 0.f;
    {
        const int n = std::min(slots[0].count, 64);
        vcm_stats acc; memset(&acc, 0, sizeof(acc));
        for (int ago = 0; ago < n; ago++) { vcm_stats st; VCMOK(vcm_get_stats_at(slots[0].ctx, ago, &st)); stats_accumulate(acc, st); }
        stats_scale(acc, n);
        iterMs = acc.msTotal;
        {   /* times, radius, the grid's vertex count: world rank 0's; the WORK counters: summed over the shards of the first
               renderer that this process hosts (rank < shards), so that a sharded renderer reports what a single one would */
            std::lock_guard<std::mutex> g(*a.resultMutex);
            vcm_stats &m = a.result->meanStats;
            if (a.rank == 0) {
                const vcm_stats work = m;   /* what other shards have added already */
                m = acc;
                m.lightVertices += work.lightVertices; m.lightRays += work.lightRays; m.cameraRays += work.cameraRays; m.shadowRays += work.shadowRays;
                m.mergeQueries += work.mergeQueries; m.mergeCandidates += work.mergeCandidates; m.mergeAccepted += work.mergeAccepted;
                m.connections += work.connections; m.lightSplats += work.lightSplats;
            } else if (a.rank < cfg.shards) {
                m.lightVertices += acc.lightVertices; m.lightRays += acc.lightRays; m.cameraRays += acc.cameraRays; m.shadowRays += acc.shadowRays;
                m.mergeQueries += acc.mergeQueries; m.mergeCandidates += acc.mergeCandidates; m.mergeAccepted += acc.mergeAccepted;
                m.connections += acc.connections; m.lightSplats += acc.lightSplats;
            }
        }
    }
    {   // wall = max over ranks, iteration ms per rank: one small all-reduce of a table every rank fills its row of
        const size_t n = 2 * (size_t)cfg.ranks;
        float *dev = NULL;
        std::vector<float> host(n, 0.f);
        host[(size_t)a.rank] = (float)mine; host[(size_t)cfg.ranks + (size_t)a.rank] = iterMs;
        HIPOK(hipMalloc((void **)&dev, n * sizeof(float)));
        HIPOK(hipMemcpyAsync(dev, host.data(), n * sizeof(float), hipMemcpyHostToDevice, commStream));
        if (a.world->size() > 1 && !a.world->allReduceSum(sh, a.rank, dev, n, commStream)) return false;
        HIPOK(hipMemcpyAsync(host.data(), dev, n * sizeof(float), hipMemcpyDeviceToHost, commStream));
        HIPOK(hipStreamSynchronize(commStream));
        (void)hipFree(dev);
        if (a.rank == cfg.firstRank) {
            std::lock_guard<std::mutex> g(*a.resultMutex);
            double wall = 0;
            for (int r = 0; r < cfg.ranks; r++) wall = std::max(wall, (double)host[(size_t)r]);
            a.result->wallSeconds = wall;
            a.result->rankIterationMs.assign(host.begin() + cfg.ranks, host.end());
        }
    }

    // read-out (smallvcm.cxx:116-142): mean over the used renderers of (running sum / own iterations); a renderer's
    // shards hold partial sums of it, so ONE all-reduce over all ranks does both sums
    int used = 0;
    for (int rid = 0; rid < R; rid++) { int f, c; renderer_schedule(cfg, R, rid, &f, &c); if (c > 0) used++; }
    const size_t n3 = (size_t)((int)cfg.scene.camera.resolution[0]) * (size_t)((int)cfg.scene.camera.resolution[1]) * 3;
    float *acc = NULL, *tmp = NULL;
    HIPOK(hipMalloc((void **)&acc, n3 * sizeof(float)));
    HIPOK(hipMalloc((void **)&tmp, n3 * sizeof(float)));
    HIPOK(hipMemsetAsync(acc, 0, n3 * sizeof(float), slots[0].stream));
    HIPOK(hipStreamSynchronize(slots[0].stream));
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
    if (a.world->size() > 1) {
        if (!a.world->allReduceSum(sh, a.rank, acc, n3, commStream)) return false;
        HIPOK(hipStreamSynchronize(commStream));
    }
    if (a.rank == 0) {
        std::lock_guard<std::mutex> g(*a.resultMutex);
        a.result->image.resize(n3);
        HIPOK(hipMemcpy(a.result->image.data(), acc, n3 * sizeof(float), hipMemcpyDeviceToHost));
    }
    (void)hipFree(acc); (void)hipFree(tmp);
    if (!line()) return false;   // nobody tears a communicator's peer down while a collective runs
    for (Slot &sl : slots) {
        vcm_destroy(sl.ctx);
        if (sl.local) (void)hipFree(sl.local);
        if (sl.gathered) (void)hipFree(sl.gathered);
        if (sl.spLight) { (void)hipFree(sl.spLight); (void)hipFree(sl.spLightIn); }
        if (sl.spQ) { (void)hipFree(sl.spQ); (void)hipFree(sl.spQIn); (void)hipFree(sl.spRes); (void)hipFree(sl.spResIn); }
        (void)hipEventDestroy(sl.evRecords); (void)hipEventDestroy(sl.evGathered);
        (void)hipStreamDestroy(sl.stream);
    }
    (void)hipStreamDestroy(commStream);
    return true;
}

} // namespace

FarmResult farm_render(const FarmConfig &cfg)
{
    FarmResult res;
    res.wallSeconds = 0;
    res.renderers = 0;
    res.rcclRanks = 0;
    memset(&res.meanStats, 0, sizeof(res.meanStats));
    const bool multiProcess = cfg.localRanks != cfg.ranks;
    if (cfg.ranks < 1 || cfg.ranks > VCM_FARM_MAX_RANKS || cfg.shards < 1 || cfg.ranks % cfg.shards || cfg.inflight < 1 ||
        cfg.localRanks < 1 || cfg.firstRank < 0 || cfg.firstRank + cfg.localRanks > cfg.ranks || (int)cfg.devices.size() != cfg.localRanks) {
        res.error = "ranks must be a multiple of shards (at most 64), one device per local rank, inflight >= 1";
        return res;
    }
    const int groups = cfg.ranks / cfg.shards;
    res.renderers = groups * cfg.inflight;
    if (cfg.sameWindow && cfg.iterations % res.renderers) { res.error = "sameWindow: iterations must be a multiple of the renderer count"; return res; }
    if (multiProcess && !cfg.rccl) { res.error = "the in-process stand-in for the collectives needs every rank in one process"; return res; }
    if ((multiProcess || !cfg.uniqueIds.empty()) && cfg.uniqueIds.size() != (size_t)(1 + groups) * sizeof(ncclUniqueId)) {
        res.error = "one process per GPU: pass 1 + ranks / shards ids of vcm_farm_unique_ids";
        return res;
    }
    Shared sh;
    Barrier startLine(cfg.localRanks);
    sh.barriers.push_back(&startLine);
    const ncclUniqueId *ids = cfg.uniqueIds.empty() ? NULL : reinterpret_cast<const ncclUniqueId *>(cfg.uniqueIds.data());   // shipped ids are used even when every rank is local
    // communicators: the world's and ONE per group (vcm_farm.hpp: "Collective order")
    std::vector<Collectives *> groupComms((size_t)groups, (Collectives *)NULL);
    const int lo = cfg.firstRank, hi = cfg.firstRank + cfg.localRanks;
    for (int g = 0; g < groups && cfg.shards > 1; g++) {
        const int g0 = g * cfg.shards, g1 = g0 + cfg.shards;
        const int a = std::max(lo, g0), b = std::min(hi, g1);   // members of group g hosted here: world ranks [a, b)
        if (a >= b) continue;
        std::vector<int> devs(cfg.devices.begin() + (a - lo), cfg.devices.begin() + (b - lo));
        groupComms[(size_t)g] = cfg.rccl ? (Collectives *)new RcclCollectives(sh, cfg.shards, a - g0, devs, ids ? ids + 1 + g : NULL)
                                         : (Collectives *)new ThreadCollectives(sh, cfg.shards);
    }
    Collectives *world = cfg.rccl ? (Collectives *)new RcclCollectives(sh, cfg.ranks, cfg.firstRank, cfg.devices, ids)
                                  : (Collectives *)new ThreadCollectives(sh, cfg.ranks);
    res.rcclRanks = world->rcclRanks();
    std::mutex resultMutex;
    std::vector<RankArgs> args((size_t)cfg.localRanks);
    std::vector<std::thread> threads;
    if (!sh.failed) {
        for (int i = 0; i < cfg.localRanks; i++) {
            RankArgs &a = args[(size_t)i];
            const int r = cfg.firstRank + i;
            a.cfg = &cfg; a.sh = &sh; a.rank = r; a.device = cfg.devices[(size_t)i]; a.group = r / cfg.shards; a.shard = r % cfg.shards;
            a.groupComm = groupComms[(size_t)a.group]; a.world = world; a.startLine = &startLine; a.result = &res; a.resultMutex = &resultMutex;
            threads.emplace_back([&a, &sh] { if (!rank_main(a)) sh.fail("rank failed"); });
        }
        for (std::thread &t : threads) t.join();
    }
    for (Collectives *c : groupComms) delete c;
    delete world;
    res.error = sh.error;
    return res;
}

// ---- C-ABI (include/smallvcm_amd_farm.h) ----
static thread_local std::string g_farmError;

extern "C" {

const char *vcm_farm_last_error(void) { return g_farmError.c_str(); }

int vcm_farm_unique_id_bytes(void) { return (int)sizeof(ncclUniqueId); }
unsigned vcm_farm_sizeof_config(void) { return (unsigned)sizeof(vcm_farm_config); }
unsigned vcm_farm_sizeof_result(void) { return (unsigned)sizeof(vcm_farm_result); }

int vcm_farm_unique_ids(void *out, int n)
{
    if (!out || n < 1) { g_farmError = "vcm_farm_unique_ids: bad argument"; return -1; }
    for (int i = 0; i < n; i++) {
        ncclUniqueId id;
        const ncclResult_t r = ncclGetUniqueId(&id);
        if (r != ncclSuccess) { g_farmError = std::string("ncclGetUniqueId: ") + ncclGetErrorString(r); return -1; }
        memcpy((char *)out + (size_t)i * sizeof(id), &id, sizeof(id));
    }
    return 0;
}

int vcm_farm_render(const vcm_farm_config *c, vcm_farm_result *out, float *imageOut)
{
    if (!c || !out) { g_farmError = "vcm_farm_render: NULL argument"; return -1; }
    memset(out, 0, sizeof(*out));
    if (c->localRanks < 1 || c->localRanks > VCM_FARM_MAX_RANKS) { g_farmError = "vcm_farm_render: localRanks out of range"; return -1; }
    FarmConfig fc;
    fc.scene = c->scene; fc.algorithm = c->algorithm; fc.radiusFactor = c->radiusFactor; fc.radiusAlpha = c->radiusAlpha;
    fc.baseSeed = c->baseSeed; fc.minLen = c->minLen; fc.maxLen = c->maxLen; fc.iterations = c->iterations; fc.warmup = c->warmup;
    fc.sameWindow = c->sameWindow != 0; fc.ranks = c->ranks; fc.firstRank = c->firstRank; fc.localRanks = c->localRanks;
    fc.devices.assign(c->devices, c->devices + c->localRanks);
    fc.shards = c->shards; fc.inflight = c->inflight; fc.rccl = c->collectives == 0;
    if (c->uniqueIds && c->nUniqueIds > 0)
        fc.uniqueIds.assign((const char *)c->uniqueIds, (const char *)c->uniqueIds + (size_t)c->nUniqueIds * sizeof(ncclUniqueId));
    const FarmResult r = farm_render(fc);
    if (!r.error.empty()) { g_farmError = r.error; return -1; }
    out->wallSeconds = r.wallSeconds; out->renderers = r.renderers; out->rcclRanks = r.rcclRanks; out->meanStats = r.meanStats;
    for (size_t i = 0; i < r.rankIterationMs.size() && i < VCM_FARM_MAX_RANKS; i++) out->rankIterationMs[i] = r.rankIterationMs[i];
    if (imageOut && !r.image.empty()) memcpy(imageOut, r.image.data(), r.image.size() * sizeof(float));
    return 0;
}

} // extern "C"
