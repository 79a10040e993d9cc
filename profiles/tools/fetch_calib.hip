// fetch_calib.hip -- what do rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for the access patterns of THIS
// code base?  MI355X_MICROARCH.md (section HBM) calibrates one pattern only: wide coalesced streaming reads
// (16 B/lane) report exactly half their bytes; "other access widths and WRITE_SIZE are uncalibrated: calibrate on a
// known byte count in your own access pattern".  Most kernels of the VCM iteration gather: 4-byte table entries,
// 12-byte positions, 16-byte list entries, 80-byte vertex records.
//
// Every kernel below moves a KNOWN number of useful bytes over a buffer far larger than L2 + Infinity Cache (so a
// line is touched by one kernel instance at most once unless the pattern says otherwise), and prints
//     CALIB <kernel> useful_bytes=<n> line_bytes=<n> ms=<t>
// (line_bytes = distinct 128-byte lines touched x 128).  Run it three times:
//     ./fetch_calib                                            timing only
//     rocprofv3 --pmc FETCH_SIZE --kernel-trace ... ./fetch_calib
//     rocprofv3 --pmc WRITE_SIZE --kernel-trace ... ./fetch_calib
// profiles/tools/fetch_calib.py joins the three into profiles/<tag>_fetch_calib.json: per pattern the factor
// useful / reported and line / reported, and the GB/s either reading implies (a factor that implies more than the
// 8 TB/s peak is not what the hardware moved).
//
// build: hipcc -O3 --offload-arch=gfx950 -o fetch_calib fetch_calib.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>

#define CHECK(e) do { hipError_t r_ = (e); if (r_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #e, hipGetErrorString(r_)); exit(1); } } while (0)

// a bijection of [0, n) for n a power of two: odd multiplier, then xor-shift (both invertible mod 2^k)
__device__ __forceinline__ uint64_t scramble(uint64_t i, uint64_t mask)
{
    i = (i * 0x9E3779B97F4A7C15ull) & mask;
    i ^= i >> 17; i &= mask;
    i = (i * 0xBF58476D1CE4E5B9ull) & mask;
    return i;
}

// ---- reads ----
template <int W4>   // W4 floats per lane, coalesced: lane i reads floats [i*W4, (i+1)*W4)
__global__ void __launch_bounds__(256) read_stream(const float *__restrict__ p, uint64_t nElems, float *sink)
{
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nElems; i += (uint64_t)gridDim.x * blockDim.x) {
        const float *q = p + i * W4;
#pragma unroll
        for (int k = 0; k < W4; k++) acc += q[k];
    }
    if (acc == 123.456f) *sink = acc;
}

// G4 floats per element; elements PITCH4 floats apart; lane i reads element scramble(i): a random gather.
// PITCH4 == G4: packed records (neighbouring records share lines, but nobody reads the neighbour soon);
// PITCH4 >= 64: every element in a 128-byte line (or two) of its own.
template <int G4, int PITCH4>
__global__ void __launch_bounds__(256) read_gather(const float *__restrict__ p, uint64_t nElems, uint64_t nRead, float *sink)
{
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nRead; i += (uint64_t)gridDim.x * blockDim.x) {
        const float *q = p + scramble(i, nElems - 1) * PITCH4;
#pragma unroll
        for (int k = 0; k < G4; k++) acc += q[k];
    }
    if (acc == 123.456f) *sink = acc;
}

// the merge kernel's pattern: every lane walks its OWN short contiguous run (RUN4 floats, 16 bytes per step) at a
// random place; runs of neighbouring lanes are unrelated
template <int RUN4>
__global__ void __launch_bounds__(256) read_runs(const float *__restrict__ p, uint64_t nRuns, uint64_t nRead, float *sink)
{
    typedef float f4 __attribute__((ext_vector_type(4), aligned(4)));
    float acc = 0.f;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nRead; i += (uint64_t)gridDim.x * blockDim.x) {
        const float *q = p + scramble(i, nRuns - 1) * RUN4;
#pragma unroll 1
        for (int k = 0; k < RUN4; k += 4) { const f4 v = *(const f4 *)(q + k); acc += v.x + v.y + v.z + v.w; }
    }
    if (acc == 123.456f) *sink = acc;
}

// ---- writes ----
template <int W4>
__global__ void __launch_bounds__(256) write_stream(float *p, uint64_t nElems)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nElems; i += (uint64_t)gridDim.x * blockDim.x) {
        float *q = p + i * W4;
#pragma unroll
        for (int k = 0; k < W4; k++) q[k] = (float)k;
    }
}
template <int G4, int PITCH4>
__global__ void __launch_bounds__(256) write_scatter(float *p, uint64_t nElems, uint64_t nWrite)
{
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nWrite; i += (uint64_t)gridDim.x * blockDim.x) {
        float *q = p + scramble(i, nElems - 1) * PITCH4;
#pragma unroll
        for (int k = 0; k < G4; k++) q[k] = (float)k;
    }
}
// returning atomics on a table (the counting sorts): one 4-byte RMW per lane at a random word
__global__ void __launch_bounds__(256) atomic_scatter(int *p, uint64_t nWords, uint64_t n, int *sink)
{
    int acc = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (uint64_t)gridDim.x * blockDim.x)
        acc += atomicAdd(&p[scramble(i, nWords - 1)], 1);
    if (acc == -12345) *sink = acc;
}

static hipEvent_t e0, e1;
static void report(const char *name, double useful, double lines, float ms)
{
    printf("CALIB %s useful_bytes=%.0f line_bytes=%.0f ms=%.4f\n", name, useful, lines, ms);
    fflush(stdout);
}
#define TIMED(name, useful, lines, launch) do {                       \
        launch; CHECK(hipDeviceSynchronize());   /* warm (TLB, clocks) */ \
        CHECK(hipEventRecord(e0, 0)); launch; CHECK(hipEventRecord(e1, 0)); CHECK(hipEventSynchronize(e1)); \
        float ms_ = 0; CHECK(hipEventElapsedTime(&ms_, e0, e1)); report(name, (double)(useful), (double)(lines), ms_); } while (0)

static double lines_of(double nElems, int bytes, int pitchBytes)
{   // expected distinct 128-byte lines per element when elements start at multiples of pitchBytes
    if (pitchBytes % 128 == 0) return nElems * ((bytes + 127) / 128) * 128.0;
    // packed / unaligned: an element of b bytes at a 4-byte aligned offset spans 1 + (b - 4) / 128 lines on average,
    // but packed elements share lines: all lines of the buffer are touched once the whole buffer is read
    return nElems * (double)pitchBytes;
}

int main(int argc, char **argv)
{
    const uint64_t GiB = 1ull << 30;
    const uint64_t bufBytes = (argc > 1 ? (uint64_t)atoll(argv[1]) : 8) * GiB;   // >> 256 MiB Infinity Cache
    float *buf = NULL, *sink = NULL;
    CHECK(hipMalloc((void **)&buf, bufBytes));
    CHECK(hipMalloc((void **)&sink, 256));
    CHECK(hipMemset(buf, 0, bufBytes));
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    const dim3 g(256 * 16), b(256);
    const uint64_t nFloats = bufBytes / 4;

    // streaming reads, 4 / 8 / 16 bytes per lane over 4 GiB
    const uint64_t sBytes = 4 * GiB;
    TIMED("read_stream_4B", sBytes, sBytes, (read_stream<1><<<g, b>>>(buf, sBytes / 4, sink)));
    TIMED("read_stream_8B", sBytes, sBytes, (read_stream<2><<<g, b>>>(buf, sBytes / 8, sink)));
    TIMED("read_stream_16B", sBytes, sBytes, (read_stream<4><<<g, b>>>(buf, sBytes / 16, sink)));

    // random gathers, one element per 256-byte pitch (a line of its own): 16 M elements
    {
        const uint64_t nEl = nFloats / 64 > (1ull << 24) ? (1ull << 24) : (1ull << 23), nRead = nEl;
        TIMED("read_gather_4B_sparse", nRead * 4, lines_of((double)nRead, 4, 256), (read_gather<1, 64><<<g, b>>>(buf, nEl, nRead, sink)));
        TIMED("read_gather_12B_sparse", nRead * 12, lines_of((double)nRead, 12, 256), (read_gather<3, 64><<<g, b>>>(buf, nEl, nRead, sink)));
        TIMED("read_gather_16B_sparse", nRead * 16, lines_of((double)nRead, 16, 256), (read_gather<4, 64><<<g, b>>>(buf, nEl, nRead, sink)));
        TIMED("read_gather_80B_sparse", nRead * 80, lines_of((double)nRead, 80, 256), (read_gather<20, 64><<<g, b>>>(buf, nEl, nRead, sink)));
    }
    // random gathers over PACKED records, every record read exactly once (the whole array = 2^26 records)
    {
        const uint64_t nEl = 1ull << 26;
        TIMED("read_gather_4B_packed", nEl * 4, nEl * 4, (read_gather<1, 1><<<g, b>>>(buf, nEl, nEl, sink)));
        TIMED("read_gather_16B_packed", nEl * 16, nEl * 16, (read_gather<4, 4><<<g, b>>>(buf, nEl, nEl, sink)));
    }
    {
        const uint64_t nEl = 1ull << 24;
        TIMED("read_gather_80B_packed", nEl * 80, nEl * 80, (read_gather<20, 20><<<g, b>>>(buf, nEl, nEl, sink)));
        TIMED("read_gather_64B_packed", nEl * 64, nEl * 64, (read_gather<16, 16><<<g, b>>>(buf, nEl, nEl, sink)));
    }
    // per-lane runs (k_merge_walk): 2^24 runs of 64 floats = 256 bytes each, 16 bytes per step
    {
        const uint64_t nRuns = 1ull << 24;
        TIMED("read_runs_256B", nRuns * 256, nRuns * 256, (read_runs<64><<<g, b>>>(buf, nRuns, nRuns, sink)));
        TIMED("read_runs_64B", nRuns * 64, nRuns * 64, (read_runs<16><<<g, b>>>(buf, nRuns, nRuns, sink)));
    }
    // writes
    TIMED("write_stream_4B", sBytes, sBytes, (write_stream<1><<<g, b>>>(buf, sBytes / 4)));
    TIMED("write_stream_16B", sBytes, sBytes, (write_stream<4><<<g, b>>>(buf, sBytes / 16)));
    {
        const uint64_t nEl = 1ull << 24;
        TIMED("write_scatter_4B_sparse", nEl * 4, nEl * 128, (write_scatter<1, 64><<<g, b>>>(buf, nEl, nEl)));
        TIMED("write_scatter_16B_sparse", nEl * 16, nEl * 128, (write_scatter<4, 64><<<g, b>>>(buf, nEl, nEl)));
        TIMED("write_scatter_80B_packed", nEl * 80, nEl * 80, (write_scatter<20, 20><<<g, b>>>(buf, nEl, nEl)));
    }
    {
        const uint64_t nEl = 1ull << 26;
        TIMED("write_scatter_4B_packed", nEl * 4, nEl * 4, (write_scatter<1, 1><<<g, b>>>(buf, nEl, nEl)));
        TIMED("write_scatter_16B_packed", nEl * 16, nEl * 16, (write_scatter<4, 4><<<g, b>>>(buf, nEl, nEl)));
    }
    {   // 2^24 returning atomics over a 2^22-word table (16 MB: the cell / bucket tables of a 2048^2 frame)
        CHECK(hipMemset(buf, 0, 1ull << 24));
        const uint64_t n = 1ull << 24;
        TIMED("atomic_add_4B_table16MB", n * 4, n * 4, (atomic_scatter<<<g, b>>>((int *)buf, 1ull << 22, n, (int *)sink)));
    }
    CHECK(hipFree(buf)); CHECK(hipFree(sink));
    return 0;
}
