// What the LDS operations of a pair-dense merge cost on gfx950 (round 6): cycles per wave-instruction, every CU busy,
// 4 waves per SIMD.   hipcc -O3 --offload-arch=gfx950 -o lds_bench lds_bench.hip && ./lds_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
#define REP 256
template <int MODE>
__global__ void __launch_bounds__(256) k(const int *__restrict__ idx, float *out, long long *cycles)
{
    __shared__ __attribute__((aligned(16))) float lds[256 * 20 + 3 * 256 + 64];
    const int tid = threadIdx.x;
    for (int i = tid; i < 256 * 20 + 3 * 256 + 64; i += 256) lds[i] = (float)i;
    __syncthreads();
    int r = idx[(blockIdx.x * 256 + tid) & 65535];   // row 0..63 of this wave's quarter, per lane
    const int base = (tid & ~63);
    float acc = 0.f;
    f4 a4 = {0, 0, 0, 0};
    const long long t0 = clock64();
    for (int it = 0; it < REP; it++) {
        if (MODE == 0) {   // ds_read_b128 of a random row (stride 20 words), 5 per iteration
            const f4 *row = (const f4 *)(lds + (base + r) * 20);
            a4 += row[0] + row[1] + row[2] + row[3] + row[4];
        } else if (MODE == 1) {   // ds_read_b128 of the lane's own row
            const f4 *row = (const f4 *)(lds + tid * 20);
            a4 += row[0] + row[1] + row[2] + row[3] + row[4];
        } else if (MODE == 2) {   // 3 x ds_add_f32, random query (duplicates as they fall)
            atomicAdd(&lds[256 * 20 + base + r], 1.f); atomicAdd(&lds[256 * 20 + 256 + base + r], 1.f); atomicAdd(&lds[256 * 20 + 512 + base + r], 1.f);
        } else if (MODE == 3) {   // 3 x ds_add_f32, distinct addresses
            atomicAdd(&lds[256 * 20 + tid], 1.f); atomicAdd(&lds[256 * 20 + 256 + tid], 1.f); atomicAdd(&lds[256 * 20 + 512 + tid], 1.f);
        } else if (MODE == 4) {   // 5 x ds_bpermute_b32
            int v = __float_as_int(acc) + it;
            for (int j = 0; j < 5; j++) v = __builtin_amdgcn_ds_bpermute(r << 2, v);
            acc += __int_as_float(v);
        } else if (MODE == 5) {   // 5 x ds_read_b32 of a random row, column layout [field][lane]
            for (int j = 0; j < 5; j++) acc += lds[j * 256 + base + r];
        } else if (MODE == 6) {   // 3 x ds_add_u32 random
            atomicAdd((unsigned *)&lds[256 * 20 + base + r], 1u); atomicAdd((unsigned *)&lds[256 * 20 + 256 + base + r], 1u); atomicAdd((unsigned *)&lds[256 * 20 + 512 + base + r], 1u);
        } else if (MODE == 7) {   // 4 x ds_write_b32 to the lane's column (the scan's queue writes)
            for (int j = 0; j < 4; j++) lds[(j + (it & 3)) * 256 + tid] = acc + j;
        } else if (MODE == 8) {   // 3 x read-add-write by the lane itself (owner-side accumulation)
            for (int j = 0; j < 3; j++) lds[256 * 20 + j * 256 + tid] += 1.f;
        }
        r = (r * 5 + 1) & 63;
        __builtin_amdgcn_wave_barrier();
    }
    const long long t1 = clock64();
    out[blockIdx.x * 256 + tid] = acc + a4.x + a4.y + a4.z + a4.w + lds[tid];
    if (tid == 0) cycles[blockIdx.x] = t1 - t0;
}
template <int MODE> void run(const char *name, int per, const int *idx, float *out, long long *cyc)
{
    const int blocks = 256 * 4;   // 4 blocks per CU: 4 waves per SIMD
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, idx, out, cyc);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, idx, out, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), cyc, blocks * sizeof(long long), hipMemcpyDeviceToHost);
    double s = 0; for (auto v : h) s += (double)v;
    // clock64 = 100 MHz wall clock on gfx9; use the kernel time: 16 waves per CU issue REP * per instructions each
    const double instPerCu = 16.0 * REP * per;
    printf("%-52s %7.3f ms  %6.1f ns per wave-instruction per CU = %5.1f cycles at 2.4 GHz\n", name, ms, ms * 1e6 / instPerCu, ms * 1e6 / instPerCu * 2.4);
}
int main()
{
    int *idx; float *out; long long *cyc;
    std::vector<int> h(65536);
    uint32_t s = 12345u;
    for (auto &v : h) { s = s * 1664525u + 1013904223u; v = (s >> 10) & 63; }
    hipMalloc(&idx, 65536 * 4); hipMalloc(&out, 1024 * 256 * 4); hipMalloc(&cyc, 1024 * 8);
    hipMemcpy(idx, h.data(), 65536 * 4, hipMemcpyHostToDevice);
    run<0>("ds_read_b128, random row of 20 words", 5, idx, out, cyc);
    run<1>("ds_read_b128, own row of 20 words", 5, idx, out, cyc);
    run<5>("ds_read_b32, random lane's column entry", 5, idx, out, cyc);
    run<4>("ds_bpermute_b32, random lane", 5, idx, out, cyc);
    run<2>("ds_add_f32, random address (duplicates)", 3, idx, out, cyc);
    run<3>("ds_add_f32, distinct addresses", 3, idx, out, cyc);
    run<6>("ds_add_u32, random address (duplicates)", 3, idx, out, cyc);
    run<7>("ds_write_b32, own column", 4, idx, out, cyc);
    run<8>("ds read + add + write, own column", 3, idx, out, cyc);
    return 0;
}
