// valu_bench.hip -- issue cost of the VALU instructions the VCM kernels are made of, on gfx950.
//   hipcc -O2 --offload-arch=gfx950 -o valu_bench profiles/tools/valu_bench.hip && ./valu_bench
// Every SIMD runs `waves` waves; a wave executes ITER x 16 instructions of one kind in 8 independent dependency
// chains (inline asm, so the compiler cannot fold or pack them).  Printed: cycles per wave-instruction and SIMD
// = time x clock / (instructions per SIMD).  Answers the questions DESIGN.md section 5 needs for its VALU diet:
// does a packed fp32 operation cost one issue slot or two, what do binary64 and the transcendental unit cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define ITER 4096
#define CHK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

#define KERNEL32(NAME, ASM)                                                                         \
__global__ void __launch_bounds__(256) NAME(float *out, float seed)                                 \
{                                                                                                   \
    float a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const float b = 1.0000001f, c = 1e-9f;                                                          \
    for (int i = 0; i < ITER; i++) {                                                                \
        _Pragma("unroll") for (int r = 0; r < 2; r++) {                                             \
            asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                           \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;             \
}
#define KERNEL64(NAME, ASM)                                                                         \
__global__ void __launch_bounds__(256) NAME(float *out, float seed)                                 \
{                                                                                                   \
    double a0 = seed + threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; \
    const double b = 1.0000001, c = 1e-9;                                                           \
    for (int i = 0; i < ITER; i++) {                                                                \
        _Pragma("unroll") for (int r = 0; r < 2; r++) {                                             \
            asm volatile(ASM : "+v"(a0) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a1) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a2) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a3) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a4) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a5) : "v"(b), "v"(c)); \
            asm volatile(ASM : "+v"(a6) : "v"(b), "v"(c)); asm volatile(ASM : "+v"(a7) : "v"(b), "v"(c)); \
        }                                                                                           \
    }                                                                                               \
    out[blockIdx.x * blockDim.x + threadIdx.x] = (float)(a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7);    \
}

KERNEL32(k_mul_f32, "v_mul_f32 %0, %0, %1")
KERNEL32(k_add_f32, "v_add_f32 %0, %0, %2")
KERNEL32(k_fma_f32, "v_fma_f32 %0, %0, %1, %2")
KERNEL32(k_mac_sgprlike, "v_fmac_f32 %0, %1, %2")
KERNEL32(k_rcp_f32, "v_rcp_f32 %0, %0")
KERNEL32(k_sqrt_f32, "v_sqrt_f32 %0, %0")
KERNEL32(k_cmp_cnd, "v_cmp_lt_f32 vcc, %0, %1\n\tv_cndmask_b32 %0, %0, %2, vcc")
KERNEL32(k_mul_lo_u32, "v_mul_lo_u32 %0, %0, %1")
KERNEL32(k_mul_hi_u32, "v_mul_hi_u32 %0, %0, %1")
KERNEL64(k_pk_mul_f32, "v_pk_mul_f32 %0, %0, %1")
KERNEL64(k_pk_add_f32, "v_pk_add_f32 %0, %0, %2")
KERNEL64(k_pk_fma_f32, "v_pk_fma_f32 %0, %0, %1, %2")
KERNEL64(k_mul_f64, "v_mul_f64 %0, %0, %1")
KERNEL64(k_add_f64, "v_add_f64 %0, %0, %2")
KERNEL64(k_fma_f64, "v_fma_f64 %0, %0, %1, %2")
KERNEL64(k_rcp_f64, "v_rcp_f64 %0, %0")
KERNEL64(k_sqrt_f64, "v_sqrt_f64 %0, %0")

typedef void (*kern_t)(float *, float);

int main()
{
    hipDeviceProp_t prop;
    CHK(hipGetDeviceProperties(&prop, 0));
    int clockKHz = 0;
    CHK(hipDeviceGetAttribute(&clockKHz, hipDeviceAttributeClockRate, 0));
    const int cus = prop.multiProcessorCount;
    printf("%s: %d CUs, nominal clock %d MHz\n", prop.name, cus, clockKHz / 1000);
    float *out;
    CHK(hipMalloc((void **)&out, (size_t)cus * 16 * 256 * sizeof(float)));
    struct { const char *name; kern_t k; int perIter; } tests[] = {
        { "v_mul_f32", k_mul_f32, 16 }, { "v_add_f32", k_add_f32, 16 }, { "v_fma_f32", k_fma_f32, 16 }, { "v_fmac_f32", k_mac_sgprlike, 16 },
        { "v_pk_mul_f32", k_pk_mul_f32, 16 }, { "v_pk_add_f32", k_pk_add_f32, 16 }, { "v_pk_fma_f32", k_pk_fma_f32, 16 },
        { "v_rcp_f32", k_rcp_f32, 16 }, { "v_sqrt_f32", k_sqrt_f32, 16 }, { "v_cmp+v_cndmask", k_cmp_cnd, 32 },
        { "v_mul_lo_u32", k_mul_lo_u32, 16 }, { "v_mul_hi_u32", k_mul_hi_u32, 16 },
        { "v_mul_f64", k_mul_f64, 16 }, { "v_add_f64", k_add_f64, 16 }, { "v_fma_f64", k_fma_f64, 16 },
        { "v_rcp_f64", k_rcp_f64, 16 }, { "v_sqrt_f64", k_sqrt_f64, 16 },
    };
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    for (int wavesPerSimd = 1; wavesPerSimd <= 4; wavesPerSimd *= 2) {
        printf("-- %d wave(s) per SIMD\n", wavesPerSimd);
        const int blocks = cus * wavesPerSimd;   /* 256 threads = 4 waves = one per SIMD of a CU */
        for (auto &t : tests) {
            hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.f);
            CHK(hipDeviceSynchronize());
            CHK(hipEventRecord(e0));
            for (int r = 0; r < 3; r++) hipLaunchKernelGGL(t.k, dim3(blocks), dim3(256), 0, 0, out, 1.f);
            CHK(hipEventRecord(e1));
            CHK(hipEventSynchronize(e1));
            float ms = 0.f;
            CHK(hipEventElapsedTime(&ms, e0, e1));
            const double instrPerSimd = 3.0 * wavesPerSimd * (double)ITER * t.perIter;
            const double cyc = (ms * 1e-3) * (clockKHz * 1e3) / instrPerSimd;
            printf("  %-18s %7.3f ms   %6.2f cycles per wave-instruction and SIMD (at the nominal clock)\n", t.name, ms / 3, cyc);
        }
    }
    return 0;
}
