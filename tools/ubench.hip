// tools/ubench.hip -- gfx950 micro-experiments backing design decisions in DESIGN.md.
//   1. exhaustive check: which cheap reciprocal / sqrt / rsqrt sequences equal the IEEE
//      correctly-rounded result for EVERY binary32 input in the range the kernel uses
//   2. VALU issue rate: v_fma_f32 vs v_pk_fma_f32 (is packed fp32 a lever on CDNA4?)
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench.hip -o tools/ubench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ float rcp_nr1(float x)
{
    float r = __builtin_amdgcn_rcpf(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
__device__ __forceinline__ float rcp_nr2(float x)
{
    float r = rcp_nr1(x);
    float e = __builtin_fmaf(-x, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}
// sqrt via rsq + 2 corrections (Markstein-style)
__device__ __forceinline__ float sqrt_fast(float x)
{
    float y = __builtin_amdgcn_rsqf(x);
    float g = x * y;
    float h = 0.5f * y;
    float r = __builtin_fmaf(-h, g, 0.5f);
    g = __builtin_fmaf(g, r, g);
    h = __builtin_fmaf(h, r, h);
    float d = __builtin_fmaf(-g, g, x);
    return __builtin_fmaf(d, h, g);
}

// mode 0: rcp_nr1 vs 1/x ; 1: rcp_nr2 vs 1/x ; 2: sqrt_fast vs sqrtf ; 3: raw v_rcp vs 1/x
__global__ void exhaustive(int mode, uint32_t lo, uint32_t hi, unsigned long long* mism, uint32_t* examples)
{
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t b = (uint64_t)lo + i; b <= hi; b += stride) {
        float x = __uint_as_float((uint32_t)b);
        float want, got;
        if (mode == 0) { want = 1.0f / x; got = rcp_nr1(x); }
        else if (mode == 1) { want = 1.0f / x; got = rcp_nr2(x); }
        else if (mode == 2) { want = __builtin_sqrtf(x); got = sqrt_fast(x); }
        else { want = 1.0f / x; got = __builtin_amdgcn_rcpf(x); }
        if (__float_as_uint(want) != __float_as_uint(got)) {
            unsigned long long k = atomicAdd(mism, 1ull);
            if (k < 16) examples[k] = (uint32_t)b;
        }
    }
}

template <int PK>
__global__ void fma_rate(float* out, int iters)
{
    typedef float f2 __attribute__((ext_vector_type(2)));
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    if (PK) {
        f2 v0 = {a0, a1}, v1 = {a2, a3}, v2 = {a4, a5}, v3 = {a6, a7};
        f2 mm = {m, m}, cc = {c, c};
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(mm), "v"(cc));
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = v0.x + v0.y + v1.x + v1.y + v2.x + v2.y + v3.x + v3.y;
    } else {
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
            }
        }
        out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    }
}

static void run_exhaustive(const char* name, int mode, float flo, float fhi)
{
    unsigned long long* d_m; uint32_t* d_e;
    CK(hipMalloc(&d_m, 8)); CK(hipMalloc(&d_e, 64));
    CK(hipMemset(d_m, 0, 8)); CK(hipMemset(d_e, 0, 64));
    uint32_t lo, hi; memcpy(&lo, &flo, 4); memcpy(&hi, &fhi, 4);
    hipLaunchKernelGGL(exhaustive, dim3(4096), dim3(256), 0, 0, mode, lo, hi, d_m, d_e);
    CK(hipDeviceSynchronize());
    unsigned long long m; uint32_t ex[16];
    CK(hipMemcpy(&m, d_m, 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(ex, d_e, 64, hipMemcpyDeviceToHost));
    printf("%-34s range [%g, %g] (%u inputs): %llu mismatches", name, flo, fhi, hi - lo + 1, m);
    for (unsigned k = 0; k < (m < 4 ? m : 4); ++k) printf(" 0x%08x", ex[k]);
    printf("\n");
    hipFree(d_m); hipFree(d_e);
}

template <int PK>
static void run_rate(const char* name, int waves_per_simd)
{
    int cus = 256;
    int blocks = cus * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    float* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    int iters = 20000;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(fma_rate<PK>, dim3(blocks), dim3(256), 0, 0, d, 100);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(fma_rate<PK>, dim3(blocks), dim3(256), 0, 0, d, iters);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    double inst = (double)blocks * 4 * iters * 8 * (PK ? 4 : 8);  // wave-instructions
    double flop = inst * 64 * 2 * (PK ? 2 : 1);
    printf("%-14s %d waves/SIMD: %.3f ms  %.1f TFLOP/s  %.2f cyc/wave-instr/SIMD (at 2.4 GHz)\n", name, waves_per_simd, ms,
           flop / ms * 1e-9, ms * 1e-3 * 2.4e9 / (inst / (cus * 4.0)));
    hipFree(d);
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    printf("device: %s (%s), %d CUs, clock %d kHz\n", p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run_exhaustive("v_rcp_f32 raw", 3, 1e-8f, 1e20f);
    run_exhaustive("rcp + 1 Newton (2 fma)", 0, 1e-8f, 1e20f);
    run_exhaustive("rcp + 2 Newton (4 fma)", 1, 1e-8f, 1e20f);
    run_exhaustive("rcp + 1 Newton (2 fma)", 0, 1.17549435e-38f, 3.4e38f);
    run_exhaustive("rsq-based sqrt (Markstein)", 2, 1e-30f, 1e30f);
    run_exhaustive("rsq-based sqrt (Markstein)", 2, 1.17549435e-38f, 3.4e38f);
    for (int w : {1, 2, 4, 8}) { run_rate<0>("v_fma_f32", w); run_rate<1>("v_pk_fma_f32", w); }
    return 0;
}
