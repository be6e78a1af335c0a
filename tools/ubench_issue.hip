// ubench_issue.hip -- issue cost of the instruction patterns the trace kernel is made of, measured
// the way the kernel runs them: many waves per SIMD, each repeating a short group of independent
// instructions.  Output: SIMD cycles per group (at the measured shader clock) -- the cost model
// behind the choices documented in DESIGN.md (survivor masks via v_addc, lane masks instead of
// per-lane booleans, binary64 only where PTSPEC demands it).
//   hipcc --offload-arch=gfx950 -O2 -o tools/ubench_issue tools/ubench_issue.hip && ./tools/ubench_issue
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); exit(1); } } while (0)

// every pattern works on 8 float accumulators a0..a7 (+ 4 doubles d0..d3) so that consecutive
// instructions are independent; REP8(X) expands X for k = 0..7
#define BODY_BEGIN(ID) template <> __device__ __forceinline__ void body<ID>(float& a0, float& a1, float& a2, float& a3, float& a4, float& a5, float& a6, float& a7, double& d0, double& d1, double& d2, double& d3, float m, float c, float sm, unsigned& u0, unsigned& u1) {
#define BODY_END }

template <int ID> __device__ __forceinline__ void body(float&, float&, float&, float&, float&, float&, float&, float&, double&, double&, double&, double&, float, float, float, unsigned&, unsigned&);

#define A8(INS) asm volatile(INS(0) INS(1) INS(2) INS(3) INS(4) INS(5) INS(6) INS(7) : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "+v"(u0), "+v"(u1) : "v"(m), "v"(c), "s"(sm) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
// operands: %0-%7 accumulators, %8 %9 unsigned, %10 m, %11 c, %12 scalar

// 0: v_fma_f32 all-VGPR
#define I0(k) "v_fma_f32 %" #k ", %" #k ", %10, %11\n\t"
BODY_BEGIN(0) A8(I0); BODY_END
// 1: v_fma_f32 with an SGPR operand
#define I1(k) "v_fma_f32 %" #k ", %" #k ", %12, %11\n\t"
BODY_BEGIN(1) A8(I1); BODY_END
// 2: v_mul_f32 with a 32-bit literal
#define I2(k) "v_mul_f32_e32 %" #k ", 0x3f7fbe77, %" #k "\n\t"
BODY_BEGIN(2) A8(I2); BODY_END
// 3: v_cmp (vcc) + v_cndmask (vcc)
#define I3(k) "v_cmp_lt_f32_e32 vcc, %10, %" #k "\n\ts_nop 1\n\tv_cndmask_b32_e32 %" #k ", %" #k ", %11, vcc\n\t"
BODY_BEGIN(3) A8(I3); BODY_END
// 4: v_cmp (sgpr pair) + v_cndmask (sgpr pair)
#define I4(k) "v_cmp_lt_f32_e64 s[20:21], %10, %" #k "\n\ts_nop 1\n\tv_cndmask_b32_e64 %" #k ", %" #k ", %11, s[20:21]\n\t"
BODY_BEGIN(4) A8(I4); BODY_END
// 5: 2 v_cmp + s_and + v_cndmask_e64 + v_lshl_or (the per-flag cost of the LSB-first survivor mask)
#define I5(k) "v_cmp_lt_f32_e64 s[20:21], %10, %" #k "\n\tv_cmp_gt_f32_e64 s[22:23], %11, %" #k "\n\ts_and_b64 s[20:21], s[20:21], s[22:23]\n\tv_cndmask_b32_e64 %9, 0, 1, s[20:21]\n\tv_lshl_or_b32 %8, %9, 3, %8\n\t"
BODY_BEGIN(5) A8(I5); BODY_END
// 6: 2 v_cmp + s_and + v_addc (the per-flag cost of the MSB-first survivor mask)
#define I6(k) "v_cmp_lt_f32_e64 s[20:21], %10, %" #k "\n\tv_cmp_gt_f32_e64 s[22:23], %11, %" #k "\n\ts_and_b64 s[20:21], s[20:21], s[22:23]\n\tv_addc_co_u32_e64 %8, s[24:25], %8, %8, s[20:21]\n\t"
BODY_BEGIN(6) A8(I6); BODY_END
// 7: v_cmp_e64 alone
#define I7(k) "v_cmp_lt_f32_e64 s[20:21], %10, %" #k "\n\t"
BODY_BEGIN(7) A8(I7); BODY_END
// 8: v_rcp_f32
#define I8(k) "v_rcp_f32_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(8) A8(I8); BODY_END
// 9: v_sqrt_f32
#define I9(k) "v_sqrt_f32_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(9) A8(I9); BODY_END
// 10: v_add_u32
#define I10(k) "v_add_u32_e32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(10) A8(I10); BODY_END
// 11: v_mul_lo_u32
#define I11(k) "v_mul_lo_u32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(11) A8(I11); BODY_END
// 12: v_lshl_or_b32
#define I12(k) "v_lshl_or_b32 %" #k ", %9, 3, %" #k "\n\t"
BODY_BEGIN(12) A8(I12); BODY_END
// 13: v_cvt_f64_f32 + v_cvt_f32_f64 round trip is pattern 16; here v_fma_f64
BODY_BEGIN(13)
    for (int k = 0; k < 2; ++k)
        asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));
BODY_END
// 14: v_mul_f64
BODY_BEGIN(14)
    for (int k = 0; k < 2; ++k)
        asm volatile("v_mul_f64 %0, %0, %4\n\tv_mul_f64 %1, %1, %4\n\tv_mul_f64 %2, %2, %4\n\tv_mul_f64 %3, %3, %4\n\t"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));
BODY_END
// 15: v_add_f64
BODY_BEGIN(15)
    for (int k = 0; k < 2; ++k)
        asm volatile("v_add_f64 %0, %0, %5\n\tv_add_f64 %1, %1, %5\n\tv_add_f64 %2, %2, %5\n\tv_add_f64 %3, %3, %5\n\t"
                     : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3) : "v"((double)m), "v"((double)c));
BODY_END
// 16: v_cvt_f64_f32 + v_cvt_f32_f64 (2 instructions per k, 4 k)
BODY_BEGIN(16)
    asm volatile("v_cvt_f64_f32 %4, %0\n\tv_cvt_f32_f64 %0, %4\n\tv_cvt_f64_f32 %5, %1\n\tv_cvt_f32_f64 %1, %5\n\t"
                 "v_cvt_f64_f32 %6, %2\n\tv_cvt_f32_f64 %2, %6\n\tv_cvt_f64_f32 %7, %3\n\tv_cvt_f32_f64 %3, %7\n\t"
                 : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
BODY_END
// 17: v_readlane_b32 into an SGPR + v_fma using it (SGPR-spill reload pattern)
#define I17(k) "v_readlane_b32 s20, %9, 3\n\ts_nop 1\n\tv_fma_f32 %" #k ", %" #k ", s20, %11\n\t"
BODY_BEGIN(17) A8(I17); BODY_END
// 18: dependent chain of v_fma_f32 (no ILP inside the wave)
#define I18(k) "v_fma_f32 %0, %0, %10, %11\n\t"
BODY_BEGIN(18) A8(I18); BODY_END
// 19: v_cmp_class + v_cndmask vcc (NaN canonicalisation pattern)
#define I19(k) "v_cmp_u_f32_e32 vcc, %" #k ", %" #k "\n\ts_nop 1\n\tv_cndmask_b32_e32 %" #k ", %" #k ", %11, vcc\n\t"
BODY_BEGIN(19) A8(I19); BODY_END
// 20: v_max_f32
#define I20(k) "v_max_f32_e32 %" #k ", %" #k ", %10\n\t"
BODY_BEGIN(20) A8(I20); BODY_END
// 21: v_rsq_f32
#define I21(k) "v_rsq_f32_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(21) A8(I21); BODY_END
// 22: v_mul_hi_u32
#define I22(k) "v_mul_hi_u32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(22) A8(I22); BODY_END
// 23: v_rcp_f64
BODY_BEGIN(23)
    for (int k = 0; k < 2; ++k)
        asm volatile("v_rcp_f64 %0, %0\n\tv_rcp_f64 %1, %1\n\tv_rcp_f64 %2, %2\n\tv_rcp_f64 %3, %3\n\t" : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
BODY_END


// ---- second batch: encodings and packed forms ----------------------------------------------------
#define I30(k) "v_fmac_f32_e32 %" #k ", %10, %11\n\t"
BODY_BEGIN(30) A8(I30); BODY_END
#define I31(k) "v_fmac_f32_e32 %" #k ", %12, %11\n\t"
BODY_BEGIN(31) A8(I31); BODY_END
#define I32(k) "v_mul_f32_e32 %" #k ", %12, %" #k "\n\t"
BODY_BEGIN(32) A8(I32); BODY_END
#define I33(k) "v_sub_f32_e32 %" #k ", %" #k ", %11\n\t"
BODY_BEGIN(33) A8(I33); BODY_END
#define I34(k) "v_subrev_f32_e32 %" #k ", %12, %" #k "\n\t"
BODY_BEGIN(34) A8(I34); BODY_END
#define I35(k) "v_mul_f32_e64 %" #k ", %" #k ", -%12\n\t"
BODY_BEGIN(35) A8(I35); BODY_END
#define I36(k) "v_fma_f32 %" #k ", -%" #k ", %10, %11\n\t"
BODY_BEGIN(36) A8(I36); BODY_END
#define I37(k) "v_add_f32_e32 %" #k ", 1.0, %" #k "\n\t"
BODY_BEGIN(37) A8(I37); BODY_END
#define I38(k) "v_mov_b32_e32 %" #k ", %10\n\t"
BODY_BEGIN(38) A8(I38); BODY_END
#define I39(k) "v_mov_b32_e32 %" #k ", %12\n\t"
BODY_BEGIN(39) A8(I39); BODY_END
#define I40(k) "v_cndmask_b32_e32 %" #k ", %" #k ", %11, vcc\n\t"
BODY_BEGIN(40) A8(I40); BODY_END
#define I41(k) "v_addc_co_u32_e64 %" #k ", s[24:25], %" #k ", %" #k ", s[20:21]\n\t"
BODY_BEGIN(41) asm volatile("s_mov_b64 s[20:21], 0x55" ::: "s20", "s21"); A8(I41); BODY_END
#define I42(k) "v_cmp_gt_f32_e64 s[20:21], |%" #k "|, %10\n\t"
BODY_BEGIN(42) A8(I42); BODY_END
#define I43(k) "v_and_b32_e32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(43) A8(I43); BODY_END
#define I44(k) "v_med3_f32 %" #k ", %" #k ", %10, %11\n\t"
BODY_BEGIN(44) A8(I44); BODY_END
#define I45(k) "v_xor_b32_e32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(45) A8(I45); BODY_END
#define I46(k) "v_lshlrev_b32_e32 %" #k ", 1, %" #k "\n\t"
BODY_BEGIN(46) A8(I46); BODY_END
#define I47(k) "v_cvt_f32_u32_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(47) A8(I47); BODY_END
#define I48(k) "v_ffbh_u32_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(48) A8(I48); BODY_END

// round 2 additions (the LBVH node step's instruction mix)
#define I60(k) "v_fma_mix_f32 %" #k ", %" #k ", %10, %11 op_sel_hi:[1,0,0]\n\t"
BODY_BEGIN(60) A8(I60); BODY_END
#define I61(k) "v_fma_mix_f32 %" #k ", %" #k ", %10, %11 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n\t"
BODY_BEGIN(61) A8(I61); BODY_END
#define I62(k) "v_cvt_f32_ubyte0_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(62) A8(I62); BODY_END
#define I63(k) "v_cvt_f32_ubyte2_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(63) A8(I63); BODY_END
#define I64(k) "v_max3_f32 %" #k ", %" #k ", %10, %11\n\t"
BODY_BEGIN(64) A8(I64); BODY_END
#define I65(k) "v_min3_f32 %" #k ", %" #k ", %10, %11\n\t"
BODY_BEGIN(65) A8(I65); BODY_END
#define I66(k) "v_perm_b32 %" #k ", %" #k ", %10, %9\n\t"
BODY_BEGIN(66) A8(I66); BODY_END
#define I67(k) "v_bfe_u32 %" #k ", %" #k ", 8, 8\n\t"
BODY_BEGIN(67) A8(I67); BODY_END
#define I68(k) "v_and_or_b32 %" #k ", %" #k ", %9, %8\n\t"
BODY_BEGIN(68) A8(I68); BODY_END
#define I69(k) "v_bcnt_u32_b32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(69) A8(I69); BODY_END
#define I70(k) "v_cvt_f32_f16_e32 %" #k ", %" #k "\n\t"
BODY_BEGIN(70) A8(I70); BODY_END
#define I71(k) "v_cvt_f32_ubyte0_e32 %" #k ", %" #k "\n\tv_fma_f32 %" #k ", %" #k ", %10, %11\n\t"
BODY_BEGIN(71) A8(I71); BODY_END
#define I72(k) "v_min_f32_e32 %" #k ", %" #k ", %10\n\t"
BODY_BEGIN(72) A8(I72); BODY_END
#define I73(k) "v_cmp_le_f32_e64 s[20:21], %10, %" #k "\n\tv_addc_co_u32_e64 %8, s[24:25], %8, %8, s[20:21]\n\t"
BODY_BEGIN(73) A8(I73); BODY_END
#define I74(k) "v_or_b32_e32 %" #k ", %" #k ", %9\n\t"
BODY_BEGIN(74) A8(I74); BODY_END
#define I75(k) "v_lshrrev_b32_e32 %" #k ", 8, %" #k "\n\t"
BODY_BEGIN(75) A8(I75); BODY_END

// packed: 4 register pairs p0..p3 (from d0..d3 reinterpretation), one instruction = 2 fp32 results
typedef float pt_f2 __attribute__((ext_vector_type(2)));
#define PK_BODY(ID, INS) BODY_BEGIN(ID) \
    pt_f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}; pt_f2 mm = {m, m}, cc = {c, c}; \
    unsigned long long ss = ((unsigned long long)__float_as_uint(sm) << 32) | __float_as_uint(sm); \
    asm volatile(INS(0) INS(1) INS(2) INS(3) INS(0) INS(1) INS(2) INS(3) : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(mm), "v"(cc), "s"(ss)); \
    a0 = p0.x; a1 = p0.y; a2 = p1.x; a3 = p1.y; a4 = p2.x; a5 = p2.y; a6 = p3.x; a7 = p3.y; BODY_END
#define P50(k) "v_pk_fma_f32 %" #k ", %" #k ", %4, %5\n\t"
PK_BODY(50, P50)
#define P51(k) "v_pk_fma_f32 %" #k ", %" #k ", %6, %5\n\t"
PK_BODY(51, P51)
#define P52(k) "v_pk_fma_f32 %" #k ", %4, %6, %" #k " op_sel_hi:[0,1,1]\n\t"
PK_BODY(52, P52)
#define P53(k) "v_pk_mul_f32 %" #k ", %" #k ", %6\n\t"
PK_BODY(53, P53)
#define P54(k) "v_pk_add_f32 %" #k ", %" #k ", %6\n\t"
PK_BODY(54, P54)
#define P55(k) "v_pk_mul_f32 %" #k ", %" #k ", %4\n\t"
PK_BODY(55, P55)

template <int ID>
__global__ __launch_bounds__(256) void rate(float* out, int iters, float m, float c, float sm)
{
    float a0 = 1.0f + threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    double d0 = a0, d1 = a1, d2 = a2, d3 = a3;
    unsigned u0 = threadIdx.x, u1 = threadIdx.x * 3 + 1;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int k = 0; k < 4; ++k) body<ID>(a0, a1, a2, a3, a4, a5, a6, a7, d0, d1, d2, d3, m, c, sm, u0, u1);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + (float)(d0 + d1 + d2 + d3) + (float)(u0 + u1);
}

static double g_clock_hz = 2.4e9;

template <int ID>
static void run(const char* name, int valu_per_group, int waves_per_simd)
{
    const int cus = 256, blocks = cus * waves_per_simd, iters = 4000;
    float* d; CK(hipMalloc(&d, (size_t)blocks * 256 * 4));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL(rate<ID>, dim3(blocks), dim3(256), 0, 0, d, 50, 0.999f, 1e-3f, 0.998f);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL(rate<ID>, dim3(blocks), dim3(256), 0, 0, d, iters, 0.999f, 1e-3f, 0.998f);
    CK(hipGetLastError());
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    CK(hipDeviceSynchronize());
    float ms; CK(hipEventElapsedTime(&ms, e0, e1));
    const double groups = (double)waves_per_simd * iters * 4 * 8;  // groups per SIMD
    const double cyc = ms * 1e-3 * g_clock_hz / groups;
    printf("%-58s %d w/SIMD  %7.2f cyc/group  (%d VALU: %.2f cyc/VALU)\n", name, waves_per_simd, cyc, valu_per_group, cyc / valu_per_group);
    hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
}

int main()
{
    hipDeviceProp_t p; CK(hipGetDeviceProperties(&p, 0));
    g_clock_hz = p.clockRate * 1e3;
    printf("device: %s (%s), %d CUs, clock %d kHz; one group = the listed instructions, 8 independent groups in flight per wave\n",
           p.name, p.gcnArchName, p.multiProcessorCount, p.clockRate);
    if (getenv("UBENCH_R02")) {  // only the round-2 additions
        for (int w : {5, 8}) {
            run<0>("v_fma_f32 v,v,v", 1, w);
            run<60>("v_fma_mix_f32 v(f16 lo),v,v", 1, w);
            run<61>("v_fma_mix_f32 v(f16 hi),v,v", 1, w);
            run<62>("v_cvt_f32_ubyte0", 1, w);
            run<63>("v_cvt_f32_ubyte2", 1, w);
            run<71>("v_cvt_f32_ubyte0 + v_fma_f32 v,v,v", 2, w);
            run<70>("v_cvt_f32_f16", 1, w);
            run<64>("v_max3_f32", 1, w);
            run<65>("v_min3_f32", 1, w);
            run<72>("v_min_f32", 1, w);
            run<73>("v_cmp_le_f32_e64 + v_addc_co_u32_e64", 2, w);
            run<66>("v_perm_b32", 1, w);
            run<67>("v_bfe_u32", 1, w);
            run<68>("v_and_or_b32", 1, w);
            run<74>("v_or_b32", 1, w);
            run<75>("v_lshrrev_b32", 1, w);
            run<69>("v_bcnt_u32_b32", 1, w);
        }
        return 0;
    }
    for (int w : {7, 8}) {
        run<0>("v_fma_f32 v,v,v", 1, w);
        run<18>("v_fma_f32 dependent chain", 1, w);
        run<1>("v_fma_f32 v,s,v", 1, w);
        run<2>("v_mul_f32 literal", 1, w);
        run<20>("v_max_f32", 1, w);
        run<3>("v_cmp_e32 vcc + s_nop 1 + v_cndmask_e32 vcc", 2, w);
        run<19>("v_cmp_u_e32 vcc + s_nop 1 + v_cndmask_e32 vcc", 2, w);
        run<4>("v_cmp_e64 s[] + s_nop 1 + v_cndmask_e64 s[]", 2, w);
        run<7>("v_cmp_e64 s[]", 1, w);
        run<5>("2 v_cmp_e64 + s_and + v_cndmask_e64 + v_lshl_or", 4, w);
        run<6>("2 v_cmp_e64 + s_and + v_addc_co_u32_e64", 3, w);
        run<17>("v_readlane -> s + s_nop 1 + v_fma v,s,v", 2, w);
        run<8>("v_rcp_f32", 1, w);
        run<21>("v_rsq_f32", 1, w);
        run<9>("v_sqrt_f32", 1, w);
        run<10>("v_add_u32", 1, w);
        run<12>("v_lshl_or_b32", 1, w);
        run<11>("v_mul_lo_u32", 1, w);
        run<22>("v_mul_hi_u32", 1, w);
        run<13>("v_fma_f64", 1, w);
        run<14>("v_mul_f64", 1, w);
        run<15>("v_add_f64", 1, w);
        run<23>("v_rcp_f64", 1, w);
        run<16>("v_cvt_f64_f32 / v_cvt_f32_f64 alternating", 1, w);

        run<30>("v_fmac_f32_e32 v,v,v", 1, w);
        run<31>("v_fmac_f32_e32 v,s,v", 1, w);
        run<32>("v_mul_f32_e32 v,s,v", 1, w);
        run<33>("v_sub_f32_e32 v,v,v", 1, w);
        run<34>("v_subrev_f32_e32 v,s,v", 1, w);
        run<35>("v_mul_f32_e64 v,v,-s", 1, w);
        run<36>("v_fma_f32 v,-v,v,v", 1, w);
        run<37>("v_add_f32_e32 v,1.0,v", 1, w);
        run<38>("v_mov_b32 v,v", 1, w);
        run<39>("v_mov_b32 v,s", 1, w);
        run<40>("v_cndmask_b32_e32 (vcc set earlier)", 1, w);
        run<41>("v_addc_co_u32_e64 (sgpr carry-in set earlier)", 1, w);
        run<42>("v_cmp_gt_f32_e64 s[], |v|, v", 1, w);
        run<43>("v_and_b32", 1, w);
        run<45>("v_xor_b32", 1, w);
        run<46>("v_lshlrev_b32", 1, w);
        run<44>("v_med3_f32", 1, w);
        run<47>("v_cvt_f32_u32", 1, w);
        run<48>("v_ffbh_u32", 1, w);
        run<50>("v_pk_fma_f32 v,v,v           (4 instr = 8 fma per group-of-8)", 1, w);
        run<51>("v_pk_fma_f32 v,s[2],v", 1, w);
        run<52>("v_pk_fma_f32 v(bcast lo),s[2],v op_sel_hi:[0,1,1]", 1, w);
        run<53>("v_pk_mul_f32 v,v,s[2]", 1, w);
        run<54>("v_pk_add_f32 v,v,s[2]", 1, w);
        run<55>("v_pk_mul_f32 v,v,v", 1, w);
    }
    return 0;
}
