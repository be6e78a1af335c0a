#include <math.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <pthread.h>
/* Exhaustive accuracy check of PTSPEC's binary32 sin/cos (oracle/pt_oracle.c ptor_sincos,
 * csrc/pt_device_math.h pt_sincos) against binary64 libm for EVERY binary32 angle in
 * [1e-19, 6.283186] (below 1e-19: sin x = x and cos x = 1 exactly).
 *   gcc -O2 -ffp-contract=off -mfma -Ioracle -o /tmp/check_sincos tools/check_sincos_f32.c -lm -lpthread && /tmp/check_sincos
 * -> max ulp err sin 1.4301 at 3.66085315, cos 1.4305 at 1.82281065 */
#include "ptor_constants.h"
#define TWO_OVER_PI_F PTOR_F32_TWO_OVER_PI
#define PIO2_A PTOR_F32_PIO2_A
#define PIO2_B PTOR_F32_PIO2_B
#define PIO2_C PTOR_F32_PIO2_C
#define PIO2_D PTOR_F32_PIO2_D
#define SS1 PTOR_F32_SIN_S1
#define SS2 PTOR_F32_SIN_S2
#define SS3 PTOR_F32_SIN_S3
#define SS4 PTOR_F32_SIN_S4
#define CC1 PTOR_F32_COS_C1
#define CC2 PTOR_F32_COS_C2
#define CC3 PTOR_F32_COS_C3
#define CC4 PTOR_F32_COS_C4
static inline void sc(float x, float* s, float* c)
{
    float kf = rintf(x * TWO_OVER_PI_F);
    float r = fmaf(-kf, PIO2_A, x);
    r = fmaf(-kf, PIO2_B, r);
    r = fmaf(-kf, PIO2_C, r);
    r = fmaf(-kf, PIO2_D, r);
    float r2 = r * r;
    float ps = SS4; ps = fmaf(ps, r2, SS3); ps = fmaf(ps, r2, SS2); ps = fmaf(ps, r2, SS1);
    float sn = fmaf(r * r2, ps, r);
    float pc = CC4; pc = fmaf(pc, r2, CC3); pc = fmaf(pc, r2, CC2); pc = fmaf(pc, r2, CC1);
    float cs = fmaf(r2, fmaf(r2, pc, -0.5f), 1.0f);
    int q = (int)kf & 3;
    *s = q == 0 ? sn : q == 1 ? cs : q == 2 ? -sn : -cs;
    *c = q == 0 ? cs : q == 1 ? -sn : q == 2 ? -cs : sn;
}
static double ulp_err(float got, double want)
{
    float w = (float)want; int e; frexpf(w == 0 ? 1e-45f : w, &e);
    double ulp = ldexp(1.0, e - 24); if (ulp < 1.4e-45) ulp = 1.4e-45;
    return fabs((double)got - want) / ulp;
}
typedef struct { uint32_t lo, hi; double ms, mc; uint32_t as, ac; } job;
static void* run(void* p)
{
    job* j = p; j->ms = j->mc = 0;
    for (uint32_t b = j->lo; b < j->hi; ++b) {
        float x; memcpy(&x, &b, 4); float s, c; sc(x, &s, &c);
        double es = ulp_err(s, sin((double)x)), ec = ulp_err(c, cos((double)x));
        if (es > j->ms) { j->ms = es; j->as = b; }
        if (ec > j->mc) { j->mc = ec; j->ac = b; }
    }
    return 0;
}
int main(void)
{
    float top = 6.2831860f; uint32_t tb; memcpy(&tb, &top, 4);
    uint32_t start = 0x20000000;  /* 1e-19: below, sin x = x exactly and cos = 1 */
    enum { T = 8 }; pthread_t th[T]; job js[T];
    for (int t = 0; t < T; ++t) { js[t].lo = start + (uint64_t)(tb - start) * t / T; js[t].hi = start + (uint64_t)(tb - start) * (t + 1) / T; pthread_create(&th[t], 0, run, &js[t]); }
    double ms = 0, mc = 0; uint32_t as = 0, ac = 0;
    for (int t = 0; t < T; ++t) { pthread_join(th[t], 0); if (js[t].ms > ms) { ms = js[t].ms; as = js[t].as; } if (js[t].mc > mc) { mc = js[t].mc; ac = js[t].ac; } }
    float xs, xc; memcpy(&xs, &as, 4); memcpy(&xc, &ac, 4);
    printf("max ulp err sin %.4f at %.9g, cos %.4f at %.9g\n", ms, xs, mc, xc);
    return 0;
}
