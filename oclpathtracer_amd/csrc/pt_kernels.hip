// pt_kernels.hip -- hand-written gfx950 kernels of the path-tracing hot path.
//
// What the reference runs as ONE OpenCL mega-kernel per frame
// (test/ClKernels/GenerateColors.cl:302-322: one work-item = one pixel = one path, then a
// read-modify-write of the gamma-encoded running mean) is restructured for CDNA4 as
//
//   pt_prep_kernel   once per scene upload: Triangle -> {p1, e1, e2, n, id}
//   pt_trace_kernel  persistent waves; every LANE owns one path at a time and, when its path
//                    ends, immediately starts the next (pixel, frame) sample taken from a
//                    wave-local range of a global batch queue (ballot + mbcnt compaction, no
//                    LDS).  The 36-triangle closest-hit loop therefore always runs with a full
//                    exec mask.  Triangle records are wave-uniform: one s_load_dwordx16 each,
//                    consumed as SGPR operands of the VALU ops (no VGPR/LDS/vector-memory cost).
//                    Path radiance goes to a staging array rad[frame][pixel].
//   pt_fold_kernel   per pixel, in ascending frame order, replays the reference's
//                    gamma -> mean -> degamma arithmetic (GenerateColors.cl:314-321) over the
//                    staged radiances: bit-identical to frame-by-frame launches.
//
// The arithmetic is PTSPEC (pt_device_math.h); results are bit-identical to oracle/pt_oracle.c.
#include "pt_kernels.h"

#include "pt_device_math.h"

typedef const __attribute__((address_space(4))) float* pt_const_f32p;  // scalar (SMEM) loads

// ------------------------------------------------------------------------------------------
// scene preparation
// ------------------------------------------------------------------------------------------
__global__ void pt_prep_kernel(const PtRawTriangle* __restrict__ raw, PtPrepTriangle* __restrict__ out, int ntri)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= ntri) return;
    f3 p1 = mk3(raw[i].p1[0], raw[i].p1[1], raw[i].p1[2]);
    f3 p2 = mk3(raw[i].p2[0], raw[i].p2[1], raw[i].p2[2]);
    f3 p3 = mk3(raw[i].p3[0], raw[i].p3[1], raw[i].p3[2]);
    f3 e1 = sub3(p2, p1);          // GenerateColors.cl:92
    f3 e2 = sub3(p3, p1);          // :93
    f3 n = cross3(e2, e1);         // :123
    PtPrepTriangle t;
    t.p1[0] = p1.x; t.p1[1] = p1.y; t.p1[2] = p1.z;
    t.e1[0] = e1.x; t.e1[1] = e1.y; t.e1[2] = e1.z;
    t.e2[0] = e2.x; t.e2[1] = e2.y; t.e2[2] = e2.z;
    t.pad0[0] = t.pad0[1] = t.pad0[2] = 0.0f;
    t.n[0] = n.x; t.n[1] = n.y; t.n[2] = n.z;
    t.id = raw[i].id;
    out[i] = t;
}

// ------------------------------------------------------------------------------------------
// camera: GenerateColors.cl:73-87, 263-288
// ------------------------------------------------------------------------------------------
PTK_DEV void pt_generate_ray(int xc, int yc, int width, int height, uint32_t& seed, f3& org, f3& dir_out)
{
    float invWidth = 1.0f / (float)width, invHeight = 1.0f / (float)height;
    float aspectratio = (float)width / (float)height;
    float angle = PTK_TAN_HALF_FOV;

    const f3 eye = mk3(0.0f, 2.75f, 4.0f);
    const f3 center = add3(eye, mk3(0.0f, 0.0f, -1.0f));
    const f3 up = mk3(0.0f, 1.0f, 0.0f);
    const f3 viewDir = normalize3(sub3(center, eye));
    const f3 holDir = normalize3(cross3(viewDir, up));
    const f3 upDir = normalize3(cross3(holDir, viewDir));

    float x = (float)xc + pt_random_float(seed) - 0.5f;
    float y = (float)yc + pt_random_float(seed) - 0.5f;
    x = (2.0f * ((x + 0.5f) * invWidth) - 1.0f) * angle * aspectratio;
    y = -(1.0f - 2.0f * ((y + 0.5f) * invHeight)) * angle;

    float my = -1.0f * y;
    f3 d = add3(add3(scale3(holDir, x), scale3(upDir, my)), viewDir);
    f3 dir = normalize3(d);
    f3 pointAimed = add3(eye, scale3(dir, 4.0f));
    org = eye;
    dir_out = normalize3(normalize3(sub3(pointAimed, eye)));  // :287 then getRay's own normalize (:75)
}

// ------------------------------------------------------------------------------------------
// trace kernel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(PT_TRACE_THREADS) void pt_trace_kernel(const PtTraceParams P)
{
    const unsigned lane = __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    pt_const_f32p T = (pt_const_f32p)(const float*)P.tris;
    const int ntri = P.ntri;

    // wave-uniform work range (kept in SGPRs)
    unsigned q_pix = 0, q_end = 0, q_frame = 0;
    bool exhausted = false;

    // per-lane path state
    bool alive = false;
    f3 o = mk3(0.0f, 0.0f, 0.0f), d = mk3(0.0f, 0.0f, 1.0f);
    f3 mask = mk3(1.0f, 1.0f, 1.0f), L = mk3(0.0f, 0.0f, 0.0f);
    uint32_t seed = 0;
    int bounce = 0;
    unsigned lp = 0, fl = 0;
    unsigned n_rays = 0, n_samples = 0;

    for (;;) {
        // ---- regeneration: dead lanes take the next samples of the wave's range ----------
        unsigned long long need = __ballot(!alive);
        while (need != 0ull && !exhausted) {
            if (q_pix == q_end) {
                unsigned b = 0;
                if (lane == 0) b = atomicAdd(P.batch_counter, 1u);
                b = __builtin_amdgcn_readfirstlane(b);
                if (b >= P.total_batches) { exhausted = true; break; }
                unsigned f = b / P.batches_per_frame;
                unsigned bi = b - f * P.batches_per_frame;
                q_frame = f;
                q_pix = bi * PT_TRACE_BATCH;
                unsigned e = q_pix + PT_TRACE_BATCH;
                q_end = e < P.npix_local ? e : P.npix_local;
            }
            unsigned n_need = (unsigned)__popcll(need);
            unsigned avail = q_end - q_pix;
            unsigned take = n_need < avail ? n_need : avail;
            unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(need >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)need, 0u));
            if (!alive && rank < take) {
                lp = q_pix + rank;
                fl = q_frame;
                // local pixel -> global pixel id (image rows dealt to ranks in stripes)
                unsigned lr = lp / (unsigned)P.width;
                unsigned x = lp - lr * (unsigned)P.width;
                unsigned sl = lr / (unsigned)P.stripe_rows;
                unsigned within = lr - sl * (unsigned)P.stripe_rows;
                unsigned grow = (sl * (unsigned)P.n_ranks + (unsigned)P.rank) * (unsigned)P.stripe_rows + within;
                unsigned gid = grow * (unsigned)P.width + x;
                int frame = P.frame_begin + (int)fl;
                seed = gid + pt_hash_u32((uint32_t)frame);                       // :308
                pt_generate_ray((int)x, (int)grow, P.width, P.height, seed, o, d);  // :310
                mask = mk3(1.0f, 1.0f, 1.0f);
                L = mk3(0.0f, 0.0f, 0.0f);
                bounce = 0;
                alive = true;
            }
            q_pix += take;
            need = __ballot(!alive);
        }
        if (__ballot(alive) == 0ull) break;

        // ---- intersectWorld (:137-154): every lane, wave-uniform triangle index -------------
        float tmax = 1e20f, hu = 0.0f, hv = 0.0f;
        int hidx = -1;
        for (int i = 0; i < ntri; ++i) {
            pt_const_f32p t = T + 16 * i;
            const float p1x = t[0], p1y = t[1], p1z = t[2];
            const float e1x = t[3], e1y = t[4], e1z = t[5];
            const float e2x = t[6], e2y = t[7], e2z = t[8];
            // pvec = cross(dir, e2); det = dot(e1, pvec)   (:96-97)
            float pvx = pt_fma(d.y, e2z, -(d.z * e2y));
            float pvy = pt_fma(d.z, e2x, -(d.x * e2z));
            float pvz = pt_fma(d.x, e2y, -(d.y * e2x));
            float det = pt_fma(e1z, pvz, pt_fma(e1y, pvy, e1x * pvx));
            if (det < 1e-8f || -det > 1e-8f) continue;  // :100
            float inv_det = 1.0f / det;
            float tvx = o.x - p1x, tvy = o.y - p1y, tvz = o.z - p1z;
            float u = pt_fma(tvz, pvz, pt_fma(tvy, pvy, tvx * pvx)) * inv_det;
            if (u < 0.0f || u > 1.0f) continue;  // :109
            float qvx = pt_fma(tvy, e1z, -(tvz * e1y));
            float qvy = pt_fma(tvz, e1x, -(tvx * e1z));
            float qvz = pt_fma(tvx, e1y, -(tvy * e1x));
            float v = pt_fma(d.z, qvz, pt_fma(d.y, qvy, d.x * qvx)) * inv_det;
            if (v < 0.0f || u + v > 1.0f) continue;  // :117
            float tt = pt_fma(e2z, qvz, pt_fma(e2y, qvy, e2x * qvx)) * inv_det;
            if (tt > 0.0f && tt < tmax) {  // :125
                tmax = tt; hu = u; hv = v; hidx = i;
            }
        }

        // ---- shade (:229-258) --------------------------------------------------------------
        if (alive) {
            bool finished = false;
            n_rays++;
            if (hidx < 0) {
                const float bg = pt_max(0.45f, 0.0f);
                L = add3(L, scale3(mask, bg));  // :235
                finished = true;
            } else {
                // deferred HitRecord of the closest hit (:127-130): same values as writing it
                // at every acceptance, only the last one is read.
                const float4 nid = *reinterpret_cast<const float4*>(reinterpret_cast<const float*>(P.tris + hidx) + 12);
                const f3 N = mk3(nid.x, nid.y, nid.z);
                int mid = __float_as_int(nid.w);
                mid = mid < 0 ? 0 : (mid >= P.nmat ? P.nmat - 1 : mid);  // never fault on a corrupt id
                f3 p = add3(o, scale3(d, tmax));
                float w = 1.0f - hu - hv;
                f3 n = normalize3(add3(add3(scale3(N, hu), scale3(N, hv)), scale3(N, w)));

                const PtRawMaterial* mat = P.mats + mid;  // :239
                const float4 alb = *reinterpret_cast<const float4*>(mat->albedo);
                const float4 emi = *reinterpret_cast<const float4*>(mat->emissive);
                const float rough = mat->roughness;
                const int type = mat->type;

                L.x = L.x + mask.x * emi.x * 3.0f;  // :241
                L.y = L.y + mask.y * emi.y * 3.0f;
                L.z = L.z + mask.z * emi.z * 3.0f;

                n = dot3(n, d) < 0.0f ? n : scale3(n, -1.0f);  // :243
                f3 wo = neg3(d);

                // sampleHemisphereCosine (:161-172) and sampleGGX (:180-192) share everything
                // except (sinTheta, cosTheta); both draw phi first, then the second uniform.
                float phi = PTK_TWO_PI * pt_random_float(seed);
                float xi = pt_random_float(seed);
                f3 axis = __builtin_fabsf(n.x) > 0.001f ? mk3(0.0f, 1.0f, 0.0f) : mk3(1.0f, 0.0f, 0.0f);
                f3 tv = normalize3(cross3(axis, n));
                f3 sv = cross3(n, tv);
                float sp, cp;
                pt_sincos(phi, sp, cp);
                float sinTheta, cosTheta;
                if (type == 2) {
                    cosTheta = __builtin_sqrtf((1.0f - xi) / (xi * (rough * rough - 1.0f) + 1.0f));
                    sinTheta = __builtin_sqrtf(pt_max(0.0f, 1.0f - cosTheta * cosTheta));
                } else {
                    sinTheta = __builtin_sqrtf(xi);
                    cosTheta = __builtin_sqrtf(1.0f - xi);
                }
                f3 a = scale3(scale3(sv, cp), sinTheta);
                f3 b = scale3(scale3(tv, sp), sinTheta);
                f3 c = scale3(n, cosTheta);
                f3 sdir = normalize3(add3(add3(a, b), c));

                f3 wi = sdir;
                f3 color = mk3(0.0f, 0.0f, 0.0f);
                float pdf = 0.0f;
                float dwin = 0.0f;
                if (type == 1) {  // DIFFUSE (:197-204)
                    dwin = dot3(wi, n);
                    pdf = dwin * PTK_INV_PI;
                    color = mk3(alb.x * PTK_INV_PI, alb.y * PTK_INV_PI, alb.z * PTK_INV_PI);
                } else if (type == 2) {  // SPECULAR (:205-218)
                    float k2 = 2.0f * dot3(wo, sdir);
                    wi = add3(neg3(wo), scale3(sdir, k2));  // reflect(wo, wh) (:156-159)
                    dwin = dot3(wi, n);
                    float dwon = dot3(wo, n);
                    if (!(dwin * dwon < 0.0f)) {
                        float r2 = rough * rough;
                        float D = r2 * PTK_INV_PI / pt_pow(cosTheta * cosTheta * (r2 - 1.0f) + 1.0f, 2.0f);
                        pdf = D * cosTheta / (4.0f * dot3(wo, sdir));
                        float g = D / (4.0f * dwin * dwon);
                        color = mk3(alb.x * g * 2.0f, alb.y * g * 2.0f, alb.z * g * 2.0f);
                    }
                }
                if (pdf <= 0.0f) {  // :251
                    finished = true;
                } else {
                    mask.x = mask.x * (color.x * dwin / pdf);  // :253-255
                    mask.y = mask.y * (color.y * dwin / pdf);
                    mask.z = mask.z * (color.z * dwin / pdf);
                    bounce++;
                    if (bounce >= P.max_bounces) {
                        finished = true;
                    } else {
                        o = add3(p, scale3(wi, 0.01f));  // :257
                        d = normalize3(wi);
                    }
                }
            }
            if (finished) {
                float4 out;
                out.x = pt_max(L.x, 0.0f);  // :260
                out.y = pt_max(L.y, 0.0f);
                out.z = pt_max(L.z, 0.0f);
                out.w = 1.0f;
                P.rad[(size_t)fl * P.npix_local + lp] = out;
                n_samples++;
                alive = false;
            }
        }
    }

    if (P.stats) {
        // wave reduction of the work counters, one atomic pair per wave
        unsigned long long r = n_rays, s = n_samples;
        for (int off = 32; off > 0; off >>= 1) {
            r += __shfl_down(r, off);
            s += __shfl_down(s, off);
        }
        if (lane == 0) {
            atomicAdd(&P.stats[0], s);
            atomicAdd(&P.stats[1], r);
        }
    }
}

// ------------------------------------------------------------------------------------------
// fold kernel: GenerateColors.cl:290-300, 314-321, frames in ascending order per pixel
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pt_fold_kernel(const PtFoldParams P)
{
    unsigned lp = blockIdx.x * blockDim.x + threadIdx.x;
    if (lp >= P.npix_local) return;
    const float inv_gamma = 1.0f / PTK_GAMMA;
    float mx = 0.0f, my = 0.0f, mz = 0.0f;
    int z = P.frame_begin;
    if (z != 0) {
        float4 cur = P.fb[lp];
        mx = cur.x; my = cur.y; mz = cur.z;
    }
    for (int f = 0; f < P.frame_count; ++f, ++z) {
        float4 c = P.rad[(size_t)f * P.npix_local + lp];
        if (z == 0) {
            mx = pt_pow(c.x, inv_gamma);
            my = pt_pow(c.y, inv_gamma);
            mz = pt_pow(c.z, inv_gamma);
        } else {
            float zm1 = (float)(z - 1), zf = (float)z;
            float ox = pt_pow(mx, PTK_GAMMA), oy = pt_pow(my, PTK_GAMMA), oz = pt_pow(mz, PTK_GAMMA);
            mx = pt_pow((ox * zm1 + c.x) / zf, inv_gamma);
            my = pt_pow((oy * zm1 + c.y) / zf, inv_gamma);
            mz = pt_pow((oz * zm1 + c.z) / zf, inv_gamma);
        }
    }
    if (P.frame_count > 0) P.fb[lp] = make_float4(mx, my, mz, 1.0f);
}

// ------------------------------------------------------------------------------------------
// multi-GPU assembly, output stage, shim smoke-test kernel
// ------------------------------------------------------------------------------------------
__global__ void pt_assemble_kernel(const float4* __restrict__ gathered, float4* __restrict__ image, int width,
                                   int height, int stripe_rows, int n_ranks, int slab_rows)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    size_t total = (size_t)width * height;
    if (i >= total) return;
    unsigned row = (unsigned)(i / (unsigned)width);
    unsigned x = (unsigned)(i - (size_t)row * width);
    unsigned stripe = row / (unsigned)stripe_rows;
    unsigned within = row - stripe * (unsigned)stripe_rows;
    unsigned rank = stripe % (unsigned)n_ranks;
    unsigned sl = stripe / (unsigned)n_ranks;
    size_t src = ((size_t)rank * slab_rows + (size_t)sl * stripe_rows + within) * width + x;
    image[i] = gathered[src];
}

// f2c(sqrtf(v)) of test/RaytraceTest.cpp:78-83,280-285: a *= 255; min((int)a, 255)
PTK_DEV int32_t pt_f2c(float v)
{
    float a = __builtin_sqrtf(v) * 255.0f;
    int32_t i;
    if (a != a) i = (int32_t)0x80000000;        // (int)NaN on the reference's x86 host
    else if (a >= 2147483648.0f) i = (int32_t)0x80000000;  // cvttss2si overflow value
    else if (a <= -2147483648.0f) i = (int32_t)0x80000000;
    else i = (int32_t)a;
    return i < 255 ? i : 255;
}

__global__ void pt_tonemap_kernel(const float4* __restrict__ fb, int32_t* __restrict__ rgb, size_t npix)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npix) return;
    float4 v = fb[i];
    rgb[3 * i + 0] = pt_f2c(v.x);
    rgb[3 * i + 1] = pt_f2c(v.y);
    rgb[3 * i + 2] = pt_f2c(v.z);
}

__global__ void pt_fill_i32_kernel(int32_t* dst, int32_t value, int n)
{
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = value;
}

// ------------------------------------------------------------------------------------------
// host launchers
// ------------------------------------------------------------------------------------------
hipError_t ptk_prep_triangles(const PtRawTriangle* raw, PtPrepTriangle* out, int ntri, hipStream_t s)
{
    if (ntri <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_prep_kernel, dim3((ntri + 255) / 256), dim3(256), 0, s, raw, out, ntri);
    return hipGetLastError();
}

hipError_t ptk_trace(const PtTraceParams& p, int num_blocks, hipStream_t s)
{
    hipLaunchKernelGGL(pt_trace_kernel, dim3(num_blocks), dim3(PT_TRACE_THREADS), 0, s, p);
    return hipGetLastError();
}

hipError_t ptk_fold(const PtFoldParams& p, hipStream_t s)
{
    if (p.npix_local == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fold_kernel, dim3((p.npix_local + 255) / 256), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t ptk_assemble_stripes(const float4* gathered, float4* image, int width, int height, int stripe_rows,
                                int n_ranks, int slab_rows, hipStream_t s)
{
    size_t total = (size_t)width * height;
    if (total == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_assemble_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, gathered, image,
                       width, height, stripe_rows, n_ranks, slab_rows);
    return hipGetLastError();
}

hipError_t ptk_tonemap_ppm(const float4* fb, int32_t* rgb, size_t npix, hipStream_t s)
{
    if (npix == 0) return hipSuccess;
    hipLaunchKernelGGL(pt_tonemap_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, s, fb, rgb, npix);
    return hipGetLastError();
}

hipError_t ptk_fill_i32(int32_t* dst, int32_t value, int n, hipStream_t s)
{
    if (n <= 0) return hipSuccess;
    hipLaunchKernelGGL(pt_fill_i32_kernel, dim3((n + 255) / 256), dim3(256), 0, s, dst, value, n);
    return hipGetLastError();
}

int ptk_trace_blocks_per_cu(void)
{
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, pt_trace_kernel, PT_TRACE_THREADS, 0) != hipSuccess || nb < 1)
        nb = 2;
    return nb;
}
