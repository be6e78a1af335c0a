// Random 64-byte record gather, the LBVH search's memory pattern, in isolation (gfx950).
// Every lane follows its own dependent chain of records (the next index comes from the fetched data), as a
// lane of pt_trace_bvh_kernel follows its own ray:
//   mode 0: the lane fetches its record with 4 x global_load_dwordx4 (what the search does): each wave
//           instruction touches 64 different 64-byte lines
//   mode 1: quad-cooperative: instruction g fetches the records of lanes 16g..16g+15, four adjacent lanes reading
//           one record's four 16-byte pieces (16 lines per instruction); pieces go to their owner through LDS
//           (ds_write_b128 + 4 x ds_read_b128)
//   mode 2: as 0, but only ONE dwordx4 of the record is fetched (a 16-byte record: what the address path costs)
//   mode 3: as 0 with two independent chains per lane (memory-level parallelism x 2)
//   mode 4: the same table as 128-byte records, 8 x dwordx4 per lane (an 8-child node)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/ubench_gather tools/ubench_gather.hip
// run:   tools/ubench_gather [records (64-B each)] [steps] [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ __forceinline__ unsigned mix(unsigned x)
{
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}

template <int MODE>
__global__ __launch_bounds__(256) void gather_kernel(const uint4* __restrict__ tab, unsigned nrec, int steps, unsigned* out)
{
    __shared__ uint4 stage[4][4][64];  // [wave][group][lane]
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned idx = mix(blockIdx.x * 256u + threadIdx.x) % nrec;
    unsigned idx2 = mix(idx + 12345u) % nrec;
    unsigned acc = 0u;
    for (int s = 0; s < steps; ++s) {
        uint4 a, b, c, d;
        if (MODE == 0 || MODE == 3) {
            const uint4* p = tab + (size_t)idx * 4u;
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
            if (MODE == 3) {
                const uint4* q = tab + (size_t)idx2 * 4u;
                const uint4 a2 = q[0], b2 = q[1], c2 = q[2], d2 = q[3];
                const unsigned h2 = a2.x ^ b2.y ^ c2.z ^ d2.w;
                acc += h2;
                idx2 = mix(h2 + (unsigned)s) % nrec;
            }
        } else if (MODE == 4) {  // 128-byte records: 8 x dwordx4
            const uint4* p = tab + (size_t)(idx >> 1) * 8u;
            const uint4 e = p[4], f = p[5], g = p[6], hh = p[7];
            a = p[0]; b = p[1]; c = p[2]; d = p[3];
            a.x ^= e.x ^ f.y; b.y ^= g.z ^ hh.w;
        } else if (MODE == 2) {
            const uint4* p = tab + (size_t)idx * 4u;
            a = p[0]; b = a; c = a; d = a;
        } else {
            for (int g = 0; g < 4; ++g) {
                const unsigned src = 16u * g + (lane >> 2);
                const unsigned ni = (unsigned)__builtin_amdgcn_ds_bpermute((int)(src << 2), (int)idx);
                stage[wave][g][lane] = tab[(size_t)ni * 4u + (lane & 3u)];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint4* m = &stage[wave][lane >> 4][(lane & 15u) * 4u];
            a = m[0]; b = m[1]; c = m[2]; d = m[3];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        const unsigned h = a.x ^ b.y ^ c.z ^ d.w;
        acc += h;
        idx = mix(h + (unsigned)s) % nrec;
    }
    if (acc == 0x12345678u) out[0] = acc;
}

template <int MODE>
static void run(const uint4* tab, unsigned nrec, int steps, int blocks, unsigned* out, const char* what, int bytes_per_step)
{
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    gather_kernel<MODE><<<blocks, 256>>>(tab, nrec, steps / 8, out);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    gather_kernel<MODE><<<blocks, 256>>>(tab, nrec, steps, out);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double recs = (double)blocks * 256.0 * steps * (MODE == 3 ? 2 : 1);
    printf("  %-58s %8.2f ms  %7.2f G records/s  %6.2f TB/s\n", what, ms, recs / ms * 1e-6, recs * bytes_per_step / ms * 1e-9);
}

int main(int argc, char** argv)
{
    const unsigned nrec = argc > 1 ? (unsigned)atol(argv[1]) : 350000u;
    const int steps = argc > 2 ? atoi(argv[2]) : 2000;
    const int wps = argc > 3 ? atoi(argv[3]) : 6;
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    const int blocks = prop.multiProcessorCount * wps;  // 256 threads = one wave per SIMD
    std::vector<unsigned> h((size_t)nrec * 16u);
    unsigned x = 1u;
    for (auto& v : h) { x = x * 1664525u + 1013904223u; v = x; }
    uint4* tab; unsigned* out;
    CK(hipMalloc(&tab, (size_t)nrec * 64u)); CK(hipMalloc(&out, 64));
    CK(hipMemcpy(tab, h.data(), (size_t)nrec * 64u, hipMemcpyHostToDevice));
    printf("%u records of 64 B (%.1f MB), %d dependent steps per lane, %d waves per SIMD (%d blocks)\n", nrec, nrec * 64e-6, steps, wps, blocks);
    run<0>(tab, nrec, steps, blocks, out, "0: 4 x dwordx4 per lane (64 lines per instruction)", 64);
    run<1>(tab, nrec, steps, blocks, out, "1: quad-cooperative + LDS transpose (16 lines/instr)", 64);
    run<2>(tab, nrec, steps, blocks, out, "2: one dwordx4 per lane (16 B of the record)", 16);
    run<3>(tab, nrec, steps, blocks, out, "3: two chains per lane, 4 x dwordx4 each", 64);
    run<4>(tab, nrec, steps, blocks, out, "4: 128-byte records (same table), 8 x dwordx4 per lane", 128);
    return 0;
}
