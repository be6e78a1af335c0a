// pt_shim.hip -- implementation of the C ABI in include/pt_shim.h on the HIP runtime.
//
// Replaces, for the path-tracing hot path, what the reference does through
// Adl/CL/AdlCL.{inl,cpp} + Adl/CL/AdlKernelUtilsCL.{inl,cpp} + Adl/AdlKernel.cpp
// (OpenCL context/queue, cl_mem buffers, map/unmap, source->binary kernel cache,
// clSetKernelArg + clEnqueueNDRangeKernel).  There is no CPU path in this file.
#include "../../include/pt_shim.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "pt_kernels.h"

// ------------------------------------------------------------------------------------------
// error plumbing
// ------------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

static int fail(int code, const char* fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess) return fail(PT_ERR_HIP, "%s failed: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

extern "C" const char* pt_last_error(void) { return g_err; }
extern "C" int pt_abi_version(void) { return PT_SHIM_ABI_VERSION; }

// ------------------------------------------------------------------------------------------
// objects
// ------------------------------------------------------------------------------------------
enum { KERNEL_GENERATE_COLORS = 0, KERNEL_FILL = 1, KERNEL_MATH = 2, KERNEL_FOLD_CHECK = 3, KERNEL_COUNT = 4 };

struct pt_kernel_s {
    int id;
    const char* file;
    const char* func;
};

struct pt_buffer_s {
    pt_device_s* dev;
    void* dptr;
    size_t bytes;
    bool owned;
    uint64_t version;   // bumped by every API call that may change the contents
    void* staging;      // pinned host range of the last map (kept until the buffer dies)
    size_t staging_bytes;
    bool mapped;
    size_t mapped_bytes;  // size of the range handed out by the last pt_buffer_map (what unmap copies back)
    bool exposed;       // the raw device pointer has been handed out (pt_buffer_device_ptr): the caller may
                        // read or write the memory behind the shim's back, like wrapped memory
};

struct pt_event_s {
    pt_device_s* dev;
    hipEvent_t start, stop;
    bool recorded;
};

struct PendingFrames {  // deferred GenerateColors launches (PT_OPT_BATCH_FRAMES)
    bool active;
    pt_buffer_s *tris, *mats, *fb;
    int width, height, z_begin, z_count;
    uint32_t pixel_count;
};

struct pt_device_s {
    int idx;
    hipDeviceProp_t prop;
    hipStream_t own_stream, stream;
    bool external;           // `stream` is the caller's (pt_device_set_stream)
    uint64_t used, peak;
    int live_buffers;
    int64_t opt_batch, opt_chunk, opt_profile, opt_quads, opt_accel, opt_tally, opt_pmask, opt_bvh_stack, opt_lanes, opt_carry;
    pt_kernel_s kernels[KERNEL_COUNT];
    // prepared-scene cache
    PtPrepTriangle* prep;
    size_t prep_capacity;  // triangles
    const pt_buffer_s* prep_src;
    uint64_t prep_version;
    int prep_ntri;
    uint64_t prep_hash;         // checksum of the raw records the prepared scene was made from (pt_prep_kernel)
    bool prep_hash_valid;
    uint64_t scene_gen;         // bumped whenever the prepared scene's CONTENTS changed
    uint64_t bvh_builds;        // LBVH builds so far (PT_OPT_BVH_BUILD_COUNT)
    bool prep_det_bounded;      // scene extent allows the short exact reciprocal
    int prep_quads;             // 0: independent triangles; 3: every pair (2k, 2k+1) is a quad (a,b,c),(c,d,a),
                                // finite radius, margins and the packed table prepared
    float prep_delta1, prep_ray_radius;  // quad modes 2, 3 (pt_quad2_pass1)
    float prep_radius;                   // max |vertex - eye|_inf of the prepared scene (NaN when not finite)
    int big_quads;                       // the LBVH's table of big triangles: 3 = made of quads, its filter table (big_p1tab) and bounds prepared
    float* big_p1tab;
    float big_delta1, big_ray_radius, big_p1_lo, big_p1_hi;
    PtBvh8Node* bvh;             // LBVH of the prepared scene (built on demand: ensure_bvh): its 64-byte records, sized with prep
    PtBvhGrid bvh_grid;         // the grid of its nodes' origins
    size_t bvh_records;         // records in use (nodes + leaves)
    PtPrepTriangle* bigtab;     // the triangles kept out of the hierarchy (PT_BVH_BIG_MAX records + indices + count)
    int* bigidx;
    int nbig;
    int bvh_blocks_per_cu;
    bool bvh_valid;
    float* p1tab;               // quad mode 3: packed pass-1 table (pt_quad3_pass1), sized with prep
    float prep_p1_lo, prep_p1_hi;
    unsigned int* det_bound_dev;  // PT_PREP_WORDS device words written by the prep kernel
    // fused-render workspace: the STREAMING renderer.  A render walks its frames in chunks of S frames (as many as a ring slot
    // holds) through a ring of two radiance slots; chunk number seq (a running number over all renders of the handle) uses slot
    // seq % 2.  All launches of ONE render go to one lane, in order:
    //     T(0)  T(1) F(0)  T(2) F(1)  ...  T(n-1) F(n-2)  D  F(n-1)
    // T(c) = trace launch of chunk c, CHECKPOINTED (PtTraceParams::carry): it ends the moment its queue has handed out the last
    // batch, every wave saving the paths it still holds, and T(c+1) resumes them beside its own samples -- a launch has no tail
    // of waves running out of paths, so chunks as short as a 192 MiB slot forces cost 5 % over one long launch (the waves stop over ~100 us, and a fold follows: DESIGN.md S6).  F(c), the
    // fold of chunk c into the framebuffer, therefore follows T(c+1), which finishes chunk c's last paths; D is a launch with an
    // empty queue that finishes the last chunk's.  (The LBVH kernel's checkpoint is one SEARCH deep: a stopping launch starts no new
    // search, the lanes still searching finish theirs and are shaded, and the paths between two searches go to the next launch.
    // With PT_OPT_CHECKPOINT 0 every launch runs its paths out and the CHUNKS alternate lanes: T(0) F(0) T(1) F(1) ...)
    // Consecutive renders ALTERNATE between the two lanes: render k+1's T(0) needs the slot that F(n-2) of render k released, not
    // the one F(n-1) is still to read, so it fills the machine while D of render k runs dry.  The folds of all chunks of all renders
    // form ONE chain (events): every pixel folds its frames in ascending order (GenerateColors.cl:314-321).
    float* ring;             // PT_RING_SLOTS x ring_slot_bytes: 12 bytes per (frame, pixel) of a chunk
    size_t ring_slot_bytes;
    hipStream_t lane[2];
    hipEvent_t ev_fork;      // recorded on `stream`: what the lanes must wait for before a render's first launches
    hipEvent_t ev_fold[2];   // recorded on lane L behind its latest fold
    hipEvent_t ev_slot[2];   // recorded behind the latest fold that read ring slot s: the slot may be written again
    hipEvent_t ev_ext;       // pt_device_wait_stream
    bool fold_recorded[2], slot_recorded[2];
    int last_fold_lane;      // the lane of the newest fold (its event is behind every earlier launch of both lanes)
    bool lanes_busy;         // lane work is enqueued that `stream` has not been ordered behind
    bool main_dirty;         // work has been enqueued on `stream` that the lanes have not been ordered behind
    uint64_t chunk_seq, render_seq;
    uint32_t* carry[2];      // per lane: the checkpoint regions of its renders' trace launches (PT_CARRY_STRIDE_DW dwords per wave)
    size_t carry_waves;      // waves each of them holds
    uint2* pmask;            // primary-ray candidate masks of the local pixels (pt_primary_mask_kernel)
    size_t pmask_pixels;
    struct { uint64_t scene_gen; int32_t g[7]; bool valid; } pmask_key;  // what the masks in hand were made for
    unsigned int* counters;  // PT_QUEUE_COUNTERS work-queue counters, PT_QUEUE_STRIDE words apart: zero between launches
    bool counters_dirty;     // a failed call may have left one non-zero
    unsigned int* trav_host; // the LBVH's sticky "search cut short" words: host memory the kernels store to (PT_ERR_TRAVERSAL)
    unsigned int* trav_dev;  // ... as the device addresses it
    uint64_t workspace;      // device bytes held by this handle for itself
    int blocks_per_cu;
    PendingFrames pending;
    // per-kernel timing (pt_profile_*)
    bool prof_on;
    std::vector<std::pair<hipEvent_t, hipEvent_t>>* prof_pairs;  // [PT_PROF_KINDS]
    size_t prof_used[PT_PROF_KINDS];
};

#define PT_RING_SLOTS 2
#define PT_DEFAULT_SLOT_BYTES ((size_t)192 << 20)   // sixteen 1024 x 1024 frames of 12-byte radiance
#define PT_QUEUE_COUNTERS 64                        // (the last two are the lanes' draining launches')
#define PT_QUEUE_STRIDE PT_QUEUE_WORDS              // words per queue: sharded counters + stop word (pt_kernels.h)

static int prof_begin(pt_device_s* d, int kind, hipStream_t st, hipEvent_t* stop_out)
{
    *stop_out = nullptr;
    if (!d->prof_on) return PT_OK;
    auto& v = d->prof_pairs[kind];
    if (d->prof_used[kind] == v.size()) {
        hipEvent_t a, b;
        HIP_TRY(hipEventCreate(&a));
        HIP_TRY(hipEventCreate(&b));
        v.push_back({ a, b });
    }
    auto& pr = v[d->prof_used[kind]++];
    HIP_TRY(hipEventRecord(pr.first, st));
    *stop_out = pr.second;
    return PT_OK;
}

static int prof_end(hipStream_t st, hipEvent_t stop)
{
    if (stop) HIP_TRY(hipEventRecord(stop, st));
    return PT_OK;
}

#ifndef PT_DEFAULT_PRIMARY_MASKS
#define PT_DEFAULT_PRIMARY_MASKS 1  // PT_OPT_PRIMARY_MASKS of a new device handle (A/B builds set 0)
#endif

static int g_init_count = 0;

// device memory the handle holds for itself (pt_device_workspace_memory)
static hipError_t ws_malloc(pt_device_s* d, void** p, size_t bytes)
{
    hipError_t e = hipMalloc(p, bytes);
    if (e == hipSuccess) d->workspace += bytes;
    else { *p = nullptr; (void)hipGetLastError(); }
    return e;
}
template <class T> static hipError_t ws_malloc(pt_device_s* d, T** p, size_t bytes) { return ws_malloc(d, reinterpret_cast<void**>(p), bytes); }

static void ws_free(pt_device_s* d, void* p, size_t bytes)
{
    if (!p) return;
    hipFree(p);
    d->workspace -= bytes;
}

static const size_t WS_BIGTAB = PT_BVH_BIG_MAX * sizeof(PtPrepTriangle);
static const size_t WS_BIGIDX = (PT_BVH_BIG_MAX + 1) * sizeof(int) + sizeof(PtBvhGrid) + sizeof(unsigned);  // indices, count, the LBVH's grid, records in use
static const size_t WS_COUNTERS = (size_t)PT_QUEUE_COUNTERS * PT_QUEUE_STRIDE * sizeof(unsigned int);
static const size_t WS_DETBOUND = PT_PREP_WORDS * sizeof(unsigned int);
static size_t ws_big_p1tab() { return ptk_p1tab_floats(PT_BVH_BIG_MAX) * sizeof(float) + PT_BVH_BIG_MAX * sizeof(PtRawTriangle); }  // + the big triangles' raw records

static int use_device(pt_device_s* d)
{
    if (!d) return fail(PT_ERR_INVALID, "null device handle");
    HIP_TRY(hipSetDevice(d->idx));
    return PT_OK;
}

// ------------------------------------------------------------------------------------------
// library / device lifetime
// ------------------------------------------------------------------------------------------
extern "C" int pt_init(void)
{
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(PT_ERR_NO_DEVICE, "hipGetDeviceCount: %s", hipGetErrorString(e));
    if (n <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible");
    ++g_init_count;
    return PT_OK;
}

extern "C" void pt_quit(void)
{
    if (g_init_count > 0) --g_init_count;
}

extern "C" int pt_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// everything a device handle owns (also the unwinding path of a pt_device_create that failed half-way)
static void destroy_device_objects(pt_device_s* d)
{
    ws_free(d, d->prep, d->prep_capacity * sizeof(PtPrepTriangle));
    ws_free(d, d->p1tab, ptk_p1tab_floats((int)d->prep_capacity) * sizeof(float));
    ws_free(d, d->bvh, ptk_bvh_record_count((int)d->prep_capacity) * sizeof(PtBvh8Node));
    ws_free(d, d->ring, PT_RING_SLOTS * d->ring_slot_bytes);
    ws_free(d, d->pmask, d->pmask_pixels * sizeof(uint2));
    ws_free(d, d->counters, WS_COUNTERS);
    ws_free(d, d->bigtab, WS_BIGTAB);
    ws_free(d, d->bigidx, WS_BIGIDX);
    ws_free(d, d->big_p1tab, ws_big_p1tab());
    ws_free(d, d->det_bound_dev, WS_DETBOUND);
    if (d->trav_host) hipHostFree(d->trav_host);
    if (d->prof_pairs) {
        for (int k = 0; k < PT_PROF_KINDS; ++k)
            for (auto& pr : d->prof_pairs[k]) {
                hipEventDestroy(pr.first);
                hipEventDestroy(pr.second);
            }
        delete[] d->prof_pairs;
    }
    for (int k = 0; k < 2; ++k) {
        if (d->ev_fold[k]) hipEventDestroy(d->ev_fold[k]);
        if (d->ev_slot[k]) hipEventDestroy(d->ev_slot[k]);
        ws_free(d, d->carry[k], d->carry_waves * PT_CARRY_STRIDE_DW * sizeof(uint32_t));
        if (d->lane[k]) hipStreamDestroy(d->lane[k]);
    }
    if (d->ev_fork) hipEventDestroy(d->ev_fork);
    if (d->ev_ext) hipEventDestroy(d->ev_ext);
    if (d->own_stream) hipStreamDestroy(d->own_stream);
    delete d;
}

extern "C" int pt_device_create(int device_idx, pt_device_t* out)
{
    if (!out) return fail(PT_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0) return fail(PT_ERR_NO_DEVICE, "no HIP device visible (%s)", hipGetErrorString(e));
    if (device_idx < 0 || device_idx >= n) return fail(PT_ERR_NO_DEVICE, "device index %d out of range [0,%d)", device_idx, n);
    pt_device_s* d = new (std::nothrow) pt_device_s();
    if (!d) return fail(PT_ERR_OOM, "host allocation failed");
    memset(static_cast<void*>(d), 0, sizeof *d);
    d->idx = device_idx;
    if (hipSetDevice(device_idx) != hipSuccess || hipGetDeviceProperties(&d->prop, device_idx) != hipSuccess) {
        delete d;
        return fail(PT_ERR_HIP, "cannot open device %d", device_idx);
    }
    if (strncmp(d->prop.gcnArchName, "gfx950", 6) != 0) {
        char arch[64];
        snprintf(arch, sizeof arch, "%s", d->prop.gcnArchName);
        delete d;
        return fail(PT_ERR_NO_DEVICE, "device %d is %s; this library carries gfx950 (MI355X) code only", device_idx, arch);
    }
    if (hipStreamCreateWithFlags(&d->own_stream, hipStreamNonBlocking) != hipSuccess) {
        delete d;
        return fail(PT_ERR_HIP, "hipStreamCreate failed");
    }
    d->stream = d->own_stream;
    d->opt_batch = 1;
    d->opt_chunk = 0;
    d->opt_profile = 0;
    d->opt_quads = 0;
    d->opt_accel = 0;
    d->opt_pmask = PT_DEFAULT_PRIMARY_MASKS;
    d->opt_bvh_stack = 64;
    d->opt_lanes = 2;
    d->opt_carry = 1;
    d->kernels[KERNEL_GENERATE_COLORS] = { KERNEL_GENERATE_COLORS, "GenerateColors", "GenerateColors" };
    d->kernels[KERNEL_FILL] = { KERNEL_FILL, "PtShimTest", "FillKernel" };
    d->kernels[KERNEL_MATH] = { KERNEL_MATH, "PtShimTest", "MathKernel" };
    d->kernels[KERNEL_FOLD_CHECK] = { KERNEL_FOLD_CHECK, "PtShimTest", "FoldCheckKernel" };
    d->prof_pairs = new std::vector<std::pair<hipEvent_t, hipEvent_t>>[PT_PROF_KINDS];
    bool ok = ws_malloc(d, &d->bigtab, WS_BIGTAB) == hipSuccess && ws_malloc(d, &d->bigidx, WS_BIGIDX) == hipSuccess &&
              ws_malloc(d, &d->counters, WS_COUNTERS) == hipSuccess && ws_malloc(d, &d->big_p1tab, ws_big_p1tab()) == hipSuccess &&
              ws_malloc(d, &d->det_bound_dev, WS_DETBOUND) == hipSuccess;
    // the render lanes and their events (no timing: they only order streams)
    for (int k = 0; ok && k < 2; ++k)
        ok = hipStreamCreateWithFlags(&d->lane[k], hipStreamNonBlocking) == hipSuccess &&
             hipEventCreateWithFlags(&d->ev_fold[k], hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&d->ev_slot[k], hipEventDisableTiming) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&d->ev_fork, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&d->ev_ext, hipEventDisableTiming) == hipSuccess;
    // the sticky words of PT_ERR_TRAVERSAL: host memory the LBVH kernels store to, read here without waiting for anything
    ok = ok && hipHostMalloc(reinterpret_cast<void**>(&d->trav_host), 64, hipHostMallocMapped) == hipSuccess &&
         hipHostGetDevicePointer(reinterpret_cast<void**>(&d->trav_dev), d->trav_host, 0) == hipSuccess;
    if (ok) {
        d->trav_host[0] = d->trav_host[1] = 0u;
        ok = hipMemsetAsync(d->counters, 0, WS_COUNTERS, d->own_stream) == hipSuccess && hipStreamSynchronize(d->own_stream) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        destroy_device_objects(d);
        return fail(PT_ERR_OOM, "workspace allocation failed");
    }
    d->blocks_per_cu = ptk_trace_blocks_per_cu(36);
    d->bvh_blocks_per_cu = ptk_trace_bvh_blocks_per_cu();
    *out = d;
    return PT_OK;
}

static int flush_pending(pt_device_s* d);

static int lanes_join(pt_device_s* d);

extern "C" int pt_device_destroy(pt_device_t d)
{
    int rc = use_device(d);
    if (rc) return rc;
    rc = flush_pending(d);
    lanes_join(d);
    hipStreamSynchronize(d->stream);
    if (d->live_buffers != 0)
        return fail(PT_ERR_INVALID, "%d buffer(s) of this device are still alive", d->live_buffers);
    destroy_device_objects(d);
    return rc;
}

extern "C" int pt_device_info(pt_device_t d, int kind, char out[128])
{
    if (!d || !out) return fail(PT_ERR_INVALID, "null argument");
    out[0] = 0;
    switch (kind) {
    case PT_INFO_NAME:
    case PT_INFO_BOARD:
        // some driver stacks report an empty marketing name; fall back to the ISA name
        if (d->prop.name[0]) snprintf(out, 128, "%s", d->prop.name);
        else snprintf(out, 128, "AMD Instinct (%.*s)", 96, d->prop.gcnArchName);
        break;
    case PT_INFO_VENDOR: snprintf(out, 128, "Advanced Micro Devices, Inc."); break;
    case PT_INFO_VERSION: {
        int rt = 0;
        hipRuntimeGetVersion(&rt);
        snprintf(out, 128, "HIP %d %.*s", rt, 96, d->prop.gcnArchName);
        break;
    }
    default: return fail(PT_ERR_INVALID, "unknown info kind %d", kind);
    }
    return PT_OK;
}

extern "C" uint64_t pt_device_max_alloc(pt_device_t d) { return d ? (uint64_t)d->prop.totalGlobalMem : 0; }
extern "C" uint64_t pt_device_mem_size(pt_device_t d) { return d ? (uint64_t)d->prop.totalGlobalMem : 0; }
extern "C" uint64_t pt_device_used_memory(pt_device_t d) { return d ? d->used : 0; }
extern "C" uint64_t pt_device_peak_memory(pt_device_t d) { return d ? d->peak : 0; }
extern "C" uint64_t pt_device_workspace_memory(pt_device_t d) { return d ? d->workspace : 0; }
extern "C" int pt_device_num_cus(pt_device_t d) { return d ? d->prop.multiProcessorCount : 0; }

// ---- stream discipline -------------------------------------------------------------------------------------------------
// The handle's stream carries every call but the fused renders, whose launches live on the two lanes (pt_device_s).  Two
// one-way hand-overs keep the whole in call order:
//   fork  (render_internal): the lanes wait for what is on `stream` -- only when something has been put there since the
//         last fork (main_dirty), or the stream is the caller's, who may have enqueued on it without telling us;
//   join  (lanes_join): `stream` waits for the newest fold -- only when a call needs the renders' results on `stream`, so
//         that back-to-back renders never meet a join and overlap on the device.
// a caller's stream handle: PT_STREAM_LEGACY (the value of HIP's hipStreamLegacy sentinel) is the legacy default stream, which every
// HIP entry point knows as the null stream (the sentinel itself is not accepted by every runtime version torch ships)
static hipStream_t as_stream(void* h) { return h == PT_STREAM_LEGACY ? nullptr : (hipStream_t)h; }

static int lanes_join(pt_device_s* d)
{
    if (!d->lanes_busy) return PT_OK;
    if (d->fold_recorded[d->last_fold_lane]) HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_fold[d->last_fold_lane], 0));
    d->lanes_busy = false;
    return PT_OK;
}

// every entry point that enqueues on, or waits for, the handle's stream: deferred frames are submitted and the stream is
// ordered behind the lanes first
static int enter_stream(pt_device_s* d)
{
    int rc = flush_pending(d);
    if (!rc) rc = lanes_join(d);
    d->main_dirty = true;
    return rc;
}

// PT_ERR_TRAVERSAL, deferred (include/pt_shim.h): read -- and clear -- the words the LBVH kernels raise.  Called wherever the
// host observes the device; never waits for anything itself.
static int check_traversal(pt_device_s* d)
{
    if (!d->trav_host) return PT_OK;
    const unsigned st = __atomic_exchange_n(&d->trav_host[0], 0u, __ATOMIC_ACQ_REL);
    const unsigned bu = __atomic_exchange_n(&d->trav_host[1], 0u, __ATOMIC_ACQ_REL);
    if (!st && !bu) return PT_OK;
    return fail(PT_ERR_TRAVERSAL, "LBVH search cut short (%s%s%s): the framebuffers of the renders since the last check are not valid",
                st ? "stack capacity" : "", st && bu ? ", " : "", bu ? "step budget" : "");
}

extern "C" int pt_device_set_stream(pt_device_t d, void* hip_stream)
{
    int rc = use_device(d);
    if (rc) return rc;
    if ((rc = enter_stream(d))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));  // keep the one-queue ordering across the switch
    // NULL restores the handle's own (non-blocking) stream.  The legacy default stream -- what a
    // caller's "stream 0" means -- is named by PT_STREAM_LEGACY and used as HIP's null stream.
    d->stream = hip_stream ? as_stream(hip_stream) : d->own_stream;
    d->external = hip_stream != nullptr;
    return PT_OK;
}

extern "C" void* pt_device_get_stream(pt_device_t d) { return !d ? nullptr : d->external && !d->stream ? PT_STREAM_LEGACY : (void*)d->stream; }

extern "C" int pt_device_wait_stream(pt_device_t d, void* hip_stream)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!hip_stream) return fail(PT_ERR_INVALID, "hip_stream == NULL (the legacy default stream is PT_STREAM_LEGACY)");
    // (an event may be recorded again while a wait on its earlier record is pending: the wait keeps the record it saw)
    HIP_TRY(hipEventRecord(d->ev_ext, as_stream(hip_stream)));
    HIP_TRY(hipStreamWaitEvent(d->stream, d->ev_ext, 0));
    d->main_dirty = true;
    return PT_OK;
}

extern "C" int pt_device_wait_hip_event(pt_device_t d, void* hip_event)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!hip_event) return fail(PT_ERR_INVALID, "hip_event == NULL");
    HIP_TRY(hipStreamWaitEvent(d->stream, (hipEvent_t)hip_event, 0));
    d->main_dirty = true;
    return PT_OK;
}

extern "C" int pt_sync(pt_device_t d)
{
    int rc = use_device(d);
    if (rc) return rc;
    // Deferred frames (PT_OPT_BATCH_FRAMES) touch shim-owned, never-exposed buffers only
    // (launch_generate_colors), which can be observed through this ABI alone: every such
    // observation flushes.  Should a pending buffer have been exposed since (pt_buffer_device_ptr
    // flushes, so this is belt and braces), clFinish semantics are restored here.
    if (d->pending.active && (d->pending.tris->exposed || d->pending.mats->exposed || d->pending.fb->exposed) &&
        (rc = flush_pending(d)))
        return rc;
    if ((rc = lanes_join(d))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));
    return check_traversal(d);
}

extern "C" int pt_flush(pt_device_t d)
{
    int rc = use_device(d);
    if (rc) return rc;
    return flush_pending(d);
}

static int free_ring(pt_device_s* d)
{
    int rc = lanes_join(d);
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));   // nothing in flight may still write it
    ws_free(d, d->ring, PT_RING_SLOTS * d->ring_slot_bytes);
    d->ring = nullptr;
    d->ring_slot_bytes = 0;
    return PT_OK;
}

static int alloc_ring(pt_device_s* d, size_t slot_bytes)
{
    slot_bytes = (slot_bytes + 255) & ~(size_t)255;
    hipError_t e = ws_malloc(d, &d->ring, PT_RING_SLOTS * slot_bytes);
    if (e != hipSuccess) return fail(PT_ERR_OOM, "radiance staging ring (%d x %zu bytes) allocation failed: %s", PT_RING_SLOTS, slot_bytes, hipGetErrorString(e));
    d->ring_slot_bytes = slot_bytes;
    return PT_OK;
}

extern "C" int pt_device_reserve_staging(pt_device_t d, size_t bytes)
{
    int rc = use_device(d);
    if (rc) return rc;
    const size_t slot = bytes ? (bytes + PT_RING_SLOTS - 1) / PT_RING_SLOTS : PT_DEFAULT_SLOT_BYTES;
    if (slot < 12) return fail(PT_ERR_INVALID, "a staging slot holds at least one 12-byte sample");
    if ((rc = flush_pending(d))) return rc;
    if (d->ring && ((slot + 255) & ~(size_t)255) == d->ring_slot_bytes) return PT_OK;
    if (d->ring && (rc = free_ring(d))) return rc;
    return alloc_ring(d, slot);
}

extern "C" int pt_device_set_option(pt_device_t d, int option, int64_t value)
{
    int rc = use_device(d);
    if (rc) return rc;
    switch (option) {
    case PT_OPT_BATCH_FRAMES:
        rc = flush_pending(d);
        d->opt_batch = value ? 1 : 0;
        return rc;
    case PT_OPT_CHUNK_FRAMES:
        if (value < 0) return fail(PT_ERR_INVALID, "chunk frames must be >= 0");
        d->opt_chunk = value;
        return PT_OK;
    case PT_OPT_PROFILE_RETURN_TIME:
        d->opt_profile = value ? 1 : 0;
        return PT_OK;
    case PT_OPT_QUAD_FILTER:
        if (value < 0 || value > 4) return fail(PT_ERR_INVALID, "quad filter must be 0 (auto), 1..3 (independent triangles) or 4 (packed shared u)");
        d->opt_quads = value;
        return PT_OK;
    case PT_OPT_ACCEL:
        if (value < 0 || value > 2) return fail(PT_ERR_INVALID, "accel must be 0 (auto), 1 (brute force) or 2 (BVH)");
        d->opt_accel = value;
        return PT_OK;
    case PT_OPT_BVH_TALLY:
        d->opt_tally = value ? 1 : 0;
        return PT_OK;
    case PT_OPT_PRIMARY_MASKS:
        d->opt_pmask = value ? 1 : 0;
        return PT_OK;
    case PT_OPT_BVH_STACK_LIMIT:
        if (value < 1 || value > 64) return fail(PT_ERR_INVALID, "the LBVH stack limit must be 1..64");
        d->opt_bvh_stack = value;
        return PT_OK;
    case PT_OPT_RENDER_LANES:
        if (value < 1 || value > 2) return fail(PT_ERR_INVALID, "render lanes must be 1 or 2");
        rc = enter_stream(d);   // the next render forks: both lanes start behind everything enqueued under the old setting
        d->opt_lanes = value;
        return rc;
    case PT_OPT_CHECKPOINT:
        d->opt_carry = value ? 1 : 0;
        return PT_OK;
    case PT_OPT_RESERVED_3:   // ABI version 1's kernel-variant switch: both of its values select the one variant left
        if (value != 0 && value != 1) return fail(PT_ERR_INVALID, "option 3 took 0 or 1");
        return PT_OK;
    default: return fail(PT_ERR_INVALID, "unknown option %d", option);
    }
}

extern "C" int64_t pt_device_get_option(pt_device_t d, int option)
{
    if (!d) return -1;
    switch (option) {
    case PT_OPT_BATCH_FRAMES: return d->opt_batch;
    case PT_OPT_CHUNK_FRAMES: return d->opt_chunk;
    case PT_OPT_PROFILE_RETURN_TIME: return d->opt_profile;
    case PT_OPT_QUAD_FILTER: return d->opt_quads;
    case PT_OPT_ACCEL: return d->opt_accel;
    case PT_OPT_BVH_TALLY: return d->opt_tally;
    case PT_OPT_PRIMARY_MASKS: return d->opt_pmask;
    case PT_OPT_BVH_STACK_LIMIT: return d->opt_bvh_stack;
    case PT_OPT_RENDER_LANES: return d->opt_lanes;
    case PT_OPT_CHECKPOINT: return d->opt_carry;
    case PT_OPT_BVH_BUILD_COUNT: return (int64_t)d->bvh_builds;
    case PT_OPT_RESERVED_3: return 0;
    default: return -1;
    }
}

// ------------------------------------------------------------------------------------------
// buffers
// ------------------------------------------------------------------------------------------
extern "C" int pt_buffer_alloc(pt_device_t d, size_t bytes, pt_buffer_t* out)
{
    if (!out) return fail(PT_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int rc = use_device(d);
    if (rc) return rc;
    pt_buffer_s* b = new (std::nothrow) pt_buffer_s();
    if (!b) return fail(PT_ERR_OOM, "host allocation failed");
    memset(static_cast<void*>(b), 0, sizeof *b);
    b->dev = d;
    b->bytes = bytes;
    b->owned = true;
    if (bytes) {
        hipError_t e = hipMalloc(&b->dptr, bytes);
        if (e != hipSuccess) {
            delete b;
            (void)hipGetLastError();  // clear the sticky error: later launches check hipGetLastError()
            return fail(PT_ERR_OOM, "hipMalloc(%zu) failed: %s (used %llu bytes)", bytes, hipGetErrorString(e),
                        (unsigned long long)d->used);
        }
    }
    d->used += bytes;
    d->peak = std::max(d->peak, d->used);
    d->live_buffers++;
    *out = b;
    return PT_OK;
}

extern "C" int pt_buffer_wrap(pt_device_t d, void* device_ptr, size_t bytes, pt_buffer_t* out)
{
    if (!out) return fail(PT_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int rc = use_device(d);
    if (rc) return rc;
    if (!device_ptr && bytes) return fail(PT_ERR_INVALID, "null device pointer");
    pt_buffer_s* b = new (std::nothrow) pt_buffer_s();
    if (!b) return fail(PT_ERR_OOM, "host allocation failed");
    memset(static_cast<void*>(b), 0, sizeof *b);
    b->dev = d;
    b->dptr = device_ptr;
    b->bytes = bytes;
    b->owned = false;
    d->live_buffers++;
    *out = b;
    return PT_OK;
}

extern "C" int pt_buffer_free(pt_buffer_t b)
{
    if (!b) return fail(PT_ERR_INVALID, "null buffer handle");
    pt_device_s* d = b->dev;
    int rc = use_device(d);
    if (rc) return rc;
    if (d->pending.active && (d->pending.tris == b || d->pending.mats == b || d->pending.fb == b)) rc = flush_pending(d);
    lanes_join(d);
    hipStreamSynchronize(d->stream);  // nothing in flight may still touch it
    if (d->prep_src == b) d->prep_src = nullptr;
    if (b->staging) hipHostFree(b->staging);
    if (b->owned) {
        if (b->dptr) hipFree(b->dptr);
        d->used -= b->bytes;
    }
    d->live_buffers--;
    delete b;
    return rc;
}

extern "C" size_t pt_buffer_size(pt_buffer_t b) { return b ? b->bytes : 0; }
extern "C" void* pt_buffer_address(pt_buffer_t b) { return b ? b->dptr : nullptr; }

extern "C" void* pt_buffer_device_ptr(pt_buffer_t b)
{
    if (!b) return nullptr;
    // From here on the caller can observe (or change) the memory without going through this ABI:
    // submit what is deferred now and never defer launches on this buffer again.
    if (!b->exposed) {
        b->exposed = true;
        pt_device_s* d = b->dev;
        if (d->pending.active && (d->pending.tris == b || d->pending.mats == b || d->pending.fb == b) &&
            (use_device(d) || flush_pending(d)))
            return nullptr;
    }
    return b->dptr;
}

static int check_range(const pt_buffer_s* b, size_t off, size_t bytes, const char* what)
{
    if (!b) return fail(PT_ERR_INVALID, "null buffer handle (%s)", what);
    if (off > b->bytes || bytes > b->bytes - off)
        return fail(PT_ERR_RANGE, "%s: range [%zu, +%zu) outside buffer of %zu bytes", what, off, bytes, b->bytes);
    return PT_OK;
}

static int event_begin(pt_device_s* d, pt_event_s* ev)
{
    if (!ev) return PT_OK;
    if (ev->dev != d) return fail(PT_ERR_INVALID, "event belongs to another device");
    HIP_TRY(hipEventRecord(ev->start, d->stream));
    return PT_OK;
}

static int event_end(pt_device_s* d, pt_event_s* ev)
{
    if (!ev) return PT_OK;
    HIP_TRY(hipEventRecord(ev->stop, d->stream));
    ev->recorded = true;
    return PT_OK;
}

extern "C" int pt_buffer_write(pt_buffer_t dst, const void* host_src, size_t bytes, size_t dst_offset, pt_event_t ev)
{
    int rc = check_range(dst, dst_offset, bytes, "pt_buffer_write");
    if (rc) return rc;
    pt_device_s* d = dst->dev;
    if ((rc = use_device(d)) || (rc = enter_stream(d))) return rc;
    if (!host_src && bytes) return fail(PT_ERR_INVALID, "null host pointer");
    if ((rc = event_begin(d, ev))) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync((char*)dst->dptr + dst_offset, host_src, bytes, hipMemcpyHostToDevice, d->stream));
    dst->version++;
    return event_end(d, ev);
}

extern "C" int pt_buffer_read(pt_buffer_t src, void* host_dst, size_t bytes, size_t src_offset, pt_event_t ev)
{
    int rc = check_range(src, src_offset, bytes, "pt_buffer_read");
    if (rc) return rc;
    pt_device_s* d = src->dev;
    if ((rc = use_device(d)) || (rc = enter_stream(d))) return rc;
    if (!host_dst && bytes) return fail(PT_ERR_INVALID, "null host pointer");
    if ((rc = event_begin(d, ev))) return rc;
    if (bytes) HIP_TRY(hipMemcpyAsync(host_dst, (const char*)src->dptr + src_offset, bytes, hipMemcpyDeviceToHost, d->stream));
    return event_end(d, ev);
}

extern "C" int pt_buffer_copy(pt_buffer_t dst, pt_buffer_t src, size_t bytes, size_t dst_offset, size_t src_offset,
                              pt_event_t ev)
{
    int rc = check_range(dst, dst_offset, bytes, "pt_buffer_copy(dst)");
    if (rc) return rc;
    if ((rc = check_range(src, src_offset, bytes, "pt_buffer_copy(src)"))) return rc;
    if (dst->dev != src->dev) return fail(PT_ERR_INVALID, "buffers belong to different devices");
    pt_device_s* d = dst->dev;
    if ((rc = use_device(d)) || (rc = enter_stream(d))) return rc;
    if ((rc = event_begin(d, ev))) return rc;
    if (bytes)
        HIP_TRY(hipMemcpyAsync((char*)dst->dptr + dst_offset, (const char*)src->dptr + src_offset, bytes,
                               hipMemcpyDeviceToDevice, d->stream));
    dst->version++;
    return event_end(d, ev);
}

extern "C" int pt_host_alloc(size_t bytes, void** out)
{
    if (!out) return fail(PT_ERR_INVALID, "null out pointer");
    *out = nullptr;
    if (g_init_count <= 0) return fail(PT_ERR_NO_DEVICE, "pt_init has not succeeded");
    hipError_t e = hipHostMalloc(out, std::max<size_t>(bytes, 1), hipHostMallocDefault);
    if (e != hipSuccess) { (void)hipGetLastError(); *out = nullptr; return fail(PT_ERR_OOM, "pinned host allocation of %zu bytes failed: %s", bytes, hipGetErrorString(e)); }
    return PT_OK;
}

extern "C" int pt_host_free(void* host_ptr)
{
    if (!host_ptr) return PT_OK;
    HIP_TRY(hipHostFree(host_ptr));
    return PT_OK;
}

extern "C" void* pt_buffer_map(pt_buffer_t b, size_t bytes, int blocking)
{
    if (!b) { fail(PT_ERR_INVALID, "null buffer handle"); return nullptr; }
    pt_device_s* d = b->dev;
    if (use_device(d) || enter_stream(d)) return nullptr;
    if (bytes == (size_t)-1) bytes = b->bytes;
    if (check_range(b, 0, bytes, "pt_buffer_map")) return nullptr;
    if (b->mapped) { fail(PT_ERR_INVALID, "buffer is already mapped"); return nullptr; }
    size_t need = std::max<size_t>(bytes, 1);
    if (b->staging_bytes < need) {
        if (b->staging) {
            hipStreamSynchronize(d->stream);
            hipHostFree(b->staging);
            b->staging = nullptr;
            b->staging_bytes = 0;
        }
        hipError_t e = hipHostMalloc(&b->staging, need, hipHostMallocDefault);
        if (e != hipSuccess) { fail(PT_ERR_OOM, "hipHostMalloc(%zu) failed: %s", need, hipGetErrorString(e)); return nullptr; }
        b->staging_bytes = need;
    }
    if (bytes) {
        hipError_t e = hipMemcpyAsync(b->staging, b->dptr, bytes, hipMemcpyDeviceToHost, d->stream);
        if (e != hipSuccess) { fail(PT_ERR_HIP, "map copy failed: %s", hipGetErrorString(e)); return nullptr; }
    }
    if (blocking) {
        hipError_t e = hipStreamSynchronize(d->stream);
        if (e != hipSuccess) { fail(PT_ERR_HIP, "map sync failed: %s", hipGetErrorString(e)); return nullptr; }
        if (check_traversal(d)) return nullptr;   // (the contents would be those of a failed render)
    }
    b->mapped = true;
    b->mapped_bytes = bytes;
    return b->staging;
}

extern "C" int pt_buffer_unmap(pt_buffer_t b, void* host_ptr)
{
    if (!b) return fail(PT_ERR_INVALID, "null buffer handle");
    if (!b->mapped || host_ptr != b->staging) return fail(PT_ERR_INVALID, "pointer was not returned by pt_buffer_map of this buffer");
    pt_device_s* d = b->dev;
    int rc;
    if ((rc = use_device(d)) || (rc = enter_stream(d))) return rc;
    // exactly the range pt_buffer_map handed out: the staging allocation is kept at its largest size
    // ever, and whatever lies beyond this map's range is a stale snapshot
    size_t bytes = std::min(b->mapped_bytes, b->bytes);
    if (bytes) HIP_TRY(hipMemcpyAsync(b->dptr, b->staging, bytes, hipMemcpyHostToDevice, d->stream));
    b->mapped = false;
    b->version++;
    return PT_OK;
}

// ------------------------------------------------------------------------------------------
// events
// ------------------------------------------------------------------------------------------
extern "C" int pt_event_create(pt_device_t d, pt_event_t* out)
{
    if (!out) return fail(PT_ERR_INVALID, "out == NULL");
    *out = nullptr;
    int rc = use_device(d);
    if (rc) return rc;
    pt_event_s* e = new (std::nothrow) pt_event_s();
    if (!e) return fail(PT_ERR_OOM, "host allocation failed");
    e->dev = d;
    e->recorded = false;
    if (hipEventCreate(&e->start) != hipSuccess || hipEventCreate(&e->stop) != hipSuccess) {
        delete e;
        return fail(PT_ERR_HIP, "hipEventCreate failed");
    }
    *out = e;
    return PT_OK;
}

extern "C" int pt_event_destroy(pt_event_t e)
{
    if (!e) return fail(PT_ERR_INVALID, "null event handle");
    hipEventDestroy(e->start);
    hipEventDestroy(e->stop);
    delete e;
    return PT_OK;
}

extern "C" int pt_event_wait(pt_event_t e)
{
    if (!e) return fail(PT_ERR_INVALID, "null event handle");
    if (!e->recorded) return PT_OK;
    int rc = use_device(e->dev);
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(e->stop));
    return check_traversal(e->dev);
}

extern "C" int pt_event_wait_on(pt_event_t e, void* hip_stream)
{
    if (!e) return fail(PT_ERR_INVALID, "null event handle");
    if (!hip_stream) return fail(PT_ERR_INVALID, "hip_stream == NULL (the legacy default stream is PT_STREAM_LEGACY)");
    if (!e->recorded) return PT_OK;
    int rc = use_device(e->dev);
    if (rc) return rc;
    HIP_TRY(hipStreamWaitEvent(as_stream(hip_stream), e->stop, 0));
    return PT_OK;
}

extern "C" int pt_event_is_complete(pt_event_t e)
{
    if (!e) { fail(PT_ERR_INVALID, "null event handle"); return -1; }
    if (!e->recorded) return 1;
    if (use_device(e->dev)) return -1;
    hipError_t s = hipEventQuery(e->stop);
    if (s == hipSuccess) return 1;
    if (s == hipErrorNotReady) return 0;
    fail(PT_ERR_HIP, "hipEventQuery: %s", hipGetErrorString(s));
    return -1;
}

extern "C" int pt_event_elapsed_ns(pt_event_t e, uint64_t* ns_out)
{
    if (!e || !ns_out) return fail(PT_ERR_INVALID, "null argument");
    if (!e->recorded) return fail(PT_ERR_INVALID, "event was never recorded");
    int rc = use_device(e->dev);
    if (rc) return rc;
    HIP_TRY(hipEventSynchronize(e->stop));
    float ms = 0.f;
    HIP_TRY(hipEventElapsedTime(&ms, e->start, e->stop));
    *ns_out = (uint64_t)((double)ms * 1.0e6);
    return check_traversal(e->dev);
}

// ------------------------------------------------------------------------------------------
// the fused renderer
// ------------------------------------------------------------------------------------------
extern "C" int pt_local_rows(int height, int stripe_rows, int n_ranks, int rank)
{
    if (height < 0 || stripe_rows < 1 || n_ranks < 1 || rank < 0 || rank >= n_ranks) return -1;
    long long rows = 0;
    long long period = (long long)stripe_rows * n_ranks;
    long long full = height / period;
    rows = full * stripe_rows;
    long long rem = height - full * period;          // rows of the last, partial period
    long long start = (long long)rank * stripe_rows;  // this rank's stripe inside it
    if (rem > start) rows += std::min<long long>(stripe_rows, rem - start);
    return (int)rows;
}

static int ensure_prep(pt_device_s* d, const pt_buffer_s* tris, int ntri)
{
    // wrapped (caller-owned) memory can change behind our back, and so can a buffer whose device pointer has been handed
    // out (pt_buffer_device_ptr: writes through it bump no version): always re-prepare those
    if (d->prep_capacity >= (size_t)ntri && d->prep_src == tris && d->prep_version == tris->version && d->prep_ntri == ntri && tris->owned && !tris->exposed)
        return PT_OK;
    // the prepared scene is about to change under whatever the lanes still run: the handle's stream goes behind them
    int rc = lanes_join(d);
    if (rc) return rc;
    d->main_dirty = true;
    if (d->prep_capacity < (size_t)ntri) {
        HIP_TRY(hipStreamSynchronize(d->stream));
        ws_free(d, d->prep, d->prep_capacity * sizeof(PtPrepTriangle));
        ws_free(d, d->p1tab, ptk_p1tab_floats((int)d->prep_capacity) * sizeof(float));
        ws_free(d, d->bvh, ptk_bvh_record_count((int)d->prep_capacity) * sizeof(PtBvh8Node));
        d->prep = nullptr;
        d->p1tab = nullptr;
        d->bvh = nullptr;
        d->bvh_valid = false;
        d->prep_capacity = 0;
        d->prep_src = nullptr;
        d->prep_hash_valid = false;
        size_t cap = std::max<size_t>((size_t)ntri, 64);
        hipError_t e = ws_malloc(d, &d->prep, cap * sizeof(PtPrepTriangle));
        if (e == hipSuccess) e = ws_malloc(d, &d->p1tab, ptk_p1tab_floats((int)cap) * sizeof(float));
        if (e != hipSuccess) {
            ws_free(d, d->prep, cap * sizeof(PtPrepTriangle));
            d->prep = nullptr;
            return fail(PT_ERR_OOM, "scene workspace allocation failed: %s", hipGetErrorString(e));
        }
        d->prep_capacity = cap;
    }
    HIP_TRY(ptk_prep_triangles((const PtRawTriangle*)tris->dptr, d->prep, ntri, d->det_bound_dev, d->stream));
    unsigned int words[PT_PREP_WORDS] = { 0u, 1u, 0x7fc00000u, 1u };
    HIP_TRY(hipMemcpyAsync(words, d->det_bound_dev, sizeof words, hipMemcpyDeviceToHost, d->stream));
    HIP_TRY(hipStreamSynchronize(d->stream));  // once per scene upload
    float bound, radius;
    memcpy(&bound, &words[0], sizeof bound);
    memcpy(&radius, &words[2], sizeof radius);
    d->prep_det_bounded = bound <= PT_DET_BOUND_MAX;  // false for NaN / Inf too
    d->prep_radius = radius;
    const bool pairs = words[1] == 0u && ntri > 0 && (ntri & 1) == 0;  // every (2k, 2k+1) has e2' == -e2
    d->prep_quads = 0;
    d->prep_delta1 = 0.0f;
    d->prep_ray_radius = 0.0f;
    if (pairs && d->prep_det_bounded && words[3] == 0u && radius <= 1.0e15f) {
        // quad mode 2: secondary rays start 0.01 off a surface (GenerateColors.cl:253), so their
        // origins stay within ray_radius of the eye; D bounds every coordinate difference between
        // two points of that box; delta1 = 128 u D^2 (derivation: pt_quad2_pass1)
        const float ray_radius = radius * 1.001f + 0.05f;
        const float diameter = 2.0f * ray_radius * 1.001f;
        const float delta1 = 128.0f * 5.9604645e-8f * diameter * diameter * 1.001f;
        HIP_TRY(ptk_prep_quad_margins(d->prep, ntri, diameter, delta1, d->p1tab, d->stream));
        d->prep_quads = 3;
        d->prep_delta1 = delta1;
        d->prep_ray_radius = ray_radius;
        // mode 3 (pt_quad3_pass1): deltaP = 192 u D^2; first triangle un >= -deltaP, second un <= delta1 + deltaP
        const float deltaP = 192.0f * 5.9604645e-8f * diameter * diameter * 1.001f;
        d->prep_p1_lo = -deltaP;
        d->prep_p1_hi = (delta1 + deltaP) * 1.001f;
    }
    d->blocks_per_cu = ptk_trace_blocks_per_cu(ntri);  // the LDS footprint follows the scene
    // Memory the caller can write behind the ABI is prepared again for every render; the checksum of the raw records says whether
    // that changed anything -- if not, the LBVH (a radix sort, a build, host waits) and the primary-ray masks still stand
    const uint64_t hash = (uint64_t)words[4] | ((uint64_t)words[5] << 32);
    if (!(d->prep_src == tris && d->prep_ntri == ntri && d->prep_hash_valid && d->prep_hash == hash)) {
        d->bvh_valid = false;
        d->scene_gen++;
    }
    d->prep_hash = hash;
    d->prep_hash_valid = true;
    d->prep_src = tris;
    d->prep_version = tris->version;
    d->prep_ntri = ntri;
    return PT_OK;
}

// LBVH of the current prepared scene: built on the GPU the first time a render
// of this scene wants it (PT_OPT_ACCEL), kept until the scene changes
static int ensure_bvh(pt_device_s* d, const pt_buffer_s* tris, int ntri)
{
    if (d->bvh_valid) return PT_OK;
    int rc = lanes_join(d);   // (a render through the old hierarchy may still be running)
    if (rc) return rc;
    d->main_dirty = true;
    if (!d->bvh) {
        hipError_t e = ws_malloc(d, &d->bvh, ptk_bvh_record_count((int)d->prep_capacity) * sizeof(PtBvh8Node));
        if (e != hipSuccess) return fail(PT_ERR_OOM, "BVH allocation failed: %s", hipGetErrorString(e));
    }
    const size_t temp_bytes = ptk_bvh_temp_bytes(ntri);
    void* temp = nullptr;
    hipError_t e = hipMalloc(&temp, temp_bytes);
    if (e != hipSuccess) { (void)hipGetLastError(); return fail(PT_ERR_OOM, "BVH build workspace allocation failed: %s", hipGetErrorString(e)); }
    PtBvhGrid* grid_dev = reinterpret_cast<PtBvhGrid*>(d->bigidx + PT_BVH_BIG_MAX + 1);
    e = ptk_bvh_build((const PtRawTriangle*)tris->dptr, d->prep, ntri, d->bvh, d->bigtab, d->bigidx, d->bigidx + PT_BVH_BIG_MAX, grid_dev,
                      reinterpret_cast<unsigned*>(grid_dev + 1), temp, temp_bytes, d->stream);
    struct { int nbig; PtBvhGrid grid; unsigned used; } back;
    memset(&back, 0, sizeof back);
    static_assert(sizeof back == sizeof(int) + sizeof(PtBvhGrid) + sizeof(unsigned), "count, grid and records in use are read back together");
    if (e == hipSuccess) e = hipMemcpyAsync(&back, d->bigidx + PT_BVH_BIG_MAX, sizeof back, hipMemcpyDeviceToHost, d->stream);
    int& nbig = back.nbig;
    hipError_t e2 = hipStreamSynchronize(d->stream);  // once per scene upload; the workspace is freed right after
    hipFree(temp);
    if (e != hipSuccess || e2 != hipSuccess) return fail(PT_ERR_HIP, "BVH build failed: %s", hipGetErrorString(e != hipSuccess ? e : e2));
    d->nbig = nbig < 0 ? 0 : (nbig > PT_BVH_BIG_MAX ? PT_BVH_BIG_MAX : nbig);
    d->bvh_grid = back.grid;
    d->bvh_records = std::min<size_t>(back.used, ptk_bvh_record_count(ntri));   // (indices are checked against this: a tighter bound than the capacity)
    if (getenv("PT_SHIM_DEBUG"))
        fprintf(stderr, "pt_shim: LBVH over %d triangles: %u records in use (%u nodes, %.2f children per node), %d big triangles outside\n", ntri, back.used,
                back.used > (unsigned)ntri ? back.used - (unsigned)(ntri - d->nbig) : 0u,
                back.used > (unsigned)ntri ? (double)(back.used - 1) / (double)(back.used - (unsigned)(ntri - d->nbig)) : 0.0, d->nbig);
    // The big triangles are searched by the brute-force two-pass search before every traversal.  When their table is made
    // of quads (a,b,c),(c,d,a) -- the Cornell box's walls among a soup of small triangles -- it gets the packed shared-u
    // filter (pt_quad3_pass1) like a quad scene on the brute-force path: same preparation (ensure_prep), on the table.  The
    // filter's error bound is about where RAYS start: anywhere on the scene, so the radius is the whole scene's.
    d->big_quads = 0;
#ifndef PT_EXP_NO_BIGQ  // (A/B builds: the big triangles' search with the plain filter)
    if (d->nbig >= 2 && (d->nbig & 1) == 0 && d->prep_det_bounded && d->prep_radius <= 1.0e15f) {
        PtRawTriangle* rawbig = reinterpret_cast<PtRawTriangle*>(d->big_p1tab + ptk_p1tab_floats(PT_BVH_BIG_MAX));
        HIP_TRY(ptk_bvh_big_raw((const PtRawTriangle*)tris->dptr, d->bigidx, d->nbig, rawbig, d->stream));
        HIP_TRY(ptk_prep_triangles(rawbig, d->bigtab, d->nbig, d->det_bound_dev, d->stream));  // (rewrites the same records)
        unsigned int words[PT_PREP_WORDS] = { 0u, 1u, 0x7fc00000u, 1u };
        HIP_TRY(hipMemcpyAsync(words, d->det_bound_dev, sizeof words, hipMemcpyDeviceToHost, d->stream));
        HIP_TRY(hipStreamSynchronize(d->stream));
        if (words[1] == 0u && words[3] == 0u) {
            const float ray_radius = d->prep_radius * 1.001f + 0.05f;
            const float diameter = 2.0f * ray_radius * 1.001f;
            const float delta1 = 128.0f * 5.9604645e-8f * diameter * diameter * 1.001f;
            HIP_TRY(ptk_prep_quad_margins(d->bigtab, d->nbig, diameter, delta1, d->big_p1tab, d->stream));
            const float deltaP = 192.0f * 5.9604645e-8f * diameter * diameter * 1.001f;
            d->big_quads = 3;
            d->big_delta1 = delta1;
            d->big_ray_radius = ray_radius;
            d->big_p1_lo = -deltaP;
            d->big_p1_hi = (delta1 + deltaP) * 1.001f;
        }
    }
#endif
    d->bvh_valid = true;
    d->bvh_builds++;
    return PT_OK;
}

// one render of at most 32 768 frames: its launches on one lane (pt_device_s).  ev_start / ev_stop: the caller's event, when
// this part begins / ends the call
static int render_part(pt_device_s* d, pt_buffer_s* tris, pt_buffer_s* mats, pt_buffer_s* fb, const pt_render_params& rp, uint32_t npix,
                       pt_buffer_s* stats, pt_event_s* ev_start, pt_event_s* ev_stop)
{
    int rc;
    // PT_OPT_ACCEL: 0 = BVH for scenes of PT_BVH_AUTO_MIN triangles or more, 1 = brute force, 2 = BVH (needs >= 2 triangles)
    const bool use_bvh = rp.num_triangles >= 2 &&
                         (d->opt_accel == 2 || (d->opt_accel == 0 && rp.num_triangles >= PT_BVH_AUTO_MIN));
    if (!use_bvh && rp.num_triangles >= (1 << 26))
        return fail(PT_ERR_INVALID, "the brute-force search packs a triangle index in 26 bits: use PT_OPT_ACCEL 0 or 2 for %d triangles", rp.num_triangles);
    if (use_bvh && rp.num_triangles >= (1 << 25))
        return fail(PT_ERR_INVALID, "the LBVH search packs a record index (2 x triangles) in 26 bits: %d triangles are too many", rp.num_triangles);
    if ((rc = ensure_prep(d, tris, rp.num_triangles))) return rc;
    if (use_bvh && (rc = ensure_bvh(d, tris, rp.num_triangles))) return rc;

    // ---- the staging ring: whole frames per chunk, as many as a slot holds -------------------------------------------------
    // Steady state allocates nothing: the ring exists from the first render on (pt_device_reserve_staging) and is replaced
    // only when ONE frame of this image does not fit a slot.
    const uint64_t frame_bytes = (uint64_t)npix * 12;
    if (!d->ring || d->ring_slot_bytes < frame_bytes) {
        if (d->ring && (rc = free_ring(d))) return rc;
        if ((rc = alloc_ring(d, (size_t)std::max<uint64_t>(frame_bytes, PT_DEFAULT_SLOT_BYTES)))) return rc;
    }
    uint64_t slot_frames = std::min<uint64_t>(d->ring_slot_bytes / frame_bytes, 16383);   // (fl + ring_phase stays below 65 536: pt_kernels.h, ring_magic)
    if (d->opt_chunk > 0) slot_frames = std::min<uint64_t>(slot_frames, (uint64_t)d->opt_chunk);
    const int nchunks = (int)(((uint64_t)rp.frame_count + slot_frames - 1) / slot_frames);
    const int chunk = (rp.frame_count + nchunks - 1) / nchunks;      // S: equal chunks -- 40 frames through a 16-frame slot go 14, 14, 12

    // primary-ray candidate masks: quad scenes of up to 64 triangles on the brute-force path (PT_OPT_PRIMARY_MASKS).  They are a
    // function of the image geometry and the prepared scene: made when either changes, on the handle's stream, behind the lanes
    const int quads_sel = (d->opt_quads == 0 || d->opt_quads == 4) ? d->prep_quads : 0;
    const bool use_pmask = d->opt_pmask && quads_sel == 3 && !use_bvh && rp.num_triangles <= 64 && d->prep_det_bounded;
    bool make_pmask = false;
    if (use_pmask) {
        const int32_t g[7] = { rp.width, rp.height, rp.stripe_rows, rp.n_ranks, rp.rank, (int32_t)npix, rp.num_triangles };
        make_pmask = !d->pmask_key.valid || d->pmask_key.scene_gen != d->scene_gen || memcmp(d->pmask_key.g, g, sizeof g) != 0;
        if (make_pmask) {
            if ((rc = lanes_join(d))) return rc;   // (a render with the old masks may still be running)
            d->main_dirty = true;
            d->pmask_key.valid = false;
            if (d->pmask_pixels < npix) {
                HIP_TRY(hipStreamSynchronize(d->stream));
                ws_free(d, d->pmask, d->pmask_pixels * sizeof(uint2));
                d->pmask = nullptr;
                d->pmask_pixels = 0;
                hipError_t e = ws_malloc(d, &d->pmask, (size_t)npix * sizeof(uint2));
                if (e != hipSuccess) return fail(PT_ERR_OOM, "primary-mask allocation (%zu bytes) failed: %s", (size_t)npix * sizeof(uint2), hipGetErrorString(e));
                d->pmask_pixels = npix;
            }
            d->pmask_key.scene_gen = d->scene_gen;
            memcpy(d->pmask_key.g, g, sizeof g);
        }
    }
    if (d->counters_dirty) {   // an earlier call failed between a trace launch and the fold that resets its counter
        if ((rc = lanes_join(d))) return rc;
        HIP_TRY(hipMemsetAsync(d->counters, 0, WS_COUNTERS, d->stream));
        d->main_dirty = true;
        d->counters_dirty = false;
    }

    // Samples per work-queue grab.  A wave that finds the queue empty idles until the last wave is
    // done, on average for half a batch: large batches (fewer atomics) when every wave gets many of
    // them, smaller ones when the launch is short (a rank's share of a multi-GPU render, small images).
    const uint64_t resident_waves = (uint64_t)d->prop.multiProcessorCount * (uint64_t)(use_bvh ? d->bvh_blocks_per_cu : d->blocks_per_cu) * (PT_TRACE_THREADS / 64);
    const uint64_t chunk_samples = (uint64_t)npix * (uint64_t)chunk;
    // (not below 128: at 64 the ONE queue counter takes 4 M atomics per launch of configs[2] and the
    // L2 atomic unit saturates -- measured +30 % launch time)
    uint32_t batch = PT_TRACE_BATCH;
    while (batch > 128u && chunk_samples / batch < 64u * resident_waves) batch >>= 1;
    if (getenv("PT_SHIM_BATCH")) batch = (uint32_t)std::max(64, atoi(getenv("PT_SHIM_BATCH")));   // (experiments)
    const uint32_t bpf = (npix + batch - 1) / batch;
    if ((uint64_t)bpf * (uint64_t)chunk + PT_QUEUE_SHARDS > 0xffffffffull) return fail(PT_ERR_INVALID, "chunk too large");

    PtTraceParams tp;
    memset(&tp, 0, sizeof tp);
    tp.tris = d->prep;
    tp.mats = (const PtRawMaterial*)mats->dptr;
    tp.stats = stats ? (unsigned long long*)stats->dptr : nullptr;
    tp.width = rp.width;
    tp.height = rp.height;
    tp.inv_width = 1.0f / (float)rp.width;     // IEEE quotients: the translation unit is compiled without fast-math
    tp.inv_height = 1.0f / (float)rp.height;
    tp.aspect = (float)rp.width / (float)rp.height;
    tp.max_bounces = rp.max_bounces;
    tp.ntri = rp.num_triangles;
    tp.nmat = rp.num_materials;
    tp.stripe_rows = rp.stripe_rows;
    tp.n_ranks = rp.n_ranks;
    tp.rank = rp.rank;
    tp.npix_local = npix;
    tp.batches_per_frame = bpf;
    tp.batch = batch;
    tp.quad_delta1 = use_bvh ? d->big_delta1 : d->prep_delta1;
    tp.ray_radius = use_bvh ? d->big_ray_radius : d->prep_ray_radius;
    tp.p1tab = use_bvh ? d->big_p1tab : d->p1tab;
    tp.p1_lo = use_bvh ? d->big_p1_lo : d->prep_p1_lo;
    tp.p1_hi = use_bvh ? d->big_p1_hi : d->prep_p1_hi;
    tp.bvh = d->bvh;
    tp.bvh_records = (int32_t)d->bvh_records;
    tp.grid = d->bvh_grid;
    tp.bigtab = d->bigtab;
    tp.bigidx = d->bigidx;
    tp.nbig = use_bvh ? d->nbig : 0;
    tp.pmask = use_pmask ? d->pmask : nullptr;
    tp.bvh_flags = d->trav_dev;
    tp.bvh_stack_limit = (int32_t)d->opt_bvh_stack;
    tp.slot_frames = (uint32_t)chunk;
    tp.ring_magic = (uint32_t)(0x100000000ull / (2u * (uint32_t)chunk)) + 1u;
    tp.rad = d->ring;
    tp.rad1 = reinterpret_cast<float*>(reinterpret_cast<char*>(d->ring) + d->ring_slot_bytes);
    if (make_pmask) {
        HIP_TRY(ptk_primary_masks(tp, d->stream));  // (cheap: one thread per pixel)
        d->pmask_key.valid = true;
    }
    // PT_OPT_QUAD_FILTER: 0 / 4 = the packed shared-u filter when the scene allows it, 1..3 = independent triangles
    const int quads = (d->opt_quads == 0 || d->opt_quads == 4) ? (use_bvh ? d->big_quads : d->prep_quads) : 0;

    // persistent grid: fill the chip, but never more waves than a chunk has batches; ONE grid for all launches of the render (a
    // checkpoint is resumed by the wave of the same number)
    const int wg_waves = PT_TRACE_THREADS / 64;
    int blocks = d->prop.multiProcessorCount * (use_bvh ? d->bvh_blocks_per_cu : d->blocks_per_cu);
    {
        const uint64_t blocks_needed = ((uint64_t)bpf * (uint64_t)chunk + wg_waves - 1) / wg_waves;
        if ((uint64_t)blocks > blocks_needed) blocks = (int)blocks_needed;
    }
    // checkpointed launches (PtTraceParams::carry): the table kernels stop at a fresh-phase boundary, the LBVH kernel between two searches
    // (a render of ONE chunk has nothing to hand on: its only launch runs its paths out, beside the next render's first launch on the other lane --
    // a checkpoint would add the draining launch and the checkpoint's traffic for nothing: 30.64 against 30.41 ms per one-launch configs[2] render)
    const bool carry = d->opt_carry != 0 && nchunks > 1;
    // the lane: one per render when its launches are checkpointed (each resumes its predecessor); otherwise the CHUNKS alternate, so that
    // chunk c+1's launch becomes resident while chunk c's runs its paths out (an LBVH launch's tail is milliseconds of falling lane use)
    int ln = d->opt_lanes == 2 ? (int)(d->render_seq++ & 1u) : 0;
    hipStream_t st = d->lane[ln];
    if (carry) {
        const size_t waves = (size_t)d->prop.multiProcessorCount * 8 * wg_waves;   // (no kernel has more than 8 workgroups per CU resident)
        if ((size_t)blocks * wg_waves > waves) return fail(PT_ERR_INVALID, "grid larger than the checkpoint regions");
        for (int k = 0; k < 2; ++k)
            if (!d->carry[k]) {   // once per device handle, at its first such render
                hipError_t e = ws_malloc(d, &d->carry[k], waves * PT_CARRY_STRIDE_DW * sizeof(uint32_t));
                if (e != hipSuccess) return fail(PT_ERR_OOM, "checkpoint buffer allocation failed: %s", hipGetErrorString(e));
                d->carry_waves = waves;
            }
    }

    // ---- fork: the lanes go behind the handle's stream (what uploaded the scene, cleared the framebuffer, made the masks ...)
    if (d->main_dirty || d->external) {
        HIP_TRY(hipEventRecord(d->ev_fork, d->stream));
        HIP_TRY(hipStreamWaitEvent(d->lane[0], d->ev_fork, 0));
        HIP_TRY(hipStreamWaitEvent(d->lane[1], d->ev_fork, 0));
        d->main_dirty = false;
    }
    d->lanes_busy = true;
    d->counters_dirty = true;   // until the last fold of this call is enqueued
    if (ev_start) HIP_TRY(hipEventRecord(ev_start->start, st));
    tp.ring_phase = (uint32_t)((d->chunk_seq & 1u) ? chunk : 0);   // the render's first chunk goes to slot chunk_seq % 2
    tp.carry = carry ? d->carry[ln] : nullptr;
    unsigned int* const drain_counter = d->counters + (size_t)(PT_QUEUE_COUNTERS - 1 - ln) * PT_QUEUE_STRIDE;
    unsigned int* prev_counter = nullptr;
    int prev_slot = 0, prev_f0 = 0, prev_nf = 0;
    hipEvent_t pstop;
    // fold of the chunk [f0, f0 + nf) staged in ring slot `slot`: behind the newest fold of the other lane (the chain), then
    // the slot is free again
    bool chained = false;   // this render's folds are behind the previous render's (on the other lane)
    auto fold = [&](int slot, int f0, int nf, unsigned int* reset, unsigned int* reset2) -> int {
        const bool last = f0 + nf >= rp.frame_count || !carry, last_two = f0 + nf + chunk >= rp.frame_count;
        if ((!chained || !carry) && d->fold_recorded[ln ^ 1]) HIP_TRY(hipStreamWaitEvent(st, d->ev_fold[ln ^ 1], 0));
        chained = true;
        PtFoldParams fp;
        fp.rad = slot ? tp.rad1 : tp.rad;
        fp.fb = (float4*)fb->dptr;
        fp.npix_local = npix;
        fp.frame_begin = rp.frame_begin + f0;
        fp.frame_count = nf;
        fp.reset_counter = reset;
        fp.reset_counter2 = reset2;
        int r = prof_begin(d, PT_PROF_FOLD, st, &pstop);
        if (r) return r;
        HIP_TRY(ptk_fold(fp, st));
        if ((r = prof_end(st, pstop))) return r;
        // what other lanes wait for: the slot is free again (the next render's first two chunks), the fold chain's newest link (its
        // first fold; lanes_join) -- only a render's last two folds can be either when all its launches share a lane
        if (last_two) {
            HIP_TRY(hipEventRecord(d->ev_slot[slot], st));
            d->slot_recorded[slot] = true;
        }
        if (last) {
            HIP_TRY(hipEventRecord(d->ev_fold[ln], st));
            d->fold_recorded[ln] = true;
            d->last_fold_lane = ln;
        }
        return PT_OK;
    };
    auto trace = [&](int f0, int nf, unsigned int* counter, bool carry_in, bool carry_out) -> int {
        tp.batch_counter = counter;
        tp.frame_begin = rp.frame_begin + f0;
        tp.frame_count = nf;
        tp.chunk_f0 = (uint32_t)f0;
        tp.carry_in_waves = carry_in ? (uint32_t)(blocks * wg_waves) : 0u;
        tp.carry_out = carry_out ? 1u : 0u;
        tp.total_batches = (uint32_t)((uint64_t)bpf * (uint64_t)nf);
        int r = prof_begin(d, PT_PROF_TRACE, st, &pstop);
        if (r) return r;
        HIP_TRY(ptk_trace(tp, blocks, d->prep_det_bounded, quads, use_bvh, d->opt_tally != 0 && stats != nullptr, st));
        return prof_end(st, pstop);
    };
    for (int c = 0; c < nchunks; ++c) {
        const int f0 = c * chunk;
        const int nf = std::min(chunk, rp.frame_count - f0);
        const uint64_t seq = d->chunk_seq++;
        const int slot = (int)(seq & 1u);
        unsigned int* counter = d->counters + (size_t)(seq % (PT_QUEUE_COUNTERS - 2)) * PT_QUEUE_STRIDE;
        if (!carry && d->opt_lanes == 2 && c > 0) {
            ln ^= 1;
            st = d->lane[ln];
        }
        // the slot must have been folded out (by the other lane, when this is one of a render's first two chunks)
        if (c < 2 && d->slot_recorded[slot]) HIP_TRY(hipStreamWaitEvent(st, d->ev_slot[slot], 0));
        if ((rc = trace(f0, nf, counter, carry && c > 0, carry))) return rc;
        if (!carry) {
            if ((rc = fold(slot, f0, nf, counter, nullptr))) return rc;
        } else {
            // chunk c - 1 is complete now: this launch finished the paths its own left unfinished
            if (c > 0 && (rc = fold(prev_slot, prev_f0, prev_nf, prev_counter, nullptr))) return rc;
            prev_slot = slot; prev_f0 = f0; prev_nf = nf; prev_counter = counter;
        }
    }
    if (carry) {
        // D: an empty queue, no checkpoint at its end -- the last chunk's paths run out here (beside the next render's first
        // launch on the other lane) -- and the last fold
        if ((rc = trace(rp.frame_count, 0, drain_counter, true, false))) return rc;
        if ((rc = fold(prev_slot, prev_f0, prev_nf, prev_counter, drain_counter))) return rc;
    }
    d->counters_dirty = false;
    if (ev_stop) {
        HIP_TRY(hipEventRecord(ev_stop->stop, st));
        ev_stop->recorded = true;
    }
    // a stream of the caller's (pt_device_set_stream) promises stream order to code we cannot see: it goes behind the render now
    if (d->external && (rc = lanes_join(d))) return rc;
    return PT_OK;
}

// pixel_count: 0 = all local pixels; otherwise the first pixel_count local pixels (n_ranks must be 1)
static int render_internal(pt_device_s* d, pt_buffer_s* tris, pt_buffer_s* mats, pt_buffer_s* fb,
                           const pt_render_params& rp, uint32_t pixel_count, pt_buffer_s* stats, pt_event_s* ev)
{
    if (!tris || !mats || !fb) return fail(PT_ERR_INVALID, "null buffer handle");
    if (tris->dev != d || mats->dev != d || fb->dev != d || (stats && stats->dev != d))
        return fail(PT_ERR_INVALID, "buffer belongs to another device");
    if (rp.width < 1 || rp.height < 1 || rp.frame_begin < 0 || rp.frame_count < 0 || rp.max_bounces < 1 ||
        rp.num_triangles < 0 || rp.num_materials < 1 || rp.stripe_rows < 1 || rp.n_ranks < 1 || rp.rank < 0 ||
        rp.rank >= rp.n_ranks)
        return fail(PT_ERR_INVALID, "invalid render parameters");
    for (int i = 0; i < 6; ++i)
        if (rp.reserved[i] != 0) return fail(PT_ERR_INVALID, "reserved fields must be zero");
    if ((long long)rp.width * rp.height > 0x7fffffffLL) return fail(PT_ERR_INVALID, "image too large");
    if ((long long)rp.frame_begin + rp.frame_count > 0x7fffffffLL) return fail(PT_ERR_INVALID, "frame index overflow");
    if (rp.max_bounces > 65535) return fail(PT_ERR_INVALID, "max_bounces above 65535 (a parked path packs its bounce count in 16 bits)");
    int rows = pt_local_rows(rp.height, rp.stripe_rows, rp.n_ranks, rp.rank);
    uint64_t npix64 = (uint64_t)rows * (uint64_t)rp.width;
    if (pixel_count) {
        if (rp.n_ranks != 1) return fail(PT_ERR_INVALID, "pixel_count needs n_ranks == 1");
        npix64 = std::min<uint64_t>(npix64, pixel_count);
    }
    uint32_t npix = (uint32_t)npix64;
    if ((size_t)rp.num_triangles * sizeof(PtRawTriangle) > tris->bytes)
        return fail(PT_ERR_RANGE, "triangle buffer holds %zu bytes, %d triangles need %zu", tris->bytes, rp.num_triangles,
                    (size_t)rp.num_triangles * sizeof(PtRawTriangle));
    if ((size_t)rp.num_materials * sizeof(PtRawMaterial) > mats->bytes)
        return fail(PT_ERR_RANGE, "material buffer holds %zu bytes, %d materials need %zu", mats->bytes, rp.num_materials,
                    (size_t)rp.num_materials * sizeof(PtRawMaterial));
    if ((size_t)npix * sizeof(float4) > fb->bytes)
        return fail(PT_ERR_RANGE, "framebuffer holds %zu bytes, %u pixels need %zu", fb->bytes, npix, (size_t)npix * sizeof(float4));
    if (stats && stats->bytes < PT_STAT_WORDS * sizeof(uint64_t)) return fail(PT_ERR_RANGE, "stats buffer too small");

    // a search that was cut short in an earlier render is reported before anything new is enqueued (PT_ERR_TRAVERSAL is deferred)
    int rc = check_traversal(d);
    if (rc) return rc;
    if (ev && ev->dev != d) return fail(PT_ERR_INVALID, "event belongs to another device");
    if (npix == 0 || rp.frame_count == 0) {
        if ((rc = lanes_join(d)) || (rc = event_begin(d, ev))) return rc;
        return event_end(d, ev);
    }
    // a path keeps its frame, counted from the render's first, in 16 bits: longer renders go in parts (each its own sequence
    // of launches on one lane; the fold chain joins them)
    const int PART = 32768;
    pt_render_params part = rp;
    for (int done = 0; done < rp.frame_count; done += PART) {
        part.frame_begin = rp.frame_begin + done;
        part.frame_count = std::min(PART, rp.frame_count - done);
        if ((rc = render_part(d, tris, mats, fb, part, npix, stats, done == 0 ? ev : nullptr, done + PART >= rp.frame_count ? ev : nullptr))) return rc;
    }
    fb->version++;
    if (stats) stats->version++;
    return PT_OK;
}

static int flush_pending(pt_device_s* d)
{
    if (!d->pending.active) return PT_OK;
    PendingFrames p = d->pending;
    d->pending.active = false;
    pt_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.width = p.width;
    rp.height = p.height;
    rp.frame_begin = p.z_begin;
    rp.frame_count = p.z_count;
    rp.max_bounces = 16;    // BOUNCES       GenerateColors.cl:5
    rp.num_triangles = 36;  // NUM_TRIANGLES GenerateColors.cl:6
    rp.num_materials = (int)std::min<size_t>(p.mats->bytes / sizeof(PtRawMaterial), 0x7fffffff);
    rp.stripe_rows = 1;
    rp.n_ranks = 1;
    rp.rank = 0;
    return render_internal(d, p.tris, p.mats, p.fb, rp, p.pixel_count, nullptr, nullptr);
}

extern "C" int pt_render_frames(pt_device_t d, pt_buffer_t triangles, pt_buffer_t materials, pt_buffer_t framebuffer,
                                const pt_render_params* params, pt_buffer_t stats, pt_event_t ev)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!params) return fail(PT_ERR_INVALID, "params == NULL");
    if ((rc = flush_pending(d))) return rc;
    return render_internal(d, triangles, materials, framebuffer, *params, 0, stats, ev);
}

static int assemble_check(pt_device_s* d, pt_buffer_s* gathered, pt_buffer_s* image, int width, int height, int stripe_rows, int n_ranks, int slab_rows);

extern "C" int pt_assemble_stripes_on(pt_device_t d, pt_buffer_t gathered, pt_buffer_t image, int width, int height,
                                      int stripe_rows, int n_ranks, int slab_rows, void* hip_stream)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!hip_stream) return fail(PT_ERR_INVALID, "hip_stream == NULL (pt_assemble_stripes uses the handle's own stream)");
    if ((rc = assemble_check(d, gathered, image, width, height, stripe_rows, n_ranks, slab_rows)) || (rc = flush_pending(d))) return rc;
    HIP_TRY(ptk_assemble_stripes((const float4*)gathered->dptr, (float4*)image->dptr, width, height, stripe_rows, n_ranks,
                                 slab_rows, as_stream(hip_stream)));
    image->version++;
    return PT_OK;
}

static int assemble_check(pt_device_s* d, pt_buffer_s* gathered, pt_buffer_s* image, int width, int height, int stripe_rows, int n_ranks, int slab_rows)
{
    if (!gathered || !image) return fail(PT_ERR_INVALID, "null buffer handle");
    if (gathered->dev != d || image->dev != d) return fail(PT_ERR_INVALID, "buffer belongs to another device");
    if (width < 1 || height < 1 || stripe_rows < 1 || n_ranks < 1 || slab_rows < 0) return fail(PT_ERR_INVALID, "invalid geometry");
    for (int r = 0; r < n_ranks; ++r)
        if (pt_local_rows(height, stripe_rows, n_ranks, r) > slab_rows) return fail(PT_ERR_INVALID, "slab_rows too small for rank %d", r);
    if ((size_t)n_ranks * slab_rows * width * sizeof(float4) > gathered->bytes) return fail(PT_ERR_RANGE, "gathered buffer too small");
    if ((size_t)width * height * sizeof(float4) > image->bytes) return fail(PT_ERR_RANGE, "image buffer too small");
    return PT_OK;
}

extern "C" int pt_assemble_stripes(pt_device_t d, pt_buffer_t gathered, pt_buffer_t image, int width, int height,
                                   int stripe_rows, int n_ranks, int slab_rows, pt_event_t ev)
{
    int rc = use_device(d);
    if (rc) return rc;
    if ((rc = assemble_check(d, gathered, image, width, height, stripe_rows, n_ranks, slab_rows))) return rc;
    if ((rc = enter_stream(d)) || (rc = event_begin(d, ev))) return rc;
    HIP_TRY(ptk_assemble_stripes((const float4*)gathered->dptr, (float4*)image->dptr, width, height, stripe_rows, n_ranks,
                                 slab_rows, d->stream));
    image->version++;
    return event_end(d, ev);
}

extern "C" int pt_tonemap_ppm(pt_device_t d, pt_buffer_t framebuffer, pt_buffer_t rgb_i32, size_t num_pixels, pt_event_t ev)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!framebuffer || !rgb_i32) return fail(PT_ERR_INVALID, "null buffer handle");
    if (framebuffer->dev != d || rgb_i32->dev != d) return fail(PT_ERR_INVALID, "buffer belongs to another device");
    if (num_pixels * sizeof(float4) > framebuffer->bytes) return fail(PT_ERR_RANGE, "framebuffer too small");
    if (num_pixels * 3 * sizeof(int32_t) > rgb_i32->bytes) return fail(PT_ERR_RANGE, "rgb buffer too small");
    if ((rc = enter_stream(d)) || (rc = event_begin(d, ev))) return rc;
    HIP_TRY(ptk_tonemap_ppm((const float4*)framebuffer->dptr, (int32_t*)rgb_i32->dptr, num_pixels, d->stream));
    rgb_i32->version++;
    return event_end(d, ev);
}

extern "C" int pt_profile_enable(pt_device_t d, int on)
{
    int rc = use_device(d);
    if (rc) return rc;
    d->prof_on = on != 0;
    return PT_OK;
}

extern "C" int pt_profile_reset(pt_device_t d)
{
    int rc = use_device(d);
    if (rc) return rc;
    if ((rc = enter_stream(d))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));
    for (int k = 0; k < PT_PROF_KINDS; ++k) d->prof_used[k] = 0;
    return PT_OK;
}

extern "C" int pt_profile_query(pt_device_t d, int kind, double* total_ms, uint64_t* launches)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (kind < 0 || kind >= PT_PROF_KINDS || !total_ms || !launches) return fail(PT_ERR_INVALID, "bad profile query");
    if ((rc = enter_stream(d))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));
    double sum = 0.0;
    for (size_t i = 0; i < d->prof_used[kind]; ++i) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, d->prof_pairs[kind][i].first, d->prof_pairs[kind][i].second));
        sum += ms;
    }
    *total_ms = sum;
    *launches = d->prof_used[kind];
    return check_traversal(d);
}

extern "C" int pt_profile_query_union(pt_device_t d, int kind, double* union_ms)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (kind < 0 || kind >= PT_PROF_KINDS || !union_ms) return fail(PT_ERR_INVALID, "bad profile query");
    if ((rc = enter_stream(d))) return rc;
    HIP_TRY(hipStreamSynchronize(d->stream));
    // launches are recorded in enqueue order and two lanes alternate: [start, stop] intervals relative to the first start,
    // swept in order of their starts
    const size_t n = d->prof_used[kind];
    std::vector<std::pair<double, double>> iv(n);
    for (size_t i = 0; i < n; ++i) {
        float a = 0.f, b = 0.f;
        if (i) HIP_TRY(hipEventElapsedTime(&a, d->prof_pairs[kind][0].first, d->prof_pairs[kind][i].first));
        HIP_TRY(hipEventElapsedTime(&b, d->prof_pairs[kind][0].first, d->prof_pairs[kind][i].second));
        iv[i] = { (double)a, (double)b };
    }
    std::sort(iv.begin(), iv.end());
    double total = 0.0, end = -1.0e300;
    for (auto& x : iv) {
        if (x.second <= end) continue;
        total += x.second - std::max(x.first, end);
        end = x.second;
    }
    *union_ms = total;
    return PT_OK;
}

// ------------------------------------------------------------------------------------------
// kernel registry + generic launcher
// ------------------------------------------------------------------------------------------
static const char* base_name(const char* path)
{
    const char* b = path;
    for (const char* p = path; *p; ++p)
        if (*p == '/' || *p == '\\') b = p + 1;
    return b;
}

extern "C" int pt_kernel_get(pt_device_t d, const char* file_name, const char* func_name, pt_kernel_t* out)
{
    if (!out) return fail(PT_ERR_INVALID, "out == NULL");
    *out = nullptr;
    if (!d || !file_name || !func_name) return fail(PT_ERR_INVALID, "null argument");
    std::string base = base_name(file_name);
    size_t dot = base.rfind('.');
    if (dot != std::string::npos) base.erase(dot);  // SELECT_KERNELPATH1 appends no extension; accept either
    for (int i = 0; i < KERNEL_COUNT; ++i)
        if (base == d->kernels[i].file && strcmp(func_name, d->kernels[i].func) == 0) {
            *out = &d->kernels[i];
            return PT_OK;
        }
    return fail(PT_ERR_NOT_FOUND, "no native kernel registered for (%s, %s)", file_name, func_name);
}

static int launch_generate_colors(pt_device_s* d, const pt_launch_arg* a, int nargs, long long n, pt_event_s* ev, float* ms_out)
{
    // __kernel void GenerateColors(tBuffer, matBuffer, gDst, const int4 cRes)  GenerateColors.cl:302-303
    if (nargs != 4 || !a[0].is_buffer || !a[1].is_buffer || !a[2].is_buffer || a[3].is_buffer || a[3].size != 16)
        return fail(PT_ERR_ARGS, "GenerateColors expects (buffer, buffer, buffer, int4)");
    pt_buffer_s *t = a[0].buffer, *m = a[1].buffer, *fb = a[2].buffer;
    if (!t || !m || !fb) return fail(PT_ERR_ARGS, "GenerateColors: null buffer argument");
    int32_t c[4];
    memcpy(c, a[3].data, 16);  // {W, H, frame, unused}: test/RaytraceTest.cpp:252-253
    if (c[0] < 1 || c[1] < 1 || c[2] < 0) return fail(PT_ERR_ARGS, "GenerateColors: invalid cRes {%d,%d,%d}", c[0], c[1], c[2]);
    long long npix = std::min<long long>(n, (long long)c[0] * c[1]);  // the kernel guards gid < W*H
    if (npix <= 0) return PT_OK;
    if (npix > 0xffffffffLL) return fail(PT_ERR_ARGS, "too many work-items");
    if (m->bytes < sizeof(PtRawMaterial)) return fail(PT_ERR_RANGE, "material buffer too small");

    // Deferral is invisible only while every buffer involved can be observed through this ABI alone.
    // Caller-owned (wrapped) memory, or memory whose device pointer has been handed out, may be read
    // after launch1D + waitForCompletion (clFinish in the reference) or rewritten before the next
    // launch: such launches execute at once, in order.
    const bool observable = !t->owned || !m->owned || !fb->owned || t->exposed || m->exposed || fb->exposed;
    bool immediate = !d->opt_batch || ev || d->opt_profile || observable;
    PendingFrames& p = d->pending;
    if (p.active && !(p.tris == t && p.mats == m && p.fb == fb && p.width == c[0] && p.height == c[1] &&
                      p.pixel_count == (uint32_t)npix && p.z_begin + p.z_count == c[2] && !immediate)) {
        int rc = flush_pending(d);
        if (rc) return rc;
    }
    if (!immediate) {
        // validate now what flush would reject later, so errors surface at the launch
        if ((size_t)36 * sizeof(PtRawTriangle) > t->bytes) return fail(PT_ERR_RANGE, "triangle buffer smaller than 36 triangles");
        if ((size_t)npix * sizeof(float4) > fb->bytes) return fail(PT_ERR_RANGE, "framebuffer smaller than the launch");
        if (t->dev != d || m->dev != d || fb->dev != d) return fail(PT_ERR_INVALID, "buffer belongs to another device");
        if (p.active) {
            p.z_count++;
        } else {
            p.active = true;
            p.tris = t; p.mats = m; p.fb = fb;
            p.width = c[0]; p.height = c[1];
            p.z_begin = c[2]; p.z_count = 1;
            p.pixel_count = (uint32_t)npix;
        }
        if (ms_out) *ms_out = 0.f;
        return PT_OK;
    }
    pt_render_params rp;
    memset(&rp, 0, sizeof rp);
    rp.width = c[0]; rp.height = c[1];
    rp.frame_begin = c[2]; rp.frame_count = 1;
    rp.max_bounces = 16; rp.num_triangles = 36;
    rp.num_materials = (int)std::min<size_t>(m->bytes / sizeof(PtRawMaterial), 0x7fffffff);
    rp.stripe_rows = 1; rp.n_ranks = 1; rp.rank = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (d->opt_profile) {
        int jrc = lanes_join(d);   // earlier renders are not part of this launch's time
        if (jrc) return jrc;
        d->main_dirty = true;      // ... and the lanes start behind e0
        HIP_TRY(hipEventCreate(&e0));
        HIP_TRY(hipEventCreate(&e1));
        HIP_TRY(hipEventRecord(e0, d->stream));
    }
    int rc = render_internal(d, t, m, fb, rp, (uint32_t)npix, nullptr, ev);
    if (d->opt_profile) {
        float ms = 0.f;
        if (!rc) rc = lanes_join(d);   // PROFILE_RETURN_TIME: the launch's duration as the handle's stream sees it
        if (!rc) {
            hipEventRecord(e1, d->stream);
            hipEventSynchronize(e1);
            hipEventElapsedTime(&ms, e0, e1);
        }
        hipEventDestroy(e0);
        hipEventDestroy(e1);
        if (ms_out) *ms_out = ms;
    } else if (ms_out) {
        *ms_out = 0.f;
    }
    return rc;
}

static int launch_fill(pt_device_s* d, const pt_launch_arg* a, int nargs, long long n, pt_event_s* ev)
{
    // FillKernel(int* dst, int value): the shim smoke kernel (test/main.cpp:128-151 uses one from a
    // TestKernel.cl the reference does not ship)
    if (nargs != 2 || !a[0].is_buffer || a[1].is_buffer || a[1].size != 4) return fail(PT_ERR_ARGS, "FillKernel expects (buffer, int)");
    pt_buffer_s* b = a[0].buffer;
    if (!b || b->dev != d) return fail(PT_ERR_ARGS, "FillKernel: bad buffer");
    int32_t v;
    memcpy(&v, a[1].data, 4);
    long long cnt = std::min<long long>(n, (long long)(b->bytes / 4));
    int rc = enter_stream(d);
    if (rc || (rc = event_begin(d, ev))) return rc;
    HIP_TRY(ptk_fill_i32((int32_t*)b->dptr, v, (int)std::min<long long>(cnt, 0x7fffffff), d->stream));
    b->version++;
    return event_end(d, ev);
}

static int launch_math(pt_device_s* d, const pt_launch_arg* a, int nargs, long long n, pt_event_s* ev)
{
    // MathKernel(const float* in, float* out): out[4i..4i+3] = sin, cos, pow(.,2.2f), pow(.,1/2.2f) of in[i]
    if (nargs != 2 || !a[0].is_buffer || !a[1].is_buffer) return fail(PT_ERR_ARGS, "MathKernel expects (buffer, buffer)");
    pt_buffer_s *in = a[0].buffer, *out = a[1].buffer;
    if (!in || !out || in->dev != d || out->dev != d) return fail(PT_ERR_ARGS, "MathKernel: bad buffer");
    long long cnt = std::min<long long>(n, std::min<long long>((long long)(in->bytes / 4), (long long)(out->bytes / 16)));
    int rc = enter_stream(d);
    if (rc || (rc = event_begin(d, ev))) return rc;
    HIP_TRY(ptk_math((const float*)in->dptr, (float*)out->dptr, (int)std::min<long long>(cnt, 0x7fffffff), d->stream));
    out->version++;
    return event_end(d, ev);
}

static int launch_fold_check(pt_device_s* d, const pt_launch_arg* a, int nargs, pt_event_s* ev)
{
    // FoldCheckKernel(ulong* out, int mode, uint first, ulong count): the fold kernel's short forms of pow and "/"
    // against the literal operations, operand by operand (csrc/pt_kernels.hip, pt_fold_check_kernel); the work-item
    // count of the launch is ignored, `count` operands are checked
    if (nargs != 4 || !a[0].is_buffer || a[1].is_buffer || a[1].size != 4 || a[2].is_buffer || a[2].size != 4 || a[3].is_buffer ||
        a[3].size != 8)
        return fail(PT_ERR_ARGS, "FoldCheckKernel expects (buffer, int, uint, ulong)");
    pt_buffer_s* out = a[0].buffer;
    if (!out || out->dev != d || out->bytes < 6 * sizeof(unsigned long long)) return fail(PT_ERR_ARGS, "FoldCheckKernel: bad buffer");
    int32_t mode;
    uint32_t first;
    uint64_t count;
    memcpy(&mode, a[1].data, 4);
    memcpy(&first, a[2].data, 4);
    memcpy(&count, a[3].data, 8);
    if (mode < 0 || mode > 4) return fail(PT_ERR_ARGS, "FoldCheckKernel: mode %d", mode);
    int rc = enter_stream(d);
    if (rc || (rc = event_begin(d, ev))) return rc;
    HIP_TRY(ptk_fold_check((unsigned long long*)out->dptr, mode, first, count, d->stream));
    out->version++;
    return event_end(d, ev);
}

extern "C" int pt_launch_2d(pt_device_t d, pt_kernel_t k, const pt_launch_arg* args, int nargs, int ntx, int nty, int lx,
                            int ly, pt_event_t ev, float* ms_out)
{
    int rc = use_device(d);
    if (rc) return rc;
    if (!k) return fail(PT_ERR_NOT_FOUND, "null kernel (Device::getKernel returned 0)");
    if (k < d->kernels || k >= d->kernels + KERNEL_COUNT) return fail(PT_ERR_INVALID, "kernel belongs to another device");
    if (nargs < 0 || nargs > PT_MAX_ARG_COUNT || (nargs && !args)) return fail(PT_ERR_ARGS, "bad argument list");
    for (int i = 0; i < nargs; ++i)
        if (!args[i].is_buffer && args[i].size > PT_MAX_ARG_SIZE) return fail(PT_ERR_ARGS, "constant %d larger than %d bytes", i, PT_MAX_ARG_SIZE);
    if (ntx < 0 || nty < 0 || lx < 1 || ly < 1) return fail(PT_ERR_ARGS, "bad launch geometry");
    if (ev && ev->dev != d) return fail(PT_ERR_INVALID, "event belongs to another device");
    long long n = (long long)ntx * nty;  // work-items the caller asked for (kernels guard the rounded-up tail)
    if (ms_out) *ms_out = 0.f;
    switch (k->id) {
    case KERNEL_GENERATE_COLORS: return launch_generate_colors(d, args, nargs, n, ev, ms_out);
    case KERNEL_FILL: return launch_fill(d, args, nargs, n, ev);
    case KERNEL_MATH: return launch_math(d, args, nargs, n, ev);
    case KERNEL_FOLD_CHECK: return launch_fold_check(d, args, nargs, ev);
    default: return fail(PT_ERR_NOT_FOUND, "unknown kernel id");
    }
}
