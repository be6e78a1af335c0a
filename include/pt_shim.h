/*
 * pt_shim.h -- C ABI of libptshim.so, the MI355X-native replacement for the Adl CL
 * device / buffer / kernel / launcher layer that the reference's RaytraceTest harness
 * drives (reference file:line cited per entry point; all paths relative to the
 * reference repository).
 *
 * Everything behind this boundary is HIP for gfx950.  There is no CPU fallback: every
 * entry point that needs the GPU fails with PT_ERR_NO_DEVICE / PT_ERR_HIP when none is
 * usable.  Plain pointers and sizes only; no C++ or torch types.
 *
 * Conventions
 *   - every int-returning function returns PT_OK (0) on success, a PT_ERR_* code
 *     otherwise; pt_last_error() gives the message for the calling thread.
 *   - handles are opaque.  Calls on one device handle are not thread-safe (the reference
 *     is single-threaded, one in-order queue per device: Adl/CL/AdlCL.cpp:215); different
 *     handles may be driven from different threads / processes (one per GPU).
 *   - all device work of a handle takes effect in call order, as on the reference's in-order
 *     cl_command_queue.  Everything but the fused renderer is enqueued on ONE stream (the handle's);
 *     pt_render_frames runs its trace / fold launches on two internal streams ("lanes") so that the
 *     next trace launch fills the machine while the current one drains, and the handle's stream is
 *     ordered behind them (by events, never a host wait) as soon as any other call needs its results.
 */
#ifndef PT_SHIM_H
#define PT_SHIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 2: PT_ERR_TRAVERSAL, PT_OPT_BVH_STACK_LIMIT, PT_OPT_RENDER_LANES, PT_STAT_BVH_*, pt_assemble_stripes_on, the staging ring
 *    (pt_device_reserve_staging, pt_device_workspace_memory), device-side hand-overs (pt_device_wait_stream, pt_device_wait_hip_event,
 *    pt_event_wait_on), pt_profile_query_union; option 3 (a kernel-variant switch of version 1) is accepted and ignored */
#define PT_SHIM_ABI_VERSION 2

enum pt_status {
    PT_OK = 0,
    PT_ERR_INVALID = 1,    /* bad handle / argument                                   */
    PT_ERR_NO_DEVICE = 2,  /* no usable gfx950 device (Adl: device with isValid()==0) */
    PT_ERR_OOM = 3,        /* allocation failed (Adl: m_size=0,m_ptr=0 + log)         */
    PT_ERR_HIP = 4,        /* a HIP runtime call failed                               */
    PT_ERR_NOT_FOUND = 5,  /* unknown kernel (Adl: getKernel returns 0)               */
    PT_ERR_ARGS = 6,       /* launch arguments do not match the kernel's signature    */
    PT_ERR_RANGE = 7,      /* offset/size outside a buffer                            */
    PT_ERR_TRAVERSAL = 8   /* an LBVH search was cut short (stack capacity or step budget): the pixels of the renders
                              enqueued since the last successful observation may be wrong and must be discarded.  The
                              reference's brute force (GenerateColors.cl:137-154) cannot skip a triangle, so this is an
                              error, never a silent approximation; it cannot occur with a hierarchy the library built
                              (csrc/pt_kernels.hip, PT_BVH_STACK) unless PT_OPT_BVH_STACK_LIMIT lowers the stack.
                              Renders are asynchronous, so the error is DEFERRED: the kernels raise a sticky word in
                              host-visible memory and the first call that observes the device afterwards reports it --
                              pt_sync, pt_event_wait / pt_event_elapsed_ns, a blocking pt_buffer_map, pt_profile_query,
                              or the next pt_render_frames (which then renders nothing).  Reporting clears the word. */
};

typedef struct pt_device_s* pt_device_t;
typedef struct pt_buffer_s* pt_buffer_t;
typedef struct pt_kernel_s* pt_kernel_t;
typedef struct pt_event_s* pt_event_t;

/* message of the last failing call on this thread ("" if none) */
const char* pt_last_error(void);
int pt_abi_version(void);

/* ---- library / device lifetime ---------------------------------------------------- */
/* adl::init(TYPE_CL) / adl::quit : Adl/Adl.cpp:39-58, 60-82 (CL: clewInit dlopen).      */
int pt_init(void);
void pt_quit(void);
/* DeviceUtils::getNDevices : Adl/Adl.cpp:84-110 */
int pt_device_count(void);
/* DeviceUtils::allocate(TYPE_CL, cfg{m_deviceIdx}) -> DeviceCL::initialize :
 * Adl/Adl.cpp:160-198, Adl/CL/AdlCL.cpp:68-271 (context + in-order queue + KernelManager) */
int pt_device_create(int device_idx, pt_device_t* out);
/* DeviceUtils::deallocate -> DeviceCL::release : Adl/Adl.cpp:200-208, AdlCL.cpp:273-280.
 * Flushes pending work, frees the device's kernels; returns PT_ERR_INVALID if buffers of
 * this device are still alive (the reference asserts used-memory == 0). */
int pt_device_destroy(pt_device_t dev);

enum pt_info_kind {
    PT_INFO_NAME = 0,    /* Device::getDeviceName    AdlCL.cpp:237-240 */
    PT_INFO_BOARD = 1,   /* Device::getBoardName                         */
    PT_INFO_VENDOR = 2,  /* Device::getDeviceVendor                      */
    PT_INFO_VERSION = 3  /* Device::getDeviceVersion  (used for the PPM file name, test/TestBase.h:45-51) */
};
int pt_device_info(pt_device_t dev, int kind, char out[128]);
uint64_t pt_device_max_alloc(pt_device_t dev);   /* Device::getMaxAllocationSize */
uint64_t pt_device_mem_size(pt_device_t dev);    /* Device::getMemSize           */
uint64_t pt_device_used_memory(pt_device_t dev); /* Device::getUsedMemory  (Adl.h:168) */
uint64_t pt_device_peak_memory(pt_device_t dev); /* Device::getPeakMemory  (Adl.h:170) */
/* Device memory the handle holds for ITSELF (not counted by pt_device_used_memory, which is the caller's buffers as in the
 * reference): the radiance staging ring, the prepared scene, the LBVH, the primary-ray masks, the side tables. */
uint64_t pt_device_workspace_memory(pt_device_t dev);
/* The fused renderer stages path radiance (12 B per sample) between its trace and fold kernels in a ring of TWO equal slots,
 * allocated ONCE per device handle: `bytes` is the size of the whole ring (0 = the default, 2 x 192 MiB: sixteen 1024^2
 * frames per slot).  pt_render_frames walks its frames through the ring in chunks of as many whole frames as a slot holds
 * and, once the ring exists, neither allocates, frees nor waits for the device.  Called implicitly with 0 by the first
 * render; calling it again waits for the device and replaces the ring.  An image of which ONE frame does not fit a slot
 * (more than 16.7 M pixels with the default) makes the render grow the ring to fit -- the only allocation a render can
 * still make; reserve enough beforehand to avoid it.  No reference counterpart (the reference keeps no staging: one launch
 * per frame, 32 B of framebuffer traffic per sample, GenerateColors.cl:314-321). */
int pt_device_reserve_staging(pt_device_t dev, size_t bytes);
int pt_device_num_cus(pt_device_t dev);          /* DeviceUtils::getNCUs               */

/* Plumbing, no reference counterpart: run this handle's work on an existing hipStream_t
 * (e.g. a torch stream, so torch.distributed collectives can be ordered after renders with
 * stream events).  NULL restores the handle's own (non-blocking) stream.  A caller whose
 * "stream handle 0" means the legacy default stream (torch's default stream reports 0) must
 * pass PT_STREAM_LEGACY, HIP's own sentinel hipStreamLegacy, so that it is not mistaken for NULL. */
#define PT_STREAM_LEGACY ((void*)1)
int pt_device_set_stream(pt_device_t dev, void* hip_stream);
void* pt_device_get_stream(pt_device_t dev);
/* Device-side hand-overs for a caller that keeps the handle on its OWN stream (what the N-rank driver does: renders of
 * consecutive images then overlap, which a shared stream would serialise).  No reference counterpart.
 *   pt_device_wait_stream    : the handle's later work starts after everything enqueued so far on hip_stream (a hipStream_t)
 *   pt_device_wait_hip_event : ... after a hipEvent_t the caller has recorded (e.g. torch.cuda.Event.cuda_event)
 *   pt_event_wait_on         : work enqueued later on hip_stream starts after the call the event was passed to has completed
 * None of them blocks the host. */
int pt_device_wait_stream(pt_device_t dev, void* hip_stream);
int pt_device_wait_hip_event(pt_device_t dev, void* hip_event);
int pt_event_wait_on(pt_event_t ev, void* hip_stream);

/* DeviceUtils::waitForCompletion(device) -> clFinish : Adl/Adl.cpp:210-213, AdlCL.cpp:282-285.
 * NOTE: with frame batching enabled (pt_device_set_option) this does not force deferred
 * GenerateColors frames to execute; observing a buffer, pt_flush or an event does.  Only launches
 * whose three buffers are all shim-allocated and whose device pointers were never handed out
 * (pt_buffer_device_ptr) are ever deferred -- memory the caller can reach behind this ABI
 * (pt_buffer_wrap, pt_buffer_device_ptr) gets clFinish semantics: the launch runs at once. */
int pt_sync(pt_device_t dev);
/* DeviceUtils::flush -> clFlush : AdlCL.cpp:303-306.  Submits deferred frames. */
int pt_flush(pt_device_t dev);

enum pt_option {
    /* 1 (default): consecutive GenerateColors launches on the same buffers with frame
     * indices z, z+1, z+2 ... are coalesced and executed as one fused multi-frame render
     * when a result is observed.  Pixel results are bit-identical either way.
     * 0: each launch executes immediately (the reference's behaviour). */
    PT_OPT_BATCH_FRAMES = 0,
    /* max frames traced per chunk of the fused renderer (radiance staging = 16 B x pixels x
     * frames per chunk).  0 = auto (fit staging in ~1/16 of device memory). */
    PT_OPT_CHUNK_FRAMES = 1,
    /* Device::toggleProfiling(PROFILE_RETURN_TIME) (Adl.h:171): launches synchronise and
     * return their duration in ms (AdlKernelUtilsCL.cpp:470-487). */
    PT_OPT_PROFILE_RETURN_TIME = 2,
    /* (3 selected between trace-kernel variants until only one was left: values 0 and 1 are accepted and ignored) */
    PT_OPT_RESERVED_3 = 3,
    /* conservative pass-1 filter of the closest-hit search, for A/B timing and parity tests:
     * 0 = the strongest the uploaded scene allows, 1..3 = independent triangles (pt_tri_pass1),
     * 4 = the packed shared-u filter when the scene is made of (a,b,c),(c,d,a) quads (two quads per
     * instruction, pt_quad3_pass1; same as 0).  All settings produce identical pixels. */
    PT_OPT_QUAD_FILTER = 4,
    /* closest-hit search: 0 = brute force below 512 triangles, LBVH from 512 on;
     * 1 = brute force (the reference's intersectWorld loop, GenerateColors.cl:137-154); 2 = LBVH
     * (built on the GPU when the scene is first rendered; scenes of >= 2 triangles).  The LBVH
     * applies the same exact triangle test to a conservative candidate set and resolves ties to the
     * lower index, as the reference's ascending loop does; see csrc/pt_bvh.hip for the one
     * theoretical caveat (rays within ~0.05 degrees of a triangle's plane). */
    PT_OPT_ACCEL = 5,
    /* measurement only (bench.py's roofline of LBVH runs): 1 = renders that pass a stats buffer and
     * take the LBVH use the tallying build of the search, which adds its work counters to
     * stats[PT_STAT_BVH_*].  Same pixels; slower; never the timed kernel.  Default 0. */
    PT_OPT_BVH_TALLY = 6,
    /* 1 (default): for quad scenes of up to 64 triangles on the brute-force path, every render first computes,
     * per pixel, a conservative candidate set of triangles its primary rays can meet (the camera is fixed:
     * GenerateColors.cl:265-272), and waves of fresh primary rays skip the pass-1 filter.  0 = off (A/B timing,
     * parity tests).  Identical pixels either way. */
    PT_OPT_PRIMARY_MASKS = 7,
    /* test hook of the LBVH's overflow report: the number of stack entries a ray's search may use (1..64; default 64,
     * which no hierarchy built by the library can exceed).  A search that needs more raises the sticky word behind
     * PT_ERR_TRAVERSAL (reported by the next call that observes the device; no render waits for the device to read it). */
    PT_OPT_BVH_STACK_LIMIT = 8,
    /* streams ("lanes") consecutive renders alternate between: 2 (default) = the first trace launch of render k+1 fills the
     * machine while the last launch of render k runs its paths out (about 0.3 ms of falling lane use,
     * profiles/r03/launch_overhead.txt), folds ordered by events so that every pixel folds its frames in ascending order
     * (GenerateColors.cl:314-321); 1 = one stream, every launch waits for the previous one (A/B timing).  Identical pixels. */
    PT_OPT_RENDER_LANES = 9,
    /* 1 (default): the trace launches of a render are CHECKPOINTED -- a launch ends the moment its work queue has handed out
     * the last batch, every wave saving the paths it still holds, and the render's next launch resumes them -- so that walking
     * a render through the bounded staging ring in many short launches costs 5 % over one long launch (measured: DESIGN.md S6), not a
     * tail of falling lane use per launch; a render of ONE chunk is one launch and takes no checkpoint; 0 = every launch runs its
     * paths out (A/B timing).  The LBVH kernel's checkpoint is one search deep: a stopping launch lets every lane finish its
     * current search and hands the shaded paths on.  Identical pixels either way. */
    PT_OPT_CHECKPOINT = 10,
    /* read-only (pt_device_get_option): LBVH builds this handle has made.  A scene is built once; a triangle buffer the caller can
     * write behind the ABI (pt_buffer_wrap, pt_buffer_device_ptr) is prepared again for every render, and built again only when the
     * checksum of its records has changed. */
    PT_OPT_BVH_BUILD_COUNT = 11
};
int pt_device_set_option(pt_device_t dev, int option, int64_t value);
int64_t pt_device_get_option(pt_device_t dev, int option);

/* ---- buffers ------------------------------------------------------------------------ */
/* Buffer<T>::allocate -> DeviceCL::allocate -> clCreateBuffer(READ_WRITE) :
 * Adl/Adl.inl:185-201, Adl/CL/AdlCL.inl:170-249.  bytes == 0 is allowed (no storage). */
int pt_buffer_alloc(pt_device_t dev, size_t bytes, pt_buffer_t* out);
/* Buffer<T>::setRawPtr (Adl.h:214): adopt device memory owned by the caller
 * (e.g. a torch tensor); pt_buffer_free does not free it. */
int pt_buffer_wrap(pt_device_t dev, void* device_ptr, size_t bytes, pt_buffer_t* out);
/* ~Buffer -> DeviceCL::deallocate : Adl.inl:153-165, AdlCL.inl:251-268 */
int pt_buffer_free(pt_buffer_t buf);
size_t pt_buffer_size(pt_buffer_t buf);
/* Buffer<T>::getInternalObject / m_ptr.  Submits deferred frames that touch the buffer and
 * switches frame batching off for it from now on (the caller can see the memory directly). */
void* pt_buffer_device_ptr(pt_buffer_t buf);
/* The buffer's device address as a VALUE (the facade's public Buffer<T>::m_ptr member, which in the reference holds
 * the opaque cl_mem: printing, identity).  It does not license access to the memory behind this ABI and changes
 * nothing; code that wants to read or write the memory itself must obtain the pointer with pt_buffer_device_ptr. */
void* pt_buffer_address(pt_buffer_t buf);
/* Buffer<T>::write / read (host) -> clEnqueueWrite/ReadBuffer : AdlCL.inl:297-340.
 * Asynchronous w.r.t. the host like the reference (non-blocking enqueue); the host range
 * must stay valid until pt_sync / the event.  ev may be NULL. */
int pt_buffer_write(pt_buffer_t dst, const void* host_src, size_t bytes, size_t dst_offset, pt_event_t ev);
int pt_buffer_read(pt_buffer_t src, void* host_dst, size_t bytes, size_t src_offset, pt_event_t ev);
/* Buffer<T>::write(Buffer&) / read(Buffer&) -> clEnqueueCopyBuffer : AdlCL.inl:270-295 */
int pt_buffer_copy(pt_buffer_t dst, pt_buffer_t src, size_t bytes, size_t dst_offset, size_t src_offset, pt_event_t ev);
/* Buffer<T>::getHostPtr(size=-1, blocking=false) -> clEnqueueMapBuffer(READ|WRITE) :
 * AdlCL.inl:434-445.  bytes == (size_t)-1 maps the whole buffer.  Returns a pinned host
 * staging range holding the buffer contents once the device has completed
 * (pt_sync, or blocking != 0); NULL on failure. */
void* pt_buffer_map(pt_buffer_t buf, size_t bytes, int blocking);
/* Buffer<T>::returnHostPtr -> clEnqueueUnmapMemObject : AdlCL.inl:447-451.
 * Copies the mapped range (the `bytes` of the matching pt_buffer_map, no more) back to the
 * device (asynchronously) and releases it. */
int pt_buffer_unmap(pt_buffer_t buf, void* host_ptr);

/* Page-locked host memory, so that pt_buffer_read / pt_buffer_write into it are truly asynchronous
 * (the progressive driver's double-buffered readback).  No Adl counterpart: the
 * reference only has the driver-owned mapping of getHostPtr. */
int pt_host_alloc(size_t bytes, void** out);
int pt_host_free(void* host_ptr);

/* ---- events (SyncObject : Adl/AdlKernel.h:45-54, AdlCL.inl:452-478) ------------------- */
int pt_event_create(pt_device_t dev, pt_event_t* out);
int pt_event_destroy(pt_event_t ev);
int pt_event_wait(pt_event_t ev);        /* DeviceUtils::waitForCompletion(SyncObject*) */
int pt_event_is_complete(pt_event_t ev); /* DeviceUtils::isComplete : 1 / 0, <0 on error */
/* Device::getExecutionTimeNanoseconds(SyncObject*) : AdlCL.cpp:508-517 */
int pt_event_elapsed_ns(pt_event_t ev, uint64_t* ns_out);

/* ---- kernels and launches -------------------------------------------------------------- */
/* Device::getKernel(fileName, funcName) -> KernelManager::query :
 * Adl/CL/AdlCL.cpp:490-493, Adl/AdlKernel.cpp:94-224.  The registry is static (kernels
 * are compiled into the library for gfx950; nothing is built at run time).  Only the
 * basename of file_name is significant, so the reference's "../test/ClKernels/GenerateColors"
 * resolves.  Unknown kernels -> PT_ERR_NOT_FOUND and *out = NULL (Adl returns 0).
 * Registered: ("GenerateColors","GenerateColors"), ("PtShimTest","FillKernel"),
 * ("PtShimTest","MathKernel"), ("PtShimTest","FoldCheckKernel") -- the last three are smoke/parity-test kernels. */
int pt_kernel_get(pt_device_t dev, const char* file_name, const char* func_name, pt_kernel_t* out);

#define PT_MAX_ARG_SIZE 64  /* Launcher::MAX_ARG_SIZE  Adl/AdlKernel.h:129 */
#define PT_MAX_ARG_COUNT 64 /* Launcher::MAX_ARG_COUNT Adl/AdlKernel.h:130 */

/* One positional kernel argument; mirrors Launcher::Args (Adl/AdlKernel.h:133-140):
 * buffers first-class, constants by value (<= 64 bytes). */
typedef struct pt_launch_arg {
    int32_t is_buffer;   /* 1: buffer, 0: by-value constant */
    int32_t read_only;   /* BufferInfo::m_isReadOnly        */
    uint64_t size;       /* constants: byte count           */
    pt_buffer_t buffer;  /* when is_buffer                  */
    unsigned char data[PT_MAX_ARG_SIZE];
} pt_launch_arg;

/* Launcher::launch2D -> LauncherCL::launch2D : Adl/AdlKernel.inl:186-196,
 * Adl/CL/AdlKernelUtilsCL.cpp:440-500.  launch1D(n, l) is launch2D(n, 1, l, 1).  The global
 * size is rounded up to a multiple of the local size as the reference does (:461-468) but,
 * unlike the reference kernel, the HIP kernels guard gid < n so a ragged n is safe.
 * ms_out (may be NULL) receives the duration when PT_OPT_PROFILE_RETURN_TIME is set, else 0. */
int pt_launch_2d(pt_device_t dev, pt_kernel_t kernel, const pt_launch_arg* args, int nargs,
                 int num_threads_x, int num_threads_y, int local_x, int local_y,
                 pt_event_t ev, float* ms_out);

/* ---- the fused hot path -------------------------------------------------------------------
 * One call = frames [frame_begin, frame_begin+frame_count) of GenerateColors
 * (test/ClKernels/GenerateColors.cl:302-322) over this device's share of the image,
 * bit-identical to frame_count successive reference launches.  Asynchronous: the call returns when the work is enqueued;
 * `ev` (may be NULL) completes when the framebuffer holds the last frame.  Consecutive calls overlap on the device as far as
 * their data allows (the folds of all calls form one chain, so renders into the same framebuffer fold in call order).
 *
 * Image sharding: the image's rows are dealt to n_ranks devices in stripes
 * of stripe_rows rows, round-robin; this device (rank) renders the rows r with
 * (r / stripe_rows) % n_ranks == rank, in ascending order, into a LOCAL framebuffer of
 * pt_local_rows(...) x width float4.  Seeds and camera rays use the GLOBAL pixel id
 * (GenerateColors.cl:305-308), so the assembled image equals the single-device image bit
 * for bit.  n_ranks = 1 gives the plain full-image framebuffer. */
typedef struct pt_render_params {
    int32_t width, height;   /* cRes.x, cRes.y */
    int32_t frame_begin;     /* first cRes.z */
    int32_t frame_count;
    int32_t max_bounces;     /* BOUNCES (16, GenerateColors.cl:5); 2 = build-defined "direct" mode */
    int32_t num_triangles;   /* NUM_TRIANGLES (36, :6) */
    int32_t num_materials;   /* bound for the material fetch at :239 */
    int32_t stripe_rows;     /* >= 1 */
    int32_t n_ranks;         /* >= 1 */
    int32_t rank;            /* 0 .. n_ranks-1 */
    int32_t reserved[6];     /* must be 0 */
} pt_render_params;

/* number of image rows owned by rank (see above) */
int pt_local_rows(int height, int stripe_rows, int n_ranks, int rank);

/* work counters accumulated by pt_render_frames when stats != NULL (uint64 each) */
enum {
    PT_STAT_SAMPLES = 0, PT_STAT_RAYS = 1,
    /* PT_OPT_BVH_TALLY renders only: box nodes entered and triangles tested (summed over the rays), and the phases
     * the waves executed for them (a node phase enters one node per participating lane, a triangle phase tests one
     * triangle per participating lane) */
    PT_STAT_BVH_NODES = 2, PT_STAT_BVH_TRIS = 3, PT_STAT_BVH_STEPS = 4 /* node phases */, PT_STAT_BVH_TRI_STEPS = 5,
    PT_STAT_BVH_MAX_STACK = 6 /* the deepest traversal stack any ray needed (a maximum, not a sum; capacity: 64) */,
    PT_STAT_CARRIED = 7 /* samples (paths under way + samples not yet started) that checkpointed launches handed to their successors */,
    /* PT_OPT_BVH_TALLY renders only: accepted closest hits at cos(incidence) < 1e-2, i.e. outside the range over which the LBVH's box
     * margin is argued conservative (csrc/pt_bvh.hip) -- the measured exposure of a scene to the LBVH's one theoretical caveat */
    PT_STAT_BVH_GRAZING = 8,
    PT_STAT_WORDS = 16
};

int pt_render_frames(pt_device_t dev, pt_buffer_t triangles, pt_buffer_t materials,
                     pt_buffer_t framebuffer, const pt_render_params* params,
                     pt_buffer_t stats /* may be NULL; PT_STAT_WORDS uint64, accumulated */,
                     pt_event_t ev);

/* Per-kernel device timing for measurement (bench.py "roofline"): when enabled, every launch of
 * the trace / fold kernels is bracketed by a HIP event pair on the device's stream.
 * pt_profile_query synchronises the stream and returns the summed duration and launch count
 * since the last reset.  Reference hook: Device::toggleProfiling + LauncherCL::launch2D's
 * stopwatch (Adl/CL/AdlKernelUtilsCL.cpp:470-487), which the reference test never enables. */
enum { PT_PROF_TRACE = 0, PT_PROF_FOLD = 1, PT_PROF_KINDS = 2 };
int pt_profile_enable(pt_device_t dev, int on);
int pt_profile_query(pt_device_t dev, int kind, double* total_ms, uint64_t* launches);
/* The time during which AT LEAST ONE launch of the kind was executing (the union of the launches' [start, stop] intervals):
 * with PT_OPT_RENDER_LANES 2 consecutive trace launches overlap -- the next one's first workgroups start while the previous
 * one's last paths drain -- so the sum of their durations counts that time twice; the union is the machine time they took. */
int pt_profile_query_union(pt_device_t dev, int kind, double* union_ms);
int pt_profile_reset(pt_device_t dev);

/* Scatter the gathered per-rank local framebuffers (n_ranks slabs of slab_rows x width
 * float4 each, slab k = rank k) into the full image (height x width float4). */
int pt_assemble_stripes(pt_device_t dev, pt_buffer_t gathered, pt_buffer_t image, int width,
                        int height, int stripe_rows, int n_ranks, int slab_rows, pt_event_t ev);

/* The same on a stream of the caller's (a hipStream_t, e.g. torch.cuda.Stream().cuda_stream) instead of the device
 * handle's: rank 0 of an N-rank render assembles image k behind its collective while render k + 1 already occupies the
 * handle's stream (the reference is single-device and has no counterpart).  The caller orders the stream
 * against the producer of `gathered` and the consumers of `image` (events); deferred frames are submitted first. */
int pt_assemble_stripes_on(pt_device_t dev, pt_buffer_t gathered, pt_buffer_t image, int width, int height,
                           int stripe_rows, int n_ranks, int slab_rows, void* hip_stream);

/* Output stage on the device: rgb8[i] = f2c(sqrtf(fb[i].xyz))
 * of test/RaytraceTest.cpp:78-83,280-285, written as int32 triplets (what "%d %d %d " prints). */
int pt_tonemap_ppm(pt_device_t dev, pt_buffer_t framebuffer, pt_buffer_t rgb_i32, size_t num_pixels,
                   pt_event_t ev);

#ifdef __cplusplus
}
#endif
#endif /* PT_SHIM_H */
