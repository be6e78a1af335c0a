// pt_kernels.h -- host-callable launchers of the gfx950 kernels (pt_kernels.hip).
// Internal to libptshim.so; the public boundary is include/pt_shim.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

// The reference's packed 64-byte device records (GenerateColors.cl:12-28).
struct PtRawTriangle { float p1[4], p2[4], p3[4]; int32_t id; char pad[12]; };
struct PtRawMaterial { float albedo[4], emissive[4]; float roughness; int32_t type; char pad[24]; };
static_assert(sizeof(PtRawTriangle) == 64 && sizeof(PtRawMaterial) == 64, "record layout");

// Triangle as the trace kernel consumes it: one 64-byte, 64-byte-aligned record =
// one s_load_dwordx16 per wave.  e1 = p2-p1, e2 = p3-p1, n = cross(e2,e1) are the values
// intersectTriangle recomputes on every call (GenerateColors.cl:92-93,123); computing them
// once per upload yields the same bits.
struct PtPrepTriangle {
    float p1[3];
    float e1[3];
    float e2[3];
    float pad0[3];
    float n[3];
    int32_t id;
};
static_assert(sizeof(PtPrepTriangle) == 64, "prep layout");

// LBVH internal node (pt_bvh.hip): the boxes of BOTH children and their links; a link is the child's
// node index, or 0x80000000 | triangle index for a leaf.  64 bytes = four 16-byte loads.
struct PtBvhNode {
    float lmin[3]; uint32_t link_l;
    float lmax[3]; uint32_t link_r;
    float rmin[3]; uint32_t pad0;
    float rmax[3]; uint32_t pad1;
};
static_assert(sizeof(PtBvhNode) == 64, "bvh node layout");

// The node the trace kernel walks: EIGHT children in 64 bytes -- one 64-byte sector of a cache line, four 16-byte loads
// (after Ylitie, Karras, Laine, "Efficient incoherent ray traversal on GPUs through compressed wide BVHs", 2017).  Round 3
// measured what the search is bound by (profiles/r03/lbvh_bottlenecks.txt): every 16-byte load of a wave whose 64 lanes
// read 64 different lines costs the CU's texture-address path 64 cycles -- one MORE such load per node visit cost 8.7 % --
// and the L2's miss path moves sectors, not requests (tools/ubench_gather: 64-byte records 111 G/s, 128-byte records 86 G/s).
// Round 2's node was 80 bytes in a 128-byte slot (five loads, both sectors); this one is 64 (four loads, one sector, two
// nodes per line: siblings lie next to each other and a ray that enters one often enters the other):
//   * which binary nodes of the radix tree become nodes here, and which of their descendants their (at most eight)
//     children are, is chosen by dynamic programming over surface-area costs (pt_bvh.hip);
//   * nodes and leaf records are 64-byte RECORDS of ONE array; the children of a node -- nodes and leaves alike -- are
//     stored consecutively in slot order from `base`: the child in slot s is record base + popcount((imask | lmask) &
//     ((1 << s) - 1)): no links, one base;
//   * a child sits in the slot whose three bits say on which side of the node's centre it lies (x = bit 0, y = bit 1,
//     z = bit 2; greedy assignment), so "slot XOR (ray direction's octant)" orders the children front to back for
//     every ray without a sort;
//   * boxes: 8 bits per coordinate in the node's own frame (origin = lower corner of the children's union, one
//     power-of-two step per axis), rounded OUTWARD and verified at build time by decoding (fma(q, step, origin) contains
//     the fp32 box); stored per axis (qlo[axis][slot], qhi[axis][slot]) so the ray's direction signs pick the near and
//     far planes of all eight children with four selects per axis; an empty slot holds an inverted box (255, 0) AND is
//     missing from both masks;
//   * the origin itself is 16 bits per axis on a grid over the scene (PtBvhGrid: origin = fma(org, grid step, grid min),
//     rounded DOWN and checked by decoding with the very expression the traversal uses; the boxes are quantised against
//     the decoded origin, so nothing is lost but up to one grid step -- 1 / 65 535 of the scene -- of the 8-bit range).
struct alignas(64) PtBvh8Node {
    uint16_t org[3];      // origin on the scene grid
    uint8_t ex[3];        // step exponents (biased as in binary32)
    uint8_t imask;        // slots that hold nodes
    uint8_t lmask;        // slots that hold leaves
    uint8_t pad;
    uint32_t base;        // record index of the first child
    uint8_t qlo[3][8];    // [axis][slot]
    uint8_t qhi[3][8];
};
static_assert(sizeof(PtBvh8Node) == 64, "bvh node layout");
struct PtBvhGrid { float gmin[3], gstep[3]; };  // the origins' grid (steps are powers of two: org * gstep is exact)

// A LEAF of the hierarchy is ONE triangle: a 64-byte record of the same array, three 16-byte loads.  (Leaves of four
// consecutive triangles of the Morton order were measured in round 2: the 10^6-triangle soup's leaf boxes grow 9x in
// cross-section, 184 instead of 5.4 triangle tests per ray, 35 instead of 86 Msamples/s.)
struct alignas(64) PtLeafTri {
    float p1[3], e1[3], e2[3];  // as in PtPrepTriangle
    uint32_t index;             // the triangle's index in the caller's buffer (ties in t go to the lowest)
    float pad[6];
};
static_assert(sizeof(PtLeafTri) == 64, "leaf record layout");

#define PT_TRACE_BATCH 256u    // largest number of samples per work-queue grab of a wave (PtTraceParams::batch)
#define PT_TRACE_THREADS 256   // 4 waves per workgroup (variant 1)
#define PT_LDS_TRI_STRIDE 12    // dwords per triangle record in the LDS copy (p1, e1, e2, 3 pad)
#define PT_LDS_TRI_MAX 256      // scenes up to this many triangles keep the copy (12 KiB per workgroup)

struct PtTraceParams {
    const PtPrepTriangle* tris;
    const PtRawMaterial* mats;
    float* rad;                   // the staging RING's two slots, [frames per slot][npix_local][3] path radiance max(L,0) each, 12 bytes per
    float* rad1;                  // sample.  A path's `fl` is its frame counted from the render's first (frame_begin - chunk_f0 is that frame's
                                  // absolute number); chunk c of the render is frames [c S, (c + 1) S) and goes to slot (c + ring_phase / S) % 2,
                                  // so a path CARRIED into the next launch (below) still stores to its own chunk's slot
    unsigned int* batch_counter;  // the work queue (PT_QUEUE_WORDS words: sharded counters + stop word); zero at the launch (the fold kernel that follows the trace launch on its stream resets it: PtFoldParams::reset_counter)
    unsigned long long* stats;    // may be null: [0] samples, [1] rays
    int32_t width, height;
    float inv_width, inv_height, aspect;  // 1.0f / W, 1.0f / H, (float)W / (float)H (IEEE, host-computed: GenerateColors.cl:266-267)
    int32_t frame_begin;          // first frame of this chunk (global frame index)
    int32_t frame_count;          // frames in this chunk
    uint32_t chunk_f0;            // ... and that first frame counted from the render's first (a multiple of the frames per ring slot)
    uint32_t slot_frames;         // S: frames per ring slot = frames per chunk (the render's last chunk may be shorter)
    uint32_t ring_phase;          // 0 or S: which slot the render's first chunk uses
    uint32_t ring_magic;          // floor(2^32 / 2S) + 1: (fl + ring_phase) % 2S by one v_mul_hi_u32 (exact below 65 536)
    int32_t max_bounces, ntri, nmat;
    int32_t stripe_rows, n_ranks, rank;
    uint32_t npix_local;
    uint32_t batches_per_frame, total_batches;
    uint32_t batch;               // samples per work-queue grab: 64, 128 or 256 (<= PT_TRACE_BATCH)
    float quad_delta1;            // quad mode 2 (pt_quad2_pass1): slack of the shared-u bounds
    float ray_radius;             // quad modes 2, 3: rays with |origin - eye|_inf above this keep every triangle
    const float* p1tab;           // quad mode 3: packed pass-1 table, PT_P1_STRIDE floats per pair of quads
    float p1_lo, p1_hi;           // quad mode 3: bounds of the shared numerator for the first / second triangle
    const PtBvh8Node* bvh;        // accel = BVH: the hierarchy's 64-byte records (nodes and leaves), root = record 0, every node's
                                  //              children consecutive
    int32_t bvh_records;          //              capacity of that array (2 x triangles): indices are checked against it
    PtBvhGrid grid;               //              the grid of the nodes' 16-bit origins
    const PtPrepTriangle* bigtab; // accel = BVH: prepared records of the nbig triangles kept out of the hierarchy (brute-force searched)
    const int32_t* bigidx;        //              their triangle indices, ascending
    int32_t nbig;
    const uint2* pmask;           // quad mode 3, <= 64 triangles: per local pixel the primary rays' candidate masks of the two
                                  // 32-triangle chunks (pt_primary_mask_kernel); null = run pass 1 for primary rays too
    unsigned int* bvh_flags;      // accel = BVH: the sticky word a search that was cut short raises (PT_BVH_FLAG_* bits): host memory mapped into
                                  //              the device's address space, written with plain system-scope stores, read and cleared by the host
    int32_t bvh_stack_limit;      //              stack entries a lane may use, <= PT_BVH_STACK (lower only to test the overflow report)
    // CHECKPOINTED launches (table trace kernels): a launch with carry_out set ends the moment its work queue has
    // handed out the last batch -- every wave, at its next fresh-phase boundary, stores what it still holds (its live paths,
    // the unstarted rest of its batch) to its region of `carry` and exits -- and the next launch of the render (carry_in_waves = the grid of this one, in
    // waves) resumes them beside its own work: no launch but a render's last (carry_out = 0: total_batches may be 0) has a tail
    // of waves running out of paths.  Which launch finishes a path never affects its result.
    uint32_t* carry;              // PT_CARRY_STRIDE_DW dwords per wave of the grid
    uint32_t carry_in_waves;
    uint32_t carry_out;
};
// a work queue: PT_QUEUE_SHARDS counters and, behind them, the stop word of checkpointed launches (one bit per shard found empty), each on
// a line of its own, PT_QUEUE_SHARD_WORDS apart (4 KiB + 128 B: whatever the address-to-channel map is, neighbours differ in both fields)
#define PT_QUEUE_SHARDS 8
#define PT_QUEUE_SHARD_WORDS 1056
#define PT_QUEUE_STOP_WORD (PT_QUEUE_SHARDS * PT_QUEUE_SHARD_WORDS)
#define PT_QUEUE_WORDS ((PT_QUEUE_SHARDS + 1) * PT_QUEUE_SHARD_WORDS)
#define PT_CARRY_RECORDS 64       // a wave stops with an empty pool and parks its (at most 64) live paths
#define PT_CARRY_STRIDE_DW (16 + PT_CARRY_RECORDS * 15)   // header (paths, pix, end, ring frame, absolute frame) + the parked-path record as arrays

struct PtFoldParams {
    const float* rad;   // [frame_count][npix_local][3]
    float4* fb;         // [npix_local] gamma-encoded running mean (GenerateColors.cl:314-321)
    uint32_t npix_local;
    int32_t frame_begin, frame_count;
    unsigned int* reset_counter;  // the work-queue counter of the trace launch whose chunk this is: set back to zero for its next user
    unsigned int* reset_counter2; // (may be null) a second one: the counter of the render's last, draining launch
};

// det_bound_bits: FOUR device words: [0] bit pattern of max_i (|e1|_1 * |e2|_1), [1] number of odd
// triangles whose e2 is not the exact negation of their predecessor's (0 = the scene is all quads),
// [2] bit pattern of max |vertex - eye|_inf, [3] number of odd triangles whose p1 is not their
// predecessor's p3 (0 = every pair is (a,b,c),(c,d,a))
// [4], [5] (one 64-bit word): checksum of the raw records
#define PT_PREP_WORDS 6
hipError_t ptk_prep_triangles(const PtRawTriangle* raw, PtPrepTriangle* out, int ntri, unsigned int* det_bound_bits,
                              hipStream_t s);
// quad mode 2: writes every odd record's pad0[0] = slack of its shared-u bound (needs the scene
// diameter bound D from word [2] of the first pass, hence a second tiny launch)
// p1tab (may be null): quad mode 3's table, PT_P1_STRIDE floats per pair of quads (pt_quad3_pass1)
hipError_t ptk_prep_quad_margins(PtPrepTriangle* out, int ntri, float diameter, float delta1, float* p1tab, hipStream_t s);
#define PT_P1_STRIDE 24  // floats per quad pair: nx ny nz e2x e2y e2z Kx Ky Kz dhi, each {quad 2p, quad 2p+1}, 4 pad
static inline size_t ptk_p1tab_floats(int ntri) { return (size_t)((ntri / 2 + 1) / 2) * PT_P1_STRIDE; }
// fills p.pmask (when not null) for the image geometry of p; needs p.p1tab (quad mode 3)
hipError_t ptk_primary_masks(const PtTraceParams& p, hipStream_t s);
// det_bounded: every triangle satisfies |e1|_1*|e2|_1 <= PT_DET_BOUND_MAX (short exact reciprocal valid)
// quads: 0 = independent triangles (pt_tri_pass1); 3 = ntri is even, every pair (2k, 2k+1) is a quad
//        (a,b,c),(c,d,a), the margins and the packed table p.p1tab are prepared (pt_quad3_pass1)
// bvh: traverse p.bvh instead of the brute-force two-pass search; then `quads` is about the table of the big triangles kept out
//      of the hierarchy (p.bigtab: 3 = made of quads, p.p1tab / p1_lo / p1_hi / quad_delta1 / ray_radius prepared for IT)
// tally: (bvh only) the measurement variant that adds the search's work counters to p.stats[2..5]
hipError_t ptk_trace(const PtTraceParams& p, int num_blocks, bool det_bounded, int quads, bool bvh, bool tally, hipStream_t s);
// out[k] = raw[bigidx[k]], k < nbig <= PT_BVH_BIG_MAX
hipError_t ptk_bvh_big_raw(const PtRawTriangle* raw, const int* bigidx, int nbig, PtRawTriangle* out, hipStream_t s);
static inline size_t ptk_bvh_record_count(int ntri) { return 2 * (size_t)(ntri > 0 ? ntri : 0); }  // < ntri nodes + ntri leaves
size_t ptk_bvh_temp_bytes(int ntri);
// prep: the prepared records of the same triangles.  bigtab[PT_BVH_BIG_MAX] / bigidx[PT_BVH_BIG_MAX] / *nbig_dev (device memory)
// receive the triangles kept OUT of the hierarchy (pt_bvh.hip: PT_BVH_BIG_DIV): their prepared records and indices, ascending
#define PT_BVH_BIG_MAX 64
// recs[ptk_bvh_record_count(ntri)] (device memory) receives the hierarchy the trace kernel walks, *grid_dev its origin grid
// *used_dev (device memory, may be null) receives the number of records in use
hipError_t ptk_bvh_build(const PtRawTriangle* raw, const PtPrepTriangle* prep, int ntri, PtBvh8Node* recs,
                         PtPrepTriangle* bigtab, int* bigidx, int* nbig_dev, PtBvhGrid* grid_dev, unsigned* used_dev, void* temp, size_t temp_bytes,
                         hipStream_t s);
#define PT_DET_BOUND_MAX 2.0e19f
#define PT_BVH_AUTO_MIN 512      // PT_OPT_ACCEL = 0 uses the BVH from this many triangles on
hipError_t ptk_fold(const PtFoldParams& p, hipStream_t s);
hipError_t ptk_assemble_stripes(const float4* gathered, float4* image, int width, int height, int stripe_rows,
                                int n_ranks, int slab_rows, hipStream_t s);
hipError_t ptk_tonemap_ppm(const float4* fb, int32_t* rgb, size_t npix, hipStream_t s);
hipError_t ptk_fill_i32(int32_t* dst, int32_t value, int n, hipStream_t s);
hipError_t ptk_math(const float* in, float* out, int n, hipStream_t s);
// the fold kernel's short forms against the literal operations (pt_fold_check_kernel); out: 6 counters
hipError_t ptk_fold_check(unsigned long long* out, int mode, unsigned first, unsigned long long count, hipStream_t s);
// dynamic LDS of variant 1: the triangle table (scenes up to PT_LDS_TRI_MAX) + one camera-ray slot
// per sample of every wave's current batch
size_t ptk_trace_lds_bytes(int ntri);
int ptk_trace_blocks_per_cu(int ntri);
int ptk_trace_bvh_blocks_per_cu(void);
size_t ptk_trace_bvh_lds_bytes(void);
