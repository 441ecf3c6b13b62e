// Soft-silhouette rasteriser for gfx950: forward, backward and fused forward+L1+backward.
//
// Replaces (reference): smal_fitter/p3d_renderer.py:41-52,142-146 - MeshRasterizer(bin_size=0,
// faces_per_pixel=100, blur_radius=log(1/1e-4-1)*1e-4, perspective-correct, clipped barycentrics) +
// SoftSilhouetteShader (sigmoid_alpha_blend) - and, in the fused entry point, the silhouette L1 term of
// SMALFitter.forward (fitter.py:332-333) with its gradient.  The per-(pixel,face) arithmetic restates
// pytorch3d 0.7.x (un-vendored): CheckPixelInsideFace / RasterizeMeshesBackward / geometry_utils.
//
// Design (see DESIGN.md "raster"):
//  * The reference visits all F faces for all S^2 pixels and stores (S,S,K) fragments in HBM (157 MB / image at 256^2).
//  * k_raster_setup (one workgroup per image): per-face validity + blurred bbox in 8x8-pixel tile units (4 x u8 packed),
//    the union of those boxes per 64 consecutive faces, per-face vertex-depth range, a tile-occupancy bitmap in LDS, and
//    the compacted list of touched tiles appended to a global work list.  Untouched tiles are never visited (their
//    silhouette is 0).
//  * k_raster_dense (persistent single-wave workgroups, work item = one 8x8 tile).  Every (face, pixel) pair is
//    evaluated ONCE, by a lane that exists only for pairs inside the face's pixel box; every later step is a dense sweep
//    (lane = record) over the records that pass produced.  An earlier design (lane = pixel, each face broadcast to the 64
//    pixels of the tile, three re-evaluating passes, K smallest depths in a sorted register array) spent ~75 % of its
//    lanes on pairs that do not exist and 100 v_med3 per face on the selection; it is gone.
//      list    faces whose tile box contains the tile (via the 64-face group boxes), with their nearest vertex depth; tiles
//              that may truncate (list > K) order it NEAR TO FAR by the first radix digit of that depth (counting sort).
//      pass 1  lane = pair.  Per DCHUNK-face chunk: two lanes per face stage its record and its pixel box inside the tile
//              (shrunk to the pixels that are still open), a prefix sum lays the boxes end to end, and the wave sweeps that
//              pair list 64 at a time (a start-bit map gives each pair its face with two v_mbcnt and two ds_bpermute; face
//              record and pixel coordinates are gathered from LDS).  Accepted pairs are ballot-compacted into the
//              workgroup's record stream in global memory (reused for every tile): 12 bytes {depth, pixel | list position
//              | inside | edge, signed squared distance}, one array of structures.  The first radix digit of every depth
//              (64 buckets, saturating 8-bit counts) is histogrammed per pixel on the way.  CLOSING: a record's depth is
//              at least its face's nearest vertex depth, so once the walk reaches digit d the per-pixel counts of digits
//              < d are final; a pixel holding >= K records there takes no further record, and when no pixel is open the
//              rest of the list is skipped.
//      blend   one sweep over the records.  A record of a pixel with <= K candidates, or whose first digit lies below the
//              digit that holds the pixel's K-th depth, is kept for certain: log2 of its factor is added to the pixel's
//              sum (fp64 LDS atomics).  A record inside that digit goes on to a compact stream {key, meta, log} and has
//              its second digit counted; one above it is dropped.
//      select  (only where a pixel has more than K candidates) the remaining digits by radix select over the compact
//              stream, SEL_BITS per sweep, per-pixel histograms in LDS; the stream shrinks in place with every sweep.
//              Exact, including the number of faces tied at the threshold; when a tie group straddles K, further sweeps
//              select on the face id (fetched through the list only for records at the threshold).  A last sweep adds the
//              logs of the records that made it; alpha = exp2(sum) (the fp32 product and the fp64-accumulated log-sum are
//              both ~1e-6 relative from the exact product).
//      pass 3  gradient of every kept record into per-face LDS accumulators ((x, y) packed as two 32-bit fixed-point numbers
//              in one 64-bit word), per 64 faces; the closest point on the record's edge is recomputed from a table of the
//              group's vertices.  Flush: one global atomic per touched vertex - two floats, or, from 64 images per launch
//              on, one packed 64-bit integer decoded afterwards (k_unpack_dndc); these atomics execute memory-side.
//  * A tile whose records would not fit the stream (REC_CAP) is processed in sub-tiles: power-of-two runs of its 64 pixels,
//    halved until pass 1 fits.  A single pixel always fits because F <= REC_CAP is required on the host.
//  * Deviation from the reference kept on purpose: the K faces a truncated pixel keeps are the K smallest by (depth, face
//    id).  The reference's unsorted-queue eviction picks among equal depths by visiting history (measured effect on the
//    L1 loss: ~1e-5 relative at K = 100; tests/test_gpu_kernels.py checks this rule exactly against the oracle's
//    select_mode(1) and the faithful queue within the documented tolerance).
#include "common.h"

#define TILE 8
#define DCHUNK 32           // faces staged per chunk
#define FREC 28             // floats per staged face record
#define FSTR 28             // its stride in LDS: 112 bytes = 28 banks, so the 16-byte rows of 16 consecutive faces start in 16 different
                            // bank quads (a 128-byte stride would put the same row of every face in the same banks)
#define K_EPS 1e-8f
#define ALPHA_GRAD_EPS 1e-12f  // pixels whose transmittance is below this contribute no gradient
#ifndef SEL_BITS
#define SEL_BITS 5
#endif
//      SEL_BITS            // radix-select digit width (two 16-bit counts per LDS word, 16 words per pixel)
#ifndef SEL1_BITS
#define SEL1_BITS 6
#endif
//      SEL1_BITS           // width of the FIRST digit, the one pass 1 counts and the closing rule works with: 64 buckets in the same
                            // 16 words per pixel as four 8-bit counts that stop at SAT8 (a count only ever matters up to K <= 128)
#define SAT8 160u           // a byte takes no further increment from here on; at most 63 more arrive with the instruction that crosses it
#ifndef DGROUP
#define DGROUP 4            // 64-record rows per buffer in the dense walks (two buffers)
#endif
#ifndef KGROUP
#define KGROUP 4            // 64-key rows per buffer in the selection sweeps (two buffers)
#endif
#ifndef SELR
#define SELR 6              // compact records a lane holds once the selection runs in registers (the stream is then at most SELR * 64 long)
#endif
#ifndef REC_CAP
#define REC_CAP 65536       // pair records one (sub-)tile may produce
#endif
#define REC_PAD 64          // slack so that a clamped read stays inside the allocation
#ifndef RESIDENT_PER_CU
#define RESIDENT_PER_CU 16
#endif
//      RESIDENT_PER_CU     // single-wave workgroups per CU: what 128 VGPRs and 9.9 KB of LDS per workgroup allow (measured 10 ... 14: every
                            // further workgroup still shortens the launch)

#ifndef WAVES_PER_SIMD
#define WAVES_PER_SIMD 4     // what the tile kernel's register budget is set for: RESIDENT_PER_CU / 4
#endif
enum { MODE_FWD = 0, MODE_BWD = 1, MODE_FUSED = 2 };

// Instrumentation hooks (phase timers, work counters, cut-off / wrap experiments): empty in libsmilfit.so.  `make variant` builds the
// instrumented libraries of tools/dbg with -DSMIL_INSTRUMENTED, which pulls their definitions from tools/dbg/raster_hooks_dbg.h; this
// translation unit itself holds no experiment code, and smil_version() says which kind of build a library is.
#include "raster_hooks.h"

// Work items (touched tiles) are queued in four cost classes by the number of (face, pixel) pairs the tile will evaluate
// (the sum of its faces' pixel boxes), and handed out heaviest class first: a persistent kernel whose longest items take
// a fifth of the whole launch must not start them last.
// Per partition two arrays of ceil(N / N_PARTS) * tiles entries hold two classes each (one filled from the front, one
// from the back).
#define N_CLASSES 4
#ifndef PACKED_MIN_IMAGES
#define PACKED_MIN_IMAGES 64  // from this many images per launch the fused entry point packs its gradient atomics (see image_fx_scale)
#endif
#ifndef CLASS_T0
#define CLASS_T0 65536        // class 0 can be dealt out in pieces (SPLIT0_LOG)
#endif
#define CLASS_T1 16384
#define CLASS_T2 4096
#ifndef SPLIT0_LOG
#define SPLIT0_LOG 0          // log2 of the pieces every class-0 tile is dealt out in (0: whole; with near-to-far lists and closing the
#endif                        // tiles with the longest lists finish early, and pieces only repeat their list walk: measured 2 -> 0: mouse -9 %)
#define COUNT_TILES_MAX 4096  // per-tile cost / entry counts and list cursors live in LDS (12 bytes per tile); larger images (S > 512) queue
                              // everything in the last class and build their lists in the tile kernel
// XCD-aware dealing.  Each of the 8 XCDs of an MI355X has its own 4 MB L2, and the tiles of one image read the same
// per-image tables (projected vertices, face tile boxes, depth ranges: ~180 KB on STICK).  Images are therefore dealt to
// N_PARTS work-list partitions (image % N_PARTS); a workgroup drains the partition of the XCD it runs on first
// (HW_REG_XCC_ID - placement is whatever the dispatcher chose, only speed depends on it) and then helps the others, so an
// image's tables are fetched into one L2 instead of eight while the launch is busy, and the tail still balances.
#define N_PARTS 8
struct RasterCounters {
    unsigned int n_class[N_PARTS][N_CLASSES];
    struct { unsigned int next, pad[15]; } deal[N_PARTS];  // one cache line per partition's cursor
    unsigned int straddling;  // faces that cross z_clip in this launch: cut at the plane (smil_raster_stats) ...
    unsigned int unclipped;   // ... except these: beyond the per-image clip tables, rendered whole or dropped
    unsigned int tie_pixels;  // (tie_rule 1) pixels left to k_raster_tie_replay
    unsigned int tie_next;    // ... and its ticket counter
};

// The tile kernel's dealing policy, shared by the kernel and the host: with fewer tiles than workgroup slots every tile is dealt out
// as 2, 4 or 8 runs of pixels (round 4, from a sweep over 1 ... 64 images x workgroups per CU x pieces, profiles/r4_small_launches.txt:
// the launch is fastest with ~2.3 pieces per WORKING workgroup and about 1.2 pieces per resident slot in all), and only
// max(slots / 8, pieces x 7 / 16) workgroups take part.
__host__ __device__ __forceinline__ unsigned int deal_split_log(unsigned int n_items, unsigned int slots) {
    return n_items * 8u <= slots * 5u / 8u ? 3u : (n_items * 4u <= slots * 5u / 4u ? 2u : (n_items * 2u <= slots * 5u / 4u ? 1u : 0u));
}
__host__ __device__ __forceinline__ unsigned int deal_working(unsigned int n_items, unsigned int split_log, unsigned int slots) {
    const unsigned int w = (n_items << split_log) * 7u / 16u;
    return w > slots / 8u ? w : slots / 8u;
}

struct Rec3 { uint32_t a, b, c; };  // one 12-byte record: loaded / stored as one dwordx3
// Per list position of the current tile, left by pass 1 (which has them in registers) for pass 3: the face's projected vertices
// and its vertex ids.  Pass 3 used to fetch them per group of 64 faces through the chain list -> face -> vertex: three dependent
// memory round trips per group and 28 % of pass 3 (profiles/r4_pass3_timers.txt).  Vertices as three float2 arrays, see stage_faces.
struct TriIds { int a, b, c; };

// clip_faces (pytorch3d renderer/mesh/clip.py, as MeshRasterizer applies it with z_clip_value = znear / 2; the reference leaves that
// default on, p3d_renderer.py:36-47): a face with one or two vertices nearer than z_clip is cut at the plane and its front part
// (one triangle, or a quadrilateral as two) rendered instead.  Such faces are rare (the mesh must reach the camera), so they are
// handled beside the mesh, per image: up to CLIP_FX front-part triangles get face ids from FP = F rounded up to 64 on, their
// new vertices (on the plane) vertex ids from V on, both in small side tables; every fetch of a face's vertex indices or of a
// vertex's coordinates / gradient row goes through one compare that picks the table.  A new vertex is
// c_a xy[a] + c_b xy[b] of the cut edge's end points (interpolated in view space); its gradient goes back to them with the
// coefficients held constant (k_clip_backward).  Faces beyond the tables' capacity are rendered as before and counted.
#define CLIP_CUTS 1024           // cut faces per image (ONE capacity: each owns two front-part triangle slots and two new-vertex slots;
                                 // round 4: 256 -> 1024 - with the camera inside the 17 420-face mouse a third of the fuzzed scenes had exceeded 256)
#define CLIP_FX (2 * CLIP_CUTS)  // front-part triangles per image
#define CLIP_VX (2 * CLIP_CUTS)  // new vertices per image
struct ClipTables {
    float *xv;          // (N, CLIP_VX, 3) new vertices (x_ndc, y_ndc, z_clip)
    int *xf;            // (N, CLIP_FX, 3) vertex ids of the front-part triangles (>= V: new vertices)
    int2 *xsrc;         // (N, CLIP_VX) end points (a, b) of the edge a new vertex lies on
    float2 *xcoef;      // (N, CLIP_VX) (c_a, c_b)
    float *xg;          // (N, CLIP_VX, 2) gradient rows of the new vertices (same representation as d_ndc)
    uint32_t *xcount;   // (N) new vertices of the image
};
__host__ __device__ __forceinline__ int faces_padded(int F) { return (F + WAVE - 1) / WAVE * WAVE; }
// vertex ids of face f of image n / coordinates of vertex i of image n, through the clip tables
__device__ __forceinline__ int face_vertex(const int *__restrict__ faces, const int *__restrict__ xf_n, int F, int f, int k) {
    return f < F ? faces[3 * f + k] : xf_n[3 * (f - faces_padded(F)) + k];
}
__device__ __forceinline__ const float *vertex_ptr(const float *__restrict__ vn, const float *__restrict__ xv_n, int V, int i) {
    return i < V ? vn + 3 * i : xv_n + 3 * (i - V);
}

struct RasterArgs {
    const float *verts_ndc;  // (N,V,3)
    const int *faces;        // (F,3)
    const uint32_t *tbox;    // (N,F) tile box of every face
    const uint32_t *gbox;    // (N, ceil(F/64)) union of the tile boxes of 64 consecutive faces
    const uint4 *items;      // work lists of {tile code, first list entry, entries (0xFFFFFFFF: build the list here), depth extent of the
                             // image's deepest face}, per partition q at 2 q cap: [0, cap) classes 0 (front) / 1 (back), [cap, 2 cap) classes 2 / 3
    uint32_t item_cap;       // entries of one array of ONE partition: ceil(N / N_PARTS) * tiles
    const float2 *fzr;       // (N,F) nearest / farthest vertex depth of every face
    RasterCounters *ctr;
    int N, V, F, S, tiles_x, K;
    int FT;                  // rows of the per-image face tables: F rounded up to 64 + CLIP_FX (front parts of cut faces)
    ClipTables clip;
    float blur, sqrt_blur, inv_sigma, inv_sigma_log2e;
    // outputs / inputs per mode
    float *sil;              // (N,S,S) FWD (or optional in FUSED)
    const float *grad_sil;   // BWD
    const float *target;     // FUSED (fp32 targets) ...
    const uint8_t *target_u8; // ... or binary {0,1} targets stored as bytes
    const float *pix_scale;  // FUSED (N,)
    const float *img_bound;  // (N,) setup kernel: 0.4 x valence x largest face box (pixels), the geometric part of the bound on a vertex's gradient
    int packed;              // FUSED: d_ndc is accumulated as (x, y) fixed point packed in 64 bits (one memory-side atomic per vertex, not two)
    float *loss_img;         // FUSED (N,)
    unsigned long long *loss_acc;  // FUSED (N,) the tiles' loss terms as 2^-32 fixed point: integer adds, the same bits in any order of
                             // arrival (round 5; a float atomic per tile before); k_clip_backward adds the sum to loss_img afterwards
    float *d_ndc;            // (N,V,2)
    // scratch per resident workgroup
    const uint2 *lists;      // (N, list_cap) tile lists binned by the setup kernel: {face id, bits of its nearest vertex depth}
    uint32_t list_cap;
    uint2 *slist;            // the current tile's faces when it builds its list itself (ascending id; same entry layout) ...
    uint32_t *slist2;        // ... and the ids the tile walks: near to far by the first radix digit of that depth when the tile may
                             // truncate (the sort reads the depths it needs from slist instead of gathering them per face)
    uint32_t *scfirst;       // F / DCHUNK + 2: first record of every chunk from the 128th on (the others live in registers)
    float2 *sxy;             // (3, list_stride) projected vertices v0 / v1 / v2 of the tile's faces by list position ...
    TriIds *sid;             // (list_stride) ... and their vertex ids
    // record streams, REC_CAP + REC_PAD entries each (structure of arrays: every sweep reads only what it needs)
    // pair records, 12 bytes each in ONE stream per workgroup (an append or a sweep step then touches one contiguous run of
    // memory instead of three): {depth bits, pixel | list position << 6 | inside << 22 | closest edge << 23, signed squared
    // distance to the closest edge (pass 3 recomputes the closest point itself)}
    Rec3 *srec;
    // records that survive the first selection digit: {key = depth bits - tile minimum, meta, log2 of the blend factor}
    Rec3 *crec;
    int list_stride, n_cf;   // entries of slist / scfirst per workgroup
    unsigned int slots;      // resident workgroup slots of the device (the dealing policy's yardstick; gridDim.x <= slots)
    SmilClipDepth cd;        // where k_clip_backward leaves the depth gradients of cut edges' end points (range == NULL: nowhere)
    int image0;              // index of the call's first image in the caller's batch (cd.range)
    int tie_rule;            // SmilRasterSettings.tie_rule (0: K smallest by (depth, face id); 1: the reference's queue, k_raster_tie_replay)
    unsigned long long *tie_mask;  // tie_rule 1: per work item (same index as `items`) the pixels of the tile whose K-th depth is a
                             // tie group that K cuts through: left out by the tile kernel, rendered by k_raster_tie_replay
    HOOK_ARGS_FIELDS         // (instrumented builds: counter buffer, cut-off phase, forced split)
};

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// Packed gradient accumulation (fused entry point, large launches).  The flush of pass 3 goes to memory-side atomics (the
// per-XCD L2s forward every atomic), whose cost is proportional to their number: (x, y) of a vertex travel as two 32-bit
// fixed-point numbers in ONE 64-bit integer atomic instead of two float atomics.  The scale is a power of two per image,
// chosen so that no vertex component can overflow: |sum| <= img_bound * |pix_scale| / sqrt(sigma) (see k_raster_setup) maps into
// [2^29, 2^30].  Integer sums are order independent: the gradient becomes reproducible bit for bit.  k_unpack_dndc turns the
// buffer into the (N,V,2) floats the interface promises, in place.
__device__ __forceinline__ float image_fx_scale(float img_bound, float pix_scale, float inv_sigma) {
    const float bound = img_bound * fabsf(pix_scale) * sqrtf(inv_sigma);
    return (bound > 0.f && bound < 3.0e38f) ? exp2f(fminf(29.0f - floorf(log2f(bound)), 100.0f)) : 0.f;
}

// Element i of a per-workgroup stream: uniform base pointer + 32-bit byte offset, which hipcc turns into the SGPR-base /
// VGPR-offset form of the global load / store (a 64-bit address per lane costs two extra VALU instructions per access).
// (12-byte elements: the index must be below 2^24, so that the full-rate 24-bit multiply is exact; left to itself hipcc emits the
// quarter-rate v_mul_lo_u32, also for the shift-and-add spelling.  ONLY for per-workgroup / per-image arrays whose length the
// host bounds - REC_CAP + REC_PAD records, list_stride entries, both checked in raster_common.  An array whose index grows with
// the number of images must not come through here: a round-4 experiment stored its 12-byte work items this way, the index
// reaches 16 * ceil(N / 8) * tiles = 18.9e6 > 2^24 at 2 304 images @512^2, partition 7's items landed 2^24 elements early and the
// tile kernel read stale words as work items - the GPU abort of gpurun_out/r4/tests_itb.txt, DESIGN.md section 8.  The shipped
// work items are 16 bytes and plainly indexed.)
template <typename T>
__device__ __forceinline__ uint32_t byte_offset(uint32_t i) {
    if (sizeof(T) == 12) {
        uint32_t r;
        asm("v_mul_u32_u24 %0, %1, 12" : "=v"(r) : "v"(i));
        return r;
    }
    return i * (uint32_t)sizeof(T);
}
template <typename T>
__device__ __forceinline__ T &at(T *base, uint32_t i) {
    i = HOOK_WRAP_IDX(i);
    return *reinterpret_cast<T *>(reinterpret_cast<char *>(base) + byte_offset<T>(i));
}
template <typename T>
__device__ __forceinline__ const T &at(const T *base, uint32_t i) {
    i = HOOK_WRAP_IDX(i);
    return *reinterpret_cast<const T *>(reinterpret_cast<const char *>(base) + byte_offset<T>(i));
}

// inclusive wave64 prefix sum in DPP (row_shr within 16-lane rows, then row_bcast across rows)
__device__ __forceinline__ int wave_scan_add(int x) {
#define SCAN_STEP(ctrl, rows) { x += __builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xF, false); }
    SCAN_STEP(0x111, 0xF) SCAN_STEP(0x112, 0xF) SCAN_STEP(0x114, 0xF) SCAN_STEP(0x118, 0xF)
    SCAN_STEP(0x142, 0xA) SCAN_STEP(0x143, 0xC)
#undef SCAN_STEP
    return x;
}

// ---------------------------------------------------------------------------------------------
// setup: per-face tile boxes + touched-tile work list
// ---------------------------------------------------------------------------------------------
#ifndef SETUP_THREADS
#define SETUP_THREADS 1024
#endif
#ifndef LIST_CAP_PER_FACE
#define LIST_CAP_PER_FACE 8   // (tile, face) list entries an image may have per face at S <= 256 (a face's blurred box covers ~4 tiles there,
                              // ~8 at 512^2: the blur radius is a fixed fraction of the image); doubled above 256
#endif
struct SetupArgs {
    ClipTables clip;
    const float *verts_ndc; const int *faces;
    uint32_t *tbox, *gbox; uint4 *items; uint32_t item_cap; float2 *fzr;
    RasterCounters *ctr;
    int V, F, S, tiles_x; float sqrt_blur, z_clip;
    float *d_ndc_zero; const float *loss_src; float *loss_dst; unsigned long long *loss_acc; float *img_bound; int max_valence;
    float *dndc_scale; const float *pix_scale; float inv_sigma; int packed;
    uint2 *lists;       // (N, list_cap) binned tile lists: {face id, bits of its nearest vertex depth} (8 bytes: the farthest depth only ever fed the
                        // tile's depth range, and farthest <= nearest + the image's largest face extent bounds that as well)
    uint32_t list_cap;  // entries per image (0: no binning)
    uint32_t *cd_counter;  // SmilClipDepth.counter of a gradient call with image0 == 0: reset here (block 0), or NULL
    int copies;         // (round 5) per-tile counters / list cursors are kept in this many copies (1, 2 or 4: what fits 48 KB of LDS), a
                        // face using copy (face id % copies): consecutive faces hit the same tiles, and LDS atomics of one wave
                        // instruction on ONE address execute one after the other - the two atomic passes were two thirds of this kernel
};
// One workgroup per image.  Pass 1: per face validity, blurred pixel box -> tile box, depth range; per covered tile ONE LDS
// atomic adds the face's cost and list entry (64-bit: entries << 32 | cost).  Then the touched tiles go to the work lists by cost
// class and - new in round 3 - the faces are BINNED: a prefix sum over the tiles' entry counts lays the image's tile lists
// end to end, and pass 2 walks the faces again and appends each to the lists of the tiles its box covers.  The tile kernel
// then starts from its list instead of scanning the tile boxes of every 64-face group that reaches its tile (build_list:
// 13 % of the tile kernel in round 2).  Images whose lists exceed list_cap keep the old way.
__global__ void __launch_bounds__(SETUP_THREADS, 8) k_raster_setup(SetupArgs q) {  // (two blocks per CU: <= 64 VGPRs)
    __shared__ uint32_t s_maxpx;  // largest blurred pixel box of a face
    __shared__ uint32_t s_straddle;
    __shared__ uint32_t s_zext;   // bits of the largest depth extent (farthest - nearest vertex) of a rendered face
    if (threadIdx.x == 0) { s_maxpx = 0u; s_straddle = 0u; s_zext = 0u; }
    uint32_t my_px = 0u, my_straddle = 0u;
    float my_zext = 0.f;
    // per tile: entries << 32 | cost (counted), or a touched-tile bitmap when the image has too many tiles; behind it the tiles'
    // list cursors
    extern __shared__ __align__(16) unsigned long long tcnt64[];
    TSETUP_INIT
    const int n = blockIdx.x;
    const int V = q.V, F = q.F, S = q.S, tiles_x = q.tiles_x;
    // the fused entry point's per-image initialisation rides along (saves a 100 MB memset and a copy launch per iteration):
    // the vertex gradient of this image starts at zero, its loss at sum |0 - target|
    if (q.d_ndc_zero) {
        float2 *z = reinterpret_cast<float2 *>(q.d_ndc_zero) + (size_t)n * V;
        for (int i = threadIdx.x; i < V; i += blockDim.x) z[i] = make_float2(0.f, 0.f);
    }
    if (q.loss_dst && threadIdx.x == 0) { q.loss_dst[n] = q.loss_src[n]; q.loss_acc[n] = 0ull; }
    if (q.cd_counter && n == 0 && threadIdx.x == 0) { q.cd_counter[0] = 0u; q.cd_counter[1] = 0u; }
    const int n_tiles = tiles_x * tiles_x;
    const bool counted = n_tiles <= COUNT_TILES_MAX;
    uint32_t *const tbits = reinterpret_cast<uint32_t *>(tcnt64);        // (!counted) touched-tile bitmap
    const int KC = counted ? q.copies : 1;                                   // copies of the per-tile words: [copy][tile]
    uint32_t *const tcur = reinterpret_cast<uint32_t *>(tcnt64 + KC * n_tiles);  // (counted) list cursor of every tile and copy
    if (counted) { for (int i = threadIdx.x; i < KC * n_tiles; i += blockDim.x) tcnt64[i] = 0ull; }
    else { for (int i = threadIdx.x; i < (n_tiles + 31) >> 5; i += blockDim.x) tbits[i] = 0u; }
    __syncthreads();
    const float *vn = q.verts_ndc + (size_t)n * V * 3;
    const float fS = (float)S;
    const int FP = faces_padded(F), FT = FP + CLIP_FX;  // ids of the front parts of cut faces start at FP; tables are FT long
    const int n_groups = FT / WAVE;
    __shared__ uint32_t s_ncut, s_unclipped;
    __shared__ int s_cut[CLIP_CUTS];
    if (threadIdx.x == 0) { s_ncut = 0u; s_unclipped = 0u; }
    for (int i = threadIdx.x; i < FT - F; i += blockDim.x) q.tbox[(size_t)n * FT + F + i] = 0x0000FFFFu;  // (ids F .. FT-1: empty unless a cut face fills them)
    __syncthreads();
    TSETUP(1)
    float *const xv_n = q.clip.xv + (size_t)n * CLIP_VX * 3;
    int *const xf_n = q.clip.xf + (size_t)n * CLIP_FX * 3;
    // one face (an original one or the front part of a cut one): validity, blurred pixel box -> tile box, cost / entry per tile
    auto emit = [&](int fid, float x0, float y0, float z0, float x1, float y1, float z1, float x2, float y2, float z2) -> uint32_t {
        uint32_t box = 0x0000FFFFu;  // empty: tx0 = ty0 = 255 > tx1 = ty1 = 0
        const float zmin = fminf(fminf(z0, z1), z2), zmax = fmaxf(fmaxf(z0, z1), z2);
        const float area = edge_fn(x0, y0, x1, y1, x2, y2);
        const bool finite = (x0 == x0) && (x1 == x1) && (x2 == x2) && (y0 == y0) && (y1 == y1) && (y2 == y2);
        // zmin < 1e-8: the rasteriser's own rule; zmax < z_clip: the face lies entirely nearer than MeshRasterizer's
        // z_clip_value (znear / 2) and clip_faces() removes it
        if (finite && !(zmin < K_EPS) && !(zmax < q.z_clip) && !(area <= K_EPS && area >= -K_EPS)) {
            const float xlo = fminf(fminf(x0, x1), x2) - q.sqrt_blur, xhi = fmaxf(fmaxf(x0, x1), x2) + q.sqrt_blur;
            const float ylo = fminf(fminf(y0, y1), y2) - q.sqrt_blur, yhi = fmaxf(fmaxf(y0, y1), y2) + q.sqrt_blur;
            // pixel index i (flipped axis) has centre -1 + (2i+1)/S: centres inside [lo,hi] are ceil(v_lo)..floor(v_hi)
            // with v = ((x+1) S - 1)/2; 0.01 px of slack covers the float rounding of both sides (clamped in float first: the
            // front part of a cut face can reach far outside the image)
            const float vxl = fminf(fmaxf(((xlo + 1.0f) * fS - 1.0f) * 0.5f - 0.01f, -1.0f), fS), vxh = fminf(fmaxf(((xhi + 1.0f) * fS - 1.0f) * 0.5f + 0.01f, -1.0f), fS);
            const float vyl = fminf(fmaxf(((ylo + 1.0f) * fS - 1.0f) * 0.5f - 0.01f, -1.0f), fS), vyh = fminf(fmaxf(((yhi + 1.0f) * fS - 1.0f) * 0.5f + 0.01f, -1.0f), fS);
            int xi_lo = (int)ceilf(vxl), xi_hi = (int)floorf(vxh), yi_lo = (int)ceilf(vyl), yi_hi = (int)floorf(vyh);
            xi_lo = max(xi_lo, 0); yi_lo = max(yi_lo, 0);
            xi_hi = min(xi_hi, S - 1); yi_hi = min(yi_hi, S - 1);
            if (xi_lo <= xi_hi && yi_lo <= yi_hi) {
                my_px = max(my_px, (uint32_t)((xi_hi - xi_lo + 1) * (yi_hi - yi_lo + 1)));
                my_zext = fmaxf(my_zext, zmax - zmin);
                // output column xo = S-1-xi
                const int tx0 = (S - 1 - xi_hi) / TILE, tx1 = (S - 1 - xi_lo) / TILE;
                const int ty0 = (S - 1 - yi_hi) / TILE, ty1 = (S - 1 - yi_lo) / TILE;
                box = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
                const int xo0 = S - 1 - xi_hi, xo1 = S - 1 - xi_lo, yo0 = S - 1 - yi_hi, yo1 = S - 1 - yi_lo;
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx) {
                        const int t = ty * tiles_x + tx;
                        if (counted) {  // cost of this face in this tile: its (face, pixel) pairs plus a bit for staging it; one list entry
                            const int wx = min(xo1, tx * TILE + TILE - 1) - max(xo0, tx * TILE) + 1;
                            const int wy = min(yo1, ty * TILE + TILE - 1) - max(yo0, ty * TILE) + 1;
                            HOOK_SETUP_COUNT(atomicAdd(&tcnt64[(fid & (KC - 1)) * n_tiles + t], (1ull << 32) | (unsigned long long)(uint32_t)(wx * wy + 8));)
                        } else {
                            atomicOr(&tbits[t >> 5], 1u << (t & 31));
                        }
                    }
            }
        }
        q.tbox[(size_t)n * FT + fid] = box;
        q.fzr[(size_t)n * FT + fid] = make_float2(zmin, zmax);
        return box;
    };
    // (the vertex ids of the NEXT round's face are requested before this round's vertices are gathered: a round is one memory
    // round trip - ids -> vertices was two - and an image is a chain of F / blockDim.x rounds)
    int nx0 = 0, nx1 = 0, nx2 = 0;
    if ((int)threadIdx.x < F) { nx0 = q.faces[3 * threadIdx.x]; nx1 = q.faces[3 * threadIdx.x + 1]; nx2 = q.faces[3 * threadIdx.x + 2]; }
    for (int f0 = 0; f0 < FP; f0 += blockDim.x) {  // every wave handles 64 consecutive faces per round
        const int f = f0 + threadIdx.x;
        uint32_t box = 0x0000FFFFu;
        const int ii[3] = {nx0, nx1, nx2};
        {
            const int fn = min(f + (int)blockDim.x, F - 1);
            nx0 = q.faces[3 * fn]; nx1 = q.faces[3 * fn + 1]; nx2 = q.faces[3 * fn + 2];
        }
        if (f < F) {
            float X[3], Y[3], Z[3];
#pragma unroll
            for (int k = 0; k < 3; ++k) { X[k] = vn[3 * ii[k]]; Y[k] = vn[3 * ii[k] + 1]; Z[k] = vn[3 * ii[k] + 2]; }
            const int nb = (Z[0] < q.z_clip ? 1 : 0) + (Z[1] < q.z_clip ? 1 : 0) + (Z[2] < q.z_clip ? 1 : 0);  // vertices behind the plane
            const bool finite = (X[0] == X[0]) && (X[1] == X[1]) && (X[2] == X[2]) && (Y[0] == Y[0]) && (Y[1] == Y[1]) && (Y[2] == Y[2]) &&
                                (Z[0] == Z[0]) && (Z[1] == Z[1]) && (Z[2] == Z[2]);
            bool cut = false;
            if (finite && (nb == 1 || nb == 2)) {  // crosses the plane: set aside for the cut loop below (rare: kept out of this loop's registers)
                ++my_straddle;
                const uint32_t c = q.clip.xv ? atomicAdd(&s_ncut, 1u) : (uint32_t)CLIP_CUTS;
                if (c < (uint32_t)CLIP_CUTS) {
                    s_cut[c] = f;
                    cut = true;
                    q.tbox[(size_t)n * FT + f] = 0x0000FFFFu;  // the face itself is replaced by its front part
                    q.fzr[(size_t)n * FT + f] = make_float2(fminf(fminf(Z[0], Z[1]), Z[2]), fmaxf(fmaxf(Z[0], Z[1]), Z[2]));
                } else {
                    atomicAdd(&s_unclipped, 1u);  // beyond the tables: rendered whole, or dropped when a vertex is nearer than 1e-8 (counted)
                }
            }
            if (!cut) box = emit(f, X[0], Y[0], Z[0], X[1], Y[1], Z[1], X[2], Y[2], Z[2]);
        }
        // union of the wave's 64 boxes: the tile kernel skips whole groups of faces with one test (images that are not binned)
        int gx0 = box & 0xFF, gy0 = (box >> 8) & 0xFF, gx1 = (box >> 16) & 0xFF, gy1 = box >> 24;
        for (int o = 32; o > 0; o >>= 1) {
            gx0 = min(gx0, __shfl_xor(gx0, o, WAVE)); gy0 = min(gy0, __shfl_xor(gy0, o, WAVE));
            gx1 = max(gx1, __shfl_xor(gx1, o, WAVE)); gy1 = max(gy1, __shfl_xor(gy1, o, WAVE));
        }
        const int grp = (f0 + (int)threadIdx.x) / WAVE;
        if ((threadIdx.x & (WAVE - 1)) == 0 && grp < FP / WAVE)
            q.gbox[(size_t)n * n_groups + grp] = (uint32_t)gx0 | ((uint32_t)gy0 << 8) | ((uint32_t)gx1 << 16) | ((uint32_t)gy1 << 24);
    }
    __syncthreads();
    TSETUP(2)
    // the faces that cross the plane: cut c owns the new vertices 2c, 2c + 1 and the front-part faces FP + 2c, FP + 2c + 1
    const uint32_t n_cut = min(s_ncut, (uint32_t)CLIP_CUTS);
    for (uint32_t c = threadIdx.x; c < n_cut; c += blockDim.x) {
        const int f = s_cut[c];
        const int ii[3] = {q.faces[3 * f], q.faces[3 * f + 1], q.faces[3 * f + 2]};
        float X[3], Y[3], Z[3];
        for (int k = 0; k < 3; ++k) { X[k] = vn[3 * ii[k]]; Y[k] = vn[3 * ii[k] + 1]; Z[k] = vn[3 * ii[k] + 2]; }
        const int nb = (Z[0] < q.z_clip ? 1 : 0) + (Z[1] < q.z_clip ? 1 : 0) + (Z[2] < q.z_clip ? 1 : 0);
        // the isolated vertex first (the one behind, or the one in front), cyclic order kept
        const int k1 = nb == 1 ? (Z[0] < q.z_clip ? 0 : (Z[1] < q.z_clip ? 1 : 2)) : (!(Z[0] < q.z_clip) ? 0 : (!(Z[1] < q.z_clip) ? 1 : 2));
        const int o1 = k1, o2 = (k1 + 1) % 3, o3 = (k1 + 2) % 3;
        const uint32_t jv = 2u * c;
        float nx[2], ny[2];
        for (int e = 0; e < 2; ++e) {  // where the edges p1-p2 and p1-p3 cross the plane (view-space interpolation)
            const int a = o1, b = e == 0 ? o2 : o3;
            const float wb = (Z[a] - q.z_clip) / (Z[a] - Z[b]);
            const float ca = Z[a] * (1.0f - wb) / q.z_clip, cb = Z[b] * wb / q.z_clip;
            nx[e] = ca * X[a] + cb * X[b]; ny[e] = ca * Y[a] + cb * Y[b];
            float *o = xv_n + 3 * (jv + e);
            o[0] = nx[e]; o[1] = ny[e]; o[2] = q.z_clip;
            q.clip.xsrc[(size_t)n * CLIP_VX + jv + e] = make_int2(ii[a], ii[b]);
            q.clip.xcoef[(size_t)n * CLIP_VX + jv + e] = make_float2(ca, cb);
        }
        const int v4 = V + (int)jv, v5 = v4 + 1;
        int *xf = xf_n + 3 * (2 * c);
        if (nb == 1) {  // quadrilateral (p4, p2, p3, p5) as (p4, p2, p3) + (p4, p3, p5)
            xf[0] = v4; xf[1] = ii[o2]; xf[2] = ii[o3]; xf[3] = v4; xf[4] = ii[o3]; xf[5] = v5;
            emit(FP + 2 * (int)c, nx[0], ny[0], q.z_clip, X[o2], Y[o2], Z[o2], X[o3], Y[o3], Z[o3]);
            emit(FP + 2 * (int)c + 1, nx[0], ny[0], q.z_clip, X[o3], Y[o3], Z[o3], nx[1], ny[1], q.z_clip);
        } else {        // triangle (p1, p4, p5)
            xf[0] = ii[o1]; xf[1] = v4; xf[2] = v5;
            emit(FP + 2 * (int)c, X[o1], Y[o1], Z[o1], nx[0], ny[0], q.z_clip, nx[1], ny[1], q.z_clip);
        }
    }
    __syncthreads();  // the front parts' tile boxes are in place (written by whichever thread cut their face)
    TSETUP(3)
    static_assert(CLIP_FX % WAVE == 0 && SETUP_THREADS % WAVE == 0, "whole waves of front-part faces");
    for (int i = threadIdx.x; i < CLIP_FX; i += blockDim.x) {  // their group boxes (wave = group of 64; rows behind the last cut are empty)
        const uint32_t box = i < 2 * (int)n_cut ? q.tbox[(size_t)n * FT + FP + i] : 0x0000FFFFu;
        int gx0 = box & 0xFF, gy0 = (box >> 8) & 0xFF, gx1 = (box >> 16) & 0xFF, gy1 = box >> 24;
        for (int o = 32; o > 0; o >>= 1) {
            gx0 = min(gx0, __shfl_xor(gx0, o, WAVE)); gy0 = min(gy0, __shfl_xor(gy0, o, WAVE));
            gx1 = max(gx1, __shfl_xor(gx1, o, WAVE)); gy1 = max(gy1, __shfl_xor(gy1, o, WAVE));
        }
        if ((i & (WAVE - 1)) == 0)
            q.gbox[(size_t)n * n_groups + FP / WAVE + i / WAVE] = (uint32_t)gx0 | ((uint32_t)gy0 << 8) | ((uint32_t)gx1 << 16) | ((uint32_t)gy1 << 24);
    }
    if (q.clip.xcount) {
        const uint32_t nxv = 2u * n_cut;
        if (threadIdx.x == 0) q.clip.xcount[n] = nxv;
        for (int i = threadIdx.x; i < (int)nxv * 2; i += blockDim.x) q.clip.xg[(size_t)n * CLIP_VX * 2 + i] = 0.f;
    }
    if (threadIdx.x == 0 && s_unclipped) atomicAdd(&q.ctr->unclipped, s_unclipped);
    // Bound on what one vertex component of this image can receive from pass 3, up to the factor |upstream gradient| /
    // sqrt(sigma): a kept record of probability p = sigmoid(-+r^2 / sigma) adds at most 2 r p alpha |g| / sigma to an end point,
    // alpha <= 1 - p, and r p (1 - p) <= 0.197 sqrt(sigma) for every r (maximum of sqrt(u) s(u) (1 - s(u)), u = r^2 / sigma);
    // a face has at most its blurred pixel box of records; a vertex has at most max_valence faces.
    {  // over the wave first, then one LDS atomic per wave (1 024 lanes on one LDS word took 16 k cycles of a one-image launch's 103 k)
        uint32_t zb = __float_as_uint(my_zext);  // (non-negative floats order like their bit patterns)
        for (int o = 32; o > 0; o >>= 1) {
            my_px = max(my_px, (uint32_t)__shfl_xor((int)my_px, o, WAVE));
            zb = max(zb, (uint32_t)__shfl_xor((int)zb, o, WAVE));
            my_straddle += (uint32_t)__shfl_xor((int)my_straddle, o, WAVE);
        }
        if ((threadIdx.x & (WAVE - 1)) == 0) {
            if (my_px) atomicMax(&s_maxpx, my_px);
            if (zb) atomicMax(&s_zext, zb);
            if (my_straddle) atomicAdd(&s_straddle, my_straddle);
        }
    }
    __syncthreads();
    TSETUP(4)
    if (threadIdx.x == 0 && q.img_bound) {
        const float bound = 1.02f * 0.4f * (float)q.max_valence * (float)s_maxpx;
        q.img_bound[n] = bound;
        if (q.dndc_scale) {  // what the consumer of a packed gradient row multiplies by (0: the row holds plain floats)
            // (an image with cut faces accumulates in plain floats: the gradients its new vertices hand back are scaled by z / z_clip
            // factors that no a-priori bound covers)
            const bool pk = q.packed && s_ncut == 0u;
            const float sc = pk ? image_fx_scale(bound, q.pix_scale[n], q.inv_sigma) : 0.f;
            q.dndc_scale[n] = sc > 0.f ? 1.0f / sc : (pk ? -1.0f : 0.f);  // (-1: packed row that received nothing: decodes to zeros)
        }
    }
    if (threadIdx.x == 0 && s_straddle) atomicAdd(&q.ctr->straddling, s_straddle);
    // touched tiles -> the work list of their cost class
    __shared__ uint32_t s_cnt[N_CLASSES], s_base[N_CLASSES], s_ents[SETUP_THREADS / WAVE], s_binned;
    if (threadIdx.x < N_CLASSES) s_cnt[threadIdx.x] = 0u;
    __syncthreads();
    auto tile_class = [&](int t) -> int {  // -1: untouched
        if (!counted) return ((tbits[t >> 5] >> (t & 31)) & 1u) ? N_CLASSES - 1 : -1;
        uint32_t c = 0u;
        for (int k = 0; k < KC; ++k) c += (uint32_t)tcnt64[k * n_tiles + t];
        return c == 0u ? -1 : (c >= CLASS_T0 ? 0 : (c >= CLASS_T1 ? 1 : (c >= CLASS_T2 ? 2 : 3)));
    };
    uint32_t mine[N_CLASSES] = {0u, 0u, 0u, 0u};
    uint32_t my_ents = 0u;  // list entries of this thread's tiles (t = thread, thread + block, ...)
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
        const int c = tile_class(t);
#pragma unroll
        for (int k = 0; k < N_CLASSES; ++k) mine[k] += (c == k) ? 1u : 0u;
        if (counted)
            for (int k = 0; k < KC; ++k) my_ents += (uint32_t)(tcnt64[k * n_tiles + t] >> 32);
    }
    uint32_t off[N_CLASSES];
#pragma unroll
    for (int k = 0; k < N_CLASSES; ++k) off[k] = mine[k] ? atomicAdd(&s_cnt[k], mine[k]) : 0u;
    // lists end to end: exclusive prefix of the entry counts over the block (thread order, each thread's tiles consecutive)
    const uint32_t incl = (uint32_t)wave_scan_add((int)my_ents);
    if ((threadIdx.x & (WAVE - 1)) == WAVE - 1) s_ents[threadIdx.x / WAVE] = incl;
    __syncthreads();
    TSETUP(5)
    uint32_t ent_off = incl - my_ents;
    for (int w = 0; w < (int)(threadIdx.x / WAVE); ++w) ent_off += s_ents[w];
    if (threadIdx.x == blockDim.x - 1) s_binned = (counted && q.list_cap != 0u && ent_off + my_ents <= q.list_cap) ? 1u : 0u;
    const int part = n % N_PARTS;
    if (threadIdx.x < N_CLASSES) s_base[threadIdx.x] = s_cnt[threadIdx.x] ? atomicAdd(&q.ctr->n_class[part][threadIdx.x], s_cnt[threadIdx.x]) : 0u;
    __syncthreads();
    TSETUP(6)
    const bool binned = s_binned != 0u;
#pragma unroll
    for (int k = 0; k < N_CLASSES; ++k) off[k] += s_base[k];
    uint32_t run = ent_off;
    const uint32_t zext_bits = s_zext;  // (final since the barrier behind the atomicMax above)
    for (int t = threadIdx.x; t < n_tiles; t += blockDim.x) {
        const int c = tile_class(t);
        uint32_t e = 0u;
        const uint32_t first = run;
        if (counted)
            for (int k = 0; k < KC; ++k) {  // a tile's entries lie copy by copy inside its list
                if (binned) tcur[k * n_tiles + t] = run + e;
                e += (uint32_t)(tcnt64[k * n_tiles + t] >> 32);
            }
        run += e;
        if (c < 0) continue;
        uint32_t slot = 0u;
#pragma unroll
        for (int k = 0; k < N_CLASSES; ++k)
            if (c == k) slot = off[k]++;
        // classes 0 and 2 grow from the front of their array, 1 and 3 from the back
        const uint32_t idx = (uint32_t)(2 * part + (c >> 1)) * q.item_cap + ((c & 1) ? q.item_cap - 1u - slot : slot);
        // the work item carries its tile's list with it: {image * tiles + tile, first entry, entries (0xFFFFFFFF: the tile kernel builds the
        // list), largest depth extent of a face of this image} - one 16-byte load in the tile kernel where the item code and a tile
        // descriptor were two dependent ones
        q.items[idx] = make_uint4((uint32_t)n * (uint32_t)n_tiles + (uint32_t)t, binned ? first : 0u, binned ? e : 0xFFFFFFFFu, zext_bits);
    }
    if (!binned) return;  // (block-uniform)
    __syncthreads();
    TSETUP(7)
    // pass 2: every face to the lists of the tiles of its box (its own tile box and depth range come back from L1 / L2)
    uint2 *const lists = q.lists + (size_t)n * q.list_cap;
    // (consecutive faces cover the same tiles: their entries take consecutive slots, so a wave's stores land in few cache lines;
    // spreading the lanes over distant faces to thin out the same-address atomics was measured slower, 601 -> 658 us)
    const int f_end = FP + 2 * (int)n_cut;  // (the rows behind the last cut face's front parts are empty)
    for (int f = threadIdx.x; f < f_end; f += blockDim.x) {
        const uint32_t box = q.tbox[(size_t)n * FT + f];
        const uint2 ent = make_uint2((uint32_t)f, __float_as_uint(q.fzr[(size_t)n * FT + f].x));  // (requested with the box: one round trip)
        const int tx0 = box & 0xFF, ty0 = (box >> 8) & 0xFF, tx1 = (box >> 16) & 0xFF, ty1 = box >> 24;
        if (tx0 > tx1) continue;
        uint32_t *const cur = tcur + (f & (KC - 1)) * n_tiles;
        // (a fast path for boxes of at most 2 x 2 tiles - the four returning atomics issued before the four stores - measured no
        // different, profiles/r5_experiments.md)
        for (int ty = ty0; ty <= ty1; ++ty)
            for (int tx = tx0; tx <= tx1; ++tx) at(lists, atomicAdd(&cur[ty * tiles_x + tx], 1u)) = ent;
    }
    TSETUP(8)
    TSETUP_REPORT
}

// ---------------------------------------------------------------------------------------------
// per-(pixel, face) evaluation
// ---------------------------------------------------------------------------------------------
// Face record staged in LDS (32 floats = 8 x 16 B).  Everything that does not depend on the pixel is folded in once
// per (tile, face): coordinates are relative to the tile centre (cx, cy) so the affine forms below do not cancel
// catastrophically.
//   w_i(p) = A_i dx + B_i dy + C_i  = b_i(p) * z_j z_k   (perspective-correct barycentric numerators; the
//            common denominator is positive, so inside <=> all w_i > 0)
// The fields are ordered so that what the evaluation computes in pairs sits in adjacent registers after the 16-byte LDS reads:
// (w0, w1), the projections on the two edges leaving v0, ... become one packed fp32 instruction each (v_pk_fma_f32 / v_pk_mul_f32 /
// v_pk_add_f32) without register moves; the third of each kind stays scalar.
struct alignas(16) FaceRec {
    float A0, A1, B0, B1;
    float C0, C1, A2, B2;
    float C2, z0, z1, z2;
    float x0c, x1c, y0c, y1c;       // v0, v1 relative to the tile centre
    float e01x, e02x, e01y, e02y;   // edge vectors and 1/|e|^2 (0 for a degenerate edge)
    float rl01, rl02, e12x, e12y;
    float rl12;
    int i0, i1, i2;
};
static_assert(sizeof(FaceRec) == FREC * sizeof(float), "FaceRec layout");
// (Round 4: the record no longer carries the blurred bounding box.  A lane only ever sees pixels of its face's pixel box, a superset of
// the bounding box by 0.01 px, and a pixel outside the box is farther than sqrt(blur) from the face, so the distance test rejects it
// as the box test did; the two can differ only for a pixel centre within rounding of the box edge.)

typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 splat2(float x) { return (f32x2){x, x}; }
__device__ __forceinline__ f32x2 clamp01(f32x2 v) {  // (folds into the clamp bit of the producing instruction)
    return __builtin_elementwise_min(__builtin_elementwise_max(v, splat2(0.f)), splat2(1.f));
}
__device__ __forceinline__ float vmax_raw(float a, float b) {
    // (v_max_f32 spelled out: hipcc puts a canonicalising v_max x, x in front of every fmaxf whose input it cannot prove canonical,
    // and these inputs - results of fma instructions - always are)
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

// Two horizontally adjacent pixels of one face per lane (round 4).  Everything a lane does per (face, pixel) pair that is not
// arithmetic - finding its face and pixel, gathering the face record from LDS, the loop around it - is paid once per TWO pairs, and
// the arithmetic itself packs over the two pixels (v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32, the face's constants as op_sel
// splats): the two pixels share dy, and with it the y parts of every projection.
// The face record as seven 16-byte rows read straight into registers.  (Reading it through a FaceRec in private memory let the
// optimiser turn `w0 > 0 ? z0 : z1` into an INDEXED load from that private copy - which then lives in scratch memory, with a
// scratch store and six scratch loads per sweep step.)
struct FaceRows { float4 r0, r1, r2, r3, r4, r5, r6; };
__device__ __forceinline__ FaceRows load_face_rows(const float *rec) {
    const float4 *r = reinterpret_cast<const float4 *>(rec);
    return FaceRows{r[0], r[1], r[2], r[3], r[4], r[5], r[6]};
}
struct PairEval2 {
    f32x2 w0, w1, w2;     // perspective-correct barycentric numerators, .x = left pixel (even column), .y = right pixel
    f32x2 sd;             // signed squared distance
    bool cand0, cand1, inside0, inside1;
    uint32_t ebits0, ebits1;  // closest edge << 23 (record layout)
};
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// clamp(v * s, 0, 1) for both pixels in one instruction, s = the low / high half of the pair `s2` (hipcc leaves the clamp of a packed
// product as two separate v_max)
__device__ __forceinline__ f32x2 pk_mul_clamp_lo(f32x2 v, f32x2 s2) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel_hi:[1,0] clamp" : "=v"(r) : "v"(v), "v"(s2));
    return r;
}
__device__ __forceinline__ f32x2 pk_mul_clamp_hi(f32x2 v, f32x2 s2) {
    f32x2 r;
    asm("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] clamp" : "=v"(r) : "v"(v), "v"(s2));
    return r;
}
__device__ __forceinline__ void eval_pair2(const FaceRows &q, float dx0, float dx1, float dyp, float blur, PairEval2 &e) {
    const f32x2 DX = {dx0, dx1};
    const f32x2 base01 = pk_fma((f32x2){q.r0.z, q.r0.w}, splat2(dyp), (f32x2){q.r1.x, q.r1.y});
    const float base2 = fmaf(q.r1.w, dyp, q.r2.x);
    e.w0 = pk_fma(splat2(q.r0.x), DX, splat2(base01.x));
    e.w1 = pk_fma(splat2(q.r0.y), DX, splat2(base01.y));
    e.w2 = pk_fma(splat2(q.r1.z), DX, splat2(base2));
    e.inside0 = fminf(fminf(e.w0.x, e.w1.x), e.w2.x) > 0.f;   // (all three positive; the numerators are finite)
    e.inside1 = fminf(fminf(e.w0.y, e.w1.y), e.w2.y) > 0.f;
    // pixels relative to v0 and to v1; the y parts are the same for both pixels
    const f32x2 QX0 = DX - splat2(q.r3.x), QX1 = DX - splat2(q.r3.y);
    const f32x2 qy = splat2(dyp) - (f32x2){q.r3.z, q.r3.w};           // .x relative to v0, .y relative to v1
    const f32x2 eyq0 = (f32x2){q.r4.z, q.r4.w} * splat2(qy.x);        // y parts of the projections on the edges leaving v0
    const float eyq12 = q.r5.w * qy.y;
    const f32x2 rl0102 = {q.r5.x, q.r5.y}, rl12_ = {q.r6.x, q.r6.y};
    const f32x2 T01 = pk_mul_clamp_lo(pk_fma(splat2(q.r4.x), QX0, splat2(eyq0.x)), rl0102);
    const f32x2 T02 = pk_mul_clamp_hi(pk_fma(splat2(q.r4.y), QX0, splat2(eyq0.y)), rl0102);
    const f32x2 T12 = pk_mul_clamp_lo(pk_fma(splat2(q.r5.z), QX1, splat2(eyq12)), rl12_);
    const f32x2 RX01 = pk_fma(T01, splat2(q.r4.x), -QX0), RY01 = pk_fma(T01, splat2(q.r4.z), -splat2(qy.x));
    const f32x2 RX02 = pk_fma(T02, splat2(q.r4.y), -QX0), RY02 = pk_fma(T02, splat2(q.r4.w), -splat2(qy.x));
    const f32x2 RX12 = pk_fma(T12, splat2(q.r5.z), -QX1), RY12 = pk_fma(T12, splat2(q.r5.w), -splat2(qy.y));
    const f32x2 D01 = pk_fma(RX01, RX01, RY01 * RY01), D02 = pk_fma(RX02, RX02, RY02 * RY02), D12 = pk_fma(RX12, RX12, RY12 * RY12);
    const float dist0 = fminf(fminf(D01.x, D02.x), D12.x), dist1 = fminf(fminf(D01.y, D02.y), D12.y);
    e.cand0 = e.inside0 || dist0 < blur;
    e.cand1 = e.inside1 || dist1 < blur;
    e.sd = (f32x2){e.inside0 ? -dist0 : dist0, e.inside1 ? -dist1 : dist1};
    // closest edge in the reference's order e01, e02, e12 with <= ties: the first whose distance IS the minimum
    e.ebits0 = D01.x == dist0 ? 0u : (D02.x == dist0 ? 1u << 23 : 2u << 23);
    e.ebits1 = D01.y == dist1 ? 0u : (D02.y == dist1 ? 1u << 23 : 2u << 23);
}
// depth at the clipped, renormalised perspective-correct barycentrics, both pixels:
// c_i = max(p_i,0) / max(sum, 1e-5), p_i = w_i / den; 1/den cancels: c_i = max(w_i,0) / max(sum max(w,0), 1e-5 den).
// When a single weight survives the clip the depth is EXACTLY that vertex's depth, so faces sharing the vertex tie
// exactly (as x / x == 1 does in the reference) and the (depth, face id) order stays well defined.
__device__ __forceinline__ f32x2 pair_depth2(const FaceRows &q, const PairEval2 &e) {
    // (the vertex depths as opaque scalars: selecting among the ELEMENTS of a row makes the optimiser index the row dynamically,
    // through scratch memory)
    float z0 = q.r2.y, z1 = q.r2.z, z2 = q.r2.w;
    asm("" : "+v"(z0), "+v"(z1), "+v"(z2));
    const f32x2 den3 = e.w0 + e.w1 + e.w2;
    const f32x2 den = {fmaxf(den3.x, K_EPS), fmaxf(den3.y, K_EPS)};
    const f32x2 m0 = {vmax_raw(e.w0.x, 0.f), vmax_raw(e.w0.y, 0.f)}, m1 = {vmax_raw(e.w1.x, 0.f), vmax_raw(e.w1.y, 0.f)},
                m2 = {vmax_raw(e.w2.x, 0.f), vmax_raw(e.w2.y, 0.f)};
    const f32x2 msum = m0 + m1 + m2, floor_ = splat2(1e-5f) * den;
    const f32x2 cs = {fmaxf(msum.x, floor_.x), fmaxf(msum.y, floor_.y)};
    const f32x2 rc = {__builtin_amdgcn_rcpf(cs.x), __builtin_amdgcn_rcpf(cs.y)};
    // (every fused multiply-add spelled out: left to the compiler, the last one is contracted in one inlining context and not in
    // another - k_raster_tie_replay must reproduce these depths bit for bit, or an exact tie here is no tie there)
    const f32x2 pz = pk_fma(splat2(z2), m2 * rc, pk_fma(splat2(z0), m0 * rc, splat2(z1) * (m1 * rc)));
    // one survivor <=> the sum of the clipped weights equals their maximum (and was not lifted by the 1e-5 floor)
    const float mx0 = fmaxf(fmaxf(m0.x, m1.x), m2.x), mx1 = fmaxf(fmaxf(m0.y, m1.y), m2.y);
    const bool single0 = (msum.x == mx0) && (mx0 >= cs.x), single1 = (msum.y == mx1) && (mx1 >= cs.y);
    const float zv0 = m0.x > 0.f ? z0 : (m1.x > 0.f ? z1 : z2), zv1 = m0.y > 0.f ? z0 : (m1.y > 0.f ? z1 : z2);
    return (f32x2){single0 ? zv0 : pz.x, single1 ? zv1 : pz.y};
}

// float -> nearest integer in ONE instruction (v_cvt_rpi_i32_f32 = floor(x + 0.5); __float2int_rn is v_rndne + v_cvt; the two differ
// only on exact halves, which round up here)
__device__ __forceinline__ int cvt_round(float x) {
    int r;
    asm("v_cvt_rpi_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return r;
}

__device__ __forceinline__ float face_prob(float sd, float inv_sigma_log2e) {
    // sigmoid(-dist / sigma) = 1 / (1 + 2^{dist log2(e) / sigma}); v_exp_f32 + v_rcp_f32 (1 ulp each), the two constant factors
    // of the exponent folded into one on the host
    return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(sd * inv_sigma_log2e));
}

__global__ void __launch_bounds__(256) k_unpack_dndc(float *__restrict__ d_ndc, const float *__restrict__ img_bound,
                                                     const float *__restrict__ pix_scale, float inv_sigma, int V,
                                                     const uint32_t *__restrict__ xcount) {
    const int n = blockIdx.x, v = blockIdx.y * blockDim.x + threadIdx.x;
    if (v >= V || (xcount && xcount[n] != 0u)) return;  // (images with cut faces hold plain floats already)
    const float sc = image_fx_scale(img_bound[n], pix_scale[n], inv_sigma);
    const float inv = sc > 0.f ? 1.0f / sc : 0.f;
    unsigned long long *p = reinterpret_cast<unsigned long long *>(d_ndc) + (size_t)n * V + v;
    const unsigned long long tot = *p;
    const int qy = (int)(uint32_t)tot;
    const int qx = (int)(uint32_t)((tot - (unsigned long long)(long long)qy) >> 32);
    *reinterpret_cast<float2 *>(p) = make_float2((float)qx * inv, (float)qy * inv);
}

// Gradient of the new vertices of cut faces back to the end points of the edges they lie on: xy_new = c_a xy_a + c_b xy_b with the
// coefficients held constant (see ClipTables).  One workgroup per image; images without cut faces leave at once.  Images with
// cut faces accumulate in plain floats, so these are float atomics on d_ndc (two new vertices may share an end point).
__global__ void __launch_bounds__(64) k_clip_backward(ClipTables c, float *__restrict__ d_ndc, int V, float *__restrict__ loss_img,
                                                      const unsigned long long *__restrict__ loss_acc, const float *__restrict__ verts_ndc,
                                                      float z_clip, SmilClipDepth cd, int image0) {
    const int n = blockIdx.x;
    // (fused entry point: the image's loss = what the setup kernel seeded it with + the tiles' terms, summed as integers)
    if (loss_img && threadIdx.x == 0) loss_img[n] += (float)((double)(long long)loss_acc[n] * (1.0 / 4294967296.0));
    const uint32_t nx = c.xcount[n];
    // Depth channel (round 5): the crossing point xy_new = (xy_a z_a (1 - w) + xy_b z_b w) / z_clip, w = (z_a - z_clip) / (z_a - z_b), also
    // depends on the end points' DEPTHS - pytorch3d's autograd differentiates clip_faces through them.  d_ndc has no depth
    // component, so these (rare) terms travel as a sparse list: two entries {vertex, d / d z} per new vertex, the image's run
    // recorded in cd.range; the LBS backward / smil_clip_depth_backward carry them through the camera.
    __shared__ uint32_t s_first;
    if (cd.range) {
        if (threadIdx.x == 0) {
            uint32_t first = 0u, cnt = 0u;
            if (nx) {
                first = atomicAdd(&cd.counter[0], 2u * nx);
                if (first + 2u * nx <= (uint32_t)cd.capacity) cnt = 2u * nx;
                else atomicAdd(&cd.counter[1], 2u * nx);  // (does not fit: dropped, counted; the run stays reserved but unused)
            }
            cd.range[2 * (size_t)(image0 + n)] = first;
            cd.range[2 * (size_t)(image0 + n) + 1] = cnt;
            s_first = cnt ? first : 0xFFFFFFFFu;
        }
        __syncthreads();
    }
    const uint32_t zfirst = cd.range ? s_first : 0xFFFFFFFFu;
    const float *vn = verts_ndc + (size_t)n * V * 3;
    for (uint32_t j = threadIdx.x; j < nx; j += blockDim.x) {
        const float gx = c.xg[((size_t)n * CLIP_VX + j) * 2], gy = c.xg[((size_t)n * CLIP_VX + j) * 2 + 1];
        const int2 ab = c.xsrc[(size_t)n * CLIP_VX + j];
        if (zfirst != 0xFFFFFFFFu) {
            const float xa = vn[3 * ab.x], ya = vn[3 * ab.x + 1], za = vn[3 * ab.x + 2];
            const float xb = vn[3 * ab.y], yb = vn[3 * ab.y + 1], zb = vn[3 * ab.y + 2];
            // xy_new = xy_a (1 - s) + xy_b s with s = z_b w / z_clip (the two weights sum to one: the crossing's depth is z_clip), so both
            // derivatives point along the edge: d xy_new / d z_a = (xy_b - xy_a) z_b (z_clip - z_b) / (z_clip (z_a - z_b)^2) and
            // d xy_new / d z_b = (xy_b - xy_a) z_a (z_a - z_clip) / (z_clip (z_a - z_b)^2).  Expanding them from the w form instead
            // cancels two terms of size |xy| |z| / z_clip against each other in fp32 (profiles/r5_fuzz.md).
            const float inv = 1.0f / (za - zb);
            const float ge = fmaf(gx, xb - xa, gy * (yb - ya)) * (inv * inv) * (1.0f / z_clip);
            cd.vertex[zfirst + 2u * j] = ab.x;
            cd.dz[zfirst + 2u * j] = ge * (zb * (z_clip - zb));
            cd.vertex[zfirst + 2u * j + 1u] = ab.y;
            cd.dz[zfirst + 2u * j + 1u] = ge * (za * (za - z_clip));
        }
        if (gx == 0.f && gy == 0.f) continue;
        const float2 co = c.xcoef[(size_t)n * CLIP_VX + j];
        float *da = d_ndc + ((size_t)n * V + ab.x) * 2, *db = d_ndc + ((size_t)n * V + ab.y) * 2;
        atomicAdd(da, co.x * gx); atomicAdd(da + 1, co.x * gy);
        atomicAdd(db, co.y * gx); atomicAdd(db + 1, co.y * gy);
    }
}

// ---------------------------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------------------------
#ifndef GCHUNK
#define GCHUNK 64            // faces whose gradient accumulators are live in pass 3 (a multiple of WAVE)
#endif
#ifndef GCOPIES
#define GCOPIES 2
#endif
//      GCOPIES              // private copies of those accumulators (measured: 1 -> 2 copies -3 %, 4 copies lose it again to zeroing and flushing)
struct alignas(16) DenseLds {
    union {
        float rec[DCHUNK * FSTR];    // pass 1: staged face records
        struct {
            // pass 3: gradient accumulators of GCHUNK faces x 3 vertices, (x, y) packed as two 32-bit fixed-point numbers
            // in one 64-bit word so that one ds_add_u64 adds both; GCOPIES private copies indexed by lane & (GCOPIES - 1)
            // keep the consecutive lanes of one face's run of records off each other's address (measured: the four
            // 13-way conflicting ds_add_f64 per record this replaces were more than half of pass 3)
            unsigned long long gacc[GCOPIES][GCHUNK * 3];
            double plog[WAVE];       // pass 2: sum of log2(1 - p_k) (fp64: ds_add_f64 runs at full rate on gfx950,
                                     //         ds_add_f32 at ~3 cycles per active lane)
            float4 pgrad[WAVE];      // after pass 1: {gradient coefficient, threshold depth bits, tie cut (face id), -}
        };
    };
    // select: [bucket / 2][pixel], two 16-bit counts per word; the first digit is counted by pass 1
    uint32_t hist[(1 << SEL_BITS) / 2 * WAVE];
    float2 pixt[WAVE];               // pass 1: pixel centre (px, py) in NDC
    uint2 psel[WAVE];                // select: {prefix of the wanted key, rank wanted among the keys sharing it (0: none)}
    int start[WAVE];                 // pass 1: 2048-bit map of the pairs that start a face's run (list phase: bucket counters)
    uint16_t bstart[1 << SEL1_BITS]; // first list position of every depth bucket of the near-to-far list (clamped to 65535)
};
static_assert(sizeof(DenseLds) * RESIDENT_PER_CU <= 160 * 1024, "the resident workgroups of a CU must fit its 160 KB of LDS");

// Record-stream accesses: written once, read once or twice, never shared between workgroups (non-temporal forms measured 12 % slower
// in round 2: the streams do live on L2 / Infinity Cache hits between pass 1 and the sweeps).
template <typename T> __device__ __forceinline__ T ld_stream(const T *base, uint32_t i) { return at(base, i); }
template <typename T> __device__ __forceinline__ void st_stream(T *base, uint32_t i, T v) { at(base, i) = v; }

// Single-wave workgroups: lanes exchange data through LDS without s_barrier, but the compiler must not forward a lane's
// own store to its later load, and the LDS queue must have drained.  Unlike __syncthreads() this does NOT wait for
// outstanding global stores (vmcnt), which in pass 1 would stall every sweep step on the previous step's record stores.
__device__ __forceinline__ void lds_fence() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// Ordered list of the faces whose tile box contains (tx,ty), written to `list` (global).  Also the range of the nearest /
// farthest vertex depth over those faces: every pair depth lies inside it (a convex combination of the face's vertex
// depths), which fixes the radix-select digits before pass 1 starts.  Depths are positive: the bit patterns order like
// the values.
__device__ __forceinline__ bool box_has(uint32_t b, int tx, int ty) {
    const int tx0 = b & 0xFF, ty0 = (b >> 8) & 0xFF, tx1 = (b >> 16) & 0xFF, ty1 = b >> 24;
    return (tx >= tx0) && (tx <= tx1) && (ty >= ty0) && (ty <= ty1);
}

#ifndef LGROUP
#define LGROUP 8  // 64-face groups whose tile boxes / depth ranges are requested together by the list build
#endif
__device__ __forceinline__ int build_list(const RasterArgs &a, int n, int tx, int ty, uint2 *list, int lane, uint32_t &kmin,
                                          uint32_t &kmax) {
    const uint32_t *__restrict__ tbox_n = a.tbox + (size_t)n * a.FT;
    const float2 *__restrict__ fzr_n = a.fzr + (size_t)n * a.FT;
    const int n_groups = a.FT / WAVE;
    const uint32_t *__restrict__ gbox_n = a.gbox + (size_t)n * n_groups;
    int cnt = 0;
    float zlo = 3.0e38f, zhi = 0.f;
    for (int g0 = 0; g0 < n_groups; g0 += WAVE) {
        // lane = group of 64 consecutive faces: does its box union reach this tile?
        const int g = g0 + lane;
        unsigned long long gm = __ballot(g < n_groups && box_has(gbox_n[min(g, n_groups - 1)], tx, ty));
        while (gm) {  // wave-uniform: the groups that do, in ascending order, LGROUP at a time (independent loads in flight)
            int fidx[LGROUP];
            uint32_t tb[LGROUP];
            float2 zz[LGROUP];
#pragma unroll
            for (int u = 0; u < LGROUP; ++u) {
                const int gi = gm ? g0 + (int)__builtin_ctzll(gm) : -1;
                gm &= gm - 1ull;  // 0 stays 0
                fidx[u] = gi >= 0 ? gi * WAVE + lane : a.FT;
                const int fc = min(fidx[u], a.FT - 1);
                tb[u] = tbox_n[fc];
                zz[u] = fzr_n[fc];
            }
#pragma unroll
            for (int u = 0; u < LGROUP; ++u) {
                const bool hit = fidx[u] < a.FT && box_has(tb[u], tx, ty);
                const unsigned long long mask = __ballot(hit);
                if (hit) {
                    at(list, (uint32_t)(cnt + __popcll(mask & ((1ull << lane) - 1ull)))) = make_uint2((uint32_t)fidx[u], __float_as_uint(zz[u].x));
                    zlo = fminf(zlo, zz[u].x);
                    zhi = fmaxf(zhi, zz[u].y);
                }
                cnt += __popcll(mask);
            }
        }
    }
    uint32_t lo = __float_as_uint(zlo), hi = __float_as_uint(zhi);
    for (int o = 32; o > 0; o >>= 1) {
        lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, WAVE));
        hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, WAVE));
    }
    kmin = lo;
    kmax = hi;
    return cnt;
}

// Near-to-far order for the tile's list: a counting sort of the faces by the first radix digit (the same digit the
// records' depths are histogrammed by) of their NEAREST vertex depth.  A record's depth is at least its face's nearest
// vertex depth, so once every face of digit <= d has been processed the per-pixel record counts of digits <= d are final:
// pass 1 uses that to stop collecting records for pixels that already hold K nearer ones (the reference keeps the K = 100
// nearest per pixel, p3d_renderer.py:42-47), and to leave the tile when no pixel is open any more.
// Order inside a bucket is arbitrary; depth ties between records are broken by face id, which the records' list position
// recovers through `out`.  bstart[d] = first position of bucket d.
__device__ __forceinline__ void sort_list_near_to_far(const uint2 *list, uint32_t *out, int n, uint32_t kmin, int shift1, int b1,
                                                      DenseLds &lds, int lane) {
    const int n_buckets = 1 << b1;
    lds.start[lane] = 0;
    __syncthreads();
    auto digit_of = [&](const uint2 &e) { return (int)(((e.y - kmin) >> shift1) & (uint32_t)(n_buckets - 1)); };
    // (four rows per step: the loads of a step are in flight together - at small launches a tile's time is its chain of
    // memory round trips)
    for (int i0 = 0; i0 < n; i0 += 4 * WAVE) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = at(list, (uint32_t)min(i0 + u * WAVE + lane, n - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * WAVE + lane < n) atomicAdd(&lds.start[digit_of(e[u])], 1);
    }
    __syncthreads();
    const int c = lane < n_buckets ? lds.start[lane] : 0;
    const int incl = wave_scan_add(c);
    __syncthreads();
    if (lane < n_buckets) lds.bstart[lane] = (uint16_t)min(incl - c, 65535);
    lds.start[lane] = incl - c;  // running cursor of every bucket
    __syncthreads();
    for (int i0 = 0; i0 < n; i0 += 4 * WAVE) {
        uint2 e[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) e[u] = at(list, (uint32_t)min(i0 + u * WAVE + lane, n - 1));
#pragma unroll
        for (int u = 0; u < 4; ++u)
            if (i0 + u * WAVE + lane < n) out[atomicAdd(&lds.start[digit_of(e[u])], 1)] = e[u].x;
    }
    __syncthreads();
}

// Staging of a chunk of DCHUNK = 32 faces by all 64 lanes: lanes l and l + 32 share face l.  The LOW lane builds the affine
// forms (rows 1-3 of the record) and the pixel ROWS the face's blurred box covers inside the open part of the tile; the
// HIGH lane the bounding box (row 0), the edge data (rows 4-7) and the pixel COLUMNS; the columns then cross over (one
// ds_bpermute each) and the low lane leaves with the face's pair count `cf` and the word pairs decode their pixel from.
// Both lanes load the same nine vertex floats (one transaction); the face's vertex indices (i0, i1, i2) come from the
// caller, which fetched them while the previous chunk was being evaluated.  Pixel index i (flipped axis) has centre -1 + (2i+1)/S:
// centres inside [lo, hi] are ceil(v_lo) .. floor(v_hi) with v = ((x+1) S - 1)/2; 0.01 px of slack covers the float rounding
// (a superset; eval_pair applies the exact test).
static_assert(2 * DCHUNK == WAVE, "two lanes per staged face");
struct Tri9 { float x0, y0, z0, x1, y1, z1, x2, y2, z2; };
// the nine vertex floats of a lane's face (both lanes of a face load the same: one transaction), through the clip tables
__device__ __forceinline__ Tri9 load_tri(const RasterArgs &a, const float *__restrict__ vn, const float *__restrict__ xv_n, int i0, int i1, int i2) {
    const float *p0 = vertex_ptr(vn, xv_n, a.V, i0), *p1 = vertex_ptr(vn, xv_n, a.V, i1), *p2 = vertex_ptr(vn, xv_n, a.V, i2);
    return Tri9{p0[0], p0[1], p0[2], p1[0], p1[1], p1[2], p2[0], p2[1], p2[2]};
}
// The seven rows of a face record from its vertices, in two halves (stage_faces: one lane each).  Every rounding is spelled out -
// no contraction left to the compiler (which fuses a*b - c*d one way in one inlining context and another way in the next): the
// tile kernel and k_raster_tie_replay must get the SAME bits from the same face, or an exact depth tie in one is no tie in the other.
__device__ __forceinline__ float edge_fx(float px, float py, float ax, float ay, float bx, float by) {
#pragma clang fp contract(off)
    return fmaf(px - ax, by - ay, -((py - ay) * (bx - ax)));
}
__device__ __forceinline__ void face_rows_lo(const Tri9 &tv, float cx, float cy, float4 &r0, float4 &r1, float4 &r2) {
#pragma clang fp contract(off)
    const float x0 = tv.x0, y0 = tv.y0, z0 = tv.z0, x1 = tv.x1, y1 = tv.y1, z1 = tv.z1, x2 = tv.x2, y2 = tv.y2, z2 = tv.z2;
    // (only the signs of the w_i and their ratios are used: the scale's last bits do not matter)
    const float rcp_area = __builtin_amdgcn_rcpf(edge_fx(x2, y2, x0, y0, x1, y1) + K_EPS);
    // edge function e_k(p) = (px - ax)(by - ay) - (py - ay)(bx - ax), linear in p; value at the tile centre + slopes
    const float s0 = rcp_area * (z1 * z2), s1 = rcp_area * (z0 * z2), s2 = rcp_area * (z0 * z1);
    r0 = make_float4((y2 - y1) * s0, (y0 - y2) * s1, -(x2 - x1) * s0, -(x0 - x2) * s1);                                      // A0 A1 B0 B1
    r1 = make_float4(edge_fx(cx, cy, x1, y1, x2, y2) * s0, edge_fx(cx, cy, x2, y2, x0, y0) * s1, (y1 - y0) * s2, -(x1 - x0) * s2);  // C0 C1 A2 B2
    r2 = make_float4(edge_fx(cx, cy, x0, y0, x1, y1) * s2, z0, z1, z2);
}
__device__ __forceinline__ void face_rows_hi(const Tri9 &tv, float cx, float cy, float4 &r3, float4 &r4, float4 &r5, float &rl12_out) {
#pragma clang fp contract(off)
    const float x0 = tv.x0, y0 = tv.y0, x1 = tv.x1, y1 = tv.y1, x2 = tv.x2, y2 = tv.y2;
    const float e01x = x1 - x0, e01y = y1 - y0, e02x = x2 - x0, e02y = y2 - y0, e12x = x2 - x1, e12y = y2 - y1;
    const float l01 = fmaf(e01x, e01x, e01y * e01y), l02 = fmaf(e02x, e02x, e02y * e02y), l12 = fmaf(e12x, e12x, e12y * e12y);
    const float rl01 = l01 <= K_EPS ? 0.f : __builtin_amdgcn_rcpf(l01);
    const float rl02 = l02 <= K_EPS ? 0.f : __builtin_amdgcn_rcpf(l02);
    rl12_out = l12 <= K_EPS ? 0.f : __builtin_amdgcn_rcpf(l12);
    r3 = make_float4(x0 - cx, x1 - cx, y0 - cy, y1 - cy);
    r4 = make_float4(e01x, e02x, e01y, e02y);
    r5 = make_float4(rl01, rl02, e12x, e12y);
}
__device__ __forceinline__ void stage_faces(const RasterArgs &a, const Tri9 &tv, int i0, int i1, int i2, int m,
                                            float *rec, int lane, float cx, float cy, float fS, int tx, int ty, int ox0, int ox1,
                                            int oy0, int oy1, unsigned long long open_px, int &cf, int &packed2, float2 *__restrict__ sxy,
                                            TriIds *__restrict__ sid, int c0, int list_stride) {
    const int slot = lane & (DCHUNK - 1);
    const bool hi = lane >= DCHUNK;
    int b0 = 0, b1 = -1;  // low lane: box rows by0 .. by1; high lane: box columns bx0 .. bx1
    if (slot < m) {
        const float x0 = tv.x0, y0 = tv.y0, x1 = tv.x1, y1 = tv.y1, x2 = tv.x2, y2 = tv.y2;
        float4 *r = reinterpret_cast<float4 *>(rec + slot * FSTR);
        // The tile's vertex table for pass 3: three arrays of float2 (v0, v1, v2 by list position) and the vertex ids, so that every
        // store instruction writes whole runs of bytes (one 24-byte structure per face, stored as 16 + 8 bytes, cost 0.45 ms per
        // cfg2b launch in partial-line writes; this form 0.1): v0 from the low lanes and v1 from the high lanes in one instruction
        at(sxy, (uint32_t)((hi ? list_stride : 0) + c0 + slot)) = hi ? make_float2(x1, y1) : make_float2(x0, y0);
        if (!hi) {
            at(sid, (uint32_t)(c0 + slot)) = TriIds{i0, i1, i2};
            face_rows_lo(tv, cx, cy, r[0], r[1], r[2]);
            const float ymin = fminf(fminf(y0, y1), y2) - a.sqrt_blur, ymax = fmaxf(fmaxf(y0, y1), y2) + a.sqrt_blur;
            const int yi_lo = (int)ceilf(((ymin + 1.0f) * fS - 1.0f) * 0.5f - 0.01f), yi_hi = (int)floorf(((ymax + 1.0f) * fS - 1.0f) * 0.5f + 0.01f);
            b0 = max(a.S - 1 - yi_hi - ty * TILE, oy0);
            b1 = min(a.S - 1 - yi_lo - ty * TILE, oy1);
        } else {
            const float xmin = fminf(fminf(x0, x1), x2) - a.sqrt_blur, xmax = fmaxf(fmaxf(x0, x1), x2) + a.sqrt_blur;
            float rl12;
            face_rows_hi(tv, cx, cy, r[3], r[4], r[5], rl12);
            r[6] = make_float4(rl12, __int_as_float(i0), __int_as_float(i1), __int_as_float(i2));
            at(sxy, (uint32_t)(2 * list_stride + c0 + slot)) = make_float2(x2, y2);  // (the table, see above)
            const int xi_lo = (int)ceilf(((xmin + 1.0f) * fS - 1.0f) * 0.5f - 0.01f), xi_hi = (int)floorf(((xmax + 1.0f) * fS - 1.0f) * 0.5f + 0.01f);
            b0 = max(a.S - 1 - xi_hi - tx * TILE, ox0);
            b1 = min(a.S - 1 - xi_lo - tx * TILE, ox1);
        }
    }
    int bx0 = __shfl(b0, slot + DCHUNK, WAVE), bx1 = __shfl(b1, slot + DCHUNK, WAVE);  // (all lanes: no divergent ds_bpermute)
    cf = 0;
    packed2 = 0;
    if (!hi && slot < m && bx0 <= bx1 && b0 <= b1) {
        // shrink the box to the open pixels inside it (bit 8 * row + column of `open_px`): between two closing steps the
        // open pixels of a tile are often scattered, and their common bounding box (ox0 .. oy1) is then the whole tile
        const uint32_t colb = ((2u << bx1) - (1u << bx0)) * 0x01010101u;                  // columns bx0 .. bx1 of every row
        const unsigned long long rows = (b1 >= 7 ? ~0ull : ((1ull << (8 * (b1 + 1))) - 1ull)) & ~((1ull << (8 * b0)) - 1ull);
        const unsigned long long mbox = open_px & rows & (((unsigned long long)colb << 32) | colb);
        if (mbox == 0ull) {
            b1 = b0 - 1;
        } else {
            b0 = (int)(__builtin_ctzll(mbox) >> 3);
            b1 = (int)((63 - __builtin_clzll(mbox)) >> 3);
            uint32_t c8 = (uint32_t)mbox | (uint32_t)(mbox >> 32);
            c8 |= c8 >> 16;
            c8 = (c8 | (c8 >> 8)) & 0xFFu;
            bx0 = (int)__builtin_ctz(c8);
            bx1 = 31 - (int)__builtin_clz(c8);
        }
    }
    if (!hi && slot < m && bx0 <= bx1 && b0 <= b1) {
        // the box in PIXEL PAIRS (columns 2c, 2c + 1 of one row; pair index = pixel index / 2): a lane of the sweep evaluates one
        const int cp0 = bx0 >> 1, bw = (bx1 >> 1) - cp0 + 1;   // 1 ... 4 pairs per row
        cf = bw * (b1 - b0 + 1);
        // what a lane needs to find its pixel pair: lane r of the box sits in box row r / bw, computed as (r * inv) >> 16 with
        // inv = floor(65536 / bw) + 1 (exact for r < 32, bw <= 4: the reciprocal is exact for 1, 2, 4 and 1e-4 away from an
        // integer otherwise), and its pair is first + r + (r / bw) * (4 - bw)
        const int inv = (int)(65536.0f * __builtin_amdgcn_rcpf((float)bw)) + 1;
        packed2 = inv | ((TILE / 2 - bw) << 17) | ((b0 * (TILE / 2) + cp0) << 20);
    }
}

// lane = pixel: in the histogram `hist` ([bucket / 2][pixel]) find the digit that holds the `need`-th smallest key.
// Returns the number of keys counted for this pixel; updates (pre, need) and reports the count of the chosen digit.
__device__ __forceinline__ int pick_digit(const uint32_t *hist, int lane, int b, uint32_t &pre, int &need, int &n_eq) {
    uint32_t hw[(1 << SEL_BITS) / 2];
#pragma unroll
    for (int w_ = 0; w_ < (1 << SEL_BITS) / 2; ++w_) hw[w_] = hist[w_ * WAVE + lane];  // (all reads in flight together)
    // Running sums c_k are non-decreasing: the chosen bucket is the number of c_k below `need`, the keys in lower buckets the largest
    // such c_k, and the chosen bucket ends at the first c_k that reaches `need`.  Arithmetic selects only (as a chain of `if`s this
    // was thirty-two exec-masked branches per call).
    int c = 0, sel = 0, below = 0, first_ge = 0x7FFFFFFF;
#pragma unroll
    for (int k = 0; k < (1 << SEL_BITS); ++k) {
        c += (k & 1) ? (int)(hw[k >> 1] >> 16) : (int)(hw[k >> 1] & 0xFFFFu);
        const bool lt = c < need;
        sel += lt ? 1 : 0;
        below = lt ? c : below;
        first_ge = min(first_ge, lt ? 0x7FFFFFFF : c);
    }
    if (need > 0 && c >= need) {
        pre = (pre << b) | (uint32_t)sel;
        need -= below;
        n_eq = first_ge - below;
    } else {
        need = 0;  // fewer keys than the rank asked for: this pixel keeps everything
        n_eq = 0;
    }
    return c;
}

// The same for the first digit, whose histogram holds four 8-bit counts per word ([bucket / 4][pixel]) that stop growing at
// SAT8 > K: the cumulative counts below the chosen digit are exact (they are below `need` <= K), a count that reached SAT8
// only ever compares as "more than K".
__device__ __forceinline__ int pick_digit8(const uint32_t *hist, int lane, int b, uint32_t &pre, int &need, int &n_eq) {
    int cum = 0, sel = 0, cnt_sel = 0, all = 0;
    bool found = false;
#pragma unroll
    for (int w_ = 0; w_ < (1 << SEL1_BITS) / 4; ++w_) {
        const uint32_t hw = hist[w_ * WAVE + lane];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int h = (int)((hw >> (8 * q)) & 0xFFu);
            all += h;
            if (!found && cum + h >= need) { sel = 4 * w_ + q; cnt_sel = h; found = true; }
            cum += found ? 0 : h;
        }
    }
    if (need > 0 && found) {
        pre = (pre << b) | (uint32_t)sel;
        need -= cum;
        n_eq = cnt_sel;
    } else {
        need = 0;
        n_eq = 0;
    }
    return all;
}

// One radix-select sweep over `n_rec` (key, meta) pairs: among the keys of pixel p whose bits above `nbits` equal
// psel[p].x, histogram the next `b` bits (psel[p].y == 0: pixel not taking part; key 0xFFFFFFFF: record not taking part).
template <typename KeyFn>
__device__ __forceinline__ void select_sweep(DenseLds &lds, const Rec3 *__restrict__ crec, int n_rec, int nbits, int b,
                                             int lane, uint32_t pre, int need, KeyFn key_of) {
    const int shift = nbits - b;
    lds.psel[lane] = make_uint2(pre, (uint32_t)need);
    for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
    __syncthreads();
    auto load_keys = [&](uint32_t (&kk)[KGROUP], uint32_t (&mt)[KGROUP], int g0) {
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) {
            const uint32_t idx = (uint32_t)min(g0 + u * WAVE + lane, n_rec - 1);  // unsigned 32-bit: SGPR base + VGPR offset addressing
            mt[u] = at(crec, idx).b;
            kk[u] = key_of(idx, mt[u]);
        }
    };
    auto count_keys = [&](const uint32_t (&kk)[KGROUP], const uint32_t (&mt)[KGROUP], int g0) {
        uint2 ps[KGROUP];  // all LDS gathers first: one latency, not one per row
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) ps[u] = lds.psel[mt[u] & 63u];
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) {
            const uint32_t pxl = mt[u] & 63u;
            const bool hit = (g0 + u * WAVE + lane < n_rec) & (ps[u].y > 0u) & ((kk[u] >> nbits) == ps[u].x) & (kk[u] != 0xFFFFFFFFu);
            const uint32_t bucket = (kk[u] >> shift) & ((1u << b) - 1u);
            if (hit) atomicAdd(&lds.hist[(bucket >> 1) * WAVE + pxl], (bucket & 1u) ? 0x10000u : 1u);
        }
    };
    if (n_rec > 0) {  // double-buffered: the next KGROUP rows are in flight while this one is counted
        uint32_t ka[KGROUP], ma[KGROUP], kb[KGROUP], mb[KGROUP];
        load_keys(ka, ma, 0);
        for (int g0 = 0; g0 < n_rec; g0 += 2 * KGROUP * WAVE) {
            load_keys(kb, mb, g0 + KGROUP * WAVE);
            count_keys(ka, ma, g0);
            load_keys(ka, ma, g0 + 2 * KGROUP * WAVE);
            count_keys(kb, mb, g0 + KGROUP * WAVE);
        }
    }
    __syncthreads();
}

// One refinement step of the radix select over the compact stream, which SHRINKS as it goes.  Among the records of pixel p
// (still selecting: psel[p].y > 0) the bits of the key above `nbits` are compared with the prefix psel[p].x chosen so far:
//   below it  -> the record lies in a lower bucket of the digit picked last: it is among the K nearest for certain, its log
//                factor goes to the pixel's sum and the record leaves the stream;
//   equal     -> it stays (compacted IN PLACE: the write position never passes the read position, and every lane has
//                loaded its record before any lane of the same step stores) and its next `b` bits are histogrammed;
//   above     -> dropped.
// Returns the number of records left.  Every later sweep thus reads only the records that are still undecided (a tenth per
// digit) instead of the whole compact stream, and the final pass only sees the last bucket.
__device__ __forceinline__ int refine_sweep(DenseLds &lds, Rec3 *crec, int n_rec, int nbits, int b,
                                            int lane, uint32_t pre, int need) {
    const int shift = nbits - b;
    lds.psel[lane] = make_uint2(pre, (uint32_t)need);
    for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
    __syncthreads();
    int n_out = 0;
    struct CRec { uint32_t kk, mt; float lf; };
    auto load_recs = [&](CRec (&r)[KGROUP], int g0) {
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) {
            const uint32_t idx = (uint32_t)min(g0 + u * WAVE + lane, n_rec - 1);
            const Rec3 q = at(crec, idx);
            r[u].kk = q.a; r[u].mt = q.b; r[u].lf = __uint_as_float(q.c);
        }
    };
    auto sift_recs = [&](const CRec (&r)[KGROUP], int g0) {
        uint2 ps[KGROUP];  // all LDS gathers first: one latency, not one per row
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) ps[u] = lds.psel[r[u].mt & 63u];
#pragma unroll
        for (int u = 0; u < KGROUP; ++u) {
            const uint32_t pxl = r[u].mt & 63u;
            const bool live = (g0 + u * WAVE + lane < n_rec) & (ps[u].y > 0u);
            const uint32_t top = r[u].kk >> nbits;
            const bool sure = live & (top < ps[u].x), stay = live & (top == ps[u].x);
            if (sure & (r[u].lf != 0.f)) atomicAdd(&lds.plog[pxl], (double)r[u].lf);
            const unsigned long long sm = __ballot(stay);
            const uint32_t slot = (uint32_t)n_out + __builtin_amdgcn_mbcnt_hi((uint32_t)(sm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)sm, 0u));
            if (stay) {
                at(crec, slot) = Rec3{r[u].kk, r[u].mt, __float_as_uint(r[u].lf)};
                const uint32_t bucket = (r[u].kk >> shift) & ((1u << b) - 1u);
                atomicAdd(&lds.hist[(bucket >> 1) * WAVE + pxl], (bucket & 1u) ? 0x10000u : 1u);
            }
            n_out += __popcll(sm);
        }
    };
    if (n_rec > 0) {  // double-buffered: the next KGROUP rows are in flight while this one is sifted
        CRec ra[KGROUP], rb[KGROUP];
        load_recs(ra, 0);
        for (int g0 = 0; g0 < n_rec; g0 += 2 * KGROUP * WAVE) {
            load_recs(rb, g0 + KGROUP * WAVE);
            sift_recs(ra, g0);
            load_recs(ra, g0 + 2 * KGROUP * WAVE);
            sift_recs(rb, g0 + KGROUP * WAVE);
        }
    }
    __syncthreads();
    return n_out;
}

template <int MODE>
__global__ void __launch_bounds__(64, WAVES_PER_SIMD) k_raster_dense(RasterArgs a) {
    __shared__ DenseLds lds;
    const int lane = threadIdx.x;
    uint2 *const slist = a.slist + (size_t)blockIdx.x * a.list_stride;
    uint32_t *const slist2 = a.slist2 + (size_t)blockIdx.x * a.list_stride;
    uint32_t *const scfirst = a.scfirst + (size_t)blockIdx.x * a.n_cf;
    float2 *const sxy = a.sxy + (size_t)blockIdx.x * 3 * a.list_stride;
    TriIds *const sid = a.sid + (size_t)blockIdx.x * a.list_stride;
    const size_t rec0 = (size_t)blockIdx.x * (REC_CAP + REC_PAD);
    Rec3 *const srec = a.srec + rec0, *const crec = a.crec + rec0;
    const uint32_t lane_lo = lane < 32 ? 1u << lane : 0u, lane_hi = lane >= 32 ? 1u << (lane - 32) : 0u;
    const int K = a.K;
    const int n_tiles = a.tiles_x * a.tiles_x;
    unsigned int n_items_all = 0;
    for (int q = 0; q < N_PARTS; ++q)
        for (int c = 0; c < N_CLASSES; ++c) n_items_all += a.ctr->n_class[q][c];
    // With fewer tiles than workgroups (a handful of images) every tile is dealt out as 2, 4 or 8 runs of pixels, so that
    // the launch finishes in a fraction of one tile's serial time.  Round 4, from a sweep over 1 ... 64 images x workgroups per CU x
    // pieces (profiles/r4_small_launches.txt): the launch is fastest with ~2.3 pieces per WORKING workgroup and about 1.2 pieces per
    // resident slot in all (8-pixel pieces - a whole list walk for one row of pixels - only while even they number under 0.6 per
    // slot); the workgroups beyond that leave at once - a one-image launch runs on 512 of them, not on 4 096 that queue for the same
    // ticket counter.
    const unsigned int slots = a.slots;  // (the device's resident slots; the grid may be smaller: tile_grid)
    const unsigned int split_log = HOOK_SPLIT_LOG(deal_split_log(n_items_all, slots));
    if (blockIdx.x >= deal_working(n_items_all, split_log, slots)) return;  // (workgroup-uniform, before any barrier)
    // The heaviest class can be dealt out in 2^SPLIT0_LOG pieces of pixels (see SPLIT0_LOG; off since the lists are walked
    // near to far) - and is, like the others, when the launch has workgroups to spare.
    const unsigned int split0_log = SPLIT0_LOG > split_log ? SPLIT0_LOG : split_log;
    const float fS = (float)a.S;
    unsigned int xcc;  // the XCD this workgroup runs on: which partition it drains first
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= (unsigned int)(N_PARTS - 1);

    TIMERS_INIT
    for (unsigned int turn = 0; turn < N_PARTS; ++turn) {
    const unsigned int part = (xcc + turn) & (unsigned int)(N_PARTS - 1);
    const unsigned int nc0 = a.ctr->n_class[part][0], nc1 = a.ctr->n_class[part][1], nc2 = a.ctr->n_class[part][2], nc3 = a.ctr->n_class[part][3];
    const unsigned int n_items = nc0 + nc1 + nc2 + nc3;
    const unsigned int units0 = nc0 << split0_log;
    const unsigned int n_units = units0 + ((n_items - nc0) << split_log);
    const uint4 *const items = a.items + (size_t)part * 2u * a.item_cap;
    while (n_units > 0u) {
        unsigned int unit = 0;
        TSUB(0)
        if (lane == 0) unit = atomicAdd(&a.ctr->deal[part].next, 1u);
        unit = __builtin_amdgcn_readfirstlane(unit);
        if (unit >= n_units) break;
        TUNIT_START
        const bool heavy = unit < units0;
        const unsigned int sl = heavy ? split0_log : split_log, u_ = heavy ? unit : unit - units0;
        const unsigned int item = (u_ >> sl) + (heavy ? 0u : nc0);
        const int p_begin = (int)(u_ & ((1u << sl) - 1u)) * (WAVE >> sl), p_end = p_begin + (WAVE >> sl);
        // heaviest class first
        const uint32_t item_at = item < nc0 ? item
                               : item < nc0 + nc1 ? a.item_cap - 1u - (item - nc0)
                               : item < nc0 + nc1 + nc2 ? a.item_cap + (item - nc0 - nc1)
                               : 2u * a.item_cap - 1u - (item - nc0 - nc1 - nc2);
        const uint4 it = items[item_at];
        const uint32_t code = it.x;
        const int n = (int)(code / (uint32_t)n_tiles), tile = (int)(code % (uint32_t)n_tiles);
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const int xo = tx * TILE + (lane & 7), yo = ty * TILE + (lane >> 3);
        const bool in_img = xo < a.S && yo < a.S;
        const float cx = pix_to_ndc(a.S - 1 - (tx * TILE + 4), a.S), cy = pix_to_ndc(a.S - 1 - (ty * TILE + 4), a.S);
        const float *vn = a.verts_ndc + (size_t)n * a.V * 3;
        const float *const xv_n = a.clip.xv + (size_t)n * CLIP_VX * 3;   // the image's clip tables (touched only by cut faces)
        const int *const xf_n = a.clip.xf + (size_t)n * CLIP_FX * 3;
        const size_t pix = ((size_t)n * a.S + yo) * a.S + xo;

        uint32_t kmin, kmax;  // bounds of the depth keys of this tile
        // the faces whose blurred box reaches this tile: binned by the setup kernel (any order), or found here through the tile
        // boxes of the 64-face groups (ascending id) when the image's lists did not fit
        const uint2 td = make_uint2(it.y, it.z);
        const bool binned = td.y != 0xFFFFFFFFu;  // (wave-uniform)
        TSUB(1)
        const uint2 *const list_src = binned ? a.lists + (size_t)n * a.list_cap + td.x : slist;
        int list_total;
        if (binned) {
            // every depth of the tile lies between the nearest vertex of its nearest face and the farthest vertex of any: the entries
            // carry the nearest depth only (8 bytes), and farthest <= nearest + (largest depth extent of a face of the image, from
            // the setup kernel with the work item) bounds the other end - an upper bound is all the key range needs
            list_total = (int)td.y;
            uint32_t lo = 0x7F7FFFFFu, hi = 0u;
            for (int i0 = 0; i0 < list_total; i0 += 4 * WAVE) {
                uint2 e[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) e[u] = at(list_src, (uint32_t)min(i0 + u * WAVE + lane, list_total - 1));
#pragma unroll
                for (int u = 0; u < 4; ++u) { lo = min(lo, e[u].y); hi = max(hi, e[u].y); }
            }
            for (int o = 32; o > 0; o >>= 1) {
                lo = min(lo, (uint32_t)__shfl_xor((int)lo, o, WAVE));
                hi = max(hi, (uint32_t)__shfl_xor((int)hi, o, WAVE));
            }
            kmin = lo;
            // (rounded up twice: the extent was a rounded difference, the sum rounds again)
            kmax = __float_as_uint((__uint_as_float(hi) + __uint_as_float(it.w) * 1.000001f) * 1.0000005f) + 1u;
        } else {
            list_total = build_list(a, n, tx, ty, slist, lane, kmin, kmax);
        }
        const bool may_truncate = list_total > K;
        TSUB(2)
        // radix select on key = depth bits - kmin, which lies in [0, kmax - kmin]: `nbits0` significant bits, of which the
        // first digit takes the top SEL_BITS (so it always spreads over at least half of its buckets)
        const uint32_t krange = kmax - kmin;
        const int nbits0 = krange ? 32 - __clz(krange) : 0;
        const int b1 = min(SEL1_BITS, nbits0), shift1 = nbits0 - b1;
        __syncthreads();  // the list stores are visible to the loads below
        // tiles that may truncate walk their faces near to far (see sort_list_near_to_far); the others keep the id order
        const uint32_t *const lst = slist2;
        if (may_truncate) {
            sort_list_near_to_far(list_src, slist2, list_total, kmin, shift1, b1, lds, lane);
        } else {  // at most K faces: the order of the list is kept
            for (int i = lane; i < list_total; i += WAVE) slist2[i] = at(list_src, (uint32_t)i).x;
            __syncthreads();
        }
        TMARK(0)
        TSUB(3)
        HOOK_STOP_AFTER(0, continue)

        unsigned long long tie_acc = 0ull;  // (tie_rule 1) pixels of this unit left to k_raster_tie_replay
        // Sub-tiles: runs of `span` pixels (lane order).  Start from an estimate (a quarter of the pairs pixel x face
        // exist) and halve whenever pass 1 finds that the records do not fit; span * list_total <= REC_CAP always fits.
        int span = p_end - p_begin;
        while (span > 1 && (long long)span * list_total > 4ll * REC_CAP) span >>= 1;
        for (int p_lo = p_begin; p_lo < p_end;) {
            const bool mine = lane >= p_lo && lane < p_lo + span;  // this lane's pixel belongs to the sub-tile
            const int sy0 = p_lo >> 3, sy1 = (p_lo + span - 1) >> 3;                       // its rows ...
            const int sx0 = span >= 8 ? 0 : (p_lo & 7), sx1 = span >= 8 ? 7 : ((p_lo & 7) + span - 1);  // ... and columns
            // pixels outside the image or the sub-tile get a position no bbox can contain
            const float px = (in_img && mine) ? pix_to_ndc(a.S - 1 - xo, a.S) : 3.0e38f, py = pix_to_ndc(a.S - 1 - yo, a.S);
            lds.pixt[lane] = make_float2(px, py);
            if (may_truncate)
                for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
            __syncthreads();

            // ---------------- pass 1: every pair inside a face's pixel box, once --------------------------------
            int vbase = 0;  // records written so far (wave-uniform)
            bool fits = true;
            // Pixels that cannot keep any further record ("closed"): they already hold K records in depth digits that are
            // final, i.e. below the digit of the first face not yet processed.  Lane = pixel keeps its count of final records.
            int final_digits = 0, final_cnt = 0;
            unsigned long long open_px = __ballot(in_img && mine);
            int ox0 = sx0, ox1 = sx1, oy0 = sy0, oy1 = sy1;  // bounding box of the open pixels
            int chunks_done = 0;
            // first record of every chunk (pass 3 walks the records group by group): lane c of `cst0` / `cst1` holds the start of chunk c /
            // 64 + c - a register read in pass 3 instead of a memory round trip per group; chunks from 128 on (lists beyond 4096 faces)
            // go through memory
            uint32_t cst0 = 0u, cst1 = 0u;
            auto set_chunk_start = [&](int c, uint32_t v) {  // (c, v wave-uniform)
                if (c < WAVE) cst0 = lane == c ? v : cst0;
                else if (c < 2 * WAVE) cst1 = lane == c - WAVE ? v : cst1;
                else if (lane == 0) scfirst[c] = v;
            };
            auto chunk_start = [&](int c) -> uint32_t {
                if (c < WAVE) return (uint32_t)__builtin_amdgcn_readlane((int)cst0, c);
                if (c < 2 * WAVE) return (uint32_t)__builtin_amdgcn_readlane((int)cst1, c - WAVE);
                return (uint32_t)__builtin_amdgcn_readfirstlane((int)scfirst[c]);
            };
            // The staging loads form a chain list entry -> vertex indices -> vertex coordinates.  The first two links are
            // fetched ahead: while chunk k is evaluated the indices of chunk k + 1 and the list entries of chunk k + 2 are in
            // flight (four registers), so a chunk starts with one memory round trip instead of three.
            const int slot_ = lane & (DCHUNK - 1);
            auto list_at = [&](int c) { return (int)lst[min(c + slot_, list_total - 1)]; };
            // (round 4: one more link ahead - the VERTICES of chunk k + 1 are requested before chunk k is evaluated and wait in nine
            // registers, so a chunk starts with the drain of the previous sweep's stores only, not with a vertex fetch behind it)
            int f_nx = list_at(2 * DCHUNK);
            int ia, ib, ic;      // vertex ids of chunk k + 1 ...
            int ja, jb, jc;      // ... and of chunk k
            Tri9 tv_nx;          // vertices of chunk k (requested one chunk ahead)
            { const int f_ = list_at(0); ja = face_vertex(a.faces, xf_n, a.F, f_, 0); jb = face_vertex(a.faces, xf_n, a.F, f_, 1); jc = face_vertex(a.faces, xf_n, a.F, f_, 2); }
            { const int f_ = list_at(DCHUNK); ia = face_vertex(a.faces, xf_n, a.F, f_, 0); ib = face_vertex(a.faces, xf_n, a.F, f_, 1); ic = face_vertex(a.faces, xf_n, a.F, f_, 2); }
            tv_nx = load_tri(a, vn, xv_n, ja, jb, jc);
            for (int c0 = 0; c0 < list_total; c0 += DCHUNK) {
                if (may_truncate) {
                    // digit of this chunk's first face = number of buckets that start at or before it, minus one
                    const int d0 = __popcll(__ballot(lane < (1 << b1) && (int)lds.bstart[lane] <= c0)) - 1;
                    if (d0 > final_digits) {  // wave-uniform: digits [final_digits, d0) have just become final
                        for (int d = final_digits; d < d0; ++d)
                            final_cnt += (int)((lds.hist[(d >> 2) * WAVE + lane] >> (8 * (d & 3))) & 0xFFu);
                        final_digits = d0;
                        open_px &= ~__ballot(final_cnt >= K);
                        if (open_px == 0ull) break;  // every pixel of the (sub-)tile is closed: the remaining faces are all farther
                        unsigned int cols = 0u;
                        oy0 = 8; oy1 = -1;
                        for (int y = 0; y < TILE; ++y) {
                            const unsigned int row = (unsigned int)(open_px >> (8 * y)) & 0xFFu;
                            cols |= row;
                            if (row) { oy0 = min(oy0, y); oy1 = y; }
                        }
                        ox0 = (int)__builtin_ctz(cols); ox1 = 31 - (int)__builtin_clz(cols);
                    }
                }
                const int m = min(DCHUNK, list_total - c0);
                int cf, packed2, packed = 0;
                const int i0 = ja, i1 = jb, i2 = jc;
                const Tri9 tv = tv_nx;                       // this chunk's vertices (in flight since the chunk before)
                ja = ia; jb = ib; jc = ic;
                tv_nx = load_tri(a, vn, xv_n, ja, jb, jc);  // chunk c0 + DCHUNK
                ia = face_vertex(a.faces, xf_n, a.F, f_nx, 0); ib = face_vertex(a.faces, xf_n, a.F, f_nx, 1); ic = face_vertex(a.faces, xf_n, a.F, f_nx, 2);  // chunk c0 + 2 DCHUNK
                f_nx = list_at(c0 + 3 * DCHUNK);
                stage_faces(a, tv, i0, i1, i2, m, lds.rec, lane, cx, cy, fS, tx, ty, ox0, ox1, oy0, oy1, open_px, cf, packed2, sxy, sid, c0, a.list_stride);
                set_chunk_start(c0 / DCHUNK, (uint32_t)vbase);
                chunks_done = c0 / DCHUNK + 1;
                lds_fence();
                HOOK_STOP_AFTER(1, continue)
                const int incl = wave_scan_add(cf);
                const int off = incl - cf;          // first pair of this face in the chunk's pair list
                const int n_pairs = __builtin_amdgcn_readlane(incl, 63);
                packed |= off;                      // off <= DCHUNK * 32
                if (vbase + 2 * n_pairs > REC_CAP) { fits = false; break; }  // wave-uniform (n_pairs lanes of two pixels each)
                STAT(20, 2 * n_pairs)
                // pair -> face.  Every non-empty face sets the bit of its first pair in a 2048-bit map (64 words in LDS) and
                // leaves its packed box at its rank among the non-empty faces.  Lane i then keeps words 2i, 2i+1 - the start
                // bits of sweep step i - and the packed box of rank i; in step i a pair's face is (starts before the step) +
                // (start bits at or below its lane) - 1, two v_mbcnt and one ds_bpermute away.
                const unsigned long long nonempty = __ballot(cf > 0);
                STAT(31, __popcll(nonempty))  // staged faces that have any open pixel in their box
                lds.start[lane] = 0;
                lds_fence();
                if (cf > 0) {
                    atomicOr(reinterpret_cast<uint32_t *>(lds.start) + (off >> 5), 1u << (off & 31));
                    const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(nonempty >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)nonempty, 0u));
                    lds.psel[rank] = make_uint2((uint32_t)packed | ((uint32_t)lane << 13), (uint32_t)packed2);
                }
                lds_fence();
                const uint32_t fl_lo = reinterpret_cast<const uint32_t *>(lds.start)[(2 * lane) & 63];
                const uint32_t fl_hi = reinterpret_cast<const uint32_t *>(lds.start)[(2 * lane + 1) & 63];
                const uint2 pk_rank = lds.psel[lane & (DCHUNK - 1)];
                lds_fence();
                TSTAGE_MARK
                uint32_t carry = 0;                 // faces started before this step
                for (int q0 = 0; q0 < n_pairs; q0 += WAVE) {
                    const uint32_t wlo = (uint32_t)__builtin_amdgcn_readlane((int)fl_lo, q0 >> 6), whi = (uint32_t)__builtin_amdgcn_readlane((int)fl_hi, q0 >> 6);
                    const uint32_t below = __builtin_amdgcn_mbcnt_hi(whi, __builtin_amdgcn_mbcnt_lo(wlo, 0u));
                    const uint32_t own = ((wlo & lane_lo) | (whi & lane_hi)) ? 1u : 0u;
                    const int r = min((int)(carry + below + own) - 1, DCHUNK - 1);
                    carry += (uint32_t)(__popc(wlo) + __popc(whi));
                    const bool valid = q0 + lane < n_pairs;
                    const uint32_t pk = (uint32_t)__shfl((int)pk_rank.x, max(r, 0), WAVE), pk2 = (uint32_t)__shfl((int)pk_rank.y, max(r, 0), WAVE);
                    const int fs = (int)((pk >> 13) & (DCHUNK - 1));
                    const uint32_t rr = (uint32_t)(q0 + lane) - (pk & 0x1FFFu);
                    // rr < 32 and the reciprocal has 17 bits: a 24-bit multiply (full rate) is exact.  Spelled in assembly because
                    // hipcc widens __umul24 here to the quarter-rate v_mul_lo_u32 (it cannot see the range of rr)
                    uint32_t rr_inv;
                    asm("v_mul_u32_u24 %0, %1, %2" : "=v"(rr_inv) : "v"(rr), "v"(pk2 & 0x1FFFFu));
                    const uint32_t dy = rr_inv >> 16;
                    const int pp = (int)((__umul24(dy, (pk2 >> 17) & 3u) + rr + (pk2 >> 20)) & 31u);  // pixel pair: pixels 2 pp, 2 pp + 1
                    const int p = 2 * pp;
                    const float4 pc = *reinterpret_cast<const float4 *>(&lds.pixt[p]);   // (px, py) of both pixels: one 16-byte read
                    const FaceRows fr = load_face_rows(lds.rec + fs * FSTR);
                    PairEval2 e;
                    eval_pair2(fr, pc.x - cx, pc.z - cx, pc.y - cy, a.blur, e);
                    HOOK_EXTRA_VALU(pc)
                    const uint32_t open2 = (uint32_t)(open_px >> p) & 3u;
                    const bool cand0 = valid && e.cand0 && (open2 & 1u), cand1 = valid && e.cand1 && (open2 & 2u);
                    const unsigned long long cm0 = __ballot(cand0), cm1 = __ballot(cand1);
                    if ((cm0 | cm1) == 0ull) continue;
                    // depth: kept inside the tile's vertex-depth range, where the convex combination lives up to rounding
                    // ... and never nearer than the face's nearest vertex (rounding of the convex combination), so that a record's
                    // digit is at least its face's: the closing rule above relies on it
                    uint32_t zb0 = 0x7F61B1E6u, zb1 = 0x7F61B1E6u;  // (3.0e38f: tiles that cannot truncate carry no depths)
                    if (may_truncate) {
                        const f32x2 z2 = pair_depth2(fr, e);
                        const float zf = fminf(fminf(fr.r2.y, fr.r2.z), fr.r2.w);
                        zb0 = min(max(__float_as_uint(vmax_raw(z2.x, zf)), kmin), kmax);
                        zb1 = min(max(__float_as_uint(vmax_raw(z2.y, zf)), kmin), kmax);
                    }
                    // a step's left-pixel records first, then its right-pixel records: each of the two store instructions writes ONE
                    // contiguous run of 12-byte records (interleaved - a lane's two records next to each other - both instructions
                    // touched every cache line of the step's run, each with half of the bytes: the record stores are 1.35 ms of the
                    // cfg2b launch, profiles/r6_experiments.md).  No sweep depends on the order of the records inside a chunk.
                    const uint32_t slot0 = (uint32_t)vbase + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm0 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm0, 0u));
                    const uint32_t slot1 = (uint32_t)vbase + (uint32_t)__popcll(cm0) + __builtin_amdgcn_mbcnt_hi((uint32_t)(cm1 >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm1, 0u));
                    const uint32_t meta = (uint32_t)p | ((uint32_t)(c0 + fs) << 6);
                    if (cand0) {
                        st_stream(srec, slot0, Rec3{zb0, meta | (e.inside0 ? 1u << 22 : 0u) | e.ebits0, __float_as_uint(e.sd.x)});
                        if (may_truncate) {  // first radix digit, and with it the number of candidates of the pixel
                            const uint32_t bucket = ((zb0 - kmin) >> shift1) & ((1u << b1) - 1u);
                            uint32_t *const hw = &lds.hist[(bucket >> 2) * WAVE + p];
                            const uint32_t sh = 8u * (bucket & 3u);
                            if (((*hw >> sh) & 0xFFu) < SAT8) atomicAdd(hw, 1u << sh);
                        }
                    }
                    if (cand1) {
                        st_stream(srec, slot1, Rec3{zb1, (meta + 1u) | (e.inside1 ? 1u << 22 : 0u) | e.ebits1, __float_as_uint(e.sd.y)});
                        if (may_truncate) {
                            const uint32_t bucket = ((zb1 - kmin) >> shift1) & ((1u << b1) - 1u);
                            uint32_t *const hw = &lds.hist[(bucket >> 2) * WAVE + p + 1];
                            const uint32_t sh = 8u * (bucket & 3u);
                            if (((*hw >> sh) & 0xFFu) < SAT8) atomicAdd(hw, 1u << sh);
                        }
                    }
                    vbase += __popcll(cm0) + __popcll(cm1);
                }
                TSWEEP_MARK
                lds_fence();  // rec is rewritten by the next chunk
            }
            if (!fits) {  // wave-uniform: try again with half the pixels
                span >>= 1;
                __syncthreads();
                continue;
            }
            set_chunk_start(chunks_done, (uint32_t)vbase);  // (chunks behind an early exit hold no records)
            STAT(21, vbase) STAT(26, 1) STAT(27, list_total) STAT(28, chunks_done) STAT(29, (list_total + DCHUNK - 1) / DCHUNK) STAT(30, __popcll(open_px))
            __syncthreads();  // also: record stores of other lanes are visible from here on
            TMARK(1)
            HOOK_STOP_AFTER(1, { p_lo += span; continue; }) HOOK_STOP_AFTER(2, { p_lo += span; continue; })

            // ---------------- select + pass 2 ---------------------------------------------------------------------
            // K-th smallest depth of every pixel that has more than K candidates, and log2 of every kept blend factor summed
            // per pixel.  threshold: depth bits of the K-th smallest (0x7F800000 = +inf bits: keep everything); tie_cut: among
            // the faces exactly at the threshold those up to this list position are kept.
            uint32_t zt_bits = 0x7F800000u;
            int tie_cut = 0x7FFFFFFF;
            uint32_t pre = 0u;
            int need = 0, n_eq = 0, nbits = nbits0 - b1;
            bool trunc = false;
            bool defer = false;  // (tie_rule 1) this pixel's tie group at the K-th depth is cut by K: k_raster_tie_replay renders it
            if (may_truncate && vbase > 0) {
                need = K;
                const int tot = pick_digit8(lds.hist, lane, b1, pre, need, n_eq);
                trunc = tot > K;
                if (!trunc) need = 0;
            }
            const bool any_trunc = __ballot(trunc) != 0ull;
            // One sweep over all records.  A record of a pixel that is not truncated, or whose first digit is below the
            // pixel's chosen one, is kept for certain: its log goes to the pixel's sum.  One inside the chosen digit goes on
            // to the compact stream (with its log) and has its second digit counted; one above it is dropped.
            lds.plog[lane] = 0.0;
            lds.psel[lane] = make_uint2(pre, (uint32_t)need);
            const int b2 = min(SEL_BITS, nbits), shift2 = nbits - b2;
            if (any_trunc)
                for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
            __syncthreads();
            TSUB(0)
            int n_cmp = 0;
            float rmax2 = 0.f;  // largest |closest point - pixel|^2 over the records: bounds the gradient sums of pass 3
            if (vbase > 0) {
                struct Rec { uint32_t z, mt; float sd; };
                auto load_recs = [&](Rec (&r)[DGROUP], int g0) {
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const uint32_t idx = (uint32_t)min(g0 + u * WAVE + lane, vbase - 1);
                        const Rec3 q = ld_stream(srec, idx);
                        r[u].z = q.a; r[u].mt = q.b; r[u].sd = __uint_as_float(q.c);
                    }
                };
                auto blend_recs = [&](const Rec (&r)[DGROUP], int g0) {
                    uint2 ps[DGROUP];
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) ps[u] = lds.psel[r[u].mt & 63u];
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const bool valid = g0 + u * WAVE + lane < vbase;
                        const uint32_t key = r[u].z - kmin, d1 = key >> nbits;
                        const bool sure = valid & ((ps[u].y == 0u) | (d1 < ps[u].x));
                        const bool maybe = valid & (ps[u].y > 0u) & (d1 == ps[u].x);
                        rmax2 = fmaxf(rmax2, fabsf(r[u].sd));  // (the clamped tail repeats a record: harmless)
                        const float lf = __log2f(1.0f - face_prob(r[u].sd, a.inv_sigma_log2e));
                        if (sure & (lf != 0.f)) atomicAdd(&lds.plog[r[u].mt & 63u], (double)lf);
                        STAT(45, __popcll(__ballot(sure))) STAT(46, __popcll(__ballot(valid & !sure & !maybe)))
                        if (any_trunc) {  // wave-uniform
                            const unsigned long long km = __ballot(maybe);
                            const uint32_t slot = (uint32_t)n_cmp + __builtin_amdgcn_mbcnt_hi((uint32_t)(km >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)km, 0u));
                            if (maybe) {
                                at(crec, slot) = Rec3{key, r[u].mt, __float_as_uint(lf)};
                                const uint32_t bucket = (key >> shift2) & ((1u << b2) - 1u);
                                atomicAdd(&lds.hist[(bucket >> 1) * WAVE + (r[u].mt & 63u)], (bucket & 1u) ? 0x10000u : 1u);
                            }
                            n_cmp += __popcll(km);
                        }
                    }
                };
                Rec ra[DGROUP], rb[DGROUP];
                load_recs(ra, 0);
                for (int g0 = 0; g0 < vbase; g0 += 2 * DGROUP * WAVE) {
                    load_recs(rb, g0 + DGROUP * WAVE);
                    blend_recs(ra, g0);
                    load_recs(ra, g0 + 2 * DGROUP * WAVE);
                    blend_recs(rb, g0 + DGROUP * WAVE);
                }
            }
            __syncthreads();
            TSUB(4)
            if (any_trunc) {
                if (nbits > 0) {  // second digit: counted above
                    pick_digit(lds.hist, lane, b2, pre, need, n_eq);
                    nbits -= b2;
                    __syncthreads();
                }
                // refinement through memory while the compact stream is long (it shrinks about six-fold per sweep) ...
                while (nbits > 0 && __ballot(need > 0) != 0ull && n_cmp > SELR * WAVE) {
                    const int b = min(SEL_BITS, nbits);
                    n_cmp = refine_sweep(lds, crec, n_cmp, nbits, b, lane, pre, need);
                    pick_digit(lds.hist, lane, b, pre, need, n_eq);
                    nbits -= b;
                    __syncthreads();
                }
                if (n_cmp <= SELR * WAVE) {
                // ... then IN REGISTERS (round 4): a lane takes up to SELR of the remaining records and every further step - the digits
                // still to go, the cut of a tie group by face id, the sum of the logs that made it - runs on them with the per-pixel
                // histograms in LDS and no memory traffic at all.  Through memory each of those three to seven sweeps over a few
                // hundred records was two exposed round trips (the first load, the drain of the in-place stores): the selection was
                // 9.5 % of the launch for 23 % of the records.
                constexpr uint32_t INV = 0xFFFFFFFFu;  // record that has left the selection (keys are below 2^31)
                uint32_t rk[SELR], rm[SELR];
                float rl[SELR];
#pragma unroll
                for (int r_ = 0; r_ < SELR; ++r_) {
                    const int idx = r_ * WAVE + lane;
                    const Rec3 q = at(crec, (uint32_t)min(idx, max(n_cmp - 1, 0)));
                    rk[r_] = idx < n_cmp ? q.a : INV; rm[r_] = q.b; rl[r_] = __uint_as_float(q.c);
                }
                while (nbits > 0 && __ballot(need > 0) != 0ull) {
                    const int b = min(SEL_BITS, nbits), shift = nbits - b;
                    lds.psel[lane] = make_uint2(pre, (uint32_t)need);
                    for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
                    lds_fence();
                    uint2 ps[SELR];
#pragma unroll
                    for (int r_ = 0; r_ < SELR; ++r_) ps[r_] = lds.psel[rm[r_] & 63u];
#pragma unroll
                    for (int r_ = 0; r_ < SELR; ++r_) {
                        const uint32_t pxl = rm[r_] & 63u;
                        const bool live = (rk[r_] != INV) & (ps[r_].y > 0u);
                        const uint32_t top = rk[r_] >> nbits;
                        const bool sure = live & (top < ps[r_].x), stay = live & (top == ps[r_].x);
                        if (sure & (rl[r_] != 0.f)) atomicAdd(&lds.plog[pxl], (double)rl[r_]);
                        if (stay) {
                            const uint32_t bucket = (rk[r_] >> shift) & ((1u << b) - 1u);
                            atomicAdd(&lds.hist[(bucket >> 1) * WAVE + pxl], (bucket & 1u) ? 0x10000u : 1u);
                        }
                        rk[r_] = (live & !stay) ? INV : rk[r_];  // decided either way: it leaves
                    }
                    lds_fence();
                    pick_digit(lds.hist, lane, b, pre, need, n_eq);
                    nbits -= b;
                    lds_fence();
                }
                if (trunc) zt_bits = pre + kmin;
                // `need` of the n_eq faces at the threshold are kept: the ones with the smallest face ids - or, under tie_rule 1, the
                // ones the reference's queue would keep (k_raster_tie_replay).  Which `need` of them they are matters only if the tied
                // records differ: a pixel outside two faces that meet in an edge (or a fan that meets in a vertex) - the usual tie -
                // has the same closest point, depth, distance and end points on all of them, so every choice gives the same
                // silhouette value and the same vertex gradient.  Such a pixel is cut by face id here (a fifth of the cut tie groups: 237 000 ->
                // 184 000 replayed pixels per cfg2b launch); only tie groups whose records differ in distance or side go on - mostly fans
                // around a vertex, whose faces clip the pixel's barycentrics to that vertex (one depth) but are at different distances.
                bool same = false;
                if (HOOK_TIE_EQUIV && a.tie_rule && __ballot(trunc && need < n_eq) != 0ull) {  // (wave-uniform)
                    lds.psel[lane] = make_uint2((trunc && need < n_eq) ? pre : INV, 0u);
                    lds.hist[lane] = 0xFFFFFFFFu; lds.hist[WAVE + lane] = 0u;         // min / max of the tied records' log bits
                    lds.hist[2 * WAVE + lane] = 1u; lds.hist[3 * WAVE + lane] = 0u;   // and / or of their inside flags
                    lds_fence();
#pragma unroll
                    for (int r_ = 0; r_ < SELR; ++r_) {
                        const uint32_t pxl = rm[r_] & 63u;
                        if (rk[r_] != INV && rk[r_] == lds.psel[pxl].x) {
                            const uint32_t lb = __float_as_uint(rl[r_]), ins = (rm[r_] >> 22) & 1u;
                            atomicMin(&lds.hist[pxl], lb); atomicMax(&lds.hist[WAVE + pxl], lb);
                            atomicAnd(&lds.hist[2 * WAVE + pxl], ins); atomicOr(&lds.hist[3 * WAVE + pxl], ins);
                        }
                    }
                    lds_fence();
                    same = lds.hist[lane] == lds.hist[WAVE + lane] && lds.hist[2 * WAVE + lane] == lds.hist[3 * WAVE + lane];
                    lds_fence();
                }
                const bool split = trunc && need < n_eq && (!a.tie_rule || same);
                defer = trunc && need < n_eq && a.tie_rule && !same;
                uint32_t rf[SELR];  // face ids of the records at the threshold of a split pixel (fetched only in tiles that have one)
#pragma unroll
                for (int r_ = 0; r_ < SELR; ++r_) rf[r_] = INV;
                if (__ballot(split) != 0ull) {
                    lds.psel[lane] = make_uint2(split ? pre : INV, 0u);
                    lds_fence();
#pragma unroll
                    for (int r_ = 0; r_ < SELR; ++r_)
                        if (rk[r_] != INV && rk[r_] == lds.psel[rm[r_] & 63u].x) rf[r_] = lst[(rm[r_] >> 6) & 0xFFFFu];
                    int pbits = 32 - __clz(max(a.FT - 1, 1));
                    uint32_t ppre = 0u;
                    int pneed = split ? need : 0, peq = 0;
                    lds_fence();
                    while (pbits > 0 && __ballot(pneed > 0) != 0ull) {
                        const int b = min(SEL_BITS, pbits), shift = pbits - b;
                        lds.psel[lane] = make_uint2(ppre, (uint32_t)pneed);
                        for (int i_ = lane; i_ < (1 << SEL_BITS) / 2 * WAVE; i_ += WAVE) lds.hist[i_] = 0u;
                        lds_fence();
#pragma unroll
                        for (int r_ = 0; r_ < SELR; ++r_) {
                            const uint32_t pxl = rm[r_] & 63u;
                            const uint2 ps = lds.psel[pxl];
                            const bool hit = (rf[r_] != INV) & (ps.y > 0u) & ((rf[r_] >> pbits) == ps.x);
                            const uint32_t bucket = (rf[r_] >> shift) & ((1u << b) - 1u);
                            if (hit) atomicAdd(&lds.hist[(bucket >> 1) * WAVE + pxl], (bucket & 1u) ? 0x10000u : 1u);
                        }
                        lds_fence();
                        pick_digit(lds.hist, lane, b, ppre, pneed, peq);
                        pbits -= b;
                        lds_fence();
                    }
                    if (split) tie_cut = (int)ppre;
                }
                // the records still held that made it: depth below the threshold, or at it up to the tie cut
                lds.psel[lane] = make_uint2(trunc ? pre : 0u, (uint32_t)tie_cut);
                lds_fence();
#pragma unroll
                for (int r_ = 0; r_ < SELR; ++r_) {
                    const uint32_t pxl = rm[r_] & 63u;
                    const uint2 ps = lds.psel[pxl];
                    const bool keep = (rk[r_] != INV) & ((rk[r_] < ps.x) | ((rk[r_] == ps.x) & (((int)ps.y == 0x7FFFFFFF) | ((int)rf[r_] <= (int)ps.y))));
                    if (keep & (rl[r_] != 0.f)) atomicAdd(&lds.plog[pxl], (double)rl[r_]);
                }
                lds_fence();
                } else {
                // (the compact stream never got short - thousands of records tied in their first digits: everything through memory)
                while (nbits > 0 && __ballot(need > 0) != 0ull) {
                    const int b = min(SEL_BITS, nbits);
                    n_cmp = refine_sweep(lds, crec, n_cmp, nbits, b, lane, pre, need);
                    pick_digit(lds.hist, lane, b, pre, need, n_eq);
                    nbits -= b;
                    __syncthreads();
                }
                if (trunc) zt_bits = pre + kmin;
                // `need` of the n_eq faces at the threshold are kept: the first ones in list order
                const bool split = trunc && need < n_eq && !a.tie_rule;
                defer = trunc && need < n_eq && a.tie_rule;
                if (__ballot(split) != 0ull) {
                    // select on the list position among the records whose depth equals the pixel's threshold
                    lds.pgrad[lane] = make_float4(0.f, __uint_as_float(split ? pre : 0xFFFFFFFFu), 0.f, 0.f);
                    __syncthreads();
                    int pbits = 32 - __clz(max(a.FT - 1, 1));  // the tie key is the face id (the list is in near-to-far order)
                    uint32_t ppre = 0u;
                    int pneed = split ? need : 0, peq = 0;
                    auto pos_key = [&](uint32_t idx, uint32_t mt) {
                        // records of other depths get the key 0xFFFFFFFF, which select_sweep ignores
                        return at(crec, idx).a == __float_as_uint(lds.pgrad[mt & 63u].y) ? lst[(mt >> 6) & 0xFFFFu] : 0xFFFFFFFFu;
                    };
                    while (pbits > 0 && __ballot(pneed > 0) != 0ull) {
                        const int b = min(SEL_BITS, pbits);
                        select_sweep(lds, crec, n_cmp, pbits, b, lane, ppre, pneed, pos_key);
                        pick_digit(lds.hist, lane, b, ppre, pneed, peq);
                        pbits -= b;
                        __syncthreads();
                    }
                    if (split) tie_cut = (int)ppre;
                }
                // the compact records still in the stream (those of the last bucket examined) that made it: depth below the
                // threshold, or at it up to the tie cut
                lds.pgrad[lane] = make_float4(0.f, __uint_as_float(trunc ? pre : 0u), __int_as_float(tie_cut), 0.f);
                const bool any_split_sel = __ballot(tie_cut != 0x7FFFFFFF) != 0ull;
                __syncthreads();
                for (int g0 = 0; g0 < n_cmp; g0 += DGROUP * WAVE) {
                    uint32_t kk[DGROUP], mt[DGROUP];
                    float lf[DGROUP];
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const uint32_t idx = (uint32_t)min(g0 + u * WAVE + lane, n_cmp - 1);
                        const Rec3 q = at(crec, idx);
                        kk[u] = q.a; mt[u] = q.b; lf[u] = __uint_as_float(q.c);
                    }
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const float4 pg = lds.pgrad[mt[u] & 63u];
                        const uint32_t zt_ = __float_as_uint(pg.y);
                        // a record AT the threshold depth of a pixel whose tie group straddles K is kept up to the cut in face id
                        // (rare: the id is fetched only then)
                        const bool in_range = g0 + u * WAVE + lane < n_cmp;
                        bool tie_ok = true;
                        if (any_split_sel) {  // wave-uniform
                            const int cut = __float_as_int(pg.z);
                            int fid = 0;
                            if (in_range & (kk[u] == zt_) & (cut != 0x7FFFFFFF)) fid = (int)lst[(mt[u] >> 6) & 0xFFFFu];
                            tie_ok = fid <= cut;
                        }
                        const bool keep = in_range & ((kk[u] < zt_) | ((kk[u] == zt_) & tie_ok));
                        if (keep & (lf[u] != 0.f)) atomicAdd(&lds.plog[mt[u] & 63u], (double)lf[u]);
                    }
                }
                __syncthreads();
                }
            }
            TMARK(2)
            TSUB(5)
            HOOK_STOP_AFTER(3, { p_lo += span; continue; })
            STAT(22, n_cmp) STAT(23, __popcll(__ballot(trunc))) STAT(24, __popcll(__ballot(lds.plog[lane] != 0.0)))
            STAT(40, any_trunc ? 1 : 0) STAT(41, may_truncate ? 1 : 0) STAT(42, any_trunc ? vbase : 0) STAT(43, may_truncate ? vbase : 0) STAT(44, __popcll(__ballot(tie_cut != 0x7FFFFFFF)))
            const double plog_px = lds.plog[lane];
            const float alpha = exp2f((float)plog_px);
            TMARK(3)

            // ---------------- epilogue: silhouette value, loss, upstream gradient --------------------
            const float silv = 1.0f - alpha;
            const bool own = in_img && mine && !defer;
            tie_acc |= __ballot(in_img && mine && defer);
            float g = 0.f;
            if (MODE == MODE_FWD) {
                if (own) a.sil[pix] = silv;
            } else if (MODE == MODE_BWD) {
                if (own) g = a.grad_sil[pix];
            } else {
                float lsum = 0.f;
                if (own) {
                    const float tg = a.target_u8 ? (float)a.target_u8[pix] : a.target[pix];
                    const float diff = silv - tg;
                    lsum = fabsf(diff) - fabsf(tg);  // loss_img starts at sum |0 - target|
                    g = a.pix_scale[n] * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
                    if (a.sil) a.sil[pix] = silv;
                }
                lsum = wave_sum(lsum);
                if (lane == 0 && lsum != 0.f) atomicAdd(&a.loss_acc[n], (unsigned long long)(long long)rint((double)lsum * 4294967296.0));
            }

            // ---------------- pass 3: lane = record -------------------------------------------------
            // d sil / d dist_k = -alpha p_k / sigma   (alpha = prod_j (1 - p_j); exact also when 1 - p_k == 0)
            const float coef = -g * alpha * a.inv_sigma;
            // (a pixel whose records all have 1 - p == 1 in fp32 - or that has none - hands nothing back: its coefficient must not enter
            // the fixed-point bound below either, or a tile of empty pixels sets the resolution for its one contributing pixel)
            const bool active = own && (g != 0.f) && (alpha > ALPHA_GRAD_EPS) && (plog_px != 0.0);
            const bool any_split = __ballot(tie_cut != 0x7FFFFFFF) != 0ull;  // a pixel whose tie group at the K-th depth is cut by face id
            TSUB(6)
            STAT(25, __popcll(__ballot(active)))
            if (MODE != MODE_FWD && __ballot(active) != 0ull) {
                float *dn = a.d_ndc + (size_t)n * a.V * 2;
                // Fixed point for the LDS accumulators.  One accumulator component receives at most one record per pixel,
                // each of magnitude <= 2 |r| |coef_pixel| p_k max(t, 1 - t) <= 2 r_max |coef_pixel|, so no partial sum
                // exceeds bound = 2 r_max sum |coef_pixel|.  With scale = the power of two that maps `bound` into [2^29, 2^30)
                // every rounded contribution sum stays below 2^31 (plus at most 64 half-units of rounding), the scaling is
                // exact, and the resolution is bound / 2^30: ~1e-9 of the tile's largest possible gradient sum, below the
                // fp32 rounding of the global atomics the sums end in.  Integer sums are order independent.
                const float csum = wave_sum(active ? fabsf(coef) : 0.f);
                const float bound = 2.0f * sqrtf(wave_max(rmax2)) * csum;
                // Packed launches accumulate in the IMAGE's fixed-point scale right away (image_fx_scale: no vertex component of the
                // image can overflow it, so no partial sum can): every contribution is rounded once, per record, and from there on
                // all sums - LDS, flush, memory-side atomics - are integer adds, exact in any order and any grouping of faces.
                // (an image with cut faces stays on float atomics, see k_raster_setup)
                const bool img_fixed = MODE == MODE_FUSED && a.packed && a.clip.xcount[n] == 0u;  // (wave-uniform)
                const float fx_scale = img_fixed ? image_fx_scale(a.img_bound[n], a.pix_scale[n], a.inv_sigma)
                                       : (bound > 0.f && bound < 3.0e38f) ? exp2f(fminf(29.0f - floorf(log2f(bound)), 100.0f)) : 0.f;
                const float fx_inv = fx_scale > 0.f ? 1.0f / fx_scale : 0.f;
                lds.pgrad[lane] = make_float4(active ? coef * fx_scale : 0.f, __uint_as_float(zt_bits), __int_as_float(tie_cut), 0.f);
                __syncthreads();
                constexpr int GR = GCHUNK / DCHUNK;
                static_assert(GCHUNK == WAVE, "pass 3: lane = face of the group");
                const uint32_t copy_off = (uint32_t)(lane & (GCOPIES - 1)) * (GCHUNK * 3);
                float *const xg_n = a.clip.xg + (size_t)n * CLIP_VX * 2;  // gradient rows of the image's new vertices (cut faces)
                // the group's projected vertices, from which a record's edge parameter t is recomputed (4 bytes less written and
                // read per record than storing it); the table lives where the selection histograms were
                float2 *const fv = reinterpret_cast<float2 *>(lds.hist);  // [GCHUNK][3]
                static_assert(GCHUNK * 3 * sizeof(float2) <= sizeof(lds.hist), "the vertex table of a group lives in the histogram area");
                TP3_START
                for (int ch = 0; ch < chunks_done; ch += GR) {
                    const int i_beg = (int)chunk_start(ch), i_end = (int)chunk_start(min(ch + GR, chunks_done));  // (registers: no memory round trip)
                    if (i_beg == i_end) continue;
                    TP3(0)
                    // lane = face of the group: its projected vertices and vertex ids as pass 1 left them in the tile's table (one round
                    // trip, requested together with the first records; the list -> face -> vertex chain they replace was three)
                    const int fch = ch * DCHUNK + lane;
                    const bool staged = fch < chunks_done * DCHUNK && fch < list_total;
                    const uint32_t fcl = (uint32_t)min(fch, list_total - 1);
                    const float2 tv0 = at(sxy, fcl), tv1 = at(sxy, (uint32_t)a.list_stride + fcl), tv2 = at(sxy, 2u * (uint32_t)a.list_stride + fcl);
                    const TriIds tid = at(sid, fcl);
                    struct GRec { uint32_t z, mt; float sd; };
                    auto load_recs = [&](GRec (&r)[DGROUP], int g0) {
#pragma unroll
                        for (int u = 0; u < DGROUP; ++u) {
                            const uint32_t idx = (uint32_t)min(g0 + u * WAVE + lane, i_end - 1);  // clamped: the tail repeats the last record
                            const Rec3 q = ld_stream(srec, idx);
                            r[u].z = q.a; r[u].mt = q.b; r[u].sd = __uint_as_float(q.c);
                        }
                    };
                    GRec ra[DGROUP], rb[DGROUP];
                    load_recs(ra, i_beg);
                    fv[lane * 3 + 0] = tv0;
                    fv[lane * 3 + 1] = tv1;
                    fv[lane * 3 + 2] = tv2;
                    TP3(1)
                    for (int i_ = lane; i_ < GCOPIES * GCHUNK * 3; i_ += WAVE) (&lds.gacc[0][0])[i_] = 0ull;
                    lds_fence();
                    TP3(2)
                    // One row of records per lane and step, DGROUP rows per buffer.  Straight-line code: every lane computes its record's
                    // contribution whether it is kept or not and only the two accumulator adds are predicated, so that the LDS gathers of
                    // all rows of a buffer are in flight together (a branch per record kept each row's gathers behind the previous row's
                    // conflicting atomics: one exposed LDS round trip per row).
                    auto grad_recs = [&](const GRec (&r)[DGROUP], int g0) {
                        float4 pg[DGROUP];
                        float2 pa[DGROUP], pb[DGROUP], pc[DGROUP];
                        uint32_t oa[DGROUP], ob[DGROUP];
#pragma unroll
                        for (int u = 0; u < DGROUP; ++u) {
                            const uint32_t mt = r[u].mt;
                            const uint32_t f3 = ((mt >> 6) & (uint32_t)(GCHUNK - 1)) * 3u;  // (list position % GCHUNK) * 3
                            const uint32_t edge = mt >> 23;
                            oa[u] = f3 + (edge == 2u ? 1u : 0u); ob[u] = f3 + (edge == 0u ? 1u : 2u);  // end points of the closest edge
                            pg[u] = lds.pgrad[mt & 63u];
                            pa[u] = fv[oa[u]]; pb[u] = fv[ob[u]]; pc[u] = lds.pixt[mt & 63u];
                        }
#pragma unroll
                        for (int u = 0; u < DGROUP; ++u) {
                            const bool valid = g0 + u * WAVE + lane < i_end;
                            const uint32_t mt = r[u].mt;
                            const uint32_t zt_ = __float_as_uint(pg[u].y);
                            const bool inside = ((mt >> 22) & 1u) != 0u;
                            float gd = pg[u].x * face_prob(r[u].sd, a.inv_sigma_log2e);                 // scale * d L / d (signed dist)
                            gd = inside ? -gd : gd;                                               // ... / d (unsigned squared distance)
                            bool tie_ok = true;  // (as in the blend: only a record at the threshold of a split tie group needs its face id)
                            if (any_split) {  // wave-uniform and rare: the fetch and the wait for it stay out of the common path
                                const int cut = __float_as_int(pg[u].z);
                                int fid = 0;
                                if (valid & (r[u].z == zt_) & (cut != 0x7FFFFFFF)) fid = (int)lst[(mt >> 6) & 0xFFFFu];
                                tie_ok = fid <= cut;
                            }
                            const bool keep = valid & (gd != 0.f) & ((r[u].z < zt_) | ((r[u].z == zt_) & tie_ok));
                            // closest point of that edge: clamped projection of the pixel, as eval_pair computed it (t = 0 for a
                            // degenerate edge); r = closest point - pixel
                            const float exx = pb[u].x - pa[u].x, eyy = pb[u].y - pa[u].y;
                            const float l2 = exx * exx + eyy * eyy;
                            const float t = __builtin_amdgcn_fmed3f((exx * (pc[u].x - pa[u].x) + eyy * (pc[u].y - pa[u].y)) * (l2 <= K_EPS ? 0.f : __builtin_amdgcn_rcpf(l2)), 0.f, 1.f);
                            const float rx = fmaf(t, exx, pa[u].x - pc[u].x), ry = fmaf(t, eyy, pa[u].y - pc[u].y);
                            const float ex = 2.0f * rx * gd, ey = 2.0f * ry * gd;
                            const float bx = t * ex, by = t * ey;
                            // (x, y) -> x * 2^32 + y as 64-bit two's complement: a negative y borrows one from the high word
                            auto pack = [](float x, float y) {
                                const int qx = cvt_round(x), qy = cvt_round(y);
                                return ((unsigned long long)(uint32_t)(qx + (qy >> 31)) << 32) | (unsigned long long)(uint32_t)qy;
                            };
                            const unsigned long long ga = pack(ex - bx, ey - by), gb = pack(bx, by);
                            if (keep) {
                                unsigned long long *acc = &lds.gacc[0][0] + copy_off;
                                atomicAdd(acc + oa[u], ga);
                                atomicAdd(acc + ob[u], gb);
                            }
                        }
                    };
                    for (int g0 = i_beg; g0 < i_end; g0 += 2 * DGROUP * WAVE) {
                        load_recs(rb, g0 + DGROUP * WAVE);
                        grad_recs(ra, g0);
                        load_recs(ra, g0 + 2 * DGROUP * WAVE);
                        grad_recs(rb, g0 + DGROUP * WAVE);
                    }
                    TP3(3)
                    lds_fence();
                    TP3(4)
                    if (staged) {  // flush: sum the copies, unpack, one global atomic per touched vertex component
                        const int vi[3] = {tid.a, tid.b, tid.c};
#pragma unroll
                        for (int k = 0; k < 3; ++k) {
                            unsigned long long tot = 0ull;
#pragma unroll
                            for (int c = 0; c < GCOPIES; ++c) tot += lds.gacc[c][lane * 3 + k];
                            if (img_fixed) {  // wave-uniform: the sum is already in the image's scale
                                if (tot != 0ull) atomicAdd(reinterpret_cast<unsigned long long *>(dn) + vi[k], tot);
                                continue;
                            }
                            const int qy = (int)(uint32_t)tot;
                            const int qx = (int)(uint32_t)((tot - (unsigned long long)(long long)qy) >> 32);
                            float *const row = vi[k] < a.V ? dn + 2 * vi[k] : xg_n + 2 * (vi[k] - a.V);  // (a vertex of a cut face's front part: its own table)
                            if (qx != 0) atomicAdd(row, (float)qx * fx_inv);
                            if (qy != 0) atomicAdd(row + 1, (float)qy * fx_inv);
                        }
                    }
                    lds_fence();  // the accumulators are read before the next group clears them; unlike __syncthreads() this does
                                  // not wait for the flush's global atomics to be acknowledged (a microsecond per group)
                    TP3(5)
                }
            }
            __syncthreads();
            TMARK(4)
            TSUB(7)
            p_lo += span;
        }
        if (tie_acc != 0ull && lane == 0) {
            atomicOr(&a.tie_mask[(size_t)part * 2u * a.item_cap + item_at], tie_acc);  // (pieces of one tile add their bits)
            atomicAdd(&a.ctr->tie_pixels, (unsigned int)__popcll(tie_acc));
        }
        TUNIT_END
    }
    }  // next partition
    TIMERS_FLUSH
}

// ---------------------------------------------------------------------------------------------
// reference tie rule (SmilRasterSettings.tie_rule = SMIL_TIE_REFERENCE_QUEUE)
// ---------------------------------------------------------------------------------------------
// pytorch3d's naive rasteriser keeps a pixel's K nearest fragments in an UNSORTED array: faces are visited in index order, the
// first K candidates fill the array, and from then on a candidate nearer than the array's farthest entry replaces that entry,
// whose successor is found by a scan for the first slot holding the largest depth (strict comparisons throughout;
// RasterizeMeshesNaiveCudaKernel, selected by the reference with faces_per_pixel = 100 and bin_size = 0,
// smal_fitter/p3d_renderer.py:42-47).  The K nearest by DEPTH survive whatever the order; which members of a group of EQUAL depths
// at the K-th place survive depends on the slots the whole history put them in.  The tile kernel's rule - the smallest face ids -
// is order independent but not that one.  With tie_rule 1 the tile kernel leaves every pixel whose tie group is cut by K - and whose
// tied records are not interchangeable (see `same` there) - to this kernel (a bit per pixel and work item, `tie_mask`), which REPLAYS
// the reference's loop for that pixel: one wave per pixel, the queue in registers (slot s = lane s % 64, register s / 64; K <= 128),
// every face of the tile's list in index order (put in order through an LDS bitmap, 64 at a time, the same pair arithmetic as the tile
// kernel), the candidates fed to the queue - the fill in one step per batch, the replacements one by one.  Then the pixel's blend, loss
// term and gradient from the queue's final content, exactly as the tile kernel computes them from its records.  45 pixels per cfg2b
// image take this path (+20 % per iteration; what was measured and rebuilt on the way: profiles/r5_experiments.md section 7).
__device__ __forceinline__ FaceRows face_rows_from_tri(const Tri9 &tv, float cx, float cy) {
    FaceRows q;
    float rl12;
    face_rows_lo(tv, cx, cy, q.r0, q.r1, q.r2);
    face_rows_hi(tv, cx, cy, q.r3, q.r4, q.r5, rl12);
    q.r6 = make_float4(rl12, 0.f, 0.f, 0.f);
    return q;
}
// largest value of the wave (values > 0, or 0 for "none") in DPP: a running maximum along the lanes as wave_scan_add runs its sum
// (lanes that receive nothing read 0), read from the last lane - no LDS round trips in the replay's serial chain
__device__ __forceinline__ uint32_t wave_max_u32(uint32_t v) {
    int x = (int)v;
#define MAX_STEP(ctrl, rows) { x = (int)max((uint32_t)x, (uint32_t)__builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xF, false)); }
    MAX_STEP(0x111, 0xF) MAX_STEP(0x112, 0xF) MAX_STEP(0x114, 0xF) MAX_STEP(0x118, 0xF)
    MAX_STEP(0x142, 0xA) MAX_STEP(0x143, 0xC)
#undef MAX_STEP
    return (uint32_t)__builtin_amdgcn_readlane(x, WAVE - 1);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, WAVE);
    return v;
}

#ifndef TIE_WHOLE_TILE
#define TIE_WHOLE_TILE 6  // replayed pixels up to which a tile is one work unit (above: four, by quarters of its pixels)
#endif
#ifndef TIE_ORD_CAP
#define TIE_ORD_CAP 2048  // faces of a tile's list that the replay puts in order at once (longer lists: 64 bitmap words = up to 2 048 faces at a time)
#endif
static_assert(TIE_ORD_CAP % 64 == 0 && TIE_ORD_CAP / 32 <= WAVE && TIE_ORD_CAP >= 64 * 32, "the high bits of the ordered ids are cleared by one wave; a segment of 64 bitmap words fits");
#ifndef TIE_WAVES_PER_SIMD
#define TIE_WAVES_PER_SIMD 5  // replay waves a SIMD holds: 96 registers each, nothing spilled (6 -> 80 registers, 17 of them spilled)
#endif
template <int MODE>
__global__ void __launch_bounds__(64, TIE_WAVES_PER_SIMD) k_raster_tie_replay(RasterArgs a) {
    const int lane = threadIdx.x;
    const int K = a.K;  // <= SMIL_MAX_FACES_PER_PIXEL = 128: two queue slots per lane
    const int n_tiles = a.tiles_x * a.tiles_x;
    const int n_groups = a.FT / WAVE;
    // The reference visits the faces in INDEX order; the tile's binned list holds them in the order the setup kernel's atomics
    // handed out.  A bitmap over the face ids (FT bits, dynamic LDS) puts them in order: one LDS atomic per entry, then the set bits
    // of 64 words at a time, laid out by a prefix sum of their counts.  The ordered ids serve every replayed pixel of the tile.
    extern __shared__ uint32_t tie_lds[];
    // LDS (the fewer bytes the more waves a SIMD holds, and this kernel is one long dependent chain per wave): the bitmap (FT bits), the
    // filling queue (128 slots x 3 words), TIE_ORD_CAP ordered ids of 16 bits + one high bit each (FT < 2^17)
    uint32_t *const bm = tie_lds;
    // the queue while it fills (slot = arrival rank: written by all lanes at once), moved to registers when it is full
    uint32_t *const fz = tie_lds + a.FT / 32, *const fm = fz + 2 * WAVE;
    float *const fs = reinterpret_cast<float *>(fm + 2 * WAVE);
    uint16_t *const ord = reinterpret_cast<uint16_t *>(fm + 4 * WAVE);
    uint32_t *const ord_hi = fm + 4 * WAVE + TIE_ORD_CAP / 2;  // bit i: id i >= 65536
    const int bm_words = a.FT / 32;
    unsigned int total = 0;
    for (int q = 0; q < N_PARTS; ++q)
        for (int c = 0; c < N_CLASSES; ++c) total += a.ctr->n_class[q][c];
    // Work unit = (work item, part of its 64 pixels - a quarter in large launches), unit index = part x items + item: the replayed pixels cluster in few tiles
    // (0.8 per touched tile on average, dozens in some), and a pixel costs ~7 000 wave instructions, so the units are dealt out
    // dynamically - a ticket is 64 units, lane = unit reads its item's mask - and the quarters of one heavy tile go to
    // different waves.
    unsigned int pre[N_PARTS + 1];  // items before each partition
    pre[0] = 0u;
    for (int q = 0; q < N_PARTS; ++q) pre[q + 1] = pre[q] + a.ctr->n_class[q][0] + a.ctr->n_class[q][1] + a.ctr->n_class[q][2] + a.ctr->n_class[q][3];
    // (a pixel is ~100 us of one wave: a launch with few replayed pixels per wave - a few hundred frames - is as long as its largest
    // unit, so its tiles are cut finer: 16 units of 4 pixels below 16 pixels per wave, single pixels below 4)
    // (no more waves than replayed pixels take tickets: 6 144 waves queueing for ONE counter word were 0.2 ms of a one-frame launch)
    if (blockIdx.x >= a.ctr->tie_pixels) return;  // (workgroup-uniform, before any barrier)
    const unsigned int px_per_wave = a.ctr->tie_pixels / gridDim.x;
    unsigned int parts_log = px_per_wave >= 16u ? 2u : (px_per_wave >= 4u ? 4u : 6u);
    // (items x parts must stay a 32-bit count: a launch of 2^26 touched tiles - 16 384 images @512^2, the largest slice the host cuts -
    // cut into single pixels would wrap; such a launch has millions of replayed pixels, and coarser parts balance just as well)
    while (((unsigned long long)total << parts_log) > 0xFFFFFFC0ull) --parts_log;  // (total < 2^31: ends at parts_log >= 1)
    const unsigned int px_log = 6u - parts_log;
    const unsigned int n_units = total << parts_log, n_tickets = (n_units + (unsigned int)WAVE - 1u) / (unsigned int)WAVE;
    TIE_TIMERS_INIT
    for (;;) {
        TIE_T(5)
        unsigned int ticket = 0u;
        if (lane == 0) ticket = atomicAdd(&a.ctr->tie_next, 1u);
        ticket = (unsigned int)__builtin_amdgcn_readfirstlane((int)ticket);
        if (ticket >= n_tickets) break;  // (every wave ends here: the counter only grows)
        TIE_T(4)
        // (a ticket's 64 units lie n_tickets apart: the items are sorted by cost class, and 64 neighbours of the heaviest class in one
        // ticket would be a tail of their own)
        const unsigned int u = (unsigned int)lane * n_tickets + ticket;
        unsigned long long my_mask = 0ull;
        uint32_t my_slot = 0u;
        if (u < n_units) {
            const unsigned int quarter = u / total, idx = u - quarter * total;
            int part = 0;
#pragma unroll
            for (int q = 1; q < N_PARTS; ++q) part += idx >= pre[q] ? 1 : 0;
            const unsigned int item = idx - pre[part];
            const unsigned int nc0 = a.ctr->n_class[part][0], nc1 = a.ctr->n_class[part][1], nc2 = a.ctr->n_class[part][2];
            const uint32_t item_at = item < nc0 ? item
                                   : item < nc0 + nc1 ? a.item_cap - 1u - (item - nc0)
                                   : item < nc0 + nc1 + nc2 ? a.item_cap + (item - nc0 - nc1)
                                   : 2u * a.item_cap - 1u - (item - nc0 - nc1 - nc2);
            my_slot = (uint32_t)part * 2u * a.item_cap + item_at;
            // (a tile with few replayed pixels is one unit - quarter 0 takes them all - so that its face list is put in order once)
            const unsigned long long full = a.tie_mask[my_slot];
            const unsigned long long part_mask = ((px_log == 6u ? 0ull : 1ull << (1u << px_log)) - 1ull) << (quarter << px_log);
            my_mask = (parts_log == 2u && __popcll(full) <= TIE_WHOLE_TILE) ? (quarter == 0u ? full : 0ull) : full & part_mask;
        }
        unsigned long long um = __ballot(my_mask != 0ull);
        TIE_T(4)
        while (um) {
        const int ul = (int)__builtin_ctzll(um);
        um &= um - 1ull;
        const size_t slot_i = (size_t)(uint32_t)__builtin_amdgcn_readlane((int)my_slot, ul);
        unsigned long long mask = (unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)my_mask, ul) |
                                  ((unsigned long long)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(my_mask >> 32), ul) << 32);
        const uint4 it = a.items[slot_i];
        const uint32_t code = it.x;
        const int n = (int)(code / (uint32_t)n_tiles), tile = (int)(code % (uint32_t)n_tiles);
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const float cx = pix_to_ndc(a.S - 1 - (tx * TILE + 4), a.S), cy = pix_to_ndc(a.S - 1 - (ty * TILE + 4), a.S);
        const float *vn = a.verts_ndc + (size_t)n * a.V * 3;
        const float *const xv_n = a.clip.xv + (size_t)n * CLIP_VX * 3;
        const int *const xf_n = a.clip.xf + (size_t)n * CLIP_FX * 3;
        const uint32_t *__restrict__ tbox_n = a.tbox + (size_t)n * a.FT;
        const uint32_t *__restrict__ gbox_n = a.gbox + (size_t)n * n_groups;
        const float2 *__restrict__ fzr_n = a.fzr + (size_t)n * a.FT;
        // the tile's faces as a bitmap over the face ids: from its binned list, or - images whose lists did not fit - from the
        // faces' tile boxes
        __syncthreads();  // (the previous unit's readers of `bm` / `ord` are done)
        for (int w = lane; w < bm_words; w += WAVE) bm[w] = 0u;
        __syncthreads();
        if (it.z != 0xFFFFFFFFu) {  // (wave-uniform)
            const uint2 *const ls = a.lists + (size_t)n * a.list_cap + it.y;
            for (int i = lane; i < (int)it.z; i += WAVE) {
                const uint32_t f = ls[i].x;
                atomicOr(&bm[f >> 5], 1u << (f & 31u));
            }
        } else {
            for (int g = 0; g < n_groups; ++g)
                if (box_has(gbox_n[g], tx, ty)) {  // (wave-uniform: the group's box union; lane = face)
                    const unsigned long long hit = __ballot(box_has(tbox_n[g * WAVE + lane], tx, ty));
                    if (lane == 0) { bm[2 * g] = (uint32_t)hit; bm[2 * g + 1] = (uint32_t)(hit >> 32); }
                }
        }
        __syncthreads();
        // set bits of the words [w0, w1) -> ordered ids ord[0 ...): a prefix sum of the words' counts lays them out
        auto extract = [&](int w0, int w1) -> int {
            int cnt = 0;
            if (lane < TIE_ORD_CAP / 32) ord_hi[lane] = 0u;
            __syncthreads();
            for (int wb = w0; wb < w1; wb += WAVE) {
                uint32_t wv = wb + lane < w1 ? bm[wb + lane] : 0u;
                const int c = __popc(wv), inc = wave_scan_add(c);
                int o = cnt + inc - c;
                while (wv) {
                    const uint32_t id = (uint32_t)((wb + lane) * 32 + (__ffs((int)wv) - 1));
                    ord[o] = (uint16_t)id;
                    if (id >> 16) atomicOr(&ord_hi[o >> 5], 1u << (o & 31));
                    ++o;
                    wv &= wv - 1u;
                }
                cnt += __builtin_amdgcn_readlane(inc, WAVE - 1);
            }
            __syncthreads();
            return cnt;
        };
        // a list of up to TIE_ORD_CAP faces is put in order once for all the unit's pixels; a longer one 64 words (<= 2 048 faces) at a
        // time, again for every pixel (the heaviest tiles of the mouse hold more)
        int n_faces = 0;
        for (int w = lane; w < bm_words; w += WAVE) n_faces += __popc(bm[w]);
        n_faces = __builtin_amdgcn_readlane(wave_scan_add(n_faces), WAVE - 1);
        const bool whole = n_faces <= TIE_ORD_CAP;  // (wave-uniform)
        const int n_ord = whole ? extract(0, bm_words) : 0;
        TIE_T(0)
        while (mask) {
            const int p = (int)__builtin_ctzll(mask);
            mask &= mask - 1ull;
            const int xo = tx * TILE + (p & 7), yo = ty * TILE + (p >> 3);
            const float dxp = pix_to_ndc(a.S - 1 - xo, a.S) - cx, dyp = pix_to_ndc(a.S - 1 - yo, a.S) - cy;
            // (the pair arithmetic is evaluated for the pixel AND its horizontal neighbour, as the tile kernel's lanes do: with the
            // same inputs in both halves of the packed operations the compiler folds them into a different, scalar sequence whose
            // last bits differ - and an exact depth tie of the tile kernel's arithmetic is then no tie here)
            const bool odd = (p & 1) != 0;
            const float dx_even = pix_to_ndc(a.S - 1 - (xo & ~1), a.S) - cx, dx_odd = pix_to_ndc(a.S - 1 - (xo | 1), a.S) - cx;
            // ---- the queue: depth bits, {face | inside << 22 | closest edge << 23}, signed squared distance ----
            uint32_t qz0 = 0u, qz1 = 0u, qm0 = 0u, qm1 = 0u;
            float qs0 = 0.f, qs1 = 0.f;
            int qsize = 0, qmax_idx = 0;
            uint32_t qmax_z = 0u;  // (depths are positive: their bit patterns order like the values, and 0 is below all of them)
            // lane = face: is it a candidate of this pixel, and with what depth / flags / distance
            auto eval_tri = [&](const Tri9 &tv, bool &cand, uint32_t &zb, uint32_t &fl, float &sd) {
                const FaceRows fr = face_rows_from_tri(tv, cx, cy);
                PairEval2 e;
                eval_pair2(fr, dx_even, dx_odd, dyp, a.blur, e);
                cand = odd ? e.cand1 : e.cand0;
                const f32x2 z2 = pair_depth2(fr, e);
                zb = __float_as_uint(vmax_raw(odd ? z2.y : z2.x, fminf(fminf(fr.r2.y, fr.r2.z), fr.r2.w)));
                sd = odd ? e.sd.y : e.sd.x;
                fl = ((odd ? e.inside1 : e.inside0) ? 1u << 22 : 0u) | (odd ? e.ebits1 : e.ebits0);
            };
            // The candidates of up to 64 faces (lanes in ascending face order) through the reference's queue.  While it FILLS, a
            // candidate's slot is its arrival rank: all lanes store at once (LDS), and the farthest entry is looked for once, when the
            // queue is full - the first slot holding the largest depth, which is what the reference's running `>` leaves.  From then on
            // a candidate enters only if it is strictly nearer than the queue's farthest entry, and that bound only ever falls: the
            // ones at or beyond it are dropped by one compare for all lanes, the others go in one by one.
            auto farthest = [&]() {
                const uint32_t v0 = lane < min(K, WAVE) ? qz0 : 0u, v1 = lane + WAVE < K ? qz1 : 0u;
                const uint32_t mx = wave_max_u32(max(v0, v1));
                const unsigned long long b0 = __ballot(v0 == mx);
                qmax_z = mx;
                qmax_idx = b0 ? (int)__builtin_ctzll(b0) : WAVE + (int)__builtin_ctzll(__ballot(v1 == mx));
            };
            auto load_queue = [&]() {  // LDS -> registers (slot s = lane s % 64, register s / 64)
                __syncthreads();
                qz0 = lane < qsize ? fz[lane] : 0u; qm0 = lane < qsize ? fm[lane] : 0u; qs0 = lane < qsize ? fs[lane] : 0.f;
                qz1 = lane + WAVE < qsize ? fz[lane + WAVE] : 0u; qm1 = lane + WAVE < qsize ? fm[lane + WAVE] : 0u; qs1 = lane + WAVE < qsize ? fs[lane + WAVE] : 0.f;
                __syncthreads();  // (the next pixel's fill may overwrite the arrays)
            };
            auto feed = [&](bool cand, uint32_t zb, uint32_t fl, float sd, int f) {
                const uint32_t m_lane = fl | (uint32_t)f;
                if (qsize < K) {  // (wave-uniform) still filling
                    const unsigned long long cf = __ballot(cand);
                    const int rank = qsize + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cf >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cf, 0u));
                    if (cand && rank < K) { fz[rank] = zb; fm[rank] = m_lane; fs[rank] = sd; }
                    const int nc = (int)__popcll(cf);
                    if (qsize + nc < K) { qsize += nc; return; }
                    qsize = K;  // full inside this batch: the candidates of rank >= K go on below
                    load_queue();
                    farthest();
                    cand = cand && rank >= K;
                }
                unsigned long long cm = __ballot(cand && zb < qmax_z);
                while (cm) {
                    const int l = (int)__builtin_ctzll(cm);
                    cm &= cm - 1ull;
                    const uint32_t z = (uint32_t)__builtin_amdgcn_readlane((int)zb, l);
                    if (!(z < qmax_z)) continue;  // (wave-uniform; the bound fell since the compare above)
                    const uint32_t m = (uint32_t)__builtin_amdgcn_readlane((int)m_lane, l);
                    const float s = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(sd), l));
                    const int put = qmax_idx;
                    const bool w0 = lane == put, w1 = lane + WAVE == put;  // (straight-line: the loop is one dependent chain, a taken branch costs more than six selects)
                    qz0 = w0 ? z : qz0; qm0 = w0 ? m : qm0; qs0 = w0 ? s : qs0;
                    qz1 = w1 ? z : qz1; qm1 = w1 ? m : qm1; qs1 = w1 ? s : qs1;
                    // the farthest entry was replaced: the new farthest is the first slot holding the largest depth, the replaced
                    // slot itself when nothing is strictly farther than the newcomer
                    const uint32_t v0 = lane < min(K, WAVE) ? qz0 : 0u, v1 = lane + WAVE < K ? qz1 : 0u;
                    const uint32_t mx = wave_max_u32(max(v0, v1));
                    const unsigned long long b0 = __ballot(v0 == mx), b1 = __ballot(v1 == mx);
                    const int first = b0 ? (int)__builtin_ctzll(b0) : WAVE + (int)__builtin_ctzll(b1 | (1ull << 63));
                    qmax_idx = mx > z ? first : put;
                    qmax_z = mx;  // (>= z: the newcomer is in the queue)
                }
            };
            // the faces ord[0 ... n_ids), 64 at a time.  A batch is a chain ordered id -> vertex ids -> vertices -> rows; the ids (and the
            // nearest depth) of the NEXT batch are requested before this one's vertices, so that a batch exposes one round trip, not
            // two: a pixel is one wave's serial work, and in small launches the replay is as long as its slowest pixel.
            auto walk = [&](int n_ids) {
                auto request = [&](int b0, bool &have, int &f, int &i0, int &i1, int &i2, uint32_t &znear) {
                    have = b0 + lane < n_ids;
                    f = have ? (int)ord[b0 + lane] | (int)(((ord_hi[(b0 + lane) >> 5] >> ((b0 + lane) & 31)) & 1u) << 16) : 0;
                    i0 = i1 = i2 = 0; znear = 0xFFFFFFFFu;
                    if (have) {
                        i0 = face_vertex(a.faces, xf_n, a.F, f, 0); i1 = face_vertex(a.faces, xf_n, a.F, f, 1); i2 = face_vertex(a.faces, xf_n, a.F, f, 2);
                        znear = __float_as_uint(fzr_n[f].x);
                    }
                };
                bool have_n; int f_n, j0, j1, j2; uint32_t zn_n;
                request(0, have_n, f_n, j0, j1, j2, zn_n);
                for (int b0 = 0; b0 < n_ids; b0 += WAVE) {
                    const bool have = have_n;
                    const int f = f_n, i0 = j0, i1 = j1, i2 = j2;
                    const uint32_t znear = zn_n;
                    // Once the queue is full only a depth strictly below its farthest entry gets in, and a face's depth at any pixel
                    // is at least its nearest vertex's: faces at or beyond the bound are not evaluated (no vertex fetch, no rows), a
                    // batch of them is skipped whole - in the heaviest tiles (thousands of faces behind the first hundred) most are.
                    const bool live = have && (qsize < K || znear < qmax_z);
                    bool cand = false;
                    uint32_t zb = 0u, fl = 0u;
                    float sd = 0.f;
                    Tri9 tv;
                    const bool any_live = __ballot(live) != 0ull;  // (wave-uniform)
                    if (live) tv = load_tri(a, vn, xv_n, i0, i1, i2);           // this batch's vertices are requested ...
                    if (b0 + WAVE < n_ids) request(b0 + WAVE, have_n, f_n, j0, j1, j2, zn_n);  // ... then the next batch's ids
                    if (!any_live) continue;
                    if (live) eval_tri(tv, cand, zb, fl, sd);
                    TIE_T(1)
                    feed(cand, zb, fl, sd, f);
                    TIE_T(2)
                }
            };
            if (whole) {
                walk(n_ord);
            } else {
                for (int w0 = 0; w0 < bm_words; w0 += WAVE) {
                    const int cnt = extract(w0, min(w0 + WAVE, bm_words));
                    TIE_T(0)
                    walk(cnt);
                    __syncthreads();  // (`ord` is rewritten by the next segment)
                }
            }
            if (qsize < K) load_queue();  // (fewer candidates than K: cannot happen for a pixel the tile kernel deferred, handled all the same)
            // ---- blend, loss term, upstream gradient: as the tile kernel's epilogue, for this one pixel ----
            const bool ok0 = lane < qsize, ok1 = lane + WAVE < qsize;
            const float lf0 = ok0 ? __log2f(1.0f - face_prob(qs0, a.inv_sigma_log2e)) : 0.f;
            const float lf1 = ok1 ? __log2f(1.0f - face_prob(qs1, a.inv_sigma_log2e)) : 0.f;
            const double plog_px = wave_sum_f64((double)lf0 + (double)lf1);
            const float alpha = exp2f((float)plog_px);
            const float silv = 1.0f - alpha;
            const size_t pix = ((size_t)n * a.S + yo) * a.S + xo;
            float g = 0.f;
            if (MODE == MODE_FWD) {
                if (lane == 0) a.sil[pix] = silv;
            } else if (MODE == MODE_BWD) {
                g = a.grad_sil[pix];
            } else {
                const float tg = a.target_u8 ? (float)a.target_u8[pix] : a.target[pix];
                const float diff = silv - tg;
                const float lsum = fabsf(diff) - fabsf(tg);  // loss_img starts at sum |0 - target|
                g = a.pix_scale[n] * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
                if (lane == 0) {
                    if (a.sil) a.sil[pix] = silv;
                    if (lsum != 0.f) atomicAdd(&a.loss_acc[n], (unsigned long long)(long long)rint((double)lsum * 4294967296.0));
                }
            }
            TIE_T(6)
            if (MODE == MODE_FWD || !((g != 0.f) && (alpha > ALPHA_GRAD_EPS) && (plog_px != 0.0))) continue;  // (wave-uniform)
            // ---- gradient of every kept entry, straight to the vertices (no accumulators: at most K entries) ----
            const float coef = -g * alpha * a.inv_sigma;
            float *dn = a.d_ndc + (size_t)n * a.V * 2;
            float *const xg_n = a.clip.xg + (size_t)n * CLIP_VX * 2;
            const bool img_fixed = MODE == MODE_FUSED && a.packed && a.clip.xcount[n] == 0u;
            const float fx_scale = img_fixed ? image_fx_scale(a.img_bound[n], a.pix_scale[n], a.inv_sigma) : 1.0f;
            const float px = dxp + cx, py = dyp + cy;
            auto entry_grad = [&](bool ok, uint32_t m, float sdv) {
                if (!ok) return;
                const int f = (int)(m & 0x3FFFFFu);
                const uint32_t edge = (m >> 23) & 3u;
                const bool inside = ((m >> 22) & 1u) != 0u;
                const int ia = face_vertex(a.faces, xf_n, a.F, f, edge == 2u ? 1 : 0), ib = face_vertex(a.faces, xf_n, a.F, f, edge == 0u ? 1 : 2);
                const float *pa_ = vertex_ptr(vn, xv_n, a.V, ia), *pb_ = vertex_ptr(vn, xv_n, a.V, ib);
                const float pax = pa_[0], pay = pa_[1], pbx = pb_[0], pby = pb_[1];
                float gd = coef * fx_scale * face_prob(sdv, a.inv_sigma_log2e);
                gd = inside ? -gd : gd;
                if (gd == 0.f) return;
                const float exx = pbx - pax, eyy = pby - pay;
                const float l2 = exx * exx + eyy * eyy;
                const float t = __builtin_amdgcn_fmed3f((exx * (px - pax) + eyy * (py - pay)) * (l2 <= K_EPS ? 0.f : __builtin_amdgcn_rcpf(l2)), 0.f, 1.f);
                const float rx = fmaf(t, exx, pax - px), ry = fmaf(t, eyy, pay - py);
                const float ex = 2.0f * rx * gd, ey = 2.0f * ry * gd;
                const float bx = t * ex, by = t * ey;
                if (img_fixed) {  // the image's packed fixed-point rows: one 64-bit integer add per end point
                    auto pack = [](float x, float y) {
                        const int qx = cvt_round(x), qy = cvt_round(y);
                        return ((unsigned long long)(uint32_t)(qx + (qy >> 31)) << 32) | (unsigned long long)(uint32_t)qy;
                    };
                    const unsigned long long ga = pack(ex - bx, ey - by), gb = pack(bx, by);
                    if (ga) atomicAdd(reinterpret_cast<unsigned long long *>(dn) + ia, ga);
                    if (gb) atomicAdd(reinterpret_cast<unsigned long long *>(dn) + ib, gb);
                } else {
                    float *const ra = ia < a.V ? dn + 2 * ia : xg_n + 2 * (ia - a.V), *const rb = ib < a.V ? dn + 2 * ib : xg_n + 2 * (ib - a.V);
                    atomicAdd(ra, ex - bx); atomicAdd(ra + 1, ey - by);
                    atomicAdd(rb, bx); atomicAdd(rb + 1, by);
                }
            };
            HOOK_TIE_GRADIENT(entry_grad(ok0, qm0, qs0); entry_grad(ok1, qm1, qs1);)
            TIE_T(3)
        }
        }  // next unit of the ticket
    }
    TIE_TIMERS_FLUSH
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

// Compute units of the current device (256 on MI355X), asked once: the persistent grid is sized to fill them.
static int device_cus() {
    static int cus = 0;
    if (cus == 0) {
        int dev = 0, n = 0;
        if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0)
            n = 256;
        cus = n;
    }
    return cus;
}

// resident workgroup slots of the device: what the tile kernel's dealing policy (pieces per tile) is tuned against
static long long tile_slots() {
    long long resident = (long long)device_cus() * RESIDENT_PER_CU;
    HOOK_RESIDENT(resident)
    return resident;
}
static int tile_grid(int N, int tiles_x) {
    // The grid - and with it the scratch arena, 1.6 MB per workgroup - is the largest number of workgroups the policy can put to
    // work on at most N x tiles touched tiles (the policy is piecewise monotone in the number of touched tiles: its maximum lies at
    // the upper end of one of its four ranges), so a small launch neither starts nor pays scratch for workgroups that would leave
    // at once: one 256^2 image gets 1 792 workgroups, one 128^2 image 896, not 4 096.
    const long long resident = tile_slots();
    const unsigned long long n_max = (unsigned long long)N * tiles_x * tiles_x;
    const unsigned int slots = (unsigned int)resident;
    unsigned int best = slots / 8u;
    const unsigned long long ends[4] = {slots * 5ull / 64ull, slots * 5ull / 16ull, slots * 5ull / 8ull, n_max};
    for (int k = 0; k < 4; ++k) {
        const unsigned long long n = ends[k] < n_max ? ends[k] : n_max;
        if (n == 0ull || n > 0x7FFFFFFull) { if (n) best = slots; continue; }
        const unsigned int w = deal_working((unsigned int)n, deal_split_log((unsigned int)n, slots), slots);
        best = w > best ? w : best;
    }
    return (int)(best < slots ? best : slots);
}

// per resident workgroup: F x {face id, nearest depth} in id order (tiles of images that are not binned), F face ids in walking order,
// F x {projected vertices (24 B), vertex ids (12 B)} by list position, F / DCHUNK + 2 chunk starts,
// (REC_CAP + REC_PAD) x (one 12-byte record + one 12-byte compact record)
#define N_STREAMS 6
static inline size_t scratch_bytes(int grid, int F) {
    return (size_t)grid * (12 * align256((size_t)F * sizeof(uint32_t)) + align256((size_t)(F / DCHUNK + 2) * sizeof(uint32_t)) +
                           (size_t)(REC_CAP + REC_PAD) * N_STREAMS * sizeof(uint32_t));
}

// binned tile lists: LIST_CAP_PER_FACE entries per face and image (8 bytes each) + one depth range per tile; images with more than
// COUNT_TILES_MAX tiles are never binned
static inline uint32_t list_cap_of(const SmilModel *m, int S) {
    return ceil_div(S, TILE) * ceil_div(S, TILE) <= COUNT_TILES_MAX ? (uint32_t)LIST_CAP_PER_FACE * (S > 256 ? 2u : 1u) * (uint32_t)m->F : 0u;
}

static inline int face_rows(const SmilModel *m) { return faces_padded(m->F) + CLIP_FX; }
// per-image clip tables: new vertices (12 B) + their gradient rows (8 B) + end points (8 B) + coefficients (8 B), front-part faces (12 B), count
static inline size_t clip_bytes(int N) {
    return align256((size_t)N * CLIP_VX * 12) + 3 * align256((size_t)N * CLIP_VX * 8) + align256((size_t)N * CLIP_FX * 12) + align256((size_t)N * 4);
}

extern "C" size_t smil_raster_workspace_bytes(const SmilModel *m, int32_t N, int32_t S) {
    if (!m || N <= 0 || S <= 0) return 0;
    const size_t tiles = (size_t)ceil_div(S, TILE) * ceil_div(S, TILE);
    const size_t FT = (size_t)face_rows(m);
    // tile boxes (N,FT), counters, work lists (2, N, tiles), per-face depth ranges (N,FT), binned lists, clip tables, per-workgroup scratch
    return align256((size_t)N * FT * sizeof(uint32_t)) + align256(sizeof(RasterCounters)) +
           align256((size_t)2 * N_PARTS * ceil_div(N, N_PARTS) * tiles * sizeof(uint4)) +
           align256((size_t)2 * N_PARTS * ceil_div(N, N_PARTS) * tiles * sizeof(unsigned long long)) +  // tie masks (tie_rule 1)
           align256((size_t)N * FT * sizeof(float2)) + align256((size_t)N * (FT / WAVE) * sizeof(uint32_t)) +
           align256((size_t)N * sizeof(float)) + align256((size_t)N * sizeof(unsigned long long)) + align256((size_t)N * list_cap_of(m, S) * sizeof(uint2)) +
           clip_bytes(N) + 256 +
           scratch_bytes(tile_grid(N, ceil_div(S, TILE)), face_rows(m));
}

// Counters of the most recent rasteriser call that used `workspace` with this N (device -> host copy: synchronises the stream).
extern "C" int smil_raster_stats(const SmilModel *m, int32_t N, const void *workspace, void *stream_, uint32_t *out4) {
    SMIL_REQUIRE(m && workspace && out4 && N > 0, "smil_raster_stats: bad argument");
    const RasterCounters *ctr = (const RasterCounters *)((const char *)workspace + align256((size_t)N * face_rows(m) * sizeof(uint32_t)));
    RasterCounters h;
    SMIL_HIP(hipMemcpyAsync(&h, ctr, sizeof(h), hipMemcpyDeviceToHost, (hipStream_t)stream_));
    SMIL_HIP(hipStreamSynchronize((hipStream_t)stream_));
    unsigned int tiles = 0;
    for (int q = 0; q < N_PARTS; ++q)
        for (int c = 0; c < N_CLASSES; ++c) tiles += h.n_class[q][c];
    out4[0] = h.straddling; out4[1] = tiles; out4[2] = h.unclipped; out4[3] = h.tie_pixels;
    return SMIL_OK;
}

static int raster_common(const SmilModel *m, const float *verts_ndc, int N, int S, const SmilRasterSettings *rs,
                         void *workspace, hipStream_t stream, RasterArgs &a, float *d_ndc_zero = nullptr,
                         const float *loss_src = nullptr, float *loss_dst = nullptr, float *dndc_scale = nullptr,
                         const float *pix_scale = nullptr, int packed = 0, bool grads = false) {
    SMIL_REQUIRE(m && verts_ndc && rs && workspace, "raster: null argument");
    SMIL_REQUIRE(N > 0 && S > 0 && S <= TILE * 256, "raster: bad sizes N=%d S=%d", N, S);
    SMIL_REQUIRE(rs->faces_per_pixel > 0 && rs->faces_per_pixel <= SMIL_MAX_FACES_PER_PIXEL,
                 "raster: faces_per_pixel=%d outside 1..%d", rs->faces_per_pixel, SMIL_MAX_FACES_PER_PIXEL);
    SMIL_REQUIRE(rs->sigma > 0.f && rs->blur_radius >= 0.f, "raster: bad blend settings");
    SMIL_REQUIRE(rs->tie_rule == SMIL_TIE_DEPTH_FACE_ID || rs->tie_rule == SMIL_TIE_REFERENCE_QUEUE, "raster: tie_rule=%d is neither 0 nor 1", rs->tie_rule);
    SMIL_REQUIRE(face_rows(m) <= REC_CAP, "raster: %d faces exceed the %d a single pixel's records may hold", m->F, REC_CAP - CLIP_FX - WAVE);
    const int tiles_x = ceil_div(S, TILE);
    SMIL_REQUIRE((double)N * tiles_x * tiles_x < 2147483647.0, "raster: N * tiles exceeds the work-item index range (2^31); launch in slices");
    static_assert(REC_CAP + REC_PAD < (1 << 24), "at<12-byte>() multiplies 24-bit indices");
    SMIL_REQUIRE(face_rows(m) + 64 < (1 << 24) && (double)list_cap_of(m, S) * sizeof(uint2) < 4294967296.0,
                 "raster: per-workgroup / per-image tables exceed the 24-bit index / 32-bit byte-offset range of at()");
    char *ws = (char *)workspace;
    const int FT = face_rows(m);
    uint32_t *tbox = (uint32_t *)ws;
    ws += align256((size_t)N * FT * sizeof(uint32_t));
    RasterCounters *ctr = (RasterCounters *)ws;  // (the probe tool reads the counters right behind the tile boxes)
    ws += align256(sizeof(RasterCounters));
    uint4 *items = (uint4 *)ws;
    const uint32_t item_cap = (uint32_t)ceil_div(N, N_PARTS) * (uint32_t)(tiles_x * tiles_x);
    ws += align256((size_t)2 * N_PARTS * item_cap * sizeof(uint4));
    unsigned long long *tie_mask = (unsigned long long *)ws;
    ws += align256((size_t)2 * N_PARTS * item_cap * sizeof(unsigned long long));
    float2 *fzr = (float2 *)ws;
    ws += align256((size_t)N * FT * sizeof(float2));
    uint32_t *gbox = (uint32_t *)ws;
    ws += align256((size_t)N * (FT / WAVE) * sizeof(uint32_t));
    float *img_bound = (float *)ws;
    ws += align256((size_t)N * sizeof(float));
    unsigned long long *loss_acc = (unsigned long long *)ws;
    ws += align256((size_t)N * sizeof(unsigned long long));
    const uint32_t list_cap = list_cap_of(m, S);
    uint2 *lists = (uint2 *)ws;
    ws += align256((size_t)N * list_cap * sizeof(uint2));
    ClipTables clip;
    clip.xv = (float *)ws; ws += align256((size_t)N * CLIP_VX * 12);
    clip.xg = (float *)ws; ws += align256((size_t)N * CLIP_VX * 8);
    clip.xsrc = (int2 *)ws; ws += align256((size_t)N * CLIP_VX * 8);
    clip.xcoef = (float2 *)ws; ws += align256((size_t)N * CLIP_VX * 8);
    clip.xf = (int *)ws; ws += align256((size_t)N * CLIP_FX * 12);
    clip.xcount = (uint32_t *)ws; ws += align256((size_t)N * 4);
    SMIL_HIP(hipMemsetAsync(ctr, 0, sizeof(RasterCounters), stream));
    if (rs->tie_rule) SMIL_HIP(hipMemsetAsync(tie_mask, 0, (size_t)2 * N_PARTS * item_cap * sizeof(unsigned long long), stream));
    const float sqrt_blur = sqrtf(rs->blur_radius);
    const int n_tiles = tiles_x * tiles_x;
    {
        SetupArgs q;
        q.verts_ndc = verts_ndc; q.faces = m->faces; q.tbox = tbox; q.gbox = gbox; q.items = items; q.item_cap = item_cap; q.fzr = fzr;
        q.ctr = ctr; q.V = m->V; q.F = m->F; q.S = S; q.tiles_x = tiles_x; q.sqrt_blur = sqrt_blur; q.z_clip = rs->z_clip;
        q.d_ndc_zero = d_ndc_zero; q.loss_src = loss_src; q.loss_dst = loss_dst; q.loss_acc = loss_acc; q.img_bound = img_bound; q.max_valence = m->max_valence;
        q.dndc_scale = dndc_scale; q.pix_scale = pix_scale; q.inv_sigma = 1.0f / rs->sigma; q.packed = packed;
        q.lists = lists; q.list_cap = list_cap; q.clip = clip;
        q.cd_counter = (grads && rs->clip_depth && rs->image0 == 0) ? rs->clip_depth->counter : nullptr;
        // per tile and copy: 8 bytes of counts + 4 bytes of list cursor (as many copies as fit 48 KB: two workgroups per CU), or one bit
        q.copies = n_tiles * 2 * 12 <= 48 * 1024 ? 2 : 1;  // (measured: two copies -10 % STICK / -17 % mouse at 256^2, four copies -5 % / -6 %)
#ifdef SETUP_COPIES
        q.copies = SETUP_COPIES;
#endif
        const size_t setup_lds = n_tiles <= COUNT_TILES_MAX ? (size_t)n_tiles * q.copies * 12 : (size_t)((n_tiles + 31) / 32) * sizeof(uint32_t);
        hipLaunchKernelGGL(k_raster_setup, dim3(N), dim3(SETUP_THREADS), setup_lds, stream, q);
        SMIL_LAUNCH_CHECK();
    }
    {
        const size_t grid = (size_t)tile_grid(N, tiles_x);
        a.list_stride = (int)(align256((size_t)FT * sizeof(uint32_t)) / sizeof(uint32_t));
        a.n_cf = (int)(align256((size_t)(FT / DCHUNK + 2) * sizeof(uint32_t)) / sizeof(uint32_t));
        ws += 256;
        a.slist = (uint2 *)ws;
        ws += grid * (size_t)a.list_stride * sizeof(uint2);
        a.slist2 = (uint32_t *)ws;
        ws += grid * (size_t)a.list_stride * sizeof(uint32_t);
        a.scfirst = (uint32_t *)ws;
        ws += grid * (size_t)a.n_cf * sizeof(uint32_t);
        a.sxy = (float2 *)ws;
        ws += grid * (size_t)a.list_stride * 3 * sizeof(float2);
        a.sid = (TriIds *)ws;
        ws += grid * (size_t)a.list_stride * sizeof(TriIds);
        const size_t stream = grid * (size_t)(REC_CAP + REC_PAD) * sizeof(Rec3);
        a.srec = (Rec3 *)ws; ws += stream;
        a.crec = (Rec3 *)ws;
    }
    a.lists = lists; a.list_cap = list_cap; a.clip = clip; a.FT = FT; a.slots = (unsigned int)tile_slots();
    a.tie_rule = rs->tie_rule; a.tie_mask = tie_mask; a.loss_acc = loss_acc;
    a.cd = SmilClipDepth{nullptr, nullptr, nullptr, nullptr, 0};
    a.image0 = rs->image0;
    if (grads && rs->clip_depth) {
        SMIL_REQUIRE(rs->clip_depth->vertex && rs->clip_depth->dz && rs->clip_depth->range && rs->clip_depth->counter && rs->clip_depth->capacity >= 0 &&
                     rs->image0 >= 0, "raster: incomplete SmilClipDepth");
        a.cd = *rs->clip_depth;
    }
    a.verts_ndc = verts_ndc; a.faces = m->faces; a.tbox = tbox; a.gbox = gbox; a.items = items; a.item_cap = item_cap; a.fzr = fzr; a.ctr = ctr; a.img_bound = img_bound; a.packed = 0;
    a.N = N; a.V = m->V; a.F = m->F; a.S = S; a.tiles_x = tiles_x; a.K = rs->faces_per_pixel;
    a.blur = rs->blur_radius; a.sqrt_blur = sqrt_blur; a.inv_sigma = 1.0f / rs->sigma; a.inv_sigma_log2e = (float)(1.4426950408889634 / (double)rs->sigma);
    HOOK_HOST_LAUNCH_SETUP(a, stream)
    a.sil = nullptr; a.grad_sil = nullptr; a.target = nullptr; a.target_u8 = nullptr; a.pix_scale = nullptr; a.loss_img = nullptr;
    a.d_ndc = nullptr;
    return SMIL_OK;
}

// ---- optional in-process timing of the tile kernel (bench.py): HIP events recorded on the launch stream ----
#define PROF_SLOTS 512
static bool g_prof_on = false;
static hipEvent_t g_prof_ev[PROF_SLOTS][2];
static int g_prof_n = 0;
static bool g_prof_init = false;

extern "C" int smil_profile_enable(int32_t on) {
    if (on && !g_prof_init) {
        for (int i = 0; i < PROF_SLOTS; ++i) {
            SMIL_HIP(hipEventCreate(&g_prof_ev[i][0]));
            SMIL_HIP(hipEventCreate(&g_prof_ev[i][1]));
        }
        g_prof_init = true;
    }
    g_prof_on = on != 0;
    g_prof_n = 0;
    return SMIL_OK;
}

// Sum / count of the tile-kernel durations recorded since smil_profile_enable(1).  Synchronises the events.
extern "C" int smil_profile_read(float *total_ms, int32_t *launches) {
    SMIL_REQUIRE(total_ms && launches, "smil_profile_read: null argument");
    float tot = 0.f;
    const int n = g_prof_n < PROF_SLOTS ? g_prof_n : PROF_SLOTS;
    for (int i = 0; i < n; ++i) {
        SMIL_HIP(hipEventSynchronize(g_prof_ev[i][1]));
        float ms = 0.f;
        SMIL_HIP(hipEventElapsedTime(&ms, g_prof_ev[i][0], g_prof_ev[i][1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = n;
    g_prof_n = 0;
    return SMIL_OK;
}

#define PROF_BEGIN(stream) \
    const int _slot = (g_prof_on && g_prof_n < PROF_SLOTS) ? g_prof_n++ : -1; \
    if (_slot >= 0) (void)hipEventRecord(g_prof_ev[_slot][0], stream)
#define PROF_END(stream) \
    if (_slot >= 0) (void)hipEventRecord(g_prof_ev[_slot][1], stream)

template <int MODE>
static void launch_tiles(const RasterArgs &a, int N, hipStream_t stream) {
    hipLaunchKernelGGL((k_raster_dense<MODE>), dim3(tile_grid(N, a.tiles_x)), dim3(64), 0, stream, a);
}
// (tie_rule 1) the pixels the tile kernel left out: the reference's queue replayed, one wave per pixel
template <int MODE>
static void launch_tie_replay(const RasterArgs &a, hipStream_t stream) {
    // dynamic LDS: the face-id bitmap (FT bits), the filling queue (128 slots x 3 words), the tile's ordered face ids (17 bits each)
    const size_t lds = (size_t)a.FT / 8 + 6 * WAVE * sizeof(uint32_t) + (size_t)TIE_ORD_CAP * sizeof(uint16_t) + TIE_ORD_CAP / 8;
    // (as many waves as the registers let a CU hold - the ~6 KB of LDS per wave allow more; they take tickets until none is left)
    if (a.tie_rule) hipLaunchKernelGGL((k_raster_tie_replay<MODE>), dim3((unsigned int)device_cus() * 4u * TIE_WAVES_PER_SIMD), dim3(64), lds, stream, a);
}

extern "C" int smil_silhouette_forward(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                                       const SmilRasterSettings *rs, float *sil, void *workspace, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterArgs a;
    int rc = raster_common(m, verts_ndc, N, S, rs, workspace, stream, a);
    if (rc) return rc;
    SMIL_REQUIRE(sil, "smil_silhouette_forward: null output");
    SMIL_HIP(hipMemsetAsync(sil, 0, (size_t)N * S * S * sizeof(float), stream));
    a.sil = sil;
    PROF_BEGIN(stream);
    launch_tiles<MODE_FWD>(a, N, stream);
    PROF_END(stream);
    launch_tie_replay<MODE_FWD>(a, stream);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_silhouette_backward(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                                        const SmilRasterSettings *rs, const float *grad_sil, float *d_ndc,
                                        void *workspace, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterArgs a;
    int rc = raster_common(m, verts_ndc, N, S, rs, workspace, stream, a, nullptr, nullptr, nullptr, nullptr, nullptr, 0, true);
    if (rc) return rc;
    SMIL_REQUIRE(grad_sil && d_ndc, "smil_silhouette_backward: null argument");
    SMIL_HIP(hipMemsetAsync(d_ndc, 0, (size_t)N * m->V * 2 * sizeof(float), stream));
    a.grad_sil = grad_sil; a.d_ndc = d_ndc;
    PROF_BEGIN(stream);
    launch_tiles<MODE_BWD>(a, N, stream);
    PROF_END(stream);
    launch_tie_replay<MODE_BWD>(a, stream);
    SMIL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_clip_backward, dim3(N), dim3(64), 0, stream, a.clip, d_ndc, m->V, (float *)nullptr, (const unsigned long long *)nullptr, verts_ndc, rs->z_clip, a.cd, a.image0);  // (new vertices of cut faces -> their edges' end points)
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_silhouette_l1_fused(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                                        const SmilRasterSettings *rs, const void *target, int32_t target_is_u8,
                                        const float *target_sum, const float *pix_scale, float *loss_img, float *d_ndc,
                                        float *sil_out, float *d_ndc_scale, void *workspace, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterArgs a;
    SMIL_REQUIRE(target && target_sum && pix_scale && loss_img && d_ndc, "smil_silhouette_l1_fused: null argument");
    // large launches accumulate the vertex gradients as packed fixed point (half the memory-side atomics); small ones keep float
    // atomics.  The packed rows are decoded in place afterwards, unless the caller takes them as they are (d_ndc_scale).
    const int packed = (N >= PACKED_MIN_IMAGES && (reinterpret_cast<uintptr_t>(d_ndc) & 7u) == 0u) ? 1 : 0;  // (64-bit atomics need 8-byte alignment)
    int rc = raster_common(m, verts_ndc, N, S, rs, workspace, stream, a, d_ndc, target_sum, loss_img, d_ndc_scale, pix_scale, packed, true);
    if (rc) return rc;
    if (sil_out) SMIL_HIP(hipMemsetAsync(sil_out, 0, (size_t)N * S * S * sizeof(float), stream));
    if (target_is_u8) a.target_u8 = (const uint8_t *)target; else a.target = (const float *)target;
    a.pix_scale = pix_scale; a.loss_img = loss_img; a.d_ndc = d_ndc; a.sil = sil_out;
    a.packed = packed;
    PROF_BEGIN(stream);
    launch_tiles<MODE_FUSED>(a, N, stream);
    PROF_END(stream);
    launch_tie_replay<MODE_FUSED>(a, stream);
    SMIL_LAUNCH_CHECK();
    hipLaunchKernelGGL(k_clip_backward, dim3(N), dim3(64), 0, stream, a.clip, d_ndc, m->V, loss_img, (const unsigned long long *)a.loss_acc, verts_ndc, rs->z_clip, a.cd, a.image0);  // (new vertices of cut faces -> their edges' end points; the images' loss sums)
    SMIL_LAUNCH_CHECK();
    if (a.packed && !d_ndc_scale) {
        hipLaunchKernelGGL(k_unpack_dndc, dim3(N, ceil_div(m->V, 256)), dim3(256), 0, stream, d_ndc, a.img_bound, pix_scale, a.inv_sigma, m->V, a.clip.xcount);
        SMIL_LAUNCH_CHECK();
    }
    return SMIL_OK;
}
