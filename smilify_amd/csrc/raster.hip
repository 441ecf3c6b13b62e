// Soft-silhouette rasteriser for gfx950: forward, backward and fused forward+L1+backward.
//
// Replaces (reference): smal_fitter/p3d_renderer.py:41-52,142-146 - MeshRasterizer(bin_size=0,
// faces_per_pixel=100, blur_radius=log(1/1e-4-1)*1e-4, perspective-correct, clipped barycentrics) +
// SoftSilhouetteShader (sigmoid_alpha_blend) - and, in the fused entry point, the silhouette L1 term of
// SMALFitter.forward (fitter.py:332-333) with its gradient.  The per-(pixel,face) arithmetic restates
// pytorch3d 0.7.x (un-vendored): CheckPixelInsideFace / RasterizeMeshesBackward / geometry_utils.
//
// Design (see DESIGN.md "raster"):
//  * The reference visits all F faces for all S^2 pixels and stores (S,S,K) fragments in HBM
//    (157 MB / image at 256^2).  Here nothing per-fragment ever reaches HBM.
//  * k_raster_setup (one workgroup per image): per-face validity + blurred bbox in 8x8-pixel tile
//    units (4 x u8 packed), a tile-occupancy bitmap in LDS, and the compacted list of touched tiles
//    appended to a global work list.  Untouched tiles are never visited (their silhouette is 0).
//  * k_raster_tiles (persistent, one 64-lane wavefront = one 8x8 tile, lane = pixel): dequeues work
//    items, compacts the face ids whose bbox meets the tile (ballot prefix -> order preserved), stages
//    their vertices through LDS in chunks of 64 and streams them to all 64 pixels (LDS broadcast reads).
//      pass 1: candidate count, product of all candidates and the K smallest depths of every pixel kept
//              SORTED IN REGISTERS: inserting z is r[i] = med3(r[i-1], r[i], z) for all i - one
//              v_med3_f32 per slot, branch-free, no LDS (an LDS heap capped occupancy at 1 wave/SIMD and
//              its sift loops were dependent-LDS-latency chains).  Pixels with <= K candidates are done.
//      pass 2: (only if some pixel has > K candidates) product over the K nearest: depth < t, plus the
//              first r faces (ascending face id) with depth == t, t = K-th smallest depth.
//      pass 3: (backward / fused) per-pair gradient added to per-face LDS accumulators (ds_add_f32),
//              flushed once per 64-face chunk with one global atomic per touched vertex component.
//  * Tile lists that can truncate are sorted front to back (16-bit quantised nearest-vertex depth | face id, bitonic
//    sort in LDS), so every pass stops at the first 64-face chunk that lies wholly beyond what any pixel still needs.
//  * Deviation from the reference kept on purpose: ties at the K-th depth are resolved by visiting order (depth
//    bucket, then face id); the reference's unsorted-queue eviction depends on visiting history (measured effect on
//    the L1 loss: ~1e-5 relative; tests/test_gpu_kernels.py::test_silhouette_forward).
#include "common.h"

#define TILE 8
#define LIST_CAP 2048       // face ids per list segment (LDS)
#define FCHUNK 64           // faces staged per chunk
#define FREC 32             // floats per staged face record
#define K_EPS 1e-8f
#define ALPHA_GRAD_EPS 1e-12f  // pixels whose transmittance is below this contribute no gradient
#define DGROUP 4            // 64-record rows in flight in the dense pass 3
#define SGROUP 8            // stream records in flight per lane in passes 2 and 3
#define VAL_CAP 65536       // pair records (16 B) a workgroup can carry from pass 1 to passes 2 and 3

enum { MODE_FWD = 0, MODE_BWD = 1, MODE_FUSED = 2 };

struct RasterCounters {
    unsigned int n_items;
    unsigned int next;
};

struct RasterArgs {
    const float *verts_ndc;  // (N,V,3)
    const int *faces;        // (F,3)
    const uint32_t *tbox;    // (N,F)
    const float *fzmin;      // (N,F) nearest vertex depth of every face
    const uint32_t *items;   // work list
    RasterCounters *ctr;
    int N, V, F, S, tiles_x, K;
    float blur, sqrt_blur, inv_sigma;
    // outputs / inputs per mode
    float *sil;              // (N,S,S) FWD (or optional in FUSED)
    const float *grad_sil;   // BWD
    const float *target;     // FUSED (fp32 targets) ...
    const uint8_t *target_u8; // ... or binary {0,1} targets stored as bytes
    const float *pix_scale;  // FUSED (N,)
    float *loss_img;         // FUSED (N,)
    float *d_ndc;            // (N,V,2)
    // pair stream (STREAM kernels): per resident workgroup LIST_CAP headers + VAL_CAP records
    uint4 *shdr;
    float4 *sval;
    uint32_t *smeta;         // per record: pixel (lane) | list position << 6
    unsigned long long *dbg; // DBG_TIMERS builds: per-phase cycle sums
};

__device__ __forceinline__ float pix_to_ndc(int i, int S) { return -1.0f + (2.0f * (float)i + 1.0f) / (float)S; }

__device__ __forceinline__ float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

// ---------------------------------------------------------------------------------------------
// setup: per-face tile boxes + touched-tile work list
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_raster_setup(const float *__restrict__ verts_ndc, const int *__restrict__ faces,
                                                      uint32_t *__restrict__ tbox, float *__restrict__ fzmin,
                                                      uint32_t *__restrict__ items,
                                                      RasterCounters *ctr, int V, int F, int S, int tiles_x,
                                                      float sqrt_blur) {
    extern __shared__ uint32_t bitmap[];  // tiles_x*tiles_x bits, then 256 scan slots
    const int n = blockIdx.x;
    const int n_tiles = tiles_x * tiles_x;
    const int n_words = (n_tiles + 31) >> 5;
    uint32_t *scan = bitmap + n_words;
    for (int i = threadIdx.x; i < n_words; i += blockDim.x) bitmap[i] = 0u;
    __syncthreads();
    const float *vn = verts_ndc + (size_t)n * V * 3;
    const float fS = (float)S;
    for (int f = threadIdx.x; f < F; f += blockDim.x) {
        const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
        const float x0 = vn[3 * i0], y0 = vn[3 * i0 + 1], z0 = vn[3 * i0 + 2];
        const float x1 = vn[3 * i1], y1 = vn[3 * i1 + 1], z1 = vn[3 * i1 + 2];
        const float x2 = vn[3 * i2], y2 = vn[3 * i2 + 1], z2 = vn[3 * i2 + 2];
        uint32_t box = 0x0000FFFFu;  // empty: tx0 = ty0 = 255 > tx1 = ty1 = 0
        const float zmin = fminf(fminf(z0, z1), z2);
        const float area = edge_fn(x0, y0, x1, y1, x2, y2);
        const bool finite = (x0 == x0) && (x1 == x1) && (x2 == x2) && (y0 == y0) && (y1 == y1) && (y2 == y2);
        if (finite && !(zmin < K_EPS) && !(area <= K_EPS && area >= -K_EPS)) {
            const float xlo = fminf(fminf(x0, x1), x2) - sqrt_blur, xhi = fmaxf(fmaxf(x0, x1), x2) + sqrt_blur;
            const float ylo = fminf(fminf(y0, y1), y2) - sqrt_blur, yhi = fmaxf(fmaxf(y0, y1), y2) + sqrt_blur;
            // pixel index i (flipped axis) has centre -1 + (2i+1)/S: centres inside [lo,hi] are ceil(v_lo)..floor(v_hi)
            // with v = ((x+1) S - 1)/2; 0.01 px of slack covers the float rounding of both sides
            int xi_lo = (int)ceilf(((xlo + 1.0f) * fS - 1.0f) * 0.5f - 0.01f), xi_hi = (int)floorf(((xhi + 1.0f) * fS - 1.0f) * 0.5f + 0.01f);
            int yi_lo = (int)ceilf(((ylo + 1.0f) * fS - 1.0f) * 0.5f - 0.01f), yi_hi = (int)floorf(((yhi + 1.0f) * fS - 1.0f) * 0.5f + 0.01f);
            xi_lo = max(xi_lo, 0); yi_lo = max(yi_lo, 0);
            xi_hi = min(xi_hi, S - 1); yi_hi = min(yi_hi, S - 1);
            if (xi_lo <= xi_hi && yi_lo <= yi_hi) {
                // output column xo = S-1-xi
                const int tx0 = (S - 1 - xi_hi) / TILE, tx1 = (S - 1 - xi_lo) / TILE;
                const int ty0 = (S - 1 - yi_hi) / TILE, ty1 = (S - 1 - yi_lo) / TILE;
                box = (uint32_t)tx0 | ((uint32_t)ty0 << 8) | ((uint32_t)tx1 << 16) | ((uint32_t)ty1 << 24);
                for (int ty = ty0; ty <= ty1; ++ty)
                    for (int tx = tx0; tx <= tx1; ++tx) {
                        const int t = ty * tiles_x + tx;
                        atomicOr(&bitmap[t >> 5], 1u << (t & 31));
                    }
            }
        }
        tbox[(size_t)n * F + f] = box;
        fzmin[(size_t)n * F + f] = zmin;
    }
    __syncthreads();
    // ordered compaction of touched tiles -> global work list
    uint32_t mine = 0;
    for (int w = threadIdx.x; w < n_words; w += blockDim.x) mine += __popc(bitmap[w]);
    scan[threadIdx.x] = mine;
    __syncthreads();
    __shared__ uint32_t base_slot;
    if (threadIdx.x == 0) {
        uint32_t run = 0;
        for (int i = 0; i < (int)blockDim.x; ++i) { const uint32_t c = scan[i]; scan[i] = run; run += c; }
        base_slot = run ? atomicAdd(&ctr->n_items, run) : 0u;
    }
    __syncthreads();
    uint32_t pos = base_slot + scan[threadIdx.x];
    for (int w = threadIdx.x; w < n_words; w += blockDim.x) {
        uint32_t bits = bitmap[w];
        while (bits) {
            const int bit = __ffs(bits) - 1;
            bits &= bits - 1;
            items[pos++] = (uint32_t)n * (uint32_t)n_tiles + (uint32_t)(w * 32 + bit);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// per-(pixel, face) evaluation
// ---------------------------------------------------------------------------------------------
// Face record staged in LDS (32 floats = 8 x 16 B, read with ds_read_b128 broadcasts).  Everything that does
// not depend on the pixel is folded in once per (tile, face): coordinates are relative to the tile centre
// (cx, cy) so the affine forms below do not cancel catastrophically.
//   w_i(p) = A_i dx + B_i dy + C_i  = b_i(p) * z_j z_k   (perspective-correct barycentric numerators; the
//            common denominator is positive, so inside <=> all w_i > 0)
struct alignas(16) FaceRec {
    float xmin, xmax, ymin, ymax;   // blurred bbox (absolute NDC)
    float A0, B0, C0, A1;
    float B1, C1, A2, B2;
    float C2, z0, z1, z2;
    float x0c, y0c, x1c, y1c;       // v0, v1 relative to the tile centre
    float e01x, e01y, rl01, e02x;   // edge vectors and 1/|e|^2 (0 for a degenerate edge)
    float e02y, rl02, e12x, e12y;
    float rl12;
    int i0, i1, i2;
};
static_assert(sizeof(FaceRec) == FREC * sizeof(float), "FaceRec layout");

struct PairEval {
    bool cand, inside;
    float sd;             // signed squared distance
    float w0, w1, w2;
    // closest edge in the reference's order e01, e02, e12 with <= ties (its backward treats t as a constant):
    // edge 0 = (v0,v1), 1 = (v0,v2), 2 = (v1,v2); t = clamped projection; (rx, ry) = closest point minus pixel,
    // sq2(rx, ry) == |sd| bit for bit
    int edge;
    float t, rx, ry;
};

// |r|^2 with one fixed rounding sequence, so that a distance recomputed from a stored (rx, ry) is bit-identical
__device__ __forceinline__ float sq2(float x, float y) { return __fmaf_rn(x, x, y * y); }

// Branch-free: every lane computes everything; `cand` says whether the pair exists.
__device__ __forceinline__ void eval_pair(const FaceRec &f, float px, float py, float dxp, float dyp, float blur, PairEval &e) {
    const bool in_bb = !(px > f.xmax || px < f.xmin || py > f.ymax || py < f.ymin);
    e.w0 = fmaf(f.A0, dxp, fmaf(f.B0, dyp, f.C0));
    e.w1 = fmaf(f.A1, dxp, fmaf(f.B1, dyp, f.C1));
    e.w2 = fmaf(f.A2, dxp, fmaf(f.B2, dyp, f.C2));
    e.inside = (e.w0 > 0.f) && (e.w1 > 0.f) && (e.w2 > 0.f);
    const float dx0 = dxp - f.x0c, dy0 = dyp - f.y0c, dx1 = dxp - f.x1c, dy1 = dyp - f.y1c;
    const float t01 = __builtin_amdgcn_fmed3f((f.e01x * dx0 + f.e01y * dy0) * f.rl01, 0.f, 1.f);
    const float t02 = __builtin_amdgcn_fmed3f((f.e02x * dx0 + f.e02y * dy0) * f.rl02, 0.f, 1.f);
    const float t12 = __builtin_amdgcn_fmed3f((f.e12x * dx1 + f.e12y * dy1) * f.rl12, 0.f, 1.f);
    const float r01x = fmaf(t01, f.e01x, -dx0), r01y = fmaf(t01, f.e01y, -dy0);
    const float r02x = fmaf(t02, f.e02x, -dx0), r02y = fmaf(t02, f.e02y, -dy0);
    const float r12x = fmaf(t12, f.e12x, -dx1), r12y = fmaf(t12, f.e12y, -dy1);
    const float d01 = sq2(r01x, r01y), d02 = sq2(r02x, r02y), d12 = sq2(r12x, r12y);
    const float dist = fminf(fminf(d01, d02), d12);
    e.cand = in_bb && (e.inside || dist < blur);
    e.sd = e.inside ? -dist : dist;
    const bool c01 = (d01 <= d02) && (d01 <= d12);
    const bool c02 = !c01 && (d02 <= d01) && (d02 <= d12);
    e.edge = c01 ? 0 : (c02 ? 1 : 2);
    e.t = c01 ? t01 : (c02 ? t02 : t12);
    e.rx = c01 ? r01x : (c02 ? r02x : r12x);
    e.ry = c01 ? r01y : (c02 ? r02y : r12y);
}

// depth at the clipped, renormalised perspective-correct barycentrics:
// c_i = max(p_i,0) / max(sum, 1e-5), p_i = w_i / den; 1/den cancels: c_i = max(w_i,0) / max(sum max(w,0), 1e-5 den).
// When a single weight survives the clip the depth is EXACTLY that vertex's depth, so faces sharing the vertex tie
// exactly (as x / x == 1 does in the reference) and the (depth, face id) order stays well defined.
__device__ __forceinline__ float pair_depth(const FaceRec &f, const PairEval &e) {
    const float den = fmaxf(e.w0 + e.w1 + e.w2, K_EPS);
    const float m0 = fmaxf(e.w0, 0.f), m1 = fmaxf(e.w1, 0.f), m2 = fmaxf(e.w2, 0.f);
    const float cs = fmaxf(m0 + m1 + m2, 1e-5f * den);
    const float rc = __builtin_amdgcn_rcpf(cs);
    float pz = (m0 * rc) * f.z0 + (m1 * rc) * f.z1 + (m2 * rc) * f.z2;
    pz = (m1 == 0.f && m2 == 0.f && m0 >= cs) ? f.z0 : pz;
    pz = (m0 == 0.f && m2 == 0.f && m1 >= cs) ? f.z1 : pz;
    pz = (m0 == 0.f && m1 == 0.f && m2 >= cs) ? f.z2 : pz;
    return pz;
}

__device__ __forceinline__ float face_prob(float sd, float inv_sigma) {
    // sigmoid(-dist / sigma) = 1 / (1 + e^{dist/sigma}); v_exp_f32 + v_rcp_f32 (1 ulp each)
    return __builtin_amdgcn_rcpf(1.0f + __expf(sd * inv_sigma));
}

// ---------------------------------------------------------------------------------------------
// tile kernel
// ---------------------------------------------------------------------------------------------
struct alignas(16) TileLds {
    float rec[FCHUNK * FREC];
    float gacc[FCHUNK * 6];  // per staged face: d/d(x0,y0,x1,y1,x2,y2), pass 3
    uint32_t list[LIST_CAP];
    // stream form of pass 3: per-pixel state (gathered by the lane that owns a record) and, per 64-face chunk of the
    // list, the index of its first record
    float pcoef[WAVE], pzt[WAVE];
    int ptie[WAVE];
    uint32_t cfirst[LIST_CAP / FCHUNK + 1];
};

// Build the ordered list of faces of [seg0, seg1) whose tile box contains (tx,ty). Returns the count.
__device__ __forceinline__ int build_list(const uint32_t *__restrict__ tbox_n, int seg0, int seg1, int tx, int ty,
                                          uint32_t *list, int lane, int cap = LIST_CAP) {
    int cnt = 0;
    for (int base = seg0; base < seg1; base += 4 * WAVE) {
        // four independent loads in flight per lane
        uint32_t b[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int f = base + u * WAVE + lane;
            b[u] = f < seg1 ? tbox_n[f] : 0x0000FFFFu;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int tx0 = b[u] & 0xFF, ty0 = (b[u] >> 8) & 0xFF, tx1 = (b[u] >> 16) & 0xFF, ty1 = b[u] >> 24;
            const bool hit = (tx >= tx0) && (tx <= tx1) && (ty >= ty0) && (ty <= ty1);
            const unsigned long long mask = __ballot(hit);
            const int pos = cnt + __popcll(mask & ((1ull << lane) - 1ull));
            if (hit && pos < cap) list[pos] = (uint32_t)(base + u * WAVE + lane);
            cnt += __popcll(mask);
        }
    }
    return cnt;
}

__device__ __forceinline__ void stage_faces(const RasterArgs &a, const float *__restrict__ vn, const uint32_t *list,
                                            int c0, int m, float *rec, int lane, float cx, float cy, uint32_t id_mask) {
    if (lane < m) {
        const int f = (int)(list[c0 + lane] & id_mask);
        const int i0 = a.faces[3 * f], i1 = a.faces[3 * f + 1], i2 = a.faces[3 * f + 2];
        const float x0 = vn[3 * i0], y0 = vn[3 * i0 + 1], z0 = vn[3 * i0 + 2];
        const float x1 = vn[3 * i1], y1 = vn[3 * i1 + 1], z1 = vn[3 * i1 + 2];
        const float x2 = vn[3 * i2], y2 = vn[3 * i2 + 1], z2 = vn[3 * i2 + 2];
        FaceRec r;
        r.xmin = fminf(fminf(x0, x1), x2) - a.sqrt_blur; r.xmax = fmaxf(fmaxf(x0, x1), x2) + a.sqrt_blur;
        r.ymin = fminf(fminf(y0, y1), y2) - a.sqrt_blur; r.ymax = fmaxf(fmaxf(y0, y1), y2) + a.sqrt_blur;
        const float rcp_area = 1.0f / (edge_fn(x2, y2, x0, y0, x1, y1) + K_EPS);
        // edge function e_k(p) = (px - ax)(by - ay) - (py - ay)(bx - ax), linear in p; value at the tile centre + slopes
        const float s0 = rcp_area * (z1 * z2), s1 = rcp_area * (z0 * z2), s2 = rcp_area * (z0 * z1);
        r.A0 = (y2 - y1) * s0; r.B0 = -(x2 - x1) * s0; r.C0 = edge_fn(cx, cy, x1, y1, x2, y2) * s0;
        r.A1 = (y0 - y2) * s1; r.B1 = -(x0 - x2) * s1; r.C1 = edge_fn(cx, cy, x2, y2, x0, y0) * s1;
        r.A2 = (y1 - y0) * s2; r.B2 = -(x1 - x0) * s2; r.C2 = edge_fn(cx, cy, x0, y0, x1, y1) * s2;
        r.z0 = z0; r.z1 = z1; r.z2 = z2;
        r.x0c = x0 - cx; r.y0c = y0 - cy; r.x1c = x1 - cx; r.y1c = y1 - cy;
        r.e01x = x1 - x0; r.e01y = y1 - y0; r.e02x = x2 - x0; r.e02y = y2 - y0; r.e12x = x2 - x1; r.e12y = y2 - y1;
        const float l01 = r.e01x * r.e01x + r.e01y * r.e01y, l02 = r.e02x * r.e02x + r.e02y * r.e02y,
                    l12 = r.e12x * r.e12x + r.e12y * r.e12y;
        r.rl01 = l01 <= K_EPS ? 0.f : 1.0f / l01;
        r.rl02 = l02 <= K_EPS ? 0.f : 1.0f / l02;
        r.rl12 = l12 <= K_EPS ? 0.f : 1.0f / l12;
        r.i0 = i0; r.i1 = i1; r.i2 = i2;
        *reinterpret_cast<FaceRec *>(rec + lane * FREC) = r;
    }
}

// Sort the cached tile list front to back, in place.  Every entry becomes (16-bit quantised nearest-vertex depth
// of the face << 16) | face id, and the u32 values are sorted with a bitonic network in LDS (ties: ascending face
// id).  Needs F < 65536.  (lo, step) let a pass recover a LOWER BOUND of the depth of every remaining face from the
// entry it is looking at: depth >= lo + (entry >> 16) * step, because a pair's depth is a convex combination of
// the face's vertex depths.
__device__ __forceinline__ void depth_sort_list(const float *__restrict__ fz_n, uint32_t *list, int n, int lane, float &zlo,
                                                float &zstep) {
    float mn = 3.0e38f, mx = 0.f;
    for (int i = lane; i < n; i += WAVE) {
        const float z = fz_n[list[i]];
        mn = fminf(mn, z); mx = fmaxf(mx, z);
    }
    for (int o = 32; o > 0; o >>= 1) { mn = fminf(mn, __shfl_xor(mn, o, 64)); mx = fmaxf(mx, __shfl_xor(mx, o, 64)); }
    zlo = mn;
    zstep = fmaxf((mx - mn) * (1.0f / 65535.0f), 1e-30f);
    const float inv = 1.0f / zstep;
    int np2 = 64;
    while (np2 < n) np2 <<= 1;
    for (int i = lane; i < np2; i += WAVE) {
        uint32_t key = 0xFFFFFFFFu;
        if (i < n) {
            const uint32_t f = list[i];
            // floor minus one step of slack against rounding: lo + q*step must never exceed the true depth
            int q = (int)((fz_n[f] - zlo) * inv) - 1;
            q = min(max(q, 0), 65534);
            key = ((uint32_t)q << 16) | f;
        }
        list[i] = key;
    }
    __syncthreads();
    for (int k = 2; k <= np2; k <<= 1)
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = lane; t < (np2 >> 1); t += WAVE) {
                const int lo_i = ((t & ~(j - 1)) << 1) | (t & (j - 1)), hi_i = lo_i | j;
                const uint32_t x = list[lo_i], y = list[hi_i];
                const bool asc = (lo_i & k) == 0;
                if ((x > y) == asc) { list[lo_i] = y; list[hi_i] = x; }
            }
            __syncthreads();
        }
}

// Loop skeleton shared by the three passes: ordered face list per 1024-face segment, 64-face chunks staged in LDS.
// When the whole tile list fits the LDS buffer it is built once (list_cached) and reused by every pass.
// CUT_EXPR (evaluated once per chunk, wave-uniform float): when the list is depth sorted, every face from the chunk's
// first entry on is at least `lo + q*step` deep; once that bound reaches CUT_EXPR nothing further can matter.
#define CHUNK_LOOP_BEGIN(ZERO_GACC, CUT_EXPR)                                                  \
    for (int seg0 = 0; seg0 < (list_cached ? 1 : a.F); seg0 += LIST_CAP) {                     \
        const int seg1 = min(a.F, seg0 + LIST_CAP);                                            \
        const int ln = list_cached ? list_total : build_list(tbox_n, seg0, seg1, tx, ty, lds.list, lane); \
        __syncthreads();                                                                       \
        for (int c0 = 0; c0 < ln; c0 += FCHUNK) {                                              \
            if (sorted) {                                                                      \
                const float zlb = zlo + (float)(lds.list[c0] >> 16) * zstep;                   \
                if (zlb >= (CUT_EXPR)) break;                                                  \
            }                                                                                  \
            const int m = min(FCHUNK, ln - c0);                                                \
            stage_faces(a, vn, lds.list, c0, m, lds.rec, lane, cx, cy, id_mask);               \
            if (ZERO_GACC)                                                                     \
                for (int i_ = lane; i_ < FCHUNK * 6; i_ += WAVE) lds.gacc[i_] = 0.f;           \
            __syncthreads();
#define CHUNK_LOOP_END                                                                         \
            __syncthreads();                                                                   \
        }                                                                                      \
    }

// Pass 3 (stream form): add the LDS accumulators of list chunk `chunk` to the vertex gradients, one global atomic per
// touched vertex component.  Lane = face of the chunk.
__device__ __forceinline__ void flush_gacc(const RasterArgs &a, TileLds &lds, float *dn, int chunk, int list_total,
                                           uint32_t id_mask, int lane) {
    __syncthreads();
    const int c0 = chunk * FCHUNK;
    if (c0 + lane < list_total) {
        const int f = (int)(lds.list[c0 + lane] & id_mask);
        const float *acc = lds.gacc + lane * 6;
        const int i0 = a.faces[3 * f], i1 = a.faces[3 * f + 1], i2 = a.faces[3 * f + 2];
        if (acc[0] != 0.f) atomicAdd(&dn[2 * i0], acc[0]);
        if (acc[1] != 0.f) atomicAdd(&dn[2 * i0 + 1], acc[1]);
        if (acc[2] != 0.f) atomicAdd(&dn[2 * i1], acc[2]);
        if (acc[3] != 0.f) atomicAdd(&dn[2 * i1 + 1], acc[3]);
        if (acc[4] != 0.f) atomicAdd(&dn[2 * i2], acc[4]);
        if (acc[5] != 0.f) atomicAdd(&dn[2 * i2 + 1], acc[5]);
    }
    __syncthreads();
}

// EXACT: K == KT is known at compile time (the common K = 100 case): the K-th smallest is simply the last slot.
// Otherwise the slot is picked with a chain of selects, which costs ~KT scalar lane masks - kept off the hot path.
template <int KT, bool EXACT>
__device__ __forceinline__ float kth_smallest(const float (&r)[KT], int K) {
    if (EXACT) return r[KT - 1];
    float v = 3.0e38f;
#pragma unroll
    for (int i = 0; i < KT; ++i)
        if (i == K - 1) v = r[i];
    return v;
}

// KT = number of register slots holding the smallest depths (>= K); 2 waves per SIMD.
// STREAM: pass 1 appends, for every (face, pixel) pair it accepts, a 16-byte record {depth, rx, ry, t|code} to a
// per-workgroup stream in global memory (ballot-compacted: one contiguous store per face, plus a 16-byte header
// {pixel mask, list position, first record}); passes 2 and 3 then walk that stream instead of re-evaluating every
// face against every pixel.  The region is reused for every tile the workgroup processes, so it lives in L2 / MALL.
// A tile whose pairs do not fit (VAL_CAP) or whose list is not cached falls back to the re-evaluating passes.
template <int MODE, int KT, bool EXACT, bool STREAM>
__global__ void __launch_bounds__(64, 2) k_raster_tiles(RasterArgs a) {
    __shared__ TileLds lds;
    const int lane = threadIdx.x;
    uint4 *const shdr = STREAM ? a.shdr + (size_t)blockIdx.x * LIST_CAP : nullptr;
    float4 *const sval = STREAM ? a.sval + (size_t)blockIdx.x * VAL_CAP : nullptr;
    uint32_t *const smeta = STREAM ? a.smeta + (size_t)blockIdx.x * VAL_CAP : nullptr;
    const uint32_t lane_lo = lane < 32 ? 1u << lane : 0u, lane_hi = lane >= 32 ? 1u << (lane - 32) : 0u;
    const int K = EXACT ? KT : a.K;
    const int n_tiles = a.tiles_x * a.tiles_x;
    const unsigned int n_items = a.ctr->n_items;

#ifdef DBG_TIMERS
    unsigned long long tph[6] = {0, 0, 0, 0, 0, 0};
    unsigned long long tlast = __builtin_readcyclecounter();
#define TMARK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[k] += now_ - tlast; tlast = now_; }
#else
#define TMARK(k)
#endif
    while (true) {
        unsigned int item = 0;
        if (lane == 0) item = atomicAdd(&a.ctr->next, 1u);
        item = __builtin_amdgcn_readfirstlane(item);
        if (item >= n_items) break;
        const uint32_t code = a.items[item];
        const int n = (int)(code / (uint32_t)n_tiles), tile = (int)(code % (uint32_t)n_tiles);
        const int tx = tile % a.tiles_x, ty = tile / a.tiles_x;
        const int xo = tx * TILE + (lane & 7), yo = ty * TILE + (lane >> 3);
        const bool in_img = xo < a.S && yo < a.S;
        // pixels outside the image get a position no bbox can contain
        const float px = in_img ? pix_to_ndc(a.S - 1 - xo, a.S) : 3.0e38f, py = pix_to_ndc(a.S - 1 - yo, a.S);
        const float cx = pix_to_ndc(a.S - 1 - (tx * TILE + 4), a.S), cy = pix_to_ndc(a.S - 1 - (ty * TILE + 4), a.S);
        const float dxp = px - cx, dyp = py - cy;
        const float *vn = a.verts_ndc + (size_t)n * a.V * 3;
        const uint32_t *tbox_n = a.tbox + (size_t)n * a.F;
        const size_t pix = ((size_t)n * a.S + yo) * a.S + xo;

        const int list_total = build_list(tbox_n, 0, a.F, tx, ty, lds.list, lane);
        const bool list_cached = list_total <= LIST_CAP;
        const bool may_truncate = list_total > K;  // otherwise no pixel can see more than K faces
        // depth ordering pays only where truncation can happen; it needs the whole list in LDS and 16-bit face ids
        const bool sorted = list_cached && may_truncate && a.F < 65536 && list_total > FCHUNK;
        float zlo = 0.f, zstep = 0.f;
        __syncthreads();
        if (sorted) depth_sort_list(a.fzmin + (size_t)n * a.F, lds.list, list_total, lane, zlo, zstep);
        const uint32_t id_mask = sorted ? 0xFFFFu : 0xFFFFFFFFu;

        TMARK(0)
        // ---------------- pass 1: count, product of all, K smallest depths (sorted, in registers) ---------
        int cnt = 0;
        float prod_all = 1.0f;
        int n_emit = 0, vbase = 0;            // headers / records written so far (wave-uniform)
        bool stream_ok = STREAM && list_cached;
        float r[KT];
#pragma unroll
        for (int i = 0; i < KT; ++i) r[i] = 3.0e38f;
        // a pixel is settled once it holds K depths and its K-th smallest is not beyond the next face; the pass may
        // stop when every pixel of the tile is settled (pixels with fewer than K candidates never are)
        int n_chunks = 0;  // 64-face chunks pass 1 visited (it may stop early on a depth-sorted list)
        CHUNK_LOOP_BEGIN(false, wave_max(!in_img ? -3.0e38f : (cnt < K ? 3.0e38f : kth_smallest<KT, EXACT>(r, K))))
        {
            if (STREAM && stream_ok) {
                if (lane == 0) lds.cfirst[c0 >> 6] = (uint32_t)vbase;
                n_chunks = (c0 >> 6) + 1;
            }
            for (int i = 0; i < m; ++i) {
                const FaceRec f = *reinterpret_cast<const FaceRec *>(lds.rec + i * FREC);
                if (__ballot(!(px > f.xmax || px < f.xmin || py > f.ymax || py < f.ymin)) == 0ull) continue;
                PairEval e;
                eval_pair(f, px, py, dxp, dyp, a.blur, e);
                const unsigned long long cm = __ballot(e.cand);
                if (cm == 0ull) continue;
                const float fac = 1.0f - face_prob(e.sd, a.inv_sigma);
                prod_all *= e.cand ? fac : 1.0f;
                cnt += e.cand ? 1 : 0;
                const float z = (may_truncate && e.cand) ? pair_depth(f, e) : 3.0e38f;
                if (STREAM && stream_ok) {
                    const int nc = __popcll(cm);
                    if (vbase + nc > VAL_CAP) {
                        stream_ok = false;  // wave-uniform: this tile re-evaluates in passes 2 and 3
                    } else {
                        // 3 code bits (inside, edge) replace the low mantissa bits of t in [0,1] (<= 4e-7 relative)
                        const uint32_t tb = (__float_as_uint(e.t) & ~7u) | (e.inside ? 1u : 0u) | ((uint32_t)e.edge << 1);
                        const int slot = vbase + (int)__builtin_amdgcn_mbcnt_hi((uint32_t)(cm >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)cm, 0u));
                        if (e.cand) {
                            sval[slot] = make_float4(z, e.rx, e.ry, __uint_as_float(tb));
                            smeta[slot] = (uint32_t)lane | ((uint32_t)(c0 + i) << 6);
                        }
                        if (lane == 0) shdr[n_emit] = make_uint4((uint32_t)cm, (uint32_t)(cm >> 32), (uint32_t)(c0 + i), (uint32_t)vbase);
                        ++n_emit;
                        vbase += nc;
                    }
                }
                if (!may_truncate) continue;
                if (__ballot(z < r[KT - 1]) == 0ull) continue;  // nobody's K-nearest set changes
                // sorted insert, dropping the largest: r'[i] = med3(r[i-1], r[i], z); r'[0] = min(r[0], z)
#pragma unroll
                for (int s_ = KT - 1; s_ >= 1; --s_) r[s_] = __builtin_amdgcn_fmed3f(r[s_ - 1], r[s_], z);
                r[0] = fminf(r[0], z);
            }
        }
        CHUNK_LOOP_END
        if (STREAM && stream_ok && lane == 0) lds.cfirst[n_chunks] = (uint32_t)vbase;
        TMARK(1)
        // cnt may have stopped early at >= K: then "all candidates" and "the K nearest" only coincide when cnt == K,
        // and pass 2 computes the right product in both cases
        const bool trunc = sorted ? (cnt >= K) : (cnt > K);
        float alpha = prod_all;
        float zt = 3.0e38f;  // depth threshold (K-th smallest)
        int r_ties = 0;
        int tie_cut = -1;    // list position of the last face kept among those exactly at the threshold (stream form)
        if (__ballot(trunc) != 0ull) {
            zt = kth_smallest<KT, EXACT>(r, K);
#pragma unroll
            for (int i = 0; i < KT; ++i) r_ties += ((EXACT || i < K) && r[i] == zt) ? 1 : 0;
            // ------------- pass 2: product over the K nearest for truncated pixels ---------------
            float prod = 1.0f;
            int ties = 0;
            // faces whose nearest vertex is beyond every unfinished truncated pixel's threshold cannot be among its K
            // nearest.  A pixel is finished once its product is below ALPHA_GRAD_EPS: 1 - alpha already rounds to 1.0f
            // and the pixel is below the gradient threshold, so no output can change any more.
            if (STREAM && stream_ok) {
                __syncthreads();  // the records were written by other lanes of this workgroup
                uint4 hn = make_uint4(0u, 0u, 0u, 0u);
                if (lane < n_emit) hn = shdr[lane];
                for (int e0 = 0; e0 < n_emit; e0 += WAVE) {
                    const bool live = trunc && prod > ALPHA_GRAD_EPS;
                    if (__ballot(live) == 0ull) break;
                    const int mm = min(WAVE, n_emit - e0);
                    const uint4 h = hn;  // headers of this batch; the next batch is requested right away
                    hn = make_uint4(0u, 0u, 0u, 0u);
                    if (e0 + WAVE + lane < n_emit) hn = shdr[e0 + WAVE + lane];
                    if (sorted) {
                        const uint32_t p0 = (uint32_t)__builtin_amdgcn_readfirstlane((int)h.z);
                        const float zlb = zlo + (float)(lds.list[p0] >> 16) * zstep;
                        if (zlb >= nextafterf(wave_max(live ? zt : -3.0e38f), 3.0e38f)) break;
                    }
                    // SGROUP records are requested before the first is consumed (the stream lives in L2 / MALL: hundreds
                    // of ns per access, and only two waves per SIMD to hide it).  Lanes >= mm hold all-zero headers.
                    for (int j0 = 0; j0 < mm; j0 += SGROUP) {
                        float4 v[SGROUP];
                        bool mine[SGROUP];
#pragma unroll
                        for (int u = 0; u < SGROUP; ++u) {
                            const uint32_t mlo = (uint32_t)__builtin_amdgcn_readlane((int)h.x, j0 + u), mhi = (uint32_t)__builtin_amdgcn_readlane((int)h.y, j0 + u);
                            const int first = __builtin_amdgcn_readlane((int)h.w, j0 + u);
                            mine[u] = ((mlo & lane_lo) | (mhi & lane_hi)) != 0u;
                            v[u] = sval[first + (mine[u] ? (int)__builtin_amdgcn_mbcnt_hi(mhi, __builtin_amdgcn_mbcnt_lo(mlo, 0u)) : 0)];
                        }
#pragma unroll
                        for (int u = 0; u < SGROUP; ++u) {
                            const bool c = mine[u] && trunc;
                            const bool tie = c && (v[u].x == zt) && (ties < r_ties);
                            const bool keep = c && ((v[u].x < zt) || tie);
                            ties += tie ? 1 : 0;
                            tie_cut = tie ? __builtin_amdgcn_readlane((int)h.z, j0 + u) : tie_cut;
                            const float dist = sq2(v[u].y, v[u].z);
                            const float sd = (__float_as_uint(v[u].w) & 1u) ? -dist : dist;
                            const float fac = 1.0f - face_prob(sd, a.inv_sigma);
                            prod *= keep ? fac : 1.0f;
                        }
                    }
                }
            } else
            CHUNK_LOOP_BEGIN(false, nextafterf(wave_max((trunc && prod > ALPHA_GRAD_EPS) ? zt : -3.0e38f), 3.0e38f))
            {
                if (__ballot(trunc && prod > ALPHA_GRAD_EPS) != 0ull) {
                    for (int i = 0; i < m; ++i) {
                        const FaceRec f = *reinterpret_cast<const FaceRec *>(lds.rec + i * FREC);
                        if (__ballot(trunc && !(px > f.xmax || px < f.xmin || py > f.ymax || py < f.ymin)) == 0ull) continue;
                        PairEval e;
                        eval_pair(f, px, py, dxp, dyp, a.blur, e);
                        if (__ballot(e.cand && trunc) == 0ull) continue;
                        const float pz = pair_depth(f, e);
                        const bool tie = e.cand && trunc && (pz == zt) && (ties < r_ties);
                        const bool keep = e.cand && trunc && ((pz < zt) || tie);
                        ties += tie ? 1 : 0;
                        const float fac = 1.0f - face_prob(e.sd, a.inv_sigma);
                        prod *= keep ? fac : 1.0f;
                    }
                }
            }
            CHUNK_LOOP_END
            if (trunc) alpha = prod;
        }

        TMARK(2)
        // ---------------- epilogue: silhouette value, loss, upstream gradient --------------------
        const float silv = 1.0f - alpha;
        float g = 0.f;
        if (MODE == MODE_FWD) {
            if (in_img) a.sil[pix] = silv;
        } else if (MODE == MODE_BWD) {
            if (in_img) g = a.grad_sil[pix];
        } else {
            float lsum = 0.f;
            if (in_img) {
                const float tg = a.target_u8 ? (float)a.target_u8[pix] : a.target[pix];
                const float diff = silv - tg;
                lsum = fabsf(diff) - fabsf(tg);  // loss_img starts at sum |0 - target|
                g = a.pix_scale[n] * (diff > 0.f ? 1.f : (diff < 0.f ? -1.f : 0.f));
                if (a.sil) a.sil[pix] = silv;
            }
            lsum = wave_sum(lsum);
            if (lane == 0 && lsum != 0.f) atomicAdd(&a.loss_img[n], lsum);
        }
        if (MODE == MODE_FWD) continue;

        TMARK(3)
        // ---------------- pass 3: gradients ------------------------------------------------------
        // d sil / d dist_k = -alpha p_k / sigma   (alpha = prod_j (1 - p_j); exact also when 1 - p_k == 0)
        const float coef = -g * alpha * a.inv_sigma;
        const bool active = in_img && (g != 0.f) && (alpha > ALPHA_GRAD_EPS);
        if (__ballot(active) == 0ull) continue;
        float *dn = a.d_ndc + (size_t)n * a.V * 2;
        const bool any_trunc = __ballot(trunc && active) != 0ull;
        int ties = 0;
        const float cut3 = wave_max(active ? (trunc ? zt : 3.0e38f) : -3.0e38f);
        if (STREAM && stream_ok) {
            // Dense walk: lane = record, not pixel.  Every lane fetches the state of the pixel its record belongs to from
            // LDS; a record at the threshold depth is kept iff its face comes no later than the pixel's last kept tie.
            lds.pcoef[lane] = active ? coef : 0.f;
            lds.pzt[lane] = trunc ? zt : __builtin_inff();
            lds.ptie[lane] = tie_cut;
            __syncthreads();  // also orders pass 1's record stores before the loads below
            for (int ch = 0; ch < n_chunks; ++ch) {
                const int c0 = ch * FCHUNK;
                if (sorted && zlo + (float)(lds.list[c0] >> 16) * zstep >= nextafterf(cut3, 3.0e38f)) break;
                const int i_beg = (int)lds.cfirst[ch], i_end = (int)lds.cfirst[ch + 1];
                if (i_beg == i_end) continue;
                for (int i_ = lane; i_ < FCHUNK * 6; i_ += WAVE) lds.gacc[i_] = 0.f;
                __syncthreads();
                for (int g0 = i_beg; g0 < i_end; g0 += DGROUP * WAVE) {
                    float4 v[DGROUP];
                    uint32_t mt[DGROUP];
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const int idx = min(g0 + u * WAVE + lane, i_end - 1);  // clamped: the tail repeats the last record
                        v[u] = sval[idx];
                        mt[u] = smeta[idx];
                    }
#pragma unroll
                    for (int u = 0; u < DGROUP; ++u) {
                        const bool valid = g0 + u * WAVE + lane < i_end;
                        const int pxl = (int)(mt[u] & 63u), pos = (int)(mt[u] >> 6);
                        const float pc = lds.pcoef[pxl], pz = lds.pzt[pxl];
                        const int pt = lds.ptie[pxl];
                        const uint32_t tb = __float_as_uint(v[u].w);
                        const bool inside = (tb & 1u) != 0u;
                        const float dist = sq2(v[u].y, v[u].z);
                        float gd = pc * face_prob(inside ? -dist : dist, a.inv_sigma);
                        gd = inside ? -gd : gd;
                        const bool keep = valid && (gd != 0.f) && ((v[u].x < pz) || (v[u].x == pz && pos <= pt));
                        const float t = __uint_as_float(tb & ~7u);
                        const int edge = (int)((tb >> 1) & 3u);
                        const int ia = edge == 2 ? 2 : 0, ib = edge == 0 ? 2 : 4;
                        const float ex = 2.0f * v[u].y * gd, ey = 2.0f * v[u].z * gd;
                        if (keep) {
                            float *acc = lds.gacc + (pos & 63) * 6;
                            atomicAdd(acc + ia, (1.0f - t) * ex);
                            atomicAdd(acc + ia + 1, (1.0f - t) * ey);
                            atomicAdd(acc + ib, t * ex);
                            atomicAdd(acc + ib + 1, t * ey);
                        }
                    }
                }
                flush_gacc(a, lds, dn, ch, list_total, id_mask, lane);
            }
        } else
        CHUNK_LOOP_BEGIN(true, nextafterf(cut3, 3.0e38f))
        {
            for (int i = 0; i < m; ++i) {
                const FaceRec f = *reinterpret_cast<const FaceRec *>(lds.rec + i * FREC);
                if (__ballot(active && !(px > f.xmax || px < f.xmin || py > f.ymax || py < f.ymin)) == 0ull) continue;
                PairEval e;
                eval_pair(f, px, py, dxp, dyp, a.blur, e);
                bool keep = e.cand && active;
                if (__ballot(keep) == 0ull) continue;
                if (any_trunc) {
                    const float pz = pair_depth(f, e);
                    const bool tie = keep && trunc && (pz == zt) && (ties < r_ties);
                    ties += tie ? 1 : 0;
                    keep = keep && (!trunc || (pz < zt) || tie);
                }
                float gd = coef * face_prob(e.sd, a.inv_sigma);  // d L / d (signed dist)
                gd = e.inside ? -gd : gd;                          // d L / d (unsigned squared distance)
                keep = keep && (gd != 0.f);
                if (__ballot(keep) == 0ull) continue;
                const float t = e.t;
                const int ia = e.edge == 2 ? 2 : 0, ib = e.edge == 0 ? 2 : 4;  // accumulator slots of the edge's end points
                const float ex = 2.0f * e.rx * gd, ey = 2.0f * e.ry * gd;
                if (keep) {
                    float *acc = lds.gacc + i * 6;
                    atomicAdd(acc + ia, (1.0f - t) * ex);
                    atomicAdd(acc + ia + 1, (1.0f - t) * ey);
                    atomicAdd(acc + ib, t * ex);
                    atomicAdd(acc + ib + 1, t * ey);
                }
            }
            __syncthreads();
            // flush: lane = staged face, one global atomic per touched vertex component
            if (lane < m) {
                const FaceRec &f = *reinterpret_cast<const FaceRec *>(lds.rec + lane * FREC);
                const float *acc = lds.gacc + lane * 6;
                if (acc[0] != 0.f) atomicAdd(&dn[2 * f.i0], acc[0]);
                if (acc[1] != 0.f) atomicAdd(&dn[2 * f.i0 + 1], acc[1]);
                if (acc[2] != 0.f) atomicAdd(&dn[2 * f.i1], acc[2]);
                if (acc[3] != 0.f) atomicAdd(&dn[2 * f.i1 + 1], acc[3]);
                if (acc[4] != 0.f) atomicAdd(&dn[2 * f.i2], acc[4]);
                if (acc[5] != 0.f) atomicAdd(&dn[2 * f.i2 + 1], acc[5]);
            }
        }
        CHUNK_LOOP_END
        TMARK(4)
    }
#ifdef DBG_TIMERS
    if (a.dbg && lane == 0)
        for (int k = 0; k < 5; ++k) atomicAdd(&a.dbg[k], tph[k]);
#endif
}

// ---------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------
static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }

static int tile_grid(int N, int tiles_x) {
    const long long max_items = (long long)N * tiles_x * tiles_x;
    const long long resident = 256LL * 8;  // 256 CUs x 8 single-wave workgroups (2 waves per SIMD)
    return (int)(max_items < resident ? max_items : resident);
}

// per resident workgroup: LIST_CAP stream headers + VAL_CAP pair records
static inline size_t stream_bytes(int grid) {
    return (size_t)grid * ((size_t)LIST_CAP * sizeof(uint4) + (size_t)VAL_CAP * (sizeof(float4) + sizeof(uint32_t)));
}

extern "C" size_t smil_raster_workspace_bytes(const SmilModel *m, int32_t N, int32_t S) {
    if (!m || N <= 0 || S <= 0) return 0;
    const size_t tiles = (size_t)ceil_div(S, TILE) * ceil_div(S, TILE);
    return 2 * align256((size_t)N * m->F * sizeof(uint32_t)) + 256 + align256((size_t)N * tiles * sizeof(uint32_t)) +
           256 + stream_bytes(tile_grid(N, ceil_div(S, TILE)));
}

static int raster_common(const SmilModel *m, const float *verts_ndc, int N, int S, const SmilRasterSettings *rs,
                         void *workspace, hipStream_t stream, RasterArgs &a) {
    SMIL_REQUIRE(m && verts_ndc && rs && workspace, "raster: null argument");
    SMIL_REQUIRE(N > 0 && S > 0 && S <= TILE * 256, "raster: bad sizes N=%d S=%d", N, S);
    SMIL_REQUIRE(rs->faces_per_pixel > 0 && rs->faces_per_pixel <= SMIL_MAX_FACES_PER_PIXEL,
                 "raster: faces_per_pixel=%d outside 1..%d", rs->faces_per_pixel, SMIL_MAX_FACES_PER_PIXEL);
    SMIL_REQUIRE(rs->sigma > 0.f && rs->blur_radius >= 0.f, "raster: bad blend settings");
    const int tiles_x = ceil_div(S, TILE);
    SMIL_REQUIRE((double)N * tiles_x * tiles_x < 4294967295.0, "raster: N * tiles exceeds the 32-bit work-item code");
    char *ws = (char *)workspace;
    uint32_t *tbox = (uint32_t *)ws;
    ws += align256((size_t)N * m->F * sizeof(uint32_t));
    RasterCounters *ctr = (RasterCounters *)ws;  // (the probe tool reads the counters right behind the tile boxes)
    float *fzmin = (float *)(ws + 256 + align256((size_t)N * tiles_x * tiles_x * sizeof(uint32_t)));
    ws += 256;
    uint32_t *items = (uint32_t *)ws;
    SMIL_HIP(hipMemsetAsync(ctr, 0, sizeof(RasterCounters), stream));
    const float sqrt_blur = sqrtf(rs->blur_radius);
    const int n_words = (tiles_x * tiles_x + 31) / 32;
    hipLaunchKernelGGL(k_raster_setup, dim3(N), dim3(256), (size_t)(n_words + 256) * sizeof(uint32_t), stream, verts_ndc,
                       m->faces, tbox, fzmin, items, ctr, m->V, m->F, S, tiles_x, sqrt_blur);
    SMIL_LAUNCH_CHECK();
    a.verts_ndc = verts_ndc; a.faces = m->faces; a.tbox = tbox; a.fzmin = fzmin; a.items = items; a.ctr = ctr;
    a.N = N; a.V = m->V; a.F = m->F; a.S = S; a.tiles_x = tiles_x; a.K = rs->faces_per_pixel;
    a.blur = rs->blur_radius; a.sqrt_blur = sqrt_blur; a.inv_sigma = 1.0f / rs->sigma;
    {
        char *sp = (char *)fzmin + align256((size_t)N * m->F * sizeof(float));
        const int grid = tile_grid(N, tiles_x);
        a.shdr = (uint4 *)sp;
        a.sval = (float4 *)(sp + (size_t)grid * LIST_CAP * sizeof(uint4));
        a.smeta = (uint32_t *)((char *)a.sval + (size_t)grid * VAL_CAP * sizeof(float4));
    }
    a.dbg = nullptr;
#ifdef DBG_TIMERS
    {
        static unsigned long long *dbg_dev = nullptr;
        if (!dbg_dev) { (void)hipMalloc(&dbg_dev, 64); (void)hipMemset(dbg_dev, 0, 64); }
        unsigned long long h[8];
        (void)hipMemcpy(h, dbg_dev, 64, hipMemcpyDeviceToHost);  // totals of the launches so far
        fprintf(stderr, "[dbg timers] setup+list %.3e  p1 %.3e  p2 %.3e  epi %.3e  p3 %.3e cycles\n", (double)h[0], (double)h[1],
                (double)h[2], (double)h[3], (double)h[4]);
        (void)hipMemset(dbg_dev, 0, 64);
        a.dbg = dbg_dev;
    }
#endif
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
    const dim3 grid(tile_grid(N, a.tiles_x)), block(64);
    constexpr bool ST = MODE != MODE_FWD;  // backward passes read pass 1's pair stream instead of re-evaluating
    if (a.K == 100) hipLaunchKernelGGL((k_raster_tiles<MODE, 100, true, ST>), grid, block, 0, stream, a);  // the reference's K
    else if (a.K <= 16) hipLaunchKernelGGL((k_raster_tiles<MODE, 16, false, false>), grid, block, 0, stream, a);
    else hipLaunchKernelGGL((k_raster_tiles<MODE, SMIL_MAX_FACES_PER_PIXEL, false, false>), grid, block, 0, stream, a);
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
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_silhouette_backward(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                                        const SmilRasterSettings *rs, const float *grad_sil, float *d_ndc,
                                        void *workspace, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterArgs a;
    int rc = raster_common(m, verts_ndc, N, S, rs, workspace, stream, a);
    if (rc) return rc;
    SMIL_REQUIRE(grad_sil && d_ndc, "smil_silhouette_backward: null argument");
    SMIL_HIP(hipMemsetAsync(d_ndc, 0, (size_t)N * m->V * 2 * sizeof(float), stream));
    a.grad_sil = grad_sil; a.d_ndc = d_ndc;
    PROF_BEGIN(stream);
    launch_tiles<MODE_BWD>(a, N, stream);
    PROF_END(stream);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_silhouette_l1_fused(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                                        const SmilRasterSettings *rs, const void *target, int32_t target_is_u8,
                                        const float *target_sum, const float *pix_scale, float *loss_img, float *d_ndc,
                                        float *sil_out, void *workspace, void *stream_) {
    hipStream_t stream = (hipStream_t)stream_;
    RasterArgs a;
    int rc = raster_common(m, verts_ndc, N, S, rs, workspace, stream, a);
    if (rc) return rc;
    SMIL_REQUIRE(target && target_sum && pix_scale && loss_img && d_ndc, "smil_silhouette_l1_fused: null argument");
    SMIL_HIP(hipMemsetAsync(d_ndc, 0, (size_t)N * m->V * 2 * sizeof(float), stream));
    SMIL_HIP(hipMemcpyAsync(loss_img, target_sum, (size_t)N * sizeof(float), hipMemcpyDeviceToDevice, stream));
    if (sil_out) SMIL_HIP(hipMemsetAsync(sil_out, 0, (size_t)N * S * S * sizeof(float), stream));
    if (target_is_u8) a.target_u8 = (const uint8_t *)target; else a.target = (const float *)target;
    a.pix_scale = pix_scale; a.loss_img = loss_img; a.d_ndc = d_ndc; a.sil = sil_out;
    PROF_BEGIN(stream);
    launch_tiles<MODE_FUSED>(a, N, stream);
    PROF_END(stream);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}
