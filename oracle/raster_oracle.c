/*
 * ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 *
 * CPU restatement of the soft-silhouette rasteriser the reference calls through
 * smal_fitter/p3d_renderer.py:41-52,142-146 (MeshRasterizer(bin_size=0,
 * faces_per_pixel=100, blur_radius=log(1/1e-4-1)*1e-4) + SoftSilhouetteShader).
 * The arithmetic lives in the third-party dependency pytorch3d (pinned 0.7.8,
 * reference environment.yml:35), which is NOT vendored under /root/reference and not
 * installed here, so this file restates its published algorithm:
 *   csrc/rasterize_meshes/rasterize_meshes.cu   CheckPixelInsideFace,
 *                                               RasterizeMeshesNaiveCudaKernel,
 *                                               RasterizeMeshesBackwardCudaKernel
 *   csrc/utils/geometry_utils.cuh               EdgeFunctionForward, BarycentricCoordsForward,
 *                                               BarycentricPerspectiveCorrectionForward,
 *                                               BarycentricClipForward, PointLineDistance*,
 *                                               PointTriangleDistance*
 *   renderer/blending.py                        sigmoid_alpha_blend
 * PARITY UNPINNED for this file: the reference holds no golden silhouette; it is pinned
 * only by analytic known-answer cases in tests/test_oracle_raster.py.
 *
 * Layout: verts_ndc (N,V,3) fp32 = (x_ndc, y_ndc, z_view) per image; faces (F,3) int32
 * shared by all images; output sil (N,S,S) fp32, row = image y (top row first).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define K_EPS 1e-8f
/* MeshRasterizer passes z_clip_value = znear / 2 (renderer/mesh/rasterizer.py; the reference sets znear = 1e-3,
 * p3d_renderer.py:24,36-38) to clip_faces(): a face whose three vertices are all nearer than that is removed before the
 * kernel runs.  Faces that straddle the value are split there: oracle/render_ref.py::clip_faces_np restates that step and
 * hands this file the clipped mesh of such an image (new vertices on the plane, the front parts as extra faces); called
 * directly, this file rasterises a straddling face whole unless the kernel rule zmin < 1e-8 drops it. */
static float g_z_clip = 5e-4f;
void oracle_set_z_clip(float z) { g_z_clip = z; }

typedef struct {
    float z;
    int f;
    float dist; /* signed squared distance */
    int inside; /* unclipped perspective-corrected barycentrics all > 0 */
} Frag;

static inline float edge_fn(float px, float py, float ax, float ay, float bx, float by) {
    return (px - ax) * (by - ay) - (py - ay) * (bx - ax);
}

static inline float seg_d2(float px, float py, float ax, float ay, float bx, float by) {
    const float bax = bx - ax, bay = by - ay;
    const float l2 = bax * bax + bay * bay;
    if (l2 <= K_EPS) {
        const float dx = px - bx, dy = py - by;
        return dx * dx + dy * dy;
    }
    float t = (bax * (px - ax) + bay * (py - ay)) / l2;
    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
    const float qx = ax + t * bax, qy = ay + t * bay;
    const float dx = qx - px, dy = qy - py;
    return dx * dx + dy * dy;
}

/* Evaluate one (pixel, face). Returns 1 and fills (pz, sdist) when the face is a candidate. */
static inline int eval_face(const float *v0, const float *v1, const float *v2, float px, float py,
                            float blur, float sqrt_blur, float *pz_out, float *sd_out, int *inside_out) {
    const float zmax = fmaxf(fmaxf(v0[2], v1[2]), v2[2]);
    const float zmin = fminf(fminf(v0[2], v1[2]), v2[2]);
    const float xmin = fminf(fminf(v0[0], v1[0]), v2[0]) - sqrt_blur;
    const float xmax = fmaxf(fmaxf(v0[0], v1[0]), v2[0]) + sqrt_blur;
    const float ymin = fminf(fminf(v0[1], v1[1]), v2[1]) - sqrt_blur;
    const float ymax = fmaxf(fmaxf(v0[1], v1[1]), v2[1]) + sqrt_blur;
    const int outside = (px > xmax) || (px < xmin) || (py > ymax) || (py < ymin) || (zmin < K_EPS);
    const float face_area = edge_fn(v0[0], v0[1], v1[0], v1[1], v2[0], v2[1]);
    const int zero_area = (face_area <= K_EPS) && (face_area >= -K_EPS);
    if (zmax < 0.f || zmax < g_z_clip || outside || zero_area) return 0;

    const float area = edge_fn(v2[0], v2[1], v0[0], v0[1], v1[0], v1[1]) + K_EPS;
    const float b0 = edge_fn(px, py, v1[0], v1[1], v2[0], v2[1]) / area;
    const float b1 = edge_fn(px, py, v2[0], v2[1], v0[0], v0[1]) / area;
    const float b2 = edge_fn(px, py, v0[0], v0[1], v1[0], v1[1]) / area;
    /* perspective correction */
    const float w0 = b0 * v1[2] * v2[2];
    const float w1 = b1 * v0[2] * v2[2];
    const float w2 = b2 * v0[2] * v1[2];
    const float den = fmaxf(w0 + w1 + w2, K_EPS);
    const float p0 = w0 / den, p1 = w1 / den, p2 = w2 / den;
    /* clip + renormalise */
    float c0 = fmaxf(p0, 0.f), c1 = fmaxf(p1, 0.f), c2 = fmaxf(p2, 0.f);
    const float cs = fmaxf(c0 + c1 + c2, 1e-5f);
    c0 /= cs; c1 /= cs; c2 /= cs;
    const float pz = c0 * v0[2] + c1 * v1[2] + c2 * v2[2];
    if (pz < 0.f) return 0;
    const float d01 = seg_d2(px, py, v0[0], v0[1], v1[0], v1[1]);
    const float d02 = seg_d2(px, py, v0[0], v0[1], v2[0], v2[1]);
    const float d12 = seg_d2(px, py, v1[0], v1[1], v2[0], v2[1]);
    const float dist = fminf(fminf(d01, d02), d12);
    const int inside = (p0 > 0.f) && (p1 > 0.f) && (p2 > 0.f);
    if (!inside && dist >= blur) return 0;
    *pz_out = pz;
    *sd_out = inside ? -dist : dist;
    *inside_out = inside;
    return 1;
}

static inline float pix_to_ndc(int i, int S) {
    /* PixToNonSquareNdc for a square image: -1 + (2 i + 1)/S */
    return -1.0f + (2.0f * (float)i + 1.0f) / (float)S;
}

static int frag_less(const void *a, const void *b);
static int g_select_mode;

/* Sequential K-nearest queue exactly as the naive kernel keeps it: unsorted array, tracked max,
 * strict '<' replacement. Returns number kept; q must hold K entries. */
static int gather_pixel(const float *vn, const int32_t *faces, int F, float px, float py, float blur,
                        float sqrt_blur, int K, Frag *q, int *n_candidates) {
    int qsize = 0, qmax_idx = -1, ncand = 0;
    float qmax_z = -1000.f;
    if (g_select_mode == 1) {
        /* analysis variant: gather everything, keep the K smallest (z, face) keys */
        int cap = 256;
        Frag *all = (Frag *)malloc(sizeof(Frag) * (size_t)cap);
        for (int f = 0; f < F; ++f) {
            const float *v0 = vn + 3 * (size_t)faces[3 * f + 0];
            const float *v1 = vn + 3 * (size_t)faces[3 * f + 1];
            const float *v2 = vn + 3 * (size_t)faces[3 * f + 2];
            float pz, sd;
            int ins;
            if (!eval_face(v0, v1, v2, px, py, blur, sqrt_blur, &pz, &sd, &ins)) continue;
            if (ncand == cap) { cap *= 2; all = (Frag *)realloc(all, sizeof(Frag) * (size_t)cap); }
            all[ncand].z = pz; all[ncand].f = f; all[ncand].dist = sd; all[ncand].inside = ins;
            ++ncand;
        }
        qsort(all, (size_t)ncand, sizeof(Frag), frag_less);
        qsize = ncand < K ? ncand : K;
        memcpy(q, all, sizeof(Frag) * (size_t)qsize);
        free(all);
        if (n_candidates) *n_candidates = ncand;
        return qsize;
    }
    for (int f = 0; f < F; ++f) {
        const float *v0 = vn + 3 * (size_t)faces[3 * f + 0];
        const float *v1 = vn + 3 * (size_t)faces[3 * f + 1];
        const float *v2 = vn + 3 * (size_t)faces[3 * f + 2];
        float pz, sd;
        int ins;
        if (!eval_face(v0, v1, v2, px, py, blur, sqrt_blur, &pz, &sd, &ins)) continue;
        ++ncand;
        if (qsize < K) {
            q[qsize].z = pz; q[qsize].f = f; q[qsize].dist = sd; q[qsize].inside = ins;
            if (pz > qmax_z) { qmax_z = pz; qmax_idx = qsize; }
            ++qsize;
        } else if (pz < qmax_z) {
            q[qmax_idx].z = pz; q[qmax_idx].f = f; q[qmax_idx].dist = sd; q[qmax_idx].inside = ins;
            qmax_z = pz;
            for (int i = 0; i < K; ++i)
                if (q[i].z > qmax_z) { qmax_z = q[i].z; qmax_idx = i; }
        }
    }
    if (n_candidates) *n_candidates = ncand;
    return qsize;
}

static int g_select_mode; /* 0 = faithful sequential queue (the reference); 1 = K smallest by (z, face): the documented rule of the HIP rasteriser, used by tests to check that rule exactly */
void oracle_set_select_mode(int m) { g_select_mode = m; }

static int frag_less(const void *a, const void *b) {
    const Frag *x = (const Frag *)a, *y = (const Frag *)b;
    if (x->z < y->z) return -1;
    if (x->z > y->z) return 1;
    return (x->f > y->f) - (x->f < y->f);
}

static inline float sigmoidf(float x) { return 1.0f / (1.0f + expf(-x)); }

/* Forward: sil[n,y,x] = 1 - prod_k (1 - sigmoid(-dist_k / sigma)).
 * Optional outputs: n_cand (N,S,S) int32 candidates before truncation; frag_face/frag_dist/frag_z
 * (N,S,S,K) fragment dump (face = -1 padding) sorted by (z, face) like the reference output. */
int oracle_silhouette_forward(const float *verts_ndc, const int32_t *faces, int N, int V, int F, int S,
                              float blur, float sigma, int K, float *sil, int32_t *n_cand,
                              int32_t *frag_face, float *frag_dist, float *frag_z) {
    if (K <= 0 || K > 4096) return -1;
    const float sqrt_blur = sqrtf(blur);
#pragma omp parallel
    {
        Frag *q = (Frag *)malloc(sizeof(Frag) * (size_t)K);
#pragma omp for collapse(2) schedule(dynamic, 4)
        for (int n = 0; n < N; ++n) {
            for (int yo = 0; yo < S; ++yo) {
                const float *vn = verts_ndc + (size_t)n * V * 3;
                const int yi = S - 1 - yo;
                const float yf = pix_to_ndc(yi, S);
                for (int xo = 0; xo < S; ++xo) {
                    const int xi = S - 1 - xo;
                    const float xf = pix_to_ndc(xi, S);
                    int nc = 0;
                    const int kept = gather_pixel(vn, faces, F, xf, yf, blur, sqrt_blur, K, q, &nc);
                    qsort(q, (size_t)kept, sizeof(Frag), frag_less);
                    float alpha = 1.0f;
                    for (int k = 0; k < kept; ++k) alpha *= (1.0f - sigmoidf(-q[k].dist / sigma));
                    const size_t pix = ((size_t)n * S + yo) * S + xo;
                    sil[pix] = 1.0f - alpha;
                    if (n_cand) n_cand[pix] = nc;
                    if (frag_face) {
                        for (int k = 0; k < K; ++k) {
                            frag_face[pix * K + k] = k < kept ? q[k].f : -1;
                            frag_dist[pix * K + k] = k < kept ? q[k].dist : -1.f;
                            frag_z[pix * K + k] = k < kept ? q[k].z : -1.f;
                        }
                    }
                }
            }
        }
        free(q);
    }
    return 0;
}

/* Backward: grad_sil (N,S,S) -> grad_verts_ndc (N,V,3) (z component stays 0: only the signed
 * distances carry gradient, SoftSilhouetteShader ignores zbuf/bary).  Accumulation is done in
 * double, per image, in pixel order (the reference uses float atomics, order undefined). */
int oracle_silhouette_backward(const float *verts_ndc, const int32_t *faces, int N, int V, int F, int S,
                               float blur, float sigma, int K, const float *grad_sil, float *grad_verts) {
    if (K <= 0 || K > 4096) return -1;
    const float sqrt_blur = sqrtf(blur);
    memset(grad_verts, 0, sizeof(float) * (size_t)N * V * 3);
#pragma omp parallel for schedule(dynamic, 1)
    for (int n = 0; n < N; ++n) {
        Frag *q = (Frag *)malloc(sizeof(Frag) * (size_t)K);
        float *pk = (float *)malloc(sizeof(float) * (size_t)K);
        double *acc = (double *)calloc((size_t)V * 2, sizeof(double));
        const float *vn = verts_ndc + (size_t)n * V * 3;
        for (int yo = 0; yo < S; ++yo) {
            const float yf = pix_to_ndc(S - 1 - yo, S);
            for (int xo = 0; xo < S; ++xo) {
                const float g = grad_sil[((size_t)n * S + yo) * S + xo];
                if (g == 0.f) continue;
                const float xf = pix_to_ndc(S - 1 - xo, S);
                const int kept = gather_pixel(vn, faces, F, xf, yf, blur, sqrt_blur, K, q, NULL);
                if (!kept) continue;
                qsort(q, (size_t)kept, sizeof(Frag), frag_less);
                for (int k = 0; k < kept; ++k) pk[k] = sigmoidf(-q[k].dist / sigma);
                for (int k = 0; k < kept; ++k) {
                    /* d alpha / d (1-p_k) = prod_{j != k} (1 - p_j)  (exact product, no division) */
                    float others = 1.0f;
                    for (int j = 0; j < kept; ++j)
                        if (j != k) others *= (1.0f - pk[j]);
                    /* sil = 1 - alpha ; d(1-p)/d dist = +p(1-p)/sigma */
                    const float gdist = -g * others * pk[k] * (1.0f - pk[k]) / sigma;
                    if (gdist == 0.f) continue;
                    const int f = q[k].f;
                    const int i0 = faces[3 * f], i1 = faces[3 * f + 1], i2 = faces[3 * f + 2];
                    const float *v0 = vn + 3 * (size_t)i0, *v1 = vn + 3 * (size_t)i1, *v2 = vn + 3 * (size_t)i2;
                    const float gd = q[k].inside ? -gdist : gdist; /* dist = inside ? -d : d */
                    const float d01 = seg_d2(xf, yf, v0[0], v0[1], v1[0], v1[1]);
                    const float d02 = seg_d2(xf, yf, v0[0], v0[1], v2[0], v2[1]);
                    const float d12 = seg_d2(xf, yf, v1[0], v1[1], v2[0], v2[1]);
                    const float *a, *b;
                    int ia, ib;
                    if (d01 <= d02 && d01 <= d12) { a = v0; b = v1; ia = i0; ib = i1; }
                    else if (d02 <= d01 && d02 <= d12) { a = v0; b = v2; ia = i0; ib = i2; }
                    else { a = v1; b = v2; ia = i1; ib = i2; }
                    const float bax = b[0] - a[0], bay = b[1] - a[1];
                    float t = (bax * (xf - a[0]) + bay * (yf - a[1])) / (bax * bax + bay * bay);
                    t = t < 0.f ? 0.f : (t > 1.f ? 1.f : t);
                    if (!(t == t)) t = 0.f; /* saturate(NaN) = 0 */
                    const float qx = (1.0f - t) * a[0] + t * b[0], qy = (1.0f - t) * a[1] + t * b[1];
                    const float ex = 2.0f * (qx - xf), ey = 2.0f * (qy - yf);
                    acc[2 * ia + 0] += (double)(gd * (1.0f - t) * ex);
                    acc[2 * ia + 1] += (double)(gd * (1.0f - t) * ey);
                    acc[2 * ib + 0] += (double)(gd * t * ex);
                    acc[2 * ib + 1] += (double)(gd * t * ey);
                }
            }
        }
        float *gv = grad_verts + (size_t)n * V * 3;
        for (int v = 0; v < V; ++v) {
            gv[3 * v + 0] = (float)acc[2 * v + 0];
            gv[3 * v + 1] = (float)acc[2 * v + 1];
        }
        free(q); free(pk); free(acc);
    }
    return 0;
}

int oracle_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
