// FoV-perspective cameras: world -> view -> NDC -> screen, forward and backward.
//
// Replaces (reference): smal_fitter/p3d_renderer.py:34-38,112-120 (FoVPerspectiveCameras set-up),
// :137 (transform_points_screen(...)[..., [1,0]]) and the vertex transform pytorch3d's MeshRasterizer
// applies before rasterising (:145).  Arithmetic restated from pytorch3d 0.7.x (un-vendored):
//   X_view = X_world R + T ;  x_ndc = X_view.x K00 / z , y_ndc = X_view.y K11 / z , z = X_view.z
//   K00 = 2 znear / (2 aspect tan(fov/2) znear) , K11 = 2 znear / (2 tan(fov/2) znear)
//   x_s = S/2 - (S/2) x_ndc ; y_s likewise ; the reference returns (y_s, x_s).
#include "common.h"

// One or two point sets (e.g. the vertices -> NDC for the rasteriser and the joints -> screen for the 2-D loss) per launch.
struct ProjectSet {
    const float *pts;  // (frames,P,3)
    float *ndc, *yx;   // (N,P,3) / (N,P,2), either may be NULL
    int P, blocks;     // points per frame; thread blocks (of 256 points) this set occupies in grid.x
};

__device__ __forceinline__ void project_point(const SmilCameras &c, const CamParams &cp, const ProjectSet &s, int b, int n, int p) {
    const float *X = s.pts + ((size_t)b * s.P + p) * 3;
    const float x = X[0], y = X[1], z = X[2];
    const float vx = x * cp.R[0] + y * cp.R[3] + z * cp.R[6] + cp.T[0];
    const float vy = x * cp.R[1] + y * cp.R[4] + z * cp.R[7] + cp.T[1];
    const float vz = x * cp.R[2] + y * cp.R[5] + z * cp.R[8] + cp.T[2];
    const float xn = vx * cp.k00 / vz;
    const float yn = vy * cp.k11 / vz;
    const size_t o = (size_t)n * s.P + p;
    if (s.ndc) { s.ndc[o * 3] = xn; s.ndc[o * 3 + 1] = yn; s.ndc[o * 3 + 2] = vz; }
    if (s.yx) {
        const float h = 0.5f * (float)c.S;
        s.yx[o * 2] = h - h * yn;
        s.yx[o * 2 + 1] = h - h * xn;
    }
}

// grid (N, blocks of set 0 + blocks of set 1)
__global__ void __launch_bounds__(256) k_project(SmilCameras c, ProjectSet s0, ProjectSet s1) {
    const int n = blockIdx.x;  // image on x: gridDim.y stops at 65 535
    const int b = n / c.views;
    const CamParams cp = load_camera(c, n);
    const bool second = (int)blockIdx.y >= s0.blocks;
    const ProjectSet &s = second ? s1 : s0;
    const int p = ((int)blockIdx.y - (second ? s0.blocks : 0)) * blockDim.x + threadIdx.x;
    if (p < s.P) project_point(c, cp, s, b, n, p);
}

static int check_cameras(const SmilCameras *cam, const char *who) {
    SMIL_REQUIRE(cam, "%s: null camera argument", who);
    SMIL_REQUIRE(cam->N > 0 && cam->views > 0 && cam->N % cam->views == 0 && cam->S > 0, "%s: bad sizes N=%d views=%d S=%d", who, cam->N,
                 cam->views, cam->S);
    SMIL_REQUIRE(cam->R && cam->T && cam->fov && cam->nR > 0 && cam->nT > 0 && cam->nFov > 0, "%s: camera tables missing", who);
    SMIL_REQUIRE(!cam->aspect || cam->nAspect > 0, "%s: aspect table empty", who);
    return SMIL_OK;
}

extern "C" int smil_project(const SmilCameras *cam, const float *pts, int32_t P, float *ndc, float *yx, void *stream) {
    if (int rc = check_cameras(cam, "smil_project")) return rc;
    SMIL_REQUIRE(pts && P > 0, "smil_project: bad points argument (P=%d)", P);
    const ProjectSet s0 = {pts, ndc, yx, P, ceil_div(P, 256)}, none = {nullptr, nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(k_project, dim3(cam->N, s0.blocks), dim3(256), 0, (hipStream_t)stream, *cam, s0, none);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_project2(const SmilCameras *cam, const float *pts_a, int32_t Pa, float *ndc_a, float *yx_a, const float *pts_b,
                             int32_t Pb, float *ndc_b, float *yx_b, void *stream) {
    if (int rc = check_cameras(cam, "smil_project2")) return rc;
    SMIL_REQUIRE(pts_a && Pa > 0 && pts_b && Pb > 0, "smil_project2: bad points arguments (Pa=%d Pb=%d)", Pa, Pb);
    const ProjectSet s0 = {pts_a, ndc_a, yx_a, Pa, ceil_div(Pa, 256)}, s1 = {pts_b, ndc_b, yx_b, Pb, ceil_div(Pb, 256)};
    hipLaunchKernelGGL(k_project, dim3(cam->N, s0.blocks + s1.blocks), dim3(256), 0, (hipStream_t)stream, *cam, s0, s1);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

struct ProjectBwdSet {
    const float *pts, *d_ndc, *d_yx;  // (frames,P,3), (N,P,2) or NULL, (N,P,2) or NULL
    float *d_pts;                     // (frames,P,3)
    int P, blocks, accumulate;
    // (N,) or NULL: rows of d_ndc the fused rasteriser left as packed fixed point (smil_silhouette_l1_fused, d_ndc_scale):
    // > 0: (x, y) of a point are two 32-bit fixed-point numbers in one 64-bit word, times this factor; 0: plain floats
    const float *d_ndc_scale;
};

// grid (frames, blocks of set 0 + blocks of set 1): each thread owns one world point and walks the views of its frame, so
// d_pts needs no atomics; the per-image fov term is block-reduced and added once per block.
__global__ void __launch_bounds__(256) k_project_bwd(SmilCameras c, ProjectBwdSet s0, ProjectBwdSet s1, float *__restrict__ d_fov_img) {
    __shared__ float red[16];
    const int b = blockIdx.x;
    const bool second = (int)blockIdx.y >= s0.blocks;
    const ProjectBwdSet &s = second ? s1 : s0;
    const int P = s.P;
    const int p = ((int)blockIdx.y - (second ? s0.blocks : 0)) * blockDim.x + threadIdx.x;
    const bool live = p < P;
    float x = 0.f, y = 0.f, z = 0.f;
    if (live) {
        const float *X = s.pts + ((size_t)b * P + p) * 3;
        x = X[0]; y = X[1]; z = X[2];
    }
    float gx = 0.f, gy = 0.f, gz = 0.f;
    for (int v = 0; v < c.views; ++v) {
        const int n = b * c.views + v;
        const CamParams cp = load_camera(c, n);
        float fsum = 0.f;
        if (live) {
            const float vx = x * cp.R[0] + y * cp.R[3] + z * cp.R[6] + cp.T[0];
            const float vy = x * cp.R[1] + y * cp.R[4] + z * cp.R[7] + cp.T[1];
            const float vz = x * cp.R[2] + y * cp.R[5] + z * cp.R[8] + cp.T[2];
            const float iz = 1.0f / vz;
            const float xn = vx * cp.k00 * iz, yn = vy * cp.k11 * iz;
            const size_t o = (size_t)n * P + p;
            float dxn = 0.f, dyn = 0.f;
            if (s.d_ndc) {
                const float2 raw = reinterpret_cast<const float2 *>(s.d_ndc)[o];
                const float sc = s.d_ndc_scale ? s.d_ndc_scale[n] : 0.f;
                if (sc != 0.f) {  // x * 2^32 + y in two's complement: a negative y borrowed one from the high word
                    const int qy = __float_as_int(raw.x), qx = __float_as_int(raw.y) - (qy >> 31);
                    dxn = sc > 0.f ? (float)qx * sc : 0.f;
                    dyn = sc > 0.f ? (float)qy * sc : 0.f;
                } else {
                    dxn = raw.x; dyn = raw.y;
                }
            }
            if (s.d_yx) {
                const float h = 0.5f * (float)c.S;
                dyn -= h * s.d_yx[o * 2];
                dxn -= h * s.d_yx[o * 2 + 1];
            }
            const float dvx = dxn * cp.k00 * iz, dvy = dyn * cp.k11 * iz;
            const float dvz = -(xn * dxn + yn * dyn) * iz;
            gx += cp.R[0] * dvx + cp.R[1] * dvy + cp.R[2] * dvz;
            gy += cp.R[3] * dvx + cp.R[4] * dvy + cp.R[5] * dvz;
            gz += cp.R[6] * dvx + cp.R[7] * dvy + cp.R[8] * dvz;
            fsum = dxn * xn + dyn * yn;
        }
        if (d_fov_img) {
            const float r = block_sum(fsum, red);
            if (threadIdx.x == 0 && r != 0.f) atomicAdd(&d_fov_img[n], r);
        }
    }
    if (live && s.d_pts) {
        float *o = s.d_pts + ((size_t)b * P + p) * 3;
        if (s.accumulate) { o[0] += gx; o[1] += gy; o[2] += gz; }
        else { o[0] = gx; o[1] = gy; o[2] = gz; }
    }
}

extern "C" int smil_project_backward(const SmilCameras *cam, const float *pts, int32_t P, const float *d_ndc,
                                     const float *d_yx, float *d_pts, float *d_fov_img, int32_t accumulate, const float *d_ndc_scale,
                                     void *stream) {
    SMIL_REQUIRE(cam && pts, "smil_project_backward: null argument");
    SMIL_REQUIRE(cam->N > 0 && cam->views > 0 && cam->N % cam->views == 0 && P > 0, "smil_project_backward: bad sizes");
    SMIL_REQUIRE(d_ndc || d_yx, "smil_project_backward: no upstream gradient");
    const ProjectBwdSet s0 = {pts, d_ndc, d_yx, d_pts, P, ceil_div(P, 256), accumulate, d_ndc_scale}, none = {nullptr, nullptr, nullptr, nullptr, 0, 0, 0, nullptr};
    hipLaunchKernelGGL(k_project_bwd, dim3(cam->N / cam->views, s0.blocks), dim3(256), 0, (hipStream_t)stream, *cam, s0, none, d_fov_img);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_project_backward2(const SmilCameras *cam, const float *pts_a, int32_t Pa, const float *d_ndc_a, const float *d_yx_a,
                                      float *d_pts_a, const float *pts_b, int32_t Pb, const float *d_ndc_b, const float *d_yx_b,
                                      float *d_pts_b, float *d_fov_img, const float *d_ndc_scale_a, void *stream) {
    SMIL_REQUIRE(cam && pts_a && pts_b && d_pts_a && d_pts_b, "smil_project_backward2: null argument");
    SMIL_REQUIRE(cam->N > 0 && cam->views > 0 && cam->N % cam->views == 0 && Pa > 0 && Pb > 0, "smil_project_backward2: bad sizes");
    SMIL_REQUIRE((d_ndc_a || d_yx_a) && (d_ndc_b || d_yx_b), "smil_project_backward2: no upstream gradient");
    const ProjectBwdSet s0 = {pts_a, d_ndc_a, d_yx_a, d_pts_a, Pa, ceil_div(Pa, 256), 0, d_ndc_scale_a},
                        s1 = {pts_b, d_ndc_b, d_yx_b, d_pts_b, Pb, ceil_div(Pb, 256), 0, nullptr};
    hipLaunchKernelGGL(k_project_bwd, dim3(cam->N / cam->views, s0.blocks + s1.blocks), dim3(256), 0, (hipStream_t)stream, *cam, s0, s1, d_fov_img);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// Depth gradients of cut edges' end points (SmilClipDepth, written by the rasteriser's k_clip_backward) carried through the camera into
// the world-space vertex gradient: z_view = x R[2] + y R[5] + z R[8] + T_z.  One workgroup per image; images without cut faces (all of
// them on the BASELINE configurations) read two words and leave.  Two views of a frame may touch the same vertex: float atomics.
__global__ void __launch_bounds__(64) k_clip_depth_bwd(SmilCameras c, SmilClipDepth cd, int V, float *__restrict__ d_verts) {
    if (cd.counter[0] == 0u) return;  // (nothing was cut anywhere in the batch)
    const int n = blockIdx.x;
    const uint32_t first = cd.range[2 * (size_t)n], cnt = cd.range[2 * (size_t)n + 1];
    if (cnt == 0u) return;
    const CamParams cp = load_camera(c, n);
    float *dv = d_verts + (size_t)(n / c.views) * V * 3;
    for (uint32_t e = threadIdx.x; e < cnt; e += blockDim.x) {
        const int v = cd.vertex[first + e];
        const float dz = cd.dz[first + e];
        if (v < 0 || v >= V || dz == 0.f) continue;
        atomicAdd(dv + 3 * v, dz * cp.R[2]); atomicAdd(dv + 3 * v + 1, dz * cp.R[5]); atomicAdd(dv + 3 * v + 2, dz * cp.R[8]);
    }
}

extern "C" int smil_clip_depth_backward(const SmilCameras *cam, const SmilClipDepth *cd, int32_t N, int32_t V, float *d_verts, void *stream) {
    SMIL_REQUIRE(cam && cd && d_verts, "smil_clip_depth_backward: null argument");
    SMIL_REQUIRE(cd->vertex && cd->dz && cd->range && cd->counter, "smil_clip_depth_backward: incomplete SmilClipDepth");
    SMIL_REQUIRE(N > 0 && V > 0 && cam->views > 0 && N == cam->N && N % cam->views == 0, "smil_clip_depth_backward: bad sizes");
    hipLaunchKernelGGL(k_clip_depth_bwd, dim3(N), dim3(64), 0, (hipStream_t)stream, *cam, *cd, V, d_verts);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// d fov_deg[c] = sum_{n = c mod nFov}  -(pi/360) (1 + t^2)/t * d_fov_img[n]      (x_ndc, y_ndc ~ 1/t)
__global__ void k_fov_reduce(SmilCameras c, const float *__restrict__ d_fov_img, float *__restrict__ d_fov) {
    __shared__ float red[16];
    const int col = blockIdx.x;
    float acc = 0.f;
    for (int n = col + threadIdx.x * c.nFov; n < c.N; n += blockDim.x * c.nFov) acc += d_fov_img[n];
    const float r = block_sum(acc, red);
    if (threadIdx.x == 0) {
        const float t = tanf((c.fov[col] * 0.017453292519943295f) / 2.0f);
        d_fov[col] = -(0.008726646259971648f) * (1.0f + t * t) / t * r;
    }
}

extern "C" int smil_fov_reduce(const SmilCameras *cam, const float *d_fov_img, float *d_fov, void *stream) {
    SMIL_REQUIRE(cam && d_fov_img && d_fov, "smil_fov_reduce: null argument");
    SMIL_REQUIRE(cam->nFov > 0 && cam->N % cam->nFov == 0, "smil_fov_reduce: N=%d not a multiple of nFov=%d", cam->N, cam->nFov);
    hipLaunchKernelGGL(k_fov_reduce, dim3(cam->nFov), dim3(256), 0, (hipStream_t)stream, *cam, d_fov_img, d_fov);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}
