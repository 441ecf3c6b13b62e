// Loss terms that do not need the renderer, the 2-D joint loss, and Adam.
//
// Replaces (reference): smal_fitter/fitter.py:292-331 (joint / limit / pose / splay / betas terms of
// SMALFitter.forward), :337-350 (get_temporal), smal_fitter/optimize_to_joints.py:117-127,173-175
// (torch.optim.Adam(betas=(0.5,0.999)) step).  Losses follow the reference's "sum over windows of the
// window mean" (optimize_to_joints.py:154-157): a frame's terms are divided by the size of ITS window.
#include "common.h"

__device__ __forceinline__ int window_size_of(const SmilFitConfig &c, int local_frame) {
    const int gi = c.frame0 + local_frame;
    const int w = c.window > 0 ? c.window : c.N_total;
    const int start = (gi / w) * w;
    return min(w, c.N_total - start);
}

// element e of frame i: [0,3) global rotation, [3,3J) joint rotations, [3J,3J+3) translation.
// pose is the combined (N,J,3) parameter buffer, mask the combined (J,3) mask.
__device__ __forceinline__ void prior_losses_body(const SmilFitConfig &c, const float *__restrict__ pose,
                                                  const float *__restrict__ trans, const float *__restrict__ mask_tab,
                                                  const float *__restrict__ halo_prev, const float *__restrict__ halo_next,
                                                  float *__restrict__ objs, float *__restrict__ d_pose,
                                                  float *__restrict__ d_t, int accumulate, int block, int n_blocks, float *red) {
    const int P3 = 3 * c.J;
    const int E = P3 + 3;
    const long long total = (long long)c.N * E;
    float o_limit = 0.f, o_pose = 0.f, o_splay = 0.f, o_tj = 0.f, o_tg = 0.f, o_tt = 0.f;
    for (long long idx = (long long)block * blockDim.x + threadIdx.x; idx < total; idx += (long long)n_blocks * blockDim.x) {
        const int i = (int)(idx / E), e = (int)(idx - (long long)i * E);
        const float bw = (float)window_size_of(c, i);
        const int gi = c.frame0 + i;
        const bool has_prev = gi > 0, has_next = gi + 1 < c.N_total;
        float mask, cur, prv = 0.f, nxt = 0.f, tnorm, train;
        float *dst;
        float grad = 0.f;
        if (e < P3) {
            mask = mask_tab[e];
            cur = pose[(size_t)i * P3 + e] * mask;
            if (has_prev) prv = (i > 0 ? pose[(size_t)(i - 1) * P3 + e] : halo_prev[e]) * mask;
            if (has_next) nxt = (i + 1 < c.N ? pose[(size_t)(i + 1) * P3 + e] : halo_next[e]) * mask;
            dst = d_pose + (size_t)i * P3 + e;
            if (e < 3) {
                tnorm = 3.f;
                train = c.train_global ? 1.f : 0.f;
            } else {
                tnorm = (float)(P3 - 3);
                train = c.train_joints ? 1.f : 0.f;
                // joint limits (fitter.py:303-307): mean over b_w*(J-1)*3
                if (c.w_limit > 0.f) {
                    const float s = c.w_limit / (bw * tnorm);
                    o_limit += s * (fmaxf(cur - c.limit, 0.f) + fmaxf(-c.limit - cur, 0.f));
                    grad += s * ((cur > c.limit ? 1.f : 0.f) - (cur < -c.limit ? 1.f : 0.f));
                }
                // pose prior (identity precision, root excluded; fitter.py:25-52,310-316): mean over b_w*3J
                if (c.w_pose > 0.f) {
                    const float s = c.w_pose / (bw * (float)P3);
                    o_pose += s * cur * cur;
                    grad += 2.f * s * cur;
                }
                // splay (fitter.py:319): SUM over the x and z components
                if (c.w_splay > 0.f && (e % 3) != 1) {
                    o_splay += c.w_splay * cur * cur;
                    grad += 2.f * c.w_splay * cur;
                }
            }
        } else {
            const int k = e - P3;
            mask = 1.f;
            train = c.train_trans ? 1.f : 0.f;
            cur = trans[3 * i + k];
            if (has_prev) prv = i > 0 ? trans[3 * (i - 1) + k] : halo_prev[e];
            if (has_next) nxt = i + 1 < c.N ? trans[3 * (i + 1) + k] : halo_next[e];
            tnorm = 3.f;
            dst = d_t + 3 * i + k;
        }
        if (c.w_temp > 0.f) {
            const float s = c.w_temp / tnorm;
            float tl = 0.f;
            if (has_next) { const float d = cur - nxt; tl = s * d * d; grad += 2.f * s * d; }  // pair (i,i+1) is owned by i
            if (has_prev) { const float d = cur - prv; grad += 2.f * s * d; }
            if (e < 3) o_tg += tl; else if (e < P3) o_tj += tl; else o_tt += tl;
        }
        const float upstream = accumulate ? *dst : 0.f;
        *dst = (upstream + grad) * mask * train;
    }
    float v;
    v = block_sum(o_limit, red); if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[1], v);
    v = block_sum(o_pose, red);  if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[2], v);
    v = block_sum(o_splay, red); if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[3], v);
    v = block_sum(o_tj, red);    if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[6], v);
    v = block_sum(o_tg, red);    if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[7], v);
    v = block_sum(o_tt, red);    if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[8], v);
}

__global__ void __launch_bounds__(256) k_prior_losses(SmilFitConfig c, const float *__restrict__ pose,
                                                      const float *__restrict__ trans, const float *__restrict__ mask_tab,
                                                      const float *__restrict__ halo_prev, const float *__restrict__ halo_next,
                                                      float *__restrict__ objs, float *__restrict__ d_pose,
                                                      float *__restrict__ d_t, int accumulate) {
    __shared__ float red[16];
    prior_losses_body(c, pose, trans, mask_tab, halo_prev, halo_next, objs, d_pose, d_t, accumulate, blockIdx.x, gridDim.x, red);
}

__global__ void __launch_bounds__(256) k_mask_rows(const float *__restrict__ in, const float *__restrict__ mask, long long n,
                                                   int cols, float *__restrict__ out) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x)
        out[i] = in[i] * mask[i % cols];
}

extern "C" int smil_mask_rows(const float *in, const float *mask, int64_t rows, int32_t cols, float *out, void *stream_) {
    SMIL_REQUIRE(in && mask && out && rows > 0 && cols > 0, "smil_mask_rows: bad argument");
    const long long n = (long long)rows * cols;
    hipLaunchKernelGGL(k_mask_rows, dim3((int)std::min<long long>(2048, (n + 255) / 256)), dim3(256), 0, (hipStream_t)stream_, in,
                       mask, n, cols, out);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// shape prior (fitter.py:321-330): per window mean(((beta - mean) @ prec)^2); identical for every window
__device__ __forceinline__ void betas_prior_body(const SmilFitConfig &c, const float *__restrict__ betas, const float *__restrict__ mean_betas,
                                                 const float *__restrict__ prec, float *__restrict__ objs, float *__restrict__ d_betas,
                                                 float *diff, float *res) {
    const int nB = c.nB, t = threadIdx.x;
    const int w = c.window > 0 ? c.window : c.N_total;
    // windows whose first frame lies in this rank's shard
    const int first = (c.frame0 + w - 1) / w, last = (c.frame0 + c.N + w - 1) / w;
    const float n_win = (float)(last - first);
    if (t < nB) diff[t] = betas[t] - mean_betas[t];
    __syncthreads();
    if (t < nB) {
        float r = 0.f;
        for (int m = 0; m < nB; ++m) r += diff[m] * prec[m * nB + t];
        res[t] = r;
    }
    __syncthreads();
    if (t < nB) {
        float g = 0.f;
        for (int k = 0; k < nB; ++k) g += res[k] * prec[t * nB + k];
        atomicAdd(&d_betas[t], n_win * c.w_betas * 2.f * g / (float)nB);
    }
    if (t == 0) {
        float s = 0.f;
        for (int k = 0; k < nB; ++k) s += res[k] * res[k];
        atomicAdd(&objs[4], n_win * c.w_betas * s / (float)nB);
    }
}

__global__ void k_betas_prior(SmilFitConfig c, const float *__restrict__ betas, const float *__restrict__ mean_betas,
                              const float *__restrict__ prec, float *__restrict__ objs, float *__restrict__ d_betas) {
    __shared__ float diff[SMIL_MAX_BETAS], res[SMIL_MAX_BETAS];
    betas_prior_body(c, betas, mean_betas, prec, objs, d_betas, diff, res);
}

extern "C" int smil_prior_losses(const SmilFitConfig *cfg, const float *pose, const float *trans, const float *betas,
                                 const float *mean_betas, const float *betas_prec, const float *mask, const float *halo_prev,
                                 const float *halo_next, float *objs, float *d_pose, float *d_trans, float *d_betas,
                                 int32_t accumulate, void *stream_) {
    SMIL_REQUIRE(cfg && pose && trans && mask && objs && d_pose && d_trans, "smil_prior_losses: null argument");
    SMIL_REQUIRE(cfg->N > 0 && cfg->J > 1 && cfg->N_total >= cfg->frame0 + cfg->N, "smil_prior_losses: bad sizes");
    SMIL_REQUIRE(cfg->frame0 == 0 || halo_prev || cfg->w_temp <= 0.f,
                 "smil_prior_losses: halo_prev required for a shard that does not start the sequence");
    SMIL_REQUIRE(cfg->frame0 + cfg->N == cfg->N_total || halo_next || cfg->w_temp <= 0.f,
                 "smil_prior_losses: halo_next required for a shard that does not end the sequence");
    hipStream_t stream = (hipStream_t)stream_;
    SmilFitConfig c = *cfg;
    if (!halo_prev && c.frame0 > 0 && c.w_temp <= 0.f) halo_prev = pose;  // never read as a neighbour value that matters
    if (!halo_next && c.frame0 + c.N < c.N_total && c.w_temp <= 0.f) halo_next = pose;
    const long long total = (long long)c.N * (3 * c.J + 3);
    const int grid = (int)std::min<long long>(512, (total + 255) / 256);
    hipLaunchKernelGGL(k_prior_losses, dim3(grid), dim3(256), 0, stream, c, pose, trans, mask, halo_prev, halo_next, objs, d_pose,
                       d_trans, accumulate);
    SMIL_LAUNCH_CHECK();
    if (c.w_betas > 0.f && c.nB > 0) {
        SMIL_REQUIRE(betas && mean_betas && betas_prec && d_betas, "smil_prior_losses: shape prior tables missing");
        SMIL_REQUIRE(c.nB <= SMIL_MAX_BETAS, "smil_prior_losses: nB too large");
        hipLaunchKernelGGL(k_betas_prior, dim3(1), dim3(64), 0, stream, c, betas, mean_betas, betas_prec, objs, d_betas);
        SMIL_LAUNCH_CHECK();
    }
    return SMIL_OK;
}

// ---------------------------------------------------------------------------------------------
// Everything that closes a fit iteration's loss / gradient evaluation in ONE launch (blocks take roles): the prior terms,
// the shape prior, the silhouette objective sum_n scale[n] loss_img[n] (fitter.py:332-333 after the fused rasteriser) and
// the reduction of the per-image fov sums to d_fov.  Replaces four dependent ~5 us launches at the end of every iteration.
// ---------------------------------------------------------------------------------------------
struct EpilogueArgs {
    SmilFitConfig c;
    const float *pose, *trans, *betas, *mean_betas, *prec, *mask, *halo_prev, *halo_next;
    float *objs, *d_pose, *d_trans, *d_betas;
    int accumulate, prior_blocks, do_betas;
    const float *loss_img, *pix_scale;
    int n_img, sil_blocks;
    SmilCameras cam;
    const float *d_fov_img;
    float *d_fov;
};

__global__ void __launch_bounds__(256) k_fit_epilogue(EpilogueArgs a) {
    __shared__ float red[16];
    __shared__ float diff[SMIL_MAX_BETAS], res[SMIL_MAX_BETAS];
    int blk = (int)blockIdx.x;
    if (blk < a.prior_blocks) {
        prior_losses_body(a.c, a.pose, a.trans, a.mask, a.halo_prev, a.halo_next, a.objs, a.d_pose, a.d_trans, a.accumulate, blk,
                          a.prior_blocks, red);
        return;
    }
    blk -= a.prior_blocks;
    if (a.do_betas) {
        if (blk == 0) { betas_prior_body(a.c, a.betas, a.mean_betas, a.prec, a.objs, a.d_betas, diff, res); return; }
        blk -= 1;
    }
    if (blk < a.sil_blocks) {
        float acc = 0.f;
        for (int n = blk * blockDim.x + threadIdx.x; n < a.n_img; n += a.sil_blocks * blockDim.x) acc += a.loss_img[n] * a.pix_scale[n];
        const float v = block_sum(acc, red);
        if (threadIdx.x == 0 && v != 0.f) atomicAdd(&a.objs[5], v);
        return;
    }
    blk -= a.sil_blocks;
    if (a.d_fov && blk < a.cam.nFov) {  // same arithmetic as k_fov_reduce (project.hip)
        float acc = 0.f;
        for (int n = blk + threadIdx.x * a.cam.nFov; n < a.cam.N; n += blockDim.x * a.cam.nFov) acc += a.d_fov_img[n];
        const float r = block_sum(acc, red);
        if (threadIdx.x == 0) {
            const float t = tanf((a.cam.fov[blk] * 0.017453292519943295f) / 2.0f);
            a.d_fov[blk] = -(0.008726646259971648f) * (1.0f + t * t) / t * r;
        }
    }
}

extern "C" int smil_fit_epilogue(const SmilFitConfig *cfg, const float *pose, const float *trans, const float *betas,
                                 const float *mean_betas, const float *betas_prec, const float *mask, const float *halo_prev,
                                 const float *halo_next, float *objs, float *d_pose, float *d_trans, float *d_betas, int32_t accumulate,
                                 const float *loss_img, const float *pix_scale, int32_t n_img, const SmilCameras *cam,
                                 const float *d_fov_img, float *d_fov, void *stream_) {
    SMIL_REQUIRE(cfg && pose && trans && mask && objs && d_pose && d_trans, "smil_fit_epilogue: null argument");
    SMIL_REQUIRE(cfg->N > 0 && cfg->J > 1 && cfg->N_total >= cfg->frame0 + cfg->N, "smil_fit_epilogue: bad sizes");
    SMIL_REQUIRE(cfg->frame0 == 0 || halo_prev || cfg->w_temp <= 0.f, "smil_fit_epilogue: halo_prev required for a shard that does not start the sequence");
    SMIL_REQUIRE(cfg->frame0 + cfg->N == cfg->N_total || halo_next || cfg->w_temp <= 0.f,
                 "smil_fit_epilogue: halo_next required for a shard that does not end the sequence");
    SMIL_REQUIRE(!loss_img || (pix_scale && n_img > 0), "smil_fit_epilogue: silhouette objective needs pix_scale and n_img");
    SMIL_REQUIRE(!d_fov || (cam && d_fov_img && cam->nFov > 0 && cam->N % cam->nFov == 0), "smil_fit_epilogue: fov reduction needs the cameras and per-image sums");
    EpilogueArgs a;
    a.c = *cfg;
    a.pose = pose; a.trans = trans; a.betas = betas; a.mean_betas = mean_betas; a.prec = betas_prec; a.mask = mask;
    a.halo_prev = (!halo_prev && a.c.frame0 > 0) ? pose : halo_prev;  // (only reachable with w_temp <= 0: never read where it matters)
    a.halo_next = (!halo_next && a.c.frame0 + a.c.N < a.c.N_total) ? pose : halo_next;
    a.objs = objs; a.d_pose = d_pose; a.d_trans = d_trans; a.d_betas = d_betas; a.accumulate = accumulate;
    a.do_betas = (a.c.w_betas > 0.f && a.c.nB > 0) ? 1 : 0;
    if (a.do_betas) {
        SMIL_REQUIRE(betas && mean_betas && betas_prec && d_betas, "smil_fit_epilogue: shape prior tables missing");
        SMIL_REQUIRE(a.c.nB <= SMIL_MAX_BETAS, "smil_fit_epilogue: nB too large");
    }
    const long long total = (long long)a.c.N * (3 * a.c.J + 3);
    a.prior_blocks = (int)std::min<long long>(512, (total + 255) / 256);
    a.loss_img = loss_img; a.pix_scale = pix_scale; a.n_img = loss_img ? n_img : 0;
    a.sil_blocks = loss_img ? std::min(64, ceil_div(n_img, 256)) : 0;
    if (cam) a.cam = *cam; else { a.cam = SmilCameras(); a.cam.nFov = 0; }
    a.d_fov_img = d_fov_img; a.d_fov = d_fov;
    const int grid = a.prior_blocks + a.do_betas + a.sil_blocks + (d_fov ? a.cam.nFov : 0);
    hipLaunchKernelGGL(k_fit_epilogue, dim3(grid), dim3(256), 0, (hipStream_t)stream_, a);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// 2-D joint loss (fitter.py:292-296). proj/d_proj (Nimg,J,2) over all model joints; canon (Jc,) selects the
// annotated ones (NULL = first Jc); target (Nimg,Jc,2), visibility (Nimg,Jc).  Mean over b_w*views*Jc*2 entries
// with invisible entries contributing 0 but counted in the denominator.
__global__ void __launch_bounds__(256) k_joint_loss(SmilFitConfig c, int views, int Jc, const int *__restrict__ canon,
                                                    const float *__restrict__ proj, const float *__restrict__ target,
                                                    const int *__restrict__ vis, float *__restrict__ objs,
                                                    float *__restrict__ d_proj) {
    __shared__ float red[16];
    const long long total = (long long)c.N * views * Jc;
    float acc = 0.f;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total; idx += (long long)gridDim.x * blockDim.x) {
        const int img = (int)(idx / Jc), k = (int)(idx - (long long)img * Jc);
        const int frame = img / views;
        const int j = canon ? canon[k] : k;
        const float s = c.w_j2d / ((float)window_size_of(c, frame) * (float)views * (float)Jc * 2.f);
        const size_t po = ((size_t)img * c.J + j) * 2, to = ((size_t)img * Jc + k) * 2;
        float gy = 0.f, gx = 0.f;
        if (vis[(size_t)img * Jc + k] != 0) {
            const float dy = proj[po] - target[to], dx = proj[po + 1] - target[to + 1];
            acc += s * (dy * dy + dx * dx);
            gy = 2.f * s * dy; gx = 2.f * s * dx;
        }
        d_proj[po] = gy; d_proj[po + 1] = gx;
    }
    const float v = block_sum(acc, red);
    if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[0], v);
}

extern "C" int smil_joint_loss(const SmilFitConfig *cfg, int32_t views, int32_t Jc, const int32_t *canon, const float *proj,
                               const float *target, const int32_t *visibility, float *objs, float *d_proj, void *stream_) {
    SMIL_REQUIRE(cfg && proj && target && visibility && objs && d_proj, "smil_joint_loss: null argument");
    SMIL_REQUIRE(views > 0 && Jc > 0 && Jc <= cfg->J, "smil_joint_loss: bad sizes views=%d Jc=%d", views, Jc);
    hipStream_t stream = (hipStream_t)stream_;
    if (canon || Jc < cfg->J) SMIL_HIP(hipMemsetAsync(d_proj, 0, (size_t)cfg->N * views * cfg->J * 2 * sizeof(float), stream));
    const long long total = (long long)cfg->N * views * Jc;
    const int grid = (int)std::min<long long>(256, (total + 255) / 256);
    hipLaunchKernelGGL(k_joint_loss, dim3(grid), dim3(256), 0, stream, *cfg, views, Jc, canon, proj, target, visibility, objs, d_proj);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// The six terms of SMALFitter.forward (fitter.py:292-333) WINDOW BY WINDOW, from what one evaluation of the whole batch left behind:
// the reference's driver calls forward once per WINDOW_SIZE frames (optimize_to_joints.py:153-157) and wants each window's own
// loss and terms back.  The fused iteration evaluates all windows at once (objs = their sum); this kernel restates the per-window
// sums - one workgroup per window, fixed summation order, no atomics - so that the drop-in forward() can serve every window of an
// epoch from ONE evaluation (SMALFitter._epoch in fitter.py).  objs_win (windows of this shard, 6): joint, limit, pose, splay,
// betas, sil_reproj - the same arithmetic per element as k_joint_loss / prior_losses_body / betas_prior_body / the silhouette objective.
struct WindowTermArgs {
    SmilFitConfig c;
    int views, Jc;
    const int *canon;
    const float *proj, *target;
    const int *vis;
    const float *pose, *mask, *objs_total, *loss_img, *pix_scale;
    float *objs_win;
};
__global__ void __launch_bounds__(256) k_window_terms(WindowTermArgs a) {
    __shared__ float red[16];
    const SmilFitConfig &c = a.c;
    const int w = c.window > 0 ? c.window : c.N_total;
    const int first_win = c.frame0 / w;                 // (shards start at window boundaries: optimize.plan_shards)
    const int gw = first_win + (int)blockIdx.x;
    const int g0 = max(gw * w, c.frame0), g1 = min(min((gw + 1) * w, c.N_total), c.frame0 + c.N);  // global frames of this window held here
    const int i0 = g0 - c.frame0, n = g1 - g0;          // local frames [i0, i0 + n)
    float *out = a.objs_win + 6 * (size_t)blockIdx.x;
    float o_joint = 0.f, o_limit = 0.f, o_pose = 0.f, o_splay = 0.f, o_sil = 0.f;
    if (n > 0) {
        const float bw = (float)window_size_of(c, i0);
        if (a.proj && c.w_j2d > 0.f) {  // fitter.py:292-296
            const float s = c.w_j2d / (bw * (float)a.views * (float)a.Jc * 2.f);
            const int total = n * a.views * a.Jc;
            for (int idx = threadIdx.x; idx < total; idx += blockDim.x) {
                const int img = i0 * a.views + idx / a.Jc, k = idx % a.Jc;
                const int j = a.canon ? a.canon[k] : k;
                if (a.vis[(size_t)img * a.Jc + k] != 0) {
                    const size_t po = ((size_t)img * c.J + j) * 2, to = ((size_t)img * a.Jc + k) * 2;
                    const float dy = a.proj[po] - a.target[to], dx = a.proj[po + 1] - a.target[to + 1];
                    o_joint += s * (dy * dy + dx * dx);
                }
            }
        }
        const int P3 = 3 * c.J;
        for (int idx = threadIdx.x; idx < n * (P3 - 3); idx += blockDim.x) {  // joint rotations only (fitter.py:303-319)
            const int i = i0 + idx / (P3 - 3), e = 3 + idx % (P3 - 3);
            const float cur = a.pose[(size_t)i * P3 + e] * a.mask[e];
            if (c.w_limit > 0.f) o_limit += c.w_limit / (bw * (float)(P3 - 3)) * (fmaxf(cur - c.limit, 0.f) + fmaxf(-c.limit - cur, 0.f));
            if (c.w_pose > 0.f) o_pose += c.w_pose / (bw * (float)P3) * cur * cur;
            if (c.w_splay > 0.f && (e % 3) != 1) o_splay += c.w_splay * cur * cur;
        }
        if (a.loss_img)  // fitter.py:332-333
            for (int k = threadIdx.x; k < n * a.views; k += blockDim.x) o_sil += a.loss_img[i0 * a.views + k] * a.pix_scale[i0 * a.views + k];
    }
    float v;
    v = block_sum(o_joint, red); if (threadIdx.x == 0) out[0] = v;
    v = block_sum(o_limit, red); if (threadIdx.x == 0) out[1] = v;
    v = block_sum(o_pose, red);  if (threadIdx.x == 0) out[2] = v;
    v = block_sum(o_splay, red); if (threadIdx.x == 0) out[3] = v;
    v = block_sum(o_sil, red);   if (threadIdx.x == 0) out[5] = v;
    if (threadIdx.x == 0) {  // the shape prior is the same for every window: the iteration's sum over the shard's windows, shared out
        const int first = (c.frame0 + w - 1) / w, last = (c.frame0 + c.N + w - 1) / w;
        out[4] = (n > 0 && last > first) ? a.objs_total[4] / (float)(last - first) : 0.f;
    }
}

extern "C" int smil_window_terms(const SmilFitConfig *cfg, int32_t views, int32_t Jc, const int32_t *canon, const float *proj,
                                 const float *target, const int32_t *visibility, const float *pose, const float *mask,
                                 const float *objs_total, const float *loss_img, const float *pix_scale, float *objs_win,
                                 int32_t n_windows, void *stream_) {
    SMIL_REQUIRE(cfg && pose && mask && objs_total && objs_win && n_windows > 0, "smil_window_terms: null argument");
    SMIL_REQUIRE(cfg->N > 0 && cfg->J > 1 && cfg->N_total >= cfg->frame0 + cfg->N, "smil_window_terms: bad sizes");
    SMIL_REQUIRE(!proj || (target && visibility && views > 0 && Jc > 0 && Jc <= cfg->J), "smil_window_terms: the joint term needs targets and visibility");
    SMIL_REQUIRE(!loss_img || (pix_scale && views > 0), "smil_window_terms: the silhouette term needs pix_scale");
    const int w = cfg->window > 0 ? cfg->window : cfg->N_total;
    SMIL_REQUIRE(cfg->frame0 % w == 0, "smil_window_terms: the shard starts inside a window (frame0=%d window=%d)", cfg->frame0, w);
    SMIL_REQUIRE(n_windows == ceil_div(cfg->N, w), "smil_window_terms: n_windows=%d but the shard holds %d", n_windows, ceil_div(cfg->N, w));
    WindowTermArgs a;
    a.c = *cfg; a.views = views; a.Jc = Jc; a.canon = canon; a.proj = proj; a.target = target; a.vis = visibility;
    a.pose = pose; a.mask = mask; a.objs_total = objs_total; a.loss_img = loss_img; a.pix_scale = pix_scale; a.objs_win = objs_win;
    hipLaunchKernelGGL(k_window_terms, dim3(n_windows), dim3(256), 0, (hipStream_t)stream_, a);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// per-image silhouette loss -> objs[5]: sum_n scale[n] * loss_img[n]
__global__ void __launch_bounds__(256) k_sil_objective(const float *__restrict__ loss_img, const float *__restrict__ pix_scale,
                                                       int N, float *__restrict__ objs) {
    __shared__ float red[16];
    float acc = 0.f;
    for (int n = blockIdx.x * blockDim.x + threadIdx.x; n < N; n += gridDim.x * blockDim.x) acc += loss_img[n] * pix_scale[n];
    const float v = block_sum(acc, red);
    if (threadIdx.x == 0 && v != 0.f) atomicAdd(&objs[5], v);
}

extern "C" int smil_sil_objective(const float *loss_img, const float *pix_scale, int32_t N, float *objs, void *stream_) {
    SMIL_REQUIRE(loss_img && pix_scale && objs && N > 0, "smil_sil_objective: bad argument");
    hipLaunchKernelGGL(k_sil_objective, dim3(std::min(64, ceil_div(N, 256))), dim3(256), 0, (hipStream_t)stream_, loss_img,
                       pix_scale, N, objs);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// pix_scale[n] = w_reproj / (b_w * views * S^2) for image n of local frame n / views
__global__ void k_pix_scale(SmilFitConfig c, int views, int S, float *__restrict__ pix_scale) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= c.N * views) return;
    pix_scale[n] = c.w_reproj / ((float)window_size_of(c, n / views) * (float)views * (float)S * (float)S);
}

extern "C" int smil_pix_scale(const SmilFitConfig *cfg, int32_t views, int32_t S, float *pix_scale, void *stream_) {
    SMIL_REQUIRE(cfg && pix_scale && views > 0 && S > 0, "smil_pix_scale: bad argument");
    hipLaunchKernelGGL(k_pix_scale, dim3(ceil_div(cfg->N * views, 256)), dim3(256), 0, (hipStream_t)stream_, *cfg, views, S, pix_scale);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// target_sum[n] = sum_px |target[n]|  (constant over the fit; computed once)
template <typename T>
__global__ void __launch_bounds__(256) k_image_abs_sum(const T *__restrict__ img, int pixels, float *__restrict__ out) {
    __shared__ float red[16];
    const T *p = img + (size_t)blockIdx.x * pixels;
    float acc = 0.f;
    for (int i = threadIdx.x; i < pixels; i += blockDim.x) acc += fabsf((float)p[i]);
    const float v = block_sum(acc, red);
    if (threadIdx.x == 0) out[blockIdx.x] = v;
}

extern "C" int smil_image_abs_sum(const void *images, int32_t is_u8, int32_t N, int32_t pixels, float *out, void *stream_) {
    SMIL_REQUIRE(images && out && N > 0 && pixels > 0, "smil_image_abs_sum: bad argument");
    if (is_u8) hipLaunchKernelGGL(k_image_abs_sum<uint8_t>, dim3(N), dim3(256), 0, (hipStream_t)stream_, (const uint8_t *)images, pixels, out);
    else hipLaunchKernelGGL(k_image_abs_sum<float>, dim3(N), dim3(256), 0, (hipStream_t)stream_, (const float *)images, pixels, out);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// torch.optim.Adam (single tensor path, no amsgrad / weight decay / maximize)
__global__ void __launch_bounds__(256) k_adam(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                              float *__restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                              float bc1, float bc2_sqrt) {
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);     // exp_avg.lerp_(grad, 1 - beta1)
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;     // exp_avg_sq.mul_(beta2).addcmul_(grad, grad, 1 - beta2)
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

struct AdamMulti {
    SmilAdamTensor t[SMIL_ADAM_MAX_TENSORS];
    float bc1[SMIL_ADAM_MAX_TENSORS], bc2_sqrt[SMIL_ADAM_MAX_TENSORS];
};

__global__ void __launch_bounds__(256) k_adam_multi(AdamMulti a, float b1, float b2, float eps) {
    const SmilAdamTensor &t = a.t[blockIdx.y];
    const float bc1 = a.bc1[blockIdx.y], bc2_sqrt = a.bc2_sqrt[blockIdx.y], lr = t.lr;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < t.n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = t.grad[i];
        const float mi = t.exp_avg[i] + (gi - t.exp_avg[i]) * (1.0f - b1);
        const float vi = t.exp_avg_sq[i] * b2 + (1.0f - b2) * gi * gi;
        t.exp_avg[i] = mi; t.exp_avg_sq[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        t.param[i] = t.param[i] - (lr / bc1) * (mi / denom);
    }
}

extern "C" int smil_adam_step_multi(const SmilAdamTensor *tensors, int32_t count, float beta1, float beta2, float eps,
                                    void *stream_) {
    SMIL_REQUIRE(tensors && count > 0 && count <= SMIL_ADAM_MAX_TENSORS, "smil_adam_step_multi: count=%d outside 1..%d", count,
                 SMIL_ADAM_MAX_TENSORS);
    AdamMulti a;
    long long n_max = 0;
    for (int k = 0; k < count; ++k) {
        const SmilAdamTensor &t = tensors[k];
        SMIL_REQUIRE(t.param && t.grad && t.exp_avg && t.exp_avg_sq && t.n > 0 && t.step > 0, "smil_adam_step_multi: bad tensor %d", k);
        a.t[k] = t;
        a.bc1[k] = 1.0f - powf(beta1, (float)t.step);
        a.bc2_sqrt[k] = sqrtf(1.0f - powf(beta2, (float)t.step));
        n_max = std::max<long long>(n_max, t.n);
    }
    const int gx = (int)std::min<long long>(512, (n_max + 255) / 256);
    hipLaunchKernelGGL(k_adam_multi, dim3(gx, count), dim3(256), 0, (hipStream_t)stream_, a, beta1, beta2, eps);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

__global__ void __launch_bounds__(256) k_adam_dev(float *__restrict__ p, const float *__restrict__ g, float *__restrict__ m,
                                                  float *__restrict__ v, long long n, float lr, float b1, float b2, float eps,
                                                  const int *__restrict__ step_dev, int step_offset) {
    const float t = (float)(*step_dev - step_offset);
    const float bc1 = 1.0f - powf(b1, t), bc2_sqrt = sqrtf(1.0f - powf(b2, t));
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long long)gridDim.x * blockDim.x) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - b1);
        const float vi = v[i] * b2 + (1.0f - b2) * gi * gi;
        m[i] = mi; v[i] = vi;
        const float denom = sqrtf(vi) / bc2_sqrt + eps;
        p[i] = p[i] - (lr / bc1) * (mi / denom);
    }
}

extern "C" int smil_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                                  float beta1, float beta2, float eps, const int32_t *step_dev, int32_t step_offset,
                                  void *stream_) {
    SMIL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step_dev, "smil_adam_step_dev: bad argument");
    const int grid = (int)std::min<long long>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(k_adam_dev, dim3(grid), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq, (long long)n,
                       lr, beta1, beta2, eps, step_dev, step_offset);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

extern "C" int smil_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n, float lr,
                              float beta1, float beta2, float eps, int32_t step, void *stream_) {
    SMIL_REQUIRE(param && grad && exp_avg && exp_avg_sq && n > 0 && step > 0, "smil_adam_step: bad argument");
    const float bc1 = 1.0f - powf(beta1, (float)step);
    const float bc2 = 1.0f - powf(beta2, (float)step);
    const int grid = (int)std::min<long long>(2048, (n + 255) / 256);
    hipLaunchKernelGGL(k_adam, dim3(grid), dim3(256), 0, (hipStream_t)stream_, param, grad, exp_avg, exp_avg_sq, (long long)n, lr,
                       beta1, beta2, eps, bc1, sqrtf(bc2));
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}
