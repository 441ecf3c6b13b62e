// Linear blend skinning for gfx950: shape blend, rest joints, Rodrigues + kinematic chain, skinning,
// joint regression - forward and hand-written backward.
//
// Replaces (reference): smal_model/batch_lbs.py:31-50 (batch_rodrigues), :75-197
// (batch_global_rigid_transformation), smal_model/smal_torch.py:198-370 (SMAL.__call__).
//
// HBM layout: every per-frame tensor is frame-major and dense.  Transforms are stored as 3x4
// row-major (12 floats) instead of the reference's 4x4 (the last row is constant).  The skin table is
// one packed uint32 (4 x u8 bone ids) + one float4 of weights per vertex: 20 B/vertex, read
// coalesced; the per-frame joint transforms (J x 48 B) are staged in LDS by each workgroup.
#include <algorithm>

#include "common.h"
#include <atomic>
#include <mutex>

#define FRAMES_PER_BLOCK 4  // one wavefront per frame in the chain kernels: frames per block of a large batch ...
// ... and of a small one (round 4: blocks of four frames ran 13 -> 21 us (k_pose_fwd) and 21 -> 36 us (k_chain_bwd) from one frame to
// eight; a frame per block - the kernels take the count from blockDim - keeps a handful of frames at the one-frame latency)
#ifndef SMALL_BATCH_FRAMES
#define SMALL_BATCH_FRAMES 256
#endif
static inline int frames_per_block(int B) { return B <= SMALL_BATCH_FRAMES ? 1 : FRAMES_PER_BLOCK; }

__device__ __forceinline__ void vertex_upstream(const float *__restrict__ d_verts_b, const float *sDJ,
                                                const int *__restrict__ colptr, const int *__restrict__ row,
                                                const float *__restrict__ cval, int v, bool regress, float dv[3]);

struct Mat34 {
    float r[9];
    float t[3];
};

__device__ __forceinline__ void rodrigues_fwd(float tx, float ty, float tz, float R[9]) {
    // batch_lbs.py:37-38: the norm is taken of theta + 1e-8, the axis uses the unshifted theta
    const float ax = tx + 1e-8f, ay = ty + 1e-8f, az = tz + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float rx = tx / angle, ry = ty / angle, rz = tz / angle;
    const float c = cosf(angle), s = sinf(angle);
    const float k = 1.0f - c;
    R[0] = c + k * rx * rx;      R[1] = k * rx * ry - s * rz; R[2] = k * rx * rz + s * ry;
    R[3] = k * ry * rx + s * rz; R[4] = c + k * ry * ry;      R[5] = k * ry * rz - s * rx;
    R[6] = k * rz * rx - s * ry; R[7] = k * rz * ry + s * rx; R[8] = c + k * rz * rz;
}

// dR (3x3) -> dtheta for R = c I + (1-c) r r^T + s H(r), angle = |theta + eps|, r = theta / angle
__device__ __forceinline__ void rodrigues_bwd(float tx, float ty, float tz, const float dR[9], float dth[3]) {
    const float ax = tx + 1e-8f, ay = ty + 1e-8f, az = tz + 1e-8f;
    const float angle = sqrtf(ax * ax + ay * ay + az * az);
    const float inv = 1.0f / angle;
    const float r[3] = {tx * inv, ty * inv, tz * inv};
    const float c = cosf(angle), s = sinf(angle), k = 1.0f - c;
    // d angle through cos/sin:  sum dR_mn * (-s d_mn + s r_m r_n + c H_mn)
    const float trace = dR[0] + dR[4] + dR[8];
    float rdr = 0.f;  // r^T dR r
    for (int m = 0; m < 3; ++m)
        for (int n = 0; n < 3; ++n) rdr += dR[3 * m + n] * r[m] * r[n];
    // sum dR_mn H_mn with H = [[0,-r2,r1],[r2,0,-r0],[-r1,r0,0]]
    const float hx = dR[7] - dR[5], hy = dR[2] - dR[6], hz = dR[3] - dR[1];
    const float dRH = r[0] * hx + r[1] * hy + r[2] * hz;
    float dangle = -s * trace + s * rdr + c * dRH;
    // d r_k = (1-c) ((dR + dR^T) r)_k + s * (hx,hy,hz)_k
    float dr[3];
    dr[0] = k * ((dR[0] + dR[0]) * r[0] + (dR[1] + dR[3]) * r[1] + (dR[2] + dR[6]) * r[2]) + s * hx;
    dr[1] = k * ((dR[3] + dR[1]) * r[0] + (dR[4] + dR[4]) * r[1] + (dR[5] + dR[7]) * r[2]) + s * hy;
    dr[2] = k * ((dR[6] + dR[2]) * r[0] + (dR[7] + dR[5]) * r[1] + (dR[8] + dR[8]) * r[2]) + s * hz;
    // r = theta / angle
    dangle -= (dr[0] * tx + dr[1] * ty + dr[2] * tz) * inv * inv;
    dth[0] = dr[0] * inv + dangle * ax * inv;
    dth[1] = dr[1] * inv + dangle * ay * inv;
    dth[2] = dr[2] * inv + dangle * az * inv;
}

// ---------------------------------------------------------------------------------------------
// shape blend: v_shaped[s] = (v_template (+ del_v[s])) + beta[s] @ shapedirs     (smal_torch.py:240-248)
// ---------------------------------------------------------------------------------------------
// A workgroup owns a TILE of the 3V-vector (256 threads x SB_EPT elements) and a share of the frames: the tile's rows of shapedirs and of
// the template are read ONCE into registers, after which every frame costs its betas (wave-uniform: scalar loads) and one coalesced
// store per element - the kernel is the write stream of v_shaped (12 V bytes per frame) and nothing else.  Round 6: the first form - one
// thread per (frame, element), every thread re-reading nB rows of shapedirs through L2 - ran at 1.3 TB/s and was the LARGEST kernel of the
// LBS forward once betas are per frame (the reference's neural caller, smil_image_regressor.py:2663): 109 us of 330 on STICK, 426 of 780
// on the mouse at 4 096 frames (profiles/r6_shape_blend.txt).  NB: register rows (>= nB_used); 0 = any nB, rows re-read per frame.
#define SB_EPT 4
template <int NB>
__global__ void __launch_bounds__(256) k_shape_blend(const float *__restrict__ vt, const float *__restrict__ sd,
                                                     const float *__restrict__ beta, const float *__restrict__ del_v,
                                                     float *__restrict__ v_shaped, int V3, int nB_used, int beta_stride, int nS) {
    const int e0 = blockIdx.x * (256 * SB_EPT) + threadIdx.x;
    float base[SB_EPT], row[NB > 0 ? NB : 1][SB_EPT];
#pragma unroll
    for (int i = 0; i < SB_EPT; ++i) {
        const int e = e0 + 256 * i;
        base[i] = e < V3 ? vt[e] : 0.f;
#pragma unroll
        for (int k = 0; k < NB; ++k) row[k][i] = (k < nB_used && e < V3) ? sd[(size_t)k * V3 + e] : 0.f;
    }
    for (int s_ = blockIdx.y; s_ < nS; s_ += gridDim.y) {
        const float *__restrict__ b = beta + (size_t)s_ * beta_stride;  // (beta_stride 0: one row shared by every frame)
        float acc[SB_EPT];
#pragma unroll
        for (int i = 0; i < SB_EPT; ++i) acc[i] = 0.f;
        if (NB > 0) {
#pragma unroll
            for (int k = 0; k < NB; ++k) {
                const float bk = k < nB_used ? b[k] : 0.f;
#pragma unroll
                for (int i = 0; i < SB_EPT; ++i) acc[i] += bk * row[k][i];
            }
        } else {
            for (int k = 0; k < nB_used; ++k) {
                const float bk = b[k];
#pragma unroll
                for (int i = 0; i < SB_EPT; ++i) { const int e = e0 + 256 * i; acc[i] += bk * (e < V3 ? sd[(size_t)k * V3 + e] : 0.f); }
            }
        }
#pragma unroll
        for (int i = 0; i < SB_EPT; ++i) {
            const int e = e0 + 256 * i;
            if (e >= V3) continue;
            float v = base[i];
            if (del_v) v += del_v[(size_t)s_ * V3 + e];
            v_shaped[(size_t)s_ * V3 + e] = v + acc[i];
        }
    }
}

// rest joints: J = J_static or J_regressor^T v_shaped (CSR gather)               (smal_torch.py:257-264)
// One wave per joint (strided over the block's waves), lanes over the non-zeros of its regressor row.
__global__ void __launch_bounds__(1024) k_rest_joints(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                     const float *__restrict__ val, const float *__restrict__ v_shaped,
                                                     const float *__restrict__ J_static, float *__restrict__ J_rest, int V, int J,
                                                     int is_static) {
    const int s = blockIdx.x;
    if (is_static) {
        for (int idx = threadIdx.x; idx < 3 * J; idx += blockDim.x) J_rest[(size_t)s * J * 3 + idx] = J_static[idx];
        return;
    }
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    const float *vs = v_shaped + (size_t)s * V * 3;
    for (int j = wid; j < J; j += nw) {
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int e = rowptr[j] + lane; e < rowptr[j + 1]; e += WAVE) {
            const float w = val[e];
            const float *p = vs + 3 * col[e];
            a0 += p[0] * w; a1 += p[1] * w; a2 += p[2] * w;
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            float *o = J_rest + ((size_t)s * J + j) * 3;
            o[0] = a0; o[1] = a1; o[2] = a2;
        }
    }
}

// ---------------------------------------------------------------------------------------------
// pose: Rodrigues + level-synchronous kinematic chain, one wavefront per frame, lane = joint.
// World transforms live in LDS while the chain is walked (parents precede children, so level d only
// reads level d-1).
// ---------------------------------------------------------------------------------------------
struct PoseArgs {
    const float *theta, *theta_mask, *Rs_in, *logscale, *btrans, *J_rest;
    const int *parents, *depth;
    float *Rs, *G, *A, *new_J, *joints_static;
    const float *joints_trans;  // added to the static joints (SMALFitter semantics) or NULL
    int B, J, max_depth, nS, logscale_shared, btrans_shared, propagate, use_scale;
};

// NJ = joints per lane (J <= 64 NJ: the skin table's 8-bit bone ids bound J by 256).  Round 4: everything a joint reads from memory -
// rotation, scale, offset to its parent - is requested BEFORE the chain is walked and waits in registers; the level loop touches LDS only.
// (With the loads inside it every level of the chain was a memory round trip of its own: 13 us for ONE frame.)
template <int NJ>
__global__ void __launch_bounds__(64 * FRAMES_PER_BLOCK) k_pose_fwd(PoseArgs a) {
    extern __shared__ float smem[];
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * (int)(blockDim.x >> 6) + wid;
    const bool live = b < a.B;
    const int J = a.J;
    float *sG = smem + (size_t)wid * J * 15;  // (J,12) world transforms
    float *sIS = sG + J * 12;                  // (J,3) inverse scales
    const float *Jr = a.J_rest + (size_t)(a.nS == 1 ? 0 : (live ? b : 0)) * J * 3;

    int dep[NJ], par[NJ];
    float Rq[NJ][9], Sq[NJ][3], Tq[NJ][3], Jq[NJ][3];
    const bool jt = live && a.joints_static && a.joints_trans;
    const float t0 = jt ? a.joints_trans[3 * b] : 0.f, t1 = jt ? a.joints_trans[3 * b + 1] : 0.f, t2 = jt ? a.joints_trans[3 * b + 2] : 0.f;
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = lane + WAVE * q;
        dep[q] = -1; par[q] = 0;
        if (live && j < J) {
            dep[q] = a.depth[j];
            if (a.Rs_in) {
                for (int i = 0; i < 9; ++i) Rq[q][i] = a.Rs_in[((size_t)b * J + j) * 9 + i];
            } else {
                const float *th = a.theta + ((size_t)b * J + j) * 3;
                const float *mk = a.theta_mask ? a.theta_mask + 3 * j : nullptr;
                rodrigues_fwd(mk ? th[0] * mk[0] : th[0], mk ? th[1] * mk[1] : th[1], mk ? th[2] * mk[2] : th[2], Rq[q]);
            }
            Sq[q][0] = Sq[q][1] = Sq[q][2] = 1.f;
            if (a.use_scale) {
                const float *ls = a.logscale + ((size_t)(a.logscale_shared ? 0 : b) * J + j) * 3;
                Sq[q][0] = expf(ls[0]); Sq[q][1] = expf(ls[1]); Sq[q][2] = expf(ls[2]);
            }
            Jq[q][0] = Jr[3 * j]; Jq[q][1] = Jr[3 * j + 1]; Jq[q][2] = Jr[3 * j + 2];
            Tq[q][0] = Tq[q][1] = Tq[q][2] = 0.f;
            if (dep[q] > 0) {
                const int p = par[q] = a.parents[j];
                Tq[q][0] = Jq[q][0] - Jr[3 * p]; Tq[q][1] = Jq[q][1] - Jr[3 * p + 1]; Tq[q][2] = Jq[q][2] - Jr[3 * p + 2];
                if (a.btrans) {
                    const float *bt = a.btrans + ((size_t)(a.btrans_shared ? 0 : b) * J + j) * 3;
                    Tq[q][0] += bt[0]; Tq[q][1] += bt[1] * -1.0f; Tq[q][2] += bt[2];  // y flipped (batch_lbs.py:148)
                }
            }
        }
    }

    for (int d = 0; d <= a.max_depth; ++d) {
#pragma unroll
        for (int q = 0; q < NJ; ++q) {
            if (dep[q] != d) continue;
            const int j = lane + WAVE * q;
            const float *R = Rq[q], *S = Sq[q], *t = Tq[q];
            if (a.Rs)
                for (int i = 0; i < 9; ++i) a.Rs[((size_t)b * J + j) * 9 + i] = R[i];
            sIS[3 * j + 0] = 1.0f / S[0]; sIS[3 * j + 1] = 1.0f / S[1]; sIS[3 * j + 2] = 1.0f / S[2];
            float G[12];
            const float jx = Jq[q][0], jy = Jq[q][1], jz = Jq[q][2];
            if (d == 0) {
                // root: rotation only, own scale never applied (batch_lbs.py:151)
                G[0] = R[0]; G[1] = R[1]; G[2] = R[2]; G[3] = jx;
                G[4] = R[3]; G[5] = R[4]; G[6] = R[5]; G[7] = jy;
                G[8] = R[6]; G[9] = R[7]; G[10] = R[8]; G[11] = jz;
            } else {
                const int p = par[q];
                float L[9];
                for (int m = 0; m < 3; ++m) {
                    const float isp = a.propagate ? 1.0f : sIS[3 * p + m];
                    for (int n = 0; n < 3; ++n) L[3 * m + n] = (isp * R[3 * m + n]) * S[n];
                }
                const float *P = sG + 12 * p;
                for (int m = 0; m < 3; ++m) {
                    const float p0 = P[4 * m], p1 = P[4 * m + 1], p2 = P[4 * m + 2], p3 = P[4 * m + 3];
                    G[4 * m + 0] = p0 * L[0] + p1 * L[3] + p2 * L[6];
                    G[4 * m + 1] = p0 * L[1] + p1 * L[4] + p2 * L[7];
                    G[4 * m + 2] = p0 * L[2] + p1 * L[5] + p2 * L[8];
                    G[4 * m + 3] = p0 * t[0] + p1 * t[1] + p2 * t[2] + p3;
                }
            }
            for (int i = 0; i < 12; ++i) sG[12 * j + i] = G[i];
            // outputs
            const size_t o = ((size_t)b * J + j);
            float *Go = a.G + o * 12, *Ao = a.A + o * 12;
            for (int m = 0; m < 3; ++m) {
                Go[4 * m] = G[4 * m]; Go[4 * m + 1] = G[4 * m + 1]; Go[4 * m + 2] = G[4 * m + 2]; Go[4 * m + 3] = G[4 * m + 3];
                Ao[4 * m] = G[4 * m]; Ao[4 * m + 1] = G[4 * m + 1]; Ao[4 * m + 2] = G[4 * m + 2];
                // A_t = G_t - G_R J   (batch_lbs.py:192-195)
                Ao[4 * m + 3] = G[4 * m + 3] - (G[4 * m] * jx + G[4 * m + 1] * jy + G[4 * m + 2] * jz);
            }
            a.new_J[o * 3 + 0] = G[3]; a.new_J[o * 3 + 1] = G[7]; a.new_J[o * 3 + 2] = G[11];
            if (a.joints_static) {
                a.joints_static[o * 3 + 0] = G[3] + t0; a.joints_static[o * 3 + 1] = G[7] + t1; a.joints_static[o * 3 + 2] = G[11] + t2;
            }
        }
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------
// pose blend shapes (legacy SMAL / SMPL models): v_posed = v_shaped + vec(Rs[1:] - I) @ posedirs   (smal_torch.py:294-301)
// A (frames x 9(J-1)) x (9(J-1) x 3V) product; every posedirs element is loaded once per PB_FRAMES frames.
// ---------------------------------------------------------------------------------------------
#define PB_FRAMES 8
__global__ void __launch_bounds__(256) k_pose_blend_fwd(const float *__restrict__ Rs, const float *__restrict__ pd,
                                                        const float *__restrict__ v_shaped, float *__restrict__ v_posed, int B,
                                                        int J, int V3, int nS) {
    extern __shared__ float sfeat[];  // (PB_FRAMES, 9(J-1))
    const int K9 = 9 * (J - 1);
    const int b0 = blockIdx.x * PB_FRAMES;
    for (int i = threadIdx.x; i < PB_FRAMES * K9; i += blockDim.x) {
        const int fb = i / K9, k = i - fb * K9;
        const int b = b0 + fb;
        float v = 0.f;
        if (b < B) v = Rs[((size_t)b * J + 1) * 9 + k] - ((k % 9) % 4 == 0 ? 1.0f : 0.0f);  // Rs[:,1:] - I
        sfeat[i] = v;
    }
    __syncthreads();
    const int e = blockIdx.y * blockDim.x + threadIdx.x;
    if (e >= V3) return;
    float acc[PB_FRAMES];
#pragma unroll
    for (int f = 0; f < PB_FRAMES; ++f) acc[f] = 0.f;
    for (int k = 0; k < K9; ++k) {
        const float p = pd[(size_t)k * V3 + e];
#pragma unroll
        for (int f = 0; f < PB_FRAMES; ++f) acc[f] += sfeat[f * K9 + k] * p;
    }
#pragma unroll
    for (int f = 0; f < PB_FRAMES; ++f) {
        const int b = b0 + f;
        if (b < B) v_posed[(size_t)b * V3 + e] = acc[f] + v_shaped[(size_t)(nS == 1 ? 0 : b) * V3 + e];
    }
}

// d_feat[b][k] = sum_e posedirs[k][e] d_vposed[b][e]; grid (B, ceil(K9/8))
__global__ void __launch_bounds__(256) k_pose_blend_bwd(const float *__restrict__ pd, const float *__restrict__ d_vposed,
                                                        float *__restrict__ d_feat, int K9, int V3) {
    __shared__ float red[16];
    const int b = blockIdx.x, k0 = blockIdx.y * 8;
    float acc[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) acc[q] = 0.f;
    const float *dv = d_vposed + (size_t)b * V3;
    for (int e = threadIdx.x; e < V3; e += blockDim.x) {
        const float g = dv[e];
#pragma unroll
        for (int q = 0; q < 8; ++q)
            if (k0 + q < K9) acc[q] += pd[(size_t)(k0 + q) * V3 + e] * g;
    }
#pragma unroll
    for (int q = 0; q < 8; ++q) {
        const float r = block_sum(acc[q], red);
        if (threadIdx.x == 0 && k0 + q < K9) d_feat[(size_t)b * K9 + k0 + q] = r;
    }
}

// d_vposed[b][v] = (sum_k w_k A_k[:3,:3])^T dv  - the part of k_shape_bwd that the pose-blend backward needs first
__global__ void __launch_bounds__(256) k_vposed_bwd(const float *__restrict__ d_verts, const float *__restrict__ d_joints,
                                                    const float *__restrict__ A, const uint32_t *__restrict__ skin_idx,
                                                    const float4 *__restrict__ skin_w, const int *__restrict__ colptr,
                                                    const int *__restrict__ row, const float *__restrict__ cval,
                                                    float *__restrict__ d_vposed, int V, int J, int regress) {
    extern __shared__ float smem[];
    float *sA = smem, *sDJ = smem + J * 12;
    const int b = blockIdx.x;
    const bool reg = regress && d_joints;
    for (int i = threadIdx.x; i < J * 12; i += blockDim.x) sA[i] = A[(size_t)b * J * 12 + i];
    for (int i = threadIdx.x; i < J * 3; i += blockDim.x) sDJ[i] = reg ? d_joints[(size_t)b * J * 3 + i] : 0.f;
    __syncthreads();
    const int v = blockIdx.y * blockDim.x + threadIdx.x;
    if (v >= V) return;
    float dv[3];
    vertex_upstream(d_verts ? d_verts + (size_t)b * V * 3 : nullptr, sDJ, colptr, row, cval, v, reg, dv);
    const uint32_t ids = skin_idx[v];
    const float4 w4 = skin_w[v];
    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
    float T[9];
    for (int i = 0; i < 9; ++i) T[i] = 0.f;
    for (int k = 0; k < SMIL_MAX_BONES; ++k) {
        if (w[k] == 0.f) continue;
        const float *Ak = sA + 12 * ((ids >> (8 * k)) & 0xFF);
        for (int m = 0; m < 3; ++m) { T[3 * m] += w[k] * Ak[4 * m]; T[3 * m + 1] += w[k] * Ak[4 * m + 1]; T[3 * m + 2] += w[k] * Ak[4 * m + 2]; }
    }
    float *o = d_vposed + ((size_t)b * V + v) * 3;
    for (int n = 0; n < 3; ++n) o[n] = T[n] * dv[0] + T[3 + n] * dv[1] + T[6 + n] * dv[2];
}

// ---------------------------------------------------------------------------------------------
// skinning: verts = (sum_k w_k A_k) [v;1] + trans                               (smal_torch.py:320-340)
// grid (B, ceil(V/256)); the frame's J transforms are staged in LDS.
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_skin_fwd(const float *__restrict__ A, const float *__restrict__ v_posed,
                                                  const uint32_t *__restrict__ skin_idx,
                                                  const float4 *__restrict__ skin_w, const float *__restrict__ trans,
                                                  float *__restrict__ verts, int V, int J, int nS) {
    extern __shared__ float sA[];
    const int b = blockIdx.x;
    const float *Ab = A + (size_t)b * J * 12;
    for (int i = threadIdx.x; i < J * 12; i += blockDim.x) sA[i] = Ab[i];
    __syncthreads();
    const int v = blockIdx.y * blockDim.x + threadIdx.x;
    if (v >= V) return;
    const uint32_t ids = skin_idx[v];
    const float4 w4 = skin_w[v];
    const float w[4] = {w4.x, w4.y, w4.z, w4.w};
    float T[12];
    for (int i = 0; i < 12; ++i) T[i] = 0.f;
    for (int k = 0; k < SMIL_MAX_BONES; ++k) {
        if (w[k] == 0.f) continue;
        const float *Ak = sA + 12 * ((ids >> (8 * k)) & 0xFF);
        for (int i = 0; i < 12; ++i) T[i] += w[k] * Ak[i];
    }
    const float *vp = v_posed + ((size_t)(nS == 1 ? 0 : b) * V + v) * 3;
    const float x = vp[0], y = vp[1], z = vp[2];
    float ox = T[0] * x + T[1] * y + T[2] * z + T[3];
    float oy = T[4] * x + T[5] * y + T[6] * z + T[7];
    float oz = T[8] * x + T[9] * y + T[10] * z + T[11];
    if (trans) { ox += trans[3 * b]; oy += trans[3 * b + 1]; oz += trans[3 * b + 2]; }
    float *o = verts + ((size_t)b * V + v) * 3;
    o[0] = ox; o[1] = oy; o[2] = oz;
}

// posed joints by regression from the posed vertices                           (smal_torch.py:348-351)
// trans_after != NULL: the vertices already carry the frame translation but the reference regresses the
// joints from the untranslated vertices and adds the translation afterwards (fitter.py:280-281).
__global__ void __launch_bounds__(1024) k_regress_joints(const int *__restrict__ rowptr, const int *__restrict__ col,
                                                        const float *__restrict__ val, const float *__restrict__ verts,
                                                        const float *__restrict__ trans_after, float *__restrict__ joints, int V,
                                                        int J) {
    const int b = blockIdx.x;
    const float *vb = verts + (size_t)b * V * 3;
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = blockDim.x / WAVE;
    const float t0 = trans_after ? trans_after[3 * b] : 0.f, t1 = trans_after ? trans_after[3 * b + 1] : 0.f,
                t2 = trans_after ? trans_after[3 * b + 2] : 0.f;
    for (int j = wid; j < J; j += nw) {  // one wave per joint, lanes over the non-zeros of its regressor row
        float a0 = 0.f, a1 = 0.f, a2 = 0.f;
        for (int e = rowptr[j] + lane; e < rowptr[j + 1]; e += WAVE) {
            const float w = val[e];
            const float *p = vb + 3 * col[e];
            a0 += (p[0] - t0) * w; a1 += (p[1] - t1) * w; a2 += (p[2] - t2) * w;
        }
        a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
        if (lane == 0) {
            float *o = joints + ((size_t)b * J + j) * 3;
            o[0] = a0 + t0; o[1] = a1 + t1; o[2] = a2 + t2;
        }
    }
}


// ---------------------------------------------------------------------------------------------
// skinning + joint regression + projection of one frame in one workgroup: the posed vertices are kept in LDS for the joint
// regressor (a gather) and projected through the frame's cameras as they are produced, so `verts` is written once and never
// read back by the forward pass (k_skin_fwd + k_regress_joints + k_project read it twice).  Same arithmetic, in the same
// order, as those three kernels: the outputs are bit-identical.
// ---------------------------------------------------------------------------------------------
#define FWD_FUSED_THREADS 512
#define FWD_FUSED_MAX_VIEWS 32
#ifndef FWD_UNR
#define FWD_UNR 3
#endif
#ifndef FWD_MIN_WAVES
#define FWD_MIN_WAVES 4
#endif

struct SkinProjectArgs {
    SmilCameras cam;
    const float *A, *v_skin, *trans, *trans_after;   // (B,J,12), (nS,V,3), (B,3) or NULL, = trans when it is added after the regression
    const uint32_t *skin_idx;
    const float4 *skin_w;
    const int *rowptr, *col;
    const float *val;
    float *verts, *joints, *ndc, *yx;   // (B,V,3), (B,J,3), (N,V,3) or NULL, (N,J,2) or NULL
    int B, V, J, nS, regress;           // regress 0: `joints` already holds the frame's joints (static joints, written by k_pose_fwd)
    int nnz_lds;                        // regressor entries staged in LDS (all of them, or 0: read from memory)
};

// NT threads per frame: FWD_FUSED_THREADS (two workgroups per CU), or twice that for a batch that leaves CUs idle anyway (round 4)
template <int NT>
__global__ void __launch_bounds__(NT, FWD_MIN_WAVES) k_skin_project_fwd(SkinProjectArgs a) {
    extern __shared__ float smem[];
    const int V = a.V, J = a.J, views = a.cam.views;
    constexpr int NW = NT / WAVE;
    float *vL = smem;                              // (V,3) posed vertices of the frame: what the joint regressor gathers from (models with
                                                   // static joints keep none: any mesh size fits - round 4, the mouse)
    float *sA = smem + (((a.regress ? 3 * V : 0) + 3) & ~3);  // (J,12)
    float *sCam = sA + 12 * J;                     // (views,16)
    // the joint regressor (CSR) is staged once per workgroup when it fits: a joint is then LDS reads only (from memory every joint
    // is a chain of two round trips, seven joints deep per wave - measured: 126 -> see profiles/r3_small_kernels.md)
    int *sRow = reinterpret_cast<int *>(sCam + 16 * views);  // (J+1)
    int *sCol = sRow + J + 1;                                 // (nnz)
    float *sVal = reinterpret_cast<float *>(sCol + a.nnz_lds);
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const float hS = 0.5f * (float)a.cam.S;
    const bool reg_lds = a.regress && a.nnz_lds > 0;
    if (reg_lds) {
        for (int i = tid; i <= J; i += NT) sRow[i] = a.rowptr[i];
        for (int i = tid; i < a.nnz_lds; i += NT) { sCol[i] = a.col[i]; sVal[i] = a.val[i]; }
    }
    const int *const rowp = reg_lds ? sRow : a.rowptr, *const colp = reg_lds ? sCol : a.col;
    const float *const valp = reg_lds ? sVal : a.val;
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        for (int i = tid; i < 12 * J; i += NT) sA[i] = a.A[(size_t)b * J * 12 + i];
        if (tid < views) {
            const CamParams cp = load_camera(a.cam, b * views + tid);
            float *o = sCam + 16 * tid;
            for (int i = 0; i < 9; ++i) o[i] = cp.R[i];
            for (int i = 0; i < 3; ++i) o[9 + i] = cp.T[i];
            o[12] = cp.k00; o[13] = cp.k11;
        }
        __syncthreads();
        const float *vpb = a.v_skin + (size_t)(a.nS == 1 ? 0 : b) * V * 3;
        const float tx = a.trans ? a.trans[3 * b] : 0.f, ty = a.trans ? a.trans[3 * b + 1] : 0.f, tz = a.trans ? a.trans[3 * b + 2] : 0.f;
        for (int v0 = tid; v0 < V; v0 += FWD_UNR * NT) {
            uint32_t ids[FWD_UNR];
            float4 w4[FWD_UNR];
            float P[FWD_UNR][3];
#pragma unroll
            for (int u = 0; u < FWD_UNR; ++u) {
                const int v = min(v0 + u * NT, V - 1);
                ids[u] = a.skin_idx[v];
                w4[u] = a.skin_w[v];
                P[u][0] = vpb[3 * v]; P[u][1] = vpb[3 * v + 1]; P[u][2] = vpb[3 * v + 2];
            }
#pragma unroll
            for (int u = 0; u < FWD_UNR; ++u) {
                const int v = v0 + u * NT;
                if (v >= V) continue;
                const float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
                float T[12];
#pragma unroll
                for (int i = 0; i < 12; ++i) T[i] = 0.f;
#pragma unroll
                for (int k = 0; k < SMIL_MAX_BONES; ++k) {
                    if (w[k] == 0.f) continue;
                    const float4 *Ak = reinterpret_cast<const float4 *>(sA) + 3 * ((ids[u] >> (8 * k)) & 0xFF);  // (three 16-byte LDS reads per bone)
#pragma unroll
                    for (int m_ = 0; m_ < 3; ++m_) {
                        const float4 r = Ak[m_];
                        T[4 * m_] += w[k] * r.x; T[4 * m_ + 1] += w[k] * r.y; T[4 * m_ + 2] += w[k] * r.z; T[4 * m_ + 3] += w[k] * r.w;
                    }
                }
                const float x = P[u][0], y = P[u][1], z = P[u][2];
                float ox = T[0] * x + T[1] * y + T[2] * z + T[3];
                float oy = T[4] * x + T[5] * y + T[6] * z + T[7];
                float oz = T[8] * x + T[9] * y + T[10] * z + T[11];
                if (a.trans) { ox += tx; oy += ty; oz += tz; }
                if (a.regress) { vL[3 * v] = ox; vL[3 * v + 1] = oy; vL[3 * v + 2] = oz; }
                float *o = a.verts + ((size_t)b * V + v) * 3;
                o[0] = ox; o[1] = oy; o[2] = oz;
                if (a.ndc)
                    for (int view = 0; view < views; ++view) {
                        const float *cp = sCam + 16 * view;
                        const float vx = ox * cp[0] + oy * cp[3] + oz * cp[6] + cp[9];
                        const float vy = ox * cp[1] + oy * cp[4] + oz * cp[7] + cp[10];
                        const float vz = ox * cp[2] + oy * cp[5] + oz * cp[8] + cp[11];
                        float *q = a.ndc + ((size_t)(b * views + view) * V + v) * 3;
                        q[0] = vx * cp[12] / vz; q[1] = vy * cp[13] / vz; q[2] = vz;
                    }
            }
        }
        __syncthreads();  // the frame's vertices are in LDS
        // joints: one wave per joint, lanes over the non-zeros of its regressor row (k_regress_joints), then through the cameras
        const float t0 = a.trans_after ? a.trans_after[3 * b] : 0.f, t1 = a.trans_after ? a.trans_after[3 * b + 1] : 0.f,
                    t2 = a.trans_after ? a.trans_after[3 * b + 2] : 0.f;
        // joints: sixteen lanes per joint (a regressor row holds ~30 non-zeros), four joints per wave at a time; the sum inside
        // the 16-lane row is four DPP adds.  Then the joint goes through the cameras (lane = view, sixteen at a time).
        for (int j0_ = 0; j0_ < J; j0_ += 4 * NW) {
            const int j = j0_ + 4 * wid + (lane >> 4), sub = lane & 15;
            const bool live = j < J;
            float q0 = 0.f, q1 = 0.f, q2 = 0.f;
            if (a.regress) {
                if (live)
                    for (int e = rowp[j] + sub; e < rowp[j + 1]; e += 16) {
                        const float w = valp[e];
                        const float *p = vL + 3 * colp[e];
                        q0 += (p[0] - t0) * w; q1 += (p[1] - t1) * w; q2 += (p[2] - t2) * w;
                    }
                q0 = row_sum16(q0) + t0; q1 = row_sum16(q1) + t1; q2 = row_sum16(q2) + t2;
                if (live && sub == 0) { float *o = a.joints + ((size_t)b * J + j) * 3; o[0] = q0; o[1] = q1; o[2] = q2; }
            } else if (live) {
                const float *o = a.joints + ((size_t)b * J + j) * 3;
                q0 = o[0]; q1 = o[1]; q2 = o[2];
            }
            if (a.yx && live)
                for (int view = sub; view < views; view += 16) {
                    const float *cp = sCam + 16 * view;
                    const float vx = q0 * cp[0] + q1 * cp[3] + q2 * cp[6] + cp[9];
                    const float vy = q0 * cp[1] + q1 * cp[4] + q2 * cp[7] + cp[10];
                    const float vz = q0 * cp[2] + q1 * cp[5] + q2 * cp[8] + cp[11];
                    const float xn = vx * cp[12] / vz, yn = vy * cp[13] / vz;
                    float *q = a.yx + ((size_t)(b * views + view) * J + j) * 2;
                    q[0] = hS - hS * yn; q[1] = hS - hS * xn;
                }
        }
        __syncthreads();  // the next frame overwrites vL, sA, sCam
    }
}

#define FWD_REG_LDS_MAX 4096  // regressor non-zeros staged in LDS (32 KB)
static int fwd_fused_nnz_lds(const SmilModel *m) { return (!m->static_joints && m->jreg_nnz <= FWD_REG_LDS_MAX) ? m->jreg_nnz : 0; }
// CUs and LDS of the current device, asked once per device (the grids of the per-frame persistent kernels; what a workgroup may
// take of the CU's LDS).  Thread-safe: one mutex-guarded table.
#define SMIL_MAX_DEVICES 16
struct DeviceLimits { int cus; size_t lds_block, lds_cu; };
static int current_device_slot() {  // index of the current device into per-device tables (0 when it cannot be told)
    int dev = 0;
    return (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < SMIL_MAX_DEVICES) ? dev : 0;
}
static DeviceLimits device_limits() {
    static std::mutex mu;
    static DeviceLimits table[16];
    static bool known[16] = {};
    int dev = 0;
    DeviceLimits q = {0, 64 * 1024, 64 * 1024};
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return q;
    std::lock_guard<std::mutex> lock(mu);
    if (!known[dev]) {
        int cus = 0, lds = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess) q.cus = cus;
        if (hipDeviceGetAttribute(&lds, hipDeviceAttributeMaxSharedMemoryPerBlock, dev) == hipSuccess && lds > 0) q.lds_block = (size_t)lds;
        q.lds_cu = q.lds_block;  // (MI355X: a workgroup may take all 160 KB of its CU)
        table[dev] = q;
        known[dev] = true;
    }
    return table[dev];
}
static int device_cu_count() { return device_limits().cus; }
// LDS a fused per-frame kernel may ask for so that TWO workgroups fit a CU: meshes beyond it take the separate kernels
static size_t fused_lds_limit() { return device_limits().lds_cu / 2; }

static size_t fwd_fused_lds_bytes(const SmilModel *m, int views) {
    return ((size_t)(m->static_joints ? 0 : 3 * m->V) + 4 + 12 * m->J + 16 * views + m->J + 1 + 2 * fwd_fused_nnz_lds(m)) * sizeof(float);
}

static int lbs_forward_impl(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, const SmilCameras *cam, float *ndc,
                            float *yx, hipStream_t stream);

extern "C" int smil_lbs_forward(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, void *stream_) {
    return lbs_forward_impl(m, in, out, nullptr, nullptr, nullptr, (hipStream_t)stream_);
}

extern "C" int smil_project2(const SmilCameras *cam, const float *pts_a, int32_t Pa, float *ndc_a, float *yx_a, const float *pts_b,
                             int32_t Pb, float *ndc_b, float *yx_b, void *stream);

extern "C" int smil_lbs_forward_project(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, const SmilCameras *cam,
                                        float *ndc, float *yx, void *stream_) {
    SMIL_REQUIRE(cam && (ndc || yx), "smil_lbs_forward_project: cameras and at least one of ndc / yx are required");
    SMIL_REQUIRE(in && cam->N > 0 && cam->views > 0 && cam->N == in->B * cam->views, "smil_lbs_forward_project: %d images for %d frames x %d views",
                 cam->N, in ? in->B : 0, cam->views);
    return lbs_forward_impl(m, in, out, cam, ndc, yx, (hipStream_t)stream_);
}

static int lbs_forward_impl(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, const SmilCameras *cam, float *ndc,
                            float *yx, hipStream_t stream) {
    SMIL_REQUIRE(m && in && out, "smil_lbs_forward: null argument");
    const int B = in->B, V = m->V, J = m->J;
    SMIL_REQUIRE(B > 0, "smil_lbs_forward: B=%d", B);
    SMIL_REQUIRE(in->nB_used >= 0 && in->nB_used <= m->nB, "smil_lbs_forward: nB_used=%d but the model has %d betas",
                 in->nB_used, m->nB);
    SMIL_REQUIRE(in->nB_used == 0 || in->beta, "smil_lbs_forward: beta missing");
    SMIL_REQUIRE(in->theta || in->Rs_in, "smil_lbs_forward: theta or Rs_in required");
    SMIL_REQUIRE(out->v_shaped && out->J_rest && out->G && out->A && out->new_J && out->verts && out->joints,
                 "smil_lbs_forward: null output");
    const int nS = (in->shared_beta && !in->del_v) ? 1 : B;
    const float *vt = in->v_template ? in->v_template : m->v_template;
    {
        // tiles of the 3V-vector on x, frames strided over y: as many blocks as keep every CU busy eight deep, no more (each block
        // first loads its tile's rows of shapedirs)
        const int tiles = ceil_div(3 * V, 256 * SB_EPT);
        const dim3 grid(tiles, std::max(1, std::min(nS, ceil_div(device_cu_count() * 8, tiles))));
        const int bstride = in->shared_beta ? 0 : in->nB_used;
#define SB_LAUNCH(NB) hipLaunchKernelGGL(k_shape_blend<NB>, grid, dim3(256), 0, stream, vt, m->shapedirs, in->beta, in->del_v, out->v_shaped, 3 * V, in->nB_used, bstride, nS)
        if (in->nB_used <= 8) SB_LAUNCH(8);
        else if (in->nB_used <= 16) SB_LAUNCH(16);
        else SB_LAUNCH(0);
#undef SB_LAUNCH
        SMIL_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(k_rest_joints, dim3(nS), dim3(1024), 0, stream,  // 16 waves: 3-4 joints each (latency of a single frame)
                       m->jreg_rowptr, m->jreg_col, m->jreg_val,
                       out->v_shaped, m->J_static, out->J_rest, V, J, m->static_joints ? 1 : 0);
    SMIL_LAUNCH_CHECK();
    {
        PoseArgs a;
        a.theta = in->theta; a.theta_mask = in->theta_mask; a.Rs_in = in->Rs_in;
        a.use_scale = (in->logscale && in->allow_limb_scaling) ? 1 : 0;
        a.logscale = in->logscale; a.btrans = in->btrans; a.J_rest = out->J_rest;
        a.parents = m->parents; a.depth = m->depth;
        a.Rs = out->Rs; a.G = out->G; a.A = out->A; a.new_J = out->new_J;
        a.joints_static = m->static_joints ? out->joints : nullptr;
        a.joints_trans = (in->trans_after_joints && in->trans) ? in->trans : nullptr;
        a.B = B; a.J = J; a.max_depth = m->max_depth; a.nS = nS;
        a.logscale_shared = in->logscale_shared; a.btrans_shared = in->btrans_shared;
        a.propagate = in->propagate_scaling;
        const int fpb = frames_per_block(B);
        const size_t lds = (size_t)fpb * J * 15 * sizeof(float);
        SMIL_REQUIRE(J <= 4 * WAVE, "smil_lbs_forward: J=%d exceeds the 256 joints the pose kernels hold in registers", J);
        const dim3 pg(ceil_div(B, fpb)), pb(64 * fpb);
        if (J <= WAVE) hipLaunchKernelGGL(k_pose_fwd<1>, pg, pb, lds, stream, a);
        else if (J <= 2 * WAVE) hipLaunchKernelGGL(k_pose_fwd<2>, pg, pb, lds, stream, a);
        else if (J <= 3 * WAVE) hipLaunchKernelGGL(k_pose_fwd<3>, pg, pb, lds, stream, a);
        else hipLaunchKernelGGL(k_pose_fwd<4>, pg, pb, lds, stream, a);
        SMIL_LAUNCH_CHECK();
    }
    const float *v_skin = out->v_shaped;
    int nS_skin = nS;
    if (m->posedirs) {
        SMIL_REQUIRE(out->v_posed && out->Rs, "smil_lbs_forward: this model has pose blend shapes: v_posed and Rs outputs required");
        dim3 grid(ceil_div(B, PB_FRAMES), ceil_div(3 * V, 256));
        hipLaunchKernelGGL(k_pose_blend_fwd, grid, dim3(256), (size_t)PB_FRAMES * 9 * (J - 1) * sizeof(float), stream, out->Rs,
                           m->posedirs, out->v_shaped, out->v_posed, B, J, 3 * V, nS);
        SMIL_LAUNCH_CHECK();
        v_skin = out->v_posed;
        nS_skin = B;
    }
    const float *trans_after = (in->trans_after_joints && in->trans) ? in->trans : nullptr;
    if (cam && cam->views <= FWD_FUSED_MAX_VIEWS && fwd_fused_lds_bytes(m, cam->views) <= fused_lds_limit()) {
        // skinning, joint regression and both projections in one kernel per frame (the frame's vertices stay in LDS)
        SkinProjectArgs a;
        a.cam = *cam;
        a.A = out->A; a.v_skin = v_skin; a.trans = in->trans; a.trans_after = trans_after;
        a.skin_idx = m->skin_idx; a.skin_w = m->skin_w;
        a.rowptr = m->jreg_rowptr; a.col = m->jreg_col; a.val = m->jreg_val;
        a.verts = out->verts; a.joints = out->joints; a.ndc = ndc; a.yx = yx;
        a.B = B; a.V = V; a.J = J; a.nS = nS_skin; a.regress = m->static_joints ? 0 : 1;
        a.nnz_lds = fwd_fused_nnz_lds(m);
        const size_t lds = fwd_fused_lds_bytes(m, cam->views);
        const int cus = device_cu_count();
        const int per_cu = std::max(1, std::min(FWD_MIN_WAVES / 2, (int)(device_limits().lds_cu / lds)));
        if (B <= std::max(1, cus))  // at most a frame per CU: 1024 threads per frame, fewer vertices (and round trips) per thread
            hipLaunchKernelGGL(k_skin_project_fwd<2 * FWD_FUSED_THREADS>, dim3(B), dim3(2 * FWD_FUSED_THREADS), lds, stream, a);
        else
            hipLaunchKernelGGL(k_skin_project_fwd<FWD_FUSED_THREADS>, dim3(std::min(B, std::max(1, cus) * per_cu)), dim3(FWD_FUSED_THREADS), lds, stream, a);
        SMIL_LAUNCH_CHECK();
        return SMIL_OK;
    }
    {
        dim3 grid(B, ceil_div(V, 256));
        hipLaunchKernelGGL(k_skin_fwd, grid, dim3(256), (size_t)J * 12 * sizeof(float), stream, out->A, v_skin,
                           m->skin_idx, m->skin_w, in->trans, out->verts, V, J, nS_skin);
        SMIL_LAUNCH_CHECK();
    }
    if (!m->static_joints) {
        hipLaunchKernelGGL(k_regress_joints, dim3(B), dim3(1024), 0, stream, m->jreg_rowptr, m->jreg_col, m->jreg_val,
                           out->verts, trans_after, out->joints, V, J);
        SMIL_LAUNCH_CHECK();
    }
    if (cam) {  // (a mesh or a camera rig beyond the fused kernel's LDS: the projection as its own launch)
        if (ndc && yx) return smil_project2(cam, out->verts, V, ndc, nullptr, out->joints, J, nullptr, yx, stream);
        if (ndc) return smil_project(cam, out->verts, V, ndc, nullptr, stream);
        return smil_project(cam, out->joints, J, nullptr, yx, stream);
    }
    return SMIL_OK;
}

// =============================================================================================
// backward
// =============================================================================================

// total upstream gradient on a posed vertex: d_verts + J_regressor (CSC gather) d_joints
__device__ __forceinline__ void vertex_upstream(const float *__restrict__ d_verts_b, const float *sDJ,
                                                const int *__restrict__ colptr, const int *__restrict__ row,
                                                const float *__restrict__ cval, int v, bool regress, float dv[3]) {
    dv[0] = dv[1] = dv[2] = 0.f;
    if (d_verts_b) { dv[0] = d_verts_b[3 * v]; dv[1] = d_verts_b[3 * v + 1]; dv[2] = d_verts_b[3 * v + 2]; }
    if (regress) {
        for (int e = colptr[v]; e < colptr[v + 1]; ++e) {
            const float w = cval[e];
            const float *dj = sDJ + 3 * row[e];
            dv[0] += w * dj[0]; dv[1] += w * dj[1]; dv[2] += w * dj[2];
        }
    }
}

// d_A[b][j] = sum_{v in bone j} w (dv (x) [v_posed;1]).  One block per frame, one wave per bone
// (strided); deterministic (no atomics).
#ifndef SKIN_BWD_THREADS
#define SKIN_BWD_THREADS 512
#endif
__global__ void __launch_bounds__(1024) k_skin_bwd_transforms(
    const float *__restrict__ d_verts, const float *__restrict__ d_joints, const float *__restrict__ v_posed,
    const int *__restrict__ bone_ptr, const int *__restrict__ bone_vid, const float *__restrict__ bone_w,
    const int *__restrict__ colptr, const int *__restrict__ row, const float *__restrict__ cval,
    float *__restrict__ d_A, int V, int J, int nS, int regress) {
    extern __shared__ float sDJ[];  // (J,3)
    const int b = blockIdx.x;
    const bool reg = regress && d_joints;
    if (reg)
        for (int i = threadIdx.x; i < 3 * J; i += blockDim.x) sDJ[i] = d_joints[(size_t)b * J * 3 + i];
    __syncthreads();
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63, nw = blockDim.x >> 6;
    const float *dvb = d_verts ? d_verts + (size_t)b * V * 3 : nullptr;
    const float *vpb = v_posed + (size_t)(nS == 1 ? 0 : b) * V * 3;
    for (int j = wid; j < J; j += nw) {
        float acc[12];
        for (int i = 0; i < 12; ++i) acc[i] = 0.f;
        for (int e = bone_ptr[j] + lane; e < bone_ptr[j + 1]; e += WAVE) {
            const int v = bone_vid[e];
            const float w = bone_w[e];
            float dv[3];
            vertex_upstream(dvb, sDJ, colptr, row, cval, v, reg, dv);
            const float x = vpb[3 * v], y = vpb[3 * v + 1], z = vpb[3 * v + 2];
            for (int r = 0; r < 3; ++r) {
                const float g = w * dv[r];
                acc[4 * r] += g * x; acc[4 * r + 1] += g * y; acc[4 * r + 2] += g * z; acc[4 * r + 3] += g;
            }
        }
        for (int i = 0; i < 12; ++i) acc[i] = wave_sum(acc[i]);
        if (lane == 0)
            for (int i = 0; i < 12; ++i) d_A[((size_t)b * J + j) * 12 + i] = acc[i];
    }
}

// Deterministic sum over frames of the SHARED shape gradient (the one quantity ranks all-reduce, SMALFitter.betas).  Every block of
// every contributing kernel leaves its partial sum as a row of `rows` (a plain store; the order of the frames inside a block is fixed
// by the launch), and the last block of the LAST kernel of the call to finish - found by a counter, no extra launch - adds the rows
// in a fixed order.  Two runs on the same inputs give the same bits, whatever order the blocks ran in (the float atomics this
// replaces did not).
struct BetaSum {
    float *rows;        // this kernel's rows [gridDim.x][n] (NULL: it contributes nothing)
    const float *all;   // finishing kernel: every row of the call, [n_all][n] ...
    int n_all;
    unsigned int *ctr;  // ... and the call's block counter; NULL in a kernel that does not finish the sum
    unsigned int *clear_ctr;  // a kernel that runs BEFORE the finishing one clears the counter (block 0; the kernel boundary orders it)
    float *out;         // (n) the gradient
    int accumulate;     // add to what `out` holds instead of overwriting it
    int n;              // shape coefficients in use
};
__device__ __forceinline__ void beta_row_store(const BetaSum &q, int k, float v) {
    __hip_atomic_store(&q.rows[(size_t)blockIdx.x * q.n + k], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);  // (write-through: another XCD reads it)
}
// Called by every thread of every block of the finishing kernel, after the block's own rows are stored.  `red`: LDS, one float per wave.
__device__ __forceinline__ void beta_sum_finish(const BetaSum &q, float *red) {
    if (!q.ctr) return;  // (uniform)
    __shared__ unsigned int s_last;
    __syncthreads();
    if (gridDim.x > 1u) {  // (a single block - a handful of frames - is its own last block: no fences, no counter)
        if (threadIdx.x == 0) {
            __threadfence();  // this block's rows are visible device-wide before it counts as done
            s_last = atomicAdd(q.ctr, 1u) == gridDim.x - 1u ? 1u : 0u;
        }
        __syncthreads();
        if (!s_last) return;  // (uniform)
        __threadfence();
    }
    const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = (blockDim.x + WAVE - 1) / WAVE;
    for (int k = 0; k < q.n; ++k) {
        float r = 0.f;
        for (int i = threadIdx.x; i < q.n_all; i += blockDim.x)  // thread t adds rows t, t + T, ... in this order
            r += __hip_atomic_load(&q.all[(size_t)i * q.n + k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        r = wave_sum(r);  // (a fixed shuffle pattern)
        if (lane == 0) red[wid] = r;
        __syncthreads();
        if (threadIdx.x == 0) {
            float tot = q.accumulate ? q.out[k] : 0.f;
            for (int w = 0; w < nw; ++w) tot += red[w];
            q.out[k] = tot;
        }
        __syncthreads();
    }
}

struct ChainBwdArgs {
    const float *theta, *theta_mask, *Rs, *logscale, *btrans, *J_rest, *G, *d_A, *d_newJ;
    const float *d_posefeat;  // (B,9(J-1)) gradient on vec(Rs[1:] - I) from the pose blend shapes, or NULL
    const float *d_Rs_up;     // (B,J,9) upstream gradient on the rotation matrices SMAL.__call__ returned, or NULL
    const int *parents, *depth;
    float *d_theta, *d_logscale, *d_btrans, *d_Jrest;
    float *d_Rs_out;          // (B,J,9) gradient on the rotation matrices themselves (matrix-valued theta), or NULL
    int B, J, max_depth, nS, logscale_shared, btrans_shared, propagate, use_scale;
    // the shape gradient that flows through the REST JOINTS, d beta[k] += sum_j d J_rest[j] . (J_regressor shapedirs[k])[j]
    // (model table jreg_shape): added here when the caller's vertex pass leaves it out (smil_lbs_backward_ndc), else NULL
    const float *jreg_shape;
    float *d_beta_frame;      // (B,nB_used) rows ADDED to (per-frame betas), or NULL
    BetaSum beta;             // shared betas: this kernel's partial rows and, when it is the call's last kernel, the final sum
    int nB_used;
};

// Reverse walk of the kinematic chain, one wavefront per frame, LDS accumulators for the gradients
// that flow child -> parent.  NJ = joints per lane; as in k_pose_fwd everything a joint reads from memory waits in registers before
// the walk starts (round 4: the walk was a memory round trip per level, 21 us for one frame).
template <int NJ>
__global__ void __launch_bounds__(64 * FRAMES_PER_BLOCK) k_chain_bwd(ChainBwdArgs a) {
    extern __shared__ float smem[];
    if (a.beta.clear_ctr && blockIdx.x == 0 && threadIdx.x == 0) *a.beta.clear_ctr = 0u;
    const int wid = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int b = blockIdx.x * (int)(blockDim.x >> 6) + wid;
    const bool live = b < a.B;
    const int J = a.J;
    float *sG = smem + (size_t)wid * J * 30;  // (J,12) world transforms
    float *sdG = sG + J * 12;                  // (J,12) their gradients
    float *sdJ = sdG + J * 12;                 // (J,3) gradient on rest joints
    float *sdS = sdJ + J * 3;                  // (J,3) gradient on log scales
    const size_t fb = live ? b : 0;
    const float *Jr = a.J_rest + (size_t)(a.nS == 1 ? 0 : fb) * J * 3;
    const float *ls = a.use_scale ? a.logscale + (size_t)(a.logscale_shared ? 0 : fb) * J * 3 : nullptr;

    if (live) {
        for (int j = lane; j < J; j += WAVE) {
            const size_t o = fb * J + j;
            float G[12], dA[12];
            for (int i = 0; i < 12; ++i) { G[i] = a.G[o * 12 + i]; dA[i] = a.d_A[o * 12 + i]; }
            const float jx = Jr[3 * j], jy = Jr[3 * j + 1], jz = Jr[3 * j + 2];
            for (int m = 0; m < 3; ++m) {
                const float dt = dA[4 * m + 3];
                sG[12 * j + 4 * m] = G[4 * m]; sG[12 * j + 4 * m + 1] = G[4 * m + 1];
                sG[12 * j + 4 * m + 2] = G[4 * m + 2]; sG[12 * j + 4 * m + 3] = G[4 * m + 3];
                // A_R = G_R ; A_t = G_t - G_R J
                sdG[12 * j + 4 * m] = dA[4 * m] - dt * jx;
                sdG[12 * j + 4 * m + 1] = dA[4 * m + 1] - dt * jy;
                sdG[12 * j + 4 * m + 2] = dA[4 * m + 2] - dt * jz;
                sdG[12 * j + 4 * m + 3] = dt + (a.d_newJ ? a.d_newJ[o * 3 + m] : 0.f);
            }
            for (int n = 0; n < 3; ++n) {
                sdJ[3 * j + n] = -(G[n] * dA[3] + G[4 + n] * dA[7] + G[8 + n] * dA[11]);
                sdS[3 * j + n] = 0.f;
            }
        }
    }
    int dep[NJ], par[NJ];
    float Rq[NJ][9], Sq[NJ][3], ISq[NJ][3], Tq[NJ][3], THq[NJ][3];
#pragma unroll
    for (int q = 0; q < NJ; ++q) {
        const int j = lane + WAVE * q;
        dep[q] = -1; par[q] = 0;
        if (live && j < J) {
            const size_t o = fb * J + j;
            dep[q] = a.depth[j];
            for (int k = 0; k < 3; ++k) { Sq[q][k] = 1.f; ISq[q][k] = 1.f; Tq[q][k] = 0.f; THq[q][k] = 0.f; }
            for (int i = 0; i < 9; ++i) Rq[q][i] = 0.f;
            if (a.d_theta && a.theta) {
                const float *th = a.theta + o * 3;
                const float *mk = a.theta_mask ? a.theta_mask + 3 * j : nullptr;
                THq[q][0] = mk ? th[0] * mk[0] : th[0]; THq[q][1] = mk ? th[1] * mk[1] : th[1]; THq[q][2] = mk ? th[2] * mk[2] : th[2];
            }
            if (dep[q] > 0) {
                const int p = par[q] = a.parents[j];
                for (int i = 0; i < 9; ++i) Rq[q][i] = a.Rs[o * 9 + i];
                if (ls) {
                    Sq[q][0] = expf(ls[3 * j]); Sq[q][1] = expf(ls[3 * j + 1]); Sq[q][2] = expf(ls[3 * j + 2]);
                    if (!a.propagate) {
                        ISq[q][0] = 1.0f / expf(ls[3 * p]); ISq[q][1] = 1.0f / expf(ls[3 * p + 1]); ISq[q][2] = 1.0f / expf(ls[3 * p + 2]);
                    }
                }
                Tq[q][0] = Jr[3 * j] - Jr[3 * p]; Tq[q][1] = Jr[3 * j + 1] - Jr[3 * p + 1]; Tq[q][2] = Jr[3 * j + 2] - Jr[3 * p + 2];
                if (a.btrans) {
                    const float *bt = a.btrans + ((size_t)(a.btrans_shared ? 0 : fb) * J + j) * 3;
                    Tq[q][0] += bt[0]; Tq[q][1] -= bt[1]; Tq[q][2] += bt[2];
                }
            }
        }
    }
    __syncthreads();
    for (int d = a.max_depth; d >= 0; --d) {
        {
#pragma unroll
            for (int q = 0; q < NJ; ++q) {
                if (dep[q] != d) continue;
                const int j = lane + WAVE * q;
                const size_t o = fb * J + j;
                float dG[12];
                for (int i = 0; i < 12; ++i) dG[i] = sdG[12 * j + i];
                float dR[9];
                if (d == 0) {
                    for (int m = 0; m < 3; ++m) {
                        dR[3 * m] = dG[4 * m]; dR[3 * m + 1] = dG[4 * m + 1]; dR[3 * m + 2] = dG[4 * m + 2];
                        sdJ[3 * j + m] += dG[4 * m + 3];
                    }
                } else {
                    const int p = par[q];
                    const float *R = Rq[q], *S = Sq[q], *isp = ISq[q], *t = Tq[q];
                    float L[9];
                    for (int m = 0; m < 3; ++m)
                        for (int n = 0; n < 3; ++n) L[3 * m + n] = (isp[m] * R[3 * m + n]) * S[n];
                    const float *P = sG + 12 * p;
                    // parent: dGp_R += dG_R L^T + dG_t (x) t ; dGp_t += dG_t
                    for (int m = 0; m < 3; ++m) {
                        for (int k = 0; k < 3; ++k) {
                            const float g = dG[4 * m] * L[3 * k] + dG[4 * m + 1] * L[3 * k + 1] + dG[4 * m + 2] * L[3 * k + 2] +
                                            dG[4 * m + 3] * t[k];
                            atomicAdd(&sdG[12 * p + 4 * m + k], g);
                        }
                        atomicAdd(&sdG[12 * p + 4 * m + 3], dG[4 * m + 3]);
                    }
                    // local: dL = Gp_R^T dG_R ; dt = Gp_R^T dG_t
                    float dL[9], dt[3];
                    for (int k = 0; k < 3; ++k) {
                        for (int n = 0; n < 3; ++n)
                            dL[3 * k + n] = P[k] * dG[n] + P[4 + k] * dG[4 + n] + P[8 + k] * dG[8 + n];
                        dt[k] = P[k] * dG[3] + P[4 + k] * dG[7] + P[8 + k] * dG[11];
                    }
                    for (int k = 0; k < 3; ++k) {
                        sdJ[3 * j + k] += dt[k];
                        atomicAdd(&sdJ[3 * p + k], -dt[k]);
                    }
                    if (a.d_btrans && a.btrans) {
                        float *o_bt = a.d_btrans + o * 3;
                        o_bt[0] = dt[0]; o_bt[1] = -dt[1]; o_bt[2] = dt[2];
                    }
                    for (int m = 0; m < 3; ++m) {
                        float dis = 0.f;
                        for (int n = 0; n < 3; ++n) {
                            dR[3 * m + n] = isp[m] * dL[3 * m + n] * S[n];
                            dis += R[3 * m + n] * S[n] * dL[3 * m + n];
                        }
                        if (ls && !a.propagate) atomicAdd(&sdS[3 * p + m], -dis * isp[m]);
                    }
                    if (ls) {
                        for (int n = 0; n < 3; ++n) {
                            float dS = 0.f;
                            for (int m = 0; m < 3; ++m) dS += isp[m] * R[3 * m + n] * dL[3 * m + n];
                            sdS[3 * j + n] += dS * S[n];
                        }
                    }
                }
                if (a.d_posefeat && j > 0)
                    for (int i = 0; i < 9; ++i) dR[i] += a.d_posefeat[fb * 9 * (J - 1) + (size_t)(j - 1) * 9 + i];
                if (a.d_Rs_up)
                    for (int i = 0; i < 9; ++i) dR[i] += a.d_Rs_up[o * 9 + i];
                if (a.d_Rs_out)
                    for (int i = 0; i < 9; ++i) a.d_Rs_out[o * 9 + i] = dR[i];
                if (a.d_theta && a.theta) {
                    float dth[3];
                    rodrigues_bwd(THq[q][0], THq[q][1], THq[q][2], dR, dth);
                    a.d_theta[o * 3] = dth[0]; a.d_theta[o * 3 + 1] = dth[1]; a.d_theta[o * 3 + 2] = dth[2];
                }
            }
        }
        __syncthreads();
    }
    if (live) {
        for (int j = lane; j < J; j += WAVE) {
            const size_t o = fb * J + j;
            for (int k = 0; k < 3; ++k) {
                if (a.d_Jrest) a.d_Jrest[o * 3 + k] = sdJ[3 * j + k];
                if (a.d_logscale) a.d_logscale[o * 3 + k] = a.use_scale ? sdS[3 * j + k] : 0.f;
            }
            if (a.d_btrans && (!a.btrans || j == 0))
                for (int k = 0; k < 3; ++k) a.d_btrans[o * 3 + k] = 0.f;
        }
    }
    if (a.jreg_shape) {  // (uniform)
        float *red = sG;  // this wave's transforms are no longer needed: [wave][k] partial sums at the start of every wave's area
        for (int k = 0; k < a.nB_used; ++k) {
            float r = 0.f;
            if (live)
                for (int i = lane; i < 3 * J; i += WAVE) r += sdJ[i] * a.jreg_shape[(size_t)k * 3 * J + i];
            r = wave_sum(r);
            if (lane == 0) {
                if (a.d_beta_frame) { if (live) a.d_beta_frame[fb * a.nB_used + k] += r; }
                else red[k] = r;
            }
        }
        if (a.beta.rows) {  // the block's frames in a fixed order: one row per block
            __syncthreads();
            if ((int)threadIdx.x < a.nB_used) {
                float r = 0.f;
                for (int w = 0; w < (int)(blockDim.x >> 6); ++w) r += smem[(size_t)w * J * 30 + threadIdx.x];
                beta_row_store(a.beta, threadIdx.x, r);
            }
        }
    }
    __syncthreads();  // (smem is free from here on)
    beta_sum_finish(a.beta, smem);
}

#define BETA_CHUNK 8
#define SHAPE_TERMS (BETA_CHUNK + 3)
#ifndef SHAPE_BWD_THREADS
#define SHAPE_BWD_THREADS 256
#endif

// Per frame: d v_posed = T_R^T dv (+ regressor^T d J_rest), reduced against shapedirs -> d_beta[b],
// and d_trans[b] = sum_v dv.  One block per frame; the per-frame sums are deterministic (fixed reduction order), the SHARED
// shape gradient is the sum of those over the frames by float atomics, i.e. reproducible to rounding only (the last bits depend
// on the order in which the blocks arrive).
__global__ void __launch_bounds__(1024) k_shape_bwd(
    const float *__restrict__ d_verts, const float *__restrict__ d_joints, const float *__restrict__ d_Jrest,
    const float *__restrict__ A, const uint32_t *__restrict__ skin_idx, const float4 *__restrict__ skin_w,
    const int *__restrict__ colptr, const int *__restrict__ row, const float *__restrict__ cval,
    const float *__restrict__ sd, float *__restrict__ d_beta_frame, BetaSum beta, float *__restrict__ d_trans,
    float *__restrict__ d_vshaped, int V, int J, int nB_used, int regress, int trans_after,
    const float *__restrict__ up_vshaped_all, int up_rows /* upstream gradient on the returned v_shaped (up_rows = 1 or B rows), or NULL */) {
    extern __shared__ float smem[];
    float *sA = smem;            // (J,12)
    float *sDJ = sA + J * 12;    // (J,3) upstream on posed joints
    float *sDR = sDJ + J * 3;    // (J,3) gradient on rest joints
    float *red = sDR + J * 3;    // (waves, SHAPE_TERMS)
    const int b = blockIdx.x;
    const bool reg_j = regress && d_joints;
    const bool reg_r = regress && d_Jrest;
    for (int i = threadIdx.x; i < J * 12; i += blockDim.x) sA[i] = A[(size_t)b * J * 12 + i];
    for (int i = threadIdx.x; i < J * 3; i += blockDim.x) {
        sDJ[i] = reg_j ? d_joints[(size_t)b * J * 3 + i] : 0.f;
        sDR[i] = reg_r ? d_Jrest[(size_t)b * J * 3 + i] : 0.f;
    }
    __syncthreads();
    const float *dvb = d_verts ? d_verts + (size_t)b * V * 3 : nullptr;
    const float *up_vshaped = !up_vshaped_all ? nullptr : (up_rows == 1 ? (b == 0 ? up_vshaped_all : nullptr) : up_vshaped_all + (size_t)b * V * 3);
    const int V3 = 3 * V;
    for (int k0 = 0; k0 < (nB_used > 0 ? nB_used : 1); k0 += BETA_CHUNK) {
        float bsum[BETA_CHUNK];
        for (int k = 0; k < BETA_CHUNK; ++k) bsum[k] = 0.f;
        float tsum[3] = {0.f, 0.f, 0.f};
        for (int v = threadIdx.x; v < V; v += blockDim.x) {
            float dv[3];
            vertex_upstream(dvb, sDJ, colptr, row, cval, v, reg_j, dv);
            if (trans_after) {
                // translation bypasses the regressor: d_trans = sum_v d_verts + sum_j d_joints
                if (dvb) { tsum[0] += dvb[3 * v]; tsum[1] += dvb[3 * v + 1]; tsum[2] += dvb[3 * v + 2]; }
            } else {
                tsum[0] += dv[0]; tsum[1] += dv[1]; tsum[2] += dv[2];
            }
            if (nB_used == 0 && !d_vshaped) continue;
            const uint32_t ids = skin_idx[v];
            const float4 w4 = skin_w[v];
            const float w[4] = {w4.x, w4.y, w4.z, w4.w};
            float T[9];
            for (int i = 0; i < 9; ++i) T[i] = 0.f;
            for (int k = 0; k < SMIL_MAX_BONES; ++k) {
                if (w[k] == 0.f) continue;
                const float *Ak = sA + 12 * ((ids >> (8 * k)) & 0xFF);
                for (int m = 0; m < 3; ++m) {
                    T[3 * m] += w[k] * Ak[4 * m]; T[3 * m + 1] += w[k] * Ak[4 * m + 1]; T[3 * m + 2] += w[k] * Ak[4 * m + 2];
                }
            }
            float dvp[3];
            for (int n = 0; n < 3; ++n) dvp[n] = T[n] * dv[0] + T[3 + n] * dv[1] + T[6 + n] * dv[2];
            if (reg_r) {
                for (int e = colptr[v]; e < colptr[v + 1]; ++e) {
                    const float wv = cval[e];
                    const float *dr = sDR + 3 * row[e];
                    dvp[0] += wv * dr[0]; dvp[1] += wv * dr[1]; dvp[2] += wv * dr[2];
                }
            }
            if (up_vshaped) { dvp[0] += up_vshaped[3 * v]; dvp[1] += up_vshaped[3 * v + 1]; dvp[2] += up_vshaped[3 * v + 2]; }
            if (d_vshaped && k0 == 0) {  // = gradient on del_v: v_shaped = v_template + blend + del_v
                float *o = d_vshaped + ((size_t)b * V + v) * 3;
                o[0] = dvp[0]; o[1] = dvp[1]; o[2] = dvp[2];
            }
            for (int k = 0; k < BETA_CHUNK; ++k) {
                if (k0 + k < nB_used) {
                    const float *s3 = sd + (size_t)(k0 + k) * V3 + 3 * v;
                    bsum[k] += s3[0] * dvp[0] + s3[1] * dvp[1] + s3[2] * dvp[2];
                }
            }
        }
        // one batched reduction of the chunk's sums (BETA_CHUNK shape terms + 3 translation terms): wave sums into
        // red[wave][term], one barrier, then one thread per term adds the waves in a fixed order
        const bool with_trans = k0 == 0 && d_trans;
        if (with_trans && trans_after && d_joints)
            for (int j = threadIdx.x; j < J; j += blockDim.x) {
                const float *dj = d_joints + ((size_t)b * J + j) * 3;
                tsum[0] += dj[0]; tsum[1] += dj[1]; tsum[2] += dj[2];
            }
        const int lane = threadIdx.x & (WAVE - 1), wid = threadIdx.x / WAVE, nw = (blockDim.x + WAVE - 1) / WAVE;
        __syncthreads();  // `red` may still be read by the previous chunk
#pragma unroll
        for (int k = 0; k < BETA_CHUNK + 3; ++k) {
            const float r = wave_sum(k < BETA_CHUNK ? bsum[k] : tsum[k - BETA_CHUNK]);
            if (lane == 0) red[wid * SHAPE_TERMS + k] = r;
        }
        __syncthreads();
        if (threadIdx.x < SHAPE_TERMS) {
            const int k = threadIdx.x;
            float r = 0.f;
            for (int w = 0; w < nw; ++w) r += red[w * SHAPE_TERMS + k];
            if (k < BETA_CHUNK) {
                if (k0 + k < nB_used) {
                    if (d_beta_frame) d_beta_frame[(size_t)b * nB_used + k0 + k] = r;
                    if (beta.rows) beta_row_store(beta, k0 + k, r);  // (one row per frame; summed over frames by the last block)
                }
            } else if (with_trans) {
                d_trans[3 * b + k - BETA_CHUNK] = r;
            }
        }
    }
    __syncthreads();
    beta_sum_finish(beta, red);
}


// =============================================================================================
// backward from the image plane in one pass per frame: projection backward (vertices and joints), skinning backward and
// shape backward, with the frame's world-space vertex gradient held in LDS instead of a (B,V,3) tensor in memory.
// Replaces k_project_bwd + k_skin_bwd_transforms + k_shape_bwd for the fit iteration (SMALFitter.forward's backward through
// p3d_renderer.py:137-146 and smal_torch.py:240-351); the shape gradient through the rest joints is added by k_chain_bwd from
// the jreg_shape table.  One workgroup walks frames blockIdx.x, + gridDim.x, ...; thread t owns vertices t, t + NT, ...
// in every phase, so the phases of a frame need one barrier (before the bone lists gather other threads' vertices).
// =============================================================================================
#ifndef NDC_BWD_THREADS
#define NDC_BWD_THREADS 512
#endif
#ifndef NDC_BWD_MIN_WAVES
#define NDC_BWD_MIN_WAVES 4   // waves per SIMD the register allocation is held to (two workgroups of 512 threads per CU)
#endif
#define NDC_BWD_MAX_BETAS 9   // + 3 translation terms = the twelve values wave_sum12 folds at once
#define NDC_BWD_MAX_VIEWS 32

struct LbsBwdNdcArgs {
    SmilCameras cam;
    const float *verts, *joints, *A, *v_skin;   // saved forward tensors (B,V,3), (B,J,3), (B,J,12), (nS,V,3)
    const float *d_ndc, *d_ndc_scale, *d_yx;    // (N,V,2) [packed rows: see smil_project_backward], (N,), (N,J,2); any may be NULL
    const uint32_t *skin_idx;
    const float4 *skin_w;
    const int *bone_ptr, *bone_vid, *bone_order;
    const float *bone_w;
    const int *colptr, *row;
    const int2 *vfirst;
    const float *cval, *sd;
    float *d_A, *d_joints, *d_beta_frame, *d_trans, *d_fov_img;
    BetaSum beta;                               // shared betas: one row per workgroup (summed by the chain kernel's last block)
    SmilClipDepth cd;                           // depth gradients of cut edges' end points from the rasteriser (range == NULL: none)
    int B, V, J, nS, nB_used, regress, trans_after, bone_slots;
};

__device__ __forceinline__ void project_point_bwd(const float *cp /* 15 floats: R, T, k00, k11, - */, float x, float y, float z,
                                                  float dxn, float dyn, float &gx, float &gy, float &gz, float &fsum) {
    const float vx = x * cp[0] + y * cp[3] + z * cp[6] + cp[9];
    const float vy = x * cp[1] + y * cp[4] + z * cp[7] + cp[10];
    const float vz = x * cp[2] + y * cp[5] + z * cp[8] + cp[11];
    const float iz = 1.0f / vz;
    const float xn = vx * cp[12] * iz, yn = vy * cp[13] * iz;
    const float dvx = dxn * cp[12] * iz, dvy = dyn * cp[13] * iz;
    const float dvz = -(xn * dxn + yn * dyn) * iz;
    gx += cp[0] * dvx + cp[1] * dvy + cp[2] * dvz;
    gy += cp[3] * dvx + cp[4] * dvy + cp[5] * dvz;
    gz += cp[6] * dvx + cp[7] * dvy + cp[8] * dvz;
    fsum += dxn * xn + dyn * yn;
}

// Every loop over the thread's vertices works on NDC_UNR of them at a time, loads first: with two workgroups per CU the time of
// a frame is its chain of memory round trips, and one round trip then covers NDC_UNR vertices (measured on 4096 STICK frames:
// 308 us with one vertex per round trip, see profiles/r3_small_kernels.md).
#define NDC_UNR 3    // phase 1 (five registers per vertex in flight)
#define NDC_UNR2 2   // phase 2 (seven + three per shape coefficient)

// NBT: shape coefficients held in registers per vertex (3, 6 or 9: the smallest that covers nB_used).
// VPL: the vertices the skinning transforms were applied to are staged in LDS next to the frame's vertex gradient (24 bytes per vertex:
// two workgroups of NT = 512 threads per CU hold meshes up to ~3 300 vertices).  Larger meshes (round 4: the mouse, V = 11 263) keep only
// the gradient in LDS - 12 bytes per vertex, ONE workgroup of NT = 1024 threads per CU with up to the CU's whole 160 KB - and phase 3
// gathers those vertices from memory (L2: one set for all frames when the betas are shared), requested one list segment ahead.
template <int NBT, bool VPL, int NT>
__global__ void __launch_bounds__(NT, NDC_BWD_MIN_WAVES) k_lbs_bwd_ndc(LbsBwdNdcArgs a) {
    extern __shared__ float smem[];
    if (a.beta.clear_ctr && blockIdx.x == 0 && threadIdx.x == 0) *a.beta.clear_ctr = 0u;
    const int V = a.V, J = a.J, views = a.cam.views;
    constexpr int NW = NT / WAVE;
    float *dvL = smem;                  // (V,3) the frame's vertex gradient
    float *vpL = dvL + 3 * V;           // (V,3) the vertices the skinning transforms were applied to (v_shaped / v_posed)   [VPL]
    float *sA = smem + (((VPL ? 6 : 3) * V + 3) & ~3);  // (J,12), 16-byte aligned
    float *sDJ = sA + 12 * J;           // (J,3) gradient on the posed joints
    float *sCam = sDJ + 3 * J;          // (views,16)
    float *sFov = sCam + 16 * views;    // (views) raw fov sums of the frame's images
    float *red = sFov + views;          // (NW,12)
    int *sBone = reinterpret_cast<int *>(red + NW * 12);  // (bone_slots,3) {first entry, end, bone} in the order the waves take them
    const int tid = threadIdx.x, lane = tid & (WAVE - 1), wid = tid / WAVE;
    const float h = 0.5f * (float)a.cam.S;
    static_assert(NW >= BONE_WAVES, "the model's bone schedule is dealt to BONE_WAVES waves (the first ones of the workgroup)");
    const int n_slots = a.bone_slots;
    for (int o = tid; o < n_slots; o += NT) {
        const int j = a.bone_order[o];
        sBone[3 * o] = j >= 0 ? a.bone_ptr[j] : 0; sBone[3 * o + 1] = j >= 0 ? a.bone_ptr[j + 1] : 0; sBone[3 * o + 2] = j;
    }
    if (VPL && a.nS == 1)  // one set of rest vertices for every frame: staged once (each thread its own vertices; phase 3 is behind a barrier)
        for (int v = tid; v < V; v += NT) { vpL[3 * v] = a.v_skin[3 * v]; vpL[3 * v + 1] = a.v_skin[3 * v + 1]; vpL[3 * v + 2] = a.v_skin[3 * v + 2]; }
    float beta_acc = 0.f;               // thread k < nB_used: the shared shape gradient summed over this workgroup's frames
    const bool have_clip = a.cd.range && a.d_ndc && a.cd.counter[0] != 0u;  // (one word: the rasteriser cut a face somewhere in the batch)
    for (int b = blockIdx.x; b < a.B; b += gridDim.x) {
        // ---- phase 0: the frame's transforms, cameras and joint gradient ----
        for (int i = tid; i < 12 * J; i += NT) sA[i] = a.A[(size_t)b * J * 12 + i];
        if (tid < views) {
            const CamParams cp = load_camera(a.cam, b * views + tid);
            float *o = sCam + 16 * tid;
            for (int i = 0; i < 9; ++i) o[i] = cp.R[i];
            for (int i = 0; i < 3; ++i) o[9 + i] = cp.T[i];
            o[12] = cp.k00; o[13] = cp.k11;
            sFov[tid] = 0.f;
        }
        __syncthreads();
        if (tid < WAVE) {  // one wave: lane = joint (strided)
            for (int j0 = 0; j0 < J; j0 += WAVE) {
                const int j = j0 + lane;
                float gx = 0.f, gy = 0.f, gz = 0.f;
                float x = 0.f, y = 0.f, z = 1.f;
                if (j < J && a.d_yx) { const float *X = a.joints + ((size_t)b * J + j) * 3; x = X[0]; y = X[1]; z = X[2]; }
                for (int view = 0; view < views; ++view) {
                    float fsum = 0.f;
                    if (j < J && a.d_yx) {
                        const size_t o = (size_t)(b * views + view) * J + j;
                        project_point_bwd(sCam + 16 * view, x, y, z, -h * a.d_yx[o * 2 + 1], -h * a.d_yx[o * 2], gx, gy, gz, fsum);
                    }
                    if (a.d_fov_img && a.d_yx) {
                        const float r = wave_sum(fsum);
                        if (lane == 0 && r != 0.f) atomicAdd(&sFov[view], r);
                    }
                }
                if (j < J) {
                    sDJ[3 * j] = gx; sDJ[3 * j + 1] = gy; sDJ[3 * j + 2] = gz;
                    if (a.d_joints) { float *o = a.d_joints + ((size_t)b * J + j) * 3; o[0] = gx; o[1] = gy; o[2] = gz; }
                }
            }
        }
        // ---- phase 1: projection backward of the vertices, view by view, into this thread's rows of dvL ----
        const float *vb = a.verts + (size_t)b * V * 3;
        if (VPL && a.nS != 1) {
            const float *vpb = a.v_skin + (size_t)b * V * 3;
            for (int v = tid; v < V; v += NT) { vpL[3 * v] = vpb[3 * v]; vpL[3 * v + 1] = vpb[3 * v + 1]; vpL[3 * v + 2] = vpb[3 * v + 2]; }
        }
        if (!a.d_ndc)
            for (int v = tid; v < V; v += NT) { dvL[3 * v] = 0.f; dvL[3 * v + 1] = 0.f; dvL[3 * v + 2] = 0.f; }
        else
            for (int view = 0; view < views; ++view) {
                const int n = b * views + view;
                const float *cp = sCam + 16 * view;
                const float sc = a.d_ndc_scale ? a.d_ndc_scale[n] : 0.f;
                const float2 *dn = reinterpret_cast<const float2 *>(a.d_ndc) + (size_t)n * V;
                float fsum = 0.f;
                for (int v0 = tid; v0 < V; v0 += NDC_UNR * NT) {
                    float2 raw[NDC_UNR];
                    float X[NDC_UNR][3];
#pragma unroll
                    for (int u = 0; u < NDC_UNR; ++u) {
                        const int v = min(v0 + u * NT, V - 1);
                        raw[u] = dn[v];
                        X[u][0] = vb[3 * v]; X[u][1] = vb[3 * v + 1]; X[u][2] = vb[3 * v + 2];
                    }
#pragma unroll
                    for (int u = 0; u < NDC_UNR; ++u) {
                        const int v = v0 + u * NT;
                        if (v >= V) continue;
                        float dxn = raw[u].x, dyn = raw[u].y;
                        if (sc != 0.f) {  // x * 2^32 + y in two's complement: a negative y borrowed one from the high word
                            const int qy = __float_as_int(raw[u].x), qx = __float_as_int(raw[u].y) - (qy >> 31);
                            dxn = sc > 0.f ? (float)qx * sc : 0.f;
                            dyn = sc > 0.f ? (float)qy * sc : 0.f;
                        }
                        float gx = 0.f, gy = 0.f, gz = 0.f;
                        project_point_bwd(cp, X[u][0], X[u][1], X[u][2], dxn, dyn, gx, gy, gz, fsum);
                        if (view > 0) { gx += dvL[3 * v]; gy += dvL[3 * v + 1]; gz += dvL[3 * v + 2]; }
                        dvL[3 * v] = gx; dvL[3 * v + 1] = gy; dvL[3 * v + 2] = gz;
                    }
                }
                if (a.d_fov_img) {
                    const float r = wave_sum(fsum);
                    if (lane == 0 && r != 0.f) atomicAdd(&sFov[view], r);
                }
            }
        __syncthreads();  // sDJ (and the fov sums) complete
        if (have_clip) {  // (uniform; never on the BASELINE configurations) depth gradients from the rasteriser's clipping plane
            for (int view = 0; view < views; ++view) {
                const size_t n = (size_t)b * views + view;
                const uint32_t first = a.cd.range[2 * n], cnt = a.cd.range[2 * n + 1];
                const float *cp = sCam + 16 * view;
                for (uint32_t e = tid; e < cnt; e += NT) {
                    const int v = a.cd.vertex[first + e];
                    const float dz = a.cd.dz[first + e];
                    if (v < 0 || v >= V || dz == 0.f) continue;
                    atomicAdd(&dvL[3 * v], dz * cp[2]); atomicAdd(&dvL[3 * v + 1], dz * cp[5]); atomicAdd(&dvL[3 * v + 2], dz * cp[8]);
                }
            }
            __syncthreads();
        }
        // ---- phase 2: + regressor^T d_joints; translation and shape terms ----
        constexpr int U2 = NBT > 6 ? 1 : NDC_UNR2;  // (nine coefficients x two vertices in flight spilled 30 registers)
        float term[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) term[i] = 0.f;
        const bool reg = a.regress && a.d_yx;
        for (int v0 = tid; v0 < V; v0 += U2 * NT) {
            uint32_t ids[U2];
            float4 w4[U2];
            int2 first[U2];  // the vertex's first regressor entry {joint | entries << 16, weight bits}
            float s3[U2][NBT][3];
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int v = min(v0 + u * NT, V - 1);
                ids[u] = a.skin_idx[v];
                w4[u] = a.skin_w[v];
                first[u] = reg ? a.vfirst[v] : make_int2(0, 0);
#pragma unroll
                for (int k = 0; k < NBT; ++k)
                    if (k < a.nB_used) {
                        const float *p = a.sd + (size_t)k * 3 * V + 3 * v;
                        s3[u][k][0] = p[0]; s3[u][k][1] = p[1]; s3[u][k][2] = p[2];
                    }
            }
#pragma unroll
            for (int u = 0; u < U2; ++u) {
                const int v = v0 + u * NT;
                if (v >= V) continue;
                float dv[3] = {dvL[3 * v], dvL[3 * v + 1], dvL[3 * v + 2]};
                if (a.trans_after) { term[9] += dv[0]; term[10] += dv[1]; term[11] += dv[2]; }
                const int n_ent = first[u].x >> 16;
                if (n_ent > 0) {
                    const float w0 = __int_as_float(first[u].y);
                    const float *dj0 = sDJ + 3 * (first[u].x & 0xFFFF);
                    dv[0] += w0 * dj0[0]; dv[1] += w0 * dj0[1]; dv[2] += w0 * dj0[2];
                    if (n_ent > 1)  // (rare: a vertex that several joints regress from)
                        for (int e = a.colptr[v] + 1; e < a.colptr[v + 1]; ++e) {
                            const float w = a.cval[e];
                            const float *dj = sDJ + 3 * a.row[e];
                            dv[0] += w * dj[0]; dv[1] += w * dj[1]; dv[2] += w * dj[2];
                        }
                    dvL[3 * v] = dv[0]; dvL[3 * v + 1] = dv[1]; dvL[3 * v + 2] = dv[2];
                }
                if (!a.trans_after) { term[9] += dv[0]; term[10] += dv[1]; term[11] += dv[2]; }
                if (a.nB_used == 0) continue;
                const float w[4] = {w4[u].x, w4[u].y, w4[u].z, w4[u].w};
                float T[9];
#pragma unroll
                for (int i = 0; i < 9; ++i) T[i] = 0.f;
#pragma unroll
                for (int k = 0; k < SMIL_MAX_BONES; ++k) {
                    if (w[k] == 0.f) continue;
                    const float4 *Ak = reinterpret_cast<const float4 *>(sA) + 3 * ((ids[u] >> (8 * k)) & 0xFF);  // (three 16-byte LDS reads per bone)
#pragma unroll
                    for (int m = 0; m < 3; ++m) {
                        const float4 r = Ak[m];
                        T[3 * m] += w[k] * r.x; T[3 * m + 1] += w[k] * r.y; T[3 * m + 2] += w[k] * r.z;
                    }
                }
                float dvp[3];
#pragma unroll
                for (int n = 0; n < 3; ++n) dvp[n] = T[n] * dv[0] + T[3 + n] * dv[1] + T[6 + n] * dv[2];
#pragma unroll
                for (int k = 0; k < NBT; ++k)
                    if (k < a.nB_used) term[k] += s3[u][k][0] * dvp[0] + s3[u][k][1] * dvp[1] + s3[u][k][2] * dvp[2];
            }
        }
        if (a.trans_after && tid < J) { term[9] += sDJ[3 * tid]; term[10] += sDJ[3 * tid + 1]; term[11] += sDJ[3 * tid + 2]; }
        {
            float q[3];
            wave_sum12(term, q);
            if ((lane & 15) == 0)
                for (int i = 0; i < 3; ++i) red[wid * 12 + i + 3 * (lane >> 4)] = q[i];
        }
        __syncthreads();  // every vertex row of dvL is final; the wave sums are in `red`
        if (tid < 12) {
            float r = 0.f;
            for (int w = 0; w < NW; ++w) r += red[w * 12 + tid];
            if (tid < NDC_BWD_MAX_BETAS) {
                if (tid < a.nB_used) {
                    if (a.d_beta_frame) a.d_beta_frame[(size_t)b * a.nB_used + tid] = r;
                    else beta_acc += r;
                }
            } else if (a.d_trans) {
                a.d_trans[3 * b + tid - NDC_BWD_MAX_BETAS] = r;
            }
        }
        if (a.d_fov_img && tid < views && sFov[tid] != 0.f) atomicAdd(&a.d_fov_img[b * views + tid], sFov[tid]);
        // ---- phase 3: d_A[j] = sum_{v in bone j} w (dv (x) [v_skin; 1]), one wave per bone (the first BONE_WAVES waves), longest lists
        // first.  The list entries of the segment after next (64 entries of this bone or of the wave's next one) are requested before
        // this segment's gathers, and - without the LDS copy - the next segment's vertices right behind them: the gathers of the gradient
        // come from LDS, so the memory round trips of a segment are all hidden behind the one before ----
        struct Seg { int o, e0, e1; };
        auto seg_next = [&](const Seg &c) -> Seg {  // (wave-uniform)
            if (c.o >= n_slots) return c;
            Seg r = {c.o, c.e0 + WAVE, c.e1};
            if (c.e0 + WAVE >= c.e1) {  // the bone is done: the wave's next one
                r.o = c.o + BONE_WAVES; r.e0 = r.e1 = 0;
                if (r.o < n_slots && sBone[3 * r.o + 2] < 0) r.o = n_slots;  // behind the wave's last bone
                if (r.o < n_slots) { r.e0 = sBone[3 * r.o]; r.e1 = sBone[3 * r.o + 1]; }
            }
            return r;
        };
        struct Ent { int vid; float w; };  // (vertex 0 with weight 0 beyond the end of a list)
        auto seg_load = [&](const Seg &c) -> Ent {
            Ent e = {0, 0.f};
            if (c.o < n_slots && c.e0 + lane < c.e1) { e.vid = a.bone_vid[c.e0 + lane]; e.w = a.bone_w[c.e0 + lane]; }
            return e;
        };
        const float *const vs = a.v_skin + (size_t)(a.nS == 1 ? 0 : b) * V * 3;
        Seg s0 = {n_slots, 0, 0};
        if (wid < BONE_WAVES && sBone[3 * wid + 2] >= 0) s0 = Seg{wid, sBone[3 * wid], sBone[3 * wid + 1]};  // (a wave's bones are its first slots)
        Seg s1 = seg_next(s0);
        Ent l0 = seg_load(s0), l1 = seg_load(s1);
        float x0 = 0.f, y0 = 0.f, z0 = 0.f;
        if (!VPL) { x0 = vs[3 * l0.vid]; y0 = vs[3 * l0.vid + 1]; z0 = vs[3 * l0.vid + 2]; }
        float acc[12];
#pragma unroll
        for (int i = 0; i < 12; ++i) acc[i] = 0.f;
        while (s0.o < n_slots) {  // (wave-uniform)
            const Seg s2 = seg_next(s1);
            const Ent l2 = seg_load(s2);
            float x1 = 0.f, y1 = 0.f, z1 = 0.f;
            if (!VPL) { x1 = vs[3 * l1.vid]; y1 = vs[3 * l1.vid + 1]; z1 = vs[3 * l1.vid + 2]; }
            {
                const int vid = l0.vid;
                const float x = VPL ? vpL[3 * vid] : x0, y = VPL ? vpL[3 * vid + 1] : y0, z = VPL ? vpL[3 * vid + 2] : z0;
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const float g = l0.w * dvL[3 * vid + r];
                    acc[4 * r] += g * x; acc[4 * r + 1] += g * y; acc[4 * r + 2] += g * z; acc[4 * r + 3] += g;
                }
            }
            if (s0.e0 + WAVE >= s0.e1) {  // the bone's last segment
                float q[3];
                wave_sum12(acc, q);
                if ((lane & 15) == 0) {
                    float *o12 = a.d_A + ((size_t)b * J + sBone[3 * s0.o + 2]) * 12 + 3 * (lane >> 4);
                    o12[0] = q[0]; o12[1] = q[1]; o12[2] = q[2];
                }
#pragma unroll
                for (int i = 0; i < 12; ++i) acc[i] = 0.f;
            }
            s0 = s1; s1 = s2; l0 = l1; l1 = l2; x0 = x1; y0 = y1; z0 = z1;
        }
        __syncthreads();  // the next frame overwrites dvL, sA, sDJ
    }
    if (a.beta.rows && tid < a.nB_used) beta_row_store(a.beta, tid, beta_acc);  // (this workgroup's frames, added in launch order)
}

// out[c] = sum_b in[b][c]; grid ceil(C/64), block (64,4)
__global__ void k_reduce_rows(const float *__restrict__ in, float *__restrict__ out, int B, int C) {
    __shared__ float part[4][64];
    const int c = blockIdx.x * 64 + threadIdx.x;
    float acc = 0.f;
    if (c < C)
        for (int b = threadIdx.y; b < B; b += 4) acc += in[(size_t)b * C + c];
    part[threadIdx.y][threadIdx.x] = acc;
    __syncthreads();
    if (threadIdx.y == 0 && c < C) out[c] = (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]);
}

int smil_reduce_rows(const float *in, float *out, int B, int C, hipStream_t stream) {
    hipLaunchKernelGGL(k_reduce_rows, dim3(ceil_div(C, 64)), dim3(64, 4), 0, stream, in, out, B, C);
    SMIL_LAUNCH_CHECK();
    return SMIL_OK;
}

// What smil_lbs_backward_ndc passes instead of (B,V,3) / (B,J,3) upstream gradients.
struct NdcUpstream {
    const SmilCameras *cam;
    const float *d_ndc, *d_ndc_scale, *d_yx;
    float *d_joints, *d_fov_img;
};

#define NDC_BWD_THREADS_WIDE 1024  // the one-workgroup-per-CU form for meshes whose vertex state does not fit twice
// vp_lds: the skinned-from vertices are staged next to the gradient (24 instead of 12 bytes per vertex)
static size_t ndc_bwd_lds_bytes(const SmilModel *m, int views, bool vp_lds, int threads = 0) {
    const int waves = (threads ? threads : (vp_lds ? NDC_BWD_THREADS : NDC_BWD_THREADS_WIDE)) / WAVE;
    return ((size_t)(vp_lds ? 6 : 3) * m->V + 4 + 15 * m->J + 3 * m->bone_slots + 17 * views + waves * 12) * sizeof(float);
}
// 1: two workgroups of 512 threads per CU, all vertex state in LDS; 2: one workgroup of 1024 threads with the vertex gradient in LDS
// (meshes up to ~13 000 vertices on the 160 KB of an MI355X CU); 0: the separate kernels
static int ndc_bwd_form(const SmilModel *m, int views) {
    if (ndc_bwd_lds_bytes(m, views, true) <= fused_lds_limit()) return 1;
    if (ndc_bwd_lds_bytes(m, views, false) <= device_limits().lds_block) return 2;
    return 0;
}

extern "C" int smil_lbs_backward_ndc_supported(const SmilModel *m, int32_t nB_used, int32_t views) {
    return m && !m->posedirs && nB_used >= 0 && nB_used <= NDC_BWD_MAX_BETAS && views >= 1 && views <= NDC_BWD_MAX_VIEWS &&
           ndc_bwd_form(m, views) != 0;
}

static int lbs_backward_impl(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *sv, const SmilLbsGrads *g,
                             const NdcUpstream *up, hipStream_t stream);

extern "C" int smil_lbs_backward(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *sv,
                                 const SmilLbsGrads *g, void *stream_) {
    SMIL_REQUIRE(m && in && sv && g, "smil_lbs_backward: null argument");
    SMIL_REQUIRE(g->d_verts || g->d_joints || g->up_Rs || g->up_v_shaped, "smil_lbs_backward: no upstream gradient");
    return lbs_backward_impl(m, in, sv, g, nullptr, (hipStream_t)stream_);
}

extern "C" int smil_lbs_backward_ndc(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *sv, const SmilLbsGrads *g,
                                     const SmilCameras *cam, const float *d_ndc, const float *d_ndc_scale, const float *d_yx_joints,
                                     float *d_joints, float *d_fov_img, void *stream_) {
    SMIL_REQUIRE(m && in && sv && g && cam, "smil_lbs_backward_ndc: null argument");
    SMIL_REQUIRE(d_ndc || d_yx_joints, "smil_lbs_backward_ndc: no upstream gradient");
    SMIL_REQUIRE(!g->d_verts && !g->d_joints && !g->d_del_v && !g->up_v_shaped,
                 "smil_lbs_backward_ndc: d_verts / d_joints / d_del_v / up_v_shaped belong to smil_lbs_backward");
    SMIL_REQUIRE(cam->N > 0 && cam->views > 0 && cam->N == in->B * cam->views, "smil_lbs_backward_ndc: %d images for %d frames x %d views",
                 cam->N, in->B, cam->views);
    SMIL_REQUIRE(smil_lbs_backward_ndc_supported(m, g->d_beta ? in->nB_used : 0, cam->views),
                 "smil_lbs_backward_ndc: not available for this model / call (pose blend shapes, more than %d shape coefficients or %d views, "
                 "or a mesh beyond 80 KB of LDS): use smil_project_backward + smil_lbs_backward", NDC_BWD_MAX_BETAS, NDC_BWD_MAX_VIEWS);
    SMIL_REQUIRE(sv->verts && sv->joints && d_joints, "smil_lbs_backward_ndc: saved verts / joints and the d_joints buffer are required");
    const NdcUpstream up = {cam, d_ndc, d_ndc_scale, d_yx_joints, d_joints, d_fov_img};
    return lbs_backward_impl(m, in, sv, g, &up, (hipStream_t)stream_);
}

static int lbs_backward_impl(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *sv, const SmilLbsGrads *g,
                             const NdcUpstream *up, hipStream_t stream) {
    const int B = in->B, V = m->V, J = m->J;
    SMIL_REQUIRE(B > 0, "smil_lbs_backward: B=%d", B);
    SMIL_REQUIRE(g->d_A && g->d_Jrest && g->d_Rs, "smil_lbs_backward: scratch buffers missing");
    SMIL_REQUIRE(sv->v_shaped && sv->J_rest && sv->G && sv->A && sv->Rs, "smil_lbs_backward: saved forward tensors missing");
    const int nS = (in->shared_beta && !in->del_v) ? 1 : B;
    const int regress = m->static_joints ? 0 : 1;
    const int use_scale = (in->logscale && in->allow_limb_scaling) ? 1 : 0;

    const float *v_skin = m->posedirs ? sv->v_posed : sv->v_shaped;
    const int nS_skin = m->posedirs ? B : nS;
    SMIL_REQUIRE(!m->posedirs || (sv->v_posed && g->d_vposed), "smil_lbs_backward: pose blend shapes need v_posed and d_vposed");
    // one block per frame: many frames -> smaller blocks (more of them resident, phases decoupled); a handful of frames -> the
    // widest block (the launch is one block's latency)
    const int few_frames = B < 64;
    const int nBu_all = g->d_beta ? in->nB_used : 0;
    float *dbeta_frame_all = nullptr;
    // shared betas: every block of the kernels below leaves a partial row in g->beta_rows and the last block of the last kernel
    // adds them in a fixed order (BetaSum: bit-reproducible); per-frame betas: one row per frame
    const bool beta_shared = g->d_beta && nBu_all > 0 && in->shared_beta;
    if (g->d_beta && nBu_all > 0 && !in->shared_beta) dbeta_frame_all = g->d_beta;
    SMIL_REQUIRE(!beta_shared || g->beta_rows, "smil_lbs_backward: shared betas need the beta_rows scratch (2 B nB_used + 16 floats)");
    BetaSum bsum;  // template: rows / all / n_all / ctr are set per kernel
    bsum.rows = nullptr; bsum.all = g->beta_rows; bsum.n_all = 0; bsum.ctr = nullptr; bsum.clear_ctr = nullptr; bsum.out = g->d_beta;
    bsum.accumulate = g->accumulate_shared_beta ? 1 : 0; bsum.n = nBu_all;
    int rows_used = 0;
    // The "last block finishes" counter of the shared shape gradient is a word of THIS CALL's scratch, behind its rows, cleared by the
    // kernel that runs before the finishing one (the fused vertex pass before the chain kernel, the chain kernel before the shape
    // kernel): calls on different streams cannot interleave on it and a call that dies half way leaves nothing behind (round 4 kept
    // one counter per model, reset by the finishing block), and no memset node is added to the iteration.
    unsigned int *const beta_ctr = beta_shared ? reinterpret_cast<unsigned int *>(g->beta_rows + (size_t)2 * B * nBu_all) : nullptr;
    const float *d_joints_up = up ? up->d_joints : g->d_joints;
    if (up) {
        LbsBwdNdcArgs a;
        a.cam = *up->cam;
        a.verts = sv->verts; a.joints = sv->joints; a.A = sv->A; a.v_skin = v_skin;
        a.d_ndc = up->d_ndc; a.d_ndc_scale = up->d_ndc_scale; a.d_yx = up->d_yx;
        a.skin_idx = m->skin_idx; a.skin_w = m->skin_w;
        a.bone_ptr = m->bone_ptr; a.bone_vid = m->bone_vid; a.bone_order = m->bone_order; a.bone_w = m->bone_w;
        a.colptr = m->jreg_colptr; a.row = m->jreg_row; a.cval = m->jreg_cval; a.sd = m->shapedirs; a.vfirst = m->jreg_vfirst;
        a.d_A = g->d_A; a.d_joints = up->d_joints; a.d_beta_frame = dbeta_frame_all;
        a.d_trans = g->d_trans; a.d_fov_img = up->d_fov_img;
        a.cd = SmilClipDepth{nullptr, nullptr, nullptr, nullptr, 0};
        if (g->clip_depth) {
            SMIL_REQUIRE(g->clip_depth->vertex && g->clip_depth->dz && g->clip_depth->range && g->clip_depth->counter, "smil_lbs_backward_ndc: incomplete SmilClipDepth");
            a.cd = *g->clip_depth;
        }
        a.B = B; a.V = V; a.J = J; a.nS = nS_skin; a.nB_used = nBu_all; a.regress = regress; a.bone_slots = m->bone_slots;
        a.trans_after = in->trans_after_joints ? 1 : 0;
        const int form = ndc_bwd_form(m, up->cam->views);
        SMIL_REQUIRE(form != 0, "smil_lbs_backward_ndc: V=%d does not fit the fused kernel's LDS (ask smil_lbs_backward_ndc_supported first)", V);
        const bool wide = form == 2;
        const int cus = device_cu_count();
        // a batch that leaves CUs idle anyway (at most a frame per CU) takes 1024 threads per frame: fewer vertices per thread, fewer
        // dependent memory round trips per phase (round 4: 25 us for one frame were ~11 round trips)
        const bool few = !wide && B <= std::max(1, cus);
        const size_t lds = ndc_bwd_lds_bytes(m, up->cam->views, !wide, few ? NDC_BWD_THREADS_WIDE : 0);
        const int per_cu = (wide || few) ? 1 : std::max(1, std::min(2, (int)(device_limits().lds_cu / lds)));
        const int grid = std::min(B, std::max(1, cus) * per_cu);
        a.beta = bsum;
        if (beta_shared) { a.beta.rows = g->beta_rows; rows_used = grid; a.beta.clear_ctr = beta_ctr; }
        const int dev_slot = current_device_slot();
#define NDC_LAUNCH(NBT) \
        do { \
            if (wide) { \
                auto kern = k_lbs_bwd_ndc<NBT, false, NDC_BWD_THREADS_WIDE>; \
                static std::atomic<int> lds_allowed[SMIL_MAX_DEVICES];  /* (once per variant, size AND device - the attribute is per device; \
                                                                          not a stream operation, kept out of replays) */ \
                if (lds_allowed[dev_slot].load() < (int)lds) { \
                    SMIL_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    lds_allowed[dev_slot].store((int)lds); \
                } \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(NDC_BWD_THREADS_WIDE), lds, stream, a); \
            } else if (few) { \
                auto kern = k_lbs_bwd_ndc<NBT, true, NDC_BWD_THREADS_WIDE>; \
                static std::atomic<int> lds_allowed[SMIL_MAX_DEVICES]; \
                if (lds > 64 * 1024 && lds_allowed[dev_slot].load() < (int)lds) { \
                    SMIL_HIP(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds)); \
                    lds_allowed[dev_slot].store((int)lds); \
                } \
                hipLaunchKernelGGL(kern, dim3(grid), dim3(NDC_BWD_THREADS_WIDE), lds, stream, a); \
            } else { \
                hipLaunchKernelGGL((k_lbs_bwd_ndc<NBT, true, NDC_BWD_THREADS>), dim3(grid), dim3(NDC_BWD_THREADS), lds, stream, a); \
            } \
        } while (0)
        if (nBu_all <= 3) NDC_LAUNCH(3);
        else if (nBu_all <= 6) NDC_LAUNCH(6);
        else NDC_LAUNCH(NDC_BWD_MAX_BETAS);
#undef NDC_LAUNCH
        SMIL_LAUNCH_CHECK();
    } else {
    hipLaunchKernelGGL(k_skin_bwd_transforms, dim3(B), dim3(few_frames ? 1024 : SKIN_BWD_THREADS), (size_t)J * 3 * sizeof(float), stream, g->d_verts,
                       g->d_joints, v_skin, m->bone_ptr, m->bone_vid, m->bone_w, m->jreg_colptr, m->jreg_row,
                       m->jreg_cval, g->d_A, V, J, nS_skin, regress);
    SMIL_LAUNCH_CHECK();
    }
    const float *d_posefeat = nullptr;
    if (m->posedirs && ((g->d_theta && !in->Rs_in) || (g->d_Rs_in && in->Rs_in))) {
        // gradient through v_posed -> vec(Rs[1:] - I): d_vposed, then the transposed product with posedirs;
        // the (B,9(J-1)) result lives in the d_Rs scratch behind the per-frame scale / translation gradients
        const int K9 = 9 * (J - 1);
        dim3 gridv(B, ceil_div(V, 256));
        hipLaunchKernelGGL(k_vposed_bwd, gridv, dim3(256), (size_t)J * 15 * sizeof(float), stream, g->d_verts, g->d_joints, sv->A,
                           m->skin_idx, m->skin_w, m->jreg_colptr, m->jreg_row, m->jreg_cval, g->d_vposed, V, J, regress);
        SMIL_LAUNCH_CHECK();
        SMIL_REQUIRE(g->d_posefeat, "smil_lbs_backward: pose blend shapes need the d_posefeat scratch");
        dim3 gridf(B, ceil_div(K9, 8));
        hipLaunchKernelGGL(k_pose_blend_bwd, gridf, dim3(256), 0, stream, m->posedirs, g->d_vposed, g->d_posefeat, K9, 3 * V);
        SMIL_LAUNCH_CHECK();
        d_posefeat = g->d_posefeat;
    }

    // per-frame scale / translation gradients go to the output directly, or to scratch (d_Rs) when the
    // table is shared by all frames and has to be reduced over frames afterwards
    float *scratch = g->d_Rs;  // (B,J,9)
    float *dls_frame = nullptr, *dbt_frame = nullptr;
    if (g->d_logscale) dls_frame = in->logscale_shared ? scratch : g->d_logscale;
    if (g->d_btrans) dbt_frame = in->btrans_shared ? scratch + (size_t)B * J * 3 : g->d_btrans;
    {
        ChainBwdArgs a;
        a.theta = in->Rs_in ? nullptr : in->theta; a.theta_mask = in->theta_mask; a.Rs = sv->Rs;
        a.logscale = in->logscale; a.btrans = in->btrans; a.J_rest = sv->J_rest; a.G = sv->G; a.d_A = g->d_A;
        a.d_newJ = m->static_joints ? d_joints_up : nullptr;
        // (the fused vertex pass leaves the shape gradient through the rest joints to this kernel)
        const bool js = up && !m->static_joints && nBu_all > 0;
        a.jreg_shape = js ? m->jreg_shape : nullptr;
        a.d_beta_frame = js ? dbeta_frame_all : nullptr; a.nB_used = nBu_all;
        a.beta = bsum;
        const int fpb = frames_per_block(B);
        const int chain_blocks = ceil_div(B, fpb);
        if (beta_shared && js) { a.beta.rows = g->beta_rows + (size_t)rows_used * nBu_all; rows_used += chain_blocks; }
        if (beta_shared && up) { a.beta.n_all = rows_used; a.beta.ctr = beta_ctr; }  // (the fused route ends here: this kernel finishes the sum)
        if (beta_shared && !up) a.beta.clear_ctr = beta_ctr;                          // (the other route: the shape kernel below finishes it)
        a.d_posefeat = d_posefeat;
        a.d_Rs_up = g->up_Rs;
        a.parents = m->parents; a.depth = m->depth;
        a.d_Rs_out = in->Rs_in ? g->d_Rs_in : nullptr;
        a.d_theta = g->d_theta; a.d_logscale = dls_frame; a.d_btrans = dbt_frame; a.d_Jrest = g->d_Jrest;
        a.B = B; a.J = J; a.max_depth = m->max_depth; a.nS = nS;
        a.logscale_shared = in->logscale_shared; a.btrans_shared = in->btrans_shared;
        a.propagate = in->propagate_scaling; a.use_scale = use_scale;
        const size_t lds = (size_t)fpb * J * 30 * sizeof(float);
        SMIL_REQUIRE(J <= 4 * WAVE, "smil_lbs_backward: J=%d exceeds the 256 joints the chain kernels hold in registers", J);
        const dim3 cg(chain_blocks), cb(64 * fpb);
        if (J <= WAVE) hipLaunchKernelGGL(k_chain_bwd<1>, cg, cb, lds, stream, a);
        else if (J <= 2 * WAVE) hipLaunchKernelGGL(k_chain_bwd<2>, cg, cb, lds, stream, a);
        else if (J <= 3 * WAVE) hipLaunchKernelGGL(k_chain_bwd<3>, cg, cb, lds, stream, a);
        else hipLaunchKernelGGL(k_chain_bwd<4>, cg, cb, lds, stream, a);
        SMIL_LAUNCH_CHECK();
    }
    if (g->d_logscale && in->logscale_shared) {
        int rc = smil_reduce_rows(dls_frame, g->d_logscale, B, J * 3, stream);
        if (rc) return rc;
    }
    if (g->d_btrans && in->btrans_shared) {
        int rc = smil_reduce_rows(dbt_frame, g->d_btrans, B, J * 3, stream);
        if (rc) return rc;
    }
    if (!up && (g->d_beta || g->d_trans || g->d_del_v)) {
        const int nBu = nBu_all;
        float *dbeta_frame = dbeta_frame_all;
        BetaSum bshape = bsum;
        if (beta_shared) { bshape.rows = g->beta_rows; bshape.n_all = B; bshape.ctr = beta_ctr; }  // (one row per frame; this kernel finishes the sum)
        const int shape_threads = few_frames ? 1024 : SHAPE_BWD_THREADS;
        const size_t lds = ((size_t)J * 18 + (shape_threads / WAVE) * SHAPE_TERMS) * sizeof(float);
        hipLaunchKernelGGL(k_shape_bwd, dim3(B), dim3(shape_threads), lds, stream, g->d_verts, g->d_joints,
                           m->static_joints ? nullptr : g->d_Jrest, sv->A, m->skin_idx, m->skin_w, m->jreg_colptr,
                           m->jreg_row, m->jreg_cval, m->shapedirs, dbeta_frame, bshape, g->d_trans, g->d_del_v, V, J, nBu, regress,
                           in->trans_after_joints ? 1 : 0, g->up_v_shaped, nS);
        SMIL_LAUNCH_CHECK();
    }
    return SMIL_OK;
}
