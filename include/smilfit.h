/*
 * libsmilfit - C ABI of the MI355X-native SMIL fitting inner loop.
 *
 * The reference (FabianPlum/SMILify) has no FFI boundary for this path: the path is three
 * Python classes over torch + pytorch3d.  This header is the boundary a binding would use
 * instead; every entry point names the reference code it replaces.  All pointers are DEVICE
 * pointers unless marked "host"; all arrays are dense, row-major fp32 / int32; `stream` is a
 * hipStream_t passed as void*.  Every function returns 0 on success or a negative SMIL_E_*
 * code (message via smil_last_error()); nothing throws, nothing synchronises the device,
 * nothing allocates after smil_model_create() - the caller owns every buffer.
 */
#ifndef SMILFIT_H_
#define SMILFIT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SMIL_OK 0
#define SMIL_E_INVALID (-1)  /* bad argument / shape */
#define SMIL_E_DEVICE (-2)   /* HIP runtime error */
#define SMIL_E_UNSUPPORTED (-3)

#define SMIL_MAX_BONES 4     /* bones per vertex in the skin table */
#define SMIL_MAX_JOINTS 256
#define SMIL_MAX_BETAS 64
#define SMIL_MAX_FACES_PER_PIXEL 128

typedef struct SmilModel SmilModel; /* opaque: device-resident model constants */

/* Host-side description of a model.  Replaces the buffers built in SMAL.__init__
 * (reference smal_model/smal_torch.py:104-196).  All pointers are HOST pointers. */
typedef struct {
    int32_t V, F, J, nB;
    const float *v_template;    /* (V,3) */
    const float *shapedirs;     /* (nB,3V), inner index v*3+c */
    const int32_t *faces;       /* (F,3) */
    const int32_t *parents;     /* (J,), parents[0] = -1, parents[i] < i */
    const int32_t *skin_idx;    /* (V,4) bone ids (padding: id 0, weight 0) */
    const float *skin_w;        /* (V,4) */
    const int32_t *jreg_rowptr; /* (J+1,) joint regressor, CSR by joint */
    const int32_t *jreg_col;    /* (nnz,) vertex ids */
    const float *jreg_val;      /* (nnz,) */
    int32_t static_joints;      /* config.STATIC_JOINT_LOCATIONS (smal_torch.py:175,257,343) */
    const float *J_static;      /* (J,3) when static_joints */
    const float *posedirs;      /* (9(J-1),3V) pose blend shapes (smal_torch.py:178-190) or NULL when empty / all zero */
} SmilModelDesc;

int smil_model_create(const SmilModelDesc *desc, SmilModel **out);
void smil_model_destroy(SmilModel *m);
int smil_model_dims(const SmilModel *m, int32_t dims[4]); /* V,F,J,nB */
const char *smil_last_error(void);
const char *smil_version(void);   /* "smilfit 0.3 (gfx950)".  0.3 = the layout of rounds 5 - 6: SmilLbsGrads carries clip_depth in front of
                                    * beta_rows (which must hold 2 * B * nB_used + 16 floats), SmilRasterSettings ends in {tie_rule, clip_depth,
                                    * image0}, smil_window_terms exists.  Callers zero-initialise every struct they pass and rebuild against
                                    * this header when the number changes: there is no binary compatibility across it. */

/* ------------------------------------------------------------------------------------------
 * Linear blend skinning.  Replaces SMAL.__call__ (smal_torch.py:198-370) including
 * batch_rodrigues (batch_lbs.py:31-50) and batch_global_rigid_transformation (batch_lbs.py:75-197).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t B;
    int32_t shared_beta;        /* 1: beta is one (nB_used,) row used by every frame */
    int32_t nB_used;            /* beta.shape[1] in the reference call (<= nB) */
    const float *beta;          /* (B,nB_used) or (nB_used,) */
    const float *theta;         /* (B,J,3) axis-angle, or NULL when Rs_in is given */
    const float *Rs_in;         /* (B,J,3,3) rotation matrices passed directly (smal_torch.py:288) */
    const float *logscale;      /* betas_logscale (B,J,3) / (J,3) / NULL */
    int32_t logscale_shared;    /* 1: one (J,3) table for every frame */
    const float *btrans;        /* betas_trans, same conventions */
    int32_t btrans_shared;
    const float *trans;         /* (B,3) or NULL */
    int32_t trans_after_joints; /* 0: SMAL.__call__ semantics - joints are regressed from the translated
                                   vertices (smal_torch.py:340-351); 1: SMALFitter semantics - trans is added
                                   to verts AND joints after regression (fitter.py:280-281) */
    const float *del_v;         /* (B,V,3) or NULL */
    const float *v_template;    /* (V,3) override or NULL */
    int32_t propagate_scaling;  /* batch_lbs.py:163-168 */
    int32_t allow_limb_scaling; /* config.ALLOW_LIMB_SCALING (batch_lbs.py:123) */
    const float *theta_mask;    /* (J,3) or NULL: theta is used as theta * mask (global_mask / rotation_mask of
                                   SMALFitter.forward, fitter.py:242-243) without a masked copy; d_theta is the gradient
                                   with respect to the MASKED pose */
} SmilLbsInputs;

typedef struct {
    float *v_shaped; /* (nS,V,3); nS = 1 when shared_beta && !del_v, else B */
    float *J_rest;   /* (nS,J,3) rest joints */
    float *Rs;       /* (B,J,3,3) */
    float *G;        /* (B,J,3,4) world transforms, saved for backward */
    float *A;        /* (B,J,3,4) relative skinning transforms */
    float *new_J;    /* (B,J,3) = SMAL.J_transformed */
    float *verts;    /* (B,V,3) */
    float *joints;   /* (B,J,3) */
    float *v_posed;  /* (B,V,3) v_shaped + pose blend shapes; required iff the model has posedirs, else NULL */
} SmilLbsOutputs;

int smil_lbs_forward(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, void *stream);

/* Depth gradients of the end points of edges that cross the rasteriser's clipping plane (pytorch3d clip_faces differentiates the
 * crossing point through w = (z_a - z_clip) / (z_a - z_b) and the explicit depth factors; the reference leaves clipping on,
 * p3d_renderer.py:36-47).  A sparse side channel next to d_ndc (N,V,2): the silhouette backward entry points append, per image that
 * has cut faces, entries {vertex, d loss / d z_view[vertex]} and record the image's run in `range`; smil_lbs_backward_ndc and
 * smil_clip_depth_backward carry them through the camera (z_view = X_world . R[:,2] + T_z) into the world-space vertex gradient.
 * All buffers are the caller's.  No BASELINE configuration cuts a face: the channel stays empty there and costs one word read. */
typedef struct {
    int32_t *vertex;      /* (capacity) */
    float *dz;            /* (capacity) */
    uint32_t *range;      /* (N_total, 2): first entry and number of entries of every image of the caller's batch */
    uint32_t *counter;    /* [0] entries in use (reset by the call whose image0 == 0), [1] entries that did not fit (dropped, counted) */
    int32_t capacity;
} SmilClipDepth;

typedef struct {
    const float *d_verts;  /* (B,V,3) upstream gradient or NULL */
    const float *d_joints; /* (B,J,3) upstream gradient or NULL */
    float *d_beta;         /* (B,nB_used) or (nB_used,); NULL to skip */
    float *d_theta;        /* (B,J,3); NULL to skip (ignored when Rs_in was used) */
    float *d_logscale;     /* (B,J,3) or (J,3) when shared; NULL to skip */
    float *d_btrans;       /* same */
    float *d_trans;        /* (B,3); NULL to skip */
    /* scratch owned by the caller */
    float *d_A;            /* (B,J,3,4) */
    float *d_Jrest;        /* (B,J,3) */
    float *d_Rs;           /* (B,J,3,3) */
    float *d_vposed;       /* (B,V,3): required iff the model has posedirs */
    float *d_posefeat;     /* (B,9(J-1)): required iff the model has posedirs */
    float *d_del_v;        /* (B,V,3) or NULL: gradient on the per-frame vertex offsets del_v (smal_torch.py:244-248);
                              also the gradient on v_shaped / v_template rows of each frame */
    float *d_Rs_in;        /* (B,J,3,3) or NULL: gradient on the rotation matrices when theta was given as
                              matrices (Rs_in; smal_torch.py:288-289) */
    int32_t accumulate_shared_beta; /* shared_beta only: ADD the sum over frames to d_beta (caller zeroes it or holds
                              other terms there) instead of overwriting it */
    const float *up_Rs;       /* (B,J,3,3) or NULL: upstream gradient on the rotation matrices the forward returned
                              (SMAL.__call__ hands Rs to its caller, smal_torch.py:367-370); flows to d_theta / d_Rs_in */
    const float *up_v_shaped; /* (nS,V,3) or NULL: upstream gradient on the returned v_shaped; flows to d_beta and d_del_v */
    const SmilClipDepth *clip_depth; /* or NULL (smil_lbs_backward_ndc only): depth gradients from the rasteriser's clipping plane,
                              added to the frame's vertex gradient through the cameras */
    float *beta_rows;         /* scratch, 2 * B * nB_used + 16 floats: required iff shared_beta and d_beta.  The kernels leave one partial
                              sum per block there and the last block to finish adds them in a fixed order; the word behind the rows
                              counts the finished blocks of THIS call (cleared by the call's own kernels: calls with different
                              scratch may run on different streams) */
} SmilLbsGrads;            /* every output is overwritten; tables shared by all frames (shared_beta,
                              logscale_shared, btrans_shared) receive the sum over frames.  All of these sums are taken in a
                              fixed order: two calls on the same inputs return the same bits (the shared shape gradient - the
                              quantity ranks all-reduce - included, since round 4) */

int smil_lbs_backward(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *saved,
                      const SmilLbsGrads *g, void *stream);

/* ------------------------------------------------------------------------------------------
 * Cameras + projection.  Replaces FoVPerspectiveCameras as configured by Renderer
 * (p3d_renderer.py:34-38,112-120) and transform_points_screen(...)[..., [1,0]] (:137).
 * Image n = frame * views + view.  A table with k rows is indexed n % k (k = 1, views or N).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N;          /* images = frames * views */
    int32_t views;      /* views per frame (>= 1) */
    int32_t S;          /* square image side */
    const float *R;     /* (nR,3,3) row-vector convention: X_view = X_world R + T */
    int32_t nR;
    const float *T;     /* (nT,3) */
    int32_t nT;
    const float *fov;   /* (nFov,) degrees */
    int32_t nFov;
    const float *aspect; /* (nAspect,) or NULL = 1 */
    int32_t nAspect;
} SmilCameras;

/* pts (frames,P,3) world -> ndc (N,P,3) = (x_ndc, y_ndc, z_view); yx (N,P,2) = (y_s, x_s) px. Either
 * output may be NULL. */
int smil_project(const SmilCameras *cam, const float *pts, int32_t P, float *ndc, float *yx, void *stream);

/* Two point sets through the same cameras in one launch (the vertices for the rasteriser and the joints for the 2-D
 * loss of one fit iteration); arguments per set as smil_project. */
int smil_project2(const SmilCameras *cam, const float *pts_a, int32_t Pa, float *ndc_a, float *yx_a, const float *pts_b,
                  int32_t Pb, float *ndc_b, float *yx_b, void *stream);

/* smil_project_backward for two point sets in one launch (d_pts of both overwritten; d_fov_img added to).  d_ndc_scale_a:
 * as in smil_project_backward, for set a. */
int smil_project_backward2(const SmilCameras *cam, const float *pts_a, int32_t Pa, const float *d_ndc_a, const float *d_yx_a,
                           float *d_pts_a, const float *pts_b, int32_t Pb, const float *d_ndc_b, const float *d_yx_b,
                           float *d_pts_b, float *d_fov_img, const float *d_ndc_scale_a, void *stream);

/* Backward of smil_project.  d_ndc (N,P,2) and/or d_yx (N,P,2) -> d_pts (frames,P,3) (summed over
 * views; overwritten unless accumulate) and d_fov_img (N,): per-image raw sums
 * sum_p (d x_ndc * x_ndc + d y_ndc * y_ndc), ATOMICALLY ADDED (caller zeroes once, then may call this for
 * several point sets).  smil_fov_reduce turns them into d_fov (nFov,), overwritten.
 * d_ndc_scale (N,) or NULL: the per-image decode factors smil_silhouette_l1_fused returned with a d_ndc it left packed
 * (> 0: the row is 64-bit packed fixed point, times this factor; 0: plain floats; < 0: a packed row of zeros). */
int smil_project_backward(const SmilCameras *cam, const float *pts, int32_t P, const float *d_ndc,
                          const float *d_yx, float *d_pts, float *d_fov_img, int32_t accumulate, const float *d_ndc_scale,
                          void *stream);
int smil_fov_reduce(const SmilCameras *cam, const float *d_fov_img, float *d_fov, void *stream);

/* smil_lbs_forward followed by the projection of its vertices and joints through `cam` (N = in->B * cam->views images):
 * ndc (N,V,3) as smil_project(verts) and yx (N,J,2) as smil_project(joints); either may be NULL.  Replaces SMAL.__call__
 * + Renderer's two projections of one fit iteration (fitter.py:270-290, p3d_renderer.py:137-146).  Skinning, joint regression and
 * both projections are ONE kernel per frame and `verts` is written once and not read back where the model's joints are static
 * (nothing is gathered from the frame's vertices: any mesh size) or the frame's vertices fit half a CU's LDS (3 V floats <= 80 KB);
 * otherwise the separate kernels run.  Outputs are those of the separate calls. */
int smil_lbs_forward_project(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *out, const SmilCameras *cam,
                             float *ndc, float *yx, void *stream);

/* smil_lbs_backward taking its upstream gradients on the IMAGE PLANE, as the fit iteration has them: d_ndc (N,V,2) on the
 * projected vertices (rows may be the packed fixed point smil_silhouette_l1_fused leaves, with d_ndc_scale (N,) as in
 * smil_project_backward) and d_yx_joints (N,J,2) on the projected joints (y, x) in pixels; either may be NULL.  One kernel per
 * frame does what smil_project_backward2 + smil_lbs_backward do (backward of p3d_renderer.py:137-146 into the backward of
 * smal_torch.py:240-351) without writing the (B,V,3) vertex gradient to memory.  g->d_verts, g->d_joints and g->d_del_v
 * must be NULL; d_joints (B,J,3) receives the world-space joint gradient (an output); d_fov_img (N,) or NULL is ADDED to
 * as by smil_project_backward.  saved->verts and saved->joints are read.  Results equal the two-call route up to fp32
 * summation order.  smil_lbs_backward_ndc_supported: 1 when this entry handles the model with nB_used shape coefficients (<= 9)
 * and `views` views per frame (<= 32): no pose blend shapes, and either the frame's vertex gradient AND rest vertices fit half a
 * CU's LDS (24 V bytes <= 80 KB: two workgroups of 512 threads per CU) or the vertex gradient alone fits the CU's whole LDS
 * (12 V bytes <= 160 KB, V <= ~13 000: one workgroup of 1024 threads per CU, rest vertices gathered from memory). */
int smil_lbs_backward_ndc(const SmilModel *m, const SmilLbsInputs *in, const SmilLbsOutputs *saved, const SmilLbsGrads *g,
                          const SmilCameras *cam, const float *d_ndc, const float *d_ndc_scale, const float *d_yx_joints,
                          float *d_joints, float *d_fov_img, void *stream);
int smil_lbs_backward_ndc_supported(const SmilModel *m, int32_t nB_used, int32_t views);
/* The separate-kernel route's counterpart of SmilLbsGrads.clip_depth: d_verts (B,V,3) += the depth gradients of `cd` carried through
 * the cameras (image n belongs to frame n / views).  Call after smil_project_backward{,2}. */
int smil_clip_depth_backward(const SmilCameras *cam, const SmilClipDepth *cd, int32_t N, int32_t V, float *d_verts, void *stream);

/* ------------------------------------------------------------------------------------------
 * Soft silhouette.  Replaces MeshRasterizer(naive, K faces per pixel, blur) + SoftSilhouetteShader
 * (p3d_renderer.py:41-52,142-146; arithmetic in un-vendored pytorch3d 0.7.8).
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    float blur_radius;        /* NDC^2; reference: log(1/1e-4 - 1) * 1e-4 */
    float sigma;              /* 1e-4 */
    int32_t faces_per_pixel;  /* K = 100 */
    float z_clip;             /* MeshRasterizer's z_clip_value = znear / 2 = 5e-4: faces whose three vertices are all
                                 nearer than this are culled, faces that cross it are cut there (clip_faces) */
    int32_t tie_rule;         /* which faces a pixel keeps among EQUAL depths at its K-th place.  SMIL_TIE_DEPTH_FACE_ID (0, default):
                                 the smallest face ids - order independent, what the tile kernel computes.  SMIL_TIE_REFERENCE_QUEUE (1):
                                 what pytorch3d's unsorted K-queue ends up with when it visits the faces in index order (the
                                 reference's rasteriser, p3d_renderer.py:42-47): such pixels (~2 % of the truncated ones) are replayed
                                 one by one by a second kernel.  Measured difference on the L1 term: ~1e-5 relative at K = 100 */
    const SmilClipDepth *clip_depth; /* or NULL: where smil_silhouette_backward / smil_silhouette_l1_fused leave the depth gradients of
                                 cut edges' end points (NULL: dropped - the xy gradients are complete either way) */
    int32_t image0;           /* index of this call's first image in the caller's batch (`clip_depth->range` is indexed by it): a
                                 batch cut into several calls passes 0, N_1, N_1 + N_2, ... */
} SmilRasterSettings;
#define SMIL_TIE_DEPTH_FACE_ID 0
#define SMIL_TIE_REFERENCE_QUEUE 1

/* Caller-owned scratch for N images of side S: per-face tile boxes / depth ranges, the tile work list, and the pair-record
 * streams of the resident workgroups (about 1.6 MB each, 16 per CU: 6.6 GB once N * tiles exceeds that many - size the
 * buffer once and reuse it).  Meshes with more than 65536 faces are rejected (SMIL_E_INVALID). */
size_t smil_raster_workspace_bytes(const SmilModel *m, int32_t N, int32_t S);

/* Counters of the most recent rasteriser call that used `workspace` with the same N, copied to out4[4] (synchronises the
 * stream): [0] faces that cross z_clip (one or two vertices nearer than znear / 2): cut at the plane like pytorch3d's
 * clip_faces, which p3d_renderer.py:36-47 leaves on - the front part is rendered as one or two extra triangles whose new
 * vertices hand their gradient back to the cut edge's end points: to their xy through d_ndc, to their DEPTHS - the crossing
 * point also moves with z_a and z_b - through SmilRasterSettings.clip_depth (a sparse list; dropped when NULL).  [1] touched 8x8 tiles.  [2] faces that cross the plane beyond the capacity of the per-image clip tables - 1024 cut faces per
 * image, each with up to two front-part triangles and two new vertices: rendered whole, or not at all when a vertex is nearer than 1e-8 - the one case in which a call still deviates.
 * [3] pixels whose tie group at the K-th depth was cut by K and that were therefore replayed through the reference's queue
 * (tie_rule = SMIL_TIE_REFERENCE_QUEUE only; 0 otherwise). */
int smil_raster_stats(const SmilModel *m, int32_t N, const void *workspace, void *stream, uint32_t *out4);

/* verts_ndc (N,V,3) -> sil (N,S,S) */
int smil_silhouette_forward(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                            const SmilRasterSettings *rs, float *sil, void *workspace, void *stream);
/* grad_sil (N,S,S) -> d_ndc (N,V,2) (overwritten) */
int smil_silhouette_backward(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                             const SmilRasterSettings *rs, const float *grad_sil, float *d_ndc,
                             void *workspace, void *stream);
/* Fused forward + L1 against a target + backward, no silhouette materialised (SMALFitter path,
 * fitter.py:332-333): loss_img[n] = sum_px |sil - target|, d_ndc (N,V,2) = d(sum_n pix_scale[n] *
 * loss_img[n]) / d ndc.  target_sum[n] = sum_px target (constant, computed once by the caller) lets
 * untouched tiles skip their target read.  target is (N,S,S) fp32, or uint8 holding binary {0,1} masks when
 * target_is_u8 (a quarter of the memory and read traffic).  sil_out may be NULL.  loss_img is bit-reproducible: the tiles'
 * terms are summed as 2^-32 fixed-point integers (any order of arrival) and added to target_sum[n] once.
 * From 64 images per call on (and an 8-byte aligned d_ndc), d_ndc is accumulated as 64-bit packed fixed point in the same
 * buffer: every (face, pixel) contribution is rounded once to 2^-30 of a per-image worst-case bound (vertex valence x largest
 * face pixel box x 0.4 |pix_scale| / sqrt(sigma)) and all further sums are integer adds - independent of the order in which
 * tiles finish and of how faces are grouped (bit-reproducible).  Relative to an image's largest gradient component that is
 * ~2e-6 on ordinary meshes and up to a few 1e-4 when one face fills the image (the bound grows with the largest face box);
 * smaller calls, and images with faces cut at z_clip, use float atomics (order-dependent last bits).
 * d_ndc_scale == NULL: packed rows are decoded in place before the call's work ends on the stream - the caller always sees
 * floats.  d_ndc_scale (N,) given: no decode pass; d_ndc_scale[n] says how image n's row is to be read (see
 * smil_project_backward, which takes the pair as it is and decodes while it reads). */
int smil_silhouette_l1_fused(const SmilModel *m, const float *verts_ndc, int32_t N, int32_t S,
                             const SmilRasterSettings *rs, const void *target, int32_t target_is_u8,
                             const float *target_sum, const float *pix_scale, float *loss_img, float *d_ndc,
                             float *sil_out, float *d_ndc_scale, void *workspace, void *stream);

/* Measurement hook (bench.py): when enabled, every launch of the tile kernel is bracketed by HIP events on its
 * launch stream; smil_profile_read synchronises those events and returns their summed duration + count. */
int smil_profile_enable(int32_t on);
int smil_profile_read(float *total_ms, int32_t *launches);

/* ------------------------------------------------------------------------------------------
 * Priors / joint loss / temporal / Adam.  Replaces the loss block of SMALFitter.forward
 * (fitter.py:292-333), get_temporal (:337-350) and the Adam step of optimize_to_joints.py:117-175.
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    int32_t N;            /* frames held by this rank */
    int32_t J;            /* joints incl. root */
    int32_t nB;
    int32_t window;       /* config.WINDOW_SIZE: losses are means per window, summed over windows */
    int32_t frame0;       /* global index of local frame 0 (multi-GPU shard offset) */
    int32_t N_total;      /* frames of the whole sequence */
    float w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay, w_temp;
    float limit;          /* joint limit half-width (0.01) */
    int32_t train_global, train_joints, train_trans; /* requires_grad of the per-frame groups
                                                        (optimize_to_joints.py:129-144): 0 -> gradient zeroed */
} SmilFitConfig;

/* objs is a 10-float accumulator, every entry ADDED to (caller zeroes once per iteration):
 * [0] joint  [1] limit  [2] pose  [3] splay  [4] betas  [5] sil_reproj
 * [6] temporal joint-rotations  [7] temporal global-rotation  [8] temporal translation  [9] unused */
#define SMIL_N_OBJS 10

/* Prior terms (limit, pose, splay, betas, temporal) and their gradients on the per-frame parameters.
 * pose (N,J,3): row 0 of each frame = global_rotation, rows 1.. = joint_rotations (UNMASKED parameters);
 * mask (J,3): row 0 = global_mask, rows 1.. = rotation_mask (fitter.py:213-219,242-243); trans (N,3);
 * betas (nB,).  halo_prev / halo_next: the (3J+3,) row [pose, trans] of the frame just before / after this
 * shard (NULL at the sequence ends).
 * accumulate = 1: d_pose holds gradients w.r.t. the MASKED pose on entry (from smil_lbs_backward) and the
 * final parameter gradients on exit ((in + prior) * mask * train flag); d_trans is added to.
 * accumulate = 0: overwritten.  d_betas (nB,) is ADDED to. */
int smil_prior_losses(const SmilFitConfig *cfg, const float *pose, const float *trans, const float *betas,
                      const float *mean_betas, const float *betas_prec, const float *mask,
                      const float *halo_prev, const float *halo_next, float *objs, float *d_pose,
                      float *d_trans, float *d_betas, int32_t accumulate, void *stream);

/* out[i][c] = in[i][c] * mask[c]  (masked pose fed to smil_lbs_forward) */
int smil_mask_rows(const float *in, const float *mask, int64_t rows, int32_t cols, float *out, void *stream);

/* smil_prior_losses + the silhouette objective (objs[5] += sum_n pix_scale[n] * loss_img[n]; loss_img NULL to skip;
 * fitter.py:332-333) + the reduction of the per-image fov sums of smil_project_backward to d_fov (cam->nFov,),
 * overwritten (d_fov NULL to skip): the tail of one fit iteration in ONE launch. */
int smil_fit_epilogue(const SmilFitConfig *cfg, const float *pose, const float *trans, const float *betas,
                      const float *mean_betas, const float *betas_prec, const float *mask, const float *halo_prev,
                      const float *halo_next, float *objs, float *d_pose, float *d_trans, float *d_betas, int32_t accumulate,
                      const float *loss_img, const float *pix_scale, int32_t n_img, const SmilCameras *cam,
                      const float *d_fov_img, float *d_fov, void *stream);

/* 2-D joint loss (fitter.py:283,292-296).  proj / d_proj (N*views,J,2) in (y,x) px over ALL model joints;
 * canon (Jc,) = config.CANONICAL_MODEL_JOINTS (NULL: the first Jc joints); target (N*views,Jc,2);
 * visibility (N*views,Jc) int32.  objs[0] ADDED, d_proj overwritten. */
int smil_joint_loss(const SmilFitConfig *cfg, int32_t views, int32_t Jc, const int32_t *canon, const float *proj,
                    const float *target, const int32_t *visibility, float *objs, float *d_proj, void *stream);

/* The six loss terms of SMALFitter.forward (fitter.py:292-333) for every WINDOW of this shard separately, computed from the buffers
 * one whole-batch iteration left behind (proj from smil_lbs_forward_project / smil_project, loss_img from smil_silhouette_l1_fused,
 * objs_total = the iteration's objs after smil_fit_epilogue): what the reference's per-window forward() calls of one epoch return
 * (optimize_to_joints.py:153-157), without evaluating the windows one by one.  objs_win (n_windows,6), overwritten:
 * [joint, limit, pose, splay, betas, sil_reproj].  proj NULL: no joint term; loss_img NULL: no silhouette term.  The shard must start
 * at a window boundary and n_windows = ceil(N / window). */
int smil_window_terms(const SmilFitConfig *cfg, int32_t views, int32_t Jc, const int32_t *canon, const float *proj,
                      const float *target, const int32_t *visibility, const float *pose, const float *mask,
                      const float *objs_total, const float *loss_img, const float *pix_scale, float *objs_win,
                      int32_t n_windows, void *stream);

/* Helpers of the fused silhouette term: pix_scale[n] = w_reproj / (b_w views S^2); target_sum[n] =
 * sum_px |target[n]| (once per fit); objs[5] += sum_n pix_scale[n] loss_img[n]. */
int smil_pix_scale(const SmilFitConfig *cfg, int32_t views, int32_t S, float *pix_scale, void *stream);
int smil_image_abs_sum(const void *images, int32_t is_u8, int32_t N, int32_t pixels, float *out, void *stream);
int smil_sil_objective(const float *loss_img, const float *pix_scale, int32_t N, float *objs, void *stream);

/* torch.optim.Adam semantics (no amsgrad, no weight decay). step = 1-based step count. */
int smil_adam_step(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                   float lr, float beta1, float beta2, float eps, int32_t step, void *stream);
/* The same update for up to SMIL_ADAM_MAX_TENSORS parameter tensors in ONE launch (a fit iteration updates five small
 * tensors; five launches cost more than the arithmetic).  Each tensor has its own learning rate and step count. */
#define SMIL_ADAM_MAX_TENSORS 8
typedef struct {
    float *param;
    const float *grad;
    float *exp_avg, *exp_avg_sq;
    int64_t n;
    float lr;
    int32_t step;             /* 1-based */
} SmilAdamTensor;
int smil_adam_step_multi(const SmilAdamTensor *tensors, int32_t count, float beta1, float beta2, float eps, void *stream);

/* Same update with the step count read from device memory: step = *step_dev - step_offset.  Lets a whole fit iteration be
 * captured once in a hipGraph and replayed (the host only bumps the counter - or the graph does, with an increment node). */
int smil_adam_step_dev(float *param, const float *grad, float *exp_avg, float *exp_avg_sq, int64_t n,
                       float lr, float beta1, float beta2, float eps, const int32_t *step_dev, int32_t step_offset,
                       void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SMILFIT_H_ */
