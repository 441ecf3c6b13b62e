// Shared internals of libsmilfit (gfx950).  Not installed; the public surface is include/smilfit.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "smilfit.h"

#define WAVE 64
#define BONE_WAVES 8  // waves of the workgroup that walks a frame's bone lists (k_lbs_bwd_ndc: 512 threads)

void smil_set_error(const char *fmt, ...);

#define SMIL_HIP(expr)                                                                              \
    do {                                                                                            \
        hipError_t _e = (expr);                                                                     \
        if (_e != hipSuccess) {                                                                     \
            smil_set_error("%s:%d: %s failed: %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return SMIL_E_DEVICE;                                                                   \
        }                                                                                           \
    } while (0)

#define SMIL_REQUIRE(cond, ...)              \
    do {                                     \
        if (!(cond)) {                       \
            smil_set_error(__VA_ARGS__);     \
            return SMIL_E_INVALID;           \
        }                                    \
    } while (0)

#define SMIL_LAUNCH_CHECK()                                                                        \
    do {                                                                                            \
        hipError_t _e = hipGetLastError();                                                          \
        if (_e != hipSuccess) {                                                                     \
            smil_set_error("%s:%d: kernel launch failed: %s", __FILE__, __LINE__, hipGetErrorString(_e)); \
            return SMIL_E_DEVICE;                                                                   \
        }                                                                                           \
    } while (0)

// Device-resident model constants.
struct SmilModel {
    int V = 0, F = 0, J = 0, nB = 0;
    int max_depth = 0;
    int max_valence = 0;          // most faces sharing one vertex (bounds the gradient a vertex can receive, raster.hip)
    bool static_joints = false;
    int jreg_nnz = 0;
    int bone_nnz = 0;
    float *v_template = nullptr;  // (V,3)
    float *shapedirs = nullptr;   // (nB,3V)
    int *faces = nullptr;         // (F,3)
    int *parents = nullptr;       // (J)
    int *depth = nullptr;         // (J)
    uint32_t *skin_idx = nullptr; // (V) four u8 bone ids packed little-endian
    float4 *skin_w = nullptr;     // (V)
    int *jreg_rowptr = nullptr;   // CSR by joint
    int *jreg_col = nullptr;
    float *jreg_val = nullptr;
    int *jreg_colptr = nullptr;   // CSC by vertex (V+1)
    int *jreg_row = nullptr;      // joint ids
    float *jreg_cval = nullptr;
    int2 *jreg_vfirst = nullptr;  // (V) the first CSC entry of every vertex inline: {joint | entries << 16, weight bits} (most have <= 1)
    int *bone_ptr = nullptr;      // skin weights by bone (J+1)
    int *bone_vid = nullptr;
    float *bone_w = nullptr;
    float *J_static = nullptr;    // (J,3)
    float *jreg_shape = nullptr;  // (nB,J,3) = J_regressor @ shapedirs[k]: d beta through the rest joints without a pass over the vertices
    int *bone_order = nullptr;    // (bone_slots) the bone lists dealt to BONE_WAVES waves, longest processing time first: slot w + k BONE_WAVES
                                  // is wave w's k-th bone, -1 behind its last
    int bone_slots = 0;
    float *posedirs = nullptr;    // (9(J-1),3V) or null
    std::vector<void *> allocations;
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// ---- wave-level helpers (wave64) --------------------------------------------------------------
// Sum over the 64 lanes using DPP within rows of 16 and readlane across rows; result valid in all lanes.
__device__ __forceinline__ float wave_sum(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0));
    const float r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32));
    const float r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return (r0 + r1) + (r2 + r3);
}

// Sum over each row of 16 lanes (four independent sums per wave); result valid in every lane of the row.
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)); // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xF, 0xF, true)); // row_mirror
    return v;
}

__device__ __forceinline__ float wave_max(float v) {
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// Block-wide sum for blocks of up to 1024 threads; result valid in thread 0 (and all threads of wave 0).
__device__ __forceinline__ float block_sum(float v, float *smem /* >= 16 floats */) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6, nw = (blockDim.x + 63) >> 6;
    __syncthreads();
    if (lane == 0) smem[wid] = v;
    __syncthreads();
    float r = 0.f;
    if (wid == 0) {
        r = lane < nw ? smem[lane] : 0.f;
        r = wave_sum(r);
    }
    return r;
}

// Sums of up to TWELVE values per lane over the 64 lanes in 30 instructions (twelve wave_sum calls: 130).  gfx950's
// v_permlane32_swap / v_permlane16_swap exchange half-waves / odd and even 16-lane rows between two registers, so one swap and
// one add fold a PAIR of values to half the lanes each: 12 values -> 6 registers (component i in lanes 0-31, i + 6 in lanes
// 32-63) -> 3 registers (rows hold components i, i + 3, i + 6, i + 9), then four DPP adds inside the rows.
// On return every lane of row r (lanes 16 r .. 16 r + 15) holds the sum of component i + 3 r in q[i], i = 0 .. 2.
// (spelled in assembly: with the __builtin_amdgcn_permlane*_swap builtins hipcc 7.2 adds the FIRST result to itself - it loses the
// second, in-place updated operand; tools/dbg/sum12_test.hip checks this function against a serial sum)
__device__ __forceinline__ float swap32_add(float a, float b) {  // lanes 0-31: a[l] + a[l + 32]; lanes 32-63: b[l - 32] + b[l]
    asm("s_nop 1\n\tv_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
__device__ __forceinline__ float swap16_add(float a, float b) {  // rows 0, 2: a's row + a's next row; rows 1, 3: b's previous row + b's row
    asm("s_nop 1\n\tv_permlane16_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    return a + b;
}
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__)
#error "libsmilfit is written for gfx950 (MI355X) only: v_permlane32_swap / v_permlane16_swap and the tile kernel's LDS and register budgets have no other target"
#endif
__device__ __forceinline__ void wave_sum12(const float (&v)[12], float (&q)[3]) {
    float h[6];
#pragma unroll
    for (int i = 0; i < 6; ++i) h[i] = swap32_add(v[i], v[i + 6]);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        float x = swap16_add(h[i], h[i + 3]);
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0xB1, 0xF, 0xF, true));   // quad_perm [1,0,3,2]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x4E, 0xF, 0xF, true));   // quad_perm [2,3,0,1]
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x141, 0xF, 0xF, true));  // row_half_mirror
        x += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), 0x140, 0xF, 0xF, true));  // row_mirror
        q[i] = x;
    }
}

// ---- FoV-perspective camera of image n (project.hip; also read by the fused LBS backward in lbs.hip) ----
#define SMIL_ZNEAR 0.001f  // Renderer.DEFAULT_ZNEAR (p3d_renderer.py:24)

struct CamParams {
    float R[9];
    float T[3];
    float k00, k11, tanh_;  // tan(fov/2)
};

__device__ __forceinline__ CamParams load_camera(const SmilCameras &c, int n) {
    CamParams p;
    const float *R = c.R + (size_t)(n % c.nR) * 9;
    const float *T = c.T + (size_t)(n % c.nT) * 3;
    for (int i = 0; i < 9; ++i) p.R[i] = R[i];
    for (int i = 0; i < 3; ++i) p.T[i] = T[i];
    const float fov = c.fov[n % c.nFov];
    const float asp = c.aspect ? c.aspect[n % c.nAspect] : 1.0f;
    const float t = tanf((fov * 0.017453292519943295f) / 2.0f);
    const float max_y = t * SMIL_ZNEAR;
    const float max_x = max_y * asp;
    p.k00 = 2.0f * SMIL_ZNEAR / (max_x - (-max_x));
    p.k11 = 2.0f * SMIL_ZNEAR / (max_y - (-max_y));
    p.tanh_ = t;
    return p;
}

