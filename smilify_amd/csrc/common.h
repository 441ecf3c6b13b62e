// Shared internals of libsmilfit (gfx950).  Not installed; the public surface is include/smilfit.h.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "smilfit.h"

#define WAVE 64

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
    int *bone_ptr = nullptr;      // skin weights by bone (J+1)
    int *bone_vid = nullptr;
    float *bone_w = nullptr;
    float *J_static = nullptr;    // (J,3)
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
