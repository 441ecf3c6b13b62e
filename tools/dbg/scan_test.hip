#include <hip/hip_runtime.h>
#include <cstdio>
template <bool IS_MAX>
__device__ __forceinline__ int wave_scan(int x) {
#define SCAN_STEP(ctrl, rows) { const int t_ = __builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xF, false); x = IS_MAX ? max(x, t_) : x + t_; }
    SCAN_STEP(0x111, 0xF) SCAN_STEP(0x112, 0xF) SCAN_STEP(0x114, 0xF) SCAN_STEP(0x118, 0xF)
    SCAN_STEP(0x142, 0xA) SCAN_STEP(0x143, 0xC)
#undef SCAN_STEP
    return x;
}
__global__ void k(const int *in, int *out) {
    const int lane = threadIdx.x;
    out[lane] = wave_scan<false>(in[lane]);
    out[64 + lane] = wave_scan<true>(in[64 + lane]);
}
int main() {
    int h[128], o[128];
    for (int i = 0; i < 64; ++i) { h[i] = (i * 7) % 5; h[64 + i] = (i % 9 == 0) ? i + 1 : 0; }
    int *d, *e; hipMalloc(&d, 512); hipMalloc(&e, 512);
    hipMemcpy(d, h, 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, 512, hipMemcpyDeviceToHost);
    int bad = 0, run = 0, mx = 0;
    for (int i = 0; i < 64; ++i) { run += h[i]; mx = h[64 + i] > mx ? h[64 + i] : mx; if (o[i] != run || o[64 + i] != mx) { if (bad < 10) printf("lane %d: sum %d (want %d) max %d (want %d)\n", i, o[i], run, o[64+i], mx); ++bad; } }
    printf("bad %d\n", bad);
    return 0;
}
