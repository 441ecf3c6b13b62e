// VALU issue rate on gfx950: cycles per wave64 instruction per SIMD for v_fma_f32, v_pk_fma_f32, v_exp_f32,
// v_cndmask, at 1, 2 and 4 waves per SIMD (one workgroup per CU: 4 / 8 / 16 waves).
//   hipcc -O3 --offload-arch=gfx950 tools/dbg/valu_rate.hip -o /tmp/valu_rate && /tmp/valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void k(float *out, int iters, unsigned long long *cyc) {
    float a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    v2f p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float b = 1.0001f, c = 0.5f;
    const v2f pb = {b, b}, pc = {c, c};
    __syncthreads();
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#define F(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(b), "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (MODE == 1) {
#define F(x) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(pb), "v"(pc));
            F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7) F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7)
#undef F
        } else if (MODE == 2) {
#define F(x) asm volatile("v_exp_f32 %0, %0" : "+v"(x));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else if (MODE == 3) {
#define F(x) asm volatile("v_max_f32 %0, %0, %1" : "+v"(x) : "v"(c));
            F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7) F(a0) F(a1) F(a2) F(a3) F(a4) F(a5) F(a6) F(a7)
#undef F
        } else {
#define F(x) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(x) : "v"(pb));
            F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7) F(p0) F(p1) F(p2) F(p3) F(p4) F(p5) F(p6) F(p7)
#undef F
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y + p4.x + p5.y + p6.x + p7.y;
}

int main() {
    float *out;
    unsigned long long *cyc, h;
    (void)hipMalloc(&out, 256 * 1024 * 4);
    (void)hipMalloc(&cyc, 8);
    const char *names[] = {"v_fma_f32", "v_pk_fma_f32", "v_exp_f32", "v_max_f32", "v_pk_mul_f32"};
    const int iters = 20000;
    for (int mode = 0; mode < 5; ++mode)
        for (int threads : {256, 512, 1024}) {
            for (int rep = 0; rep < 2; ++rep) {
                hipEvent_t e0, e1;
                (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
                (void)hipEventRecord(e0, 0);
                if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, out, iters, cyc);
                (void)hipEventRecord(e1, 0);
                (void)hipEventSynchronize(e1);
                float ms;
                (void)hipEventElapsedTime(&ms, e0, e1);
                (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
                const double waves_per_simd = threads / 256.0;
                if (rep == 1)
                    printf("%-13s %4d threads/CU (%.0f waves/SIMD): %.2f cycles per instr per wave, %.2f cycles per instr per SIMD, %.3f ms (clock %.2f GHz)\n",
                           names[mode], threads, waves_per_simd, (double)h / (iters * 16.0), (double)h / (iters * 16.0 * waves_per_simd), ms,
                           (double)h / (ms * 1e6));
            }
        }
    return 0;
}
