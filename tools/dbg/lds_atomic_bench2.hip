#include <hip/hip_runtime.h>
#include <cstdio>
// LDS atomic cost vs number of active lanes and type
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long *out, int iters, int active) {
    __shared__ unsigned long long mem[1024];
    const int lane = threadIdx.x;
    for (int i = lane; i < 1024; i += 64) mem[i] = 0;
    __syncthreads();
    int addr = lane;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (lane < active) {
            if (MODE == 0) atomicAdd(reinterpret_cast<float *>(mem) + addr, 1.0f);
            else if (MODE == 1) atomicAdd(&mem[addr], 1ull);
            else if (MODE == 2) atomicAdd(reinterpret_cast<double *>(mem) + addr, 1.0);
            else if (MODE == 3) atomicAdd(reinterpret_cast<unsigned int *>(mem) + addr, 1u);
            else if (MODE == 4) atomicMax(reinterpret_cast<unsigned int *>(mem) + addr, (unsigned)i);
        }
        addr = (addr + 64) & 1023;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (mem[lane] == 0xdeadbeef) out[1] = 1;
}
int main() {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    const char *names[] = {"f32 add", "u64 add", "f64 add", "u32 add", "u32 max"};
    for (int mode = 0; mode < 5; ++mode)
        for (int active : {4, 16, 64})
            for (int blocks : {1, 256 * 16}) {
                const int iters = 4096;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
                    if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(blocks), dim3(64), 0, 0, d, iters, active);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("%-8s active %2d blocks %5d: %.1f cycles per wave-instruction\n", names[mode], active, blocks, (double)h[0] / iters);
            }
    return 0;
}
