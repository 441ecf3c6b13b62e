#include <hip/hip_runtime.h>
#include <cstdio>
// LDS atomic throughput: cycles per wave-instruction for different address patterns, 1 and 16 waves per CU
template <int MODE>
__global__ void __launch_bounds__(64) k(unsigned long long *out, int iters, int pattern) {
    __shared__ unsigned int mem[2048];
    const int lane = threadIdx.x;
    for (int i = lane; i < 2048; i += 64) mem[i] = 0;
    __syncthreads();
    int addr;
    if (pattern == 0) addr = lane;                 // conflict-free
    else if (pattern == 1) addr = lane >> 2;       // 4 lanes per address
    else if (pattern == 2) addr = lane >> 4;       // 16 lanes per address
    else if (pattern == 3) addr = 0;               // all same
    else addr = (lane * 33) & 2047;                // scattered, bank-conflict free
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) atomicAdd(&mem[addr], 1u);
        else if (MODE == 1) atomicAdd(reinterpret_cast<float *>(&mem[addr]), 1.0f);
        else if (MODE == 2) mem[addr] = mem[addr] + 1u;   // plain read-modify-write
        else atomicOr(&mem[addr], 1u << (i & 31));
        addr = (addr + 64) & 2047;
    }
    __syncthreads();
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (lane == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (mem[lane] == 0xdeadbeef) out[1] = 1;
}
int main() {
    unsigned long long *d, h[2];
    (void)hipMalloc(&d, 16);
    const char *names[] = {"u32 add", "f32 add", "plain rmw", "u32 or"};
    for (int mode = 0; mode < 4; ++mode)
        for (int pattern = 0; pattern < 5; ++pattern)
            for (int blocks : {1, 256 * 16}) {
                const int iters = 4096;
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(blocks), dim3(64), 0, 0, d, iters, pattern);
                    if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(blocks), dim3(64), 0, 0, d, iters, pattern);
                    if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(blocks), dim3(64), 0, 0, d, iters, pattern);
                    if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(blocks), dim3(64), 0, 0, d, iters, pattern);
                    (void)hipDeviceSynchronize();
                }
                (void)hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
                printf("%-10s pattern %d blocks %5d: %.1f cycles per wave-instruction\n", names[mode], pattern, blocks, (double)h[0] / iters);
            }
    return 0;
}
