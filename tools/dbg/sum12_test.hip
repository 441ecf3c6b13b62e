// wave_sum12 (common.h) against a serial sum: hipcc --offload-arch=gfx950 -I smilify_amd/csrc -I include tools/dbg/sum12_test.hip -o /tmp/sum12 && /tmp/sum12
#include "common.h"
#include <cstdlib>
void smil_set_error(const char *, ...) {}
__global__ void k(const float *in, float *out) {
    float v[12], q[3];
    for (int i = 0; i < 12; ++i) v[i] = in[i * 64 + threadIdx.x];
    wave_sum12(v, q);
    if ((threadIdx.x & 15) == 0)
        for (int i = 0; i < 3; ++i) out[i + 3 * (threadIdx.x >> 4)] = q[i];
}
int main() {
    float h[12 * 64], ref[12] = {0}, *din, *dout, got[12];
    for (int i = 0; i < 12 * 64; ++i) { h[i] = (float)(rand() % 1000) * 0.01f; ref[i / 64] += h[i]; }
    (void)hipMalloc(&din, sizeof(h)); (void)hipMalloc(&dout, sizeof(got));
    (void)hipMemcpy(din, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, din, dout);
    (void)hipMemcpy(got, dout, sizeof(got), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < 12; ++i) { printf("%d: %.3f vs %.3f\n", i, got[i], ref[i]); bad += fabsf(got[i] - ref[i]) > 1e-2f; }
    printf("%s\n", bad ? "MISMATCH" : "ok");
    return bad;
}
