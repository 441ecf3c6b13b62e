// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE for the access shapes of the tile kernel's record streams
// (MI355X_MICROARCH.md, HBM section: "other access widths are uncalibrated: calibrate on a known byte count in your own
// access pattern").  Single-wave workgroups, each streaming its own contiguous 1.5 MB slice of a 3 GiB buffer (no reuse,
// far beyond L2 + Infinity Cache): 4 bytes per lane (the record sweeps), 16 bytes per lane (the documented x2 case),
// 4-byte stores, and 4-byte ballot-compacted stores (a random ~60 % of the lanes store to consecutive slots).
//   hipcc --offload-arch=gfx950 -O3 -o fetch_calib tools/dbg/fetch_calib.hip
//   rocprofv3 --pmc FETCH_SIZE --output-format csv -d out/f -o f -- ./fetch_calib
//   rocprofv3 --pmc WRITE_SIZE --output-format csv -d out/w -o w -- ./fetch_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr size_t SLICE_WORDS = 393216;  // 1.5 MB per workgroup

__global__ void __launch_bounds__(64) read4(const uint32_t *__restrict__ p, uint32_t *__restrict__ sink) {
    const uint32_t *s = p + (size_t)blockIdx.x * SLICE_WORDS;
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < SLICE_WORDS; i += 256) {
        acc ^= s[i]; acc ^= s[i + 64]; acc ^= s[i + 128]; acc ^= s[i + 192];
    }
    if (acc == 0x12345u) sink[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(64) read16(const uint4 *__restrict__ p, uint32_t *__restrict__ sink) {
    const uint4 *s = p + (size_t)blockIdx.x * (SLICE_WORDS / 4);
    uint32_t acc = 0;
    for (uint32_t i = threadIdx.x; i < SLICE_WORDS / 4; i += 128) {
        const uint4 a = s[i], b = s[i + 64];
        acc ^= a.x ^ a.y ^ a.z ^ a.w ^ b.x ^ b.y ^ b.z ^ b.w;
    }
    if (acc == 0x12345u) sink[blockIdx.x] = acc;
}
__global__ void __launch_bounds__(64) write4(uint32_t *__restrict__ p) {
    uint32_t *s = p + (size_t)blockIdx.x * SLICE_WORDS;
    for (uint32_t i = threadIdx.x; i < SLICE_WORDS; i += 64) s[i] = i;
}
// the pass-1 store shape: every step a pseudo-random subset of the lanes appends to the slice's cursor
__global__ void __launch_bounds__(64) write4_compact(uint32_t *__restrict__ p, uint32_t *__restrict__ counts) {
    uint32_t *s = p + (size_t)blockIdx.x * SLICE_WORDS;
    uint32_t cursor = 0, rng = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 1u;
    while (cursor + 64 <= SLICE_WORDS) {
        rng = rng * 1664525u + 1013904223u;
        const bool on = (rng >> 24) < 154u;  // ~60 %
        const unsigned long long m = __ballot(on);
        const uint32_t slot = cursor + __builtin_amdgcn_mbcnt_hi((uint32_t)(m >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)m, 0u));
        if (on) s[slot] = rng;
        cursor += (uint32_t)__popcll(m);
    }
    if (threadIdx.x == 0) counts[blockIdx.x] = cursor;
}

int main() {
    const int n_wg = 2048;  // 3 GiB
    const size_t bytes = (size_t)n_wg * SLICE_WORDS * 4;
    uint32_t *buf, *sink;
    CHECK(hipMalloc(&buf, bytes));
    CHECK(hipMalloc(&sink, n_wg * 4));
    CHECK(hipMemset(buf, 1, bytes));
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(read4, dim3(n_wg), dim3(64), 0, 0, buf, sink);
    hipLaunchKernelGGL(read16, dim3(n_wg), dim3(64), 0, 0, (const uint4 *)buf, sink);
    hipLaunchKernelGGL(write4, dim3(n_wg), dim3(64), 0, 0, buf);
    hipLaunchKernelGGL(write4_compact, dim3(n_wg), dim3(64), 0, 0, buf, sink);
    CHECK(hipDeviceSynchronize());
    uint32_t *h = new uint32_t[n_wg];
    CHECK(hipMemcpy(h, sink, n_wg * 4, hipMemcpyDeviceToHost));
    size_t compact_words = 0;
    for (int i = 0; i < n_wg; ++i) compact_words += h[i];
    printf("bytes per kernel: read4 %zu read16 %zu write4 %zu write4_compact %zu\n", bytes, bytes, bytes, compact_words * 4);
    return 0;
}
