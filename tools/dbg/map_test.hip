#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define WAVE 64
template <bool IS_MAX>
__device__ __forceinline__ int wave_scan(int x) {
#define SCAN_STEP(ctrl, rows) { const int t_ = __builtin_amdgcn_update_dpp(0, x, ctrl, rows, 0xF, false); x = IS_MAX ? max(x, t_) : x + t_; }
    SCAN_STEP(0x111, 0xF) SCAN_STEP(0x112, 0xF) SCAN_STEP(0x114, 0xF) SCAN_STEP(0x118, 0xF)
    SCAN_STEP(0x142, 0xA) SCAN_STEP(0x143, 0xC)
#undef SCAN_STEP
    return x;
}
// in: per face bx0,by0,bx1,by1 (or empty). out: per pair (face, pixel)
__global__ void k(const int *box, int *out, int *npairs) {
    __shared__ int start[WAVE];
    const int lane = threadIdx.x;
    int cf = 0, packed = 0;
    const int bx0 = box[4 * lane], by0 = box[4 * lane + 1], bx1 = box[4 * lane + 2], by1 = box[4 * lane + 3];
    if (bx0 <= bx1 && by0 <= by1) { cf = (bx1 - bx0 + 1) * (by1 - by0 + 1); packed = (bx0 << 13) | (by0 << 16) | ((bx1 - bx0) << 19); }
    const int incl = wave_scan<false>(cf);
    const int off = incl - cf;
    const int n_pairs = __builtin_amdgcn_readlane(incl, 63);
    packed |= off;
    int carry = 0;
    for (int q0 = 0; q0 < n_pairs; q0 += WAVE) {
        start[lane] = 0;
        if (cf > 0 && off >= q0 && off < q0 + WAVE) start[off - q0] = lane + 1;
        __syncthreads();
        const int fi = max(wave_scan<true>(start[lane]), carry);
        carry = __builtin_amdgcn_readlane(fi, 63);
        const bool valid = q0 + lane < n_pairs;
        const int fs = max(fi - 1, 0);
        const int pk = __shfl(packed, fs, WAVE);
        const int rr = q0 + lane - (pk & 0x1FFF);
        const int bw = ((pk >> 19) & 7) + 1;
        const int dy = (int)((float)rr * __builtin_amdgcn_rcpf((float)bw) + 1e-3f);
        const int p = (((((pk >> 16) & 7) + dy) << 3) + ((pk >> 13) & 7) + (rr - dy * bw)) & 63;
        if (valid) { out[2 * (q0 + lane)] = fs; out[2 * (q0 + lane) + 1] = p; }
    }
    if (lane == 0) *npairs = n_pairs;
}
int main() {
    int box[256], *dbox, *dout, *dn; int out[2 * 4096], np;
    srand(3);
    for (int f = 0; f < 64; ++f) {
        int x0 = rand() % 8, y0 = rand() % 8, x1 = x0 + rand() % 4 - (f % 5 == 0 ? 5 : 0), y1 = y0 + rand() % 5;
        if (x1 > 7) x1 = 7; if (y1 > 7) y1 = 7;
        box[4*f] = x0; box[4*f+1] = y0; box[4*f+2] = x1; box[4*f+3] = y1;
    }
    (void)hipMalloc(&dbox, sizeof(box)); (void)hipMalloc(&dout, sizeof(out)); (void)hipMalloc(&dn, 4);
    (void)hipMemcpy(dbox, box, sizeof(box), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dbox, dout, dn);
    (void)hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost); (void)hipMemcpy(&np, dn, 4, hipMemcpyDeviceToHost);
    int q = 0, bad = 0;
    for (int f = 0; f < 64; ++f) {
        if (box[4*f] > box[4*f+2] || box[4*f+1] > box[4*f+3]) continue;
        for (int y = box[4*f+1]; y <= box[4*f+3]; ++y) for (int x = box[4*f]; x <= box[4*f+2]; ++x) {
            if (out[2*q] != f || out[2*q+1] != y*8+x) { if (bad < 10) printf("pair %d: got (%d,%d) want (%d,%d)\n", q, out[2*q], out[2*q+1], f, y*8+x); ++bad; }
            ++q;
        }
    }
    printf("pairs %d (kernel %d) bad %d\n", q, np, bad);
}
