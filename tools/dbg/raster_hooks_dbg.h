// Definitions of the rasteriser's instrumentation hooks for the instrumented libraries of tools/dbg (never part of libsmilfit.so):
//   make -C smilify_amd/csrc variant NAME=<n> VFLAGS="-DTILE_TIMERS"
// -DTILE_TIMERS: per-phase cycle sums of the tile kernel, per wave (printed by the NEXT launch).  The marks only read the cycle counter
//   into registers; the one place that touches memory is TT_FLUSH at the end of the kernel: lane 0 of every wave issues return-free
//   global atomics on a small buffer, outside any divergent region and with no barrier or spin depending on their completion.
#pragma once
#include <cstdio>
#include <cstdlib>

#ifdef TILE_TIMERS
#define N_TT 16
#define HOOK_ARGS_FIELDS unsigned long long *dbg;
#define TT_INIT unsigned long long tph[N_TT]; for (int k_ = 0; k_ < N_TT; ++k_) tph[k_] = 0ull; unsigned long long tlast = __builtin_readcyclecounter(); \
    if (a.dbg && tid == 0) { const unsigned long long c_ = atomicAdd(&a.dbg[N_TT], 1ull) + 1ull; atomicMax(&a.dbg[N_TT + 1], c_); }  /* census of resident workgroups */
#define TT(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[k] += now_ - tlast; tlast = now_; }
#define TSTAT(k, v) { const unsigned long long v_ = (unsigned long long)(v); if (a.dbg && lane == 0) atomicAdd(&a.dbg[N_TT + 2 + (k)], v_); }
#define TT_FLUSH if (a.dbg && lane == 0) { for (int k_ = 0; k_ < N_TT; ++k_) if (tph[k_]) atomicAdd(&a.dbg[k_], tph[k_]); } \
    if (a.dbg && tid == 0) atomicAdd(&a.dbg[N_TT], ~0ull);
static inline void raster_dbg_report(unsigned long long *&dbg_out) {
    static unsigned long long *dbg_dev = nullptr;
    if (!dbg_dev) { (void)hipMalloc(&dbg_dev, (N_TT + 10) * 8); (void)hipMemset(dbg_dev, 0, (N_TT + 10) * 8); }
    unsigned long long h[N_TT + 10];
    (void)hipMemcpy(h, dbg_dev, (N_TT + 10) * 8, hipMemcpyDeviceToHost);  // totals of the launches so far
    static const char *names[N_TT] = {"ticket+item", "list bounds", "sort", "subtile init", "pass1 walk", "pass1 wait", "pick1+zero", "blend sweep",
                                      "select", "epilogue", "p3 window setup", "p3 sweep", "p3 flush", "other", "-", "-"};
    double tot = 0; for (int k = 0; k < N_TT; ++k) tot += (double)h[k];
    if (tot > 0) {
        fprintf(stderr, "[tile timers] wave-cycles by phase (sum %.3e):", tot);
        for (int k = 0; k < 14; ++k) fprintf(stderr, "  %s %.1f%%", names[k], 100.0 * (double)h[k] / tot);
        fprintf(stderr, "  | most workgroups resident at once: %llu | tiles %.4e list entries %.4e pair slots %.4e records %.4e selected %.4e spilled %.4e\n", h[N_TT + 1], (double)h[N_TT + 2], (double)h[N_TT + 3], (double)h[N_TT + 4], (double)h[N_TT + 5], (double)h[N_TT + 6], (double)h[N_TT + 7]);
    }
    (void)hipMemset(dbg_dev, 0, (N_TT + 10) * 8);
    dbg_out = dbg_dev;
}
#define HOOK_HOST_LAUNCH_SETUP(a, stream) raster_dbg_report(a.dbg);
#else
#define TT_INIT
#define TT(k)
#define TSTAT(k, v)
#define TT_FLUSH
#define HOOK_ARGS_FIELDS
#define HOOK_HOST_LAUNCH_SETUP(a, stream)
#endif

// -DDBG_SETUP_TIMERS: marks inside k_raster_setup (block 0, thread 0; device printf at the end of the kernel): where one image's
// setup time goes.  Timing experiments only.
#ifdef DBG_SETUP_TIMERS
#define TSETUP_INIT long long tsu_[12]; for (int k_ = 0; k_ < 12; ++k_) tsu_[k_] = 0; tsu_[0] = clock64();
#define TSETUP(k) if (blockIdx.x == 0 && threadIdx.x == 0) tsu_[k] = clock64();
#define TSETUP_REPORT if (blockIdx.x == 0 && threadIdx.x == 0) printf("[dbg setup] cycles after start: zero+init %lld  pass1 %lld  cuts %lld  bounds %lld  classes+prefix %lld  global atomics %lld  items %lld  pass2 %lld\n", \
    tsu_[1] - tsu_[0], tsu_[2] - tsu_[0], tsu_[3] - tsu_[0], tsu_[4] - tsu_[0], tsu_[5] - tsu_[0], tsu_[6] - tsu_[0], tsu_[7] - tsu_[0], tsu_[8] - tsu_[0]);
#else
#define TSETUP_INIT
#define TSETUP(k)
#define TSETUP_REPORT
#endif

// -DABL_SETUP_NOCOUNT: k_raster_setup without the per-(face, tile) LDS atomics of its counting pass (garbage work lists: timing only)
#ifdef ABL_SETUP_NOCOUNT
#define HOOK_SETUP_COUNT(stmt)
#elif defined(ABL_SETUP_COUNT32)  // a 32-bit add of the entry alone (no cost: garbage classes, timing only)
#define HOOK_SETUP_COUNT(stmt) atomicAdd(reinterpret_cast<uint32_t *>(&tcnt64[t]) + 1, 1u);
#else
#define HOOK_SETUP_COUNT(stmt) stmt
#endif
