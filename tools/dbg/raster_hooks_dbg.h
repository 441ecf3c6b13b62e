// Definitions of the tile kernel's instrumentation hooks for the instrumented libraries of tools/dbg (never part of libsmilfit.so):
//   make -C smilify_amd/csrc variant NAME=<n> VFLAGS="-DDBG_TIMERS [-DDBG_STATS] [-DRASTER_EXPERIMENT] [-DABL_EXTRA_VALU=64]"
// -DDBG_TIMERS: per-phase cycle sums of the tile kernel (printed by the next launch).  The marks only read the cycle counter into
//   registers; the one place that touches memory is TIMERS_FLUSH / STAT: lane 0 of a single-wave workgroup issuing returning-free
//   global atomics on a 512-byte buffer, outside any divergent region and with no barrier or spin depending on their completion -
//   nothing another wave waits for, so they cannot deadlock the persistent loop (the kernel's exit condition is the work counter alone).
// -DDBG_STATS: work counters (thousands of waves adding to the same few words: the launch runs several times longer).
// -DRASTER_EXPERIMENT: ablation knobs read from the environment by the host side (SMIL_STOP: cut the kernel off after a phase - 0 list,
//   1 + staging, 2 + pair sweep, 3 + blend / select; SMIL_WRAP: wrap every stream index into a cache-resident window; SMIL_SPLIT;
//   SMIL_RESIDENT: workgroups per CU); results are garbage under SMIL_WRAP / SMIL_STOP by design - timing experiments only.
#pragma once
#include <cstdio>
#include <cstdlib>

#define HOOK_ARGS_FIELDS unsigned long long *dbg; int stop_after; int force_split;

#ifdef DBG_TIMERS
#define TIMERS_INIT unsigned long long tph[8] = {0, 0, 0, 0, 0, 0, 0, 0}; unsigned long long tlast = __builtin_readcyclecounter(); const unsigned long long tstart_ = tlast; unsigned long long tstage_ = 0, tsweep_ = 0; unsigned long long tsub[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tsub_last = tlast; unsigned long long tp3[8] = {0, 0, 0, 0, 0, 0, 0, 0}, tp3_last = tlast;
#define TSUB(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tsub[k] += now_ - tsub_last; tsub_last = now_; }  // finer marks, independent of TMARK
#define TMARK(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tph[k] += now_ - tlast; tlast = now_; }
#define TP3_START { tp3_last = __builtin_readcyclecounter(); }
#define TP3(k) { const unsigned long long now_ = __builtin_readcyclecounter(); tp3[k] += now_ - tp3_last; tp3_last = now_; }  // inside pass 3 (dbg[48 + k])
#define TUNIT_START const unsigned long long tunit_ = tlast;
#define TUNIT_END if (a.dbg && lane == 0) { \
            const int cls = item < nc0 ? 0 : (item < nc0 + nc1 ? 1 : (item < nc0 + nc1 + nc2 ? 2 : 3)); \
            const unsigned long long dt_ = tlast - tunit_; \
            atomicAdd(&a.dbg[8 + cls], dt_); atomicMax(&a.dbg[12 + cls], dt_); atomicMax(&a.dbg[16 + cls], tunit_ - tstart_); }
#define TSTAGE_MARK { const unsigned long long now_ = __builtin_readcyclecounter(); tstage_ += now_ - tlast; tlast = now_; }
#define TSWEEP_MARK { const unsigned long long now_ = __builtin_readcyclecounter(); tsweep_ += now_ - tlast; tlast = now_; }
#define TIMERS_FLUSH if (a.dbg && lane == 0) { for (int k_ = 0; k_ < 5; ++k_) atomicAdd(&a.dbg[k_], tph[k_]); \
        atomicMin(&a.dbg[5], tstart_); atomicMin(&a.dbg[6], tlast); atomicMax(&a.dbg[7], tlast); \
        atomicAdd(&a.dbg[3], tstage_); atomicAdd(&a.dbg[1], tsweep_); for (int k_ = 0; k_ < 8; ++k_) { atomicAdd(&a.dbg[32 + k_], tsub[k_]); atomicAdd(&a.dbg[48 + k_], tp3[k_]); } }
#else
#define TIMERS_INIT
#define TSUB(k)
#define TMARK(k)
#define TP3_START
#define TP3(k)
#define TUNIT_START
#define TUNIT_END
#define TSTAGE_MARK
#define TSWEEP_MARK
#define TIMERS_FLUSH
#endif

#if defined(DBG_TIMERS) && defined(DBG_STATS)
#define STAT(k, v) { const unsigned long long v_ = (unsigned long long)(v); /* (all lanes: v may hold a ballot) */ if (a.dbg && lane == 0) atomicAdd(&a.dbg[k], v_); }
#else
#define STAT(k, v)
#endif

#ifdef RASTER_EXPERIMENT
__constant__ uint32_t g_wrap_mask = 0xFFFFFFFFu;
#define HOOK_WRAP_IDX(i) ((i) & g_wrap_mask)
#define HOOK_SPLIT_LOG(x) (a.force_split >= 0 ? (unsigned int)a.force_split : (x))
#define HOOK_STOP_AFTER(k, stmt) if (a.stop_after == (k)) stmt;
#define HOOK_RESIDENT(resident) if (const char *e = getenv("SMIL_RESIDENT")) resident = (long long)device_cus() * (atoi(e) > 0 && atoi(e) <= RESIDENT_PER_CU ? atoi(e) : RESIDENT_PER_CU);
#define HOOK_HOST_EXPERIMENT(a, stream) \
    if (const char *e = getenv("SMIL_STOP")) a.stop_after = atoi(e); \
    if (const char *e = getenv("SMIL_SPLIT")) a.force_split = atoi(e); \
    { uint32_t mask = 0xFFFFFFFFu; \
      if (const char *e = getenv("SMIL_WRAP")) mask = (uint32_t)strtoul(e, nullptr, 0); \
      (void)hipMemcpyToSymbolAsync(HIP_SYMBOL(g_wrap_mask), &mask, sizeof(mask), 0, hipMemcpyHostToDevice, stream); }
#else
#define HOOK_WRAP_IDX(i) (i)
#define HOOK_SPLIT_LOG(x) (x)
#define HOOK_STOP_AFTER(k, stmt)
#define HOOK_RESIDENT(resident)
#define HOOK_HOST_EXPERIMENT(a, stream)
#endif

#ifdef ABL_EXTRA_VALU  // ABL_EXTRA_VALU dependency-free v_fma_f32 per sweep step (how VALU-bound is the launch?)
#define HOOK_EXTRA_VALU(pc) { float d0_ = pc.x, d1_ = pc.y, d2_ = pc.z, d3_ = pc.w; \
    _Pragma("unroll") for (int i_ = 0; i_ < ABL_EXTRA_VALU / 4; ++i_) \
        asm volatile("v_fma_f32 %0, %0, %0, %0\n v_fma_f32 %1, %1, %1, %1\n v_fma_f32 %2, %2, %2, %2\n v_fma_f32 %3, %3, %3, %3" : "+v"(d0_), "+v"(d1_), "+v"(d2_), "+v"(d3_)); }
#else
#define HOOK_EXTRA_VALU(pc)
#endif

#ifdef DBG_TIMERS
static inline void raster_dbg_report(unsigned long long *&dbg_out) {
    static unsigned long long *dbg_dev = nullptr;
    {
        if (!dbg_dev) { (void)hipMalloc(&dbg_dev, 512); (void)hipMemset(dbg_dev, 0, 512); }
        unsigned long long h[64];
        (void)hipMemcpy(h, dbg_dev, 512, hipMemcpyDeviceToHost);  // totals of the launches so far
        fprintf(stderr, "[dbg sub] other %.3e  item fetch %.3e  list bounds %.3e  sort %.3e  blend sweep %.3e  select %.3e  epilogue %.3e  pass3 %.3e\n",
                (double)h[32], (double)h[33], (double)h[34], (double)h[35], (double)h[36], (double)h[37], (double)h[38], (double)h[39]);
        fprintf(stderr, "[dbg pass3] chunk starts %.3e  vertex chain %.3e  zero + sync %.3e  record loop %.3e  sync %.3e  flush %.3e\n",
                (double)h[48], (double)h[49], (double)h[50], (double)h[51], (double)h[52], (double)h[53]);
        fprintf(stderr, "[dbg timers] list %.3e  pass1 sweep %.3e  blend+select %.3e  pass1 staging %.3e  pass3 %.3e cycles (summed over waves); "
                "first wave exit %.3e, last wave exit %.3e cycles after the first start\n",
                (double)h[0], (double)h[1], (double)h[2], (double)h[3], (double)h[4], (double)(h[6] - h[5]), (double)(h[7] - h[5]));
        fprintf(stderr, "[dbg timers] per class: unit time sums %.3e %.3e %.3e %.3e  longest unit %.3e %.3e %.3e %.3e  latest unit start %.3e %.3e %.3e %.3e\n",
                (double)h[8], (double)h[9], (double)h[10], (double)h[11], (double)h[12], (double)h[13], (double)h[14], (double)h[15],
                (double)h[16], (double)h[17], (double)h[18], (double)h[19]);
        fprintf(stderr, "[dbg stats] units %.4e  list entries %.4e  pairs evaluated %.4e  accepted %.4e  compact %.4e  pixels: touched %.4e truncated %.4e with gradient %.4e; chunks walked %.4e of %.4e, pixels still open at the end %.4e\n",
                (double)h[26], (double)h[27], (double)h[20], (double)h[21], (double)h[22], (double)h[24], (double)h[23], (double)h[25], (double)h[28], (double)h[29], (double)h[30]);
        fprintf(stderr, "[dbg stats] staged faces with a non-empty pixel box %.4e\n", (double)h[31]);
        fprintf(stderr, "[dbg stats] (sub-)tiles with a truncated pixel %.4e (their records %.4e), that may truncate %.4e (records %.4e); pixels with a split tie group %.4e\n",
                (double)h[40], (double)h[42], (double)h[41], (double)h[43], (double)h[44]);
        fprintf(stderr, "[dbg stats] blend sweep: records kept for certain %.4e, above their pixel's chosen first digit (dropped) %.4e\n", (double)h[45], (double)h[46]);
        (void)hipMemset(dbg_dev, 0, 512);
        { const unsigned long long big[3] = {~0ull, ~0ull, 0ull}; (void)hipMemcpy(dbg_dev + 5, big, 24, hipMemcpyHostToDevice); }
        dbg_out = dbg_dev;
    }
}
#define HOOK_HOST_LAUNCH_SETUP(a, stream) a.dbg = nullptr; a.stop_after = 99; a.force_split = -1; HOOK_HOST_EXPERIMENT(a, stream) raster_dbg_report(a.dbg);
#else
#define HOOK_HOST_LAUNCH_SETUP(a, stream) a.dbg = nullptr; a.stop_after = 99; a.force_split = -1; HOOK_HOST_EXPERIMENT(a, stream)
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

// ---- k_raster_tie_replay (tie_rule = reference_queue) ----
#ifdef DBG_TIE_TIMERS  // where a replay wave's cycles go (tools/dbg/tie_timers.py): list order, face evaluation, queue, gradient, ticket, ...
__device__ unsigned long long g_tie_t[8];
extern "C" int smil_dbg_tie_timers(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_tie_t), sizeof(g_tie_t)) != hipSuccess) return -1;
    if (reset) { unsigned long long z[8] = {}; if (hipMemcpyToSymbol(HIP_SYMBOL(g_tie_t), z, sizeof(z)) != hipSuccess) return -1; }
    return 0;
}
#define TIE_TIMERS_INIT unsigned long long tt_[8] = {}, tt0_ = __builtin_amdgcn_s_memtime();
#define TIE_T(i) { const unsigned long long n_ = __builtin_amdgcn_s_memtime(); tt_[i] += n_ - tt0_; tt0_ = n_; }
#define TIE_TIMERS_FLUSH if (lane == 0) for (int i_ = 0; i_ < 8; ++i_) atomicAdd(&g_tie_t[i_], tt_[i_]);
#else
#define TIE_TIMERS_INIT
#define TIE_T(k)
#define TIE_TIMERS_FLUSH
#endif
#ifdef ABL_TIE_NO_ATOMICS  // what the replay's gradient atomics cost (garbage gradients)
#define HOOK_TIE_GRADIENT(stmt)
#else
#define HOOK_TIE_GRADIENT(stmt) stmt
#endif
#ifdef TIE_NO_EQUIV  // every cut tie group replayed, interchangeable ones included
#define HOOK_TIE_EQUIV false
#else
#define HOOK_TIE_EQUIV true
#endif
