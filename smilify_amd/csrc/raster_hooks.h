// Instrumentation hooks of the tile kernel.  The product (libsmilfit.so) compiles them to nothing; `make variant` defines
// SMIL_INSTRUMENTED and takes the real definitions from tools/dbg/raster_hooks_dbg.h (phase timers -DDBG_TIMERS, work counters
// -DDBG_STATS, cut-off / wrap / resident experiments -DRASTER_EXPERIMENT, dummy VALU -DABL_EXTRA_VALU=n; of the replay kernel: phase timers -DDBG_TIE_TIMERS, its gradient atomics
// left out -DABL_TIE_NO_ATOMICS, interchangeable tie groups replayed all the same -DTIE_NO_EQUIV).
#pragma once
#ifdef SMIL_INSTRUMENTED
#include "raster_hooks_dbg.h"
#else
#define TIMERS_INIT
#define TSUB(k)
#define TMARK(k)
#define TP3_START
#define TP3(k)
#define STAT(k, v)
#define TIMERS_FLUSH
#define TUNIT_START
#define TUNIT_END
#define TSTAGE_MARK
#define TSWEEP_MARK
#define HOOK_WRAP_IDX(i) (i)
#define HOOK_SPLIT_LOG(x) (x)
#define HOOK_STOP_AFTER(k, stmt)
#define HOOK_RESIDENT(resident)
#define HOOK_EXTRA_VALU(pc)
#define HOOK_ARGS_FIELDS
#define HOOK_HOST_LAUNCH_SETUP(a, stream)
#define TSETUP_INIT
#define TSETUP(k)
#define TSETUP_REPORT
#define HOOK_SETUP_COUNT(stmt) stmt
#define TIE_TIMERS_INIT
#define TIE_T(k)
#define TIE_TIMERS_FLUSH
#define HOOK_TIE_GRADIENT(stmt) stmt
#define HOOK_TIE_EQUIV true
#endif
