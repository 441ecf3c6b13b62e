// Instrumentation hooks of the rasteriser kernels.  The product (libsmilfit.so) compiles them to nothing; `make variant` defines
// SMIL_INSTRUMENTED and takes the real definitions from tools/dbg/raster_hooks_dbg.h (phase timers of the tile kernel -DTILE_TIMERS,
// phase marks / ablations of the setup kernel -DDBG_SETUP_TIMERS, -DABL_SETUP_*).
#pragma once
#ifdef SMIL_INSTRUMENTED
#include "raster_hooks_dbg.h"
#else
#define TT_INIT
#define TT(k)
#define TSTAT(k, v)
#define TT_FLUSH
#define HOOK_ARGS_FIELDS
#define HOOK_HOST_LAUNCH_SETUP(a, stream)
#define TSETUP_INIT
#define TSETUP(k)
#define TSETUP_REPORT
#define HOOK_SETUP_COUNT(stmt) stmt
#endif
