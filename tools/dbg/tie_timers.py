"""Where the replay kernel's wave cycles go (`make variant NAME=tt VFLAGS=-DDBG_TIE_TIMERS`):
   SMILFIT_LIB=smilify_amd/lib/libsmilfit_tt.so python tools/dbg/tie_timers.py <raster_probe args...>"""
import ctypes, os, runpy, sys
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
sys.argv = [os.path.join(REPO, "tools", "raster_probe.py")] + sys.argv[1:] + ["--quick", "--tie-rule", "reference_queue"]
try:
    runpy.run_path(sys.argv[0], run_name="__main__")
except SystemExit:
    pass
from smilify_amd import _lib
import torch
torch.cuda.synchronize()
out = (ctypes.c_ulonglong * 8)()
assert _lib.load().smil_dbg_tie_timers(out, 1) == 0
names = ["ordered list", "face evaluation", "queue", "gradient", "ticket + masks", "loop head", "blend + loss", "group scan (list not ordered)"]
tot = sum(out)
for n, v in zip(names, out):
    if v:
        print(f"{n:18s} {v:.3e} clock ticks  {100.0 * v / tot:5.1f} %")
