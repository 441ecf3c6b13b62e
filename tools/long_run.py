"""Stability check: many fused fit iterations on the cfg2 workload; prints the loss trajectory and parameter sanity.
   python tools/long_run.py [steps] [frames] [tie rule: depth_face_id | reference_queue]"""
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from smilify_amd import engine, model_io, synthetic  # noqa: E402

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 300
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 512
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
f = synthetic.make_problem(t, frames, 1, 256, "cuda:0")
if len(sys.argv) > 3:
    f.renderer.raster_settings = engine.raster_settings(tie_rule=sys.argv[3])
f.begin_stage(synthetic.STAGE1_LR)
t0 = time.perf_counter()
for i in range(steps):
    objs = f.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)
    if i % 25 == 0 or i == steps - 1:
        o = objs.cpu()
        assert torch.isfinite(o).all(), (i, o)
        print(f"iter {i:4d}  total {o[:9].sum():12.3f}  joint {o[0]:10.3f}  sil {o[5]:9.3f}  limit {o[1]:8.3f}  temporal {o[6:9].sum():8.3f}  "
              f"fov {f.fov.item():7.3f}", flush=True)
torch.cuda.synchronize()
if len(sys.argv) > 3:
    print("replayed pixels in the last iteration:", engine.raster_stats(f.device_model, frames)["tie_pixels"])
print(f"{steps} iterations in {time.perf_counter() - t0:.1f} s; pose finite: {bool(torch.isfinite(f._pose).all())}, "
      f"betas {f.betas.detach().cpu().numpy().round(3)}")
