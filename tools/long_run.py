"""Stability check: many fused fit iterations on the cfg2 workload; prints the loss trajectory and parameter sanity.
   python tools/long_run.py [steps] [frames] [tie rule: depth_face_id | reference_queue]
   python tools/long_run.py [steps] [frames] loop      the reference's driver loop (forward per window, get_temporal, backward, torch.optim.Adam)
                                                       beside the fused fit_step on the same problem: same trajectory, stable memory"""
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
if len(sys.argv) > 3 and sys.argv[3] != "loop":
    f.renderer.raster_settings = engine.raster_settings(tie_rule=sys.argv[3])
if len(sys.argv) > 3 and sys.argv[3] == "loop":
    g = synthetic.make_problem(t, frames, 1, 256, "cuda:0")
    g.begin_stage(synthetic.STAGE1_LR)
    W = f.config.WINDOW_SIZE
    opt = torch.optim.Adam([{"params": [p for n, p in f.named_parameters() if n != "fov"], "lr": synthetic.STAGE1_LR}, {"params": [f.fov], "lr": 1}],
                           lr=synthetic.STAGE1_LR, betas=(0.5, 0.999))
    mem0 = None
    for i in range(steps):
        opt.zero_grad()
        acc = 0
        for j in range(0, frames, W):
            loss, _ = f(list(range(j, min(frames, j + W))), synthetic.STAGE1_WEIGHTS, 1)
            acc += loss.mean()
        jl, gl, tl = f.get_temporal(synthetic.STAGE1_TEMPORAL)
        acc = acc + jl + gl + tl
        acc.backward()
        opt.step()
        ref = g.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)
        if i == 5:
            torch.cuda.synchronize(); mem0 = torch.cuda.memory_allocated()
        if i % 25 == 0 or i == steps - 1:
            a, b = float(acc), float(ref[:9].sum())
            assert a == a and b == b
            print(f"epoch {i:4d}  driver loop {a:12.3f}  fit_step {b:12.3f}  rel diff {abs(a - b) / abs(b):.2e}  "
                  f"max |pose diff| {float((f._pose - g._pose).abs().max()):.2e}  fov {f.fov.item():7.3f} / {g.fov.item():7.3f}", flush=True)
    torch.cuda.synchronize()
    print(f"{steps} epochs of both; memory growth since epoch 5: {torch.cuda.memory_allocated() - mem0} bytes; windows served from one evaluation "
          f"in the last epoch: {f._epoch['served'] if f._epoch else 0} of {(frames + W - 1) // W}")
    sys.exit(0)
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
