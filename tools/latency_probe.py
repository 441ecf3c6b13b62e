import sys, time, torch
import os
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from smilify_amd import engine, model_io, synthetic
TIE = sys.argv[1] if len(sys.argv) > 1 else None  # optional: reference_queue
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
for B in (1, 8, 64, 512):
    out = []
    for mode in ("eager", "graph"):
        f = synthetic.make_problem(t, B, 1, 256, "cuda:0", window=10)
        if TIE:
            f.renderer.raster_settings = engine.raster_settings(tie_rule=TIE)
        f.begin_stage(5e-3)
        step = f.fit_step_graph if mode == "graph" else f.fit_step
        for _ in range(3): step(synthetic.STAGE1_WEIGHTS, 100.0)
        blocks = []
        for _ in range(7):  # median of seven blocks of 30 iterations (a shared box shows a slow block now and then)
            torch.cuda.synchronize(); t0 = time.perf_counter()
            for _ in range(30): step(synthetic.STAGE1_WEIGHTS, 100.0)
            torch.cuda.synchronize(); blocks.append((time.perf_counter() - t0) / 30 * 1e3)
        out.append(sorted(blocks)[3])
    print(f"B={B}: fit_step {out[0]:.3f} ms   fit_step_graph {out[1]:.3f} ms" + (f"   [tie_rule {TIE}]" if TIE else ""))
