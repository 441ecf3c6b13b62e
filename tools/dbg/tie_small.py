# a few fit iterations at a small batch under tie_rule = reference_queue (run under rocprofv3 --kernel-trace --stats):  tie_small.py <frames>
import os, sys, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
from smilify_amd import engine, model_io, synthetic
B = int(sys.argv[1])
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
f = synthetic.make_problem(t, B, 1, 256, "cuda:0", window=10)
f.renderer.raster_settings = engine.raster_settings(tie_rule="reference_queue")
f.begin_stage(5e-3)
for _ in range(20):
    f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
torch.cuda.synchronize()
print("replayed pixels:", engine.raster_stats(f.device_model, B)["tie_pixels"])
