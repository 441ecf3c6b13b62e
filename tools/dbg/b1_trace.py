import sys, os, time, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smilify_amd import model_io, synthetic
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
f = synthetic.make_problem(t, int(os.environ.get("B1_FRAMES", "1")), 1, 256, "cuda:0", window=10)
f.begin_stage(5e-3)
for _ in range(30): f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
torch.cuda.synchronize()
