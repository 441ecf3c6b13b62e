# which framework ops the one-frame fit iteration issues besides the library's kernels (fills, copies): torch.profiler over 10 iterations
import sys, os, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from smilify_amd import model_io, synthetic
from torch.profiler import profile, ProfilerActivity
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
f = synthetic.make_problem(t, int(os.environ.get("B1_FRAMES", "1")), 1, 256, "cuda:0", window=10)
f.begin_stage(5e-3)
for _ in range(5): f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    for _ in range(10): f.fit_step(synthetic.STAGE1_WEIGHTS, 100.0)
    torch.cuda.synchronize()
print(prof.key_averages().table(sort_by="cuda_time_total", row_limit=40, max_name_column_width=60))
print(prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=30, max_name_column_width=50, max_src_column_width=90))
