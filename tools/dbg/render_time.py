"""Renderer.forward (the drop-in p3d_renderer.Renderer) with autograd at 512 frames: forward + backward time and memory across calls."""
import os, sys, time
import torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
from smilify_amd import model_io, synthetic
t = model_io.load_model(os.path.join(REPO, "data", "models", "SMILy_STICK.npz"))
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
f = synthetic.make_problem(t, B, 1, 256, "cuda:0", window=10)
smal, rend = f.smal_model, f.renderer
beta = f.betas.detach()[None].expand(B, -1).contiguous().requires_grad_()
theta = f._pose.detach().clone().requires_grad_()
trans = f.trans.detach().clone().requires_grad_()
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync()
    verts, joints, _, _ = smal(beta, theta, trans=trans)
    sil, proj = rend(verts, joints, smal.faces[None].expand(B, -1, -1))
    t1 = sync()
    (sil.mean() + 1e-4 * proj.sum()).backward()
    t2 = sync()
    for p in (beta, theta, trans): p.grad = None
    print(f"B={B} it {it}: SMAL + Renderer forward {1e3*(t1-t0):.2f} ms  backward {1e3*(t2-t1):.2f} ms  alloc {torch.cuda.memory_allocated()/2**20:.0f} MiB")
