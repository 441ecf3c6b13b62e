"""Where SMAL.__call__ forward + backward spends host time (per-frame betas, 4 096 frames)."""
import os, sys, time
import torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO)
from smilify_amd import model_io
from smilify_amd.smal_torch import SMAL
name = sys.argv[1] if len(sys.argv) > 1 else "SMILy_Mouse_static_joints"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
dev = torch.device("cuda:0")
t = model_io.load_model(os.path.join(REPO, "data", "models", name + ".npz"))
smal = SMAL(dev, tables=t)
J, V, nB = t.J, t.V, t.nB
g = torch.Generator().manual_seed(1)
beta = (0.5 * torch.randn(B, nB, generator=g)).to(dev).requires_grad_()
theta = (0.15 * torch.randn(B, J, 3, generator=g)).to(dev).requires_grad_()
trans = (0.05 * torch.randn(B, 3, generator=g)).to(dev).requires_grad_()
ls = (0.05 * torch.randn(B, J, 3, generator=g)).to(dev).requires_grad_()
wv = torch.randn(B, V, 3, generator=g).to(dev); wj = torch.randn(B, J, 3, generator=g).to(dev)
def sync(): torch.cuda.synchronize(); return time.perf_counter()
for it in range(6):
    t0 = sync()
    verts, joints, Rs, v_shaped = smal(beta, theta, trans=trans, betas_logscale=ls)
    t1 = sync()
    loss = (verts * wv).sum() + (joints * wj).sum()
    t2 = sync()
    loss.backward()
    t3 = sync()
    for p in (beta, theta, trans, ls): p.grad = None
    print(f"{name} B={B} it {it}: forward {1e3*(t1-t0):.2f} ms  loss {1e3*(t2-t1):.2f} ms  backward {1e3*(t3-t2):.2f} ms   alloc {torch.cuda.memory_allocated()/2**30:.2f} GiB reserved {torch.cuda.memory_reserved()/2**30:.2f} GiB")
