"""``SMAL.__call__`` forward + backward with one beta row per frame (the reference's neural caller, smil_image_regressor.py:2663), for
rocprofv3 --kernel-trace --stats: which kernels the drop-in module spends its time in outside the fit iteration.

    rocprofv3 --kernel-trace --stats -d gpurun_out/smal -o smal -- python3 tools/smal_call_probe.py --frames 4096
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from smilify_amd import model_io  # noqa: E402
from smilify_amd.smal_torch import SMAL  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SMILy_STICK")
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
t = model_io.load_model(os.path.join(REPO, "data", "models", args.model + ".npz"))
smal = SMAL(dev, tables=t)
B, J, V, nB = args.frames, t.J, t.V, t.nB
g = torch.Generator().manual_seed(1)
beta = (0.5 * torch.randn(B, nB, generator=g)).to(dev).requires_grad_()
theta = (0.15 * torch.randn(B, J, 3, generator=g)).to(dev).requires_grad_()
trans = (0.05 * torch.randn(B, 3, generator=g)).to(dev).requires_grad_()
ls = (0.05 * torch.randn(B, J, 3, generator=g)).to(dev).requires_grad_()
wv = torch.randn(B, V, 3, generator=g).to(dev)
wj = torch.randn(B, J, 3, generator=g).to(dev)
for it in range(args.reps + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    verts, joints, Rs, v_shaped = smal(beta, theta, trans=trans, betas_logscale=ls)
    ((verts * wv).sum() + (joints * wj).sum()).backward()
    for p in (beta, theta, trans, ls):
        p.grad = None
torch.cuda.synchronize()
print(f"{args.model}: B={B} V={V} J={J} nB={nB}  SMAL.__call__ forward + backward {1e3 * (time.perf_counter() - t0) / args.reps:.3f} ms per call")
