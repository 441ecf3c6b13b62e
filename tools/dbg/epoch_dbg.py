"""Cached-epoch forward() against the window-by-window one, term by term (debug aid for SMALFitter._epoch_window)."""
import os, sys
import numpy as np
import torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from smilify_amd import synthetic, model_io
t = model_io.synthetic_model()
frames, window = 7, 2
weights, w_temp = [10.0, 500.0, 1.0, 1.0, 100.0, 0.1], 100.0
a = synthetic.make_problem(t, frames, 1, 40, "cuda:0", radius=2.2, seed=5, window=window)
b = synthetic.make_problem(t, frames, 1, 40, "cuda:0", radius=2.2, seed=5, window=window)
b.epoch_cache = False
oa = torch.optim.Adam(a.parameters(), lr=5e-3, betas=(0.5, 0.999)); ob = torch.optim.Adam(b.parameters(), lr=5e-3, betas=(0.5, 0.999))
for epoch in range(3):
    res = []
    for m in (a, b):
        for p in m.parameters(): p.grad = None
        acc = 0; rows = []
        for j in range(0, frames, window):
            loss, objs = m(list(range(j, min(frames, j + window))), weights, 1)
            rows.append([float(loss)] + [float(objs[k]) for k in ("joint", "limit", "pose", "splay", "betas", "sil_reproj")])
            acc = acc + loss.mean()
        jl, gl, tl = m.get_temporal(w_temp)
        (acc + jl + gl + tl).backward()
        res.append((np.array(rows), {n: p.grad.clone() for n, p in m.named_parameters() if p.grad is not None}))
    print("epoch", epoch, "\ncached:\n", res[0][0], "\ndirect:\n", res[1][0])
    for n in res[1][1]:
        d = (res[0][1][n] - res[1][1][n]).abs()
        print("  grad", n, "max|diff|", float(d.max()), "max|g|", float(res[1][1][n].abs().max()), "argmax", int(d.reshape(-1).argmax()))
    for n, p in a.named_parameters():
        print("  param", n, float((p.detach() - dict(b.named_parameters())[n].detach()).abs().max()))
    oa.step(); ob.step()
