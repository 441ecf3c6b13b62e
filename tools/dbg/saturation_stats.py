"""CPU analysis (oracle only, no GPU): how many (face, pixel) records would a pixel still need if it stopped taking
records once the transmittance of its certainly-kept nearest records fell below 1e-12 (the kernel's zero-gradient cut)?
Usage: python tools/dbg/saturation_stats.py [stick|mouse] [frames]"""
import sys, os
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
from oracle import lbs_ref, render_ref
from smilify_amd import model_io
from smilify_amd.synthetic import random_pose
from conftest import oracle_model, MODEL_FILES

key = sys.argv[1] if len(sys.argv) > 1 else "stick"
frames = int(sys.argv[2]) if len(sys.argv) > 2 else 2
S, K = 256, 100
t = model_io.load_model(MODEL_FILES[key])
m = oracle_model(t)
gen = torch.Generator().manual_seed(1234)
pose, trans = random_pose(frames, t.J, gen)
betas = 0.5 * torch.randn(t.nB, generator=gen)
out = lbs_ref.smal_forward(m, betas[None].expand(frames, -1), pose, trans=trans)
R, T = render_ref.look_at_view_transform(2.7 if key == "stick" else 4.0, 15.0, 0.0)
ndc = render_ref.project_to_ndc(out["verts"], R.expand(frames, 3, 3), T.expand(frames, 3), torch.full((frames,), 60.0)).numpy()
# K large enough to see every candidate
KK = 1024
sil, nc, ff, fd, fz = render_ref.silhouette_forward_np(ndc, t.faces, S, K=KK, want_fragments=True)
print("max candidates per pixel", nc.max(), " pixels with > K:", (nc > K).mean())
valid = ff >= 0
p = 1.0 / (1.0 + np.exp(np.clip(fd / 1e-4, -80, 80)))       # sigmoid(-d/sigma)
lf = np.where(valid, np.log2(np.maximum(1.0 - p, 1e-300)), 0.0)
cum = np.cumsum(lf, axis=-1)                                 # nearest first (fragments are sorted by z)
n = np.minimum(nc, KK)
kept = np.minimum(n, K)
# first index where the cumulative log2 transmittance < -40 (alpha < 1e-12)
sat = (cum < -40.0) & valid
first = np.where(sat.any(-1), sat.argmax(-1) + 1, 10 ** 9)
need = np.minimum(kept, first)
alpha_final = np.take_along_axis(cum, np.maximum(kept - 1, 0)[..., None], -1)[..., 0]
dead = (alpha_final < -40.0) & (kept > 0)
tot = n.sum()
print(f"candidates {tot}  kept(K) {kept.sum()} ({kept.sum()/tot:.3f})  needed until saturation {need.sum()} ({need.sum()/tot:.3f})")
print(f"pixels touched {np.mean(n>0):.3f}; of touched: saturated (alpha<1e-12) {dead.sum()/ (n>0).sum():.3f}; records in saturated pixels {kept[dead].sum()/kept.sum():.3f} of kept, candidates {n[dead].sum()/tot:.3f}")
trunc = nc > K
print(f"truncating pixels: {trunc.sum()/(n>0).sum():.3f} of touched; saturated among them {(dead & trunc).sum()/max(trunc.sum(),1):.3f}")
