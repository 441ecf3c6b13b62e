"""Debug helper: dense rasteriser vs oracle on a small scene; prints where they differ."""
import os, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from smilify_amd import engine, model_io
from oracle import render_ref, lbs_ref, fitter_ref
from conftest import oracle_model  # noqa

key, S, dist, K = sys.argv[1], int(sys.argv[2]), float(sys.argv[3]), int(sys.argv[4])
if key == "synthetic":
    t = model_io.synthetic_model()
else:
    t = model_io.load_model(os.path.join(REPO, "data", "models", {"stick": "SMILy_STICK", "mouse": "SMILy_Mouse_static_joints"}[key] + ".npz"))
dev = torch.device("cuda:0")
dm = engine.DeviceModel(t, dev)
N = 1
m = oracle_model(t)
g = torch.Generator().manual_seed(5)
theta = 0.15 * torch.randn(N, t.J, 3, generator=g)
theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation()) + 0.1 * torch.randn(N, 3, generator=g)
out = lbs_ref.smal_forward(m, 0.3 * torch.randn(1, t.nB, generator=g).expand(N, -1), theta)
R, T = render_ref.look_at_view_transform(dist, 15.0, torch.zeros(N))
ndc = render_ref.project_to_ndc(out["verts"], R, T, torch.full((N,), 60.0)).contiguous()
ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
got = engine.silhouette_forward(dm, ndc.to(dev), S, engine.raster_settings(K=K)).cpu().numpy()
d = np.abs(got - ref)[0]
print("F", t.F, "max ncand", ncand.max(), "mean diff", d.mean(), "frac>1e-4", (d > 1e-4).mean(), "touched", (ncand[0] > 0).sum())
bad = np.argwhere(d > 1e-4)
print("bad pixels", len(bad))
for y, x in bad[:25]:
    print(f"  y={y} x={x} tile=({x//8},{y//8}) lane={(y%8)*8+x%8} ncand={ncand[0,y,x]} ref={ref[0,y,x]:.6f} got={got[0,y,x]:.6f}")
# per-tile summary
ty, tx = bad[:, 0] // 8, bad[:, 1] // 8
import collections
print(collections.Counter(zip(tx.tolist(), ty.tolist())).most_common(10))
if os.environ.get("DBG_CNT"):
    c = np.floor(got[0] + 1e-4).astype(int); seen = np.round((got[0] - c) * 1000).astype(int)
    print("cnt mismatches", (c != ncand[0]).sum(), "of touched", (ncand[0] > 0).sum(), "seen!=cnt", (seen != c).sum())
    bad = np.argwhere(c != ncand[0])
    for y, x in bad[:20]:
        print(f"  y={y} x={x} lane={(y%8)*8+x%8} ncand={ncand[0,y,x]} cnt={c[y,x]} seen={seen[y,x]}")
with render_ref.select_mode(1):
    ref1, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
got = engine.silhouette_forward(dm, ndc.to(dev), S, engine.raster_settings(K=K)).cpu().numpy()
d1 = np.abs(got - ref1)[0]
print("vs (z, face id) rule: max diff", d1.max(), "mean", d1.mean(), " vs faithful: max", d.max(), "sum rel", abs(got.sum()-ref.sum())/ref.sum(), "sum rel mode1", abs(got.sum()-ref1.sum())/ref1.sum())
bad = np.argwhere(d1 > 1e-5)
for y, x in bad[:10]:
    print(f"  y={y} x={x} ncand={ncand[0,y,x]} ref1={ref1[0,y,x]:.7f} got={got[0,y,x]:.7f}")
