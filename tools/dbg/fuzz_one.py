# one scene of tools/dbg/fuzz_raster.py taken apart pixel by pixel: which pixels carry the gradient disagreement with the oracle
#   python tools/dbg/fuzz_one.py <seed> [clip|queue]
import os, sys
import numpy as np, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from smilify_amd import engine, model_io
from oracle import render_ref, lbs_ref, fitter_ref
from conftest import oracle_model
DEV = "cuda:0"
seed = int(sys.argv[1]); CLIP = len(sys.argv) > 2 and sys.argv[2] == "clip"
QUEUE = len(sys.argv) > 2 and sys.argv[2] == "queue"   # tie_rule="reference_queue" against the oracle's faithful queue
MODE, RULE = (0, "reference_queue") if QUEUE else (1, "depth_face_id")
rng = np.random.default_rng(seed)
key = ["synthetic", "stick", "mouse"][int(rng.integers(0, 3))]
t = {"synthetic": model_io.synthetic_model,
     "stick": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_STICK.npz")),
     "mouse": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_Mouse_static_joints.npz"))}[key]()
dm = engine.DeviceModel(t, DEV)
S = int(rng.integers(9, 140)); K = int(rng.choice([1, 2, 5, 17, 64, 100, 128]))
dist = float(np.exp(rng.uniform(np.log(1.2), np.log(40.0)))) * (1.5 if key == "mouse" else 1.0)
if CLIP:
    dist = float(rng.uniform(0.05, 0.9)) * (1.5 if key == "mouse" else 1.0)
N = int(rng.integers(1, 4))
m = oracle_model(t)
g = torch.Generator().manual_seed(seed)
theta = 0.3 * torch.randn(N, t.J, 3, generator=g)
theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation()) + 0.5 * torch.randn(N, 3, generator=g)
verts = lbs_ref.smal_forward(m, torch.zeros(N, t.nB), theta)["verts"]
R, T = render_ref.look_at_view_transform(dist, float(g.initial_seed() % 60), torch.linspace(0, 300, N))
ndc = render_ref.project_to_ndc(verts, R, T, torch.full((N,), 60.0)).contiguous()
rs = engine.raster_settings(K=K, tie_rule=RULE)
with render_ref.select_mode(MODE):
    ref1, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
got = engine.silhouette_forward(dm, ndc.to(DEV), S, rs).cpu().numpy()
gs = rng.standard_normal((N, S, S)).astype(np.float32)
print(f"seed {seed} {key} N={N} S={S} K={K} dist={dist:.2f} maxcand={ncand.max()}  pixels with candidates {int((ncand > 0).sum())}, truncated {int((ncand > K).sum())}")


fd = np.abs(got - ref1)
bad = np.argwhere(fd > 1e-4)
print(f"forward: mean|d| {fd.mean():.3e}  pixels with |d| > 1e-4: {len(bad)}")
for n_, y_, x_ in bad[:40]:
    print(f"   ({n_},{y_},{x_}) candidates {ncand[n_, y_, x_]:4d}  oracle {ref1[n_, y_, x_]:.7f}  kernel {got[n_, y_, x_]:.7f}")
if CLIP:
    st = engine.raster_stats(dm, N)
    print("raster stats:", st)


def both(gmask):
    with render_ref.select_mode(MODE):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gmask, K=K)[..., :2]
    gotg = engine.silhouette_backward(dm, ndc.to(DEV), S, torch.from_numpy(gmask).to(DEV), rs).cpu().numpy()
    return want, gotg


want, gotg = both(gs)
print(f"whole image: |want| {np.linalg.norm(want):.4e} |got| {np.linalg.norm(gotg):.4e} |diff| {np.linalg.norm(gotg - want):.4e}")
rows = []
for n, y, x in zip(*np.nonzero(ncand > 0)):
    one = np.zeros_like(gs); one[n, y, x] = gs[n, y, x]
    w, h = both(one)
    rows.append((np.linalg.norm(h - w), np.linalg.norm(w), np.linalg.norm(h), n, y, x))
rows.sort(reverse=True)
print("largest per-pixel disagreements: |diff| |want| |got|  (n, y, x)  candidates  sil oracle / kernel  upstream")
for d, w, h, n, y, x in rows[:12]:
    print(f"  {d:.4e} {w:.4e} {h:.4e}  ({n},{y},{x})  {ncand[n, y, x]:5d}  {ref1[n, y, x]:.7f} / {got[n, y, x]:.7f}  {gs[n, y, x]:+.3f}")
print(f"sum of per-pixel |diff|^2 ^.5 = {np.sqrt(sum(r[0] ** 2 for r in rows)):.4e}")
