"""CPU only: how far are the fp32 oracle and the HIP kernel from a float64 evaluation of the same scene?
   python tools/dbg/fp64_check.py <seed> clip <fuzz_one log with the kernel's values>
Builds a float64 copy of oracle/raster_oracle.c under /tmp (same source, `float` -> `double`), renders the fuzz scene of
tools/dbg/fuzz_one.py with both, and compares them at the pixels the log lists."""
import os, re, sys
import numpy as np, torch
REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from smilify_amd import model_io
from oracle import render_ref, lbs_ref, fitter_ref
from conftest import oracle_model

seed = int(sys.argv[1]); CLIP = sys.argv[2] == "clip"; log = sys.argv[3]
import fp64_oracle
lib = fp64_oracle.load(); dp, ip = fp64_oracle.DP, fp64_oracle.IP

rng = np.random.default_rng(seed)
key = ["synthetic", "stick", "mouse"][int(rng.integers(0, 3))]
t = {"synthetic": model_io.synthetic_model,
     "stick": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_STICK.npz")),
     "mouse": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_Mouse_static_joints.npz"))}[key]()
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
ndc = render_ref.project_to_ndc(verts, R, T, torch.full((N,), 60.0)).contiguous().numpy()
with render_ref.select_mode(1):
    ref32, ncand = render_ref.silhouette_forward_np(ndc, t.faces, S, K=K)
print(f"seed {seed} {key} N={N} S={S} K={K} dist={dist:.2f}; largest |x_ndc|, |y_ndc| of a vertex in front of z_clip: "
      f"{np.abs(ndc[..., :2][ndc[..., 2] > 5e-4]).max():.1f}")
ref64 = np.empty((N, S, S))
for n in range(N):  # the clipped mesh of every image, exactly as the fp32 oracle renders it
    plan = render_ref._clip_plan(ndc[n:n + 1], t.faces.astype(np.int32))[0]
    va, fa = (ndc[n], t.faces) if plan is None else (plan[0], plan[1])
    va = np.ascontiguousarray(va, np.float64); fa = np.ascontiguousarray(fa, np.int32)
    out = np.empty((1, S, S)); nc = np.empty((1, S, S), np.int32)
    rc = lib.oracle_silhouette_forward(va.ctypes.data_as(dp), fa.ctypes.data_as(ip), 1, va.shape[0], fa.shape[0], S,
                                       render_ref.BLUR_RADIUS, render_ref.SIGMA, K, out.ctypes.data_as(dp), nc.ctypes.data_as(ip),
                                       ip(), dp(), dp())
    assert rc == 0
    ref64[n] = out[0]
rows = re.findall(r"\((\d+),(\d+),(\d+)\) candidates\s+\d+\s+oracle ([\d.]+)\s+kernel ([\d.]+)", open(log).read())
print("pixel        float64     fp32 oracle (err)      HIP kernel (err)")
eo, ek = [], []
for n, y, x, o, k in rows:
    n, y, x, o, k = int(n), int(y), int(x), float(o), float(k)
    d = ref64[n, y, x]
    eo.append(abs(o - d)); ek.append(abs(k - d))
    print(f"({n},{y},{x})  {d:.7f}   {o:.7f} ({o - d:+.2e})   {k:.7f} ({k - d:+.2e})")
print(f"median |error| against float64 over these {len(rows)} pixels: fp32 oracle {np.median(eo):.2e}, HIP kernel {np.median(ek):.2e}; "
      f"pixels where the kernel is the nearer one: {sum(a > b for a, b in zip(eo, ek))}")
print(f"whole image: mean |fp32 oracle - float64| = {np.abs(ref32 - ref64).mean():.3e}")
