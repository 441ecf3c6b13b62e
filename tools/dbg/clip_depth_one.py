# one clip-mode scene of tools/dbg/fuzz_raster.py: the depth-gradient channel vertex by vertex, HIP against the oracle
#   python tools/dbg/clip_depth_one.py <seed>
import os, sys
import numpy as np, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from smilify_amd import engine, model_io
from oracle import render_ref, lbs_ref, fitter_ref
from conftest import oracle_model
DEV = "cuda:0"
seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
key = ["synthetic", "stick", "mouse"][int(rng.integers(0, 3))]
t = {"synthetic": model_io.synthetic_model,
     "stick": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_STICK.npz")),
     "mouse": lambda: model_io.load_model(os.path.join(REPO, "data/models/SMILy_Mouse_static_joints.npz"))}[key]()
dm = engine.DeviceModel(t, DEV)
S = int(rng.integers(9, 140)); K = int(rng.choice([1, 2, 5, 17, 64, 100, 128]))
dist = float(np.exp(rng.uniform(np.log(1.2), np.log(40.0)))) * (1.5 if key == "mouse" else 1.0)
dist = float(rng.uniform(0.05, 0.9)) * (1.5 if key == "mouse" else 1.0)
N = int(rng.integers(1, 4))
m = oracle_model(t)
g = torch.Generator().manual_seed(seed)
theta = 0.3 * torch.randn(N, t.J, 3, generator=g)
theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation()) + 0.5 * torch.randn(N, 3, generator=g)
verts = lbs_ref.smal_forward(m, torch.zeros(N, t.nB), theta)["verts"]
R, T = render_ref.look_at_view_transform(dist, float(g.initial_seed() % 60), torch.linspace(0, 300, N))
ndc = render_ref.project_to_ndc(verts, R, T, torch.full((N,), 60.0)).contiguous()
gs = torch.from_numpy(rng.standard_normal((N, S, S)).astype(np.float32))
with render_ref.select_mode(1):
    want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy(), K=K)
cd = engine.ClipDepth(DEV, N)
got = engine.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV), engine.raster_settings(K=K), clip_depth=cd).cpu().numpy()
gz = cd.dense(t.V).numpy()
print(f"seed {seed} {key} N={N} S={S} K={K} dist={dist:.2f}  counter {cd.counter.cpu().tolist()} ranges {cd.range.cpu().tolist()}")
for n in range(N):
    wz = want[n, :, 2].astype(np.float64)
    nz = np.linalg.norm(wz)
    cos = float((gz[n] * wz).sum() / (np.linalg.norm(gz[n]) * nz + 1e-300)) if nz > 0 else 1.0
    cxy = float((got[n] * want[n, :, :2]).sum() / (np.linalg.norm(got[n]) * np.linalg.norm(want[n, :, :2]) + 1e-300))
    print(f"image {n}: cos xy {cxy:.5f}  cos z {cos:.5f}  |want z| {nz:.3e} |got z| {np.linalg.norm(gz[n]):.3e}")
    for v in np.argsort(-np.abs(wz))[:6]:
        print(f"   vertex {v:4d} z={ndc[n, v, 2]:+.3e} xy=({ndc[n, v, 0]:+.2f},{ndc[n, v, 1]:+.2f})  oracle {wz[v]:+.4e}  kernel {gz[n, v]:+.4e}")


import fp64_oracle
for n in range(N):
    z64 = fp64_oracle.depth_gradient(render_ref, ndc[n].numpy(), t.faces, S, K, gs[n].numpy())
    scale = float((np.linalg.norm(want[n, :, :2], axis=-1) / np.maximum(np.abs(ndc[n, :, 2].numpy()), 5e-4)).max())
    d_or = np.abs(want[n, :, 2] - z64).max(); d_k = np.abs(gz[n] - z64).max()
    print(f"image {n}: float64 |z-grad| max {np.abs(z64).max():.3e}; away from it: fp32 oracle {d_or:.3e} ({d_or / scale:.1e} of the scale), "
          f"kernel {d_k:.3e} ({d_k / scale:.1e})")

# per crossing of image 0: the kernel's two entries next to the oracle's for the same edge (several cut faces may share an edge: listed in
# the order each side emits them), with |J| |g| - the size of the two terms whose difference a depth gradient is
n = 0
plan = render_ref._clip_plan(ndc[n:n + 1].numpy(), t.faces.astype(np.int32))[0]
if plan is not None:
    va, fa, src, coef = plan
    with render_ref.select_mode(1):
        g_or = render_ref.silhouette_backward_np(va[None], fa, S, gs[n:n + 1].numpy(), K=K, _clipped=True)[0]
    V0 = t.V
    by_pair = {}
    for j in range(len(src)):
        a, b = int(src[j, 0]), int(src[j, 1])
        gj = g_or[V0 + j, :2].astype(np.float64)
        dza, dzb = render_ref.clip_depth_gradient(va[a], va[b], gj, 5e-4)
        jx = render_ref.clip_depth_gradient(va[a], va[b], np.array([1.0, 0.0]), 5e-4)
        jy = render_ref.clip_depth_gradient(va[a], va[b], np.array([0.0, 1.0]), 5e-4)
        by_pair.setdefault((a, b), []).append((dza, dzb, np.hypot(jx[0], jy[0]) * np.linalg.norm(gj), np.hypot(jx[1], jy[1]) * np.linalg.norm(gj), gj, va[V0 + j]))
    rg = cd.range.cpu().numpy().reshape(-1, 2); vx = cd.vertex.cpu().numpy(); dzs = cd.dz.cpu().numpy()
    first, cnt = int(rg[n, 0]), int(rg[n, 1])
    kern = {}
    for e in range(first, first + cnt, 2):
        kern.setdefault((int(vx[e]), int(vx[e + 1])), []).append((float(dzs[e]), float(dzs[e + 1])))
    print(f"image 0: {len(src)} crossings in the oracle, {cnt // 2} in the kernel; edges only one side has: {sorted(set(by_pair) ^ set(kern))}")
    rows = []
    for pair in by_pair:
        ko = sum(x[0] for x in by_pair[pair]), sum(x[1] for x in by_pair[pair])
        kk = sum(x[0] for x in kern.get(pair, [])), sum(x[1] for x in kern.get(pair, []))
        size = sum(x[2] for x in by_pair[pair]), sum(x[3] for x in by_pair[pair])
        rows.append((max(abs(kk[0] - ko[0]) / (size[0] + 1e-300), abs(kk[1] - ko[1]) / (size[1] + 1e-300)), pair, ko, kk, size))
    for r, pair, ko, kk, size in sorted(rows, key=lambda x: -x[0])[:10]:
        print(f"   edge {pair}: oracle dz ({ko[0]:+.4e}, {ko[1]:+.4e})  kernel ({kk[0]:+.4e}, {kk[1]:+.4e})  |J||g| ({size[0]:.3e}, {size[1]:.3e})  rel {r:.1e}")
        for x in by_pair[pair]:
            print(f"        oracle crossing at xy ({x[5][0]:+.3e}, {x[5][1]:+.3e}) g_xy ({x[4][0]:+.4e}, {x[4][1]:+.4e}) dz ({x[0]:+.4e}, {x[1]:+.4e})")
        for x in kern.get(pair, []):
            print(f"        kernel entry dz ({x[0]:+.4e}, {x[1]:+.4e})")
