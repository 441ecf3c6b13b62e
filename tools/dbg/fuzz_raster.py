import os, sys
import numpy as np, torch
REPO = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests")); sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from smilify_amd import engine, model_io
from oracle import render_ref, lbs_ref, fitter_ref
from conftest import oracle_model
import fp64_oracle
DEV = "cuda:0"
models = {"synthetic": model_io.synthetic_model(),
          "stick": model_io.load_model(os.path.join(REPO, "data/models/SMILy_STICK.npz")),
          "mouse": model_io.load_model(os.path.join(REPO, "data/models/SMILy_Mouse_static_joints.npz"))}
dms = {k: engine.DeviceModel(t, DEV) for k, t in models.items()}
def scene(t, N, S, dist, seed, scale=1.0):
    m = oracle_model(t)
    g = torch.Generator().manual_seed(seed)
    theta = 0.3 * torch.randn(N, t.J, 3, generator=g)
    theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation()) + 0.5 * torch.randn(N, 3, generator=g)
    verts = lbs_ref.smal_forward(m, torch.zeros(N, t.nB), theta)["verts"] * scale
    R, T = render_ref.look_at_view_transform(dist, float(g.initial_seed() % 60), torch.linspace(0, 300, N))
    return render_ref.project_to_ndc(verts, R, T, torch.full((N,), 60.0)).contiguous()
bad = 0
CLIP = len(sys.argv) > 3 and sys.argv[3] == "clip"  # camera INSIDE the mesh's reach: faces cross z_clip and get cut (clip_faces)
QUEUE = len(sys.argv) > 3 and sys.argv[3] == "queue"  # tie_rule="reference_queue" against the oracle's FAITHFUL queue (select_mode 0)
MODE, RULE = (0, "reference_queue") if QUEUE else (1, "depth_face_id")
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    rng = np.random.default_rng(seed)
    key = ["synthetic", "stick", "mouse"][int(rng.integers(0, 3))]
    t, dm = models[key], dms[key]
    S = int(rng.integers(9, 140)); K = int(rng.choice([1, 2, 5, 17, 64, 100, 128]))
    dist = float(np.exp(rng.uniform(np.log(1.2), np.log(40.0)))) * (1.5 if key == "mouse" else 1.0)
    if CLIP:
        dist = float(rng.uniform(0.05, 0.9)) * (1.5 if key == "mouse" else 1.0)
    N = int(rng.integers(1, 4))
    ndc = scene(t, N, S, dist, seed)
    with render_ref.select_mode(MODE):
        ref1, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
    got = engine.silhouette_forward(dm, ndc.to(DEV), S, engine.raster_settings(K=K, tie_rule=RULE)).cpu().numpy()
    d1 = np.abs(got - ref1)
    # (cut faces carry vertices at |xy| up to 1e3 NDC units: fp32 cancellation on both sides; looser there)
    ok = d1.mean() < (2e-4 if CLIP else 5e-6) and np.mean(d1 > 1e-4) < (3e-2 if CLIP else 5e-3) and (CLIP or d1[ncand <= K].max(initial=0.0) < 3e-4)
    gs = torch.from_numpy(rng.standard_normal((N, S, S)).astype(np.float32))
    with render_ref.select_mode(MODE):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy(), K=K)[..., :2]
    cd = engine.ClipDepth(DEV, N) if CLIP else None  # (the depth gradients of cut edges' end points, beside d_ndc)
    gotg = engine.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV), engine.raster_settings(K=K, tie_rule=RULE), clip_depth=cd).cpu().numpy()
    nrm = np.linalg.norm(want)
    cos = (gotg * want).sum() / (np.linalg.norm(gotg) * nrm + 1e-30) if nrm > 0 else 1.0
    ok = ok and (cos > (0.97 if CLIP else 0.99) or nrm < 1e-6) and np.isfinite(gotg).all()
    st = engine.raster_stats(dm, N)
    zk = zo = 0.0
    if CLIP and st['unclipped_faces'] == 0:
        # The depth gradient of a cut edge's end points is J . g_xy(new vertex), J = d(crossing point)/d(depth), and the crossing point
        # slides ALONG the edge's line when a depth changes.  With the crossing far outside the image (the usual fuzz scene, |xy| ~ 1e3)
        # the only edge of the cut face that crosses the image IS that line, every pixel's contribution to g_xy is perpendicular to
        # it, and the product is a rounding residual |J| |g| ~ 1e6 times larger than itself: the fp32 oracle's own value changes
        # by its full size when the vertices move by one ulp (profiles/r5_fuzz.md).  So both are measured against the same pass in
        # float64 (fp64_oracle.py), on the scale the depth gradient acts on - the view-space gradient the vertices receive through
        # xy, |d_ndc| / z - and the yardstick is the fp32 oracle's own distance from float64, the largest of three samples: the scene
        # itself and two copies with every xy one ulp up or down.  The kernel may be 4x as far, or 2e-3 of that scale.
        # (Well-conditioned cuts are compared directly: tests/test_gpu_raster_clip.py.)
        with render_ref.select_mode(MODE):
            wfull = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy(), K=K).astype(np.float64)
        gz = cd.dense(t.V).numpy()
        for n in range(N):
            nd = ndc[n].numpy()
            if not bool((nd[:, 2] < render_ref._Z_CLIP).any()):
                ok = ok and bool((gz[n] == 0).all())
                continue
            z64 = fp64_oracle.depth_gradient(render_ref, nd, t.faces, S, K, gs[n].numpy(), select_mode=MODE)
            scale = float((np.linalg.norm(wfull[n, :, :2], axis=-1) / np.maximum(np.abs(nd[:, 2]), 5e-4)).max()) + 1e-300
            zk = max(zk, float(np.abs(gz[n] - z64).max() / scale)); zo = max(zo, float(np.abs(wfull[n, :, 2] - z64).max() / scale))
            ok = ok and bool((gz[n][(z64 == 0) & (wfull[n, :, 2] == 0)] == 0).all())  # (vertices off every cut edge receive nothing)
            for k in range(2):
                up = np.random.default_rng(seed * 4 + k).integers(0, 2, size=(t.V, 2)).astype(bool)
                nd2 = nd.copy()
                nd2[:, :2] = np.where(up, np.nextafter(nd[:, :2], np.float32(np.inf)), np.nextafter(nd[:, :2], np.float32(-np.inf)))
                with render_ref.select_mode(MODE):
                    w2 = render_ref.silhouette_backward_np(nd2[None], t.faces, S, gs[n:n + 1].numpy(), K=K)[0, :, 2].astype(np.float64)
                z2 = fp64_oracle.depth_gradient(render_ref, nd2, t.faces, S, K, gs[n].numpy(), select_mode=MODE)
                zo = max(zo, float(np.abs(w2 - z2).max() / scale))
        ok = ok and zk <= max(4.0 * zo, 2e-3) and np.isfinite(gz).all() and int(cd.counter[1]) == 0
    over = st['unclipped_faces'] > 0  # more than 256 cut faces in an image: the ones beyond the clip tables are rendered whole (documented
    ok = ok or over                   # capacity, counted by smil_raster_stats) - the oracle cuts them all, so the scene cannot agree
    print(f"cut={st['straddling_faces']:4d} lost={st['unclipped_faces']:3d} ", end="")
    print(f"seed {seed:3d} {key:9s} N={N} S={S:3d} K={K:3d} dist={dist:6.2f} maxcand={ncand.max():5d} mean|d|={d1.mean():.2e} frac>1e-4={np.mean(d1>1e-4):.1e} cos={cos:.5f} dz: kernel {zk:.1e} oracle {zo:.1e} {('ok (over clip capacity)' if over else 'ok') if ok else 'FAIL'}", flush=True)
    bad += (not ok)
print("failures", bad)
