"""Generate golden vectors by running the REAL reference code (build container only).

Run from the repo root:   python tests/golden/make_golden.py

* imports ``/root/reference/smal_model`` and ``smal_fitter.fitter`` with a stub ``config``
  module and import stubs for third-party packages that are absent here (cv2, nibabel,
  pytorch3d - names only; recipe: SURVEY.md Appendix C),
* converts the present model pickles to flat tables under ``data/models/*.npz``,
* writes ``tests/golden/lbs_<model>.npz`` (batch_rodrigues, batch_global_rigid_transformation,
  SMAL.__call__ outputs + autograd gradients) and ``tests/golden/fitter_<model>.npz``
  (SMALFitter.forward loss terms + gradients, with the renderer replaced by the oracle's
  restatement, because pytorch3d is not installed).

Only inputs and outputs are stored - no reference source travels.  The GPU box never runs this.
"""
import math
import os
import sys
import types

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
REF = "/root/reference"
sys.path.insert(0, REPO)

from smilify_amd import model_io  # noqa: E402
from oracle import render_ref  # noqa: E402

MODELS = {
    "stick": os.path.join(REF, "3D_model_prep", "SMILy_STICK.pkl"),
    "mouse": os.path.join(REF, "3D_model_prep", "SMILy_Mouse_static_joints.pkl"),
}
OUT = os.path.join(REPO, "tests", "golden")


def vertex_probe(shape, k):
    """Deterministic pseudo-random cotangent, reproducible from indices alone."""
    n = int(np.prod(shape))
    return torch.from_numpy(np.cos(0.37 * np.arange(n) * (k + 1) + k).astype(np.float32).reshape(shape))


def install_stubs(pkl_path):
    dd = model_io.read_model_pickle(pkl_path)
    cfg = types.ModuleType("config")
    cfg.SMAL_FILE = pkl_path
    cfg.dd = dd
    cfg.DEBUG = False
    cfg.ignore_sym = True
    cfg.ignore_hardcoded_body = True
    cfg.ALLOW_LIMB_SCALING = True
    cfg.STATIC_JOINT_LOCATIONS = bool(dd.get("static_joint_locs", False))
    cfg.joint_names = dd["J_names"]
    cfg.N_POSE = len(dd["J_names"]) - 1
    cfg.N_BETAS = dd["shapedirs"].shape[2]
    cfg.CANONICAL_MODEL_JOINTS = list(range(len(dd["J_names"])))
    cfg.MESH_COLOR = [0, 172, 223]
    cfg.MARKER_TYPE = [0] * len(dd["J_names"])
    cfg.MARKER_COLORS = [[0, 0, 0]] * len(dd["J_names"])
    sys.modules["config"] = cfg

    cv2 = types.ModuleType("cv2")
    cv2.MARKER_STAR = 0
    sys.modules["cv2"] = cv2

    nib = types.ModuleType("nibabel")
    eul = types.ModuleType("nibabel.eulerangles")

    def euler2angle_axis(z=0, y=0, x=0):
        # nibabel.eulerangles: euler2quat then quat2angle_axis (published formulas)
        z, y, x = z / 2.0, y / 2.0, x / 2.0
        cz, sz, cy, sy, cx, sx = math.cos(z), math.sin(z), math.cos(y), math.sin(y), math.cos(x), math.sin(x)
        q = np.array([cx * cy * cz - sx * sy * sz, cx * sy * sz + cy * cz * sx,
                      cx * cz * sy - sx * cy * sz, cx * cy * sz + sx * cz * sy])
        w, v = q[0], q[1:]
        n = np.linalg.norm(v)
        if n < 1e-12:
            return 0.0, np.array([1.0, 0, 0])
        return 2 * math.acos(max(-1, min(1, w))), v / n

    eul.euler2angle_axis = euler2angle_axis
    nib.eulerangles = eul
    sys.modules["nibabel"] = nib
    sys.modules["nibabel.eulerangles"] = eul

    p3d = types.ModuleType("pytorch3d")
    p3r = types.ModuleType("pytorch3d.renderer")
    p3s = types.ModuleType("pytorch3d.structures")
    for name in ["look_at_view_transform", "RasterizationSettings", "MeshRenderer", "MeshRasterizer", "BlendParams",
                 "PointLights", "HardPhongShader", "SoftSilhouetteShader", "Textures", "FoVPerspectiveCameras"]:
        setattr(p3r, name, None)
    p3s.Meshes = None
    sys.modules["pytorch3d"] = p3d
    sys.modules["pytorch3d.renderer"] = p3r
    sys.modules["pytorch3d.structures"] = p3s
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for m in [k for k in sys.modules if k.startswith("smal_model") or k.startswith("smal_fitter")]:
        del sys.modules[m]
    return cfg, dd


def lbs_goldens(key, pkl_path):
    cfg, dd = install_stubs(pkl_path)
    from smal_model import batch_lbs
    from smal_model.smal_torch import SMAL

    smal = SMAL("cpu")
    J = smal.J_regressor.shape[1]
    nB = smal.num_betas
    g = torch.Generator().manual_seed(20240 + len(key))
    out = {}

    # (i) batch_rodrigues
    th = torch.cat([torch.zeros(2, 3), torch.full((1, 3), 1e-9), 0.15 * torch.randn(8, 3, generator=g),
                    torch.randn(8, 3, generator=g), 3.0 * torch.randn(8, 3, generator=g)])
    out["rod_theta"] = th.numpy()
    out["rod_R"] = batch_lbs.batch_rodrigues(th).numpy()

    # (ii) chain with / without scale, trans, propagate
    B = 3
    Rs = batch_lbs.batch_rodrigues((0.4 * torch.randn(B * J, 3, generator=g))).view(B, J, 3, 3)
    Js = torch.randn(B, J, 3, generator=g)
    ls = 0.2 * torch.randn(B, J, 3, generator=g)
    bt = 0.1 * torch.randn(B, J, 3, generator=g)
    out["chain_Rs"], out["chain_Js"], out["chain_ls"], out["chain_bt"] = Rs.numpy(), Js.numpy(), ls.numpy(), bt.numpy()
    for tag, kw in [("plain", {}), ("scale", dict(betas_logscale=ls)), ("scale_trans", dict(betas_logscale=ls, betas_trans=bt)),
                    ("prop", dict(betas_logscale=ls, betas_trans=bt, propagate_scaling=True))]:
        nj, A = batch_lbs.batch_global_rigid_transformation(Rs, Js, smal.parents, num_joints=J, **kw)
        out[f"chain_{tag}_newJ"], out[f"chain_{tag}_A"] = nj.numpy(), A.numpy()

    # (iii) SMAL.__call__ + gradients
    B = 4
    beta = (0.5 * torch.randn(B, nB, generator=g)).requires_grad_()
    theta = (0.3 * torch.randn(B, J, 3, generator=g))
    theta[0] = 0.0  # exact zero pose exercises the 1e-8 quirk
    theta.requires_grad_()
    trans = (0.1 * torch.randn(B, 3, generator=g)).requires_grad_()
    ls = (0.1 * torch.randn(B, J, 3, generator=g)).requires_grad_()
    bt = (0.05 * torch.randn(B, J, 3, generator=g)).requires_grad_()
    verts, joints, Rs_o, v_shaped = smal(beta, theta, trans=trans, betas_logscale=ls, betas_trans=bt)
    loss = (verts * vertex_probe(verts.shape, 0)).sum() + (joints * vertex_probe(joints.shape, 1)).sum()
    loss.backward()
    for n, t in [("beta", beta), ("theta", theta), ("trans", trans), ("ls", ls), ("bt", bt)]:
        out[f"smal_{n}"] = t.detach().numpy()
        out[f"smal_grad_{n}"] = t.grad.numpy()
    out["smal_verts"], out["smal_joints"] = verts.detach().numpy(), joints.detach().numpy()
    out["smal_Rs"], out["smal_v_shaped"] = Rs_o.detach().numpy(), v_shaped.detach().numpy()
    out["smal_J_transformed"] = smal.J_transformed.detach().numpy()
    out["smal_loss"] = np.float32(loss.item())

    # (iii-b) the reference test-suite fixture (tests/test_triangulation_consistency.py:209-216)
    torch.manual_seed(42)
    betas0 = torch.zeros(2, nB)
    theta0 = torch.randn(2, J, 3) * 0.15
    theta0[:, 0, :] = 0.0
    _, joints0, _, _ = smal(betas0, theta0)
    out["fixture_theta"], out["fixture_joints"] = theta0.numpy(), joints0.numpy()

    # (iii-c) shared betas, no scale/trans args (None path)
    verts1, joints1, _, _ = smal(beta.detach()[:1].expand(2, nB), theta.detach()[1:3])
    out["plain_verts"], out["plain_joints"] = verts1.numpy(), joints1.numpy()
    np.savez_compressed(os.path.join(OUT, f"lbs_{key}.npz"), **out)
    print(f"lbs_{key}: J={J} nB={nB} loss={loss.item():.6f}")


def lbs_extra_goldens(key, pkl_path):
    """SMAL.__call__ inputs the first set does not touch: per-vertex offsets ``del_v`` (smal_torch.py:244-248),
    rotation-matrix ``theta`` (:288-289) and one-row inputs that torch broadcasts over the batch (trans (1,3),
    betas_logscale (1,J,3)); outputs and autograd gradients of the real reference."""
    install_stubs(pkl_path)
    from smal_model import batch_lbs
    from smal_model.smal_torch import SMAL

    smal = SMAL("cpu")
    J, nB, V = smal.J_regressor.shape[1], smal.num_betas, smal.v_template.shape[0]
    g = torch.Generator().manual_seed(4242 + len(key))
    B = 3
    beta = (0.5 * torch.randn(B, nB, generator=g)).requires_grad_()
    Rs = batch_lbs.batch_rodrigues(0.3 * torch.randn(B * J, 3, generator=g)).view(B, J, 3, 3).detach().clone().requires_grad_()
    trans = (0.1 * torch.randn(1, 3, generator=g)).requires_grad_()
    del_v = (0.01 * torch.randn(B, V, 3, generator=g)).requires_grad_()
    ls = (0.1 * torch.randn(1, J, 3, generator=g)).requires_grad_()
    bt = (0.05 * torch.randn(B, J, 3, generator=g)).requires_grad_()
    verts, joints, Rs_o, v_shaped = smal(beta, Rs, trans=trans, del_v=del_v, betas_logscale=ls, betas_trans=bt)
    loss = (verts * vertex_probe(verts.shape, 2)).sum() + (joints * vertex_probe(joints.shape, 3)).sum()
    loss.backward()
    out = dict(verts=verts.detach().numpy(), joints=joints.detach().numpy(), v_shaped=v_shaped.detach().numpy(), loss=np.float32(loss.item()))
    for n, t in [("beta", beta), ("Rs", Rs), ("trans", trans), ("del_v", del_v), ("ls", ls), ("bt", bt)]:
        out[n] = t.detach().numpy()
        out[f"grad_{n}"] = t.grad.numpy()
    # one shared (1,V,3) offset for the whole batch, axis-angle pose
    theta = (0.3 * torch.randn(B, J, 3, generator=g))
    dv1 = (0.01 * torch.randn(1, V, 3, generator=g)).requires_grad_()
    verts1, joints1, _, _ = smal(beta.detach(), theta, del_v=dv1)
    ((verts1 * vertex_probe(verts1.shape, 4)).sum() + (joints1 * vertex_probe(joints1.shape, 5)).sum()).backward()
    out.update(b_theta=theta.numpy(), b_del_v=dv1.detach().numpy(), b_verts=verts1.detach().numpy(), b_joints=joints1.detach().numpy(),
               b_grad_del_v=dv1.grad.numpy())
    np.savez_compressed(os.path.join(OUT, f"lbs_extra_{key}.npz"), **out)
    print(f"lbs_extra_{key}: loss={loss.item():.6f}")


def fitter_goldens(key, pkl_path, S=64, N=3):
    cfg, dd = install_stubs(pkl_path)
    import smal_fitter.fitter as ref_fitter

    J = len(dd["J_names"])
    g = torch.Generator().manual_seed(777 + len(key))
    dist = 2.7 if key == "stick" else 4.0
    R, T = render_ref.look_at_view_transform(dist, 10.0, torch.tensor([0.0, 40.0, -30.0])[:N])

    class _Cams:
        pass

    class FakeRenderer(torch.nn.Module):
        """Reference-shaped renderer whose arithmetic is the oracle restatement."""

        def __init__(self, image_size, device):
            super().__init__()
            self.image_size = image_size
            self.cameras = _Cams()
            self.cameras.fov = torch.full((N,), 60.0)

        def forward(self, vertices, points, faces, render_texture=False, joints_only=False):
            r = render_ref.OracleRenderer(self.image_size, R, T, self.cameras.fov)
            return r(vertices, points, faces, joints_only=joints_only)

    ref_fitter.Renderer = FakeRenderer
    rgb = torch.zeros(N, 3, S, S)
    sil_t = (torch.rand(N, 1, S, S, generator=g) > 0.5).float()
    tj = torch.rand(N, J, 2, generator=g) * S
    vis = (torch.rand(N, J, generator=g) > 0.25).long()
    fitter = ref_fitter.SMALFitter("cpu", (rgb, sil_t, tj.clone(), vis.clone()), N, -1, False)
    with torch.no_grad():
        fitter.global_rotation += 0.05 * torch.randn(N, 3, generator=g)
        fitter.joint_rotations += 0.08 * torch.randn(N, J - 1, 3, generator=g)
        fitter.trans += 0.05 * torch.randn(N, 3, generator=g)
        fitter.betas += 0.3 * torch.randn(fitter.betas.shape, generator=g)
        fitter.log_beta_scales += 0.05 * torch.randn(N, J, 3, generator=g)
        fitter.betas_trans += 0.02 * torch.randn(N, J, 3, generator=g)
    fitter.log_beta_scales.requires_grad = True
    fitter.betas_trans.requires_grad = True
    out = dict(S=np.int32(S), R=R.numpy(), T=T.numpy(), sil_target=sil_t.numpy(), target_joints=tj.numpy(),
               visibility=vis.numpy(), mean_betas=fitter.mean_betas.numpy(), betas_prec=fitter.betas_prec.numpy(),
               init_global_rotation=ref_fitter.eul_to_axis(np.array([-np.pi / 2, 0, -np.pi / 2])).astype(np.float32))
    for n, p in fitter.named_parameters():
        out[f"param_{n}"] = p.detach().numpy().copy()
    weights = [25.0, 500.0, 1.0, 1.0, 100.0, 0.1]
    out["weights"] = np.array(weights, np.float32)
    loss, objs = fitter(list(range(N)), weights, 1)
    jl, gl, tl = fitter.get_temporal(100.0)
    total = loss.mean() + jl + gl + tl
    total.backward()
    for k, v in objs.items():
        out[f"obj_{k}"] = np.float32(v.item())
    out["loss"] = np.float32(loss.item())
    out["temporal"] = np.array([jl.item(), gl.item(), tl.item()], np.float32)
    for n, p in fitter.named_parameters():
        out[f"grad_{n}"] = p.grad.numpy().copy() if p.grad is not None else np.zeros_like(p.detach().numpy())
    np.savez_compressed(os.path.join(OUT, f"fitter_{key}.npz"), **out)
    print(f"fitter_{key}: loss={loss.item():.6f}", {k: round(v.item(), 6) for k, v in objs.items()})


def export_golden(N=3):
    """Fit-result interchange formats written / read by the REAL reference code:

    * ``checkpoint_ref/<frame:04>/st1_ep7.pkl`` - the per-frame parameter dict ``SMALFitter.generate_visualization``
      hands to ``ImageExporter.export`` (fitter.py:241-261,507; optimize_to_joints.py:48-63: ``pkl.dump``), built with the
      reference's own expressions from a reference ``SMALFitter``; ``checkpoint_ref/expected.npz`` holds that fitter's
      parameters after the reference's ``load_checkpoint`` (fitter.py:352-371) has read the files back.
    * ``animation_ref.npz`` / ``animation_ref.json`` - written by the reference ``AnimationRecorder``
      (smal_fitter/neuralSMIL/animation_export.py:101-193) from ``animation_ref_inputs.npz``.
    """
    import json
    import pickle as pkl
    import shutil

    key, pkl_path = "stick", MODELS["stick"]
    cfg, dd = install_stubs(pkl_path)
    import smal_fitter.fitter as ref_fitter

    class _Cams:
        fov = torch.full((N,), 60.0)

    class FakeRenderer(torch.nn.Module):
        def __init__(self, image_size, device):
            super().__init__()
            self.image_size, self.cameras = image_size, _Cams()

    ref_fitter.Renderer = FakeRenderer
    J = len(dd["J_names"])
    S = 16
    g = torch.Generator().manual_seed(31337)
    data = (torch.zeros(N, 3, S, S), torch.zeros(N, 1, S, S), torch.zeros(N, J, 2), torch.ones(N, J).long())
    fitter = ref_fitter.SMALFitter("cpu", data, N, -1, False)
    with torch.no_grad():
        fitter.global_rotation += 0.1 * torch.randn(N, 3, generator=g)
        fitter.joint_rotations += 0.1 * torch.randn(N, J - 1, 3, generator=g)
        fitter.trans += 0.1 * torch.randn(N, 3, generator=g)
        fitter.betas += 0.3 * torch.randn(fitter.betas.shape, generator=g)
        fitter.log_beta_scales += 0.05 * torch.randn(N, J, 3, generator=g)
        fitter.betas_trans += 0.02 * torch.randn(N, J, 3, generator=g)
    fitter.fov = torch.nn.Parameter(torch.tensor([55.0, 60.0, 65.0])[:N])
    batch_range = list(range(N))
    batch_params = {  # the reference's expressions (fitter.py:241-256), evaluated on the reference's own module
        "global_rotation": fitter.global_rotation[batch_range] * fitter.global_mask,
        "joint_rotations": fitter.joint_rotations[batch_range] * fitter.rotation_mask,
        "betas": fitter.betas.expand(len(batch_range), fitter.n_betas),
        "trans": fitter.trans[batch_range],
        "fov": fitter.fov[batch_range],
        "log_betascale": fitter.log_beta_scales.expand(len(batch_range), fitter.joint_rotations.shape[1] + 1, 3),
        "betas_trans": fitter.betas_trans.expand(len(batch_range), fitter.joint_rotations.shape[1] + 1, 3),
    }
    root = os.path.join(OUT, "checkpoint_ref")
    shutil.rmtree(root, ignore_errors=True)
    for batch_id in batch_range:
        img_parameters = {k: v[batch_id].cpu().data.numpy() for (k, v) in batch_params.items()}  # fitter.py:507
        os.makedirs(os.path.join(root, "{0:04}".format(batch_id)))
        with open(os.path.join(root, "{0:04}".format(batch_id), "st1_ep7.pkl"), "wb") as f:
            pkl.dump(img_parameters, f)  # optimize_to_joints.py:56-60
    fresh = ref_fitter.SMALFitter("cpu", data, N, -1, False)
    with torch.no_grad():
        fresh.load_checkpoint(root, "st1_ep7")
    np.savez_compressed(os.path.join(root, "expected.npz"), **{n: p.detach().numpy() for n, p in fresh.named_parameters()},
                        written_betas=fitter.betas.detach().numpy())
    print("checkpoint_ref:", sorted(img_parameters), "loaded betas", fresh.betas.detach().numpy())

    from smal_fitter.neuralSMIL.animation_export import AnimationRecorder

    nJ, nBt, F = 6, 4, 5
    g = torch.Generator().manual_seed(99)
    frames = dict(global_rot=torch.randn(F, 1, 3, generator=g), joint_rot=torch.randn(F, 1, nJ - 1, 3, generator=g),
                  trans=torch.randn(F, 1, 3, generator=g), betas=torch.randn(F, 1, nBt, generator=g),
                  log_beta_scales=torch.randn(F, 1, nJ, 3, generator=g), betas_trans=torch.randn(F, 1, nJ, 3, generator=g),
                  mesh_scale=torch.rand(F, 1, 1, generator=g) + 0.5, cam_rot=torch.randn(F, 1, 3, 3, generator=g),
                  cam_trans=torch.randn(F, 1, 3, generator=g), fov=45.0 + torch.randn(F, 1, 1, generator=g))
    base = os.path.join(OUT, "animation_ref")
    rec = AnimationRecorder(output_path=base, rotation_representation="axis_angle", n_joints=nJ, n_betas=nBt,
                            joint_names=[f"J_{i}" for i in range(nJ)], parents=[-1] + list(range(nJ - 1)), fps=25.0,
                            static_joint_locs=True, ignore_hardcoded_body=True, source_checkpoint="ckpt.pth", source_input="clip.mp4",
                            model_id="golden")
    for i in range(F):
        rec.record({k: v[i] for k, v in frames.items()})
    out = rec.write()
    np.savez_compressed(os.path.join(OUT, "animation_ref_inputs.npz"), **{k: v.numpy() for k, v in frames.items()})
    print("animation_ref:", sorted(np.load(out["npz"]).files), json.load(open(out["json"]))["schema_version"])


def posedirs_golden():
    """Pose blend shapes: no SMIL model ships a non-empty table, so a small synthetic model is pickled in the
    reference's schema, loaded by the REAL reference SMAL, and given a random posedirs table."""
    import pickle
    import tempfile

    t = model_io.synthetic_model(seed=4)
    JR = t.dense_J_regressor().T.copy()
    dd = dict(f=t.faces.astype(np.int32), J_regressor=JR, kintree_table=np.stack([t.parents.astype(np.int64), np.arange(t.J)]),
              J=np.zeros((t.J, 3), np.float32), weights=t.dense_weights(), posedirs=np.zeros(0), v_template=t.v_template.astype(np.float64),
              shapedirs=t.shapedirs.T.reshape(t.V, 3, t.nB).astype(np.float64), J_names=list(t.joint_names), bs_style="lbs", bs_type="lrotmin")
    path = os.path.join(tempfile.mkdtemp(), "synthetic_tube.pkl")
    with open(path, "wb") as fh:
        pickle.dump(dd, fh, protocol=2)
    install_stubs(path)
    from smal_model.smal_torch import SMAL

    smal = SMAL("cpu")
    g = torch.Generator().manual_seed(99)
    pd = 0.02 * torch.randn(9 * (t.J - 1), 3 * t.V, generator=g)
    smal.posedirs = pd
    B = 3
    beta = (0.5 * torch.randn(B, t.nB, generator=g)).requires_grad_()
    theta = (0.4 * torch.randn(B, t.J, 3, generator=g)).requires_grad_()
    trans = (0.1 * torch.randn(B, 3, generator=g)).requires_grad_()
    verts, joints, Rs, v_shaped = smal(beta, theta, trans=trans)
    loss = (verts * vertex_probe(verts.shape, 0)).sum() + (joints * vertex_probe(joints.shape, 1)).sum()
    loss.backward()
    np.savez_compressed(os.path.join(OUT, "lbs_posedirs.npz"), posedirs=pd.numpy(), beta=beta.detach().numpy(), theta=theta.detach().numpy(),
                        trans=trans.detach().numpy(), verts=verts.detach().numpy(), joints=joints.detach().numpy(),
                        grad_beta=beta.grad.numpy(), grad_theta=theta.grad.numpy(), grad_trans=trans.grad.numpy(), seed=np.int32(4))
    print("lbs_posedirs: loss", loss.item())


def main():
    os.makedirs(os.path.join(REPO, "data", "models"), exist_ok=True)
    if "--extra" in sys.argv:  # only the files added in round 2 (the earlier ones stay byte-identical)
        for key, p in MODELS.items():
            lbs_extra_goldens(key, p)
        export_golden()
        return
    posedirs_golden()
    for key, p in MODELS.items():
        t = model_io.load_model(p)
        t.save_npz(os.path.join(REPO, "data", "models", os.path.basename(p).replace(".pkl", ".npz")))
        print("converted", p, "V", t.V, "F", t.F, "J", t.J, "nB", t.nB)
        lbs_goldens(key, p)
        fitter_goldens(key, p, S=64 if key == "stick" else 48)
        lbs_extra_goldens(key, p)
    export_golden()


if __name__ == "__main__":
    main()
