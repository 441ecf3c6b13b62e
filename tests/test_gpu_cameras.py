"""Camera ingest on the HIP path (``pytest -m gpu``): the only reference-held evidence for the screen convention.

The reference converts OpenCV calibrations to FoV cameras (smal_fitter/sleap_data/sleap_multiview_dataset.py:197-223:
``fov_y = 2 atan(H / 2fy)``, ``aspect = W fy / (H fx)``, ``R = R_cv^T Rz180``, ``T = Rz180 t_cv``) and expects the
result to project like the pinhole model ``u = fx X/Z + W/2``, ``v = fy Y/Z + H/2``.  That identity holds exactly only
under the ``S/2 - (S/2) ndc`` screen transform, so it pins ``smil_project`` (aspect != 1 included) and the aspect branch
of ``Renderer.set_camera_parameters`` (smal_fitter/p3d_renderer.py:72-125) without pytorch3d.
"""
import math

import numpy as np
import pytest
import torch

from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _calibrations(n, S, seed=0):
    """n random pinhole cameras looking roughly at the origin from 2.5-4 units away, fx != fy."""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        a, b = rng.uniform(-0.6, 0.6), rng.uniform(-0.4, 0.4)
        Ry = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]])
        Rx = np.array([[1, 0, 0], [0, math.cos(b), -math.sin(b)], [0, math.sin(b), math.cos(b)]])
        fx, fy = rng.uniform(1.2, 2.0) * S, rng.uniform(1.2, 2.0) * S
        K = np.array([[fx, 0, S / 2], [0, fy, S / 2], [0, 0, 1.0]])
        out.append((Ry @ Rx, np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), rng.uniform(2.5, 4.0)]), K))
    return out


def _pinhole(X, R_cv, t_cv, K):
    Xc = X @ R_cv.T + t_cv
    return K[0, 0] * Xc[:, 0] / Xc[:, 2] + K[0, 2], K[1, 1] * Xc[:, 1] / Xc[:, 2] + K[1, 2]


def _fov_cameras(cal, S):
    from smilify_amd import cameras

    conv = [cameras.opencv_to_fov_camera(R, t, K, (S, S)) for R, t, K in cal]
    R = torch.tensor(np.stack([c[0] for c in conv]))
    T = torch.tensor(np.stack([c[1] for c in conv]))
    fov = torch.tensor([c[2] for c in conv], dtype=torch.float32)
    aspect = torch.tensor([c[3] for c in conv], dtype=torch.float32)
    return R, T, fov, aspect


def test_hip_projection_reproduces_pinhole_with_aspect():
    """smil_project with per-image fov and aspect != 1 against the pinhole model, forward and backward."""
    from smilify_amd import engine

    S, n, P = 512, 5, 40
    cal = _calibrations(n, S)
    R, T, fov, aspect = _fov_cameras(cal, S)
    assert float((aspect - 1).abs().min()) > 1e-3
    rng = np.random.default_rng(1)
    X = rng.uniform(-0.5, 0.5, (n, P, 3)).astype(np.float32)
    cams = engine.CameraSet(R.to(DEV).contiguous(), T.to(DEV).contiguous(), fov.to(DEV), aspect.to(DEV), 1, S)
    ndc, yx = engine.project(cams, torch.from_numpy(X).to(DEV))
    yx = yx.cpu().numpy()
    for i, (R_cv, t_cv, K) in enumerate(cal):
        u, v = _pinhole(X[i].astype(np.float64), R_cv, t_cv, K)
        np.testing.assert_allclose(yx[i, :, 0], v, atol=2e-2)   # (row, col) = (v, u): reference p3d_renderer.py:137
        np.testing.assert_allclose(yx[i, :, 1], u, atol=2e-2)
    # the oracle restatement agrees to float rounding, and so do the gradients (points and fov) through the aspect term
    Xo = torch.from_numpy(X).requires_grad_()
    fov_o = fov.clone().requires_grad_()
    yx_o = render_ref.project_points_screen(Xo, R, T, fov_o, S, aspect)
    np.testing.assert_allclose(yx, yx_o.detach().numpy(), atol=2e-3)
    w = torch.from_numpy(rng.standard_normal((n, P, 2)).astype(np.float32))
    (yx_o * w).sum().backward()
    d_pts, d_fov_img = engine.project_backward(cams, torch.from_numpy(X).to(DEV), d_yx=w.to(DEV).contiguous())
    np.testing.assert_allclose(d_pts.cpu().numpy(), Xo.grad.numpy(), rtol=2e-4, atol=2e-3)
    d_fov = engine.fov_reduce(cams, d_fov_img)
    np.testing.assert_allclose(d_fov.cpu().numpy(), fov_o.grad.numpy(), rtol=2e-4, atol=1e-3)
    # z of the NDC triple is the view-space depth the rasteriser sorts by
    np.testing.assert_allclose(ndc[..., 2].cpu().numpy(), (np.einsum("npk,nkj->npj", X, R.numpy()) + T.numpy()[:, None])[..., 2], atol=1e-5)


@pytest.mark.parametrize("scalar_aspect", [False, True])
def test_renderer_aspect_branch(scalar_aspect, tables):
    """Renderer.set_camera_parameters(R, T, fov, aspect_ratio=...) -> forward: joints like the pinhole model, and the
    silhouette of an anisotropic camera against the oracle renderer with the same aspect."""
    from smilify_amd.p3d_renderer import Renderer
    from smilify_amd.smal_torch import SMAL

    t = tables("synthetic")
    S, n = 64, 3
    cal = _calibrations(n, S, seed=4)
    if scalar_aspect:  # one shared fy/fx ratio: the reference accepts a python float and broadcasts it (p3d_renderer.py:96-111)
        ratio = 1.25
        cal = [(R, tt, np.array([[K[1, 1] / ratio, 0, S / 2], [0, K[1, 1], S / 2], [0, 0, 1.0]])) for R, tt, K in cal]
    R, T, fov, aspect = _fov_cameras(cal, S)
    smal = SMAL(DEV, tables=t)
    g = torch.Generator().manual_seed(2)
    theta = 0.2 * torch.randn(n, t.J, 3, generator=g)
    verts, joints, _, _ = smal(torch.zeros(n, t.nB, device=DEV), theta.to(DEV))
    rend = Renderer(S, DEV)
    rend.set_camera_parameters(R, T, fov, aspect_ratio=(1.25 if scalar_aspect else aspect))
    sil, proj = rend(verts, joints, smal.faces)
    _, proj_only = rend(verts, joints, smal.faces, joints_only=True)
    assert torch.equal(proj, proj_only)
    J = joints.detach().cpu().numpy().astype(np.float64)
    for i, (R_cv, t_cv, K) in enumerate(cal):
        u, v = _pinhole(J[i], R_cv, t_cv, K)
        np.testing.assert_allclose(proj[i, :, 0].cpu().numpy(), v, atol=5e-3)
        np.testing.assert_allclose(proj[i, :, 1].cpu().numpy(), u, atol=5e-3)
    oren = render_ref.OracleRenderer(S, R, T, fov, aspect)
    sil_o, _ = oren(verts.detach().cpu(), joints.detach().cpu(), smal.faces.cpu())
    d = (sil.detach().cpu() - sil_o).abs().numpy()
    assert sil_o.sum() > 10 and d.mean() < 5e-6 and d.max() < 2e-3, (float(sil_o.sum()), d.mean(), d.max())


def test_batched_multiview_joint_projection(tables):
    """The neural multi-view path's use of the renderer (reference multiview_smil_regressor.py:1623-1705): LBS once per
    frame, joints projected through frames x views cameras in one call.  Image n = frame * views + view; gradients of the
    projected joints flow back to the (per-frame) joints summed over the views."""
    from smilify_amd.p3d_renderer import Renderer

    t = tables("synthetic")
    S, B, V = 128, 3, 4
    cal = _calibrations(V, S, seed=9)
    R, T, fov, aspect = _fov_cameras(cal, S)
    g = torch.Generator().manual_seed(5)
    joints = (0.4 * torch.randn(B, t.J, 3, generator=g)).to(DEV).requires_grad_()
    rend = Renderer(S, DEV, views=V)
    rend.set_camera_parameters(R, T, fov, aspect_ratio=aspect)
    none, proj = rend(joints, joints, None, joints_only=True)
    assert none is None and proj.shape == (B * V, t.J, 2)
    Jn = joints.detach().cpu().numpy().astype(np.float64)
    for b in range(B):
        for v, (R_cv, t_cv, K) in enumerate(cal):
            u, vv = _pinhole(Jn[b], R_cv, t_cv, K)
            np.testing.assert_allclose(proj[b * V + v, :, 0].detach().cpu().numpy(), vv, atol=1e-2)
            np.testing.assert_allclose(proj[b * V + v, :, 1].detach().cpu().numpy(), u, atol=1e-2)
    w = torch.randn(B * V, t.J, 2, generator=g).to(DEV)
    (proj * w).sum().backward()
    jo = joints.detach().cpu().clone().requires_grad_()
    tot = 0.0
    for v in range(V):
        po = render_ref.project_points_screen(jo, R[v:v + 1].expand(B, 3, 3), T[v:v + 1].expand(B, 3), fov[v:v + 1].expand(B), S, aspect[v:v + 1].expand(B))
        tot = tot + (po * w.cpu()[v::V]).sum()
    tot.backward()
    np.testing.assert_allclose(joints.grad.cpu().numpy(), jo.grad.numpy(), rtol=2e-4, atol=2e-3)


def _dlt_triangulate(yx_norm, R, T, fov_deg, aspect):
    """Linear (DLT) triangulation, written from the FoV camera model alone: X_view = X R + T, x_ndc = K00 X_view.x / X_view.z,
    x_s = S/2 - (S/2) x_ndc (likewise y), observations (y, x) / S.  Every view gives two equations linear in X:
    (K00 R[:,0] - x_ndc R[:,2]) . X = x_ndc T_z - K00 T_x."""
    V, J = yx_norm.shape[0], yx_norm.shape[1]
    out = np.zeros((J, 3))
    for j in range(J):
        rows, rhs = [], []
        for v in range(V):
            t = math.tan(math.radians(fov_deg[v]) / 2.0)
            k00, k11 = 1.0 / (aspect[v] * t), 1.0 / t
            x_ndc, y_ndc = 1.0 - 2.0 * yx_norm[v, j, 1], 1.0 - 2.0 * yx_norm[v, j, 0]
            rows += [k00 * R[v][:, 0] - x_ndc * R[v][:, 2], k11 * R[v][:, 1] - y_ndc * R[v][:, 2]]
            rhs += [x_ndc * T[v][2] - k00 * T[v][0], y_ndc * T[v][2] - k11 * T[v][1]]
        out[j] = np.linalg.lstsq(np.asarray(rows), np.asarray(rhs), rcond=None)[0]
    return out


@pytest.mark.parametrize("views", [6, 4, 2])
def test_projection_round_trip_recovers_the_reference_fixture_joints(views, golden):
    """The reference's own check on the projection convention (tests/test_triangulation_consistency.py:254-298, rig :73-107,
    projection :109-160), on the HIP path: the joints of ITS fixture (seed 42, theta = 0.15 randn, written by the real reference
    into tests/golden/lbs_stick.npz) -> smil_project through a ring of cameras (radius 3, elevation 15 deg, fov 60, 512 px)
    -> (y, x) / S -> DLT -> the joints again, within the reference's tolerances (0.05 max, 0.01 mean; met with 3 orders to
    spare, which is what pins the S/2 screen convention: half a pixel of offset alone would cost 2e-3)."""
    from smilify_amd import engine
    from smilify_amd.cameras import look_at_view_transform

    S = 512
    joints = torch.from_numpy(golden("lbs_stick")["fixture_joints"]).float()           # (2, 55, 3)
    B, J = joints.shape[0], joints.shape[1]
    az = torch.linspace(0, 360, views + 1)[:views]
    R, T = look_at_view_transform(3.0, torch.full_like(az, 15.0), az)
    fov = torch.full((views,), 60.0)
    cams = engine.CameraSet(R.to(DEV).contiguous(), T.to(DEV).contiguous(), fov.to(DEV), None, views, S)
    _, yx = engine.project(cams, joints.to(DEV).contiguous(), want_ndc=False)          # (B * views, J, 2), image = frame * views + view
    kp = (yx / S).cpu().numpy().astype(np.float64).reshape(B, views, J, 2)
    assert kp.min() > 0.0 and kp.max() < 1.0                                           # every joint is inside every image
    Rn, Tn = R.numpy().astype(np.float64), T.numpy().astype(np.float64)
    err = np.stack([np.linalg.norm(_dlt_triangulate(kp[b], Rn, Tn, [60.0] * views, [1.0] * views) - joints[b].numpy(), axis=-1)
                    for b in range(B)])
    assert err.max() < 0.05 and err.mean() < 0.01, (err.max(), err.mean())             # the reference's tolerances
    assert err.max() < 2e-4, err.max()                                                 # what exact conventions give in fp32
    # the oracle's projection says the same pixels
    for v in range(views):
        po = render_ref.project_points_screen(joints, R[v:v + 1].expand(B, 3, 3), T[v:v + 1].expand(B, 3), fov[v:v + 1].expand(B), S)
        np.testing.assert_allclose(yx.cpu().numpy().reshape(B, views, J, 2)[:, v], po.numpy(), atol=2e-3)
