"""GPU tests of the drop-in surface added in round 2 (``pytest -m gpu``): ``del_v`` / rotation-matrix ``theta`` /
one-row (broadcast) inputs of ``SMAL.__call__`` against vectors of the real reference, the reference-written per-frame
checkpoint, hipGraph invalidation, the z_clip cull and the renderer's topology cache."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, vertex_probe
from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_smal_del_v_rotation_matrices_and_broadcast_inputs_match_reference(key, golden, tables):
    """reference smal_torch.py:244-248 (del_v), :288-289 (theta given as (B,J,3,3) matrices), and the inputs torch
    broadcasts over the batch there (trans (1,3), betas_logscale (1,J,3)): outputs and every gradient of the REAL
    reference's autograd."""
    from smilify_amd.smal_torch import SMAL

    g = golden(f"lbs_extra_{key}")
    smal = SMAL(DEV, tables=tables(key))
    leaf = {n: torch.from_numpy(g[n]).to(DEV).requires_grad_() for n in ("beta", "Rs", "trans", "del_v", "ls", "bt")}
    assert leaf["trans"].shape[0] == 1 and leaf["ls"].shape[0] == 1 and leaf["Rs"].dim() == 4
    verts, joints, Rs_o, v_shaped = smal(leaf["beta"], leaf["Rs"], trans=leaf["trans"], del_v=leaf["del_v"], betas_logscale=leaf["ls"],
                                         betas_trans=leaf["bt"])
    np.testing.assert_allclose(verts.detach().cpu().numpy(), g["verts"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(joints.detach().cpu().numpy(), g["joints"], rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(v_shaped.detach().cpu().numpy(), g["v_shaped"], rtol=2e-5, atol=2e-6)
    np.testing.assert_allclose(Rs_o.detach().cpu().numpy(), g["Rs"], rtol=0, atol=1e-7)
    loss = (verts * vertex_probe(verts.shape, 2).to(DEV)).sum() + (joints * vertex_probe(joints.shape, 3).to(DEV)).sum()
    assert abs(loss.item() - float(g["loss"])) <= 2e-4 * abs(float(g["loss"])) + 1e-4
    loss.backward()
    for n, t in leaf.items():
        want = g[f"grad_{n}"]
        assert t.grad is not None and tuple(t.grad.shape) == want.shape, n
        scale = np.abs(want).max() + 1e-12
        np.testing.assert_allclose(t.grad.cpu().numpy() / scale, want / scale, rtol=0, atol=3e-4, err_msg=f"{key}/{n}")
    # one (1,V,3) offset shared by the batch, axis-angle pose
    dv1 = torch.from_numpy(g["b_del_v"]).to(DEV).requires_grad_()
    v1, j1, _, _ = smal(leaf["beta"].detach(), torch.from_numpy(g["b_theta"]).to(DEV), del_v=dv1)
    np.testing.assert_allclose(v1.detach().cpu().numpy(), g["b_verts"], rtol=1e-4, atol=5e-6)
    ((v1 * vertex_probe(v1.shape, 4).to(DEV)).sum() + (j1 * vertex_probe(j1.shape, 5).to(DEV)).sum()).backward()
    want = g["b_grad_del_v"]
    np.testing.assert_allclose(dv1.grad.cpu().numpy() / np.abs(want).max(), want / np.abs(want).max(), rtol=0, atol=3e-4)


def test_smal_rejects_shapes_that_would_read_past_a_buffer(tables):
    from smilify_amd.smal_torch import SMAL

    t = tables("synthetic")
    smal = SMAL(DEV, tables=t)
    B = 3
    beta, theta = torch.zeros(B, t.nB, device=DEV), torch.zeros(B, t.J, 3, device=DEV)
    for kw in (dict(trans=torch.zeros(2, 3, device=DEV)), dict(del_v=torch.zeros(2, t.V, 3, device=DEV)),
               dict(del_v=torch.zeros(B, t.V - 1, 3, device=DEV)), dict(betas_logscale=torch.zeros(2, t.J, 3, device=DEV)),
               dict(betas_trans=torch.zeros(B, t.J - 1, 3, device=DEV)), dict(v_template=torch.zeros(t.V + 1, 3, device=DEV))):
        with pytest.raises(ValueError):
            smal(beta, theta, **kw)
    with pytest.raises(ValueError):
        smal(torch.zeros(2, t.nB, device=DEV), theta)
    # a custom template receives the batch-summed vertex gradient
    vt = torch.from_numpy(t.v_template).to(DEV).requires_grad_()
    verts, _, _, _ = smal(beta, theta, v_template=vt)
    verts.sum().backward()
    assert vt.grad.shape == (t.V, 3) and torch.allclose(vt.grad, torch.full_like(vt.grad, float(B)), atol=1e-4)


def test_load_checkpoint_reads_what_the_reference_wrote(tables):
    """tests/golden/checkpoint_ref: per-frame pickles in the reference exporter's layout and key set
    (optimize_to_joints.py:48-63, fitter.py:241-261,507); expected.npz = the reference's own load_checkpoint result."""
    from smilify_amd import synthetic

    root = os.path.join(GOLDEN, "checkpoint_ref")
    exp = np.load(os.path.join(root, "expected.npz"))
    t = tables("stick")
    N = exp["trans"].shape[0]
    f = synthetic.make_problem(t, N, 1, 32, DEV, seed=1, window=N)
    f.load_checkpoint(root, "st1_ep7")
    for n in ("global_rotation", "joint_rotations", "trans", "betas"):
        np.testing.assert_allclose(getattr(f, n).detach().cpu().numpy(), exp[n], rtol=0, atol=1e-7, err_msg=n)
    # the reference averages the per-frame scale tables into ONE (J,3) table (fitter.py:371)
    np.testing.assert_allclose(f.log_beta_scales.detach().cpu().numpy().reshape(exp["log_beta_scales"].shape), exp["log_beta_scales"], atol=1e-7)
    loss, _ = f(list(range(N)), synthetic.STAGE1_WEIGHTS, 1)  # still a working fitter
    assert torch.isfinite(loss)
    # and what this build exports has the reference's key set, shapes and (for untouched parameters) values
    want = pickle.load(open(os.path.join(root, "0001", "st1_ep7.pkl"), "rb"))
    got = f.export_parameters(1)
    assert sorted(got) == sorted(want)
    for k in want:
        assert np.asarray(got[k]).shape == np.asarray(want[k]).shape, k
    np.testing.assert_allclose(got["joint_rotations"], want["joint_rotations"], atol=1e-7)
    np.testing.assert_allclose(got["trans"], want["trans"], atol=1e-7)


def test_graph_replay_is_invalidated_by_camera_mask_and_workspace_changes(tables):
    """A captured iteration bakes in device addresses; after set_cameras / a re-assigned mask / a regrown workspace the
    next fit_step_graph must re-capture and agree with the eager step."""
    from smilify_amd import synthetic
    from smilify_amd.cameras import look_at_view_transform

    t = tables("synthetic")

    def make():
        f = synthetic.make_problem(t, 4, 2, 40, DEV, radius=2.3, seed=9, window=2)
        f.begin_stage(synthetic.STAGE1_LR)
        return f

    fe, fg = make(), make()
    w, wt = synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL

    def both():
        a = fe.fit_step(w, wt).clone()
        b = fg.fit_step_graph(w, wt).clone()
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4, atol=1e-6)

    both()
    both()
    first = fg._graph["graph"]
    R, T = look_at_view_transform(2.6, 25.0, np.array([30.0, 200.0]), device=DEV)
    for f in (fe, fg):
        f.set_cameras(R, T)
    assert fg._graph is None
    both()
    assert fg._graph["graph"] is not first
    second = fg._graph["graph"]
    mask = torch.ones(t.J - 1, 3, device=DEV)
    mask[2:] = 0.0
    for f in (fe, fg):
        f.rotation_mask = mask.clone()
    both()
    assert fg._graph["graph"] is not second
    third = fg._graph["graph"]
    both()
    assert fg._graph["graph"] is third          # nothing changed: replayed
    for f in (fe, fg):                          # the reference's idiom: a mask edited IN PLACE (same tensor, same address)
        f.rotation_mask[0] = 0.0
        f.global_mask[0, 1] = 0.0
    both()                                      # the graph reads the refreshed mask table ...
    both()
    assert fg._graph["graph"] is third          # ... without a re-capture
    np.testing.assert_allclose(fg._pose.cpu().numpy(), fe._pose.cpu().numpy(), rtol=2e-4, atol=2e-6)
    from smilify_amd import engine as _eng

    held = dict(_eng._SHARED_WS)  # (kept alive: the next buffer gets another address)
    _eng._SHARED_WS.clear()
    fg.device_model._ws = None                  # as a larger call on this device would do: workspace replaced
    both()
    del held
    assert fg._graph["graph"] is not third


def test_z_clip_culls_faces_entirely_nearer_than_half_znear(tables):
    """MeshRasterizer's z_clip_value = znear / 2 (5e-4): a mesh lying entirely between the camera plane and z_clip renders
    nothing; with the cull disabled (z_clip = 0) the same vertices do render.  Same rule in the oracle."""
    from smilify_amd import engine as eng

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 32
    g = torch.Generator().manual_seed(0)
    ndc = torch.empty(1, t.V, 3)
    ndc[..., :2] = 0.6 * (torch.rand(1, t.V, 2, generator=g) - 0.5)
    ndc[..., 2] = 1e-4 + 3e-4 * torch.rand(1, t.V, generator=g)       # every vertex in (1e-4, 4e-4) < 5e-4
    assert float(eng.silhouette_forward(dm, ndc.to(DEV), S).abs().max()) == 0.0
    ref, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    assert float(np.abs(ref).max()) == 0.0
    rs = eng.raster_settings()
    rs.z_clip = 0.0
    got = eng.silhouette_forward(dm, ndc.to(DEV), S, rs).cpu().numpy()
    render_ref.set_z_clip(0.0)
    try:
        ref0, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    finally:
        render_ref.set_z_clip(5e-4)
    assert ref0.sum() > 1.0 and np.abs(got - ref0).mean() < 1e-4


def test_faces_straddling_z_clip_are_counted_and_the_fitter_warns(tables):
    """Faces with one or two vertices nearer than z_clip are cut at the plane (next test) and counted per launch by the setup
    kernel (smil_raster_stats); the fitter warns once: the mesh has reached the camera."""
    import warnings

    from smilify_amd import engine as eng
    from smilify_amd import synthetic

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S, N = 32, 3
    g = torch.Generator().manual_seed(1)
    ndc = torch.empty(N, t.V, 3)
    ndc[..., :2] = 0.6 * (torch.rand(N, t.V, 2, generator=g) - 0.5)
    ndc[..., 2] = 1.0 + torch.rand(N, t.V, generator=g)
    eng.silhouette_forward(dm, ndc.to(DEV), S)
    st = eng.raster_stats(dm, N)
    assert st["straddling_faces"] == 0 and st["tiles"] > 0
    ndc[1, 5, 2] = 2e-4                                                # one vertex of image 1 behind z_clip = 5e-4
    ndc[2, 7, 2] = 1e-4
    ndc[2, 9, 2] = 3e-4
    want = sum(int(((ndc[n][torch.from_numpy(t.faces.astype(np.int64))][..., 2] < 5e-4).any(1)
                    & ~(ndc[n][torch.from_numpy(t.faces.astype(np.int64))][..., 2] < 5e-4).all(1)).sum()) for n in range(N))
    eng.silhouette_forward(dm, ndc.to(DEV), S)
    got = eng.raster_stats(dm, N)["straddling_faces"]
    assert got == want > 0, (got, want)
    # the fitter: a mesh pushed into the camera plane
    f = synthetic.make_problem(t, 2, 1, S, DEV, radius=2.2, seed=3, window=2)
    f.begin_stage(synthetic.STAGE1_LR)
    f.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)
    with warnings.catch_warnings():
        warnings.simplefilter("error")
        assert f.straddling_faces() == 0
    with torch.no_grad():
        f.trans[:, 2] += 2.2 / 1.0  # towards the camera: the body now crosses z = 0
        cam = f.renderer.cameras
        # move along the viewing direction of camera 0 until the mesh centre sits on the camera plane
        f.trans.copy_((-cam.T[0] @ cam.R[0].T).expand_as(f.trans))
    f.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)
    with pytest.warns(RuntimeWarning, match="straddle"):
        assert f.straddling_faces() > 0
    with warnings.catch_warnings():
        warnings.simplefilter("error")                                 # once per fitter
        assert f.straddling_faces() > 0


def test_renderer_topology_cache_is_keyed_by_content(tables):
    from smilify_amd.p3d_renderer import Renderer
    from smilify_amd.smal_torch import SMAL

    t = tables("synthetic")
    smal = SMAL(DEV, tables=t)
    rend = Renderer(32, DEV)
    rend.bind_model(smal.device_model)
    verts, joints, _, _ = smal(torch.zeros(1, t.nB, device=DEV), torch.zeros(1, t.J, 3, device=DEV))
    sil_a, _ = rend(verts, joints, smal.faces)
    assert rend._device_model(smal.faces, t.V) is smal.device_model and not rend._topologies
    # int32 copies (a fresh temporary per call in the reference's calling code) hit the bound model by content
    assert rend._device_model(smal.faces.to(torch.int32), t.V) is smal.device_model
    # same (V, F) counts, different triangles: must NOT be rendered with the bound model's table
    other = smal.faces.clone()
    other[: t.F // 2] = other[: t.F // 2].flip(0)[:, [0, 2, 1]]
    other[::3] = other[0]
    dm_other = rend._device_model(other, t.V)
    assert dm_other is not smal.device_model and len(rend._topologies) == 1
    assert rend._device_model(other.clone().to(torch.int32), t.V) is dm_other and len(rend._topologies) == 1
    sil_b, _ = rend(verts, joints, other)
    assert (sil_a - sil_b).abs().max() > 1e-3


def test_batches_beyond_65535_frames_and_images(tables):
    """gridDim.y stops at 65 535; BASELINE config 5 holds 147 456 images per GPU and the whole 65 536-frame sequence may sit
    on one GPU.  Projection and LBS with 70 000 rows: first and last rows equal the same rows computed in a small batch."""
    from smilify_amd import engine as eng

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    B = 70000
    g = torch.Generator().manual_seed(3)
    theta = (0.2 * torch.randn(B, t.J, 3, generator=g)).to(DEV)
    trans = (0.05 * torch.randn(B, 3, generator=g)).to(DEV)
    beta = torch.zeros(t.nB, device=DEV)
    big = eng.lbs_forward(dm, beta, theta, trans=trans, shared_beta=True, trans_after_joints=True)
    sel = torch.tensor([0, 1, 65535, 65536, B - 1], device=DEV)
    small = eng.lbs_forward(dm, beta, theta[sel].contiguous(), trans=trans[sel].contiguous(), shared_beta=True, trans_after_joints=True)
    assert torch.equal(big["verts"][sel], small["verts"]) and torch.equal(big["joints"][sel], small["joints"])
    R = torch.eye(3, device=DEV)[None].contiguous()
    R[0, 0, 0] = R[0, 2, 2] = -1.0
    T = torch.tensor([[0.0, 0.0, 2.7]], device=DEV)
    cams = eng.CameraSet(R, T, torch.full((1,), 60.0, device=DEV), None, 1, 64)
    ndc, yx = eng.project(cams, big["joints"])
    ndc_s, yx_s = eng.project(cams, small["joints"])
    assert torch.equal(ndc[sel], ndc_s) and torch.equal(yx[sel], yx_s)
    w = torch.ones_like(yx)
    d_pts, d_fov_img = eng.project_backward(cams, big["joints"], d_yx=w)
    d_pts_s, _ = eng.project_backward(cams, small["joints"], d_yx=torch.ones_like(yx_s))
    assert torch.equal(d_pts[sel], d_pts_s) and d_fov_img.shape[0] == B


@pytest.mark.parametrize("key,S,radius", [("stick", 128, 2.7), ("mouse", 96, 4.0)])
def test_packed_gradient_atomics_match_float_atomics_and_are_reproducible(key, S, radius, tables):
    """Launches of >= 64 images accumulate the vertex gradient as packed 64-bit fixed point (one memory-side atomic per
    vertex) and decode it in place; smaller launches keep two float atomics.  Same images through both paths: equal to
    fixed-point resolution, and the packed path is order independent, hence bit-reproducible."""
    from smilify_amd import engine as eng
    from smilify_amd import synthetic

    t = tables(key)
    N = 96
    f = synthetic.make_problem(t, N, 1, S, DEV, radius=radius, seed=21, window=N)
    f._refresh_targets()
    dm = f.device_model
    lbs = eng.lbs_forward(dm, f.betas.detach(), f._pose, trans=f.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
    cam = f.renderer.cameras
    cams = eng.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f.fov.detach(), None, 1, S)
    ndc, _ = eng.project(cams, lbs["verts"], want_yx=False)
    scale = torch.full((N,), 3.0 / (S * S), device=DEV)
    scale[3] = 0.0          # an image without weight: no gradient either way
    scale[5] = 250.0        # a per-image scale 1e6 times the others: every image has its own fixed-point scale
    li_a, dn_a, _ = eng.silhouette_l1_fused(dm, ndc, S, f._sil_dev, f._sil_sum, scale)          # one launch of 96: packed
    li_b, dn_b, _ = eng.silhouette_l1_fused(dm, ndc, S, f._sil_dev, f._sil_sum, scale)
    assert torch.equal(dn_a, dn_b)                                                               # bit-reproducible gradient
    np.testing.assert_allclose(li_a.cpu().numpy(), li_b.cpu().numpy(), rtol=1e-6)                 # (the loss sums stay float atomics)
    dn_c = torch.empty_like(dn_a)
    li_c = torch.empty_like(li_a)
    for n0 in range(0, N, 32):                                                                   # three launches of 32: float atomics
        sl = slice(n0, n0 + 32)
        eng.silhouette_l1_fused(dm, ndc[sl].contiguous(), S, f._sil_dev[sl].contiguous(), f._sil_sum[sl].contiguous(),
                                scale[sl].contiguous(), loss_img=li_c[sl], d_ndc=dn_c[sl])
    np.testing.assert_allclose(li_a.cpu().numpy(), li_c.cpu().numpy(), rtol=1e-6)
    a, c = dn_a.cpu().numpy(), dn_c.cpu().numpy()
    assert np.abs(c).max() > 0
    assert np.abs(a[3]).max() == 0.0 and np.abs(c[3]).max() == 0.0
    per_img = np.abs(a - c).reshape(N, -1).max(1) / (np.abs(c).reshape(N, -1).max(1) + 1e-30)
    assert per_img.max() < 2e-5, per_img.max()
    # the fit iteration's hand-off: the rows stay packed and the projection backward decodes them while it reads - bit for bit
    # what the in-place decode pass followed by the plain projection backward gives (also for the weightless image and for a
    # call below the packing threshold, whose rows are plain floats with factor 0)
    _, dn_p, _, sc_p = eng.silhouette_l1_fused(dm, ndc, S, f._sil_dev, f._sil_sum, scale, packed_out=True)
    assert float(sc_p.max()) > 0.0 and not torch.equal(dn_p, dn_a)
    dv_a, fov_a = eng.project_backward(cams, lbs["verts"], d_ndc=dn_a)
    dv_p, fov_p = eng.project_backward(cams, lbs["verts"], d_ndc=dn_p, d_ndc_scale=sc_p)
    assert torch.equal(dv_a, dv_p)
    np.testing.assert_allclose(fov_a.cpu().numpy(), fov_p.cpu().numpy(), rtol=1e-5)  # (block sums meet in float atomics)
    _, dn_s, _, sc_s = eng.silhouette_l1_fused(dm, ndc[:32].contiguous(), S, f._sil_dev[:32].contiguous(), f._sil_sum[:32].contiguous(),
                                               scale[:32].contiguous(), packed_out=True)
    assert float(sc_s.abs().max()) == 0.0
    cams32 = eng.CameraSet(cam.R[:32].contiguous() if cam.R.shape[0] == N else cam.R.contiguous(),
                           cam.T[:32].contiguous() if cam.T.shape[0] == N else cam.T.contiguous(), f.fov.detach()[:32] if f.fov.numel() == N else f.fov.detach(), None, 1, S)
    dv_s, _ = eng.project_backward(cams32, lbs["verts"][:32].contiguous(), d_ndc=dn_s, d_ndc_scale=sc_s)
    dv_c, _ = eng.project_backward(cams32, lbs["verts"][:32].contiguous(), d_ndc=dn_c[:32].contiguous())
    np.testing.assert_allclose(dv_s.cpu().numpy(), dv_c.cpu().numpy(), rtol=1e-4, atol=1e-9)  # (two float-atomic runs: order noise)


def test_graph_replay_with_packed_gradients(tables):
    """64 frames: the fused launch takes the packed-gradient path, whose in-place decode kernel must be part of the captured
    iteration; three replayed iterations equal three eager ones."""
    from smilify_amd import synthetic

    t = tables("synthetic")

    def make():
        f = synthetic.make_problem(t, 64, 1, 48, DEV, radius=2.3, seed=13, window=8)
        f.begin_stage(synthetic.STAGE1_LR)
        return f

    fe, fg = make(), make()
    w, wt = synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL
    for _ in range(3):
        a = fe.fit_step(w, wt).clone()
        b = fg.fit_step_graph(w, wt).clone()
        np.testing.assert_allclose(b.cpu().numpy(), a.cpu().numpy(), rtol=2e-4, atol=1e-6)
    for n in ("joint_rotations", "trans", "betas"):
        np.testing.assert_allclose(getattr(fg, n).detach().cpu().numpy(), getattr(fe, n).detach().cpu().numpy(), rtol=1e-3, atol=1e-5)


def test_hardcoded_body_joints_of_35_joint_models():
    """reference smal_torch.py:353-365: a 35-joint model with ignore_hardcoded_body off returns six mesh vertices behind its
    joints (gradients flow to them like to any vertex)."""
    from smilify_amd import config as cfgmod
    from smilify_amd import model_io
    from smilify_amd.smal_torch import SMAL, _HARDCODED_BODY_VERTS

    t = model_io.synthetic_model(V_side=100, J=35, nB=2, seed=2)   # 3602 vertices > 3055
    assert t.J == 35 and t.V > max(_HARDCODED_BODY_VERTS)
    cfg = cfgmod.FitterConfig.from_tables(t, ignore_hardcoded_body=False)
    smal = SMAL(DEV, tables=t, config=cfg)
    beta = torch.zeros(2, t.nB, device=DEV)
    theta = (0.1 * torch.randn(2, t.J, 3, generator=torch.Generator().manual_seed(0))).to(DEV).requires_grad_()
    verts, joints, _, _ = smal(beta, theta)
    assert joints.shape == (2, 41, 3)
    assert torch.equal(joints[:, 35:], verts[:, list(_HARDCODED_BODY_VERTS)])
    assert smal(beta, theta, get_skin=False).shape == (2, 41, 3)
    joints[:, 35:].sum().backward()
    assert theta.grad is not None and float(theta.grad.abs().max()) > 0
    plain = SMAL(DEV, tables=t, config=cfgmod.FitterConfig.from_tables(t))
    assert plain(beta, theta.detach())[1].shape == (2, 35, 3)


def _mesh_through_the_clip_plane(t, N, S, seed, n_behind):
    """Random vertices in the rasteriser's input space with a few of them nearer than z_clip (some behind the camera)."""
    g = torch.Generator().manual_seed(seed)
    ndc = torch.empty(N, t.V, 3)
    ndc[..., :2] = 0.9 * (torch.rand(N, t.V, 2, generator=g) - 0.5)
    ndc[..., 2] = 0.8 + torch.rand(N, t.V, generator=g)
    for n in range(N):
        idx = torch.randperm(t.V, generator=g)[:n_behind]
        ndc[n, idx, 2] = torch.tensor([-0.4, 2e-4, -1.5, 1e-5, 4e-4, -0.05, 3e-4, -0.7][:n_behind])
        ndc[n, idx, :2] *= 0.3  # (keep the cut edges' crossings from flying off to 1e4 NDC units: fp32 of both sides stays comparable)
    return ndc


def test_faces_that_cross_z_clip_are_cut_at_the_plane(tables):
    """clip_faces (pytorch3d, left on by the reference's RasterizationSettings, p3d_renderer.py:36-47): a face with one or two
    vertices nearer than z_clip = znear / 2 is cut at the plane and its front part rendered.  HIP (per-image clip tables
    filled by the setup kernel) against the oracle's restatement (render_ref.clip_faces_np): silhouette and vertex gradient,
    the latter with the new vertices' gradients handed back to the cut edges' end points."""
    from smilify_amd import engine as eng

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S, N = 48, 3
    ndc = _mesh_through_the_clip_plane(t, N, S, 5, 6)
    faces = t.faces
    # known answer first (oracle only): the crossing points are where the VIEW-space segments meet the plane
    va, fa, src, coef = render_ref.clip_faces_np(ndc[0].numpy(), faces, 5e-4)
    assert src.shape[0] >= 4 and fa.shape[0] > faces.shape[0]
    v64 = ndc[0].numpy().astype(np.float64)
    for j in range(src.shape[0]):
        a, b = src[j]
        pa, pb = np.array([v64[a, 0] * v64[a, 2], v64[a, 1] * v64[a, 2], v64[a, 2]]), np.array([v64[b, 0] * v64[b, 2], v64[b, 1] * v64[b, 2], v64[b, 2]])
        q = pa + (5e-4 - pa[2]) / (pb[2] - pa[2]) * (pb - pa)
        np.testing.assert_allclose(va[t.V + j, :2], q[:2] / 5e-4, rtol=2e-5, atol=1e-4)
    sil = eng.silhouette_forward(dm, ndc.to(DEV), S).cpu().numpy()
    st = eng.raster_stats(dm, N)
    assert st["straddling_faces"] > 0 and st["unclipped_faces"] == 0
    with render_ref.select_mode(1):
        ref, _ = render_ref.silhouette_forward_np(ndc.numpy(), faces, S)
    d = np.abs(sil - ref)
    assert ref.sum() > 50 and d.mean() < 2e-5 and np.mean(d > 1e-3) < 2e-3, (ref.sum(), d.mean(), np.mean(d > 1e-3), d.max())
    # the cut matters: rendering the same mesh with the cut faces simply left out differs visibly
    keep = ~((ndc[0, :, 2][torch.from_numpy(faces.astype(np.int64))] < 5e-4).any(1)).numpy()
    assert np.abs(render_ref.silhouette_forward_np(ndc[:1].numpy(), faces[keep], S)[0] - ref[:1]).sum() > 1.0
    # gradient
    gs = torch.from_numpy(np.cos(0.3 * np.arange(N * S * S)).astype(np.float32).reshape(N, S, S))
    got = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV)).cpu().numpy()
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc.numpy(), faces, S, gs.numpy())[..., :2]
    scale = np.abs(want).max()
    # (the front parts carry vertices at |xy| ~ 1e2 NDC units: fp32 cancellation in both implementations)
    assert scale > 0 and np.abs(got - want).max() < 6e-3 * scale, (np.abs(got - want).max(), scale)
    # the fused path on a batch large enough for packed gradients: images with cut faces fall back to float rows, the others stay packed
    Nb = 70
    big = _mesh_through_the_clip_plane(t, Nb, S, 9, 0)
    big[5] = ndc[0]; big[33] = ndc[1]
    target = (torch.rand(Nb, S, S, generator=torch.Generator().manual_seed(2)) > 0.5).float()
    scale_img = torch.full((Nb,), 1.0 / (S * S), device=DEV)
    li, dn, _ = eng.silhouette_l1_fused(dm, big.to(DEV), S, target.to(DEV), eng.image_abs_sum(target.to(DEV)), scale_img)
    _, _, _, row_scale = eng.silhouette_l1_fused(dm, big.to(DEV), S, target.to(DEV), eng.image_abs_sum(target.to(DEV)), scale_img, packed_out=True)
    assert float(row_scale[5]) == 0.0 and float(row_scale[33]) == 0.0 and float(row_scale[6]) > 0.0  # float rows / packed rows
    for n in (5, 6, 33):
        li1, dn1, _ = eng.silhouette_l1_fused(dm, big[n:n + 1].to(DEV).contiguous(), S, target[n:n + 1].to(DEV).contiguous(),
                                              eng.image_abs_sum(target[n:n + 1].to(DEV).contiguous()), scale_img[:1].contiguous())
        np.testing.assert_allclose(li[n].item(), li1[0].item(), rtol=1e-5)
        a, b = dn[n].cpu().numpy(), dn1[0].cpu().numpy()
        assert np.abs(a - b).max() <= 2e-5 * np.abs(b).max() + 1e-12, (n, np.abs(a - b).max(), np.abs(b).max())


def test_packed_gradients_keep_their_resolution_next_to_an_image_filling_face(tables):
    """The fixed-point scale of a packed launch comes from a worst-case bound per image (valence x largest face box): one face
    that fills the image coarsens the resolution of that image's whole gradient.  With such a face in the mesh the packed
    gradient (>= 64 images) must still agree with the float-atomic one (same images in small launches) well inside the
    gradient tolerance used against the oracle (1e-3 of the largest component): every record is rounded once to 2^-30 of the
    BOUND, so with a bound ~1e3 times the actual largest component and ~1e3 records per vertex the noise is a few 1e-4 here
    (a few 1e-6 on ordinary meshes, previous test) - and with the oracle."""
    from smilify_amd import engine as eng
    from smilify_amd import model_io, synthetic

    base = tables("synthetic")
    S, N = 40, 64
    f0 = synthetic.make_problem(base, N, 1, S, DEV, radius=2.3, seed=12, window=N)
    f0._refresh_targets()
    lbs = eng.lbs_forward(f0.device_model, f0.betas.detach(), f0._pose, trans=f0.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
    cam = f0.renderer.cameras
    cams = eng.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f0.fov.detach(), None, 1, S)
    ndc, _ = eng.project(cams, lbs["verts"], want_yx=False)
    # a second topology: the same vertices plus one far, image-filling triangle over three of them
    faces = np.concatenate([base.faces, np.array([[0, 1, 2]], base.faces.dtype)])
    ndc = ndc.clone()
    ndc[:, 0] = torch.tensor([-0.95, -0.9, 9.0], device=DEV)
    ndc[:, 1] = torch.tensor([0.95, -0.9, 9.0], device=DEV)
    ndc[:, 2] = torch.tensor([0.0, 0.95, 9.0], device=DEV)
    from smilify_amd.p3d_renderer import _MeshTopology

    dm = _MeshTopology(np.ascontiguousarray(faces.astype(np.int32)), base.V, torch.device(DEV)).dm
    target = (torch.rand(N, S, S, generator=torch.Generator().manual_seed(3)) > 0.5).float().to(DEV)
    tsum = eng.image_abs_sum(target)
    scale = torch.full((N,), 1.0 / (S * S), device=DEV)
    _, dn_packed, _ = eng.silhouette_l1_fused(dm, ndc, S, target, tsum, scale)                       # 64 images: packed
    dn_float = torch.empty_like(dn_packed)
    for n0 in range(0, N, 16):
        sl = slice(n0, n0 + 16)
        eng.silhouette_l1_fused(dm, ndc[sl].contiguous(), S, target[sl].contiguous(), tsum[sl].contiguous(), scale[sl].contiguous(),
                                d_ndc=dn_float[sl], loss_img=torch.empty(16, device=DEV))
    a, b = dn_packed.cpu().numpy(), dn_float.cpu().numpy()
    per_img = np.abs(a - b).reshape(N, -1).max(1) / np.abs(b).reshape(N, -1).max(1)
    assert np.abs(b).max() > 0 and per_img.max() < 5e-4, per_img.max()
    # and against the oracle on two of the images (sign(sil - target) / S^2 is the upstream gradient of the L1 term)
    sil = eng.silhouette_forward(dm, ndc[:2].contiguous(), S).cpu()
    gs = (torch.sign(sil - target[:2].cpu()) / (S * S)).numpy()
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc[:2].cpu().numpy(), faces, S, gs)[..., :2]
    sc = np.abs(want).max()
    assert np.abs(a[:2] - want).max() < 2e-3 * sc, (np.abs(a[:2] - want).max(), sc)
