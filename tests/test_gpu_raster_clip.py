"""GPU tests of the rasteriser at ``z_clip`` (``pytest -m gpu``; pytorch3d ``clip_faces`` as ``MeshRasterizer`` applies it with the
reference's settings, smal_fitter/p3d_renderer.py:36-47): faces nearer than the plane are culled, faces that cross it are cut and
their front parts rendered, forward and gradient against the oracle and against finite differences."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, vertex_probe
from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

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

def test_gradient_of_a_cut_face_by_finite_differences(tables):
    """A face that crosses z_clip is rendered as its front part, whose new vertices are ``c_a xy_a + c_b xy_b`` of the cut edge's end
    points with coefficients that depend on the depths only (pytorch3d clip_faces; reference settings p3d_renderer.py:36-47).  Moving
    an end point in x or y therefore moves the new vertex linearly, and the analytic vertex gradient - new vertices handing theirs
    back through k_clip_backward - must agree with central finite differences of the rendered silhouette (round-3 advice: the clip
    fuzz compared forward passes only).  Scene: the posed synthetic mesh at a distance (a silhouette with a soft rim, not a filled
    image), three of its vertices pulled through the clipping plane."""
    from smilify_amd import engine as eng
    from oracle import render_ref
    from test_gpu_edge_cases import _scene

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 48
    ndc = _scene(t, 1, S, 2.5, 3).clone()
    pulled = [5, 40, 77]
    # nearer than z_clip = 5e-4 but in front of the camera: the coefficients stay O(1) (behind the camera they reach hundreds and a
    # finite step on an end point moves the new vertex by many blur radii)
    ndc[0, pulled, 2] = torch.tensor([2e-4, 1e-5, 4e-4])
    _, _, src, _ = render_ref.clip_faces_np(ndc[0].numpy(), t.faces, 5e-4)
    ends = sorted({int(v) for ab in src for v in ab if ndc[0, int(v), 2] >= 5e-4})  # the cut edges' end points in front of the plane
    assert len(src) >= 4 and ends
    gs = torch.from_numpy(np.cos(0.3 * np.arange(S * S)).astype(np.float32).reshape(1, S, S)).to(DEV)
    grad = eng.silhouette_backward(dm, ndc.to(DEV), S, gs).cpu().numpy()[0]
    assert eng.raster_stats(dm, 1)["straddling_faces"] > 0
    loss = lambda x: float((eng.silhouette_forward(dm, x.to(DEV), S).double() * gs.double()).sum())  # noqa: E731
    eps, checked = 2e-4, 0
    for v in ends:
        for c in (0, 1):
            if abs(grad[v, c]) < 0.05 * np.abs(grad).max():
                continue  # (too flat for a difference of two fp32 renders to say anything)
            hi, lo = ndc.clone(), ndc.clone()
            hi[0, v, c] += eps
            lo[0, v, c] -= eps
            fd = (loss(hi) - loss(lo)) / (2 * eps)
            assert abs(fd - grad[v, c]) <= 0.06 * abs(grad[v, c]) + 0.02 * np.abs(grad).max(), (v, c, fd, grad[v, c])
            checked += 1
    assert checked >= 3, (checked, np.abs(grad).max())

def test_depth_gradient_of_cut_edges(tables):
    """The end points of an edge that crosses z_clip receive a DEPTH gradient (pytorch3d differentiates clip_faces through the
    interpolation weight; reference settings p3d_renderer.py:36-47).  It travels beside d_ndc as a sparse list (``ClipDepth``): HIP
    against the oracle's analytic depth gradient and against central finite differences of the rendered silhouette, and the list
    carried through the cameras by both LBS backward routes (fused kernel / projection backward + ``clip_depth_backward``)."""
    from smilify_amd import engine as eng
    from oracle import render_ref
    from test_gpu_edge_cases import _scene

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S, N = 48, 2
    ndc = _scene(t, N, S, 2.5, 3).clone()
    ndc[0, [5, 40, 77], 2] = torch.tensor([2e-4, 1e-5, 4e-4])      # image 0: three vertices pulled through the plane; image 1: none
    gs = torch.from_numpy(np.cos(0.3 * np.arange(N * S * S)).astype(np.float32).reshape(N, S, S))
    want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy())
    assert np.abs(want[0, :, 2]).max() > 0 and np.abs(want[1, :, 2]).max() == 0
    cd = eng.ClipDepth(DEV, N)
    d_ndc = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV), clip_depth=cd).cpu().numpy()
    dz = cd.dense(t.V).numpy()
    assert int(cd.counter[1]) == 0 and int(cd.range[1, 1]) == 0 and int(cd.range[0, 1]) > 0
    sc = np.abs(want[0, :, 2]).max()
    assert np.abs(dz[0] - want[0, :, 2]).max() <= 2e-3 * sc, (np.abs(dz[0] - want[0, :, 2]).max(), sc)
    assert np.abs(dz[1]).max() == 0.0
    scxy = np.abs(want[..., :2]).max()
    assert np.abs(d_ndc - want[..., :2]).max() <= 2e-3 * scxy
    # finite differences in the depth of the end points in front of the plane
    loss = lambda x: float((eng.silhouette_forward(dm, x.to(DEV), S).double().cpu() * gs.double()).sum())  # noqa: E731
    checked = 0
    for v in np.argsort(-np.abs(dz[0]))[:6]:
        if float(ndc[0, v, 2]) < 5e-4:
            continue
        eps = 2e-3 * float(ndc[0, v, 2])
        hi, lo = ndc.clone(), ndc.clone()
        hi[0, v, 2] += eps
        lo[0, v, 2] -= eps
        fd = (loss(hi) - loss(lo)) / (float(hi[0, v, 2]) - float(lo[0, v, 2]))
        assert abs(fd - dz[0, v]) <= 0.1 * abs(dz[0, v]) + 0.02 * np.abs(dz[0]).max(), (v, fd, dz[0, v])
        checked += 1
    assert checked >= 2
    # the same list through the fused fit path and through the separate kernels: fused l1 entry point, both LBS backward routes
    f = None
    from smilify_amd import synthetic
    f = synthetic.make_problem(t, N, 1, S, DEV, radius=2.5, seed=3, window=N)
    f._refresh_targets()
    lbs = eng.lbs_forward(f.device_model, f.betas.detach(), f._pose, trans=f.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
    cam = f.renderer.cameras
    cams = eng.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f.fov.detach(), None, 1, S)
    d_fake = torch.zeros(N, t.V, 2, device=DEV)
    fake = eng.ClipDepth(DEV, N)                                      # two entries for image 1, none for image 0
    fake.vertex[:2] = torch.tensor([7, 33], dtype=torch.int32)
    fake.dz[:2] = torch.tensor([0.75, -1.5])
    fake.range[1] = torch.tensor([0, 2], dtype=torch.int32)
    fake.counter[0] = 2
    d_v, _ = eng.project_backward(cams, lbs["verts"], d_ndc=d_fake)
    eng.clip_depth_backward(cams, fake, d_v)
    R = cam.R[0].to(DEV) if cam.R.shape[0] == 1 else cam.R[1].to(DEV)
    expect = torch.zeros(N, t.V, 3, device=DEV)
    expect[1, 7] = 0.75 * R[:, 2]
    expect[1, 33] = -1.5 * R[:, 2]
    assert torch.allclose(d_v, expect, atol=1e-6), (d_v - expect).abs().max()
    if eng.lbs_backward_ndc_supported(f.device_model, t.nB, 1):
        g_sep = eng.lbs_backward(f.device_model, lbs, d_v, None, need_beta=True)
        g_fus = eng.lbs_backward(f.device_model, lbs, None, None, need_beta=True,
                                 ndc_upstream=dict(cams=cams, d_ndc=d_fake, d_ndc_scale=None, d_yx=None, d_fov_img=None, clip_depth=fake))
        for k in ("d_theta", "d_trans", "d_beta"):
            a_, b_ = g_sep[k].cpu().numpy(), g_fus[k].cpu().numpy()
            assert np.abs(a_ - b_).max() <= 1e-5 * (np.abs(a_).max() + 1e-12), (k, np.abs(a_ - b_).max(), np.abs(a_).max())
