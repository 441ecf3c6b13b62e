"""GPU tests of round 3 (``pytest -m gpu``): the skinning backward that takes its upstream gradients on the image plane
(``smil_lbs_backward_ndc``: projection backward + skinning backward + shape backward in one kernel per frame) against the
two-call route it replaces, which the other test files pin to the oracle and to vectors of the real reference."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _close(a, b, rtol, what):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max() / scale
    assert err < rtol, (what, err)


@pytest.mark.parametrize("key,views,shared_beta,trans_after", [
    ("stick", 1, True, True),            # the fit iteration's own call
    ("stick", 3, True, True),
    ("stick", 2, False, False),          # SMAL.__call__ semantics: joints regressed from the translated vertices, per-frame betas
    ("synthetic", 4, True, True),
    ("synthetic_static", 2, True, True),  # static joints: the joint gradient enters the chain, not the regressor
    ("synthetic_static", 1, False, False),
])
def test_backward_from_the_image_plane_equals_projection_backward_then_skinning_backward(key, views, shared_beta, trans_after, tables):
    from smilify_amd import cameras as cam_mod
    from smilify_amd import engine as eng

    t = tables(key)
    dm = eng.DeviceModel(t, DEV)
    B, S = 37, 64
    J, V, nB = dm.J, dm.V, dm.nB
    g = torch.Generator().manual_seed(5)
    beta = (0.4 * torch.randn(nB, generator=g) if shared_beta else 0.4 * torch.randn(B, nB, generator=g)).to(DEV)
    theta = (0.25 * torch.randn(B, J, 3, generator=g)).to(DEV)
    trans = (0.1 * torch.randn(B, 3, generator=g)).to(DEV)
    ls = (0.05 * torch.randn(J, 3, generator=g)).to(DEV)
    bt = (0.02 * torch.randn(J, 3, generator=g)).to(DEV)
    lbs = eng.lbs_forward(dm, beta, theta, trans=trans, logscale=ls, btrans=bt, shared_beta=shared_beta, logscale_shared=True,
                          btrans_shared=True, trans_after_joints=trans_after)
    R, T = cam_mod.look_at_view_transform(3.0, 10.0, np.linspace(0, 300, views), device=DEV)
    fov = torch.full((views,), 55.0, device=DEV)
    cams = eng.CameraSet(R.contiguous(), T.contiguous(), fov, None, views, S)
    N = B * views
    d_ndc = (1e-3 * torch.randn(N, V, 2, generator=g)).to(DEV)
    d_yx = (1e-2 * torch.randn(N, J, 2, generator=g)).to(DEV)
    assert eng.lbs_backward_ndc_supported(dm, nB, views)

    def route(fused, with_ndc=True, with_yx=True):
        dn = d_ndc if with_ndc else None
        dy = d_yx if with_yx else None
        fov_img = torch.zeros(N, device=DEV)
        if fused:
            out = eng.lbs_backward(dm, lbs, None, None, ndc_upstream=dict(cams=cams, d_ndc=dn, d_yx=dy, d_fov_img=fov_img))
        else:
            if with_ndc and with_yx:
                dv, dj = eng.project_backward_verts_and_joints(cams, lbs["verts"], dn, lbs["joints"], dy, fov_img)
            elif with_ndc:
                (dv, _), dj = eng.project_backward(cams, lbs["verts"], d_ndc=dn, d_fov_img=fov_img), None
            else:
                dv, (dj, _) = None, eng.project_backward(cams, lbs["joints"], d_yx=dy, d_fov_img=fov_img)
            out = eng.lbs_backward(dm, lbs, dv, dj)
            out["d_joints"] = dj
        out["fov_img"] = fov_img
        return out

    for with_ndc, with_yx in ((True, True), (True, False), (False, True)):
        a, b = route(True, with_ndc, with_yx), route(False, with_ndc, with_yx)
        for k in ("d_beta", "d_theta", "d_trans", "d_logscale", "d_btrans", "fov_img"):
            assert a[k] is not None and b[k] is not None, k
            _close(a[k], b[k], 2e-5, (k, with_ndc, with_yx))
        if with_yx:
            _close(a["d_joints"], b["d_joints"], 1e-6, "d_joints")


def test_backward_from_the_image_plane_decodes_packed_rows_and_declines_what_it_cannot_hold(tables):
    """The vertex gradient arrives as the fused rasteriser leaves it (packed 64-bit fixed point with per-image factors); a
    mesh whose per-frame gradient does not fit the workgroup's LDS is declined, and the fit iteration then takes the two-call route."""
    from smilify_amd import engine as eng
    from smilify_amd import synthetic

    t = tables("stick")
    N, S = 80, 96
    f = synthetic.make_problem(t, N, 1, S, DEV, seed=4, window=N)
    f._refresh_targets()
    dm = f.device_model
    lbs = eng.lbs_forward(dm, f.betas.detach(), f._pose, trans=f.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
    cam = f.renderer.cameras
    cams = eng.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f.fov.detach(), None, 1, S)
    ndc, _ = eng.project(cams, lbs["verts"], want_yx=False)
    scale = torch.full((N,), 3.0 / (S * S), device=DEV)
    scale[7] = 0.0
    _, dn_p, _, sc_p = eng.silhouette_l1_fused(dm, ndc, S, f._sil_dev, f._sil_sum, scale, packed_out=True)
    assert float(sc_p.max()) > 0.0
    fov_a, fov_b = torch.zeros(N, device=DEV), torch.zeros(N, device=DEV)
    a = eng.lbs_backward(dm, lbs, None, None, ndc_upstream=dict(cams=cams, d_ndc=dn_p, d_ndc_scale=sc_p, d_fov_img=fov_a))
    dv, _ = eng.project_backward(cams, lbs["verts"], d_ndc=dn_p, d_fov_img=fov_b, d_ndc_scale=sc_p)
    b = eng.lbs_backward(dm, lbs, dv, None)
    for k in ("d_beta", "d_theta", "d_trans"):
        _close(a[k], b[k], 2e-5, k)
    _close(fov_a, fov_b, 2e-5, "fov")
    mouse = eng.DeviceModel(tables("mouse"), DEV)
    assert eng.lbs_backward_ndc_supported(mouse, mouse.nB, 18)  # round 4: 11 263 vertices x 12 bytes in one workgroup per CU


@pytest.mark.parametrize("key,views,trans_after", [("stick", 1, True), ("stick", 3, False), ("synthetic", 5, True), ("synthetic_static", 2, True),
                                                   ("synthetic_static", 2, False), ("mouse", 2, True)])
def test_forward_with_projection_equals_forward_then_projection(key, views, trans_after, tables):
    """``smil_lbs_forward_project``: skinning + joint regression + both projections in one kernel per frame (round 4: the mouse, whose joints are
    static, keeps no vertex copy in LDS and takes the same kernel)."""
    from smilify_amd import cameras as cam_mod
    from smilify_amd import engine as eng

    t = tables(key)
    dm = eng.DeviceModel(t, DEV)
    B, S, J, nB = 21, 96, dm.J, dm.nB
    g = torch.Generator().manual_seed(8)
    beta = (0.4 * torch.randn(nB, generator=g)).to(DEV)
    theta = (0.25 * torch.randn(B, J, 3, generator=g)).to(DEV)
    trans = (0.1 * torch.randn(B, 3, generator=g)).to(DEV)
    R, T = cam_mod.look_at_view_transform(3.0, 10.0, np.linspace(0, 300, views), device=DEV)
    cams = eng.CameraSet(R.contiguous(), T.contiguous(), torch.full((views,), 50.0, device=DEV), None, views, S)
    kw = dict(trans=trans, shared_beta=True, trans_after_joints=trans_after)
    ref = eng.lbs_forward(dm, beta, theta, **kw)
    ndc_ref, yx_ref = eng.project_verts_and_joints(cams, ref["verts"], ref["joints"])
    for want in (dict(ndc=True, yx=True), dict(ndc=True, yx=False), dict(ndc=False, yx=True)):
        got = eng.lbs_forward(dm, beta, theta, project=dict(cams=cams, **want), **kw)
        for k in ("verts", "joints", "A", "new_J"):
            _close(got[k], ref[k], 1e-6, k)
        assert ("ndc" in got) == want["ndc"] and ("yx" in got) == want["yx"]
        if want["ndc"]:
            _close(got["ndc"], ndc_ref, 1e-6, "ndc")
        if want["yx"]:
            _close(got["yx"], yx_ref, 1e-6, "yx")


@pytest.mark.parametrize("views", [1, 3])
def test_fit_iteration_is_the_same_through_either_route(views, tables):
    from smilify_amd import engine as eng
    from smilify_amd import synthetic

    t = tables("stick")
    outs = []
    for fused in (True, False):
        eng.FUSED_LBS_BACKWARD = eng.FUSED_LBS_FORWARD = fused
        try:
            f = synthetic.make_problem(t, 24, views, 64, DEV, seed=9, window=8)
            objs, grads = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL, window=8)
        finally:
            eng.FUSED_LBS_BACKWARD = eng.FUSED_LBS_FORWARD = True
        outs.append((objs, grads))
    (oa, ga), (ob, gb) = outs
    _close(oa, ob, 1e-6, "objs")
    assert set(ga) == set(gb)
    for k in ga:
        if ga[k] is None:
            assert gb[k] is None
            continue
        _close(ga[k], gb[k], 3e-5, k)


@pytest.mark.parametrize("key,matrices", [("stick", False), ("synthetic", True), ("synthetic_static", False)])
def test_gradients_flow_through_all_four_tensors_smal_returns(key, matrices, tables):
    """SMAL.__call__ hands (verts, joints, Rs, v_shaped) to its caller (reference smal_torch.py:367-370) and torch would
    differentiate through every one of them: a loss on Rs and v_shaped alone, and one on all four, against the oracle's autograd."""
    from conftest import oracle_model, vertex_probe
    from oracle import lbs_ref
    from smilify_amd.smal_torch import SMAL

    t = tables(key)
    smal = SMAL(DEV, tables=t)
    m = oracle_model(t)
    B, J, nB, V = 5, t.J, t.nB, t.V
    g = torch.Generator().manual_seed(11)
    host = dict(beta=0.4 * torch.randn(B, nB, generator=g), theta=0.3 * torch.randn(B, J, 3, generator=g),
                trans=0.1 * torch.randn(B, 3, generator=g), del_v=0.01 * torch.randn(B, V, 3, generator=g))
    if matrices:
        host["theta"] = lbs_ref.rodrigues(host["theta"].reshape(-1, 3)).view(B, J, 3, 3)
    pR, pS = vertex_probe((B, J, 3, 3), 2), vertex_probe((B, V, 3), 3)
    pV, pJ = vertex_probe((B, V, 3), 0), vertex_probe((B, J, 3), 1)
    for which in ("rs_vs", "all"):
        ref_leaves = {k: v.clone().requires_grad_() for k, v in host.items()}
        o = lbs_ref.smal_forward(m, ref_leaves["beta"], ref_leaves["theta"], trans=ref_leaves["trans"], del_v=ref_leaves["del_v"])
        loss = (o["Rs"] * pR).sum() + (o["v_shaped"] * pS).sum()
        if which == "all":
            loss = loss + (o["verts"] * pV).sum() + (o["joints"] * pJ).sum()
        loss.backward()
        leaves = {k: v.clone().to(DEV).requires_grad_() for k, v in host.items()}
        verts, joints, Rs, v_shaped = smal(leaves["beta"], leaves["theta"], trans=leaves["trans"], del_v=leaves["del_v"])
        np.testing.assert_allclose(Rs.detach().cpu().numpy(), o["Rs"].detach().numpy(), atol=2e-6)
        np.testing.assert_allclose(v_shaped.detach().cpu().numpy(), o["v_shaped"].detach().numpy(), atol=2e-6)
        loss = (Rs * pR.to(DEV)).sum() + (v_shaped * pS.to(DEV)).sum()
        if which == "all":
            loss = loss + (verts * pV.to(DEV)).sum() + (joints * pJ.to(DEV)).sum()
        loss.backward()
        for k in host:
            ref = ref_leaves[k].grad
            got = leaves[k].grad
            if which == "rs_vs" and k == "trans":
                assert got is None or float(got.abs().max()) == 0.0  # neither tensor depends on the translation
                continue
            assert got is not None, (which, k)
            _close(got, ref, 3e-4, (which, k))


def test_fused_entries_refuse_what_they_cannot_do(tables):
    """Loud failures instead of wrong numbers: the image-plane backward on a mesh beyond its LDS, upstream gradients in both
    forms at once, and a camera table that does not match the batch."""
    from smilify_amd import cameras as cam_mod
    from smilify_amd import engine as eng
    from smilify_amd._lib import SmilError

    from smilify_amd import model_io

    t = model_io.synthetic_model(V_side=125, J=120, nB=3, seed=1)  # 15 127 vertices: 12 bytes each exceed a CU's 160 KB of LDS
    dm = eng.DeviceModel(t, DEV)
    assert not eng.lbs_backward_ndc_supported(dm, dm.nB, 1)
    B, S = 3, 32
    g = torch.Generator().manual_seed(2)
    beta = torch.zeros(dm.nB, device=DEV)
    theta = (0.1 * torch.randn(B, dm.J, 3, generator=g)).to(DEV)
    R, T = cam_mod.look_at_view_transform(4.0, 10.0, np.array([0.0]), device=DEV)
    cams = eng.CameraSet(R.contiguous(), T.contiguous(), torch.full((1,), 50.0, device=DEV), None, 1, S)
    lbs = eng.lbs_forward(dm, beta, theta, shared_beta=True)
    d_ndc = torch.zeros(B, dm.V, 2, device=DEV)
    with pytest.raises(SmilError, match="not available for this model"):
        eng.lbs_backward(dm, lbs, None, None, ndc_upstream=dict(cams=cams, d_ndc=d_ndc))
    small = eng.DeviceModel(tables("synthetic"), DEV)
    th = (0.1 * torch.randn(B, small.J, 3, generator=g)).to(DEV)
    lbs_s = eng.lbs_forward(small, torch.zeros(small.nB, device=DEV), th, shared_beta=True)
    with pytest.raises(ValueError, match="ndc_upstream replaces"):
        eng.lbs_backward(small, lbs_s, torch.zeros(B, small.V, 3, device=DEV), None, ndc_upstream=dict(cams=cams, d_ndc=torch.zeros(B, small.V, 2, device=DEV)))
    two_views = eng.CameraSet(R.repeat(2, 1, 1).contiguous(), T.repeat(2, 1).contiguous(), torch.full((2,), 50.0, device=DEV), None, 2, S)
    with pytest.raises(SmilError, match="images for"):
        eng.lbs_backward(small, lbs_s, None, None, ndc_upstream=dict(cams=_Fixed(two_views, B), d_ndc=torch.zeros(B, small.V, 2, device=DEV)))


class _Fixed:
    """A camera set that claims ``n`` images whatever the caller computes (to reach the library's own check)."""

    def __init__(self, cams, n):
        self._c, self._n, self.views = cams, n, cams.views

    def struct(self, _n):
        return self._c.struct(self._n)
