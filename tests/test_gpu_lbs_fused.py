"""GPU tests of the fused per-frame LBS kernels (``pytest -m gpu``): ``smil_lbs_forward_project`` (skinning + joint regression +
both projections) and ``smil_lbs_backward_ndc`` (projection backward + skinning backward + shape backward from the image plane)
against the separate-kernel route and, directly, against the CPU oracle's autograd (oracle/lbs_ref.py, oracle/render_ref.py;
reference smal_model/smal_torch.py:240-351, batch_lbs.py:155-195); the bit-reproducible shared shape gradient."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, vertex_probe
from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

def _close(a, b, rtol, what):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    scale = np.abs(b).max() + 1e-30
    err = np.abs(a - b).max() / scale
    assert err < rtol, (what, err)


class _Fixed:
    """A camera set that claims ``n`` images whatever the caller computes (to reach the library's own check)."""

    def __init__(self, cams, n):
        self._c, self._n, self.views = cams, n, cams.views

    def struct(self, _n):
        return self._c.struct(self._n)


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

@pytest.mark.parametrize("seed,wide", [(s, False) for s in (0, 1, 2, 3, 5, 8, 13, 21)] + [(s, True) for s in (100, 101, 102, 103)])
def test_fused_lbs_kernels_against_the_oracle(seed, wide):
    """smil_lbs_forward_project / smil_lbs_backward_ndc directly against the CPU oracle's autograd through LBS and projection
    (oracle/lbs_ref.py, oracle/render_ref.py; reference smal_model/smal_torch.py:240-351, batch_lbs.py:155-195): seeded random
    models with up to 120 (wide: 250) joints, up to 20 views, static and regressed joints, shared and per-frame betas.
    Tolerances (tests/lbs_cases.py): forward 2e-5, parameter gradients 5e-4 of the largest component; fused against separate
    kernels 1e-6 / 3e-5."""
    import lbs_cases

    checks, info = lbs_cases.run_case(seed, wide)
    assert any(w.startswith("oracle d_") for _, w, _ in checks), info
    fails = [(w, e) for f, w, e in checks if f is not None]
    assert not fails, (info, fails)

@pytest.mark.parametrize("seed,key", [(200, None), (201, None), (202, None), (203, "mouse"), (204, "mouse")])
def test_fused_lbs_kernels_on_meshes_beyond_half_a_cu(seed, key, tables):
    """Meshes whose per-frame vertex state (24 bytes per vertex) does not fit twice into a CU's LDS take the fused kernels' second
    form since round 4: one workgroup of 1024 threads per CU, only the vertex gradient in LDS (12 bytes per vertex), the rest
    vertices gathered from memory one bone-list segment ahead; the forward kernel keeps no vertex copy at all for models with
    static joints.  Random tubes with 3 600 - 5 000 vertices and the mouse (V = 11 263, BASELINE configs 3 and 5) against the
    CPU oracle's autograd and against the separate-kernel route (reference smal_model/smal_torch.py:320-351)."""
    import lbs_cases
    from smilify_amd import engine as eng

    t = tables(key) if key else None
    checks, info = lbs_cases.run_case(seed, big=key is None, table=t)
    assert info["V"] > 3500 and info["fused_bwd"] == 1, info
    assert any(w.startswith("oracle d_") for _, w, _ in checks) and any(w.startswith("bwd d_") for _, w, _ in checks), info
    fails = [(w, e) for f, w, e in checks if f is not None]
    assert not fails, (info, fails)
    if key:  # the form is chosen by the model alone: every BASELINE camera rig of the mouse is covered
        dm = eng.DeviceModel(t, DEV)
        assert all(eng.lbs_backward_ndc_supported(dm, dm.nB, v) for v in (1, 2, 18, 32))

@pytest.mark.parametrize("key,frames,views", [("stick", 96, 2), ("mouse", 24, 3)])
def test_the_shared_shape_gradient_is_bit_reproducible(key, frames, views, tables):
    """Two evaluations of the same fit step return the same bits in d_betas - the one quantity ranks all-reduce (reference
    fitter.py:236-335: ``betas`` is shared by every frame, so its gradient is a sum over frames).  Round 3 summed the frames with
    float atomics, whose result depends on the order the blocks arrive in; since round 4 every block leaves a partial row and the
    last block adds the rows in a fixed order (lbs.hip BetaSum).  STICK goes through the fused per-frame kernel (two workgroups per
    CU) + chain kernel, the mouse through its one-workgroup-per-CU form; several runs, because a race would only show now and then.  (At least 64 images per
    launch, so that the rasteriser's vertex gradients upstream are the integer-exact packed ones.)"""
    from smilify_amd import synthetic

    f = synthetic.make_problem(tables(key), frames, views, 64, DEV, radius=2.7 if key == "stick" else 4.0)
    ref_objs, ref = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL, window=10)
    ref = {k: v.clone() for k, v in ref.items() if v is not None}
    ref_objs = ref_objs.clone()
    assert float(ref["betas"].abs().max()) > 0
    for _ in range(5):
        objs, g = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL, window=10)
        assert torch.equal(g["betas"], ref["betas"]), (g["betas"], ref["betas"])
        # the shared scale / translation tables are sums in a fixed order as well (smil_reduce_rows)
        for k in ("log_beta_scales", "betas_trans"):
            if k in ref and g.get(k) is not None:
                assert torch.equal(g[k], ref[k]), k
        # (the fov gradient and the loss terms still end in a few float atomics per image / per block: equal to rounding)
        np.testing.assert_allclose(g["fov"].cpu().numpy(), ref["fov"].cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(objs.cpu().numpy(), ref_objs.cpu().numpy(), rtol=1e-5, atol=1e-7)
