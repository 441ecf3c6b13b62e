"""GPU tests of the rasteriser's gradient accumulation (``pytest -m gpu``): packed fixed-point atomics against float atomics,
reproducibility, resolution next to an image-filling face, hipGraph replay, a saturated pixel among empty ones
(reference behaviour at stake: smal_fitter/p3d_renderer.py:41-52,142-146 + fitter.py:332-333)."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, vertex_probe
from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

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
    assert torch.equal(li_a, li_b)                                                               # ... and loss (round 5: the tiles' terms are summed as 2^-32 fixed point)
    dn_c = torch.empty_like(dn_a)
    li_c = torch.empty_like(li_a)
    for n0 in range(0, N, 32):                                                                   # three launches of 32: float atomics
        sl = slice(n0, n0 + 32)
        eng.silhouette_l1_fused(dm, ndc[sl].contiguous(), S, f._sil_dev[sl].contiguous(), f._sil_sum[sl].contiguous(),
                                scale[sl].contiguous(), loss_img=li_c[sl], d_ndc=dn_c[sl])
    np.testing.assert_allclose(li_a.cpu().numpy(), li_c.cpu().numpy(), rtol=1e-6)                 # (a launch of 32 deals its tiles in pieces: other partial sums)
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

def test_a_saturated_pixel_keeps_its_gradient_among_empty_pixels(tables):
    """A small, far silhouette: one pixel of the tile holds every candidate and is all but saturated (transmittance 2e-7), the other
    63 pixels hold none but carry upstream gradients of ordinary size.  The fixed-point scale of the tile's gradient accumulators is
    set by the pixels that can contribute; round 3 let the empty ones in, and their coefficients (1e7 times the saturated pixel's)
    quantised its whole gradient away: |error| / |gradient| = 0.5 (long fuzz of round 4, seed 1249; profiles/r4_fuzz_one.txt).
    Reference semantics: p3d_renderer.py:41-47 - every pixel's gradient, whatever its size."""
    from smilify_amd import engine as eng
    from oracle import render_ref
    from test_gpu_edge_cases import _scene

    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 9
    ndc = _scene(t, 1, S, 8.4, 1249)
    with render_ref.select_mode(1):
        sil, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    assert int((ncand > 0).sum()) <= 4 and float(sil.max()) > 0.99, (ncand.max(), sil.max())  # (the scene this test is about)
    gs = np.random.default_rng(5).standard_normal((1, S, S)).astype(np.float32)
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs)[..., :2]
    got = eng.silhouette_backward(dm, ndc.to(DEV), S, torch.from_numpy(gs).to(DEV)).cpu().numpy()
    assert np.linalg.norm(want) > 0
    assert np.linalg.norm(got - want) <= 1e-3 * np.linalg.norm(want), (np.linalg.norm(got - want), np.linalg.norm(want))
