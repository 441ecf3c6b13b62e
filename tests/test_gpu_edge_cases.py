"""Edge cases and size-independent properties of the HIP path (``pytest -m gpu``)."""
import numpy as np
import pytest
import torch

from conftest import MODEL_FILES, oracle_model
from oracle import fitter_ref, lbs_ref, render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _eng():
    from smilify_amd import engine

    return engine


def _scene(t, N, S, dist, seed, scale=1.0):
    m = oracle_model(t)
    g = torch.Generator().manual_seed(seed)
    theta = 0.15 * torch.randn(N, t.J, 3, generator=g)
    theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation()) + 0.1 * torch.randn(N, 3, generator=g)
    verts = lbs_ref.smal_forward(m, torch.zeros(N, t.nB), theta)["verts"] * scale
    R, T = render_ref.look_at_view_transform(dist, 15.0, torch.linspace(0, 300, N))
    return render_ref.project_to_ndc(verts, R, T, torch.full((N,), 60.0)).contiguous()


@pytest.mark.parametrize("S", [20, 60, 100])
def test_image_side_not_multiple_of_tile(S, tables):
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    ndc = _scene(t, 2, S, 2.2, 1)
    ref, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    got = eng.silhouette_forward(dm, ndc.to(DEV), S).cpu().numpy()
    assert np.abs(got - ref).max() < 2e-4


def test_nothing_visible_and_behind_camera(tables):
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 32
    ndc = _scene(t, 2, S, 2.2, 1)
    behind = ndc.clone()
    behind[..., 2] = -1.0                      # every face behind the camera
    off = ndc.clone()
    off[..., 0] += 5.0                         # everything outside the image
    for v in (behind, off):
        sil = eng.silhouette_forward(dm, v.to(DEV), S)
        assert float(sil.abs().max()) == 0.0
        target = (torch.rand(2, S, S) > 0.5).float().to(DEV)
        tsum = eng.image_abs_sum(target)
        li, dn, _ = eng.silhouette_l1_fused(dm, v.to(DEV), S, target, tsum, torch.ones(2, device=DEV))
        np.testing.assert_allclose(li.cpu().numpy(), target.sum(dim=(1, 2)).cpu().numpy(), rtol=1e-6)
        assert float(dn.abs().max()) == 0.0
    # NaN vertices must not fault or poison other faces
    bad = ndc.clone()
    bad[0, 0] = float("nan")
    sil = eng.silhouette_forward(dm, bad.to(DEV), S)
    assert torch.isfinite(sil).all()


@pytest.mark.parametrize("K", [1, 16, 17, 100, 128])
def test_faces_per_pixel_variants(K, tables):
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 48
    ndc = _scene(t, 2, S, 2.2, 3)
    ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
    got = eng.silhouette_forward(dm, ndc.to(DEV), S, eng.raster_settings(K=K)).cpu().numpy()
    d = np.abs(got - ref)
    assert d[ncand <= K].max() < 2e-4
    assert np.mean(d > 1e-3) < 0.02 and d.mean() < 2e-4, (np.mean(d > 1e-3), d.mean())
    with pytest.raises(Exception):
        eng.silhouette_forward(dm, ndc.to(DEV), S, eng.raster_settings(K=129))


@pytest.mark.parametrize("key,dist", [("stick", 14.0), ("mouse", 22.0)])
def test_dense_tile_processed_in_sub_tiles(key, dist, tables):
    """A whole mesh squeezed into ~2x2 tiles: thousands of faces per tile, far more (face, pixel) records than the
    workgroup's stream holds, so the kernel works through the tile in sub-tiles (and, for the mouse, has to halve
    them again after finding that its first estimate does not fit)."""
    eng = _eng()
    t = tables(key)
    dm = eng.DeviceModel(t, DEV)
    S = 32
    ndc = _scene(t, 1, S, dist, 2)  # far away -> tiny on screen
    ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    assert ncand.max() > 1024
    got = eng.silhouette_forward(dm, ndc.to(DEV), S).cpu().numpy()
    d = np.abs(got - ref)
    assert d.mean() < 5e-4 and np.mean(d > 1e-2) < 0.01, (d.mean(), d.max())
    with render_ref.select_mode(1):
        ref1, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    d1 = np.abs(got - ref1)
    assert d1.mean() < 2e-4 and np.mean(d1 > 1e-2) < 0.005, (d1.mean(), d1.max())
    # gradient path through the same code
    gs = torch.ones(1, S, S)
    want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy())[..., :2]
    gotg = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV)).cpu().numpy()
    rel = np.linalg.norm(gotg - want) / np.linalg.norm(want)
    assert rel < 1e-4 and np.abs(gotg - want).max() < 1e-4 * np.abs(want).max(), rel   # (measured: 1.2e-6 relative L2)


def test_full_size_properties_cfg2(tables):
    """BASELINE config 2 size (512 frames, 256^2, STICK): properties that need no oracle run."""
    eng = _eng()
    from smilify_amd import synthetic

    t = tables("stick")
    f = synthetic.make_problem(t, 512, 1, 256, DEV)
    f._refresh_targets()
    dm = f.device_model
    lbs = eng.lbs_forward(dm, f.betas.detach(), f._pose, trans=f.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
    cam = f.renderer.cameras
    cams = eng.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f.fov.detach(), None, 1, 256)
    ndc, _ = eng.project(cams, lbs["verts"], want_yx=False)
    sil = eng.silhouette_forward(dm, ndc, 256)
    assert float(sil.min()) >= 0.0 and float(sil.max()) <= 1.0
    # (a) the fused kernel's loss equals the L1 distance computed from the materialised silhouette
    scale = torch.full((512,), 1.0 / (256 * 256), device=DEV)
    li, dn, sil2 = eng.silhouette_l1_fused(dm, ndc, 256, f._sil_dev, f._sil_sum, scale, want_sil=True)
    want = (sil - f._sil_dev).abs().sum(dim=(1, 2))
    np.testing.assert_allclose(li.cpu().numpy(), want.cpu().numpy(), rtol=2e-4)
    assert torch.equal(sil, sil2)                        # (b) forward is deterministic and mode-independent
    # (c) explicit backward with the same upstream gradient reproduces the fused gradient (atomics: order noise only)
    gsil = torch.sign(sil - f._sil_dev) * scale[:, None, None]
    dn2 = eng.silhouette_backward(dm, ndc, 256, gsil.contiguous())
    num = (dn - dn2).norm().item() / (dn.norm().item() + 1e-30)
    assert num < 1e-4, num
    # (d) backward is linear in the upstream gradient
    dn3 = eng.silhouette_backward(dm, ndc, 256, (2.0 * gsil).contiguous())
    assert (dn3 - 2.0 * dn2).norm().item() / dn3.norm().item() < 1e-4
    # (e) rendering the target pose itself gives a (much) lower loss than the perturbed start
    objs0, _ = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, 0.0, window=10)
    assert torch.isfinite(objs0).all()
    # (f) a few fused Adam steps reduce the objective
    f.begin_stage(synthetic.STAGE1_LR)
    first = f.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)[:9].sum().item()
    for _ in range(4):
        last = f.fit_step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL)[:9].sum().item()
    assert last < first, (first, last)


def test_single_frame_reference_config(tables):
    """BASELINE config 1: 1 frame, 1 view, 256^2 (the only shape the shipped reference fitter runs)."""
    from smilify_amd import synthetic

    t = tables("stick")
    f = synthetic.make_problem(t, 1, 1, 256, DEV, window=1)
    loss, objs = f([0], synthetic.STAGE1_WEIGHTS, 1)
    loss.backward()
    assert set(objs) == {"joint", "limit", "pose", "splay", "betas", "sil_reproj"}
    assert f.global_rotation.grad is not None and f.joint_rotations.grad.shape == (1, t.J - 1, 3)
    assert f.fov.grad.shape == (1,) and torch.isfinite(f.fov.grad).all()
    # against the oracle
    cpu = lambda x: x.detach().cpu().clone()  # noqa: E731
    m = oracle_model(t)
    params = dict(betas=cpu(f.betas), log_beta_scales=cpu(f.log_beta_scales), betas_trans=cpu(f.betas_trans), global_rotation=cpu(f.global_rotation),
                  trans=cpu(f.trans), joint_rotations=cpu(f.joint_rotations), fov=cpu(f.fov))
    targets = dict(sil=cpu(f.sil_imgs), joints=cpu(f.target_joints), visibility=cpu(f.target_visibility))
    cams = dict(R=cpu(f.renderer.cameras.R), T=cpu(f.renderer.cameras.T))
    tot, o, _ = fitter_ref.fit_losses(m, params, [0], synthetic.STAGE1_WEIGHTS, targets, cams, 256, f.mean_betas.cpu(), f.betas_prec.cpu())
    assert abs(loss.item() - tot.item()) <= 1e-4 * abs(tot.item()), (loss.item(), tot.item())
    for k in o:
        assert abs(objs[k].item() - o[k].item()) <= 1e-4 * abs(o[k].item()) + 1e-7, k


def test_renderer_with_foreign_mesh_topology():
    """Renderer.forward on a mesh that does not come from a SMAL model (face table uploaded on first use)."""
    from smilify_amd.p3d_renderer import Renderer

    verts = torch.tensor([[[-0.5, -0.5, 0.0], [0.5, -0.5, 0.0], [0.0, 0.5, 0.0], [0.0, 0.0, 0.5]]], device=DEV, requires_grad=True)
    faces = torch.tensor([[0, 1, 2], [0, 1, 3], [1, 2, 3], [2, 0, 3]], device=DEV)
    r = Renderer(64, DEV)
    sil, proj = r(verts, verts[:, :2], faces[None])
    assert sil.shape == (1, 1, 64, 64) and 0.02 < sil.mean().item() < 0.5
    sil.sum().backward()
    assert torch.isfinite(verts.grad).all() and verts.grad.abs().sum() > 0
    R, T = render_ref.look_at_view_transform(2.7, 0.0, 0.0)
    ndc = render_ref.project_to_ndc(verts.detach().cpu(), R, T, torch.tensor([60.0]))
    ref, _ = render_ref.silhouette_forward_np(ndc.numpy(), faces.cpu().numpy().astype(np.int32), 64)
    assert np.abs(sil.detach().cpu().numpy()[0, 0] - ref[0]).max() < 2e-4
    with pytest.raises(NotImplementedError):
        r(verts, verts[:, :2], faces[None], render_texture=True)


def test_uint8_targets_equal_float_targets(tables):
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 48
    ndc = _scene(t, 3, S, 2.2, 4).to(DEV)
    tgt_f = (torch.rand(3, S, S) > 0.6).float().to(DEV)
    tgt_b = tgt_f.to(torch.uint8)
    scale = torch.tensor([1.0, 0.5, 2.0], device=DEV) / (S * S)
    sum_f, sum_b = eng.image_abs_sum(tgt_f), eng.image_abs_sum(tgt_b)
    assert torch.equal(sum_f, sum_b)
    lf, gf, _ = eng.silhouette_l1_fused(dm, ndc, S, tgt_f, sum_f, scale)
    lb, gb, _ = eng.silhouette_l1_fused(dm, ndc, S, tgt_b, sum_b, scale)
    np.testing.assert_allclose(lb.cpu().numpy(), lf.cpu().numpy(), rtol=1e-6)
    assert (gb - gf).norm().item() <= 1e-5 * gf.norm().item()
    with pytest.raises(Exception):
        eng.silhouette_l1_fused(dm, ndc, S, tgt_f.double(), sum_f, scale)


@pytest.mark.parametrize("key,dist", [("stick", 2.7), ("mouse", 4.0)])
def test_full_resolution_against_oracle(key, dist, tables):
    """256^2 (the BASELINE resolution) on the real models, two frames: silhouette, fused loss and gradient against the
    CPU oracle (naive rasteriser with the faithful K = 100 queue)."""
    eng = _eng()
    t = tables(key)
    dm = eng.DeviceModel(t, DEV)
    S = 256
    ndc = _scene(t, 2, S, dist, 31)
    tgt = _scene(t, 2, S, dist, 32)
    ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    target = (torch.from_numpy(render_ref.silhouette_forward_np(tgt.numpy(), t.faces, S)[0]) > 0.5).float()
    assert (ncand > 100).mean() > 0.01
    scale = torch.full((2,), 500.0 / (2 * S * S))
    gsil = torch.sign(torch.from_numpy(ref) - target) * scale[:, None, None]
    want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gsil.numpy())[..., :2]
    li, dn, sil = eng.silhouette_l1_fused(dm, ndc.to(DEV), S, target.to(torch.uint8).to(DEV), eng.image_abs_sum(target.to(DEV)),
                                          scale.to(DEV), want_sil=True)
    got = sil.cpu().numpy()
    d = np.abs(got - ref)
    assert d.mean() < 1e-6 and np.mean(d > 1e-4) < 1e-3, (d.mean(), np.mean(d > 1e-4))
    loss_ref = np.abs(ref - target.numpy()).sum(axis=(1, 2))
    np.testing.assert_allclose(li.cpu().numpy(), loss_ref, rtol=1e-4)     # the north-star tolerance
    g = dn.cpu().numpy()
    cos = (g * want).sum() / (np.linalg.norm(g) * np.linalg.norm(want))
    rel = np.linalg.norm(g - want) / np.linalg.norm(want)
    assert cos > 0.9999 and rel < 2e-2, (cos, rel)
    # against the kernel's own documented rule, (depth, face id) = oracle select_mode(1): same upstream gradient through the
    # explicit backward entry point, error <= 1e-3 of the largest component (fp32 rounding only), and the fused loss within
    # the north-star 1e-4 of BOTH rules
    with render_ref.select_mode(1):
        ref1, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
        want1 = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gsil.numpy())[..., :2]
    g1 = eng.silhouette_backward(dm, ndc.to(DEV), S, gsil.to(DEV).contiguous()).cpu().numpy()
    err1 = np.abs(g1 - want1) / np.abs(want1).max()
    assert err1.max() < 1e-3 and np.sqrt((err1 ** 2).mean()) < 1e-5, (err1.max(), np.sqrt((err1 ** 2).mean()))
    np.testing.assert_allclose(li.cpu().numpy(), np.abs(ref1 - target.numpy()).sum(axis=(1, 2)), rtol=1e-4)
    assert np.abs(got - ref1).max() < 2e-4 or np.mean(np.abs(got - ref1) > 1e-4) < 2e-4


def test_sliced_launches_equal_one_launch(tables, monkeypatch):
    """The engine cuts very large image batches into several rasteriser launches (bounded workspace); the results must
    not depend on where the cuts fall."""
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S, N = 40, 7
    ndc = _scene(t, N, S, 2.2, 5).to(DEV)
    tgt = (eng.silhouette_forward(dm, _scene(t, N, S, 2.2, 6).to(DEV), S) > 0.5).to(torch.uint8)
    scale = torch.linspace(0.5, 1.5, N, device=DEV) / (S * S)
    ref = eng.silhouette_l1_fused(dm, ndc, S, tgt, eng.image_abs_sum(tgt), scale, want_sil=True)
    ref_fwd = eng.silhouette_forward(dm, ndc, S)
    monkeypatch.setattr(eng, "MAX_IMAGES_PER_LAUNCH", 3)
    got = eng.silhouette_l1_fused(dm, ndc, S, tgt, eng.image_abs_sum(tgt), scale, want_sil=True)
    got_fwd = eng.silhouette_forward(dm, ndc, S)
    torch.testing.assert_close(got_fwd, ref_fwd, rtol=0, atol=1e-6)
    torch.testing.assert_close(got[2], ref[2], rtol=0, atol=1e-6)
    torch.testing.assert_close(got[0], ref[0], rtol=1e-5, atol=1e-6)  # (an image's loss is summed over its tiles by float atomics)
    assert dm._last_slice == 3  # (the second pair of calls really was cut: 3 + 3 + 1 images)
    # (seven images accumulate in float atomics: order noise of the largest contributions shows on components that cancel)
    torch.testing.assert_close(got[1], ref[1], rtol=1e-4, atol=2e-6 * float(ref[1].abs().max()))


@pytest.mark.parametrize("seed", range(10))
def test_random_configurations_against_oracle(seed, tables):
    """Random image size, K, camera distance and model: silhouette against the oracle's (depth, face id) rule and against
    the faithful queue, gradient against the oracle's backward.  Covers short and long tile lists, sub-tiles, K below and
    above the candidate counts, images that are not a multiple of the tile size."""
    eng = _eng()
    rng = np.random.default_rng(1000 + seed)
    key = ["synthetic", "stick", "stick", "mouse"][int(rng.integers(0, 4))]
    t = tables(key)
    dm = eng.DeviceModel(t, DEV)
    S = int(rng.integers(17, 97))
    K = int(rng.choice([1, 3, 10, 40, 100, 128]))
    dist = float(rng.uniform(2.0, 9.0)) * (1.5 if key == "mouse" else 1.0)
    ndc = _scene(t, 2, S, dist, 50 + seed)
    rs = eng.raster_settings(K=K)
    with render_ref.select_mode(1):
        ref1, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
    ref0, _ = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S, K=K)
    got = eng.silhouette_forward(dm, ndc.to(DEV), S, rs).cpu().numpy()
    d1, d0 = np.abs(got - ref1), np.abs(got - ref0)
    msg = f"{key} S={S} K={K} dist={dist:.2f} max candidates {ncand.max()}"
    assert d1.mean() < 3e-6 and np.mean(d1 > 1e-4) < 3e-3, (msg, d1.mean(), np.mean(d1 > 1e-4), d1.max())
    assert d1[ncand <= K].max(initial=0.0) < 2e-4, msg
    if K >= 40:  # the faithful queue: with a handful of slots and many equal depths its history-dependent pick is a different image
        assert abs(got.sum() - ref0.sum()) <= 2e-3 * max(ref0.sum(), 1.0), (msg, got.sum(), ref0.sum())
    gs = torch.from_numpy(rng.standard_normal((2, S, S)).astype(np.float32))
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy(), K=K)[..., :2]
    gotg = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV), rs).cpu().numpy()
    if np.linalg.norm(want) > 0:
        cos = (gotg * want).sum() / (np.linalg.norm(gotg) * np.linalg.norm(want) + 1e-30)
        rel = np.linalg.norm(gotg - want) / np.linalg.norm(want)
        assert cos > 0.99999 and rel < 2e-3, (msg, cos, rel)


def test_more_tiles_than_the_setup_kernel_counts(tables):
    """768^2 = 9216 tiles per image: above COUNT_TILES_MAX the setup kernel keeps a touched-tile bitmap instead of per-tile
    costs and queues every tile in one class; the result must not depend on that."""
    eng = _eng()
    t = tables("synthetic")
    dm = eng.DeviceModel(t, DEV)
    S = 768
    ndc = _scene(t, 1, S, 2.6, 12)
    ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    got = eng.silhouette_forward(dm, ndc.to(DEV), S).cpu().numpy()
    d = np.abs(got - ref)
    assert ref.sum() > 1000 and d.mean() < 2e-6 and np.mean(d > 1e-4) < 1e-3, (ref.sum(), d.mean(), np.mean(d > 1e-4))
    gs = torch.ones(1, S, S) / (S * S)
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy())[..., :2]
    gotg = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV)).cpu().numpy()
    assert np.linalg.norm(gotg - want) / np.linalg.norm(want) < 2e-3


def test_sixty_thousand_faces_in_a_handful_of_tiles():
    """A mesh near the rasteriser's face limit (65 536): a tube of 3 000 slivers per ring, 60 000 faces, drawn small - every
    touched tile lists tens of thousands of faces (list positions and bucket starts near their 16-bit range, records far
    beyond the stream, depth buckets with thousands of entries that saturate the 8-bit first-digit counts)."""
    from smilify_amd import model_io

    eng = _eng()
    t = model_io.synthetic_model(V_side=3000, J=9, nB=2, seed=5)
    assert 59000 < t.F <= 65536
    dm = eng.DeviceModel(t, DEV)
    S = 40
    ndc = _scene(t, 1, S, 3.0, 3)
    with render_ref.select_mode(1):
        ref, ncand = render_ref.silhouette_forward_np(ndc.numpy(), t.faces, S)
    assert ncand.max() > 4000
    got = eng.silhouette_forward(dm, ndc.to(DEV), S).cpu().numpy()
    d = np.abs(got - ref)
    assert d.mean() < 1e-6 and d.max() < 1e-4, (d.mean(), d.max())   # (measured: 4e-9 / 4e-7)
    gs = torch.ones(1, S, S)
    with render_ref.select_mode(1):
        want = render_ref.silhouette_backward_np(ndc.numpy(), t.faces, S, gs.numpy())[..., :2]
    gotg = eng.silhouette_backward(dm, ndc.to(DEV), S, gs.to(DEV)).cpu().numpy()
    rel = np.linalg.norm(gotg - want) / np.linalg.norm(want)
    assert rel < 1e-4 and np.abs(gotg - want).max() < 1e-4 * np.abs(want).max(), rel   # (measured: 1.2e-6 relative L2)
