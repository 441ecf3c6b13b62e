"""GPU tests of the fitter's state handling (``pytest -m gpu``): the reference-written per-frame checkpoint
(optimize_to_joints.py:48-63, fitter.py:352-371), hipGraph invalidation, the renderer's topology cache, batches beyond 65 535
frames / images."""
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN, vertex_probe
from oracle import render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"

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
    """A captured iteration bakes in device addresses; after set_cameras / a re-assigned mask / a regrown workspace / new rasteriser
    settings the next fit_step_graph must re-capture and agree with the eager step."""
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
    fourth = fg._graph["graph"]
    for f in (fe, fg):                          # the rasteriser settings travel by value into the captured launches
        f.renderer.raster_settings = _eng.raster_settings(tie_rule="reference_queue")
    both()
    assert fg._graph["graph"] is not fourth

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
