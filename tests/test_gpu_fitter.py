"""GPU parity of the drop-in classes and the fused fit step (run with ``pytest -m gpu`` on the MI355X box)."""
import numpy as np
import pytest
import torch

from conftest import oracle_model, vertex_probe
from oracle import fitter_ref, lbs_ref, render_ref

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
PARAMS = ["betas", "log_beta_scales", "betas_trans", "global_rotation", "trans", "joint_rotations", "fov"]


def _fitter_from_golden(g, t, views=1):
    from smilify_amd.config import FitterConfig
    from smilify_amd.fitter import SMALFitter

    S = int(g["S"])
    N = g["param_trans"].shape[0]
    rgb = torch.zeros(N, 3, S, S)
    data = (rgb, torch.from_numpy(g["sil_target"]), torch.from_numpy(g["target_joints"]), torch.from_numpy(g["visibility"]))
    f = SMALFitter(DEV, data, N, -1, False, tables=t, config=FitterConfig.from_tables(t, WINDOW_SIZE=N), views=views)
    f.set_cameras(torch.from_numpy(g["R"]), torch.from_numpy(g["T"]), fov=torch.from_numpy(g["param_fov"]))
    with torch.no_grad():
        f.betas.copy_(torch.from_numpy(g["param_betas"]))
        f.log_beta_scales.copy_(torch.from_numpy(g["param_log_beta_scales"]))
        f.betas_trans.copy_(torch.from_numpy(g["param_betas_trans"]))
        f.global_rotation.copy_(torch.from_numpy(g["param_global_rotation"]))
        f.joint_rotations.copy_(torch.from_numpy(g["param_joint_rotations"]))
        f.trans.copy_(torch.from_numpy(g["param_trans"]))
    f.log_beta_scales.requires_grad = True
    f.betas_trans.requires_grad = True
    return f


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_fitter_forward_matches_reference_golden(key, golden, tables):
    """Loss terms + gradients of the REAL reference SMALFitter.forward (its renderer swapped for the oracle's)."""
    g = golden(f"fitter_{key}")
    f = _fitter_from_golden(g, tables(key))
    N = f.num_images
    np.testing.assert_allclose(f.mean_betas.cpu().numpy(), g["mean_betas"], atol=1e-6)
    np.testing.assert_allclose(f.betas_prec.cpu().numpy(), g["betas_prec"], rtol=1e-5, atol=1e-6)
    loss, objs = f(list(range(N)), g["weights"], 1)
    jl, gl, tl = f.get_temporal(100.0)
    (loss + jl + gl + tl).backward()
    for k in ("joint", "limit", "pose", "splay", "betas", "sil_reproj"):
        ref = float(g[f"obj_{k}"])
        assert abs(objs[k].item() - ref) <= 1e-4 * abs(ref) + 1e-7, (k, objs[k].item(), ref)
    assert abs(loss.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    np.testing.assert_allclose(sorted([jl.item(), gl.item(), tl.item()]), sorted(g["temporal"].tolist()), rtol=1e-4, atol=1e-7)
    assert abs((jl + gl + tl).item() - float(g["temporal"].sum())) <= 1e-4 * float(g["temporal"].sum())
    for n in PARAMS:
        ref = g[f"grad_{n}"]
        got = getattr(f, n).grad.cpu().numpy().reshape(ref.shape)
        scale = np.abs(ref).max() + 1e-12
        err = np.abs(got - ref) / scale
        # the silhouette gradient is a sum of float atomics over ~1e5 (pixel, face) pairs
        assert err.max() < 5e-3, (n, err.max())
        assert np.sqrt((err ** 2).mean()) < 5e-4, (n, np.sqrt((err ** 2).mean()))


def _oracle_problem(fitter, tables):
    cpu = lambda t: t.detach().cpu().clone()  # noqa: E731
    m = oracle_model(tables)
    params = dict(betas=cpu(fitter.betas), log_beta_scales=cpu(fitter.log_beta_scales), betas_trans=cpu(fitter.betas_trans),
                  global_rotation=cpu(fitter.global_rotation), trans=cpu(fitter.trans), joint_rotations=cpu(fitter.joint_rotations),
                  fov=cpu(fitter.fov))
    targets = dict(sil=cpu(fitter.sil_imgs), joints=cpu(fitter.target_joints), visibility=cpu(fitter.target_visibility))
    cams = dict(R=cpu(fitter.renderer.cameras.R), T=cpu(fitter.renderer.cameras.T))
    return m, params, targets, cams


@pytest.mark.parametrize("model_key,static", [("synthetic", False), ("synthetic_static", True)])
def test_fit_step_matches_oracle_iteration(model_key, static, tables):
    """Whole fused epoch (sum over windows of window means + temporal, backward, Adam) vs the oracle + torch.optim.Adam."""
    from smilify_amd import synthetic

    t = tables(model_key)
    frames, S, W = 5, 40, 2
    fitter = synthetic.make_problem(t, frames, 1, S, DEV, radius=2.2, seed=3, window=W)
    m, params, targets, cams = _oracle_problem(fitter, t)
    for k in ("betas", "log_beta_scales", "global_rotation", "trans", "joint_rotations", "fov"):
        params[k].requires_grad_()
    windows = [list(range(s, min(frames, s + W))) for s in range(0, frames, W)]
    weights, w_temp = synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL
    opt = torch.optim.Adam([{"params": [params[k] for k in params if k != "fov" and params[k].requires_grad], "lr": 5e-3},
                            {"params": [params["fov"]], "lr": 1.0}], lr=5e-3, betas=(0.5, 0.999))
    fitter.begin_stage(5e-3, fov_lr=1.0)
    for it in range(2):
        opt.zero_grad()
        total, _, _ = fitter_ref.fit_iteration_loss(m, params, windows, weights, w_temp, targets, cams, S, fitter.mean_betas.cpu(),
                                                    fitter.betas_prec.cpu())
        total.backward()
        opt.step()
        objs = fitter.fit_step(weights, w_temp, window=W)
        got = objs[:9].sum().item()
        assert abs(got - total.item()) <= 1e-4 * abs(total.item()), (it, got, total.item())
    # parameters after two Adam steps (Adam normalises the step, so tiny gradient noise can flip tiny components)
    for name in ("global_rotation", "joint_rotations", "trans", "betas", "fov", "log_beta_scales"):
        a, b = getattr(fitter, name).detach().cpu().numpy(), params[name].detach().numpy()
        d = np.abs(a - b.reshape(a.shape))
        assert np.median(d) < 2e-4 and np.mean(d < 2e-3) > 0.97, (name, np.median(d), d.max())


def test_fit_step_multiview_and_sharded_equivalence(tables):
    """(i) 3 cameras per frame against the oracle; (ii) two window-aligned shards + halos + summed shared
    gradients reproduce the single-rank loss and gradients exactly."""
    from smilify_amd import synthetic

    t = tables("synthetic")
    frames, views, S, W = 4, 3, 32, 2
    fitter = synthetic.make_problem(t, frames, views, S, DEV, radius=2.4, seed=5, window=W)
    weights, w_temp = synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL
    objs, grads = fitter._loss_and_grads(None, weights, w_temp, window=W)
    # oracle: loss of window w = mean over (b_w * views) images -> evaluate per view and average
    m, params, targets, cams = _oracle_problem(fitter, t)
    total = 0.0
    for v in range(views):
        sel = [f * views + v for f in range(frames)]
        tv = dict(sil=targets["sil"][sel], joints=targets["joints"][sel], visibility=targets["visibility"][sel])
        cv = dict(R=cams["R"][v:v + 1], T=cams["T"][v:v + 1])
        for s in range(0, frames, W):
            tot, o, _ = fitter_ref.fit_losses(m, params, range(s, min(frames, s + W)), weights, tv, cv, S, fitter.mean_betas.cpu(),
                                              fitter.betas_prec.cpu())
            total = total + (o["joint"] + o["sil_reproj"]) / views + (o["limit"] + o["pose"] + o["splay"] + o["betas"]) / views
    jl, gl, tl = fitter_ref.temporal(params, w_temp)
    total = total + jl + gl + tl
    assert abs(objs[:9].sum().item() - float(total)) <= 1e-4 * abs(float(total))

    # sharded: frames [0,2) and [2,4)
    rows = torch.cat([fitter._pose.reshape(frames, -1), fitter.trans.detach()], 1)
    acc_objs = torch.zeros_like(objs)
    parts = []
    for f0 in (0, 2):
        sub = synthetic.make_problem(t, 2, views, S, DEV, radius=2.4, seed=5, window=W, frame0=f0, n_frames_total=frames)
        with torch.no_grad():
            sub._pose.copy_(fitter._pose[f0:f0 + 2]); sub.trans.copy_(fitter.trans[f0:f0 + 2])
            sub.betas.copy_(fitter.betas); sub.log_beta_scales.copy_(fitter.log_beta_scales[f0:f0 + 2])
        sub.sil_imgs = fitter.sil_imgs[f0 * views:(f0 + 2) * views]
        sub.target_joints = fitter.target_joints[f0 * views:(f0 + 2) * views]
        o2, g2 = sub._loss_and_grads(None, weights, w_temp, window=W, halo_prev=rows[f0 - 1].contiguous() if f0 else None,
                                     halo_next=rows[f0 + 2].contiguous() if f0 + 2 < frames else None)
        acc_objs += o2
        parts.append(g2)
    np.testing.assert_allclose(acc_objs.cpu().numpy(), objs.cpu().numpy(), rtol=2e-5, atol=1e-6)
    np.testing.assert_allclose(torch.cat([p["pose"] for p in parts]).cpu().numpy(), grads["pose"].cpu().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose((parts[0]["betas"] + parts[1]["betas"]).cpu().numpy(), grads["betas"].cpu().numpy(), rtol=1e-3, atol=1e-5)
    np.testing.assert_allclose((parts[0]["fov"] + parts[1]["fov"]).cpu().numpy(), grads["fov"].cpu().numpy(), rtol=1e-3, atol=1e-5)


@pytest.mark.parametrize("key", ["stick", "synthetic_static"])
def test_smal_and_renderer_dropins(key, tables):
    """Reference-style use: SMAL(...)(beta, theta, ...) -> Renderer(...)(verts, joints, faces) with autograd."""
    from smilify_amd.p3d_renderer import Renderer
    from smilify_amd.smal_torch import SMAL

    t = tables(key)
    m = oracle_model(t)
    B, S = 2, 48
    g = torch.Generator().manual_seed(21)
    beta = (0.3 * torch.randn(B, t.nB, generator=g))
    theta = 0.2 * torch.randn(B, t.J, 3, generator=g)
    theta[:, 0] = torch.from_numpy(fitter_ref.default_global_rotation())
    trans = 0.05 * torch.randn(B, 3, generator=g)
    smal = SMAL(DEV, tables=t)
    assert smal.faces.dtype == torch.int64 and tuple(smal.weights.shape) == (t.V, t.J) and tuple(smal.J_regressor.shape) == (t.V, t.J)
    leaves_g = [x.clone().to(DEV).requires_grad_() for x in (beta, theta, trans)]
    verts, joints, Rs, v_shaped = smal(leaves_g[0], leaves_g[1], trans=leaves_g[2])
    assert smal.J_transformed.shape == (B, t.J, 3)
    rend = Renderer(S, DEV)
    fov = torch.tensor([55.0], device=DEV, requires_grad=True)
    rend.cameras.fov = fov
    sil, proj = rend(verts, joints, smal.faces.unsqueeze(0).expand(B, -1, -1))
    assert sil.shape == (B, 1, S, S) and proj.shape == (B, t.J, 2)
    w_s, w_p = vertex_probe(sil.shape, 4).to(DEV), vertex_probe(proj.shape, 5).to(DEV)
    ((sil * w_s).sum() + 1e-3 * (proj * w_p).sum()).backward()

    leaves_o = [x.clone().requires_grad_() for x in (beta, theta, trans)]
    fov_o = torch.tensor([55.0], requires_grad=True)
    out = lbs_ref.smal_forward(m, leaves_o[0], leaves_o[1], trans=leaves_o[2])
    R, T = render_ref.look_at_view_transform(2.7, 0.0, 0.0)
    oren = render_ref.OracleRenderer(S, R, T, fov_o)
    sil_o, proj_o = oren(out["verts"], out["joints"], m["faces"])
    ((sil_o * w_s.cpu()).sum() + 1e-3 * (proj_o * w_p.cpu()).sum()).backward()
    np.testing.assert_allclose(verts.detach().cpu().numpy(), out["verts"].detach().numpy(), rtol=1e-4, atol=5e-6)
    np.testing.assert_allclose(proj.detach().cpu().numpy(), proj_o.detach().numpy(), rtol=1e-4, atol=1e-3)
    assert np.abs(sil.detach().cpu().numpy() - sil_o.detach().numpy()).mean() < 5e-6
    for a, b, n in zip(leaves_g + [fov], leaves_o + [fov_o], ["beta", "theta", "trans", "fov"]):
        ref = b.grad.numpy()
        sc = np.abs(ref).max() + 1e-12
        err = np.abs(a.grad.cpu().numpy() - ref) / sc
        assert err.max() < 1e-2 and np.sqrt((err ** 2).mean()) < 2e-3, (n, err.max())
    # joints_only branch
    none, proj2 = rend(verts.detach(), joints.detach(), smal.faces, joints_only=True)
    assert none is None and torch.allclose(proj2, proj.detach())


@pytest.mark.parametrize("key,radius,tie_rule", [("stick", 2.7, None), ("mouse", 4.0, None), ("stick", 2.7, "reference_queue")])
def test_graph_captured_step_equals_eager_step(key, radius, tie_rule, tables):
    """fit_step_graph (one hipGraph replay per iteration) against fit_step (one launch per kernel): same losses and the
    same parameters after several iterations, also when eager steps are mixed in.  (The mouse: the one-workgroup-per-CU form of the
    fused LBS kernels, 139 KB of dynamic LDS, inside a captured graph.  Third case: the reference's tie rule - the replay kernel, its
    ticket counter and the tie masks inside the captured graph.)"""
    from smilify_amd import engine, synthetic

    t = tables(key)
    runs = {}
    for mode in ("eager", "graph", "mixed"):
        f = synthetic.make_problem(t, 6, 2, 64, DEV, radius=radius, seed=11, window=3)
        if tie_rule:
            f.renderer.raster_settings = engine.raster_settings(tie_rule=tie_rule)
        f.begin_stage(synthetic.STAGE1_LR)
        objs = []
        for it in range(6):
            use_graph = mode == "graph" or (mode == "mixed" and it not in (2, 3))
            step = f.fit_step_graph if use_graph else f.fit_step
            objs.append(step(synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL).clone())
        torch.cuda.synchronize()
        if tie_rule:
            assert engine.raster_stats(f.device_model, 12)["tie_pixels"] > 0  # (the replay kernel did run in the last iteration)
        runs[mode] = (torch.stack(objs).cpu(), {n: getattr(f, n).detach().cpu().clone() for n in PARAMS})
    ref_objs, ref_par = runs["eager"]
    for mode in ("graph", "mixed"):
        o, par = runs[mode]
        np.testing.assert_allclose(o.numpy(), ref_objs.numpy(), rtol=2e-4, atol=1e-6)
        for n in PARAMS:  # float atomics make the gradient sums order dependent: compare to Adam-step resolution
            np.testing.assert_allclose(par[n].numpy(), ref_par[n].numpy(), atol=2e-4, err_msg=f"{mode}: {n}")
    assert float(ref_objs[-1].sum()) < float(ref_objs[0].sum())


def test_temporal_terms_are_three_independent_graphs(tables):
    """Reference fitter.py:337-350 returns three scalars that each carry a graph; a caller may weight or drop any of them.
    Every term against the oracle's, each differentiated ALONE: the joint term reaches joint_rotations only, the global term
    global_rotation only, the translation term trans only."""
    from oracle import fitter_ref
    from smilify_amd import synthetic

    t = tables("synthetic")
    f = synthetic.make_problem(t, 6, 1, 32, DEV, radius=2.2, seed=4, window=3)
    cpu = lambda x: x.detach().cpu().clone().requires_grad_()  # noqa: E731
    params = dict(global_rotation=cpu(f.global_rotation), joint_rotations=cpu(f.joint_rotations), trans=cpu(f.trans))
    want = fitter_ref.temporal(params, 37.0)
    names = ("joint_rotations", "global_rotation", "trans")
    for i, own in enumerate(names):
        for p in (f.global_rotation, f.joint_rotations, f.trans):
            p.grad = None
        for p in params.values():
            p.grad = None
        got = f.get_temporal(37.0)
        assert abs(got[i].item() - want[i].item()) <= 1e-5 * abs(want[i].item()) + 1e-9
        (2.5 * got[i]).backward()
        (2.5 * want[i]).backward()
        for n in names:
            g = getattr(f, n).grad
            if n == own:
                np.testing.assert_allclose(g.cpu().numpy(), params[n].grad.numpy(), rtol=2e-4, atol=1e-7, err_msg=f"{own}/{n}")
            else:
                assert g is None or float(g.abs().max()) == 0.0, (own, n)
