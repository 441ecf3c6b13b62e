"""Stage scheduler and checkpoint format on the GPU (``pytest -m gpu``)."""
import os
import pickle

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_staged_optimisation_freezes_and_converges(tables):
    from smilify_amd import optimize, synthetic

    t = tables("synthetic")
    f = synthetic.make_problem(t, 6, 1, 40, DEV, radius=2.2, seed=11, window=3)
    f.config.TORSO_JOINTS = [0, 1, 5]
    f.config.OPT_WEIGHTS = [[25.0, 10.0], [0.0, 500.0], [0.0, 1.0], [0.0, 1.0], [0.0, 100.0], [0.0, 0.1], [500.0, 100.0], [6, 8], [2e-2, 5e-3]]
    f.log_beta_scales.requires_grad = False
    joints0, betas0, fov0 = f.joint_rotations.detach().clone(), f.betas.detach().clone(), f.fov.detach().clone()
    seen = []
    stages = optimize.stages_from_config(f.config)
    assert len(stages) == 2 and stages[0].epochs == 6

    def on_epoch(stage, epoch, objs):
        seen.append((stage, epoch, float(objs[:9].sum())))
        if stage == 0 and epoch == stages[0].epochs - 1:
            # stage 0 (optimize_to_joints.py:129-138): joint rotations / betas frozen, visibility masked to the torso
            assert torch.equal(f.joint_rotations.detach(), joints0) and torch.equal(f.betas.detach(), betas0)
            assert not torch.equal(f.fov.detach(), fov0)
            assert int(f.target_visibility.sum()) == 6 * 3

    optimize.optimize(f, stages, on_epoch=on_epoch)
    assert [s for s, _, _ in seen] == [0] * 6 + [1] * 8
    assert not torch.equal(f.joint_rotations.detach(), joints0) and not torch.equal(f.betas.detach(), betas0)
    assert int(f.target_visibility.sum()) == 6 * t.J
    s0 = [l for s, _, l in seen if s == 0]
    s1 = [l for s, _, l in seen if s == 1]
    assert s0[-1] < s0[0] and s1[-1] < s1[0], (s0, s1)


def test_checkpoint_roundtrip(tables, tmp_path):
    """Per-frame parameter dicts as the reference pickles them (optimize_to_joints.py:48-63) -> load_checkpoint."""
    from smilify_amd import synthetic

    t = tables("synthetic")
    f = synthetic.make_problem(t, 3, 1, 32, DEV, radius=2.2, seed=2)
    for i in range(3):
        d = tmp_path / "{0:04}".format(i)
        os.makedirs(d)
        p = f.export_parameters(i)
        assert set(p) >= {"global_rotation", "joint_rotations", "betas", "trans", "fov", "log_betascale"}
        with open(d / "st1_ep0.pkl", "wb") as fh:
            pickle.dump(p, fh)
    g = synthetic.make_problem(t, 3, 1, 32, DEV, radius=2.2, seed=99)
    g.load_checkpoint(str(tmp_path), "st1_ep0")
    np.testing.assert_allclose(g.global_rotation.detach().cpu().numpy(), f.global_rotation.detach().cpu().numpy(), atol=1e-7)
    np.testing.assert_allclose(g.joint_rotations.detach().cpu().numpy(), f.joint_rotations.detach().cpu().numpy(), atol=1e-7)
    np.testing.assert_allclose(g.trans.detach().cpu().numpy(), f.trans.detach().cpu().numpy(), atol=1e-7)
    np.testing.assert_allclose(g.betas.detach().cpu().numpy(), f.betas.detach().cpu().numpy(), atol=1e-6)
    # scales are averaged over frames like the reference does (fitter.py:371)
    np.testing.assert_allclose(g.log_beta_scales.detach().cpu().numpy()[0], f.log_beta_scales.detach().cpu().numpy().mean(0), atol=1e-6)
    # the loaded fitter is usable: shared (1,J,3) scale table
    loss, _ = g([0, 1, 2], synthetic.STAGE1_WEIGHTS, 1)
    assert torch.isfinite(loss)


def test_reference_style_driver_loop(tables):
    """The reference's own loop shape (optimize_to_joints.py:110-178): torch.optim.Adam over named_parameters with fov in
    its own group, in-place visibility masking, forward per window + get_temporal, one backward, one step."""
    from smilify_amd import synthetic

    t = tables("synthetic")
    N, W = 6, 2
    model = synthetic.make_problem(t, N, 1, 40, DEV, radius=2.2, seed=21, window=W)
    model.log_beta_scales.requires_grad = False
    torso = [0, 1, 5]
    full_vis = model.target_visibility.clone()
    losses = {}
    for stage_id, (weights, w_temp, epochs, lr) in enumerate([([25.0, 0, 0, 0, 0, 0], 500.0, 4, 2e-2), ([10.0, 500.0, 1, 1, 100.0, 0.1], 100.0, 4, 5e-3)]):
        optimizer = torch.optim.Adam([{"params": [p for n, p in model.named_parameters() if n != "fov"], "lr": lr},
                                      {"params": [model.fov], "lr": 1}], lr=lr, betas=(0.5, 0.999))
        if stage_id == 0:
            model.joint_rotations.requires_grad = False
            model.betas.requires_grad = False
            target_visibility = model.target_visibility.clone()
            model.target_visibility *= 0
            model.target_visibility[:, torso] = target_visibility[:, torso]      # in-place edit, as the reference does
        else:
            model.joint_rotations.requires_grad = True
            model.betas.requires_grad = True
            model.target_visibility = full_vis.clone()
        for epoch in range(epochs):
            acc_loss = 0
            optimizer.zero_grad()
            for j in range(0, N, W):
                loss, objs = model(list(range(j, min(N, j + W))), weights, stage_id)
                acc_loss += loss.mean()
            jl, gl, tl = model.get_temporal(w_temp)
            acc_loss = acc_loss + jl + gl + tl
            acc_loss.backward()
            optimizer.step()
            losses.setdefault(stage_id, []).append(float(acc_loss))
        if stage_id == 0:
            assert int(model._vis_dev.sum()) == N * len(torso), "in-place visibility mask was not picked up"
    assert losses[0][-1] < losses[0][0] and losses[1][-1] < losses[1][0], losses
    # the same two stages through the fused driver give the same first-epoch loss of stage 0
    ref = synthetic.make_problem(t, N, 1, 40, DEV, radius=2.2, seed=21, window=W)
    ref.log_beta_scales.requires_grad = False
    vis = torch.zeros_like(ref.target_visibility)
    vis[:, torso] = ref.target_visibility[:, torso]
    ref.target_visibility = vis
    ref.joint_rotations.requires_grad = False
    ref.betas.requires_grad = False
    ref.begin_stage(2e-2)
    objs = ref.fit_step([25.0, 0, 0, 0, 0, 0], 500.0, window=W)
    assert abs(float(objs[:9].sum()) - losses[0][0]) <= 1e-4 * abs(losses[0][0])


def _run_bench(*flags):
    import json
    import subprocess
    import sys

    from conftest import REPO

    out = subprocess.run([sys.executable, os.path.join(REPO, "bench.py"), *flags], capture_output=True, text=True, timeout=900, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_launches_its_own_ranks_and_shards_like_one_rank():
    """``python bench.py --gpus 2`` without a launcher starts torch.distributed.run as a child and relays rank 0's single
    JSON line; two ranks (gloo, sharing this box's one GPU) over 2 x 16 frames end at the loss one rank reaches on the
    same 32 frames: frames shard, the shared-parameter gradients are all-reduced, the temporal halo is exchanged."""
    common = ["--workload", "tiny", "--steps", "3", "--warmup", "0", "--cpu-frames", "0"]
    two = _run_bench("--gpus", "2", "--backend", "gloo", "--share-gpu", *common)
    one = _run_bench("--gpus", "1", "--frames", "32", *common)
    assert two["n_gpus"] == 2 and two["rehearsal"] is True and one["n_gpus"] == 1
    assert two["config"]["frames_per_gpu"] == 16 and one["config"]["frames_per_gpu"] == 32
    for k in ("metric", "value", "unit", "ms_per_step", "roofline", "scaling", "dtype", "config"):
        assert k in two, k
    assert "cpu_baseline" not in two
    assert abs(two["final_loss"] - one["final_loss"]) <= 2e-4 * abs(one["final_loss"]), (two["final_loss"], one["final_loss"])


def test_four_ranks_on_one_gpu_follow_one_rank():
    """The launcher path with more than two children: ``bench.py --gpus 4 --backend gloo --share-gpu`` (four ranks on this box's one
    GPU; the box admits at most six GPU processes at once, the test runner being one of them, so the eight-rank case runs on the
    CPU in tests/test_distributed_cpu.py) against one rank on the same 4 x 16 frames: three interior shard boundaries with
    both halo directions, the in-place sum of the shared block over four contributions."""
    common = ["--workload", "tiny", "--steps", "3", "--warmup", "0", "--cpu-frames", "0"]
    four = _run_bench("--gpus", "4", "--backend", "gloo", "--share-gpu", *common)
    one = _run_bench("--gpus", "1", "--frames", "64", *common)
    assert four["n_gpus"] == 4 and four["rehearsal"] is True and four["config"]["frames_per_gpu"] == 16
    assert abs(four["frame_iters_per_sec"] - 4 * 16 / (four["ms_per_step"] * 1e-3)) <= 1e-6 * four["frame_iters_per_sec"]  # whole-job frames per second
    assert abs(four["value"] - 4 * 1000.0 / four["ms_per_step"]) <= 1e-6 * four["value"]  # BASELINE's metric: fit iterations per second, summed over the GPUs
    assert four["rccl_ranks"] == 0 and len(four["rank_ms_per_step"]["all"]) == 4  # (gloo rehearsal: no RCCL ranks)
    assert abs(four["final_loss"] - one["final_loss"]) <= 2e-4 * abs(one["final_loss"]), (four["final_loss"], one["final_loss"])


_RCCL_PROBE = r"""
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, os.getcwd())
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%d" % int(sys.argv[1]), rank=0, world_size=1, device_id=dev)
from smilify_amd import optimize
shared = {"betas": torch.arange(5.0, device=dev), "fov": torch.full((1,), 3.0, device=dev)}
objs = torch.arange(10.0, device=dev)
want = {k: v.clone() for k, v in shared.items()}
optimize.allreduce_shared(shared, objs)                  # device tensors through RCCL, as bench.py --gpus N does
assert all(torch.equal(shared[k], want[k]) for k in shared) and torch.equal(objs, torch.arange(10.0, device=dev))
block = torch.arange(16.0, device=dev)                   # the fitter's shared block: summed in place, asynchronously
h = optimize.allreduce_block(block)
side = torch.ones(8, device=dev) * 2                     # (work launched beside the collective)
h.wait()
assert torch.equal(block, torch.arange(16.0, device=dev)) and float(side.sum()) == 16.0
assert optimize.exchange_halos(torch.ones(168, device=dev), torch.ones(168, device=dev), 0, 1) == (None, None)
t = torch.tensor([1.5], device=dev)
dist.all_reduce(t, op=dist.ReduceOp.MAX)                 # the bench's max-over-ranks clock
dist.barrier()
torch.cuda.synchronize()
dist.destroy_process_group()
print("RCCL_OK")
"""


def test_rccl_collectives_of_the_sharded_path_run_on_device_tensors():
    """The box has one GPU, so RCCL is exercised with a one-rank group: backend "nccl" initialises, and the fused
    all-reduce (general and in-place asynchronous forms), the MAX reduce and the barrier of the multi-GPU path accept
    this build's device tensors (the N > 1 arithmetic and the point-to-point halo are covered by the gloo tests)."""
    import subprocess
    import sys

    from conftest import REPO

    out = subprocess.run([sys.executable, "-c", _RCCL_PROBE, "29631"], capture_output=True, text=True, timeout=600, cwd=REPO)
    assert out.returncode == 0 and "RCCL_OK" in out.stdout, (out.stdout[-1000:], out.stderr[-3000:])


@pytest.mark.parametrize("use_graph", [True, False])
def test_two_stage_schedule_follows_the_reference_trajectory(use_graph, tables):
    """optimize.optimize (one hipGraph per stage, or eager) against the reference's loop restated on the oracle
    (optimize_to_joints.py:111-175): per stage a fresh torch.optim.Adam(betas=(0.5, 0.999)) over every parameter with fov in
    its own lr = 1 group; stage 0 freezes joint rotations / betas / limb scales and masks the visibility to the torso
    joints; every epoch = sum over windows of the window mean + temporal terms, one backward, one step."""
    from conftest import oracle_model
    from oracle import fitter_ref
    from smilify_amd import optimize, synthetic

    t = tables("synthetic")
    frames, S, W = 4, 32, 2
    f = synthetic.make_problem(t, frames, 1, S, DEV, radius=2.2, seed=21, window=W)
    f.config.TORSO_JOINTS = [0, 1, 4]
    f.config.OPT_WEIGHTS = [[25.0, 10.0], [0.0, 500.0], [0.0, 1.0], [0.0, 1.0], [0.0, 100.0], [0.0, 0.1], [500.0, 100.0], [3, 3], [2e-2, 5e-3]]
    cpu = lambda x: x.detach().cpu().clone()  # noqa: E731
    m = oracle_model(t)
    params = dict(betas=cpu(f.betas), log_beta_scales=cpu(f.log_beta_scales), betas_trans=cpu(f.betas_trans), global_rotation=cpu(f.global_rotation),
                  trans=cpu(f.trans), joint_rotations=cpu(f.joint_rotations), fov=cpu(f.fov))
    targets = dict(sil=cpu(f.sil_imgs).float(), joints=cpu(f.target_joints), visibility=cpu(f.target_visibility))
    cams = dict(R=cpu(f.renderer.cameras.R), T=cpu(f.renderer.cameras.T))
    mean_b, prec = f.mean_betas.cpu(), f.betas_prec.cpu()
    windows = [list(range(s0, min(frames, s0 + W))) for s0 in range(0, frames, W)]
    stages = optimize.stages_from_config(f.config)

    want = []
    full_vis = targets["visibility"].clone()
    for stage_id, st in enumerate(stages):
        frozen = ("joint_rotations", "betas", "log_beta_scales") if stage_id == 0 else ()
        for k, p in params.items():
            p.requires_grad_(k != "betas_trans" and k not in frozen)
        vis = full_vis.clone()
        if stage_id == 0:
            vis = torch.zeros_like(full_vis)
            vis[:, f.config.TORSO_JOINTS] = full_vis[:, f.config.TORSO_JOINTS]
        tg = dict(targets, visibility=vis)
        opt = torch.optim.Adam([{"params": [p for k, p in params.items() if k != "fov"], "lr": st.lr}, {"params": [params["fov"]], "lr": 1.0}],
                               lr=st.lr, betas=(0.5, 0.999))
        for _ in range(st.epochs):
            opt.zero_grad()
            total, _, _ = fitter_ref.fit_iteration_loss(m, params, windows, st.weights, st.w_temp, tg, cams, S, mean_b, prec)
            total.backward()
            opt.step()
            want.append(float(total))

    got = []
    optimize.optimize(f, stages, on_epoch=lambda s_, e_, objs: got.append(float(objs[:9].sum())), use_graph=use_graph)
    np.testing.assert_allclose(got, want, rtol=2e-4)
    assert (f._graph is not None) == use_graph
    for name in ("global_rotation", "joint_rotations", "trans", "betas", "fov", "log_beta_scales"):
        a, b = getattr(f, name).detach().cpu().numpy(), params[name].detach().numpy().reshape(getattr(f, name).shape)
        d = np.abs(a - b)
        assert np.median(d) < 2e-4 and np.mean(d < 3e-3) > 0.97, (name, np.median(d), d.max())


def _reference_epoch(model, N, W, weights, w_temp, window_weight=None):
    """One epoch body of optimize_to_joints.py:153-172 without the optimiser step: per-window losses, total, parameter gradients."""
    for p in model.parameters():
        p.grad = None
    acc, per_window, terms = 0, [], []
    for k, j in enumerate(range(0, N, W)):
        loss, objs = model(list(range(j, min(N, j + W))), weights, 1)
        per_window.append(float(loss))
        terms.append({n: float(v) for n, v in objs.items()})
        acc = acc + loss.mean() * (1.0 if window_weight is None else window_weight[k])
    jl, gl, tl = model.get_temporal(w_temp)
    (acc + jl + gl + tl).backward()
    grads = {n: (None if p.grad is None else p.grad.detach().clone()) for n, p in model.named_parameters()}
    return per_window, terms, float(acc), grads


@pytest.mark.parametrize("frames,window,views", [(7, 2, 1), (8, 4, 1), (6, 2, 2)])
def test_forward_serves_an_epoch_from_one_evaluation(tables, frames, window, views):
    """The per-window forward() calls of the reference's loop are answered from ONE whole-batch evaluation (SMALFitter._epoch_window):
    same window losses, same terms, same parameter gradients as the window-by-window evaluation - also when the windows are weighted
    differently or one is left out (the correction path of _EpochEval.backward), and a parameter update starts a new epoch."""
    from smilify_amd import synthetic

    t = tables("synthetic")
    weights, w_temp = [10.0, 500.0, 1.0, 1.0, 100.0, 0.1], 100.0
    n_win = (frames + window - 1) // window
    for ww in (None, [1.0 + 0.5 * k for k in range(n_win)], [0.0] + [1.0] * (n_win - 1)):
        a = synthetic.make_problem(t, frames, views, 40, DEV, radius=2.2, seed=5, window=window)
        b = synthetic.make_problem(t, frames, views, 40, DEV, radius=2.2, seed=5, window=window)
        b.epoch_cache = False
        opt_a = torch.optim.Adam(a.parameters(), lr=5e-3, betas=(0.5, 0.999))
        opt_b = torch.optim.Adam(b.parameters(), lr=5e-3, betas=(0.5, 0.999))
        for epoch in range(3):
            la, ta, acc_a, ga = _reference_epoch(a, frames, window, weights, w_temp, ww)
            lb, tb, acc_b, gb = _reference_epoch(b, frames, window, weights, w_temp, ww)
            if epoch > 0:
                assert a._epoch is not None and a._epoch["served"] == n_win, "the second epoch was not served from one evaluation"
            np.testing.assert_allclose(la, lb, rtol=2e-5)
            for da, db in zip(ta, tb):
                assert da.keys() == db.keys()
                np.testing.assert_allclose([da[k] for k in da], [db[k] for k in da], rtol=2e-5, atol=1e-7)
            for n in gb:
                assert (ga[n] is None) == (gb[n] is None), n
                if gb[n] is not None:
                    scale = float(gb[n].abs().max()) + 1e-12
                    assert float((ga[n] - gb[n]).abs().max()) <= 2e-4 * scale, (n, epoch, ww)
            opt_a.step()
            opt_b.step()
        np.testing.assert_allclose(a.trans.detach().cpu().numpy(), b.trans.detach().cpu().numpy(), atol=2e-6)
    # windows evaluated under no_grad (logging) must not be handed to a later differentiable call of the same parameter state
    with torch.no_grad():
        for j in range(0, frames, window):
            a(list(range(j, min(frames, j + window))), weights, 1)
    loss, _ = a(list(range(0, window)), weights, 1)
    loss2, _ = a(list(range(window, min(frames, 2 * window))), weights, 1)
    assert loss.requires_grad and loss2.requires_grad
    (loss.mean() + loss2.mean()).backward()


def test_generate_visualization_feeds_the_reference_exporter(tables):
    """The reference's driver calls ``model.generate_visualization(image_exporter)`` every VIS_FREQUENCY epochs
    (optimize_to_joints.py:177-178) and its ImageExporter pickles the parameter dict / writes the mesh it is handed
    (:48-68): every frame is exported once, with the dict ``export_parameters`` builds (= what ``load_checkpoint`` reads back),
    the posed vertices of its window and a (S, 5 S, 3) uint8 collage."""
    from smilify_amd import engine, synthetic

    t = tables("synthetic")
    N, W, S = 5, 2, 40
    model = synthetic.make_problem(t, N, 1, S, DEV, radius=2.2, seed=3, window=W)
    calls = []

    class Exporter:
        stage_id, epoch_name = 1, "0"

        def export(self, collage_np, batch_id, global_id, img_parameters, vertices, faces, img_idx=0, epoch=None):
            calls.append((collage_np, batch_id, global_id, img_parameters, vertices[batch_id].cpu().numpy().copy(), faces, img_idx, epoch))

    cams_before = model.renderer.cameras
    model.generate_visualization(Exporter(), epoch=7)
    assert model.renderer.cameras is cams_before
    assert [c[2] for c in calls] == list(range(N)) and [c[1] for c in calls] == [0, 1, 0, 1, 0] and all(c[7] == 7 for c in calls)
    lbs = engine.lbs_forward(model.device_model, model.betas.detach(), model._pose.detach().contiguous(), trans=model.trans.detach().contiguous(),
                             logscale=model.log_beta_scales.detach().contiguous(), btrans=model.betas_trans.detach().contiguous(), shared_beta=True,
                             trans_after_joints=True)
    verts = lbs["verts"].cpu().numpy()  # (the fit iteration's own posed, translated vertices)
    for collage, _, gid, params, v, faces, _, _ in calls:
        assert collage.shape == (S, 5 * S, 3) and collage.dtype == np.uint8 and collage[:, S:2 * S].max() > 0  # the render panel shows the mesh
        want = model.export_parameters(gid)
        assert params.keys() == want.keys()
        for k in want:
            np.testing.assert_array_equal(params[k], want[k])
        np.testing.assert_allclose(v, verts[gid], atol=2e-6)
        assert faces.shape == (t.F, 3)
    # ... and the reference-style epoch body still runs with the call inside it
    opt = torch.optim.Adam(model.parameters(), lr=5e-3, betas=(0.5, 0.999))
    for epoch in range(2):
        opt.zero_grad()
        acc = 0
        for j in range(0, N, W):
            loss, _ = model(list(range(j, min(N, j + W))), [10.0, 500.0, 1.0, 1.0, 100.0, 0.1], 1)
            acc += loss.mean()
        acc.backward()
        opt.step()
        if epoch % 1 == 0:
            model.generate_visualization(Exporter())
    assert len(calls) == 3 * N


def test_epoch_served_forward_with_a_joint_subset_and_frozen_parameters(tables):
    """``smil_window_terms`` and the cached epoch under the reference's stage-0 conditions: a subset of annotated joints
    (config.CANONICAL_MODEL_JOINTS), visibility masked in place, joint rotations / betas / scales frozen, zero silhouette weight
    (optimize_to_joints.py:129-138, column 0 of OPT_WEIGHTS) - window losses, terms and gradients as the window-by-window path."""
    from smilify_amd import synthetic

    t = tables("synthetic")
    N, W = 6, 3
    weights, w_temp = [25.0, 0.0, 0.0, 0.0, 0.0, 0.0], 500.0
    fits = []
    for cache in (True, False):
        f = synthetic.make_problem(t, N, 1, 40, DEV, radius=2.2, seed=9, window=W)
        f.epoch_cache = cache
        canon = [0, 2, 5, 7]
        f.config.CANONICAL_MODEL_JOINTS = canon
        f.target_joints = f.target_joints[:, canon].contiguous()
        f.target_visibility = f.target_visibility[:, canon].contiguous()
        f.target_visibility[:, 1] = 0  # in place, as the driver masks the non-torso joints
        for p in (f.joint_rotations, f.betas, f.log_beta_scales):
            p.requires_grad = False
        fits.append(f)
    a, b = fits
    for epoch in range(2):
        la, ta, _, ga = _reference_epoch(a, N, W, weights, w_temp)
        lb, tb, _, gb = _reference_epoch(b, N, W, weights, w_temp)
        np.testing.assert_allclose(la, lb, rtol=2e-5)
        for da, db in zip(ta, tb):
            assert da.keys() == db.keys() == {"joint"}
            np.testing.assert_allclose(da["joint"], db["joint"], rtol=2e-5)
        for n in gb:
            assert (ga[n] is None) == (gb[n] is None), n
            if gb[n] is not None:
                assert float((ga[n] - gb[n]).abs().max()) <= 2e-4 * (float(gb[n].abs().max()) + 1e-12), (n, epoch)
        with torch.no_grad():
            for f in (a, b):
                f.trans += 0.01  # an in-place parameter edit starts a new epoch for the cache (version counter)
    assert a._epoch is not None and a._epoch["served"] == N // W
