"""Fit-result interchange formats on the CPU: the AMASS-style animation clip and the per-frame
parameter pickle, against files the REAL reference code wrote (tests/golden/make_golden.py::export_golden).

The recorder tests mirror the reference's own round-trip tests (tests/test_animation_export.py)."""
import json
import os
import pickle

import numpy as np
import pytest
import torch

from conftest import GOLDEN
from smilify_amd.animation_export import SCHEMA_VERSION, AnimationRecorder, rotation_6d_to_axis_angle

N_JOINTS, N_BETAS = 8, 20
N_POSE = N_JOINTS - 1


def _params(seed):
    g = torch.Generator().manual_seed(seed)
    return {"global_rot": torch.randn(1, 3, generator=g), "joint_rot": torch.randn(1, N_POSE, 3, generator=g),
            "trans": torch.randn(1, 3, generator=g), "betas": torch.randn(1, N_BETAS, generator=g),
            "log_beta_scales": torch.randn(1, N_JOINTS, 3, generator=g), "betas_trans": torch.randn(1, N_JOINTS, 3, generator=g),
            "mesh_scale": torch.rand(1, 1, generator=g) + 0.5, "cam_rot": torch.eye(3).unsqueeze(0),
            "cam_trans": torch.randn(1, 3, generator=g), "fov": torch.tensor([[45.0]])}


def _recorder(tmp_path, rep="axis_angle"):
    return AnimationRecorder(output_path=tmp_path / "clip", rotation_representation=rep, n_joints=N_JOINTS, n_betas=N_BETAS,
                             joint_names=[f"J_{i}" for i in range(N_JOINTS)], parents=[-1] + list(range(N_JOINTS - 1)), fps=30.0,
                             static_joint_locs=True, ignore_hardcoded_body=False, source_checkpoint="ckpt.pth", source_input="input.mp4",
                             model_id="test_model")


def test_round_trip_axis_angle(tmp_path):
    rec = _recorder(tmp_path)
    frames = [_params(i) for i in range(5)]
    for f in frames:
        rec.record(f)
    assert rec.num_frames() == 5
    out = rec.write()
    data = np.load(out["npz"])
    assert data["poses"].shape == (5, N_JOINTS, 3) and data["trans"].shape == (5, 3) and data["betas"].shape == (N_BETAS,)
    assert data["betas_per_frame"].shape == (5, N_BETAS) and data["log_beta_scales"].shape == (5, N_JOINTS, 3)
    assert data["betas_trans"].shape == (5, N_JOINTS, 3) and data["mesh_scale"].shape == (5,)
    assert float(data["fps"]) == pytest.approx(30.0)
    for i, p in enumerate(frames):
        np.testing.assert_allclose(data["poses"][i, 0], p["global_rot"][0].numpy(), atol=1e-6)
        np.testing.assert_allclose(data["poses"][i, 1:], p["joint_rot"][0].numpy(), atol=1e-6)
        np.testing.assert_allclose(data["betas_per_frame"][i], p["betas"][0].numpy(), atol=1e-6)
    np.testing.assert_allclose(data["betas"], np.stack([p["betas"][0].numpy() for p in frames]).mean(axis=0), atol=1e-6)
    sidecar = json.load(open(out["json"]))
    assert sidecar["schema_version"] == SCHEMA_VERSION == "1.1" and sidecar["rotation_representation"] == "axis_angle"
    assert sidecar["root_joint_index"] == 0 and sidecar["static_joint_locs"] is True and sidecar["n_frames"] == 5
    assert sidecar["n_joints"] == N_JOINTS and sidecar["n_betas"] == N_BETAS and sidecar["fps"] == 30.0
    assert len(sidecar["joint_names"]) == N_JOINTS and len(sidecar["parents"]) == N_JOINTS
    assert len(sidecar["cameras"]) == 1 and sidecar["cameras"][0]["view_name"] == "view_0"


def test_6d_rotations_normalise_to_axis_angle(tmp_path):
    rec = _recorder(tmp_path, "6d")
    ident = torch.tensor([[1.0, 0.0, 0.0, 0.0, 1.0, 0.0]])
    rec.record({"global_rot": ident, "joint_rot": ident.expand(1, N_POSE, 6).clone(), "trans": torch.zeros(1, 3), "betas": torch.zeros(1, N_BETAS)})
    data = np.load(rec.write()["npz"])
    np.testing.assert_allclose(data["poses"], np.zeros((1, N_JOINTS, 3)), atol=1e-5)
    # general rotations: 6-D of Rodrigues(aa) -> aa back (the two rows of the matrix are the representation)
    g = torch.Generator().manual_seed(0)
    aa = torch.randn(64, 3, generator=g)
    aa = aa / aa.norm(dim=1, keepdim=True) * (torch.rand(64, 1, generator=g) * 3.0 + 0.05)
    aa[0] = torch.tensor([3.14159, 0.0, 0.0])   # half turn: the logarithm's singular case
    from oracle import lbs_ref
    R = lbs_ref.rodrigues(aa)
    back = rotation_6d_to_axis_angle(torch.cat([R[:, 0], R[:, 1]], dim=1))
    np.testing.assert_allclose(lbs_ref.rodrigues(back).numpy(), R.numpy(), atol=2e-5)
    np.testing.assert_allclose(back[1:].numpy(), aa[1:].numpy(), atol=2e-4)


def test_optional_fields_absent_and_errors(tmp_path):
    rec = _recorder(tmp_path)
    with pytest.raises(RuntimeError):
        rec.write()
    rec.record({"global_rot": torch.zeros(1, 3), "joint_rot": torch.zeros(1, N_POSE, 3), "trans": torch.zeros(1, 3), "betas": torch.zeros(1, N_BETAS)})
    out = rec.write()
    data = np.load(out["npz"])
    assert not {"log_beta_scales", "betas_trans", "mesh_scale"} & set(data.files)
    assert json.load(open(out["json"]))["cameras"] == []
    with pytest.raises(ValueError):
        _recorder(tmp_path, "quaternion")


def test_recorder_reproduces_the_reference_writer(tmp_path):
    """Same inputs as the file the reference's AnimationRecorder wrote: identical payload (keys, dtypes, shapes, values) and
    identical side-car."""
    inp = np.load(os.path.join(GOLDEN, "animation_ref_inputs.npz"))
    want = np.load(os.path.join(GOLDEN, "animation_ref.npz"))
    want_json = json.load(open(os.path.join(GOLDEN, "animation_ref.json")))
    F, nJ, nB = inp["global_rot"].shape[0], inp["log_beta_scales"].shape[2], inp["betas"].shape[2]
    rec = AnimationRecorder(output_path=tmp_path / "animation_ref", rotation_representation="axis_angle", n_joints=nJ, n_betas=nB,
                            joint_names=[f"J_{i}" for i in range(nJ)], parents=[-1] + list(range(nJ - 1)), fps=25.0, static_joint_locs=True,
                            ignore_hardcoded_body=True, source_checkpoint="ckpt.pth", source_input="clip.mp4", model_id="golden")
    for i in range(F):
        rec.record({k: torch.from_numpy(inp[k][i]) for k in inp.files})
    out = rec.write()
    got = np.load(out["npz"])
    assert sorted(got.files) == sorted(want.files)
    for k in want.files:
        assert got[k].dtype == want[k].dtype and got[k].shape == want[k].shape, k
        np.testing.assert_array_equal(got[k], want[k], err_msg=k)
    got_json = json.load(open(out["json"]))
    assert list(got_json) == list(want_json)  # same keys in the same order
    cams_g, cams_w = got_json.pop("cameras"), want_json.pop("cameras")
    assert got_json == want_json
    assert len(cams_g) == len(cams_w) == 1 and cams_g[0]["view_name"] == cams_w[0]["view_name"]
    np.testing.assert_allclose(cams_g[0]["R"], cams_w[0]["R"], rtol=1e-6)
    np.testing.assert_allclose(cams_g[0]["t"], cams_w[0]["t"], rtol=1e-6)
    assert cams_g[0]["fov"] == pytest.approx(cams_w[0]["fov"], rel=1e-6)


def test_reference_checkpoint_fixture_layout():
    """The per-frame pickle the reference writes (optimize_to_joints.py:48-63): key set and shapes the loader relies on."""
    root = os.path.join(GOLDEN, "checkpoint_ref")
    exp = np.load(os.path.join(root, "expected.npz"))
    J = exp["joint_rotations"].shape[1] + 1
    for frame in range(exp["trans"].shape[0]):
        p = pickle.load(open(os.path.join(root, f"{frame:04}", "st1_ep7.pkl"), "rb"))
        assert sorted(p) == ["betas", "betas_trans", "fov", "global_rotation", "joint_rotations", "log_betascale", "trans"]
        assert p["global_rotation"].shape == (3,) and p["joint_rotations"].shape == (J - 1, 3) and p["trans"].shape == (3,)
        assert p["log_betascale"].shape == (J, 3) and p["betas_trans"].shape == (J, 3) and p["fov"].shape == ()
        np.testing.assert_array_equal(p["global_rotation"], exp["global_rotation"][frame])  # what the reference loader restored


def test_record_fitter_copies_whole_parameter_buffers(tmp_path):
    """``record_fitter`` on a fitter-shaped object: every track arrives in one block, masks applied, shared tables expanded,
    one static camera per view in the side-car; the same clip comes out of the frame-by-frame path."""
    from types import SimpleNamespace

    from smilify_amd.animation_export import record_fitter

    N, J, nB, views = 7, 5, 3, 2
    g = torch.Generator().manual_seed(3)
    rnd = lambda *s: torch.randn(*s, generator=g)  # noqa: E731
    gmask, rmask = torch.tensor([1.0, 0.0, 1.0]), torch.ones(J - 1, 3)
    rmask[2] = 0.0
    fitter = SimpleNamespace(
        smal_model=SimpleNamespace(tables=SimpleNamespace(J=J, nB=nB, joint_names=[f"j{i}" for i in range(J)], parents=[-1, 0, 1, 1, 3],
                                                           static_joints=False, name="toy")),
        config=SimpleNamespace(ignore_hardcoded_body=True), num_images=N, views=views,
        global_rotation=rnd(N, 3), joint_rotations=rnd(N, J - 1, 3), global_mask=gmask, rotation_mask=rmask, trans=rnd(N, 3),
        betas=rnd(nB), log_beta_scales=rnd(1, J, 3), betas_trans=rnd(N, J, 3), fov=torch.tensor([50.0, 70.0]),
        renderer=SimpleNamespace(cameras=SimpleNamespace(R=rnd(views, 3, 3), T=rnd(views, 3))))
    out = record_fitter(fitter, tmp_path / "fit", fps=12.0, view_names=["left", "right"])
    d, side = np.load(out["npz"]), json.load(open(out["json"]))
    assert d["poses"].shape == (N, J, 3) and d["betas_per_frame"].shape == (N, nB) and d["log_beta_scales"].shape == (N, J, 3)
    np.testing.assert_array_equal(d["poses"][:, 0], (fitter.global_rotation * gmask).numpy())
    np.testing.assert_array_equal(d["poses"][:, 1:], (fitter.joint_rotations * rmask).numpy())
    np.testing.assert_array_equal(d["log_beta_scales"], fitter.log_beta_scales.expand(N, J, 3).numpy())
    np.testing.assert_array_equal(d["betas"], d["betas_per_frame"].mean(axis=0))
    assert side["n_frames"] == N and side["model_id"] == "toy" and [c["view_name"] for c in side["cameras"]] == ["left", "right"]
    assert side["cameras"][1]["fov"] == 70.0 and "mesh_scale" not in d.files
    # frame by frame through record(): the identical payload
    rec = AnimationRecorder(tmp_path / "fit2", "axis_angle", J, nB, side["joint_names"], side["parents"], 12.0, False, True)
    for i in range(N):
        rec.record({"global_rot": (fitter.global_rotation * gmask)[i:i + 1], "joint_rot": (fitter.joint_rotations * rmask)[i:i + 1],
                    "trans": fitter.trans[i:i + 1], "betas": fitter.betas[None], "log_beta_scales": fitter.log_beta_scales,
                    "betas_trans": fitter.betas_trans[i:i + 1]})
    d2 = np.load(rec.write()["npz"])
    assert sorted(d2.files) == sorted(d.files)
    for k in d.files:
        np.testing.assert_array_equal(d2[k], d[k], err_msg=k)
    with pytest.raises(KeyError):
        rec.record_block(poses=np.zeros((1, J, 3)), trans=np.zeros((1, 3)))  # no betas
