"""Model-file ingest, camera ingest and configuration - CPU only."""
import io
import math
import pickle

import numpy as np
import pytest
import scipy.sparse as sp
import torch

from conftest import MODEL_FILES
from oracle import render_ref
from smilify_amd import cameras, config, model_io


def _model_dict(t, sparse_regressor=False, static=False):
    """A pickle-schema dict (3D_model_prep/SMIL_processing_addon.py:1590-1603) rebuilt from tables."""
    JR = t.dense_J_regressor().T.copy()
    dd = dict(f=t.faces.astype(np.int32), J_regressor=sp.csc_matrix(JR) if sparse_regressor else JR,
              kintree_table=np.stack([np.where(t.parents < 0, 4294967295, t.parents).astype(np.int64), np.arange(t.J)]),
              J=np.zeros((t.J, 3), np.float32) if t.J_static is None else t.J_static, weights=t.dense_weights(),
              posedirs=np.zeros(0), v_template=t.v_template.astype(np.float64),
              shapedirs=t.shapedirs.T.reshape(t.V, 3, t.nB).astype(np.float64), J_names=list(t.joint_names), bs_style="lbs")
    if static:
        dd["static_joint_locs"] = True
    return dd


def test_pickle_roundtrip_matches_tables(tmp_path):
    t = model_io.synthetic_model()
    for sparse in (False, True):
        p = tmp_path / f"m{int(sparse)}.pkl"
        with open(p, "wb") as fh:
            pickle.dump(_model_dict(t, sparse_regressor=sparse), fh, protocol=2)
        t2 = model_io.load_model(str(p))
        np.testing.assert_allclose(t2.v_template, t.v_template)
        np.testing.assert_allclose(t2.shapedirs, t.shapedirs, atol=1e-7)
        np.testing.assert_array_equal(t2.faces, t.faces)
        np.testing.assert_array_equal(t2.parents, t.parents)       # 2^32-1 root marker of legacy files -> -1
        np.testing.assert_allclose(t2.dense_weights(), t.dense_weights())
        np.testing.assert_allclose(t2.dense_J_regressor(), t.dense_J_regressor())
        np.testing.assert_array_equal(t2.depth, t.depth)
    # npz round trip
    t.save_npz(str(tmp_path / "m.npz"))
    t3 = model_io.load_model(str(tmp_path / "m.npz"))
    np.testing.assert_array_equal(t3.skin_idx, t.skin_idx)
    np.testing.assert_allclose(t3.skin_w, t.skin_w)
    assert t3.name == t.name and t3.joint_names == t.joint_names


def test_unpickler_refuses_foreign_globals(tmp_path):
    class Evil:
        def __reduce__(self):
            import os
            return (os.system, ("echo pwned",))

    p = tmp_path / "evil.pkl"
    with open(p, "wb") as fh:
        pickle.dump({"f": Evil()}, fh)
    with pytest.raises(pickle.UnpicklingError):
        model_io.read_model_pickle(str(p))


def test_validator_rejects_malformed_models():
    t = model_io.synthetic_model()
    dd = _model_dict(t)
    bad = dict(dd)
    W = dd["weights"].copy()
    W[0, :5] = 0.2                                  # five bones on one vertex
    bad["weights"] = W
    with pytest.raises(ValueError, match="bones"):
        model_io.tables_from_dict(bad)
    bad = dict(dd)
    kt = dd["kintree_table"].copy()
    kt[0, 2] = 5                                    # parent after child
    bad["kintree_table"] = kt
    with pytest.raises(ValueError, match="precede"):
        model_io.tables_from_dict(bad)
    bad = dict(dd)
    f = dd["f"].copy()
    f[0, 0] = t.V + 3
    bad["f"] = f
    with pytest.raises(ValueError, match="face"):
        model_io.tables_from_dict(bad)


def test_shipped_tables_are_consistent():
    for key, path in MODEL_FILES.items():
        t = model_io.load_model(path)
        assert np.allclose(t.dense_weights().sum(1), 1.0, atol=1e-5)
        assert (t.skin_w >= 0).all() and t.depth.max() <= 8
        assert t.static_joints == (key == "mouse")
        colptr, rows, vals = t.jreg_csc()
        assert colptr[-1] == len(t.jreg_col) == len(rows) == len(vals)
        ptr, vid, w = t.bone_vertex_lists()
        assert ptr[-1] == int((t.skin_w != 0).sum())
        cfg = config.FitterConfig.from_tables(t)
        assert cfg.N_POSE == t.J - 1 and cfg.N_BETAS == t.nB and cfg.CANONICAL_MODEL_JOINTS == list(range(t.J))
        assert cfg.STATIC_JOINT_LOCATIONS == t.static_joints


def test_look_at_matches_oracle_restatement():
    for args in [(2.7, 0.0, 0.0), (3.0, 15.0, np.array([0.0, 60.0, 120.0, 180.0, 240.0, 300.0])), (4.0, -20.0, 45.0), (2.0, 90.0, 0.0)]:
        R, T = cameras.look_at_view_transform(*args)
        Ro, To = render_ref.look_at_view_transform(args[0], args[1], torch.as_tensor(args[2], dtype=torch.float32))
        np.testing.assert_allclose(R.numpy(), Ro.numpy(), atol=2e-6)
        np.testing.assert_allclose(T.numpy(), To.numpy(), atol=2e-6)


def test_opencv_camera_conversion_reproduces_pinhole():
    """reference sleap_multiview_dataset.py:197-223: the FoV camera must project like u = fx X/Z + W/2, v = fy Y/Z + H/2."""
    rng = np.random.default_rng(0)
    W = H = 512
    fx, fy = 800.0, 760.0
    K = np.array([[fx, 0, W / 2], [0, fy, H / 2], [0, 0, 1.0]])
    a = 0.3
    R_cv = np.array([[math.cos(a), 0, math.sin(a)], [0, 1, 0], [-math.sin(a), 0, math.cos(a)]]) @ np.array(
        [[1, 0, 0], [0, math.cos(0.2), -math.sin(0.2)], [0, math.sin(0.2), math.cos(0.2)]])
    t_cv = np.array([0.1, -0.2, 3.0])
    R, T, fov, aspect = cameras.opencv_to_fov_camera(R_cv, t_cv, K, (W, H))
    X = rng.uniform(-0.5, 0.5, (20, 3))
    Xc = X @ R_cv.T + t_cv
    u, v = fx * Xc[:, 0] / Xc[:, 2] + W / 2, fy * Xc[:, 1] / Xc[:, 2] + H / 2
    yx = render_ref.project_points_screen(torch.tensor(X[None], dtype=torch.float32), torch.tensor(R[None]), torch.tensor(T[None]),
                                          torch.tensor([fov]), H, torch.tensor([aspect]))[0].numpy()
    np.testing.assert_allclose(yx[:, 0], v, atol=2e-2)
    np.testing.assert_allclose(yx[:, 1], u, atol=2e-2)
