"""Pin the oracle's loss/step restatement against the real reference ``SMALFitter.forward``
(goldens made with the reference's own loss code; its renderer replaced by the oracle's because
pytorch3d is absent - so this pins the six loss terms, the parameter plumbing and all gradients
through LBS, not the rasteriser arithmetic)."""
import numpy as np
import pytest
import torch

from conftest import oracle_model
from oracle import fitter_ref

PARAMS = ["betas", "log_beta_scales", "betas_trans", "global_rotation", "trans", "joint_rotations", "fov"]


def _setup(g, tables_key, tables):
    m = oracle_model(tables(tables_key))
    params = {n: torch.from_numpy(g[f"param_{n}"]).clone().requires_grad_() for n in PARAMS}
    targets = dict(sil=torch.from_numpy(g["sil_target"]), joints=torch.from_numpy(g["target_joints"]),
                   visibility=torch.from_numpy(g["visibility"]))
    cams = dict(R=torch.from_numpy(g["R"]), T=torch.from_numpy(g["T"]))
    return m, params, targets, cams


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_fit_losses_match_reference(key, golden, tables):
    g = golden(f"fitter_{key}")
    m, params, targets, cams = _setup(g, key, tables)
    N = params["trans"].shape[0]
    total, objs, _ = fitter_ref.fit_losses(m, params, range(N), g["weights"], targets, cams, int(g["S"]),
                                           torch.from_numpy(g["mean_betas"]), torch.from_numpy(g["betas_prec"]))
    for k in fitter_ref.OBJ_KEYS:
        ref = float(g[f"obj_{k}"])
        assert abs(objs[k].item() - ref) <= 1e-4 * abs(ref) + 1e-7, (k, objs[k].item(), ref)
    assert abs(total.item() - float(g["loss"])) <= 1e-4 * abs(float(g["loss"]))
    jl, gl, tl = fitter_ref.temporal(params, 100.0)
    np.testing.assert_allclose([jl.item(), gl.item(), tl.item()], g["temporal"], rtol=1e-4, atol=1e-7)
    (total.mean() + jl + gl + tl).backward()
    for n in PARAMS:
        ref = g[f"grad_{n}"]
        got = params[n].grad.numpy()
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(got / scale, ref / scale, rtol=0, atol=5e-4, err_msg=n)


def test_init_rotation_and_shape_prior(golden, tables):
    g = golden("fitter_stick")
    np.testing.assert_allclose(fitter_ref.default_global_rotation(), g["init_global_rotation"], atol=1e-6)
    t = tables("stick")
    prec = fitter_ref.shape_prior_precision(t.shape_cov, t.nB)
    np.testing.assert_allclose(prec, g["betas_prec"], rtol=1e-5, atol=1e-6)
    gm = golden("fitter_mouse")
    tm = tables("mouse")
    np.testing.assert_allclose(fitter_ref.shape_prior_precision(tm.shape_cov, tm.nB), gm["betas_prec"], rtol=1e-5, atol=1e-6)
