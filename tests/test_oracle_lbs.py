"""Pin the oracle's LBS restatement against vectors produced by the real reference
(tests/golden/make_golden.py, run in the build container)."""
import numpy as np
import pytest
import torch

from conftest import oracle_model, vertex_probe
from oracle import lbs_ref

TOL = dict(rtol=2e-5, atol=2e-6)


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_rodrigues_matches_reference(key, golden):
    g = golden(f"lbs_{key}")
    R = lbs_ref.rodrigues(torch.from_numpy(g["rod_theta"])).numpy()
    np.testing.assert_allclose(R, g["rod_R"], **TOL)
    # quirk 1: theta == 0 -> identity exactly
    np.testing.assert_array_equal(R[0], np.eye(3, dtype=np.float32))


@pytest.mark.parametrize("key", ["stick", "mouse"])
@pytest.mark.parametrize("tag", ["plain", "scale", "scale_trans", "prop"])
def test_chain_matches_reference(key, tag, golden, tables):
    g = golden(f"lbs_{key}")
    t = tables(key)
    kw = {}
    if tag != "plain":
        kw["betas_logscale"] = torch.from_numpy(g["chain_ls"])
    if tag in ("scale_trans", "prop"):
        kw["betas_trans"] = torch.from_numpy(g["chain_bt"])
    if tag == "prop":
        kw["propagate_scaling"] = True
    nj, A, _ = lbs_ref.global_rigid_transformation(torch.from_numpy(g["chain_Rs"]), torch.from_numpy(g["chain_Js"]),
                                                    t.parents, **kw)
    np.testing.assert_allclose(nj.numpy(), g[f"chain_{tag}_newJ"], rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(A.numpy(), g[f"chain_{tag}_A"], rtol=1e-4, atol=1e-5)


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_smal_forward_and_grads_match_reference(key, golden, tables):
    g = golden(f"lbs_{key}")
    m = oracle_model(tables(key))
    leaf = {n: torch.from_numpy(g[f"smal_{n}"]).clone().requires_grad_() for n in ["beta", "theta", "trans", "ls", "bt"]}
    out = lbs_ref.smal_forward(m, leaf["beta"], leaf["theta"], trans=leaf["trans"], betas_logscale=leaf["ls"],
                               betas_trans=leaf["bt"])
    np.testing.assert_allclose(out["verts"].detach().numpy(), g["smal_verts"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["joints"].detach().numpy(), g["smal_joints"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["Rs"].detach().numpy(), g["smal_Rs"], **TOL)
    np.testing.assert_allclose(out["v_shaped"].detach().numpy(), g["smal_v_shaped"], **TOL)
    np.testing.assert_allclose(out["new_J"].detach().numpy(), g["smal_J_transformed"], rtol=1e-4, atol=2e-6)
    loss = (out["verts"] * vertex_probe(out["verts"].shape, 0)).sum() + (out["joints"] * vertex_probe(out["joints"].shape, 1)).sum()
    assert abs(loss.item() - float(g["smal_loss"])) < 1e-3 * max(1.0, abs(float(g["smal_loss"])))
    loss.backward()
    for n, t in leaf.items():
        ref = g[f"smal_grad_{n}"]
        scale = np.abs(ref).max() + 1e-12
        np.testing.assert_allclose(t.grad.numpy() / scale, ref / scale, rtol=0, atol=2e-4, err_msg=n)


def test_reference_test_fixture(golden, tables):
    """tests/test_triangulation_consistency.py:209-216 of the reference: seed 42, betas 0."""
    g = golden("lbs_stick")
    t = tables("stick")
    m = oracle_model(t)
    out = lbs_ref.smal_forward(m, torch.zeros(2, t.nB), torch.from_numpy(g["fixture_theta"]))
    np.testing.assert_allclose(out["joints"].numpy(), g["fixture_joints"], rtol=1e-4, atol=2e-6)


@pytest.mark.parametrize("key", ["stick", "mouse"])
def test_plain_call_shared_betas(key, golden, tables):
    g = golden(f"lbs_{key}")
    m = oracle_model(tables(key))
    beta = torch.from_numpy(g["smal_beta"][:1]).expand(2, -1)
    out = lbs_ref.smal_forward(m, beta, torch.from_numpy(g["smal_theta"][1:3]))
    np.testing.assert_allclose(out["verts"].numpy(), g["plain_verts"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["joints"].numpy(), g["plain_joints"], rtol=1e-4, atol=2e-6)


def test_pose_blend_shapes_match_reference(golden):
    """posedirs path (smal_torch.py:294-301): the real reference SMAL loaded a pickled synthetic model and was given
    a random posedirs table (no shipped SMIL model has one)."""
    from smilify_amd import model_io

    g = golden("lbs_posedirs")
    t = model_io.synthetic_model(seed=int(g["seed"]))
    m = oracle_model(t)
    m["posedirs"] = torch.from_numpy(g["posedirs"])
    leaves = {n: torch.from_numpy(g[n]).clone().requires_grad_() for n in ("beta", "theta", "trans")}
    out = lbs_ref.smal_forward(m, leaves["beta"], leaves["theta"], trans=leaves["trans"])
    np.testing.assert_allclose(out["verts"].detach().numpy(), g["verts"], rtol=1e-4, atol=2e-6)
    np.testing.assert_allclose(out["joints"].detach().numpy(), g["joints"], rtol=1e-4, atol=2e-6)
    ((out["verts"] * vertex_probe(out["verts"].shape, 0)).sum() + (out["joints"] * vertex_probe(out["joints"].shape, 1)).sum()).backward()
    for n in leaves:
        ref = g[f"grad_{n}"]
        sc = np.abs(ref).max()
        np.testing.assert_allclose(leaves[n].grad.numpy() / sc, ref / sc, atol=2e-4, err_msg=n)
