"""Property-based checks of the host logic (hypothesis)."""
import numpy as np
from hypothesis import given, settings
from hypothesis import strategies as st

from smilify_amd import model_io, optimize


@settings(max_examples=200, deadline=None)
@given(n_total=st.integers(1, 5000), world=st.integers(1, 8), window=st.integers(1, 64))
def test_shard_plans_partition_the_sequence(n_total, world, window):
    w = max(1, min(window, n_total))
    n_win = (n_total + w - 1) // w
    if n_win < world:
        return
    plans = optimize.plan_shards(n_total, world, window)
    assert len(plans) == world and plans[0].start == 0 and plans[-1].stop == n_total
    covered = 0
    for a in plans:
        assert a.n_local > 0 and a.start == covered and (a.stop % w == 0 or a.stop == n_total)
        covered = a.stop
    sizes = [p.n_local for p in plans]
    assert max(sizes) - min(sizes) <= 2 * w


@settings(max_examples=25, deadline=None)
@given(seg=st.integers(3, 14), J=st.integers(3, 12), nB=st.integers(0, 6), seed=st.integers(0, 1000), static=st.booleans())
def test_synthetic_models_are_valid_and_round_trip_dense(seg, J, nB, seed, static):
    t = model_io.synthetic_model(V_side=seg, J=J, nB=nB, seed=seed, static_joints=static)
    model_io.validate_tables(t)
    W = t.dense_weights()
    assert np.allclose(W.sum(1), 1.0, atol=1e-5) and (np.count_nonzero(W, axis=1) <= model_io.MAX_BONES_PER_VERTEX).all()
    R = t.dense_J_regressor()
    assert np.allclose(R.sum(0), 1.0, atol=1e-5)
    colptr, rows, vals = t.jreg_csc()
    R2 = np.zeros_like(R)
    for v in range(t.V):
        R2[v, rows[colptr[v]:colptr[v + 1]]] = vals[colptr[v]:colptr[v + 1]]
    assert np.array_equal(R, R2)
    ptr, vid, w = t.bone_vertex_lists()
    W2 = np.zeros_like(W)
    for j in range(t.J):
        W2[vid[ptr[j]:ptr[j + 1]], j] = w[ptr[j]:ptr[j + 1]]
    assert np.array_equal(W, W2)
    # closed, consistently wound surface: every edge is used by exactly two faces, once in each direction
    e = np.concatenate([t.faces[:, [0, 1]], t.faces[:, [1, 2]], t.faces[:, [2, 0]]])
    fwd = {(int(a), int(b)) for a, b in e}
    assert len(fwd) == len(e) and all((b, a) in fwd for a, b in fwd)
