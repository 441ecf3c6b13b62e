"""Round 4 GPU tests (run on the MI355X box with ``pytest -m gpu``)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda:0"


@pytest.mark.parametrize("seed,wide", [(s, False) for s in (0, 1, 2, 3, 5, 8, 13, 21)] + [(s, True) for s in (100, 101, 102, 103)])
def test_fused_lbs_kernels_against_the_oracle(seed, wide):
    """smil_lbs_forward_project / smil_lbs_backward_ndc directly against the CPU oracle's autograd through LBS and projection
    (oracle/lbs_ref.py, oracle/render_ref.py; reference smal_model/smal_torch.py:240-351, batch_lbs.py:155-195): seeded random
    models with up to 120 (wide: 250) joints, up to 20 views, static and regressed joints, shared and per-frame betas.
    Tolerances (tests/lbs_cases.py): forward 2e-5, parameter gradients 5e-4 of the largest component; fused against separate
    kernels 1e-6 / 3e-5."""
    import lbs_cases

    checks, info = lbs_cases.run_case(seed, wide)
    assert any(w.startswith("oracle d_") for _, w, _ in checks), info
    fails = [(w, e) for f, w, e in checks if f is not None]
    assert not fails, (info, fails)


@pytest.mark.parametrize("key,frames,views", [("stick", 96, 2), ("mouse", 24, 3)])
def test_the_shared_shape_gradient_is_bit_reproducible(key, frames, views, tables):
    """Two evaluations of the same fit step return the same bits in d_betas - the one quantity ranks all-reduce (reference
    fitter.py:236-335: ``betas`` is shared by every frame, so its gradient is a sum over frames).  Round 3 summed the frames with
    float atomics, whose result depends on the order the blocks arrive in; since round 4 every block leaves a partial row and the
    last block adds the rows in a fixed order (lbs.hip BetaSum).  STICK goes through the fused per-frame kernel + chain kernel, the
    mouse through the separate shape kernel; several runs, because a race would only show now and then.  (At least 64 images per
    launch, so that the rasteriser's vertex gradients upstream are the integer-exact packed ones.)"""
    from smilify_amd import synthetic

    f = synthetic.make_problem(tables(key), frames, views, 64, DEV, radius=2.7 if key == "stick" else 4.0)
    ref_objs, ref = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL, window=10)
    ref = {k: v.clone() for k, v in ref.items() if v is not None}
    ref_objs = ref_objs.clone()
    assert float(ref["betas"].abs().max()) > 0
    for _ in range(5):
        objs, g = f._loss_and_grads(None, synthetic.STAGE1_WEIGHTS, synthetic.STAGE1_TEMPORAL, window=10)
        assert torch.equal(g["betas"], ref["betas"]), (g["betas"], ref["betas"])
        # the shared scale / translation tables are sums in a fixed order as well (smil_reduce_rows)
        for k in ("log_beta_scales", "betas_trans"):
            if k in ref and g.get(k) is not None:
                assert torch.equal(g[k], ref[k]), k
        # (the fov gradient and the loss terms still end in a few float atomics per image / per block: equal to rounding)
        np.testing.assert_allclose(g["fov"].cpu().numpy(), ref["fov"].cpu().numpy(), rtol=1e-5)
        np.testing.assert_allclose(objs.cpu().numpy(), ref_objs.cpu().numpy(), rtol=1e-5, atol=1e-7)
