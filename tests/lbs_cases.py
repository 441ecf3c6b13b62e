"""Random procedural models and random calls through the fused per-frame LBS kernels (smil_lbs_forward_project,
smil_lbs_backward_ndc), checked DIRECTLY against the CPU oracle's autograd (oracle/lbs_ref.py + oracle/render_ref.py) and
against the separate-kernel route.  Shared by tests/test_gpu_lbs_fused.py (a dozen seeded cases under ``pytest -m gpu``) and
tools/dbg/fuzz_lbs.py (as many seeds as one likes).

Models: procedural tubes with 3 ... 120 joints (``wide``: up to 250 joints and 20 views), up to 4 bones per vertex, vertices that
several joints regress from, 0 ... 9 shape coefficients, static or regressed joints.  Reference: smal_model/smal_torch.py:240-351,
smal_model/batch_lbs.py:155-195 (through the oracle's restatement)."""
import numpy as np
import torch

from conftest import oracle_model
from oracle import lbs_ref, render_ref
from smilify_amd import cameras as cam_mod
from smilify_amd import engine as eng
from smilify_amd import model_io

DEV = "cuda:0"


def random_model(rng, wide=False, big=False):
    J = int(rng.integers(3, 251 if wide else 121))
    side = int(rng.integers(4, 41))
    while (J + 1) * side + 2 > 5000:
        side -= 1
    if big:  # more than 3 600 vertices: beyond the 80 KB of LDS two workgroups per CU can take each (24 bytes per vertex)
        J = int(rng.integers(90, 121))
        side = int(rng.integers(3600 // (J + 1) + 1, 4998 // (J + 1) + 1))
    nB = int(rng.integers(0, 10))
    static = bool(rng.integers(0, 2))
    t = model_io.synthetic_model(V_side=side, J=J, nB=nB, seed=int(rng.integers(0, 1 << 30)), static_joints=static)
    V = t.v_template.shape[0]
    # up to four bones on a third of the vertices
    idx, w = t.skin_idx.copy(), t.skin_w.copy()
    pick = rng.random(V) < 0.33
    for v in np.nonzero(pick)[0]:
        k = int(rng.integers(3, 5))
        extra = rng.choice(J, size=k, replace=False) if J >= k else np.arange(J)
        ww = rng.random(len(extra)).astype(np.float32) + 0.05
        idx[v] = 0
        w[v] = 0
        idx[v, :len(extra)] = extra
        w[v, :len(extra)] = ww / ww.sum()
    t.skin_idx, t.skin_w = idx, w
    if not static:  # a few vertices that two or three joints regress from
        rowptr, col, val = t.jreg_rowptr, t.jreg_col, t.jreg_val
        cols, vals, rp = [], [], [0]
        hot = rng.choice(V, size=min(8, V), replace=False)
        for j in range(J):
            c = list(col[rowptr[j]:rowptr[j + 1]])
            x = list(val[rowptr[j]:rowptr[j + 1]])
            if rng.random() < 0.5:
                for h in hot[: int(rng.integers(1, 4))]:
                    if h not in c:
                        c.append(int(h))
                        x.append(float(rng.random() * 0.1))
            s = sum(x)
            cols += c
            vals += [q / s for q in x]
            rp.append(len(cols))
        t.jreg_rowptr, t.jreg_col, t.jreg_val = np.array(rp, np.int32), np.array(cols, np.int32), np.array(vals, np.float32)
    model_io.validate_tables(t)
    return t


def close(a, b, tol, what):
    a, b = a.detach().cpu().numpy(), b.detach().cpu().numpy()
    err = np.abs(a - b).max() / (np.abs(b).max() + 1e-30)
    return err if err >= tol else None, what, err



def run_case(seed, wide=False, big=False, table=None):
    """One random model (or ``table``) and call.  Returns (checks, info): checks = [(failed_error_or_None, what, error)]."""
    rng = np.random.default_rng(seed)
    t = random_model(rng, wide, big) if table is None else table
    dm = eng.DeviceModel(t, DEV)
    J, V, nB = dm.J, dm.V, dm.nB
    B, views, S = int(rng.integers(1, 40)), int(rng.integers(1, 21 if wide else 7)), 64
    shared_beta, trans_after = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    ls_shared, use_mask = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
    g = torch.Generator().manual_seed(seed)
    host = dict(beta=0.4 * torch.randn(*((nB,) if shared_beta else (B, nB)), generator=g), theta=0.25 * torch.randn(B, J, 3, generator=g),
                trans=0.1 * torch.randn(B, 3, generator=g), ls=0.05 * torch.randn(*((J, 3) if ls_shared else (B, J, 3)), generator=g),
                bt=0.02 * torch.randn(J, 3, generator=g))
    dev = {k: v.to(DEV) for k, v in host.items()}
    R, T = cam_mod.look_at_view_transform(3.0, 12.0, np.linspace(0, 300, views), device=DEV)
    cams = eng.CameraSet(R.contiguous(), T.contiguous(), torch.full((views,), 52.0, device=DEV), None, views, S)
    kw = dict(trans=dev["trans"], logscale=dev["ls"], btrans=dev["bt"], shared_beta=shared_beta, logscale_shared=ls_shared, btrans_shared=True,
              trans_after_joints=trans_after)
    N = B * views
    d_ndc = (1e-3 * torch.randn(N, V, 2, generator=g)).to(DEV)
    d_yx = (1e-2 * torch.randn(N, J, 2, generator=g)).to(DEV)
    ok_sup = eng.lbs_backward_ndc_supported(dm, nB, views)
    # --- HIP, fused and separate
    ref = eng.lbs_forward(dm, dev["beta"], dev["theta"], **kw)
    ndc_ref, yx_ref = eng.project_verts_and_joints(cams, ref["verts"], ref["joints"])
    got = eng.lbs_forward(dm, dev["beta"], dev["theta"], project=dict(cams=cams, ndc=True, yx=True), **kw)
    checks = [close(got[k], ref[k], 1e-6, "fwd " + k) for k in ("verts", "joints")]
    checks += [close(got["ndc"], ndc_ref, 1e-6, "fwd ndc"), close(got["yx"], yx_ref, 1e-6, "fwd yx")]
    fov_b = torch.zeros(N, device=DEV)
    dv, dj = eng.project_backward_verts_and_joints(cams, ref["verts"], d_ndc, ref["joints"], d_yx, fov_b)
    b = eng.lbs_backward(dm, ref, dv, dj)
    if ok_sup:
        fov_a = torch.zeros(N, device=DEV)
        a = eng.lbs_backward(dm, got, None, None, ndc_upstream=dict(cams=cams, d_ndc=d_ndc, d_yx=d_yx, d_fov_img=fov_a))
        for k in ("d_beta", "d_theta", "d_trans", "d_logscale", "d_btrans"):
            if b[k] is not None and b[k].numel():
                # (the shape gradient is ONE sum over all vertices and frames per coefficient, of terms of both signs: with a single
                # coefficient the "largest component" is that cancelling sum itself, and two fp32 summation orders differ by up to 1e-4
                # of it - long fuzz seed 12025: 7e-5 between the routes, 4.5e-4 against the oracle, every other gradient 1e-7)
                checks.append(close(a[k], b[k], 2e-4 if k == "d_beta" else 3e-5, "bwd " + k))
        checks.append(close(fov_a, fov_b, 3e-5, "bwd fov"))
    # --- the oracle's autograd through LBS + projection on the same upstream gradients
    m = oracle_model(t)
    leaves = {k: v.clone().requires_grad_() for k, v in host.items()}
    beta_o = leaves["beta"][None].expand(B, -1) if shared_beta else leaves["beta"]
    ls_o = leaves["ls"][None].expand(B, -1, -1) if ls_shared else leaves["ls"]
    o = lbs_ref.smal_forward(m, beta_o, leaves["theta"], trans=None if trans_after else leaves["trans"], betas_logscale=ls_o,
                             betas_trans=leaves["bt"][None].expand(B, -1, -1))
    vo, jo = o["verts"], o["joints"]
    if trans_after:
        vo, jo = vo + leaves["trans"][:, None], jo + leaves["trans"][:, None]
    Rc, Tc = R.cpu(), T.cpu()
    rep = lambda x: x[:, None].expand(-1, views, -1, -1).reshape(N, -1, 3)  # noqa: E731
    ndc_o = render_ref.project_to_ndc(rep(vo), Rc.repeat(B, 1, 1), Tc.repeat(B, 1), torch.full((N,), 52.0))
    yx_o = render_ref.project_points_screen(rep(jo), Rc.repeat(B, 1, 1), Tc.repeat(B, 1), torch.full((N,), 52.0), S)
    ((ndc_o[..., :2] * d_ndc.cpu()).sum() + (yx_o * d_yx.cpu()).sum()).backward()
    src = a if ok_sup else b
    checks.append(close(got["ndc"].cpu(), ndc_o.detach(), 2e-5, "oracle ndc"))
    for k, n in (("d_beta", "beta"), ("d_theta", "theta"), ("d_trans", "trans"), ("d_logscale", "ls"), ("d_btrans", "bt")):
        if src[k] is not None and src[k].numel():
            checks.append(close(src[k].cpu(), leaves[n].grad, 5e-4, "oracle " + k))
    info = dict(V=V, J=J, nB=nB, static=int(t.static_joints), B=B, views=views, shared_beta=int(shared_beta), trans_after=int(trans_after),
                fused_bwd=int(ok_sup))
    return checks, info
