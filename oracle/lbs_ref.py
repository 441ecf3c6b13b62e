"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (torch, fp32, differentiable) restatement of the reference's linear-blend-skinning path:

* ``rodrigues``                  <- smal_model/batch_lbs.py:31-50 (+ batch_skew :10-28)
* ``global_rigid_transformation`` <- smal_model/batch_lbs.py:75-197
* ``smal_forward``               <- smal_model/smal_torch.py:198-370 (``SMAL.__call__``)

Pinned against outputs of the real reference imported in the build container
(tests/golden/make_golden.py -> tests/golden/lbs_*.npz; checked by tests/test_oracle_lbs.py).
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may import this.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch


def rodrigues(theta: torch.Tensor) -> torch.Tensor:
    """(N,3) axis-angle -> (N,3,3).  Keeps the reference's epsilon quirk (batch_lbs.py:37-38):
    the norm is taken of ``theta + 1e-8`` but the axis divides the *unshifted* theta."""
    angle = torch.linalg.vector_norm(theta + 1e-8, dim=1, keepdim=True)  # (N,1)
    r = theta / angle
    c = torch.cos(angle)[:, :, None]
    s = torch.sin(angle)[:, :, None]
    rx, ry, rz = r[:, 0], r[:, 1], r[:, 2]
    zero = torch.zeros_like(rx)
    # skew(r): rows (0,-rz,ry), (rz,0,-rx), (-ry,rx,0)   (batch_lbs.py:18-25 scatter pattern)
    H = torch.stack([zero, -rz, ry, rz, zero, -rx, -ry, rx, zero], dim=1).view(-1, 3, 3)
    outer = r[:, :, None] * r[:, None, :]
    eye = torch.eye(3, dtype=theta.dtype).expand(theta.shape[0], 3, 3)
    return c * eye + (1 - c) * outer + s * H


def global_rigid_transformation(
    Rs: torch.Tensor,
    Js: torch.Tensor,
    parents,
    betas_logscale: Optional[torch.Tensor] = None,
    betas_trans: Optional[torch.Tensor] = None,
    propagate_scaling: bool = False,
    allow_limb_scaling: bool = True,
):
    """World transforms along the kinematic tree.

    Returns ``new_J (B,J,3)``, ``A (B,J,4,4)`` (relative skinning transforms) like the
    reference, plus ``G (B,J,4,4)`` world transforms for tests.
    """
    B, J = Rs.shape[0], Rs.shape[1]
    if not allow_limb_scaling:  # batch_lbs.py:123-124
        betas_logscale = None
    scale = torch.exp(betas_logscale) if betas_logscale is not None else torch.ones(B, J, 3, dtype=Rs.dtype)
    inv_scale = 1.0 / scale
    flip = torch.tensor([1.0, -1.0, 1.0], dtype=Rs.dtype)  # batch_lbs.py:148
    Grot = [Rs[:, 0]]  # root: no scale applied (batch_lbs.py:151)
    Gt = [Js[:, 0]]
    for i in range(1, J):
        p = int(parents[i])
        t = Js[:, i] - Js[:, p]
        if betas_trans is not None:
            t = t + betas_trans[:, i] * flip
        R = Rs[:, i]
        if not propagate_scaling:
            R = inv_scale[:, p, :, None] * R  # S_p^-1 . R   (batch_lbs.py:164,173)
        R = R * scale[:, i, None, :]  # . S_i
        Grot.append(torch.matmul(Grot[p], R))
        Gt.append(torch.matmul(Grot[p], t[:, :, None])[:, :, 0] + Gt[p])
    Grot = torch.stack(Grot, 1)  # (B,J,3,3)
    Gt = torch.stack(Gt, 1)  # (B,J,3)
    new_J = Gt
    # A_i = G_i with translation column reduced by G_i[:3,:3] . J_i  (batch_lbs.py:192-195)
    At = Gt - torch.matmul(Grot, Js[:, :, :, None])[:, :, :, 0]
    bottom = torch.tensor([0.0, 0.0, 0.0, 1.0], dtype=Rs.dtype).expand(B, J, 1, 4)
    A = torch.cat([torch.cat([Grot, At[:, :, :, None]], 3), bottom], 2)
    G = torch.cat([torch.cat([Grot, Gt[:, :, :, None]], 3), bottom], 2)
    return new_J, A, G


def smal_forward(
    model: Dict[str, torch.Tensor],
    beta: torch.Tensor,
    theta: torch.Tensor,
    trans: Optional[torch.Tensor] = None,
    del_v: Optional[torch.Tensor] = None,
    betas_logscale: Optional[torch.Tensor] = None,
    betas_trans: Optional[torch.Tensor] = None,
    propagate_scaling: bool = False,
    allow_limb_scaling: bool = True,
):
    """``SMAL.__call__`` restated with dense tables.

    ``model`` holds dense fp32 tensors: v_template (V,3), shapedirs (nB,3V), J_regressor (V,J),
    weights (V,J), parents (J,) ints, optional J_static (J,3), optional posedirs.
    Returns dict(verts, joints, Rs, v_shaped, J_rest, new_J, A).
    """
    v_template = model["v_template"]
    V = v_template.shape[0]
    B = theta.shape[0]
    nB = beta.shape[1]
    J = model["weights"].shape[1]
    if nB > 0:
        v_shaped = v_template + torch.matmul(beta, model["shapedirs"][:nB]).view(-1, V, 3)
    else:
        v_shaped = v_template[None]
    if del_v is not None:
        v_shaped = v_shaped + del_v
    if model.get("J_static") is not None:
        J_rest = model["J_static"][None].expand(v_shaped.shape[0], -1, -1)
    else:
        J_rest = torch.stack([torch.matmul(v_shaped[:, :, c], model["J_regressor"]) for c in range(3)], dim=2)
    if theta.dim() == 4:
        Rs = theta
    else:
        Rs = rodrigues(theta.reshape(-1, 3)).view(B, J, 3, 3)
    v_posed = v_shaped
    if model.get("posedirs") is not None:
        feat = (Rs[:, 1:] - torch.eye(3, dtype=Rs.dtype)).reshape(B, -1)
        v_posed = v_shaped + torch.matmul(feat, model["posedirs"]).view(B, V, 3)
    new_J, A, _ = global_rigid_transformation(
        Rs,
        J_rest.expand(B, -1, -1),
        model["parents"],
        betas_logscale=betas_logscale,
        betas_trans=betas_trans,
        propagate_scaling=propagate_scaling,
        allow_limb_scaling=allow_limb_scaling,
    )
    T = torch.matmul(model["weights"][None], A.reshape(B, J, 16)).view(B, V, 4, 4)
    vp = v_posed.expand(B, -1, -1)
    verts = torch.matmul(T[:, :, :3, :3], vp[:, :, :, None])[:, :, :, 0] + T[:, :, :3, 3]
    if trans is not None:
        verts = verts + trans[:, None, :]
    if model.get("J_static") is not None:
        joints = new_J  # static joints: chain output, no trans (smal_torch.py:343-346)
    else:
        joints = torch.stack([torch.matmul(verts[:, :, c], model["J_regressor"]) for c in range(3)], dim=2)
    return dict(verts=verts, joints=joints, Rs=Rs, v_shaped=v_shaped, J_rest=J_rest, new_J=new_J, A=A)
