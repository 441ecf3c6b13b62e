"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU (torch fp32, differentiable) restatement of the fit iteration:

* ``fit_losses``   <- smal_fitter/fitter.py:236-335 (``SMALFitter.forward``)
* ``temporal``     <- smal_fitter/fitter.py:337-350 (``SMALFitter.get_temporal``)
* ``shape_prior_precision`` <- smal_fitter/fitter.py:121-136,170-175
* ``default_global_rotation`` <- smal_fitter/fitter.py:206-210 + smal_fitter/utils.py:76-78
* ``fit_iteration``<- smal_fitter/optimize_to_joints.py:147-175 (sum of window means + temporal,
  one backward, one Adam(betas=(0.5,0.999)) step)

The loss terms are pinned against the real reference ``SMALFitter.forward`` imported in the build
container (tests/golden/fitter_*.npz).  The silhouette they consume comes from
oracle/render_ref.py, which is "parity unpinned" (pytorch3d absent).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch

from . import lbs_ref, render_ref

OBJ_KEYS = ("joint", "limit", "pose", "splay", "betas", "sil_reproj")


def default_global_rotation() -> np.ndarray:
    """Head-on init of fitter.py:206: ``eul_to_axis([-pi/2, 0, -pi/2])`` calls nibabel's
    ``euler2angle_axis(z, y, x)`` whose rotation is R_x(x) R_y(y) R_z(z) (z applied first), i.e. the
    quaternion (cx cz, sx cz, -sx sz, cx sz) at y = 0 -> axis-angle (-1.2092, -1.2092, -1.2092).
    (nibabel is absent here; SURVEY.md quotes a different sign on y - see DESIGN.md.)"""
    ex, ey, ez = -math.pi / 2, 0.0, -math.pi / 2
    Rx = np.array([[1, 0, 0], [0, math.cos(ex), -math.sin(ex)], [0, math.sin(ex), math.cos(ex)]])
    Ry = np.array([[math.cos(ey), 0, math.sin(ey)], [0, 1, 0], [-math.sin(ey), 0, math.cos(ey)]])
    Rz = np.array([[math.cos(ez), -math.sin(ez), 0], [math.sin(ez), math.cos(ez), 0], [0, 0, 1]])
    M = Rx @ Ry @ Rz
    ang = math.acos(max(-1.0, min(1.0, (np.trace(M) - 1) / 2)))
    ax = np.array([M[2, 1] - M[1, 2], M[0, 2] - M[2, 0], M[1, 0] - M[0, 1]]) / (2 * math.sin(ang))
    return (ax * ang).astype(np.float32)


def shape_prior_precision(shape_cov: Optional[np.ndarray], n_betas: int) -> np.ndarray:
    cov = np.eye(n_betas) if shape_cov is None else np.asarray(shape_cov, np.float64)
    invcov = np.linalg.inv(cov + 1e-5 * np.eye(cov.shape[0]))
    return np.linalg.cholesky(invcov)[:n_betas, :n_betas].astype(np.float32)


def fit_losses(
    model: Dict[str, torch.Tensor],
    params: Dict[str, torch.Tensor],
    batch_range: Sequence[int],
    weights: Sequence[float],
    targets: Dict[str, torch.Tensor],
    cams: Dict[str, torch.Tensor],
    image_size: int,
    mean_betas: torch.Tensor,
    betas_prec: torch.Tensor,
    canonical_joints: Optional[Sequence[int]] = None,
    propagate_scaling: bool = False,
    global_mask: Optional[torch.Tensor] = None,
    rotation_mask: Optional[torch.Tensor] = None,
    renderer=None,
):
    """One window of ``SMALFitter.forward``.  ``params``: betas (nB,), log_beta_scales (N,J,3),
    betas_trans (N,J,3), global_rotation (N,3), trans (N,3), joint_rotations (N,J-1,3), fov (N,) or (1,).
    ``cams``: R (Nc,3,3), T (Nc,3), optional aspect.  Returns (total, objs dict, extras dict)."""
    w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay = [float(w) for w in weights]
    br = list(batch_range)
    b = len(br)
    J = params["joint_rotations"].shape[1] + 1
    gmask = torch.ones(1, 3) if global_mask is None else global_mask
    rmask = torch.ones(J - 1, 3) if rotation_mask is None else rotation_mask
    grot = params["global_rotation"][br] * gmask
    jrot = params["joint_rotations"][br] * rmask
    betas = params["betas"].expand(b, -1)
    trans = params["trans"][br]
    fov = params["fov"]
    fov_b = fov[br] if fov.shape[0] > 1 else fov.expand(b)
    lbs_scale = params["log_beta_scales"]
    lbs_scale = lbs_scale[br] if lbs_scale.shape[0] > 1 and lbs_scale.shape[0] != b else lbs_scale.expand(b, J, 3)
    btr = params["betas_trans"]
    btr = btr[br] if btr.shape[0] > 1 and btr.shape[0] != b else btr.expand(b, J, 3)

    theta = torch.cat([grot[:, None], jrot], dim=1)
    out = lbs_ref.smal_forward(model, betas, theta, betas_logscale=lbs_scale, betas_trans=btr,
                               propagate_scaling=propagate_scaling)
    verts = out["verts"] + trans[:, None]
    joints = out["joints"] + trans[:, None]
    cj = list(range(J)) if canonical_joints is None else list(canonical_joints)
    cjoints = joints[:, cj]

    R = cams["R"] if cams["R"].shape[0] == b else (cams["R"][br] if cams["R"].shape[0] > 1 else cams["R"].expand(b, 3, 3))
    T = cams["T"] if cams["T"].shape[0] == b else (cams["T"][br] if cams["T"].shape[0] > 1 else cams["T"].expand(b, 3))
    aspect = cams.get("aspect")
    if renderer is None:
        proj = render_ref.project_points_screen(cjoints, R, T, fov_b, image_size, aspect)
        sil = None
        if w_reproj > 0:
            sil = render_ref.render_silhouette(verts, model["faces"], R, T, fov_b, image_size, aspect)
    else:
        sil, proj = renderer(verts, cjoints, model["faces"])

    objs = {}
    if w_j2d > 0:
        vis = targets["visibility"][br].bool()
        tj = targets["joints"][br].clone()
        rj = torch.where(vis[:, :, None], proj, torch.full_like(proj, -1.0))
        tj = torch.where(vis[:, :, None], tj, torch.full_like(tj, -1.0))
        objs["joint"] = w_j2d * torch.mean((rj - tj) ** 2)  # denominator counts invisible joints
    if w_limit > 0:
        lim = 0.01  # joint_limits_prior.py:8-15 under ignore_hardcoded_body
        zeros = torch.zeros_like(jrot)
        objs["limit"] = w_limit * torch.mean(torch.max(jrot - lim, zeros) + torch.max(-lim - jrot, zeros))
    if w_pose > 0:
        use = torch.ones(3 * J)
        use[:3] = 0.0
        objs["pose"] = w_pose * ((theta.reshape(b, 3 * J) * use) ** 2).mean()
    if w_splay > 0:
        objs["splay"] = w_splay * torch.sum(jrot[:, :, [0, 2]] ** 2)
    if w_betas > 0:
        res = torch.matmul(betas - mean_betas[None], betas_prec)
        objs["betas"] = w_betas * (res ** 2).mean()
    if w_reproj > 0 and sil is not None:
        objs["sil_reproj"] = w_reproj * torch.mean(torch.abs(sil - targets["sil"][br]))
    total = sum(objs.values())
    return total, objs, dict(verts=verts, joints=joints, proj=proj, sil=sil)


def temporal(params, w_temp: float, global_mask=None, rotation_mask=None):
    """Frame-to-frame MSE terms; returns (joint_loss, global_loss, trans_loss)."""
    jr = params["joint_rotations"] if rotation_mask is None else params["joint_rotations"] * rotation_mask
    gr = params["global_rotation"] if global_mask is None else params["global_rotation"] * global_mask
    tr = params["trans"]
    if jr.shape[0] < 2:
        z = torch.tensor(0.0)
        return z, z.clone(), z.clone()
    gl = ((gr[1:] - gr[:-1]) ** 2).mean(dim=1).sum() * w_temp
    jl = ((jr[1:] - jr[:-1]) ** 2).mean(dim=(1, 2)).sum() * w_temp
    tl = ((tr[1:] - tr[:-1]) ** 2).mean(dim=1).sum() * w_temp
    return jl, gl, tl


def fit_iteration_loss(model, params, windows: List[Sequence[int]], weights, w_temp, targets, cams, image_size,
                       mean_betas, betas_prec, **kw):
    """Accumulated loss of one epoch: sum over windows of the window mean + temporal terms
    (optimize_to_joints.py:154-171)."""
    acc = 0.0
    all_objs = []
    for br in windows:
        total, objs, _ = fit_losses(model, params, br, weights, targets, cams, image_size, mean_betas, betas_prec, **kw)
        acc = acc + total.mean()
        all_objs.append({k: float(v) for k, v in objs.items()})
    jl, gl, tl = temporal(params, w_temp, kw.get("global_mask"), kw.get("rotation_mask"))
    return acc + jl + gl + tl, all_objs, (float(jl), float(gl), float(tl))
