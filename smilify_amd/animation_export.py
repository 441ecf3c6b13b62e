"""AMASS-style animation export of fitted / predicted SMIL parameters.

``AnimationRecorder`` keeps the constructor, ``record`` / ``set_cameras`` / ``num_frames`` / ``write`` contract and the
on-disk schema (``.npz`` payload keys + ``.json`` side-car, schema 1.1) of the reference's
``smal_fitter/neuralSMIL/animation_export.py:41-218`` so that files written here load in the reference's Blender /
video tooling and vice versa.  ``record_fitter`` feeds it from a ``SMALFitter`` (one frame per fitted frame).
Host-side format code only: nothing here runs on the hot path.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional, Union

import numpy as np
import torch

SCHEMA_VERSION = "1.1"  # reference animation_export.py:18


def rotation_6d_to_axis_angle(d6: torch.Tensor) -> torch.Tensor:
    """(..., 6) continuous rotation representation -> (..., 3) axis-angle.  The reference delegates to pytorch3d
    (animation_export.py:21-31); this is the same two published steps without it: Gram-Schmidt of the two 3-vectors
    into a rotation matrix (rows b1, b2, b1 x b2), then the matrix logarithm."""
    a1, a2 = d6[..., :3], d6[..., 3:]
    b1 = torch.nn.functional.normalize(a1, dim=-1)
    b2 = torch.nn.functional.normalize(a2 - (b1 * a2).sum(-1, keepdim=True) * b1, dim=-1)
    b3 = torch.cross(b1, b2, dim=-1)
    R = torch.stack((b1, b2, b3), dim=-2).double()
    cos = ((R[..., 0, 0] + R[..., 1, 1] + R[..., 2, 2]) - 1.0) / 2.0
    angle = torch.acos(cos.clamp(-1.0, 1.0))
    skew = torch.stack((R[..., 2, 1] - R[..., 1, 2], R[..., 0, 2] - R[..., 2, 0], R[..., 1, 0] - R[..., 0, 1]), dim=-1)
    small = angle < 1e-6
    scale = torch.where(small, torch.full_like(angle, 0.5), angle / (2.0 * torch.sin(angle).clamp_min(1e-12)))
    aa = skew * scale[..., None]
    near_pi = (torch.pi - angle) < 1e-4
    if bool(near_pi.any()):  # the skew part vanishes at pi: take the axis from the symmetric part
        diag = torch.stack((R[..., 0, 0], R[..., 1, 1], R[..., 2, 2]), dim=-1)
        axis = torch.sqrt(((diag + 1.0) / 2.0).clamp_min(0.0))
        sign = torch.stack((torch.ones_like(angle), torch.sign(R[..., 0, 1] + R[..., 1, 0]), torch.sign(R[..., 0, 2] + R[..., 2, 0])), dim=-1)
        sign = torch.where(sign == 0, torch.ones_like(sign), sign)
        aa = torch.where(near_pi[..., None], torch.nn.functional.normalize(axis * sign, dim=-1) * angle[..., None], aa)
    return aa.to(d6.dtype)


def _to_numpy(t: Any) -> np.ndarray:
    if isinstance(t, torch.Tensor):
        return t.detach().cpu().float().numpy()
    return np.asarray(t, dtype=np.float32)


class AnimationRecorder:
    """Accumulates per-frame ``predicted_params`` dicts (batch dimension 1) and writes ``<output_path>.npz`` +
    ``<output_path>.json``.  Rotations are stored as axis-angle whatever the inbound representation."""

    def __init__(self, output_path: Union[str, Path], rotation_representation: str, n_joints: int, n_betas: int,
                 joint_names: List[str], parents: List[int], fps: float, static_joint_locs: bool, ignore_hardcoded_body: bool,
                 source_checkpoint: Optional[str] = None, source_input: Optional[str] = None, model_id: Optional[str] = None) -> None:
        if rotation_representation not in ("axis_angle", "6d"):
            raise ValueError(f"rotation_representation must be 'axis_angle' or '6d', got {rotation_representation!r}")
        self.output_path = Path(output_path)
        self.rotation_representation = rotation_representation
        self.n_joints, self.n_betas = int(n_joints), int(n_betas)
        self.joint_names = [str(n) for n in joint_names]
        self.parents = [int(p) for p in parents]
        self.fps = float(fps)
        self.static_joint_locs, self.ignore_hardcoded_body = bool(static_joint_locs), bool(ignore_hardcoded_body)
        self.source_checkpoint, self.source_input, self.model_id = source_checkpoint, source_input, model_id
        self._rows: Dict[str, List[np.ndarray]] = {k: [] for k in ("poses", "trans", "betas", "log_beta_scales", "betas_trans", "mesh_scale",
                                                                  "cam_rot", "cam_trans", "fov")}
        self._cameras_sidecar: List[Dict[str, Any]] = []

    def _axis_angle(self, rot):
        if isinstance(rot, torch.Tensor):
            rot = rot.detach().cpu()
        else:
            rot = torch.as_tensor(np.asarray(rot, dtype=np.float32))
        return rotation_6d_to_axis_angle(rot) if self.rotation_representation == "6d" else rot

    def record(self, predicted_params: Dict[str, Any]) -> None:
        g, j = self._axis_angle(predicted_params["global_rot"]), self._axis_angle(predicted_params["joint_rot"])
        if g.dim() == 2:  # (B,3) -> (B,1,3)
            g = g.unsqueeze(1)
        self._rows["poses"].append(_to_numpy(torch.cat([g, j], dim=1)[0]))
        self._rows["trans"].append(_to_numpy(predicted_params["trans"][0]))
        self._rows["betas"].append(_to_numpy(predicted_params["betas"][0]))
        for key in ("log_beta_scales", "betas_trans", "cam_rot", "cam_trans", "fov"):
            if predicted_params.get(key, None) is not None:
                self._rows[key].append(_to_numpy(predicted_params[key][0]))
        if predicted_params.get("mesh_scale", None) is not None:
            self._rows["mesh_scale"].append(_to_numpy(predicted_params["mesh_scale"][0]).reshape(-1))

    def set_cameras(self, cameras: List[Dict[str, Any]]) -> None:
        """Cameras block of the side-car for multi-view clips: ``{"view_name", "R" 3x3, "t" 3, "fov"}`` per view."""
        self._cameras_sidecar = list(cameras)

    def num_frames(self) -> int:
        return len(self._rows["poses"])

    def _averaged_camera(self) -> List[Dict[str, Any]]:
        r = self._rows
        if not r["cam_rot"]:
            return []
        t = np.stack(r["cam_trans"]).mean(axis=0) if r["cam_trans"] else np.zeros(3, np.float32)
        return [{"view_name": "view_0", "R": np.stack(r["cam_rot"]).mean(axis=0).tolist(), "t": t.flatten().tolist(),
                 "fov": float(np.mean(r["fov"])) if r["fov"] else 0.0}]

    def write(self) -> Dict[str, Path]:
        r = self._rows
        if not r["poses"]:
            raise RuntimeError("AnimationRecorder has no frames to write.")
        self.output_path.parent.mkdir(parents=True, exist_ok=True)
        npz_path, json_path = self.output_path.with_suffix(".npz"), self.output_path.with_suffix(".json")
        poses = np.stack(r["poses"]).astype(np.float32)
        betas_per_frame = np.stack(r["betas"]).astype(np.float32)
        payload: Dict[str, Any] = {"poses": poses, "trans": np.stack(r["trans"]).astype(np.float32),
                                   "betas": betas_per_frame.mean(axis=0).astype(np.float32), "betas_per_frame": betas_per_frame,
                                   "fps": np.float32(self.fps)}
        for key in ("log_beta_scales", "betas_trans"):
            if r[key]:
                payload[key] = np.stack(r[key]).astype(np.float32)
        if r["mesh_scale"]:  # (F,): isotropic scale about the root joint
            payload["mesh_scale"] = np.stack(r["mesh_scale"]).astype(np.float32).reshape(-1)
        np.savez(npz_path, **payload)
        sidecar = {"schema_version": SCHEMA_VERSION, "model_id": self.model_id, "source_checkpoint": self.source_checkpoint,
                   "source_input": self.source_input, "n_frames": int(poses.shape[0]), "n_joints": self.n_joints, "n_betas": self.n_betas,
                   "joint_names": self.joint_names, "parents": self.parents, "rotation_representation": "axis_angle",
                   "root_joint_index": 0, "static_joint_locs": self.static_joint_locs, "ignore_hardcoded_body": self.ignore_hardcoded_body,
                   "fps": self.fps, "cameras": self._cameras_sidecar or self._averaged_camera()}
        with open(json_path, "w") as f:
            json.dump(sidecar, f, indent=2)
        return {"npz": npz_path, "json": json_path}


def record_fitter(fitter, output_path: Union[str, Path], fps: float = 30.0, view_names: Optional[List[str]] = None,
                  model_id: Optional[str] = None) -> Dict[str, Path]:
    """Write the current parameters of a ``SMALFitter`` (every frame of this rank) as one clip; cameras go to the
    side-car as one static entry per view."""
    t = fitter.smal_model.tables
    rec = AnimationRecorder(output_path, "axis_angle", t.J, t.nB, list(t.joint_names), [int(p) for p in t.parents], fps,
                            bool(t.static_joints), bool(fitter.config.ignore_hardcoded_body), model_id=model_id or t.name)
    N = fitter.num_images
    gr = (fitter.global_rotation.detach() * fitter.global_mask).cpu()
    jr = (fitter.joint_rotations.detach() * fitter.rotation_mask).cpu()
    tr, betas = fitter.trans.detach().cpu(), fitter.betas.detach().cpu()
    ls, bt = fitter.log_beta_scales.detach().cpu(), fitter.betas_trans.detach().cpu()
    for i in range(N):
        rec.record(dict(global_rot=gr[i:i + 1], joint_rot=jr[i:i + 1], trans=tr[i:i + 1], betas=betas[None],
                        log_beta_scales=ls[min(i, ls.shape[0] - 1)][None], betas_trans=bt[min(i, bt.shape[0] - 1)][None]))
    cam = fitter.renderer.cameras
    fov = fitter.fov.detach().reshape(-1).cpu()
    cams = []
    for v in range(fitter.views):
        cams.append({"view_name": view_names[v] if view_names else f"view_{v}", "R": cam.R[min(v, cam.R.shape[0] - 1)].cpu().tolist(),
                     "t": cam.T[min(v, cam.T.shape[0] - 1)].cpu().tolist(), "fov": float(fov[min(v, fov.numel() - 1)])})
    rec.set_cameras(cams)
    return rec.write()
