"""AMASS-style animation export of fitted / predicted SMIL parameters.

``AnimationRecorder`` answers to the constructor / ``record`` / ``set_cameras`` / ``num_frames`` / ``write`` calls of the
reference's recorder (``smal_fitter/neuralSMIL/animation_export.py:41-218``) and writes its on-disk schema (``.npz`` payload
keys + ``.json`` side-car, schema 1.1), so files written here load in the reference's Blender / video tooling and vice
versa.  Inside it is a table-driven column store with a bulk path: ``record_fitter`` copies a ``SMALFitter``'s parameter
buffers into a clip with one copy per track.
Host-side format code only: nothing here runs on the hot path.
"""
from __future__ import annotations

import json
from pathlib import Path
from typing import Any, Dict, List, Optional, Tuple, Union

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


# ---------------------------------------------------------------------------------------------------------------------
# The clip file format is the contract (``.npz`` payload + ``.json`` side-car, schema 1.1, as the reference's Blender /
# video tooling reads it: neuralSMIL/animation_export.py:160-210).  Everything below is organised around ONE table of the
# per-frame tracks a clip can hold; tracks live in preallocated float32 column stores that bulk producers (a fitter's
# parameter buffers) fill with one copy per track.
# ---------------------------------------------------------------------------------------------------------------------
class _Track:
    """Growable ``(frames, *cell)`` float32 store (capacity doubles; ``rows()`` is a contiguous view of what is filled)."""

    def __init__(self, cell: Tuple[int, ...]) -> None:
        self.cell = tuple(int(c) for c in cell)
        self._buf = np.empty((0,) + self.cell, np.float32)
        self.filled = 0

    def append_block(self, block: np.ndarray) -> None:
        block = np.asarray(block, np.float32).reshape((-1,) + self.cell)
        need = self.filled + block.shape[0]
        if need > self._buf.shape[0]:
            grown = np.empty((max(need, 2 * self._buf.shape[0], 16),) + self.cell, np.float32)
            grown[:self.filled] = self._buf[:self.filled]
            self._buf = grown
        self._buf[self.filled:need] = block
        self.filled = need

    def rows(self) -> np.ndarray:
        return self._buf[:self.filled]


def _host_f32(x: Any) -> np.ndarray:
    return x.detach().to("cpu", torch.float32).numpy() if isinstance(x, torch.Tensor) else np.asarray(x, np.float32)


# track name -> (cell shape from (n_joints, n_betas), mandatory in every frame?, payload key or None when side-car only)
_TRACKS = {
    "poses": (lambda J, nB: (J, 3), True, "poses"),
    "trans": (lambda J, nB: (3,), True, "trans"),
    "betas": (lambda J, nB: (nB,), True, "betas_per_frame"),
    "log_beta_scales": (lambda J, nB: (J, 3), False, "log_beta_scales"),
    "betas_trans": (lambda J, nB: (J, 3), False, "betas_trans"),
    "mesh_scale": (lambda J, nB: (), False, "mesh_scale"),
    "cam_rot": (lambda J, nB: (3, 3), False, None),
    "cam_trans": (lambda J, nB: (3,), False, None),
    "fov": (lambda J, nB: (), False, None),
}
# side-car layout: (key, attribute of the recorder or a callable of it), in file order
_SIDECAR = (
    ("schema_version", lambda r: SCHEMA_VERSION), ("model_id", "model_id"), ("source_checkpoint", "source_checkpoint"),
    ("source_input", "source_input"), ("n_frames", lambda r: r.num_frames()), ("n_joints", "n_joints"), ("n_betas", "n_betas"),
    ("joint_names", "joint_names"), ("parents", "parents"), ("rotation_representation", lambda r: "axis_angle"),
    ("root_joint_index", lambda r: 0), ("static_joint_locs", "static_joint_locs"), ("ignore_hardcoded_body", "ignore_hardcoded_body"),
    ("fps", "fps"), ("cameras", lambda r: r._camera_block()),
)
_ROTATION_DECODERS = {"axis_angle": lambda t: t, "6d": rotation_6d_to_axis_angle}


class AnimationRecorder:
    """Collects frames of SMIL parameters and writes ``<output_path>.npz`` + ``<output_path>.json``.

    ``record(params)`` takes one frame the way the reference's inference loop hands it over (tensors with a leading batch
    dimension of 1, rotations in ``rotation_representation``); ``record_block(**tracks)`` takes whole ``(F, ...)`` arrays at
    once.  Rotations always reach the file as axis-angle."""

    def __init__(self, output_path: Union[str, Path], rotation_representation: str, n_joints: int, n_betas: int,
                 joint_names: List[str], parents: List[int], fps: float, static_joint_locs: bool, ignore_hardcoded_body: bool,
                 source_checkpoint: Optional[str] = None, source_input: Optional[str] = None, model_id: Optional[str] = None) -> None:
        try:
            self._decode_rotation = _ROTATION_DECODERS[rotation_representation]
        except KeyError:
            raise ValueError(f"rotation_representation must be one of {sorted(_ROTATION_DECODERS)}, "
                             f"got {rotation_representation!r}") from None
        self.rotation_representation = rotation_representation
        self.output_path = Path(output_path)
        self.n_joints, self.n_betas, self.fps = int(n_joints), int(n_betas), float(fps)
        self.joint_names, self.parents = list(map(str, joint_names)), list(map(int, parents))
        self.static_joint_locs, self.ignore_hardcoded_body = bool(static_joint_locs), bool(ignore_hardcoded_body)
        self.source_checkpoint, self.source_input, self.model_id = source_checkpoint, source_input, model_id
        self._tracks = {name: _Track(cell(self.n_joints, self.n_betas)) for name, (cell, _, _) in _TRACKS.items()}
        self._static_cameras: List[Dict[str, Any]] = []

    # ---- producers --------------------------------------------------------------------------------------------------
    def record_block(self, **tracks: Any) -> None:
        """Append ``F`` frames at once: every given track is an ``(F, *cell)`` array (axis-angle ``poses`` incl. the root);
        one copy per track.  Mandatory tracks must be present; an optional track must come with every block or none."""
        given = {k: _host_f32(v) for k, v in tracks.items() if v is not None}
        unknown = set(given) - set(_TRACKS)
        if unknown:
            raise KeyError(f"unknown animation tracks {sorted(unknown)}")
        for name, (_, mandatory, _) in _TRACKS.items():
            if mandatory and name not in given:
                raise KeyError(f"animation frame without {name!r}")
        for name, block in given.items():
            self._tracks[name].append_block(block)

    def record(self, predicted_params: Dict[str, Any]) -> None:
        """One frame from a ``predicted_params`` dict (``global_rot``, ``joint_rot``, ``trans``, ``betas`` and the optional
        tracks); only batch element 0 is taken."""
        def first(key):
            v = predicted_params.get(key)
            return None if v is None else (v.detach().cpu() if isinstance(v, torch.Tensor) else torch.as_tensor(np.asarray(v, np.float32)))[0]

        root = self._decode_rotation(first("global_rot")).reshape(1, 3)
        limbs = self._decode_rotation(first("joint_rot")).reshape(-1, 3)
        frame = {"poses": torch.cat((root, limbs))[None], "trans": first("trans")[None], "betas": first("betas")[None]}
        for name in ("log_beta_scales", "betas_trans", "mesh_scale", "cam_rot", "cam_trans", "fov"):
            v = first(name)
            if v is not None:
                frame[name] = v.reshape((1,) + self._tracks[name].cell)
        self.record_block(**frame)

    def set_cameras(self, cameras: List[Dict[str, Any]]) -> None:
        """Static cameras of a multi-view clip for the side-car: ``{"view_name", "R" 3x3, "t" 3, "fov"}`` per view."""
        self._static_cameras = [dict(c) for c in cameras]

    def num_frames(self) -> int:
        return self._tracks["poses"].filled

    # ---- file ---------------------------------------------------------------------------------------------------------
    def _camera_block(self) -> List[Dict[str, Any]]:
        if self._static_cameras:
            return self._static_cameras
        rot, pos, fov = (self._tracks[k].rows() for k in ("cam_rot", "cam_trans", "fov"))
        if not len(rot):  # no per-frame camera was ever recorded
            return []
        # a single-view clip carries its (predicted, per-frame) camera as one time-averaged entry
        return [{"view_name": "view_0", "R": rot.mean(axis=0).tolist(),
                 "t": (pos.mean(axis=0) if len(pos) else np.zeros(3, np.float32)).tolist(),
                 "fov": float(fov.mean()) if len(fov) else 0.0}]

    def write(self) -> Dict[str, Path]:
        frames = self.num_frames()
        if frames == 0:
            raise RuntimeError("AnimationRecorder has no frames to write.")
        payload: Dict[str, Any] = {}
        for name, (_, mandatory, key) in _TRACKS.items():
            rows = self._tracks[name].rows()
            if key is None or not len(rows):
                continue
            if len(rows) != frames:
                raise RuntimeError(f"track {name!r} holds {len(rows)} frames, the clip {frames}")
            payload[key] = rows.copy()
        payload["betas"] = payload["betas_per_frame"].mean(axis=0).astype(np.float32)  # the clip's single shape vector
        payload["fps"] = np.float32(self.fps)
        paths = {"npz": self.output_path.with_suffix(".npz"), "json": self.output_path.with_suffix(".json")}
        self.output_path.parent.mkdir(parents=True, exist_ok=True)
        np.savez(paths["npz"], **payload)
        sidecar = {key: (src(self) if callable(src) else getattr(self, src)) for key, src in _SIDECAR}
        paths["json"].write_text(json.dumps(sidecar, indent=2))
        return paths


def record_fitter(fitter, output_path: Union[str, Path], fps: float = 30.0, view_names: Optional[List[str]] = None,
                  model_id: Optional[str] = None) -> Dict[str, Path]:
    """Write the current parameters of a ``SMALFitter`` (every frame of this rank) as one clip: each track is copied from
    the fitter's parameter buffer in one piece; cameras go to the side-car as one static entry per view."""
    t = fitter.smal_model.tables
    rec = AnimationRecorder(output_path, "axis_angle", t.J, t.nB, list(t.joint_names), [int(p) for p in t.parents], fps,
                            bool(t.static_joints), bool(fitter.config.ignore_hardcoded_body), model_id=model_id or t.name)
    N = fitter.num_images
    with torch.no_grad():
        poses = torch.cat(((fitter.global_rotation * fitter.global_mask)[:, None], fitter.joint_rotations * fitter.rotation_mask), dim=1)
        per_frame = lambda p: p.expand(N, *p.shape[1:]) if p.shape[0] != N else p  # noqa: E731  (tables shared by all frames)
        rec.record_block(poses=poses, trans=fitter.trans, betas=fitter.betas[None].expand(N, -1),
                         log_beta_scales=per_frame(fitter.log_beta_scales), betas_trans=per_frame(fitter.betas_trans))
    cam = fitter.renderer.cameras
    fov = fitter.fov.detach().reshape(-1).cpu()
    pick = lambda a, v: a[min(v, a.shape[0] - 1)]  # noqa: E731  (one shared camera or one per view)
    rec.set_cameras([{"view_name": view_names[v] if view_names else f"view_{v}", "R": pick(cam.R, v).cpu().tolist(),
                      "t": pick(cam.T, v).cpu().tolist(), "fov": float(pick(fov, v))} for v in range(fitter.views)])
    return rec.write()
