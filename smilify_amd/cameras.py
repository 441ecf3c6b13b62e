"""Camera set-up helpers (host side, run once per fit - not on the hot path).

* ``look_at_view_transform`` - the pytorch3d function the reference uses to build its default camera
  (smal_fitter/p3d_renderer.py:34) and its test rigs (tests/test_triangulation_consistency.py:73-107).
* ``FoVCameras`` - the small mutable holder behind ``Renderer.cameras``: the reference assigns
  ``renderer.cameras.fov = self.fov`` every forward (smal_fitter/fitter.py:285).
* ``opencv_to_fov_camera`` - the OpenCV/SLEAP -> FoV-camera conversion of
  smal_fitter/sleap_data/sleap_multiview_dataset.py:197-223.
"""
from __future__ import annotations

import math
from typing import Optional, Tuple

import numpy as np
import torch


def _normalize(v: np.ndarray, eps: float = 1e-5) -> np.ndarray:
    n = np.maximum(np.linalg.norm(v, axis=1, keepdims=True), eps)
    return v / n


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees: bool = True, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """World->view rotation R (n,3,3) and translation T (n,3) of cameras on a sphere around the origin,
    looking at it with +y up; row-vector convention X_view = X_world @ R + T."""
    d, e, a = np.broadcast_arrays(np.atleast_1d(np.asarray(dist, np.float64)), np.atleast_1d(np.asarray(elev, np.float64)),
                                  np.atleast_1d(np.asarray(azim, np.float64)))
    if degrees:
        e, a = np.deg2rad(e), np.deg2rad(a)
    C = np.stack([d * np.cos(e) * np.sin(a), d * np.sin(e), d * np.cos(e) * np.cos(a)], axis=1)
    up = np.tile(np.array([[0.0, 1.0, 0.0]]), (C.shape[0], 1))
    z_axis = _normalize(-C)
    x_axis = _normalize(np.cross(up, z_axis))
    y_axis = _normalize(np.cross(z_axis, x_axis))
    degenerate = np.all(np.isclose(x_axis, 0.0, atol=5e-3), axis=1)
    if degenerate.any():
        x_axis[degenerate] = _normalize(np.cross(y_axis, z_axis))[degenerate]
    R = np.stack([x_axis, y_axis, z_axis], axis=2)  # columns are the camera axes
    T = -np.einsum("nij,ni->nj", R, C)
    return (torch.tensor(R, dtype=torch.float32, device=device), torch.tensor(T, dtype=torch.float32, device=device))


class FoVCameras:
    """Mutable camera table: ``R (n,3,3)``, ``T (n,3)``, ``fov (n,)`` degrees, ``aspect_ratio (n,)`` or None."""

    def __init__(self, R: torch.Tensor, T: torch.Tensor, fov: torch.Tensor, aspect_ratio: Optional[torch.Tensor] = None,
                 znear: float = 0.001, zfar: float = 1000.0):
        self.R, self.T, self.fov, self.aspect_ratio = R, T, fov, aspect_ratio
        self.znear, self.zfar = znear, zfar

    def __len__(self) -> int:
        return int(max(self.R.shape[0], self.T.shape[0], self.fov.numel()))


def opencv_to_fov_camera(R_cv: np.ndarray, t_cv: np.ndarray, K: np.ndarray, image_size_wh) -> Tuple[np.ndarray, np.ndarray, float, float]:
    """(R, T, fov_y_degrees, aspect_ratio) of the FoV camera reproducing a pinhole calibration."""
    width, height = float(image_size_wh[0]), float(image_size_wh[1])
    fx, fy = float(K[0, 0]), float(K[1, 1])
    fov_y = float(2.0 * math.atan(height / (2.0 * fy)) * 180.0 / math.pi)
    aspect = float((width * fy) / (height * fx + 1e-12))
    flip = np.diag([-1.0, -1.0, 1.0]).astype(np.float32)
    return (np.asarray(R_cv, np.float32).T @ flip).astype(np.float32), (flip @ np.asarray(t_cv, np.float32)).astype(np.float32), fov_y, aspect
