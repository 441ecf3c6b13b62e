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


def _unit(v: torch.Tensor, eps: float = 1e-5) -> torch.Tensor:
    return v / v.norm(dim=1, keepdim=True).clamp_min(eps)


def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees: bool = True, device="cpu") -> Tuple[torch.Tensor, torch.Tensor]:
    """World->view rotation R (n,3,3) and translation T (n,3) of cameras on a sphere around the origin,
    looking at it with +y up; row-vector convention X_view = X_world @ R + T.  Computed in fp32 like the
    pytorch3d function it replaces, so degenerate set-ups (camera on the up axis) resolve the same way."""
    f32 = lambda x: torch.as_tensor(np.asarray(x, dtype=np.float32)).reshape(-1)  # noqa: E731
    d, e, a = torch.broadcast_tensors(f32(dist), f32(elev), f32(azim))
    if degrees:
        e, a = e * (math.pi / 180.0), a * (math.pi / 180.0)
    C = torch.stack([d * torch.cos(e) * torch.sin(a), d * torch.sin(e), d * torch.cos(e) * torch.cos(a)], dim=1)
    up = torch.tensor([[0.0, 1.0, 0.0]]).expand_as(C)
    z_axis = _unit(-C)
    x_axis = _unit(torch.cross(up, z_axis, dim=1))
    y_axis = _unit(torch.cross(z_axis, x_axis, dim=1))
    degenerate = torch.isclose(x_axis, torch.zeros(()), atol=5e-3).all(dim=1, keepdim=True)
    if bool(degenerate.any()):
        x_axis = torch.where(degenerate, _unit(torch.cross(y_axis, z_axis, dim=1)), x_axis)
    R = torch.stack([x_axis, y_axis, z_axis], dim=2)  # columns are the camera axes
    T = -torch.einsum("nij,ni->nj", R, C)
    return R.contiguous().to(device), T.contiguous().to(device)


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
