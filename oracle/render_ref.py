"""ORACLE - TEST INFRASTRUCTURE ONLY.  Not part of the product path.

CPU restatement of the camera/projection/silhouette arithmetic behind the reference's
``Renderer`` (smal_fitter/p3d_renderer.py:27-152).  The arithmetic itself lives in the
un-vendored third-party ``pytorch3d`` (pinned 0.7.8, environment.yml:35) - this restates the
published algorithm of ``FoVPerspectiveCameras`` / ``look_at_view_transform`` /
``transform_points_screen`` / ``MeshRasterizer.transform`` and binds the C restatement of the
naive rasteriser (oracle/raster_oracle.c).  PARITY UNPINNED by reference-owned vectors; pinned
by analytic known-answer tests only (tests/test_oracle_raster.py).
"""
from __future__ import annotations

import ctypes
import math
import os
import subprocess
from typing import Optional

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

SIGMA = 1e-4
BLUR_RADIUS = float(np.log(1.0 / 1e-4 - 1.0) * SIGMA)  # p3d_renderer.py:44
FACES_PER_PIXEL = 100  # p3d_renderer.py:45
ZNEAR, ZFAR = 0.001, 1000.0  # p3d_renderer.py:24-25


def build_oracle_lib() -> str:
    so = os.path.join(_HERE, "_build", "libraster_oracle.so")
    src = os.path.join(_HERE, "raster_oracle.c")
    if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return so


def _lib():
    global _LIB
    if _LIB is None:
        lib = ctypes.CDLL(build_oracle_lib())
        fp = ctypes.POINTER(ctypes.c_float)
        ip = ctypes.POINTER(ctypes.c_int32)
        lib.oracle_silhouette_forward.argtypes = [fp, ip] + [ctypes.c_int] * 4 + [ctypes.c_float] * 2 + [
            ctypes.c_int, fp, ip, ip, fp, fp]
        lib.oracle_silhouette_forward.restype = ctypes.c_int
        lib.oracle_silhouette_backward.argtypes = [fp, ip] + [ctypes.c_int] * 4 + [ctypes.c_float] * 2 + [
            ctypes.c_int, fp, fp]
        lib.oracle_silhouette_backward.restype = ctypes.c_int
        lib.oracle_num_threads.restype = ctypes.c_int
        lib.oracle_set_z_clip.argtypes = [ctypes.c_float]
        _LIB = lib
    return _LIB


class select_mode:
    """Context manager: which K faces survive truncation.  0 (default) = the reference's sequential queue (faithful);
    1 = the K smallest by (depth, face id), the order-independent rule the HIP rasteriser implements."""

    def __init__(self, mode: int):
        self.mode = int(mode)

    def __enter__(self):
        _lib().oracle_set_select_mode(self.mode)
        return self

    def __exit__(self, *exc):
        _lib().oracle_set_select_mode(0)
        return False


_Z_CLIP = ZNEAR / 2.0


def set_z_clip(z: float) -> None:
    """z_clip_value of the rasteriser (default znear / 2 = 5e-4, as the reference's camera gives it)."""
    global _Z_CLIP
    _Z_CLIP = float(z)
    _lib().oracle_set_z_clip(ctypes.c_float(float(z)))


def clip_faces_np(verts: np.ndarray, faces: np.ndarray, z_clip: float):
    """pytorch3d ``clip_faces`` (renderer/mesh/clip.py, 0.7.x; un-vendored, restated from its published algorithm - PARITY
    UNPINNED) for ONE mesh in the rasteriser's input space ``(x_ndc, y_ndc, z_view)``, as MeshRasterizer applies it with
    ``z_clip_value = znear / 2`` and ``cull_to_frustum = False`` (reference p3d_renderer.py:36-47 leaves both defaults):

    * a vertex is "behind" when ``z < z_clip``; faces with no vertex behind are kept, with all three behind removed;
    * ONE behind (p1; p2, p3 in front, cyclic order kept): the front part is a quadrilateral, split into (p4, p2, p3) and
      (p4, p3, p5); TWO behind (p1 in front; p2, p3 behind): the front part is the triangle (p1, p4, p5);
      p4 / p5 = where the edges p1-p2 / p1-p3 cross the plane, interpolated in VIEW space (perspective_correct): with
      ``w_b = (z_a - z_clip) / (z_a - z_b)``, ``xy = (xy_a z_a (1 - w_b) + xy_b z_b w_b) / z_clip``, ``z = z_clip``.

    Returns ``(verts_aug (V + X, 3), faces_aug (F + E, 3), src (X, 2) int, coef (X, 2) float64)``: rows 0..F-1 of ``faces_aug``
    keep the face ids (a clipped face becomes a zero-area placeholder), the front parts follow at F...; every new vertex j is
    ``coef[j,0] xy[src[j,0]] + coef[j,1] xy[src[j,1]]``.  Gradients (``silhouette_backward_np``): the rasteriser's xy gradient of
    a new vertex goes back to its two source vertices - to their xy with those coefficients, and, since round 5, to their DEPTHS
    through ``w_b`` and the explicit ``z_a``, ``z_b`` factors, as pytorch3d's autograd does (``clip_depth_gradient``)."""
    v = np.asarray(verts, np.float64)
    f = np.asarray(faces, np.int64)
    behind = v[:, 2][f] < z_clip                      # (F, 3)
    nb = behind.sum(1)
    clip_ids = np.nonzero((nb == 1) | (nb == 2))[0]
    if clip_ids.size == 0:
        return np.asarray(verts, np.float32), np.asarray(faces, np.int32), np.zeros((0, 2), np.int64), np.zeros((0, 2))
    V = v.shape[0]
    new_v, src, coef, new_f = [], [], [], []
    f_aug = f.copy()

    def cross(a, b):  # plane crossing of the segment from vertex a to vertex b
        wb = (v[a, 2] - z_clip) / (v[a, 2] - v[b, 2])
        ca, cb = v[a, 2] * (1.0 - wb) / z_clip, v[b, 2] * wb / z_clip
        new_v.append([ca * v[a, 0] + cb * v[b, 0], ca * v[a, 1] + cb * v[b, 1], z_clip])
        src.append([a, b]); coef.append([ca, cb])
        return V + len(new_v) - 1

    for fi in clip_ids:
        tri, bh = f[fi], behind[fi]
        k = int(np.nonzero(bh)[0][0]) if nb[fi] == 1 else int(np.nonzero(~bh)[0][0])  # the isolated vertex
        p1, p2, p3 = int(tri[k]), int(tri[(k + 1) % 3]), int(tri[(k + 2) % 3])
        p4, p5 = cross(p1, p2), cross(p1, p3)
        if nb[fi] == 1:
            new_f += [[p4, p2, p3], [p4, p3, p5]]
        else:
            new_f += [[p1, p4, p5]]
        f_aug[fi] = [tri[0], tri[0], tri[0]]          # placeholder: zero area, never rendered
    verts_aug = np.concatenate([v, np.asarray(new_v)]).astype(np.float32)
    faces_aug = np.concatenate([f_aug, np.asarray(new_f, np.int64)]).astype(np.int32)
    return verts_aug, faces_aug, np.asarray(src, np.int64), np.asarray(coef, np.float64)


def clip_depth_gradient(va, vb, g_xy, z_clip):
    """d (new vertex xy) / d (z_a, z_b) contracted with the new vertex's xy gradient ``g_xy``.  ``clip_faces_np`` places the new vertex at
    ``xy = (xy_a z_a (1 - w) + xy_b z_b w) / z_c`` with ``w = (z_a - z_c) / (z_a - z_b)``; the two weights ``z_a (1 - w) / z_c`` and
    ``s = z_b w / z_c`` sum to one (the crossing's depth is ``z_c``), so ``xy = xy_a + s (xy_b - xy_a)`` and
    ``d xy / d z_a = (xy_b - xy_a) z_b (z_c - z_b) / (z_c (z_a - z_b)^2)``, ``d xy / d z_b = (xy_b - xy_a) z_a (z_a - z_c) / (z_c (z_a - z_b)^2)``:
    a depth only slides the crossing ALONG the edge's line.  (What autograd obtains through the same expressions by the chain rule;
    checked against finite differences in tests/test_oracle_raster.py.)  Returns ``(dz_a, dz_b)``, float64."""
    xa, xb = np.asarray(va[:2], np.float64), np.asarray(vb[:2], np.float64)
    za, zb, zc = float(va[2]), float(vb[2]), float(z_clip)
    ge = float(np.dot(np.asarray(g_xy, np.float64), xb - xa)) / (zc * (za - zb) ** 2)
    return ge * zb * (zc - zb), ge * za * (za - zc)


def _clip_plan(verts_ndc: np.ndarray, faces: np.ndarray):
    """Per image: None (nothing crosses z_clip) or the clipped mesh of ``clip_faces_np``."""
    z = verts_ndc[..., 2]
    plans = []
    for n in range(verts_ndc.shape[0]):
        if _Z_CLIP > 0.0 and bool((z[n] < _Z_CLIP).any()):
            va, fa, src, coef = clip_faces_np(verts_ndc[n], faces, _Z_CLIP)
            plans.append((va, fa, src, coef) if src.shape[0] else None)
        else:
            plans.append(None)
    return plans


def num_threads() -> int:
    return int(_lib().oracle_num_threads())


def _fp(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_float))


def _ip(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_int32))


def silhouette_forward_np(verts_ndc, faces, S, blur=BLUR_RADIUS, sigma=SIGMA, K=FACES_PER_PIXEL,
                          want_fragments=False, _clipped=False):
    verts_ndc = np.ascontiguousarray(verts_ndc, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    if not _clipped and not want_fragments:
        plans = _clip_plan(verts_ndc, faces)
        if any(p is not None for p in plans):  # images with faces across z_clip are rendered one by one from their clipped mesh
            outs = [silhouette_forward_np(verts_ndc[n:n + 1] if p is None else p[0][None], faces if p is None else p[1], S, blur, sigma, K,
                                          _clipped=True) for n, p in enumerate(plans)]
            return np.concatenate([o[0] for o in outs]), np.concatenate([o[1] for o in outs])
    N, V, _ = verts_ndc.shape
    F = faces.shape[0]
    sil = np.empty((N, S, S), np.float32)
    ncand = np.empty((N, S, S), np.int32)
    null_i = ctypes.POINTER(ctypes.c_int32)()
    null_f = ctypes.POINTER(ctypes.c_float)()
    if want_fragments:
        ff = np.empty((N, S, S, K), np.int32)
        fd = np.empty((N, S, S, K), np.float32)
        fz = np.empty((N, S, S, K), np.float32)
        rc = _lib().oracle_silhouette_forward(_fp(verts_ndc), _ip(faces), N, V, F, S, blur, sigma, K,
                                              _fp(sil), _ip(ncand), _ip(ff), _fp(fd), _fp(fz))
        assert rc == 0
        return sil, ncand, ff, fd, fz
    rc = _lib().oracle_silhouette_forward(_fp(verts_ndc), _ip(faces), N, V, F, S, blur, sigma, K,
                                          _fp(sil), _ip(ncand), null_i, null_f, null_f)
    assert rc == 0
    return sil, ncand


def silhouette_backward_np(verts_ndc, faces, S, grad_sil, blur=BLUR_RADIUS, sigma=SIGMA, K=FACES_PER_PIXEL, _clipped=False):
    verts_ndc = np.ascontiguousarray(verts_ndc, np.float32)
    faces = np.ascontiguousarray(faces, np.int32)
    grad_sil = np.ascontiguousarray(grad_sil, np.float32)
    if not _clipped:
        plans = _clip_plan(verts_ndc, faces)
        if any(p is not None for p in plans):
            V0 = verts_ndc.shape[1]
            out = np.zeros((verts_ndc.shape[0], V0, 3), np.float32)
            for n, p in enumerate(plans):
                if p is None:
                    out[n] = silhouette_backward_np(verts_ndc[n:n + 1], faces, S, grad_sil[n:n + 1], blur, sigma, K, _clipped=True)[0]
                    continue
                va, fa, src, coef = p
                g = silhouette_backward_np(va[None], fa, S, grad_sil[n:n + 1], blur, sigma, K, _clipped=True)[0].astype(np.float64)
                acc = g[:V0].copy()
                for j in range(src.shape[0]):  # new vertex -> its two source vertices: their xy, and their depths through the coefficients
                    acc[src[j, 0], :2] += coef[j, 0] * g[V0 + j, :2]
                    acc[src[j, 1], :2] += coef[j, 1] * g[V0 + j, :2]
                    dza, dzb = clip_depth_gradient(va[src[j, 0]], va[src[j, 1]], g[V0 + j, :2], _Z_CLIP)
                    acc[src[j, 0], 2] += dza
                    acc[src[j, 1], 2] += dzb
                out[n] = acc.astype(np.float32)
            return out
    N, V, _ = verts_ndc.shape
    gv = np.empty((N, V, 3), np.float32)
    rc = _lib().oracle_silhouette_backward(_fp(verts_ndc), _ip(faces), N, V, faces.shape[0], S, blur, sigma, K,
                                           _fp(grad_sil), _fp(gv))
    assert rc == 0
    return gv


class SoftSilhouette(torch.autograd.Function):
    """verts_ndc (N,V,3) -> silhouette (N,S,S), differentiable wrt the xy of verts_ndc (and wrt the depth of the end points of edges
    that cross z_clip)."""

    @staticmethod
    def forward(ctx, verts_ndc, faces, S, blur, sigma, K):
        sil, _ = silhouette_forward_np(verts_ndc.detach().numpy(), faces.numpy(), S, blur, sigma, K)
        ctx.save_for_backward(verts_ndc.detach(), faces)
        ctx.cfg = (S, blur, sigma, K)
        return torch.from_numpy(sil)

    @staticmethod
    def backward(ctx, grad_sil):
        verts_ndc, faces = ctx.saved_tensors
        S, blur, sigma, K = ctx.cfg
        gv = silhouette_backward_np(verts_ndc.numpy(), faces.numpy(), S, grad_sil.contiguous().numpy(),
                                    blur, sigma, K)
        return torch.from_numpy(gv), None, None, None, None, None


# --------------------------------------------------------------------------------------
# cameras (pytorch3d.renderer.cameras restated)
# --------------------------------------------------------------------------------------
def look_at_view_transform(dist=1.0, elev=0.0, azim=0.0, degrees=True):
    """R (n,3,3), T (n,3) of a camera looking at the origin, up = +y (pytorch3d convention:
    X_view = X_world @ R + T).  Default Renderer camera: dist 2.7, elev 0, azim 0
    (p3d_renderer.py:34) -> R = diag(-1,1,-1), T = (0,0,2.7)."""
    dist = torch.as_tensor(dist, dtype=torch.float32).reshape(-1)
    elev = torch.as_tensor(elev, dtype=torch.float32).reshape(-1)
    azim = torch.as_tensor(azim, dtype=torch.float32).reshape(-1)
    n = max(dist.numel(), elev.numel(), azim.numel())
    dist, elev, azim = dist.expand(n), elev.expand(n), azim.expand(n)
    if degrees:
        elev = elev * (math.pi / 180.0)
        azim = azim * (math.pi / 180.0)
    x = dist * torch.cos(elev) * torch.sin(azim)
    y = dist * torch.sin(elev)
    z = dist * torch.cos(elev) * torch.cos(azim)
    C = torch.stack([x, y, z], dim=1)  # camera centre in world coordinates
    at = torch.zeros_like(C)
    up = torch.tensor([0.0, 1.0, 0.0]).expand(n, 3)
    z_axis = torch.nn.functional.normalize(at - C, eps=1e-5)
    x_axis = torch.nn.functional.normalize(torch.cross(up, z_axis, dim=1), eps=1e-5)
    y_axis = torch.nn.functional.normalize(torch.cross(z_axis, x_axis, dim=1), eps=1e-5)
    is_close = torch.isclose(x_axis, torch.tensor(0.0), atol=5e-3).all(dim=1, keepdim=True)
    if is_close.any():
        replacement = torch.nn.functional.normalize(torch.cross(y_axis, z_axis, dim=1), eps=1e-5)
        x_axis = torch.where(is_close, replacement, x_axis)
    R = torch.cat((x_axis[:, None, :], y_axis[:, None, :], z_axis[:, None, :]), dim=1).transpose(1, 2)
    T = -torch.bmm(R.transpose(1, 2), C[:, :, None])[:, :, 0]
    return R, T


def _tan_half_fov(fov_deg: torch.Tensor) -> torch.Tensor:
    return torch.tan((fov_deg * (math.pi / 180.0)) / 2)


def project_to_ndc(points: torch.Tensor, R, T, fov_deg, aspect=None):
    """points (N,P,3) world -> (x_ndc, y_ndc, z_view) (N,P,3).  ``MeshRasterizer.transform``:
    view transform first, then K00 = 1/(aspect tan), K11 = 1/tan, divide by view z."""
    fov_deg = fov_deg.reshape(-1)
    t = _tan_half_fov(fov_deg)
    a = torch.ones_like(t) if aspect is None else aspect.reshape(-1)
    view = torch.matmul(points, R) + T[:, None, :]
    # K built as 2 znear / (max - min) like compute_projection_matrix
    max_y = t * ZNEAR
    max_x = max_y * a
    k00 = 2.0 * ZNEAR / (max_x - (-max_x))
    k11 = 2.0 * ZNEAR / (max_y - (-max_y))
    z = view[..., 2]
    x = view[..., 0] * k00[:, None] / z
    y = view[..., 1] * k11[:, None] / z
    return torch.stack([x, y, z], dim=-1)


def project_points_screen(points, R, T, fov_deg, S: int, aspect=None):
    """``cameras.transform_points_screen(points)[..., [1,0]]`` (p3d_renderer.py:137): returns
    (y_s, x_s) pixels with x_s = S/2 - (S/2) x_ndc."""
    ndc = project_to_ndc(points, R, T, fov_deg, aspect)
    xs = S / 2.0 - (S / 2.0) * ndc[..., 0]
    ys = S / 2.0 - (S / 2.0) * ndc[..., 1]
    return torch.stack([ys, xs], dim=-1)


def render_silhouette(verts, faces, R, T, fov_deg, S: int, aspect=None, blur=BLUR_RADIUS, sigma=SIGMA,
                      K=FACES_PER_PIXEL):
    """verts (N,V,3) world, one camera per image -> (N,1,S,S) soft silhouette."""
    ndc = project_to_ndc(verts, R, T, fov_deg, aspect)
    return SoftSilhouette.apply(ndc, faces.to(torch.int32), S, blur, sigma, K)[:, None]


class OracleRenderer:
    """Callable with the reference Renderer's forward signature (p3d_renderer.py:127)."""

    def __init__(self, image_size: int, R=None, T=None, fov=None, aspect=None):
        self.image_size = image_size
        if R is None:
            R, T = look_at_view_transform(2.7, 0.0, 0.0)
        self.R, self.T = R, T
        self.fov = torch.tensor([60.0]) if fov is None else fov
        self.aspect = aspect

    def _expand(self, n):
        R = self.R.expand(n, 3, 3) if self.R.shape[0] != n else self.R
        T = self.T.expand(n, 3) if self.T.shape[0] != n else self.T
        fov = self.fov.reshape(-1)
        fov = fov.expand(n) if fov.shape[0] != n else fov
        a = self.aspect
        if a is not None:
            a = a.reshape(-1)
            a = a.expand(n) if a.shape[0] != n else a
        return R, T, fov, a

    def __call__(self, vertices, points, faces, joints_only=False):
        n = vertices.shape[0]
        R, T, fov, a = self._expand(n)
        proj = project_points_screen(points.float(), R, T, fov, self.image_size, a)
        if joints_only:
            return None, proj
        f = faces[0] if faces.dim() == 3 else faces
        sil = render_silhouette(vertices.float(), f, R, T, fov, self.image_size, a)
        return sil, proj
