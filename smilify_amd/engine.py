"""Thin torch-tensor front end over the C ABI (include/smilfit.h).

PyTorch is used for device memory, streams and autograd plumbing only; every arithmetic step of the hot
path is a HIP kernel inside ``libsmilfit.so``.  All functions launch on ``torch.cuda.current_stream()``.
"""
from __future__ import annotations

import ctypes
import os
from dataclasses import dataclass
from typing import Dict, Optional

import numpy as np
import torch

from . import _lib
from .model_io import SmilModelTables

# Renderer settings of the reference (smal_fitter/p3d_renderer.py:24-25,41-47)
SIGMA = 1e-4
BLUR_RADIUS = float(np.log(1.0 / 1e-4 - 1.0) * 1e-4)
FACES_PER_PIXEL = 100
ZNEAR, ZFAR = 0.001, 1000.0


def _ptr(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _f32(t: Optional[torch.Tensor], device) -> Optional[torch.Tensor]:
    if t is None:
        return None
    return t.detach().to(device=device, dtype=torch.float32).contiguous()


def require_gpu(device) -> torch.device:
    device = torch.device(device)
    if device.type != "cuda":
        raise _lib.SmilError(f"smilify_amd runs on an AMD GPU only (got device '{device}'); there is no CPU path")
    if not torch.cuda.is_available():
        raise _lib.SmilError("no GPU visible to PyTorch-ROCm; smilify_amd has no CPU fallback")
    return device


_SHARED_WS: Dict = {}  # (device type, index) -> the device's rasteriser workspace
_WS_USER: Dict = {}    # (device type, index) -> (model, stream) of the most recent rasteriser call in that workspace


class DeviceModel:
    """Model constants resident on one GPU (``SmilModel*``)."""

    def __init__(self, tables: SmilModelTables, device):
        self.device = require_gpu(device)
        self.tables = tables
        lib = _lib.load()
        t = tables
        arrs = dict(
            v_template=np.ascontiguousarray(t.v_template, np.float32),
            shapedirs=np.ascontiguousarray(t.shapedirs, np.float32),
            faces=np.ascontiguousarray(t.faces, np.int32),
            parents=np.ascontiguousarray(t.parents, np.int32),
            skin_idx=np.ascontiguousarray(t.skin_idx, np.int32),
            skin_w=np.ascontiguousarray(t.skin_w, np.float32),
            jreg_rowptr=np.ascontiguousarray(t.jreg_rowptr, np.int32),
            jreg_col=np.ascontiguousarray(t.jreg_col, np.int32),
            jreg_val=np.ascontiguousarray(t.jreg_val, np.float32),
        )
        if t.static_joints:
            arrs["J_static"] = np.ascontiguousarray(t.J_static, np.float32)
        if t.posedirs is not None:
            if t.posedirs.shape != (9 * (t.J - 1), 3 * t.V):
                raise _lib.SmilError(f"posedirs shape {t.posedirs.shape} != ({9 * (t.J - 1)}, {3 * t.V})")
            arrs["posedirs"] = np.ascontiguousarray(t.posedirs, np.float32)
        d = _lib.ModelDesc()
        d.V, d.F, d.J, d.nB = t.V, t.F, t.J, t.nB
        for k, a in arrs.items():
            setattr(d, k, a.ctypes.data if a.size else None)
        d.static_joints = 1 if t.static_joints else 0
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            _lib.check(lib.smil_model_create(ctypes.byref(d), ctypes.byref(handle)), "smil_model_create")
        self.handle = handle
        self.V, self.F, self.J, self.nB = t.V, t.F, t.J, t.nB
        self.static_joints = bool(t.static_joints)
        self.has_posedirs = t.posedirs is not None
        self._ws: Optional[torch.Tensor] = None

    def __del__(self):
        try:
            if getattr(self, "handle", None):
                _lib.load().smil_model_destroy(self.handle)
                self.handle = None
        except Exception:
            pass

    def faces_i32(self) -> torch.Tensor:
        """The face table as an int32 device tensor (lazily uploaded copy of what smil_model_create received)."""
        if getattr(self, "_faces_dev", None) is None:
            self._faces_dev = torch.from_numpy(np.ascontiguousarray(self.tables.faces, np.int32)).to(self.device)
        return self._faces_dev

    def _ws_key(self):
        return (self.device.type, self.device.index if self.device.index is not None else torch.cuda.current_device())

    def workspace(self, N: int, S: int) -> torch.Tensor:
        """Rasteriser workspace: ONE buffer per device, shared by every model and topology on it (most of it is the scratch
        arena of the resident workgroups, whatever the mesh), grown when a call needs more; every model then points at the
        grown buffer (a captured hipGraph holds the tensor it was captured with itself, ``SMALFitter._capture_step``).
        Every call rewrites what it reads, so models may take turns - in STREAM ORDER: a call on another stream than the
        previous one first waits for everything submitted to that stream (``_claim_workspace``)."""
        need = int(_lib.load().smil_raster_workspace_bytes(self.handle, N, S))
        key = self._ws_key()
        ws = _SHARED_WS.get(key)
        if ws is None or ws.numel() < need:
            ws = _SHARED_WS[key] = torch.empty(need, dtype=torch.uint8, device=self.device)
        self._ws = ws
        return ws

    def _claim_workspace(self, n_last_slice: int) -> None:
        """Book-keeping of a rasteriser call about to be launched: order it behind the previous user of the shared workspace when that
        one ran on another stream, and remember who used the workspace last and with which slice size (``raster_stats`` reads the
        counters at an offset that depends on it)."""
        key = self._ws_key()
        cur = torch.cuda.current_stream(self.device)
        prev = _WS_USER.get(key)
        if prev is not None and prev[1] != cur and not torch.cuda.is_current_stream_capturing():
            cur.wait_stream(prev[1])
        _WS_USER[key] = (self, cur)
        self._last_launch = n_last_slice


@dataclass
class CameraSet:
    """FoV-perspective cameras; tables with k rows are indexed ``image % k`` (k = 1, views or N)."""

    R: torch.Tensor  # (nR,3,3)
    T: torch.Tensor  # (nT,3)
    fov: torch.Tensor  # (nFov,) degrees
    aspect: Optional[torch.Tensor]
    views: int
    S: int

    def struct(self, N: int) -> _lib.Cameras:
        c = _lib.Cameras()
        c.N, c.views, c.S = N, self.views, self.S
        c.R, c.nR = self.R.data_ptr(), self.R.shape[0]
        c.T, c.nT = self.T.data_ptr(), self.T.shape[0]
        c.fov, c.nFov = self.fov.data_ptr(), self.fov.numel()
        if self.aspect is not None:
            c.aspect, c.nAspect = self.aspect.data_ptr(), self.aspect.numel()
        else:
            c.aspect, c.nAspect = None, 0
        for name, k in (("R", c.nR), ("T", c.nT), ("fov", c.nFov)):
            if k not in (1, self.views, N):
                raise _lib.SmilError(f"camera table {name} has {k} rows; expected 1, views={self.views} or N={N}")
        return c


TIE_RULES = {"depth_face_id": 0, "reference_queue": 1}


class ClipDepth:
    """Caller-owned buffers of the rasteriser's depth-gradient side channel (``SmilClipDepth``): the end points of edges that
    cross the clipping plane receive a gradient on their DEPTH (pytorch3d differentiates ``clip_faces`` through its
    interpolation weight), which ``d_ndc (N,V,2)`` cannot hold.  ``silhouette_backward`` / ``silhouette_l1_fused`` fill it
    (``clip_depth=``), ``lbs_backward(ndc_upstream=dict(..., clip_depth=))`` or ``clip_depth_backward`` consume it."""

    def __init__(self, device, n_images: int, capacity: int = 1 << 18):
        dev = torch.device(device)
        self.n_images, self.capacity = int(n_images), int(capacity)
        self.vertex = torch.empty(capacity, dtype=torch.int32, device=dev)  # (entries are written before they are read: range / counter say which)
        self.dz = torch.empty(capacity, dtype=torch.float32, device=dev)
        self.range = torch.zeros(n_images, 2, dtype=torch.int32, device=dev)
        self.counter = torch.zeros(2, dtype=torch.int32, device=dev)
        self._struct = _lib.ClipDepth(self.vertex.data_ptr(), self.dz.data_ptr(), self.range.data_ptr(), self.counter.data_ptr(), self.capacity)

    def pointer(self):
        return ctypes.cast(ctypes.pointer(self._struct), ctypes.c_void_p)

    def dense(self, V: int) -> torch.Tensor:
        """(n_images, V) depth gradients (tests; synchronises)."""
        out = torch.zeros(self.n_images, V, dtype=torch.float64)
        rg, vx, dz = self.range.cpu().numpy(), self.vertex.cpu().numpy(), self.dz.cpu().numpy()
        for n in range(self.n_images):
            for e in range(int(rg[n, 0]), int(rg[n, 0]) + int(rg[n, 1])):
                out[n, int(vx[e])] += float(dz[e])
        return out


def clip_depth_for(model: "DeviceModel", n_images: int) -> ClipDepth:
    """The model's cached ``ClipDepth`` for calls of ``n_images`` images (one per size: a captured graph keeps its pointers).
    The cache is shared by every fitter, renderer and stream that uses this ``DeviceModel``: correctness relies on the produce
    (rasteriser call) -> consume (LBS / projection backward) pair of one evaluation being issued back to back on ONE stream, as
    ``SMALFitter._loss_and_grads`` does; callers that interleave evaluations of the same size on several streams must bring their own
    ``ClipDepth``."""
    cache = model.__dict__.setdefault("_clip_depth_cache", {})
    if n_images not in cache:
        cache[n_images] = ClipDepth(model.device, n_images)
    return cache[n_images]


def _rs_for_slice(rs, clip_depth: Optional["ClipDepth"], image0: int):
    """A copy of the raster settings that names the depth-gradient sink and the slice's first image."""
    if clip_depth is None:
        return rs
    cp = _lib.RasterSettings()
    ctypes.pointer(cp)[0] = rs
    cp.clip_depth, cp.image0 = clip_depth.pointer(), int(image0)
    return cp


def clip_depth_backward(cams: "CameraSet", clip_depth: ClipDepth, d_verts: torch.Tensor) -> None:
    """``d_verts`` (frames,V,3) += the depth gradients of ``clip_depth`` carried through the cameras (separate-kernel route)."""
    N = d_verts.shape[0] * cams.views
    c = cams.struct(N)
    _lib.check(_lib.load().smil_clip_depth_backward(ctypes.byref(c), ctypes.byref(clip_depth._struct), N, d_verts.shape[1], _ptr(d_verts),
                                                    _stream()), "smil_clip_depth_backward")


def raster_settings(blur=BLUR_RADIUS, sigma=SIGMA, K=FACES_PER_PIXEL, tie_rule="depth_face_id") -> _lib.RasterSettings:
    """``tie_rule``: which faces a truncated pixel keeps among equal depths at its K-th place - ``"depth_face_id"`` (default: the
    smallest face ids, order independent) or ``"reference_queue"`` (what pytorch3d's unsorted K-queue keeps when it visits the
    faces in index order, the reference's rasteriser: those pixels are replayed by a second kernel)."""
    rs = _lib.RasterSettings()
    rs.blur_radius, rs.sigma, rs.faces_per_pixel, rs.z_clip = blur, sigma, K, ZNEAR / 2
    rs.tie_rule = TIE_RULES[tie_rule] if isinstance(tie_rule, str) else int(tie_rule)
    return rs


# ----------------------------------------------------------------------------------------------
# LBS
# ----------------------------------------------------------------------------------------------
def lbs_forward(model: DeviceModel, beta, theta, trans=None, logscale=None, btrans=None, del_v=None,
                v_template=None, Rs_in=None, shared_beta=False, logscale_shared=False, btrans_shared=False,
                propagate_scaling=False, allow_limb_scaling=True, trans_after_joints=False, theta_mask=None,
                project: Optional[Dict] = None) -> Dict[str, torch.Tensor]:
    """``theta_mask`` (J,3): the kernels use ``theta * mask`` without a masked copy being made (SMALFitter's rotation masks).
    ``project`` = ``dict(cams=CameraSet, ndc=bool, yx=bool)``: the vertices / joints are also projected through the cameras
    (``smil_lbs_forward_project``: one kernel per frame with skinning and joint regression); the result carries ``ndc`` (N,V,3)
    and / or ``yx`` (N,J,2)."""
    dev = model.device
    B = int((theta if theta is not None else Rs_in).shape[0])
    J, V = model.J, model.V
    nB_used = int(beta.shape[-1])
    nS = 1 if (shared_beta and del_v is None) else B
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    out = dict(v_shaped=f(nS, V, 3), J_rest=f(nS, J, 3), Rs=f(B, J, 3, 3), G=f(B, J, 3, 4), A=f(B, J, 3, 4),
               new_J=f(B, J, 3), verts=f(B, V, 3), joints=f(B, J, 3))
    if model.has_posedirs:
        out["v_posed"] = f(B, V, 3)
    inp = dict(beta=beta, theta=theta, Rs_in=Rs_in, logscale=logscale, btrans=btrans, trans=trans, del_v=del_v,
               v_template=v_template, theta_mask=theta_mask)
    i = _lib.LbsInputs()
    i.B, i.shared_beta, i.nB_used = B, int(shared_beta), nB_used
    i.logscale_shared, i.btrans_shared = int(logscale_shared), int(btrans_shared)
    i.propagate_scaling, i.allow_limb_scaling = int(propagate_scaling), int(allow_limb_scaling)
    i.trans_after_joints = int(trans_after_joints)
    for k, t in inp.items():
        setattr(i, k, None if t is None else t.data_ptr())
    o = _lib.LbsOutputs()
    for k, t in out.items():
        setattr(o, k, t.data_ptr())
    ndc = yx = None
    if project is not None:
        cams = project["cams"]
        N = B * cams.views
        ndc = f(N, V, 3) if project.get("ndc", True) else None
        yx = f(N, J, 2) if project.get("yx", True) else None
        c = cams.struct(N)
        _lib.check(_lib.load().smil_lbs_forward_project(model.handle, ctypes.byref(i), ctypes.byref(o), ctypes.byref(c), _ptr(ndc), _ptr(yx), _stream()),
                   "smil_lbs_forward_project")
    else:
        _lib.check(_lib.load().smil_lbs_forward(model.handle, ctypes.byref(i), ctypes.byref(o), _stream()), "smil_lbs_forward")
    if ndc is not None:
        out["ndc"] = ndc
    if yx is not None:
        out["yx"] = yx
    out["_inputs"] = inp
    out["_flags"] = dict(B=B, shared_beta=shared_beta, nB_used=nB_used, logscale_shared=logscale_shared,
                         btrans_shared=btrans_shared, propagate_scaling=propagate_scaling,
                         allow_limb_scaling=allow_limb_scaling, trans_after_joints=trans_after_joints)
    return out


# SMILFIT_UNFUSED_LBS=1 (A/B measurements, tools/dbg/ab_lbs.sh): the fit iteration takes the separate projection / skinning kernels
FUSED_LBS_FORWARD = os.environ.get("SMILFIT_UNFUSED_LBS") != "1"   # projection inside the skinning kernel (smil_lbs_forward_project)
FUSED_LBS_BACKWARD = os.environ.get("SMILFIT_UNFUSED_LBS") != "1"  # the fit iteration takes smil_lbs_backward_ndc where the library supports the model (tests switch it off to compare)


def lbs_backward_ndc_supported(model: DeviceModel, nB_used: int, views: int) -> bool:
    """Whether ``lbs_backward(..., ndc_upstream=...)`` (one kernel from the image plane) handles this model and call."""
    return bool(_lib.load().smil_lbs_backward_ndc_supported(model.handle, int(nB_used), int(views)))


def lbs_backward(model: DeviceModel, saved: Dict, d_verts, d_joints, need_beta=True, need_theta=True,
                 need_logscale=True, need_btrans=True, need_trans=True, need_vshaped=False,
                 need_Rs=False, d_beta_accum: Optional[torch.Tensor] = None, out_logscale: Optional[torch.Tensor] = None,
                 out_btrans: Optional[torch.Tensor] = None, ndc_upstream: Optional[Dict] = None, up_Rs: Optional[torch.Tensor] = None,
                 up_v_shaped: Optional[torch.Tensor] = None) -> Dict[str, Optional[torch.Tensor]]:
    """``d_beta_accum`` (shared betas only): the sum over frames is ADDED to this (nB,) tensor instead of a fresh one.
    ``out_logscale`` / ``out_btrans`` (shared tables only): (J,3) buffers that receive those gradients (overwritten).
    ``ndc_upstream`` = ``dict(cams=CameraSet, d_ndc=, d_ndc_scale=, d_yx=, d_fov_img=)`` instead of ``d_verts`` / ``d_joints``:
    the gradients are taken on the image plane and projected back inside the skinning backward (``smil_lbs_backward_ndc``);
    the result then carries ``d_joints`` (B,J,3).
    ``up_Rs`` (B,J,3,3) / ``up_v_shaped`` (nS,V,3): upstream gradients on the returned rotation matrices / shaped vertices."""
    dev = model.device
    inp, fl = saved["_inputs"], saved["_flags"]
    B, J = fl["B"], model.J
    f = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)  # noqa: E731
    g = dict(d_beta=None, d_theta=None, d_logscale=None, d_btrans=None, d_trans=None, d_del_v=None, d_Rs_in=None)
    if need_vshaped:  # per-frame gradient on v_shaped = on del_v (and, summed over frames, on a custom v_template)
        g["d_del_v"] = f(B, model.V, 3)
    if need_Rs and inp["Rs_in"] is not None:
        g["d_Rs_in"] = f(B, J, 3, 3)
    accumulate_beta = False
    if need_beta and fl["nB_used"] > 0:
        if fl["shared_beta"] and d_beta_accum is not None:
            g["d_beta"], accumulate_beta = d_beta_accum, True
        else:
            g["d_beta"] = f(fl["nB_used"]) if fl["shared_beta"] else f(B, fl["nB_used"])
    if need_theta and inp["theta"] is not None:
        g["d_theta"] = f(B, J, 3)
    if need_logscale and inp["logscale"] is not None and fl["allow_limb_scaling"]:
        g["d_logscale"] = (out_logscale if out_logscale is not None else f(J, 3)) if fl["logscale_shared"] else f(B, J, 3)
    if need_btrans and inp["btrans"] is not None:
        g["d_btrans"] = (out_btrans if out_btrans is not None else f(J, 3)) if fl["btrans_shared"] else f(B, J, 3)
    if need_trans:
        g["d_trans"] = f(B, 3)
    scratch = dict(d_A=f(B, J, 12), d_Jrest=f(B, J, 3), d_Rs=f(B, J, 9))
    if g["d_beta"] is not None and fl["shared_beta"]:
        scratch["beta_rows"] = f(2 * B * fl["nB_used"] + 16)  # per-block partial sums of the shared shape gradient (added in a fixed order) + the call's block counter
    i = _lib.LbsInputs()
    i.B, i.shared_beta, i.nB_used = B, int(fl["shared_beta"]), fl["nB_used"]
    i.logscale_shared, i.btrans_shared = int(fl["logscale_shared"]), int(fl["btrans_shared"])
    i.propagate_scaling, i.allow_limb_scaling = int(fl["propagate_scaling"]), int(fl["allow_limb_scaling"])
    i.trans_after_joints = int(fl["trans_after_joints"])
    for k, t in inp.items():
        setattr(i, k, None if t is None else t.data_ptr())
    o = _lib.LbsOutputs()
    for k in ("v_shaped", "J_rest", "Rs", "G", "A", "new_J", "verts", "joints"):
        setattr(o, k, saved[k].data_ptr())
    if model.has_posedirs:
        o.v_posed = saved["v_posed"].data_ptr()
        scratch["d_vposed"] = f(B, model.V, 3)
        scratch["d_posefeat"] = f(B, 9 * (J - 1))
    gs = _lib.LbsGrads()
    gs.d_verts = None if d_verts is None else d_verts.data_ptr()
    gs.d_joints = None if d_joints is None else d_joints.data_ptr()
    for k, t in {**g, **scratch}.items():
        setattr(gs, k, None if t is None else t.data_ptr())
    gs.accumulate_shared_beta = int(accumulate_beta)
    gs.up_Rs = None if up_Rs is None else up_Rs.data_ptr()
    gs.up_v_shaped = None if up_v_shaped is None else up_v_shaped.data_ptr()
    if ndc_upstream is not None:
        if d_verts is not None or d_joints is not None or need_vshaped:
            raise ValueError("ndc_upstream replaces d_verts / d_joints and has no del_v gradient")
        up = ndc_upstream
        cams = up["cams"]
        g["d_joints"] = f(B, J, 3)
        if up.get("clip_depth") is not None:
            gs.clip_depth = up["clip_depth"].pointer()
        c = cams.struct(B * cams.views)
        _lib.check(_lib.load().smil_lbs_backward_ndc(model.handle, ctypes.byref(i), ctypes.byref(o), ctypes.byref(gs), ctypes.byref(c),
                                                     _ptr(up.get("d_ndc")), _ptr(up.get("d_ndc_scale")), _ptr(up.get("d_yx")),
                                                     _ptr(g["d_joints"]), _ptr(up.get("d_fov_img")), _stream()),
                   "smil_lbs_backward_ndc")
        return g
    _lib.check(_lib.load().smil_lbs_backward(model.handle, ctypes.byref(i), ctypes.byref(o), ctypes.byref(gs), _stream()),
               "smil_lbs_backward")
    return g


# ----------------------------------------------------------------------------------------------
# projection
# ----------------------------------------------------------------------------------------------
def project(cams: CameraSet, pts: torch.Tensor, want_ndc=True, want_yx=True):
    frames, P = pts.shape[0], pts.shape[1]
    N = frames * cams.views
    ndc = torch.empty(N, P, 3, dtype=torch.float32, device=pts.device) if want_ndc else None
    yx = torch.empty(N, P, 2, dtype=torch.float32, device=pts.device) if want_yx else None
    c = cams.struct(N)
    _lib.check(_lib.load().smil_project(ctypes.byref(c), _ptr(pts), P, _ptr(ndc), _ptr(yx), _stream()), "smil_project")
    return ndc, yx


def project_backward(cams: CameraSet, pts: torch.Tensor, d_ndc=None, d_yx=None, d_pts=None, d_fov_img=None,
                     accumulate=False, d_ndc_scale=None):
    frames, P = pts.shape[0], pts.shape[1]
    N = frames * cams.views
    if d_pts is None:
        d_pts = torch.empty_like(pts)
        accumulate = False
    if d_fov_img is None:
        d_fov_img = torch.zeros(N, dtype=torch.float32, device=pts.device)
    c = cams.struct(N)
    _lib.check(_lib.load().smil_project_backward(ctypes.byref(c), _ptr(pts), P, _ptr(d_ndc), _ptr(d_yx), _ptr(d_pts),
                                                 _ptr(d_fov_img), int(accumulate), _ptr(d_ndc_scale), _stream()), "smil_project_backward")
    return d_pts, d_fov_img


def project_verts_and_joints(cams: CameraSet, verts: torch.Tensor, joints: torch.Tensor):
    """One launch: verts (frames,V,3) -> NDC (N,V,3) for the rasteriser, joints (frames,J,3) -> (y,x) pixels (N,J,2)."""
    frames, V, J = verts.shape[0], verts.shape[1], joints.shape[1]
    N = frames * cams.views
    ndc = torch.empty(N, V, 3, dtype=torch.float32, device=verts.device)
    yx = torch.empty(N, J, 2, dtype=torch.float32, device=verts.device)
    c = cams.struct(N)
    _lib.check(_lib.load().smil_project2(ctypes.byref(c), _ptr(verts), V, _ptr(ndc), None, _ptr(joints), J, None, _ptr(yx), _stream()),
               "smil_project2")
    return ndc, yx


def project_backward_verts_and_joints(cams: CameraSet, verts, d_ndc, joints, d_yx, d_fov_img, d_ndc_scale=None):
    """One launch: the backward of ``project_verts_and_joints``; returns (d_verts, d_joints), adds to d_fov_img.
    ``d_ndc_scale``: the decode factors of a ``d_ndc`` the fused rasteriser left packed (``silhouette_l1_fused(packed_out=True)``)."""
    V, J = verts.shape[1], joints.shape[1]
    N = verts.shape[0] * cams.views
    d_verts, d_joints = torch.empty_like(verts), torch.empty_like(joints)
    c = cams.struct(N)
    _lib.check(_lib.load().smil_project_backward2(ctypes.byref(c), _ptr(verts), V, _ptr(d_ndc), None, _ptr(d_verts), _ptr(joints), J, None,
                                                  _ptr(d_yx), _ptr(d_joints), _ptr(d_fov_img), _ptr(d_ndc_scale), _stream()),
               "smil_project_backward2")
    return d_verts, d_joints


def fit_epilogue(cfg, pose, trans, betas, mean_betas, betas_prec, mask, objs, d_pose, d_trans, d_betas, halo_prev=None, halo_next=None,
                 accumulate=True, loss_img=None, pix_scale=None, cams: Optional[CameraSet] = None, d_fov_img=None, d_fov=None):
    """prior_losses + sil_objective + fov_reduce in one launch (the tail of a fit iteration)."""
    n_img = 0 if loss_img is None else loss_img.numel()
    c = None if cams is None else cams.struct(d_fov_img.numel())
    _lib.check(_lib.load().smil_fit_epilogue(ctypes.byref(cfg), _ptr(pose), _ptr(trans), _ptr(betas), _ptr(mean_betas), _ptr(betas_prec),
                                             _ptr(mask), _ptr(halo_prev), _ptr(halo_next), _ptr(objs), _ptr(d_pose), _ptr(d_trans),
                                             _ptr(d_betas), int(accumulate), _ptr(loss_img), _ptr(pix_scale), n_img,
                                             None if c is None else ctypes.byref(c), _ptr(d_fov_img), _ptr(d_fov), _stream()),
               "smil_fit_epilogue")


def fov_reduce(cams: CameraSet, d_fov_img: torch.Tensor) -> torch.Tensor:
    N = d_fov_img.numel()
    d_fov = torch.empty(cams.fov.numel(), dtype=torch.float32, device=d_fov_img.device)
    c = cams.struct(N)
    _lib.check(_lib.load().smil_fov_reduce(ctypes.byref(c), _ptr(d_fov_img), _ptr(d_fov), _stream()), "smil_fov_reduce")
    return d_fov


# ----------------------------------------------------------------------------------------------
# silhouette
# ----------------------------------------------------------------------------------------------
# The per-image tables of one rasteriser launch (tile boxes, depth ranges, work lists: ~12 F + 8 tiles bytes per image) are
# part of the workspace; launches are cut into slices of this many images so that the workspace stays bounded at cfg5
# scale (147 k images per GPU).  Each slice still holds millions of tiles.
MAX_IMAGES_PER_LAUNCH = 16384
MAX_WORKSPACE_BYTES = 24 << 30  # the per-image tables (incl. the binned tile lists, ~96-192 bytes per face) shrink the slice until this holds


def _slice_images(model: "DeviceModel", N: int, S: int) -> int:
    """Images per rasteriser call: at most MAX_IMAGES_PER_LAUNCH, fewer when the per-image workspace tables would push the
    workspace past MAX_WORKSPACE_BYTES (mouse-sized meshes at 512^2)."""
    step = min(N, MAX_IMAGES_PER_LAUNCH)
    key = (N, S, step)
    cached = model.__dict__.setdefault("_slice_cache", {})
    if key not in cached:
        while step > 256 and int(_lib.load().smil_raster_workspace_bytes(model.handle, step, S)) > MAX_WORKSPACE_BYTES:
            step = (step + 1) // 2
        cached[key] = step
    return cached[key]


def _slices(N: int, step: int = MAX_IMAGES_PER_LAUNCH):
    for n0 in range(0, N, step):
        yield n0, min(N, n0 + step)


def raster_stats(model: DeviceModel, N: int) -> dict:
    """Counters of the most recent rasteriser call of ``model`` (of its last slice when the batch was cut into several launches):
    faces straddling z_clip, touched tiles, faces beyond the clip tables, pixels replayed through the reference's queue (``tie_rule``).  ``N`` is ignored (kept for callers of round 3): the slice
    size is recorded at launch time.  Zeros when the model has not rasterised since the workspace was last used by another model.
    Synchronises."""
    zero = {"straddling_faces": 0, "tiles": 0, "unclipped_faces": 0, "tie_pixels": 0}
    user = _WS_USER.get(model._ws_key())
    last = model.__dict__.get("_last_launch")
    if model._ws is None or last is None or user is None or user[0] is not model:
        return zero  # this model has not rasterised, or another model / topology has used the shared workspace since
    out = (ctypes.c_uint32 * 4)()
    _lib.check(_lib.load().smil_raster_stats(model.handle, last, _ptr(model._ws), _stream(), out), "smil_raster_stats")
    return {"straddling_faces": int(out[0]), "tiles": int(out[1]), "unclipped_faces": int(out[2]), "tie_pixels": int(out[3])}


def silhouette_forward(model: DeviceModel, verts_ndc: torch.Tensor, S: int, rs=None) -> torch.Tensor:
    rs = rs or raster_settings()
    N = verts_ndc.shape[0]
    sil = torch.empty(N, S, S, dtype=torch.float32, device=verts_ndc.device)
    step = model._last_slice = _slice_images(model, N, S)
    ws = model.workspace(step, S)
    model._claim_workspace(N - ((N - 1) // step) * step)
    for n0, n1 in _slices(N, step):
        _lib.check(_lib.load().smil_silhouette_forward(model.handle, _ptr(verts_ndc[n0:n1]), n1 - n0, S, ctypes.byref(rs),
                                                       _ptr(sil[n0:n1]), _ptr(ws), _stream()), "smil_silhouette_forward")
    return sil


def silhouette_backward(model: DeviceModel, verts_ndc: torch.Tensor, S: int, grad_sil: torch.Tensor, rs=None,
                        clip_depth: Optional[ClipDepth] = None) -> torch.Tensor:
    """``clip_depth``: receives the depth gradients of the end points of edges that cross the clipping plane (``ClipDepth``)."""
    rs = rs or raster_settings()
    N = verts_ndc.shape[0]
    d_ndc = torch.empty(N, model.V, 2, dtype=torch.float32, device=verts_ndc.device)
    step = model._last_slice = _slice_images(model, N, S)
    ws = model.workspace(step, S)
    model._claim_workspace(N - ((N - 1) // step) * step)
    for n0, n1 in _slices(N, step):
        _lib.check(_lib.load().smil_silhouette_backward(model.handle, _ptr(verts_ndc[n0:n1]), n1 - n0, S, ctypes.byref(_rs_for_slice(rs, clip_depth, n0)),
                                                        _ptr(grad_sil[n0:n1]), _ptr(d_ndc[n0:n1]), _ptr(ws), _stream()),
                   "smil_silhouette_backward")
    return d_ndc


def silhouette_l1_fused(model: DeviceModel, verts_ndc, S, target, target_sum, pix_scale, rs=None, want_sil=False,
                        loss_img=None, d_ndc=None, packed_out=False, clip_depth: Optional[ClipDepth] = None):
    """Fused soft silhouette + L1 + backward.  Returns (loss_img, d_ndc, sil), or with ``packed_out`` (loss_img, d_ndc, sil,
    d_ndc_scale): ``d_ndc`` as the kernel accumulated it (64-bit packed fixed point for large batches) plus the per-image
    decode factors ``project_backward`` takes - this saves the decode pass over the whole gradient."""
    rs = rs or raster_settings()
    N = verts_ndc.shape[0]
    dev = verts_ndc.device
    if loss_img is None:
        loss_img = torch.empty(N, dtype=torch.float32, device=dev)
    if d_ndc is None:
        d_ndc = torch.empty(N, model.V, 2, dtype=torch.float32, device=dev)
    sil = torch.empty(N, S, S, dtype=torch.float32, device=dev) if want_sil else None
    scale = torch.empty(N, dtype=torch.float32, device=dev) if packed_out else None
    step = model._last_slice = _slice_images(model, N, S)
    ws = model.workspace(step, S)
    model._claim_workspace(N - ((N - 1) // step) * step)
    if target.dtype not in (torch.float32, torch.uint8):
        raise _lib.SmilError(f"target silhouettes must be float32 or uint8, got {target.dtype}")
    for n0, n1 in _slices(N, step):
        _lib.check(_lib.load().smil_silhouette_l1_fused(
            model.handle, _ptr(verts_ndc[n0:n1]), n1 - n0, S, ctypes.byref(_rs_for_slice(rs, clip_depth, n0)), _ptr(target[n0:n1]), int(target.dtype == torch.uint8),
            _ptr(target_sum[n0:n1]), _ptr(pix_scale[n0:n1]), _ptr(loss_img[n0:n1]), _ptr(d_ndc[n0:n1]),
            _ptr(None if sil is None else sil[n0:n1]), _ptr(None if scale is None else scale[n0:n1]), _ptr(ws), _stream()),
            "smil_silhouette_l1_fused")
    return (loss_img, d_ndc, sil, scale) if packed_out else (loss_img, d_ndc, sil)


def image_abs_sum(images: torch.Tensor) -> torch.Tensor:
    N = images.shape[0]
    pixels = images[0].numel()
    out = torch.empty(N, dtype=torch.float32, device=images.device)
    _lib.check(_lib.load().smil_image_abs_sum(_ptr(images), int(images.dtype == torch.uint8), N, pixels, _ptr(out), _stream()),
               "smil_image_abs_sum")
    return out


# ----------------------------------------------------------------------------------------------
# losses / optimiser
# ----------------------------------------------------------------------------------------------
def fit_config(N, J, nB, window, weights, w_temp=0.0, frame0=0, N_total=None, limit=0.01, train_global=True,
               train_joints=True, train_trans=True) -> _lib.FitConfig:
    """weights in the reference order (fitter.py:238): w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay."""
    c = _lib.FitConfig()
    c.train_global, c.train_joints, c.train_trans = int(train_global), int(train_joints), int(train_trans)
    c.N, c.J, c.nB, c.window, c.frame0 = N, J, nB, window, frame0
    c.N_total = N if N_total is None else N_total
    c.w_j2d, c.w_reproj, c.w_betas, c.w_pose, c.w_limit, c.w_splay = [float(w) for w in weights]
    c.w_temp, c.limit = float(w_temp), float(limit)
    return c


def pix_scale(cfg: _lib.FitConfig, views: int, S: int, device) -> torch.Tensor:
    out = torch.empty(cfg.N * views, dtype=torch.float32, device=device)
    _lib.check(_lib.load().smil_pix_scale(ctypes.byref(cfg), views, S, _ptr(out), _stream()), "smil_pix_scale")
    return out


def prior_losses(cfg, pose, trans, betas, mean_betas, betas_prec, mask, objs, d_pose, d_trans, d_betas, halo_prev=None,
                 halo_next=None, accumulate=True):
    """pose (N,J,3) = [global_rotation ; joint_rotations], mask (J,3) = [global_mask ; rotation_mask]."""
    _lib.check(_lib.load().smil_prior_losses(ctypes.byref(cfg), _ptr(pose), _ptr(trans), _ptr(betas), _ptr(mean_betas),
                                             _ptr(betas_prec), _ptr(mask), _ptr(halo_prev), _ptr(halo_next), _ptr(objs),
                                             _ptr(d_pose), _ptr(d_trans), _ptr(d_betas), int(accumulate), _stream()),
               "smil_prior_losses")


def mask_rows(x: torch.Tensor, mask: torch.Tensor, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    cols = mask.numel()
    out = torch.empty_like(x) if out is None else out
    _lib.check(_lib.load().smil_mask_rows(_ptr(x), _ptr(mask), x.numel() // cols, cols, _ptr(out), _stream()), "smil_mask_rows")
    return out


def joint_loss(cfg, views, Jc, canon, proj, target, visibility, objs, d_proj):
    _lib.check(_lib.load().smil_joint_loss(ctypes.byref(cfg), views, Jc, _ptr(canon), _ptr(proj), _ptr(target),
                                           _ptr(visibility), _ptr(objs), _ptr(d_proj), _stream()), "smil_joint_loss")


def sil_objective(loss_img, pscale, objs):
    _lib.check(_lib.load().smil_sil_objective(_ptr(loss_img), _ptr(pscale), loss_img.numel(), _ptr(objs), _stream()),
               "smil_sil_objective")


def window_terms(cfg, views, Jc, canon, proj, target, visibility, pose, mask, objs_total, loss_img, pscale) -> torch.Tensor:
    """(windows of this shard, 6) = [joint, limit, pose, splay, betas, sil_reproj] of every window, from the buffers of ONE
    whole-batch iteration (``smil_window_terms``): what the reference's per-window ``forward`` calls of an epoch return."""
    w = cfg.window if cfg.window > 0 else cfg.N_total
    n_win = (cfg.N + w - 1) // w
    out = torch.empty(n_win, 6, dtype=torch.float32, device=pose.device)
    _lib.check(_lib.load().smil_window_terms(ctypes.byref(cfg), views, Jc, _ptr(canon), _ptr(proj), _ptr(target), _ptr(visibility), _ptr(pose),
                                             _ptr(mask), _ptr(objs_total), _ptr(loss_img), _ptr(pscale), _ptr(out), n_win, _stream()),
               "smil_window_terms")
    return out


def adam_step(param, grad, exp_avg, exp_avg_sq, lr, step, beta1=0.5, beta2=0.999, eps=1e-8):
    _lib.check(_lib.load().smil_adam_step(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, beta1,
                                          beta2, eps, step, _stream()), "smil_adam_step")


def adam_step_multi(items, beta1=0.5, beta2=0.999, eps=1e-8):
    """One launch for several tensors: ``items`` = [(param, grad, exp_avg, exp_avg_sq, lr, step), ...]."""
    for k0 in range(0, len(items), _lib.ADAM_MAX_TENSORS):
        chunk = items[k0:k0 + _lib.ADAM_MAX_TENSORS]
        arr = (_lib.AdamTensor * len(chunk))()
        for a, (p, g, m, v, lr, step) in zip(arr, chunk):
            a.param, a.grad, a.exp_avg, a.exp_avg_sq = p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr()
            a.n, a.lr, a.step = p.numel(), float(lr), int(step)
        _lib.check(_lib.load().smil_adam_step_multi(arr, len(chunk), beta1, beta2, eps, _stream()), "smil_adam_step_multi")


def adam_step_dev(param, grad, exp_avg, exp_avg_sq, lr, step_dev, step_offset=0, beta1=0.5, beta2=0.999, eps=1e-8):
    """Adam update whose step count is ``step_dev[0] - step_offset`` (int32 device tensor): capturable in a hipGraph."""
    _lib.check(_lib.load().smil_adam_step_dev(_ptr(param), _ptr(grad), _ptr(exp_avg), _ptr(exp_avg_sq), param.numel(), lr, beta1,
                                              beta2, eps, _ptr(step_dev), int(step_offset), _stream()), "smil_adam_step_dev")


def profile_enable(on: bool) -> None:
    _lib.check(_lib.load().smil_profile_enable(int(on)), "smil_profile_enable")


def profile_read():
    """(summed tile-kernel milliseconds, launches) since the last enable/read."""
    ms, n = ctypes.c_float(), ctypes.c_int32()
    _lib.check(_lib.load().smil_profile_read(ctypes.byref(ms), ctypes.byref(n)), "smil_profile_read")
    return float(ms.value), int(n.value)
