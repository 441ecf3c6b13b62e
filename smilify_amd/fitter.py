"""Drop-in ``SMALFitter`` plus the fused whole-sequence fit step.

``SMALFitter`` keeps the reference's constructor, parameter names and ``forward(batch_range, weights,
stage_id) -> (loss, objs)`` / ``get_temporal`` / ``load_checkpoint`` contract
(reference smal_fitter/fitter.py:55-371).  Every arithmetic step - LBS, projection, soft silhouette, the six
loss terms and all their gradients - runs in HIP kernels of ``libsmilfit.so``; autograd only sees one node
per call whose backward hands out the gradients the kernels already produced.

``SMALFitter.fit_step`` is the fast path used by ``smilify_amd.optimize`` and ``bench.py``: one call = one
epoch of reference optimize_to_joints.py:147-175 over all frames of this rank (sum over windows of window
means + temporal terms, backward, Adam step), with no autograd graph and no per-window Python.

Extensions over the reference, needed for the batched / multi-view configurations:
``fov`` may have shape (1,), (views,) or (N*views,); ``log_beta_scales`` / ``betas_trans`` may be
(1,J,3) shared or (N,J,3) per frame; cameras may be given per view; targets are (N*views, ...).
"""
from __future__ import annotations

import math
import os
from collections import Counter
import pickle as pkl
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn as nn

from . import config as _config
from . import engine, model_io
from ._lib import N_OBJS
from .cameras import FoVCameras
from .p3d_renderer import Renderer
from .smal_torch import SMAL

OBJ_NAMES = ("joint", "limit", "pose", "splay", "betas", "sil_reproj")  # reference objs keys, in objs[] order


def default_global_rotation() -> np.ndarray:
    """``eul_to_axis([-pi/2, 0, -pi/2])`` of reference fitter.py:206: nibabel's euler2angle_axis(z, y, x)
    composes R_x(x) R_y(y) R_z(z); at y = 0 the quaternion is (cx cz, sx cz, -sx sz, cx sz)."""
    hx = hz = -math.pi / 4
    q = np.array([math.cos(hx) * math.cos(hz), math.sin(hx) * math.cos(hz), -math.sin(hx) * math.sin(hz), math.cos(hx) * math.sin(hz)])
    angle = 2.0 * math.acos(q[0])
    return (q[1:] / np.linalg.norm(q[1:]) * angle).astype(np.float32)


def shape_prior_precision(shape_cov, n_betas: int) -> np.ndarray:
    """Cholesky factor of the regularised inverse shape covariance (reference fitter.py:170-175)."""
    cov = np.eye(n_betas) if shape_cov is None else np.asarray(shape_cov, np.float64)
    prec = np.linalg.cholesky(np.linalg.inv(cov + 1e-5 * np.eye(cov.shape[0])))
    return prec[:n_betas, :n_betas].astype(np.float32)


class _Prior:
    """Identity-precision pose prior (reference fitter.py:25-52); kept for API parity, evaluated in-kernel."""

    def __init__(self, n_joints: int, device):
        self.use_ind = np.ones(n_joints * 3, dtype=bool)
        self.use_ind[:3] = False
        self.use_ind_tch = torch.from_numpy(self.use_ind).float().to(device)


class _WindowLoss(torch.Tensor):
    """The scalar loss ``forward`` returns.  The reference's driver does ``acc_loss += loss.mean()`` once per window
    (optimize_to_joints.py:156): on a 0-dim tensor ``mean()`` is the identity, but as a torch op it is a kernel launch and an autograd
    node per window, forward and backward - a quarter of the host time of an epoch of the loop at 52 windows.  Here it returns the tensor
    itself (same value, same gradient); every other operation gives a plain ``torch.Tensor`` at plain-tensor cost."""

    __torch_function__ = torch._C._disabled_torch_function_impl

    def mean(self, *args, **kwargs):
        return self


class _FitWindow(torch.autograd.Function):
    """(loss, objs) of one window; gradients were computed by the kernels in forward."""

    @staticmethod
    def forward(ctx, fitter, frames, weights, w_temp, betas, log_beta_scales, betas_trans, pose, trans, fov):
        objs, grads = fitter._loss_and_grads(frames, weights, w_temp)
        ctx.grads = grads
        total = objs[:9].sum()
        return total.as_subclass(_WindowLoss), objs.clone()

    @staticmethod
    def backward(ctx, g_total, _g_objs):
        g = ctx.grads
        s = lambda t: None if t is None else t * g_total  # noqa: E731
        return (None, None, None, None, s(g["betas"]), s(g["log_beta_scales"]), s(g["betas_trans"]), s(g["pose"]), s(g["trans"]),
                s(g["fov"]))


_WEIGHT_OF = {"joint": 0, "sil_reproj": 1, "betas": 2, "pose": 3, "limit": 4, "splay": 5}  # position of a term's weight (fitter.py:238)


class _EpochEval(torch.autograd.Function):
    """Every window of an epoch in ONE evaluation: one loss scalar per window + the terms (windows, 6).  The per-window ``forward``
    calls of the reference's driver hand these out; autograd brings every window's upstream gradient to ONE backward per epoch (each
    window is an output of this node: no select / scatter nodes in between).  With the same upstream gradient on every window (the
    driver adds the window losses with weight 1) the whole-batch gradients are the answer; windows whose upstream gradient differs
    (left out, weighted differently) are handled exactly - see ``backward``."""

    @staticmethod
    def forward(ctx, fitter, weights, window, betas, log_beta_scales, betas_trans, pose, trans, fov):
        _, grads = fitter._loss_and_grads(None, weights, 0.0, window=window, window_terms=True)
        objs_win = grads.pop("_objs_win")
        ctx.fitter, ctx.weights, ctx.window, ctx.grads = fitter, weights, window, grads
        ctx.key = fitter._state_key(tuple(weights))
        ctx.set_materialize_grads(False)  # (a window nobody used arrives as None, not as a zero tensor)
        ctx.mark_non_differentiable(objs_win)
        return (*(t_.as_subclass(_WindowLoss) for t_ in objs_win.sum(1).unbind(0)), objs_win)

    @staticmethod
    def backward(ctx, *upstream):
        # sum_j g_j G_j.  Rows of per-frame parameters belong to one window each: they are scaled by their window's upstream value
        # (exact, also for a window left out: its rows are exactly zero).  Shared parameters (betas, a shared fov or scale table):
        # c G_total + sum_{g_j != c} (g_j - c) G_j with c the most frequent upstream value - nothing to correct in the driver's loop,
        # one direct evaluation per deviating window otherwise (the window an epoch's first call evaluated on its own, ...).
        f, W = ctx.fitter, ctx.window
        N, views = f.num_images, f.views
        g_win = upstream[:-1]
        used = [t for t in g_win if t is not None]
        if not used:
            return (None,) * 9
        it = iter(torch.stack(used).tolist())  # (the one host sync of an epoch's backward)
        vals = [0.0 if t is None else next(it) for t in g_win]
        c = Counter(vals).most_common(1)[0][0]
        deviating = [(j, gj) for j, gj in enumerate(vals) if gj != c]
        if deviating:
            g_losses = torch.tensor(vals, dtype=torch.float32, device=f.device)
            per_row = lambda t, rep: t * g_losses.repeat_interleave(rep)[:t.shape[0]].reshape((-1,) + (1,) * (t.dim() - 1))  # noqa: E731
        out, shared = {}, []
        for k, v in ctx.grads.items():
            if v is None:
                out[k] = None
            elif not deviating:
                out[k] = v if c == 1.0 else v * c
            elif not f._is_shared(k):
                out[k] = per_row(v, W * views if k == "fov" else W)
            else:
                out[k] = v * c
                shared.append(k)
        if deviating and shared:
            if f._state_key(tuple(ctx.weights)) != ctx.key:
                raise RuntimeError("SMALFitter: backward() through window losses after the parameters, targets or cameras they were "
                                   "evaluated with have changed")
            for j, gj in deviating:
                _, gw = f._loss_and_grads(list(range(j * W, min(N, (j + 1) * W))), ctx.weights, 0.0)
                for k in shared:
                    out[k] = out[k] + gw[k] * (gj - c)
        return (None, None, None, out["betas"], out["log_beta_scales"], out["betas_trans"], out["pose"], out["trans"], out["fov"])


class SMALFitter(nn.Module):
    epoch_cache = True  # forward() may serve the windows of an epoch from one whole-batch evaluation (False: every call on its own)

    def __init__(self, device, data_batch, batch_size, shape_family=-1, use_unity_prior=False, rgb_only=False, *,
                 tables: Optional[model_io.SmilModelTables] = None, model_path: Optional[str] = None,
                 config: Optional[_config.FitterConfig] = None, views: int = 1, frame0: int = 0, n_frames_total: Optional[int] = None):
        super().__init__()
        if use_unity_prior or shape_family != -1:
            raise NotImplementedError("the Unity / shape-family priors need MPI-licensed files that SMIL models do not use")
        self.device = engine.require_gpu(device)
        dev = self.device
        if tables is None:
            cfg0 = config or _config.current
            path = model_path or (cfg0.SMAL_FILE if cfg0 is not None else None)
            if path is None:
                raise ValueError("SMALFitter needs tables=, model_path= or config.SMAL_FILE")
            tables = model_io.load_model(path)
        self.config = config or _config.current or _config.FitterConfig.from_tables(tables, model_path)
        cfg = self.config
        self.rgb_only = rgb_only
        self.views = int(views)
        J, nB = tables.J, tables.nB
        if rgb_only:
            self.rgb_imgs = data_batch
            n_img = self.rgb_imgs.shape[0]
            self.sil_imgs = None
            self.target_joints = torch.zeros(n_img, len(cfg.CANONICAL_MODEL_JOINTS), 2)
            self.target_visibility = torch.zeros(n_img, len(cfg.CANONICAL_MODEL_JOINTS)).long()
        else:
            self.rgb_imgs, self.sil_imgs, self.target_joints, self.target_visibility = data_batch
            self.target_visibility = self.target_visibility.long()
        n_img = self.sil_imgs.shape[0] if self.sil_imgs is not None else self.rgb_imgs.shape[0]
        if n_img % self.views:
            raise ValueError(f"{n_img} target images is not a multiple of views={self.views}")
        self.num_images = n_img // self.views  # frames held by this rank
        self.image_size = int(self.sil_imgs.shape[-1] if self.sil_imgs is not None else self.rgb_imgs.shape[2])
        self.frame0 = int(frame0)
        self.n_frames_total = int(n_frames_total) if n_frames_total is not None else self.num_images
        self.use_unity_prior = False
        self.batch_size = batch_size
        self.n_betas = nB
        self.shape_family_list = np.array(shape_family)
        self.propagate_scaling = False
        N = self.num_images

        # shape prior learned from the scanned models, identity fallback (reference fitter.py:121-136,170-175)
        mean = tables.shape_mean_betas if (tables.shape_cov is not None and tables.shape_mean_betas is not None) else None
        self.mean_betas = (torch.zeros(nB) if mean is None else torch.from_numpy(np.asarray(mean, np.float32))[:nB]).to(dev).contiguous()
        self.betas_prec = torch.from_numpy(shape_prior_precision(tables.shape_cov if mean is not None else None, nB)).to(dev).contiguous()
        self.pose_prior = _Prior(J, dev)
        self.max_limits = torch.full((J - 1, 3), cfg.JOINT_LIMIT, device=dev)
        self.min_limits = -self.max_limits

        # parameters (names as in the reference: optimize_to_joints.py:118-144 addresses them by name)
        self.betas = nn.Parameter(self.mean_betas.clone())
        self.log_beta_scales = nn.Parameter(torch.zeros(N, J, 3, device=dev), requires_grad=False)
        self.betas_trans = nn.Parameter(torch.zeros(N, J, 3, device=dev), requires_grad=False)
        # global_rotation and joint_rotations are views of ONE (N,J,3) pose buffer so the kernels read them without
        # a concatenation; they stay ordinary leaf Parameters for torch.optim
        self._pose = torch.zeros(N, J, 3, device=dev)
        self._pose[:, 0] = torch.from_numpy(default_global_rotation()).to(dev)
        self.global_rotation = nn.Parameter(self._pose[:, 0])
        self.joint_rotations = nn.Parameter(self._pose[:, 1:])
        self.trans = nn.Parameter(torch.zeros(N, 3, device=dev))
        self.global_mask = torch.ones(1, 3, device=dev)
        self.rotation_mask = torch.ones(J - 1, 3, device=dev)

        self.smal_model = SMAL(dev, tables=tables, config=cfg)
        self.renderer = Renderer(self.image_size, dev, views=self.views)
        self.renderer.bind_model(self.smal_model.device_model)
        self.fov = nn.Parameter(self.renderer.cameras.fov.clone())  # (1,) = 60 deg

        # device-resident targets (the reference re-uploads them every forward, fitter.py:263-266)
        self._graph = None
        self._epoch = None
        self._targets_dirty = True
        self._adam: Dict[str, Dict] = {}
        self._adam_step = 0

    # ------------------------------------------------------------------------------------------
    # plumbing
    # ------------------------------------------------------------------------------------------
    @property
    def device_model(self) -> engine.DeviceModel:
        return self.smal_model.device_model

    def set_cameras(self, R, T, fov=None, aspect_ratio=None):
        """Install per-view or per-image cameras; ``fov`` (if given) replaces the trainable parameter."""
        f = self.fov.data if fov is None else fov
        self.renderer.set_camera_parameters(R, T, f, aspect_ratio)
        if fov is not None:
            self.fov = nn.Parameter(self.renderer.cameras.fov.clone())
        self._graph = None  # a captured iteration holds the old camera tables' addresses

    def _refresh_targets(self):
        dev = self.device
        n_img = self.num_images * self.views
        if self.sil_imgs is not None:
            sil = self.sil_imgs.to(dev).reshape(n_img, self.image_size, self.image_size)
            if sil.dtype != torch.uint8:
                sil = sil.float()
                # binary masks (the usual case) are kept as bytes: a quarter of the memory and of the read traffic
                if bool(((sil == 0) | (sil == 1)).all()):
                    sil = sil.to(torch.uint8)
            self._sil_dev = sil.contiguous()
            self._sil_sum = engine.image_abs_sum(self._sil_dev)
        else:
            self._sil_dev = self._sil_sum = None
        self._tj_dev = self.target_joints.to(dev).float().contiguous()
        self._vis_dev = self.target_visibility.to(dev).to(torch.int32).contiguous()
        canon = list(self.config.CANONICAL_MODEL_JOINTS)
        self._canon_identity = canon == list(range(self.smal_model.tables.J))
        self._canon_dev = torch.tensor(canon, dtype=torch.int32, device=dev)
        self._targets_dirty = False
        self._target_signature = self._signature()

    def _signature(self):
        """(identity, in-place version) of every target tensor: the reference driver edits ``target_visibility`` in
        place (optimize_to_joints.py:135-138), which no attribute hook can see; torch's version counter can."""
        sig = []
        for t_ in (self.target_visibility, self.target_joints, self.sil_imgs):
            sig.append(None if t_ is None else (id(t_), t_._version))
        sig.append(tuple(self.config.CANONICAL_MODEL_JOINTS))
        return tuple(sig)

    def invalidate_targets(self):
        """Force a re-upload of the targets (in-place edits and re-assignments are detected automatically)."""
        self._targets_dirty = True

    def __setattr__(self, name, value):
        if name in ("target_visibility", "target_joints", "sil_imgs") and "_targets_dirty" in self.__dict__:
            self.__dict__["_targets_dirty"] = True
        if name in ("global_mask", "rotation_mask", "renderer", "propagate_scaling") and "_graph" in self.__dict__:
            self.__dict__["_graph"] = None  # re-assigned tables: the captured iteration read the old buffers
        super().__setattr__(name, value)

    def _mask_table(self) -> torch.Tensor:
        """(J,3) = [global_mask ; rotation_mask] in ONE persistent device buffer, refreshed in place when a mask tensor was
        replaced or edited in place (the reference documents ``fitter.rotation_mask[25:32] = 0.0``): a captured iteration
        reads this buffer, so it sees the current masks at every replay."""
        key = (self.global_mask.data_ptr(), self.global_mask._version, self.rotation_mask.data_ptr(), self.rotation_mask._version)
        cached = self.__dict__.get("_mask_cache")
        if cached is None or cached[0] != key:
            table = torch.cat([self.global_mask.reshape(1, 3), self.rotation_mask.reshape(-1, 3)], 0).float().contiguous()
            if cached is not None and cached[1].shape == table.shape and cached[1].device == table.device:
                cached[1].copy_(table)
                table = cached[1]
            cached = (key, table)
            self.__dict__["_mask_cache"] = cached
        return cached[1]

    def _pix_scale(self, fc, views: int, S: int) -> torch.Tensor:
        """Per-image weight of the silhouette term (w_reproj / (window size * views * S^2)): depends on the loss weights and
        the window layout only, so it is computed once per configuration, not once per iteration."""
        key = (fc.N, views, S, fc.w_reproj, fc.window, fc.frame0, fc.N_total)
        cached = self.__dict__.get("_pix_scale_cache")
        if cached is None or cached[0] != key:
            cached = (key, engine.pix_scale(fc, views, S, self.device))
            self.__dict__["_pix_scale_cache"] = cached
        return cached[1]

    def _rows(self, p: torch.Tensor, idx: Optional[torch.Tensor], n_sel: int):
        """(tensor, shared?) for a parameter that is either one shared row or one row per frame."""
        if p.shape[0] == 1:
            return p.detach()[0].contiguous(), True
        if p.shape[0] != self.num_images:
            raise ValueError(f"parameter with {p.shape[0]} rows for {self.num_images} frames")
        return (p.detach() if idx is None else p.detach().index_select(0, idx)).contiguous(), False

    # ------------------------------------------------------------------------------------------
    # the fused loss + gradient evaluation
    # ------------------------------------------------------------------------------------------
    def _loss_and_grads(self, frames: Optional[Sequence[int]], weights, w_temp: float, window: Optional[int] = None,
                        halo_prev=None, halo_next=None, halo=None, window_terms: bool = False):
        """Evaluate every loss term and the gradient of their sum for ``frames`` (None = all frames of this rank).

        Returns ``(objs (10,), grads)`` with full-size gradient tensors (zero rows outside ``frames``).
        ``window``: frames per loss window; None = the selected frames form one window (``forward`` semantics).
        ``halo``: an ``optimize.PendingHalo`` instead of ``halo_prev`` / ``halo_next`` - waited for right before the epilogue kernel,
        the only reader of the rows, so the messages travel while skinning and rasteriser run.
        ``window_terms``: also return ``grads["_objs_win"]`` (windows, 6), the six terms of every window on its own (one more
        kernel over the buffers this evaluation leaves behind: ``smil_window_terms``).
        """
        if self._targets_dirty or self._signature() != self._target_signature:
            self._refresh_targets()
        dev, dm, cfg = self.device, self.device_model, self.config
        J, nB, V, views, S = dm.J, dm.nB, dm.V, self.views, self.image_size
        N_all = self.num_images
        w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay = [float(w) for w in weights]
        if self.rgb_only:
            w_reproj = 0.0
        if frames is None:
            idx, n = None, N_all
            frame0, n_total = self.frame0, self.n_frames_total
            win = window if window is not None else n_total
        else:
            fl = list(frames)
            n = len(fl)
            idx = torch.tensor(fl, dtype=torch.long, device=dev)
            frame0, n_total, win = 0, n, (window if window is not None else n)
            w_temp = 0.0  # a window has no temporal term; get_temporal covers the sequence
        sel = (lambda t: t) if idx is None else (lambda t: t.index_select(0, idx))
        pose = sel(self._pose.detach()).contiguous()
        trans = sel(self.trans.detach()).contiguous()
        mask = self._mask_table()
        ls, ls_shared = self._rows(self.log_beta_scales, idx, n)
        bt, bt_shared = self._rows(self.betas_trans, idx, n)
        betas = self.betas.detach().contiguous()
        fc = engine.fit_config(n, J, nB, win, [w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay], w_temp, frame0, n_total,
                               cfg.JOINT_LIMIT, self.global_rotation.requires_grad, self.joint_rotations.requires_grad,
                               self.trans.requires_grad)
        # everything the kernels ADD into lives in one buffer with one zero fill: loss terms, the shared shape gradient,
        # the per-image fov sums
        # ... laid out so that everything ranks have to SUM sits in one contiguous "shared block" at the front: the loss terms,
        # the shape gradient, the fov gradient when fov is shared, the scale-table gradients when the tables are shared
        n_img = n * views
        fov_n = self.fov.numel()
        n_fov = fov_n if fov_n in (1, views) else 0  # (a per-image fov belongs to its rank: not in the block)
        n_ls = 3 * J if (ls_shared and self.log_beta_scales.requires_grad) else 0
        n_bt = 3 * J if (bt_shared and self.betas_trans.requires_grad) else 0
        o_fov, o_ls, o_bt = N_OBJS + nB, N_OBJS + nB + n_fov, N_OBJS + nB + n_fov + n_ls
        n_shared = o_bt + n_bt
        arena = torch.zeros(n_shared + n_img, dtype=torch.float32, device=dev)
        objs, d_betas, d_fov_img = arena[:N_OBJS], arena[N_OBJS:N_OBJS + nB], arena[n_shared:]
        self.__dict__["_shared_block"] = arena[:n_shared]

        # cameras: one table row per view, per image or shared; fov may be the trainable parameter
        cam = self.renderer.cameras
        fov = self.fov.detach().reshape(-1).contiguous()
        img_idx = None
        if idx is not None:
            img_idx = (idx[:, None] * views + torch.arange(views, device=dev)[None]).reshape(-1)

        def cam_rows(t, rows):
            k = t.shape[0]
            if k in (1, views) or idx is None:
                return t.contiguous()
            if k != N_all * views:
                raise ValueError(f"camera table with {k} rows for {N_all * views} images")
            return t.index_select(0, img_idx).contiguous()

        cams = engine.CameraSet(cam_rows(cam.R, 9), cam_rows(cam.T, 3), cam_rows(fov, 1),
                                None if cam.aspect_ratio is None else cam_rows(cam.aspect_ratio.reshape(-1), 1), views, S)

        need_render = (w_j2d > 0) or (w_reproj > 0)
        g_lbs = None
        d_fov = loss_img = pscale = d_fov_sel = None
        yx = tj = vis = None
        if need_render:
            # the rotation masks are applied inside the pose kernels (theta_mask): no masked copy of the pose
            lbs = engine.lbs_forward(dm, betas, pose, trans=trans, logscale=ls, btrans=bt, shared_beta=True,
                                     logscale_shared=ls_shared, btrans_shared=bt_shared, propagate_scaling=self.propagate_scaling,
                                     allow_limb_scaling=cfg.ALLOW_LIMB_SCALING, trans_after_joints=True, theta_mask=mask,
                                     project=dict(cams=cams, ndc=w_reproj > 0, yx=w_j2d > 0) if engine.FUSED_LBS_FORWARD else None)
            both = w_j2d > 0 and w_reproj > 0
            ndc = d_yx = d_ndc = d_verts = d_joints = cd = None
            if engine.FUSED_LBS_FORWARD:  # projected by the skinning kernel (vertices -> NDC, joints -> pixels)
                ndc, yx = lbs.get("ndc"), lbs.get("yx")
            elif both:  # vertices -> NDC and joints -> pixels in one launch
                ndc, yx = engine.project_verts_and_joints(cams, lbs["verts"], lbs["joints"])
            elif w_j2d > 0:
                _, yx = engine.project(cams, lbs["joints"], want_ndc=False)
            else:
                ndc, _ = engine.project(cams, lbs["verts"], want_yx=False)
            if w_j2d > 0:
                tj = (self._tj_dev if idx is None else self._tj_dev.index_select(0, img_idx)).contiguous()
                vis = (self._vis_dev if idx is None else self._vis_dev.index_select(0, img_idx)).contiguous()
                d_yx = torch.empty_like(yx)
                Jc = self._canon_dev.numel()
                engine.joint_loss(fc, views, Jc, None if self._canon_identity else self._canon_dev, yx, tj, vis, objs, d_yx)
            if w_reproj > 0:
                tgt = self._sil_dev if idx is None else self._sil_dev.index_select(0, img_idx).contiguous()
                tsum = self._sil_sum if idx is None else self._sil_sum.index_select(0, img_idx).contiguous()
                pscale = self._pix_scale(fc, views, S)
                # (the vertex gradient stays as the tile kernel accumulated it: the projection backward decodes it while it reads)
                # (the depth gradients of edges cut at the clipping plane travel beside d_ndc; persistent buffers: a captured
                # iteration replays the same pointers)
                cd = self.__dict__["_last_clip_depth"] = engine.clip_depth_for(dm, n_img)
                loss_img, d_ndc, _, d_ndc_scale = engine.silhouette_l1_fused(dm, ndc, S, tgt, tsum, pscale, self.renderer.raster_settings,
                                                                             packed_out=True, clip_depth=cd)
            # image-plane gradients -> world space: inside the skinning backward (one kernel per frame, no (B,V,3) vertex
            # gradient in memory) where the library offers it, else by the projection backward first
            ndc_up = None
            if engine.FUSED_LBS_BACKWARD and engine.lbs_backward_ndc_supported(dm, nB if self.betas.requires_grad else 0, views):
                ndc_up = dict(cams=cams, d_ndc=d_ndc, d_ndc_scale=d_ndc_scale if d_ndc is not None else None, d_yx=d_yx, d_fov_img=d_fov_img,
                              clip_depth=cd if d_ndc is not None else None)
            elif both:
                d_verts, d_joints = engine.project_backward_verts_and_joints(cams, lbs["verts"], d_ndc, lbs["joints"], d_yx, d_fov_img,
                                                                             d_ndc_scale=d_ndc_scale)
            elif w_j2d > 0:
                d_joints, _ = engine.project_backward(cams, lbs["joints"], d_yx=d_yx, d_fov_img=d_fov_img)
            else:
                d_verts, _ = engine.project_backward(cams, lbs["verts"], d_ndc=d_ndc, d_fov_img=d_fov_img, d_ndc_scale=d_ndc_scale)
            if d_verts is not None and d_ndc is not None and cd is not None:
                engine.clip_depth_backward(cams, cd, d_verts)
            d_fov_sel = arena[o_fov:o_ls] if (n_fov and cams.fov.numel() == n_fov) else torch.empty(cams.fov.numel(), dtype=torch.float32, device=dev)
            # the shared shape gradient is accumulated straight into d_betas (where the shape prior adds its own); the shared
            # scale tables' gradients land in their slots of the shared block
            g_lbs = engine.lbs_backward(dm, lbs, d_verts, d_joints, need_beta=self.betas.requires_grad,
                                        need_logscale=self.log_beta_scales.requires_grad,
                                        need_btrans=self.betas_trans.requires_grad, need_trans=self.trans.requires_grad,
                                        d_beta_accum=d_betas, out_logscale=arena[o_ls:o_bt].view(J, 3) if n_ls else None,
                                        out_btrans=arena[o_bt:n_shared].view(J, 3) if n_bt else None, ndc_upstream=ndc_up)
        if g_lbs is not None and g_lbs["d_theta"] is not None:
            d_pose = g_lbs["d_theta"]
            d_trans = g_lbs["d_trans"] if g_lbs["d_trans"] is not None else torch.zeros(n, 3, dtype=torch.float32, device=dev)
            accumulate = True
        else:
            d_pose = torch.empty(n, J, 3, dtype=torch.float32, device=dev)
            d_trans = torch.zeros(n, 3, dtype=torch.float32, device=dev)
            accumulate = False
        if halo is not None:
            halo_prev, halo_next = halo.wait()
        # priors + temporal terms + silhouette objective + fov reduction: one launch
        engine.fit_epilogue(fc, pose, trans, betas, self.mean_betas, self.betas_prec, mask, objs, d_pose, d_trans, d_betas,
                            halo_prev=halo_prev, halo_next=halo_next, accumulate=accumulate, loss_img=loss_img, pix_scale=pscale,
                            cams=cams if d_fov_sel is not None else None, d_fov_img=d_fov_img if d_fov_sel is not None else None,
                            d_fov=d_fov_sel)
        objs_win = None
        if window_terms:  # every window's own six terms (the drop-in forward() serves the windows of an epoch from one evaluation)
            objs_win = engine.window_terms(fc, views, self._canon_dev.numel(), None if self._canon_identity else self._canon_dev,
                                           yx if w_j2d > 0 else None, tj, vis, pose, mask, objs, loss_img, pscale)
        if d_fov_sel is not None and (fov.numel() in (1, views) or idx is None):
            d_fov = d_fov_sel
        else:
            d_fov = arena[o_fov:o_ls] if n_fov else torch.zeros_like(fov)
            if d_fov_sel is not None:  # per-image fov, window of frames: scatter the selected images' gradients
                d_fov.index_add_(0, img_idx, d_fov_sel)

        def scatter(rows, like):
            if idx is None:
                return rows
            full = torch.zeros_like(like)
            full.index_copy_(0, idx, rows)
            return full

        def table_grad(g, shared, like, slot):
            if g is None:  # (a shared table keeps its - zero - slot of the shared block)
                return slot.view(like.shape) if slot.numel() else torch.zeros_like(like)
            return g.reshape(like.shape) if shared else scatter(g, like)

        grads = dict(
            betas=d_betas if self.betas.requires_grad else None,
            pose=scatter(d_pose, self._pose),
            trans=scatter(d_trans, self.trans),
            log_beta_scales=table_grad(g_lbs["d_logscale"] if g_lbs else None, ls_shared, self.log_beta_scales, arena[o_ls:o_bt])
            if self.log_beta_scales.requires_grad else None,
            betas_trans=table_grad(g_lbs["d_btrans"] if g_lbs else None, bt_shared, self.betas_trans, arena[o_bt:n_shared])
            if self.betas_trans.requires_grad else None,
            fov=d_fov.reshape(self.fov.shape) if self.fov.requires_grad else None,
        )
        if objs_win is not None:
            grads["_objs_win"] = objs_win
        return objs, grads

    # ------------------------------------------------------------------------------------------
    # reference API
    # ------------------------------------------------------------------------------------------
    def print_grads(self, grad_output):
        """Debug hook of the reference (fitter.py:233-234): prints a gradient it is registered on."""
        print(grad_output)

    def forward(self, batch_range, weights, stage_id):
        """Reference fitter.py:236-335: ``(sum of the weighted terms, dict of the terms)`` for one window.

        The reference's driver calls this once per ``WINDOW_SIZE`` frames and adds the losses up before ONE backward
        (optimize_to_joints.py:153-175) - 410 calls per epoch at 4096 frames.  The windows of an epoch are independent given the
        parameters, so from the SECOND window requested under unchanged parameters on, all windows are evaluated in one launch
        chain (``_EpochEval``) and this call - and every later one of the epoch - is served from it (``_epoch_window``)."""
        wts = tuple(float(w) for w in weights)
        j = self._epoch_window(batch_range, wts)
        if j is not None:
            total, objs = self._epoch["losses"][j], self._epoch["objs_win"][j]  # (a tuple of scalars: no autograd node per window)
        else:
            total, objs = _FitWindow.apply(self, list(batch_range), list(wts), 0.0, self.betas, self.log_beta_scales,
                                           self.betas_trans, self._pose_leaf(), self.trans, self.fov)
        terms = objs.unbind(0)
        out = {}
        for k, name in enumerate(OBJ_NAMES):
            if wts[_WEIGHT_OF[name]] > 0 and not (name == "sil_reproj" and self.rgb_only):
                out[name] = terms[k]
        return total, out

    # ---- one evaluation per epoch behind the per-window forward() ------------------------------------------------
    def _state_key(self, wts):
        """Everything a cached epoch depends on: the parameters (identity + in-place version counter: ``optimizer.step()`` and
        ``param[...] = x`` bump it), which of them train, loss weights, targets, masks, cameras and rasteriser settings.  Edits that
        bypass the counter (``param.data[...] = x``) need ``invalidate_epoch()``.  (Called once per ``forward``: plain attribute reads.)"""
        P = self._parameters
        b, ls, bt, gr, jr, tr, fv = P["betas"], P["log_beta_scales"], P["betas_trans"], P["global_rotation"], P["joint_rotations"], P["trans"], P["fov"]
        cam = self.renderer.cameras
        R, T, asp, canon = cam.R, cam.T, cam.aspect_ratio, self.config.CANONICAL_MODEL_JOINTS
        tv, tj, si, gm, rm = self.target_visibility, self.target_joints, self.sil_imgs, self.global_mask, self.rotation_mask
        return (wts, id(b), b._version, b.requires_grad, id(ls), ls._version, ls.requires_grad, id(bt), bt._version, bt.requires_grad,
                id(gr), gr._version, gr.requires_grad, id(jr), jr._version, jr.requires_grad, id(tr), tr._version, tr.requires_grad,
                id(fv), fv._version, fv.requires_grad, id(tv), tv._version, id(tj), tj._version, id(si), None if si is None else si._version,
                id(canon), len(canon), gm.data_ptr(), gm._version, rm.data_ptr(), rm._version, R.data_ptr(), R._version, T.data_ptr(), T._version,
                None if asp is None else (asp.data_ptr(), asp._version), self.propagate_scaling, self.rgb_only,
                bytes(self.renderer.raster_settings), torch.is_grad_enabled())  # (an evaluation under no_grad carries no graph)

    def invalidate_epoch(self):
        """Forget the cached epoch (after editing a parameter through ``.data`` or any other route autograd's version counters miss)."""
        self.__dict__["_epoch"] = None

    def _epoch_window(self, batch_range, wts):
        """Index of ``batch_range`` among the windows of the cached epoch evaluation, or None when this call has to be evaluated on
        its own.  Policy: the first window requested under a new parameter state is evaluated directly (a caller that only ever asks
        for one window per state - stochastic mini-batches - never pays for a whole batch); the second one switches the epoch to the
        whole-batch evaluation, and once an epoch has been served that way the next one starts with it at its first window."""
        W = int(self.batch_size) if self.batch_size else 0
        N = self.num_images
        n = len(batch_range)
        if not self.epoch_cache or W <= 0 or n == 0 or N <= W:
            return None
        j0 = int(batch_range[0])
        if j0 % W or n != min(W, N - j0) or self.frame0 % W:
            return None
        wl = self.__dict__.get("_win_lists")
        if wl is None or wl[0] != (N, W):
            wl = self.__dict__["_win_lists"] = ((N, W), [list(range(j, min(N, j + W))) for j in range(0, N, W)])
        if (batch_range if type(batch_range) is list else [int(b) for b in batch_range]) != wl[1][j0 // W]:
            return None
        key = self._state_key(wts)
        ep = self.__dict__.get("_epoch")
        if ep is not None and ep["key"] == key:
            if ep["losses"] is None:  # second window of this state: evaluate them all now
                self._evaluate_epoch(ep, wts, W)
            ep["served"] += 1
            return j0 // W
        eager = ep is not None and ep["losses"] is not None  # the last state saw a second window: it was served from one evaluation
        ep = self.__dict__["_epoch"] = dict(key=key, losses=None, objs_win=None, served=0)
        if not eager:
            return None
        self._evaluate_epoch(ep, wts, W)
        ep["served"] += 1
        return j0 // W

    def _evaluate_epoch(self, ep, wts, W):
        out = _EpochEval.apply(self, list(wts), W, self.betas, self.log_beta_scales, self.betas_trans, self._pose_leaf(), self.trans, self.fov)
        ep["losses"], ep["objs_win"] = out[:-1], out[-1]

    def _pose_leaf(self):
        """Autograd handle tying the fused pose gradient to the two rotation Parameters."""
        return torch.cat([self.global_rotation[:, None], self.joint_rotations], dim=1)

    def get_temporal(self, w_temp):
        """Reference fitter.py:337-350: (joint_loss, global_loss, trans_loss) over consecutive frames - three scalars, each
        carrying its own graph like the reference's: the joint term depends on ``joint_rotations`` only, the global term on
        ``global_rotation`` only, the translation term on ``trans`` only, so the one evaluated gradient splits exactly."""
        joint, glob, tr = _TemporalTerm.apply(self, float(w_temp), self._pose_leaf(), self.trans)
        return joint, glob, tr

    def load_checkpoint(self, checkpoint_path, epoch):
        """Reference fitter.py:352-371: per-frame ``<frame>/<epoch>.pkl`` parameter dicts; betas/scales averaged."""
        beta_list, scale_list = [], []
        with torch.no_grad():
            for frame_id in range(self.num_images):
                with open(os.path.join(checkpoint_path, "{0:04}".format(frame_id), "{0}.pkl".format(epoch)), "rb") as f:
                    p = pkl.load(f)
                self.global_rotation[frame_id] = torch.from_numpy(np.asarray(p["global_rotation"])).float().to(self.device).reshape(3)
                self.joint_rotations[frame_id] = torch.from_numpy(np.asarray(p["joint_rotations"])).float().to(self.device).view(-1, 3)
                self.trans[frame_id] = torch.from_numpy(np.asarray(p["trans"])).float().to(self.device).reshape(3)
                beta_list.append(np.asarray(p["betas"]).reshape(-1)[: self.n_betas])
                scale_list.append(np.asarray(p["log_betascale"]))
        self.betas = nn.Parameter(torch.from_numpy(np.mean(beta_list, axis=0)).float().to(self.device))
        scales = torch.from_numpy(np.mean(scale_list, axis=0)).float().to(self.device).reshape(1, -1, 3)
        self.log_beta_scales = nn.Parameter(scales, requires_grad=self.log_beta_scales.requires_grad)

    def generate_visualization(self, image_exporter, apply_UE_transform=False, img_idx=0, mesh_scale=None, epoch=None):
        """Reference fitter.py:373-517, as far as this build goes: for every frame ``image_exporter.export(collage, batch_id,
        global_id, img_parameters, verts, faces, img_idx, epoch=epoch)`` with the SAME per-frame parameter dict (what the reference
        pickles as ``st{S}_ep{E}.pkl`` and ``load_checkpoint`` reads back), the same posed vertices (the ``.ply``) and a collage of the
        same layout (target | render | overlay | silhouette agreement | view from behind).  The reference's driver calls this
        every ``VIS_FREQUENCY`` epochs (optimize_to_joints.py:177-178), so the unchanged loop needs it to work.  What differs: the
        colour / HardPhong shading (p3d_renderer.py render_texture=True) and the joint markers (SMALJointDrawer, cv2) are
        visualisation code outside this build - the "render" panels show the soft silhouette in the mesh colour, without markers.
        Frames are posed and rendered by the HIP kernels; the collage is assembled on the host."""
        cfg, dev, views, S = self.config, self.device, self.views, self.image_size
        J = self.smal_model.tables.J
        faces_np = self.smal_model.faces.detach().cpu().numpy()
        color = torch.tensor([c / 255.0 for c in getattr(cfg, "MESH_COLOR", [0, 172, 223])], dtype=torch.float32).view(1, 3, 1, 1)
        rot = torch.tensor([[-1.0, 0.0, 0.0], [0.0, 1.0, 0.0], [0.0, 0.0, -1.0]], device=dev)  # 180 degrees about y (fitter.py:388)
        cam_all = self.renderer.cameras
        W = int(self.batch_size) if self.batch_size else self.num_images
        try:
            with torch.no_grad():
                for j in range(0, self.num_images, W):
                    rows = list(range(j, min(self.num_images, j + W)))
                    idx = torch.tensor(rows, device=dev)
                    n = len(rows)
                    pick = lambda p_: (p_.detach() if p_.shape[0] == 1 else p_.detach().index_select(0, idx))  # noqa: E731
                    theta = torch.cat([(self.global_rotation.detach().index_select(0, idx) * self.global_mask)[:, None],
                                       self.joint_rotations.detach().index_select(0, idx) * self.rotation_mask], 1)
                    trans = self.trans.detach().index_select(0, idx)
                    verts, joints, _, _ = self.smal_model(self.betas.detach()[None].expand(n, -1), theta, betas_logscale=pick(self.log_beta_scales),
                                                          betas_trans=pick(self.betas_trans), propagate_scaling=self.propagate_scaling)
                    if apply_UE_transform:  # (the replicAnt convention: ten times larger about the root joint)
                        root = joints[:, :1]
                        verts, joints = (verts - root) * 10 + trans[:, None], (joints - root) * 10 + trans[:, None]
                    elif mesh_scale is not None:
                        sc = torch.as_tensor(mesh_scale, dtype=torch.float32, device=dev).reshape(-1, 1, 1)
                        root = joints[:, :1]
                        verts, joints = (verts - root) * sc + trans[:, None], (joints - root) * sc + trans[:, None]
                    else:
                        verts, joints = verts + trans[:, None], joints + trans[:, None]
                    canon = joints[:, list(cfg.CANONICAL_MODEL_JOINTS)].contiguous()
                    img_rows = (idx[:, None] * views + torch.arange(views, device=dev)[None]).reshape(-1)

                    def table(t_, per_row):  # camera tables with one row per image follow the window; shared / per-view ones stay
                        return t_ if t_ is None or t_.shape[0] != self.num_images * views else t_.index_select(0, img_rows)
                    fov = self.fov.detach().reshape(-1)
                    self.renderer.cameras = FoVCameras(table(cam_all.R, 9), table(cam_all.T, 3), table(fov, 1),
                                                       table(cam_all.aspect_ratio, 1), cam_all.znear, cam_all.zfar)
                    faces_b = self.smal_model.faces[None].expand(n, -1, -1)
                    sil, _ = self.renderer(verts.contiguous(), canon, faces_b)
                    centre = verts.mean(1, keepdim=True)
                    sil_rev, _ = self.renderer(((verts - centre) @ rot.T).contiguous(), ((canon - centre) @ rot.T).contiguous(), faces_b)
                    first = torch.arange(n, device=dev) * views  # a frame's first view stands for it
                    sil = sil.reshape(n * views, 1, S, S).index_select(0, first).cpu()
                    sil_rev = sil_rev.reshape(n * views, 1, S, S).index_select(0, first).cpu()
                    take = (idx * views).cpu()
                    rgb = self.rgb_imgs[take].float().cpu()
                    target_sil = torch.zeros_like(sil) if (self.rgb_only or self.sil_imgs is None) else self.sil_imgs[take].float().cpu().reshape(n, 1, S, S)
                    rendered = sil * color
                    agreement = (1.0 - (target_sil - sil).abs()).expand_as(rgb)
                    collage = torch.cat([rgb, rendered, 0.5 * rendered + 0.5 * rgb, agreement, sil_rev * color], dim=3).clamp(0.0, 1.0)
                    for batch_id, global_id in enumerate(rows):
                        image_exporter.export((collage[batch_id].permute(1, 2, 0).numpy() * 255.0).astype(np.uint8), batch_id, global_id,
                                              self.export_parameters(global_id), verts, faces_np, img_idx, epoch=epoch)
        finally:
            self.renderer.cameras = cam_all

    def export_parameters(self, frame_id: int) -> Dict[str, np.ndarray]:
        """The per-frame dict the reference pickles (optimize_to_joints.py:48-63, fitter.py:241-261,507)."""
        ls = self.log_beta_scales.detach()
        bt = self.betas_trans.detach()
        return dict(
            global_rotation=(self.global_rotation.detach()[frame_id] * self.global_mask[0]).cpu().numpy(),
            joint_rotations=(self.joint_rotations.detach()[frame_id] * self.rotation_mask).cpu().numpy(),
            betas=self.betas.detach().cpu().numpy(), trans=self.trans.detach()[frame_id].cpu().numpy(),
            fov=self.fov.detach().reshape(-1)[min(frame_id, self.fov.numel() - 1)].cpu().numpy(),
            log_betascale=ls[min(frame_id, ls.shape[0] - 1)].cpu().numpy(), betas_trans=bt[min(frame_id, bt.shape[0] - 1)].cpu().numpy())

    # ------------------------------------------------------------------------------------------
    # fused epoch (fast path)
    # ------------------------------------------------------------------------------------------
    def begin_stage(self, lr: float, fov_lr: float = 1.0, betas=(0.5, 0.999), eps: float = 1e-8):
        """New Adam state per stage, like the reference's new optimiser per stage (optimize_to_joints.py:117-127)."""
        self._adam = {}
        self._adam_step = 0
        self._adam_hyper = dict(lr=float(lr), fov_lr=float(fov_lr), betas=betas, eps=eps)
        self._graph = None  # a captured iteration belongs to one stage

    def _param_tensor(self, name: str) -> torch.Tensor:
        return self._pose if name == "pose" else getattr(self, name).data

    def apply_adam(self, grads: Dict[str, Optional[torch.Tensor]], advance: bool = True):
        """torch.optim.Adam(betas=(0.5,0.999)) semantics on every parameter that received a gradient.  ``advance=False``:
        a second group of the same optimiser step (the shared parameters, once their all-reduced gradient has arrived)."""
        h = self._adam_hyper
        if advance:
            self._adam_step += 1
        items = []
        for name, g in grads.items():
            if g is None:
                continue
            p = self._param_tensor(name)
            st = self._adam.get(name)
            if st is None:
                st = self._adam[name] = dict(m=torch.zeros_like(p), v=torch.zeros_like(p), t0=self._adam_step - 1)
            lr = h["fov_lr"] if name == "fov" else h["lr"]
            items.append((p, g.contiguous(), st["m"], st["v"], lr, self._adam_step - st["t0"]))
        if items:  # one launch for all of them
            engine.adam_step_multi(items, h["betas"][0], h["betas"][1], h["eps"])
            self.__dict__["_epoch"] = None  # (written through .data: the parameters' version counters do not move)

    def fit_step(self, weights, w_temp: float, window: Optional[int] = None, halo_prev=None, halo_next=None,
                 shared_grad_hook=None, halo=None):
        """One epoch over all frames of this rank: losses + gradients + Adam.  Returns objs (10,) (device).

        Several ranks: ``shared_grad_hook(block)`` receives the shared block - one contiguous tensor ``[10 loss terms | d_betas
        | d_fov | shared scale-table gradients]`` - right after backward, sums it over the ranks IN PLACE and returns a handle
        (``optimize.allreduce_block``).  The per-frame parameters take their Adam step while that collective is in flight;
        the shared ones after ``handle.wait()``."""
        window = self.config.WINDOW_SIZE if window is None else window
        objs, grads = self._loss_and_grads(None, weights, w_temp, window=window, halo_prev=halo_prev, halo_next=halo_next, halo=halo)
        if shared_grad_hook is None:
            self.apply_adam(grads)
            return objs
        handle = shared_grad_hook(self._shared_block)
        in_block = lambda g: g is not None and g.untyped_storage().data_ptr() == self._shared_block.untyped_storage().data_ptr()  # noqa: E731
        local = {k: g for k, g in grads.items() if g is not None and not self._is_shared(k)}
        shared = {k: g for k, g in grads.items() if g is not None and self._is_shared(k)}
        stray = [k for k, g in shared.items() if not in_block(g)]
        if stray:  # a shared gradient that does not live in the block (never the case for the layouts _loss_and_grads builds)
            raise RuntimeError(f"shared gradients outside the shared block: {stray}")
        self.apply_adam(local)
        if handle is not None:
            handle.wait()
        self.apply_adam(shared, advance=False)
        return objs

    # ---- the same epoch as one hipGraph: ~40 kernel launches replayed with a single call --------------------
    def _graph_key(self, weights, w_temp, window):
        """Everything a captured iteration bakes in besides the parameter buffers: loss weights, which parameters train,
        the target tensors, the rasteriser settings, and the raw device addresses of the camera tables, the rotation masks and the
        rasteriser workspace.  A replay happens only while all of them are what they were at capture time."""
        flags = tuple(bool(getattr(self, n).requires_grad) for n in
                      ("betas", "log_beta_scales", "betas_trans", "global_rotation", "joint_rotations", "trans", "fov"))
        cam = self.renderer.cameras
        ptr = lambda t: None if t is None else (t.data_ptr(), tuple(t.shape))  # noqa: E731
        ws = self.device_model._ws
        addresses = (ptr(cam.R), ptr(cam.T), ptr(cam.aspect_ratio), ptr(self.fov.data), ptr(self._mask_table()),
                     ptr(self.log_beta_scales.data), ptr(self.betas_trans.data), ptr(self.betas.data), None if ws is None else ws.data_ptr())
        rs = self.renderer.raster_settings  # (passed by value into the captured launches: blur, sigma, K, clipping plane, tie rule)
        raster = (float(rs.blur_radius), float(rs.sigma), int(rs.faces_per_pixel), float(rs.z_clip), int(rs.tie_rule))
        return (tuple(float(w) for w in weights), float(w_temp), window, flags, self._target_signature, addresses, raster)

    def fit_step_graph(self, weights, w_temp: float, window: Optional[int] = None):
        """``fit_step`` for a single rank, captured once per (stage, weights) in a hipGraph (``torch.cuda.CUDAGraph``)
        and replayed afterwards.  Worth it when the iteration is launch-bound (few frames); results are identical.
        Returns objs (10,) in a buffer that the next replay overwrites."""
        window = self.config.WINDOW_SIZE if window is None else window
        if self._targets_dirty or self._signature() != self._target_signature:
            self._refresh_targets()
        self._mask_table()  # in-place mask edits since the capture reach the buffer the graph reads (outside the graph)
        key = self._graph_key(weights, w_temp, window)
        g = getattr(self, "_graph", None)
        if g is None or g["key"] != key:
            g = self._capture_step(weights, w_temp, window)
        if g["t_mirror"] != self._adam_step:  # eager steps in between: bring the device counter back in line
            self._adam_t.fill_(self._adam_step)
        self._adam_step += 1
        g["t_mirror"] = self._adam_step
        # (the replay's rasteriser kernels run on THIS stream, not on the one the graph was captured on: order them behind the last
        # user of the device's shared workspace, and make the next user wait for them)
        self.device_model._claim_workspace(g["last_launch"])
        g["graph"].replay()
        self.__dict__["_epoch"] = None
        return g["objs"]

    def _capture_step(self, weights, w_temp, window, ranks=None):
        """Capture the iteration: one graph (single rank), or - ``ranks=(rank, world)`` - the losses + backward and the Adam update as
        two graphs, the first reading the persistent halo buffers."""
        dev = self.device
        halo_kw = {}
        if ranks is not None:
            halo_kw = dict(halo_prev=self._halo_buf[0] if ranks[0] > 0 else None,
                           halo_next=self._halo_buf[1] if ranks[0] + 1 < ranks[1] else None)
        if not hasattr(self, "_adam_t"):
            self._adam_t = torch.zeros(1, dtype=torch.int32, device=dev)
        h = self._adam_hyper
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        with torch.cuda.stream(side):
            # eager dry run (no parameter update): sizes the rasteriser workspace and tells which parameters get a gradient
            _, grads = self._loss_and_grads(None, weights, w_temp, window=window, **halo_kw)
            for name, gr in grads.items():
                if gr is not None and name not in self._adam:
                    p = self._param_tensor(name)
                    self._adam[name] = dict(m=torch.zeros_like(p), v=torch.zeros_like(p), t0=self._adam_step)
        torch.cuda.current_stream(dev).wait_stream(side)
        self._adam_t.fill_(self._adam_step)
        torch.cuda.synchronize(dev)
        def adam_all(grads):
            for name, gr in grads.items():
                if gr is None:
                    continue
                st = self._adam[name]
                lr = h["fov_lr"] if name == "fov" else h["lr"]
                engine.adam_step_dev(self._param_tensor(name), gr.contiguous(), st["m"], st["v"], lr, self._adam_t, st["t0"],
                                     h["betas"][0], h["betas"][1], h["eps"])

        graph = torch.cuda.CUDAGraph()
        graph_adam = shared_block = None
        with torch.cuda.graph(graph):
            self._adam_t.add_(1)
            objs, grads = self._loss_and_grads(None, weights, w_temp, window=window, **halo_kw)
            if ranks is None:
                adam_all(grads)
            else:
                shared_block = self._shared_block  # summed over the ranks in place between the two graphs
        if ranks is not None:
            graph_adam = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph_adam, pool=graph.pool()):
                adam_all(grads)
        # keyed on the state AFTER the dry run, which may have (re)allocated the rasteriser workspace
        self._graph = dict(key=self._graph_key(weights, w_temp, window), graph=graph, objs=objs, t_mirror=self._adam_step,
                           graph_adam=graph_adam, shared_block=shared_block, grads=grads,
                           ws=self.device_model._ws,  # (the graph's kernels hold raw pointers into this workspace tensor)
                           last_launch=self.device_model.__dict__.get("_last_launch", 0))
        return self._graph

    def fit_step_graph_ranks(self, weights, w_temp: float, window: Optional[int], rank: int, world: int, group, shared_grad_hook,
                             host_staged: bool = False):
        """``fit_step`` of one rank among several as TWO hipGraphs with the collective between them:
        ``[losses + backward] | all-reduce of the shared block | [Adam of every parameter]``.  The temporal-halo rows are received
        straight into two persistent device buffers the first graph reads (posted before the replay, waited for in front of it:
        a graph cannot wait in its middle - the eager ``fit_step`` can, and does).  Identical results to the eager step."""
        from . import optimize  # (local: optimize imports nothing from here)

        window = self.config.WINDOW_SIZE if window is None else window
        if self._targets_dirty or self._signature() != self._target_signature:
            self._refresh_targets()
        self._mask_table()
        dev = self.device
        if not hasattr(self, "_halo_buf"):
            n_row = self._pose.shape[1] * 3 + 3
            self._halo_buf = (torch.zeros(n_row, device=dev), torch.zeros(n_row, device=dev))
        first, last = self.boundary_rows()
        pending = optimize.post_halos(first, last, rank, world, group, host_staged=host_staged,
                                      recv_prev=self._halo_buf[0], recv_next=self._halo_buf[1])
        prev_row, next_row = pending.wait()
        for buf, row in zip(self._halo_buf, (prev_row, next_row)):  # (host-staged rehearsals arrive in fresh tensors)
            if row is not None and row.data_ptr() != buf.data_ptr():
                buf.copy_(row)
        key = ("ranks", rank, world) + self._graph_key(weights, w_temp, window)
        g = getattr(self, "_graph", None)
        if g is None or g["key"] != key:
            g = self._capture_step(weights, w_temp, window, ranks=(rank, world))
            g["key"] = ("ranks", rank, world) + g["key"]
        if g["t_mirror"] != self._adam_step:
            self._adam_t.fill_(self._adam_step)
        self._adam_step += 1
        g["t_mirror"] = self._adam_step
        self.device_model._claim_workspace(g["last_launch"])  # (as in fit_step_graph: the replay is ordered on this stream)
        g["graph"].replay()
        handle = shared_grad_hook(g["shared_block"]) if shared_grad_hook is not None else None
        if handle is not None:
            handle.wait()
        g["graph_adam"].replay()
        self.__dict__["_epoch"] = None
        return g["objs"]

    def straddling_faces(self) -> int:
        """Faces of the most recent silhouette launch with one or two vertices nearer than ``z_clip = znear / 2``.  They are cut
        at the plane like pytorch3d's ``clip_faces`` does (left on by the reference's settings, p3d_renderer.py:36-47): the
        part in front is rendered.  A non-zero count still deserves a look - the mesh has reached the camera - so the first
        one warns; faces beyond the per-image clip tables (1024 cut faces per image) are rendered unclipped and always
        reported.  Synchronises the stream (call it between stages, not per iteration)."""
        if self.device_model._ws is None:
            return 0
        st = engine.raster_stats(self.device_model, self.num_images * self.views)
        n, lost = int(st["straddling_faces"]), int(st["unclipped_faces"])
        cd = self.__dict__.get("_last_clip_depth")  # (the buffer of the LAST evaluation only: other sizes' counters are stale)
        dropped = int(cd.counter.tolist()[1]) if cd is not None else 0  # (both counters in one copy)
        if dropped:  # (more cut edges in one call than ClipDepth.capacity entries: their depth gradients were left out, never silently)
            import warnings

            warnings.warn(f"{dropped} depth-gradient entries of cut edges did not fit engine.ClipDepth (capacity {cd.capacity}) in the "
                          "last evaluation and were dropped.", RuntimeWarning, stacklevel=2)
        if n and not self.__dict__.get("_warned_straddling"):
            import warnings

            self.__dict__["_warned_straddling"] = True
            warnings.warn(f"{n} mesh faces straddle the camera's clipping plane (z_clip = znear / 2) and were cut there, as the "
                          f"reference's rasteriser does ({lost} of them beyond the clip tables: rendered unclipped). The mesh has "
                          "probably drifted into the camera (check `trans`).", RuntimeWarning, stacklevel=2)
        return n

    def _is_shared(self, name: str) -> bool:
        if name in ("betas",):
            return True
        if name == "fov":
            return self.fov.numel() in (1, self.views)
        if name in ("log_beta_scales", "betas_trans"):
            return getattr(self, name).shape[0] == 1
        return False

    def boundary_rows(self):
        """(first, last) parameter rows [pose, trans] of this shard for the temporal halo exchange: two rows are sliced, the
        parameter matrix is not touched."""
        pose, trans = self._pose.detach(), self.trans.detach()
        row = lambda i: torch.cat([pose[i].reshape(-1), trans[i].reshape(-1)])  # noqa: E731
        return row(0), row(self.num_images - 1)


class _TemporalTerm(torch.autograd.Function):
    """(joint, global, translation) temporal terms; rows of the pose gradient belong to exactly one of them."""

    @staticmethod
    def forward(ctx, fitter, w_temp, pose, trans):
        objs, grads = fitter._loss_and_grads(None, [0.0] * 6, w_temp, window=None)
        ctx.grads = grads
        return objs[6].clone(), objs[7].clone(), objs[8].clone()

    @staticmethod
    def backward(ctx, g_joint, g_global, g_trans):
        g = ctx.grads
        d_pose = torch.cat([g["pose"][:, :1] * g_global, g["pose"][:, 1:] * g_joint], dim=1)
        return None, None, d_pose, g["trans"] * g_trans
