"""Drop-in ``SMAL`` module: same constructor, attributes and ``__call__`` contract as the reference's
``smal_model.smal_torch.SMAL`` (reference smal_model/smal_torch.py:88-370), computed by the HIP library.

Differences that are deliberate and documented:
* the model path comes from ``model_path=`` / ``tables=`` / ``smilify_amd.config.current.SMAL_FILE``
  (the reference reads the global ``config.SMAL_FILE`` at construction, :92);
* dense ``weights`` / ``J_regressor`` / ``posedirs`` attributes exist for callers that read them, but the
  kernels use the compact tables of ``model_io``;
* gradients flow to ``beta, theta (axis-angle or (B,J,3,3) matrices), trans, del_v, betas_logscale, betas_trans,
  v_template`` through all four returned tensors (``verts``, ``joints``, ``Rs``, ``v_shaped``; reference :367-370);
* inputs with a leading dimension of 1 are broadcast over the batch like the torch expressions of the reference do;
  any other shape mismatch raises instead of reading past a buffer;
* pose blend shapes (legacy SMAL ``posedirs``) are applied when the model has a non-zero table; every SMIL
  model ships an empty one (reference :182-190) and skips that product entirely.
"""
from __future__ import annotations

from typing import Optional

import numpy as np
import torch
import torch.nn as nn

from . import config as _config
from . import engine, model_io


class _LbsFunction(torch.autograd.Function):
    _OUTPUTS = ("verts", "joints", "Rs", "v_shaped", "new_J")

    @staticmethod
    def forward(ctx, dm, flags, beta, theta, trans, logscale, btrans, del_v, v_template):
        dev = dm.device
        ctx.set_materialize_grads(False)  # a returned tensor the caller's loss does not touch arrives as None, not as zeros
        c = lambda t: None if t is None else t.detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        theta_c = c(theta)
        rot_in = theta_c is not None and theta_c.dim() == 4
        out = engine.lbs_forward(
            dm, c(beta), None if rot_in else theta_c, trans=c(trans), logscale=c(logscale), btrans=c(btrans), del_v=c(del_v),
            v_template=c(v_template), Rs_in=theta_c if rot_in else None, logscale_shared=flags["logscale_shared"],
            btrans_shared=flags["btrans_shared"], propagate_scaling=flags["propagate_scaling"],
            allow_limb_scaling=flags["allow_limb_scaling"])
        # The five RETURNED tensors go through save_for_backward: kept as plain attributes they close a cycle (output -> grad_fn -> ctx ->
        # output) that crosses into C++ where Python's collector cannot follow it, and every call's (B,V,3) outputs stayed allocated
        # for the life of the process (round 6: 1 GiB per call with the mouse at 4 096 frames).  Everything else is an intermediate no
        # output refers to.
        ctx.dm = dm
        ctx.rest = {k: v for k, v in out.items() if k not in _LbsFunction._OUTPUTS}
        ctx.save_for_backward(*(out[k] for k in _LbsFunction._OUTPUTS))
        ctx.has = (beta is not None and beta.shape[-1] > 0, not rot_in, trans is not None, logscale is not None, btrans is not None)
        ctx.rot_in = rot_in
        ctx.shared = (flags["logscale_shared"], flags["btrans_shared"])
        v_shaped = out["v_shaped"]
        ctx.mark_non_differentiable(out["new_J"])
        return out["verts"], out["joints"], out["Rs"], v_shaped, out["new_J"]

    @staticmethod
    def backward(ctx, d_verts, d_joints, d_Rs, d_vs, _dnj):
        need = ctx.needs_input_grad  # (dm, flags, beta, theta, trans, logscale, btrans, del_v, v_template)
        cont = lambda t: None if t is None else t.contiguous()  # noqa: E731
        dv, dj, dR, dvs = cont(d_verts), cont(d_joints), cont(d_Rs), cont(d_vs)
        if dv is None and dj is None and dR is None and dvs is None:
            return (None,) * 9
        saved = dict(ctx.rest)
        saved.update(zip(_LbsFunction._OUTPUTS, ctx.saved_tensors))
        g = engine.lbs_backward(ctx.dm, saved, dv, dj, need_beta=need[2] and ctx.has[0], need_theta=need[3] and ctx.has[1],
                                need_logscale=need[5] and ctx.has[3], need_btrans=need[6] and ctx.has[4],
                                need_trans=need[4] and ctx.has[2], need_vshaped=need[7] or need[8], need_Rs=need[3] and ctx.rot_in,
                                up_Rs=dR, up_v_shaped=dvs)
        d_theta = g["d_Rs_in"] if ctx.rot_in else g["d_theta"]
        d_ls = g["d_logscale"] if g["d_logscale"] is None or not ctx.shared[0] else g["d_logscale"][None]
        d_bt = g["d_btrans"] if g["d_btrans"] is None or not ctx.shared[1] else g["d_btrans"][None]
        d_delv = g["d_del_v"] if need[7] else None
        d_vt = g["d_del_v"].sum(0) if need[8] else None  # a custom template is one (V,3) table shared by the batch
        return None, None, g["d_beta"], d_theta, g["d_trans"], d_ls, d_bt, d_delv, d_vt


_HARDCODED_BODY_VERTS = (1863, 26, 2124, 150, 3055, 1097)  # end of nose, chin, right / left ear tip, left / right eye


class SMAL(nn.Module):
    def __init__(self, device, shape_family_id=-1, dtype=torch.float, model_path: Optional[str] = None,
                 tables: Optional[model_io.SmilModelTables] = None, config: Optional[_config.FitterConfig] = None):
        super().__init__()
        if shape_family_id != -1:
            raise NotImplementedError("shape families need the MPI-licensed SMAL data file, which SMIL models do not use")
        if dtype not in (torch.float, torch.float32):
            raise NotImplementedError("the HIP path computes in fp32")
        if tables is None:
            cfg0 = config or _config.current
            path = model_path or (cfg0.SMAL_FILE if cfg0 is not None else None)
            if path is None:
                raise ValueError("SMAL needs model_path=, tables= or smilify_amd.config.current.SMAL_FILE")
            tables = model_io.load_model(path)
        self.tables = tables
        self.config = config or _config.current or _config.FitterConfig.from_tables(tables, model_path)
        self.device = engine.require_gpu(device)
        self._dm = engine.DeviceModel(tables, self.device)
        t = tables
        dev = self.device
        # attributes the reference exposes (plain tensors, not registered buffers: smal_torch.py:104-196)
        self.f = t.faces
        self.faces = torch.from_numpy(t.faces.astype(np.int64)).to(dev)
        self.size = [t.V, 3]
        self.num_betas = t.nB
        self.v_template = torch.from_numpy(t.v_template).to(dev)
        self.shapedirs = torch.from_numpy(t.shapedirs).to(dev)
        self.parents = t.parents.copy()
        self.left_inds = self.right_inds = self.center_inds = np.array([])
        if t.static_joints:
            self.J = torch.from_numpy(t.J_static).to(dev)
        self.J_transformed = None
        self._dense = {}

    # dense views are built lazily: nothing on the hot path reads them
    def _lazy(self, name, builder):
        if name not in self._dense:
            self._dense[name] = torch.from_numpy(builder()).to(self.device)
        return self._dense[name]

    @property
    def weights(self):
        return self._lazy("weights", self.tables.dense_weights)

    @property
    def J_regressor(self):
        return self._lazy("J_regressor", self.tables.dense_J_regressor)

    @property
    def posedirs(self):
        t = self.tables
        return self._lazy("posedirs", lambda: t.posedirs if t.posedirs is not None else np.zeros(((t.J - 1) * 9, t.V * 3), np.float32))

    @property
    def device_model(self) -> engine.DeviceModel:
        return self._dm

    def __call__(self, beta, theta, trans=None, del_v=None, betas_logscale=None, betas_trans=None, get_skin=True,
                 v_template=None, propagate_scaling=False):
        J, V = self.tables.J, self.tables.V
        if theta.shape[1] != J:  # reference :282-287: wrong joint count -> zero pose
            theta = torch.zeros(beta.shape[0] if beta.shape[1] > 0 else 1, J, 3, device=self.device)
        if tuple(theta.shape[1:]) not in ((J, 3), (J, 3, 3)):
            raise ValueError(f"theta must be (B,{J},3) axis-angle or (B,{J},3,3) rotation matrices, got {tuple(theta.shape)}")
        B = theta.shape[0]
        if beta.dim() != 2 or beta.shape[0] not in (1, B):
            raise ValueError(f"beta must be (B,nB) or (1,nB) with B={B}, got {tuple(beta.shape)}")
        if beta.shape[0] != B:
            beta = beta.expand(B, -1)
        if beta.shape[1] > self.tables.nB:
            raise ValueError(f"beta has {beta.shape[1]} columns but the model has {self.tables.nB} shape directions")

        def rows(name, t, tail):
            """(tensor laid out as the kernels read it, shared?) for an optional input that is (B,*tail) or broadcasts
            from (1,*tail) / (*tail) like the torch expressions of the reference (smal_torch.py:244-248,320-340,
            batch_lbs.py:131-160); anything else would make a kernel read past the buffer, so it raises."""
            if t is None:
                return None, False
            if tuple(t.shape) == tail:
                t = t[None]
            if t.dim() != len(tail) + 1 or tuple(t.shape[1:]) != tail or t.shape[0] not in (1, B):
                raise ValueError(f"{name} must be (B,{','.join(map(str, tail))}) or broadcast from one row, with B={B}; got {tuple(t.shape)}")
            return t, (t.shape[0] == 1 and B > 1)

        trans, tr_shared = rows("trans", trans, (3,))
        if tr_shared:
            trans = trans.expand(B, 3)
        del_v, dv_shared = rows("del_v", del_v, (V, 3))
        if dv_shared:
            del_v = del_v.expand(B, V, 3)
        betas_logscale, ls_shared = rows("betas_logscale", betas_logscale, (J, 3))
        betas_trans, bt_shared = rows("betas_trans", betas_trans, (J, 3))
        if v_template is not None and tuple(v_template.shape) != (V, 3):
            raise ValueError(f"v_template must be ({V},3), got {tuple(v_template.shape)}")
        flags = dict(propagate_scaling=bool(propagate_scaling), allow_limb_scaling=bool(self.config.ALLOW_LIMB_SCALING),
                     logscale_shared=ls_shared, btrans_shared=bt_shared)
        verts, joints, Rs, v_shaped, new_J = _LbsFunction.apply(self._dm, flags, beta, theta, trans, betas_logscale, betas_trans,
                                                                del_v, v_template)
        self.J_transformed = new_J
        if J == 35 and not self.config.ignore_hardcoded_body:
            # legacy SMAL / WLDO body: six vertices (nose, chin, ear tips, eyes) ride along as extra joints
            # (reference smal_torch.py:353-365; never taken by SMIL models, whose config sets ignore_hardcoded_body)
            if V <= max(_HARDCODED_BODY_VERTS):
                raise ValueError(f"the hard-coded body vertices need a mesh with more than {max(_HARDCODED_BODY_VERTS)} vertices, this one has {V}")
            joints = torch.cat([joints, verts[:, list(_HARDCODED_BODY_VERTS)]], dim=1)
        if get_skin:
            return verts, joints, Rs, v_shaped
        return joints
