"""Synthetic fitting problems: seeded poses, a camera ring, and targets rendered by this
library itself from a second pose.  Used by bench.py, ``__graft_entry__.smoke`` and the tests; there is no
dataset on the GPU box.
"""
from __future__ import annotations

from typing import Optional

import torch

from . import engine, model_io
from .cameras import look_at_view_transform
from .config import FitterConfig
from .fitter import SMALFitter, default_global_rotation

STAGE1_WEIGHTS = [10.0, 500.0, 1.0, 1.0, 100.0, 0.1]  # column 1 of OPT_WEIGHTS (reference config.py:64-71)
STAGE1_TEMPORAL = 100.0
STAGE1_LR = 5e-3


def camera_ring(views: int, radius: float, elevation: float = 15.0, device="cpu"):
    """Evenly spaced azimuths, as reference tests/test_triangulation_consistency.py:73-107."""
    az = torch.linspace(0, 360, views + 1)[:views]
    return look_at_view_transform(radius, elevation, az.numpy(), device=device)


def random_pose(n: int, J: int, gen: torch.Generator, amp: float = 0.15):
    pose = amp * torch.randn(n, J, 3, generator=gen)
    pose[:, 0] = torch.from_numpy(default_global_rotation()) + 0.05 * torch.randn(n, 3, generator=gen)
    trans = 0.05 * torch.randn(n, 3, generator=gen)
    return pose, trans


def make_problem(tables: model_io.SmilModelTables, frames: int, views: int, S: int, device, radius: float = 2.7,
                 seed: int = 1234, window: int = 10, frame0: int = 0, n_frames_total: Optional[int] = None,
                 target_chunk: int = 0) -> SMALFitter:
    """A ``SMALFitter`` on ``device`` holding ``frames`` synthetic frames x ``views`` cameras of side ``S``.

    Targets: hard silhouettes (threshold 0.5) and projected joints (+ N(0,1 px)) of a second random pose
    (seed + 10^6); visibility all 1.  The fitted parameters start at the first random pose."""
    dev = torch.device(device)
    cfg = FitterConfig.from_tables(tables, WINDOW_SIZE=window)
    J, nB = tables.J, tables.nB
    # every rank draws the WHOLE sequence from the same seed and keeps its own frames, so that shards of a multi-GPU run
    # are slices of the problem a single rank would hold (a few MB even at 65 k frames)
    total = frames if n_frames_total is None else int(n_frames_total)
    own = slice(frame0, frame0 + frames)
    gen = torch.Generator().manual_seed(seed)
    gen_t = torch.Generator().manual_seed(seed + 10 ** 6)
    pose0, trans0 = (x[own] for x in random_pose(total, J, gen))
    pose1, trans1 = (x[own] for x in random_pose(total, J, gen_t))
    betas0 = 0.5 * torch.randn(nB, generator=gen)
    betas1 = 0.5 * torch.randn(nB, generator=gen_t)
    ls0 = (0.05 * torch.randn(total, J, 3, generator=gen))[own]
    R, T = camera_ring(views, radius, device=dev)
    fov = torch.full((1,), 60.0, device=dev)

    dm = engine.DeviceModel(tables, dev)
    cams = engine.CameraSet(R.contiguous(), T.contiguous(), fov, None, views, S)
    sil = torch.empty(frames * views, 1, S, S, dtype=torch.uint8, device=dev)  # binary masks
    tj = torch.empty(frames * views, J, 2, dtype=torch.float32, device=dev)
    if target_chunk <= 0:  # keep the temporary (chunk*views, S, S) fp32 render around 1 GiB
        target_chunk = max(1, (1 << 28) // (views * S * S))
    for f0 in range(0, frames, target_chunk):
        f1 = min(frames, f0 + target_chunk)
        out = engine.lbs_forward(dm, betas1.to(dev), pose1[f0:f1].to(dev).contiguous(), trans=trans1[f0:f1].to(dev).contiguous(),
                                 shared_beta=True, trans_after_joints=True)
        ndc, _ = engine.project(cams, out["verts"], want_yx=False)
        _, yx = engine.project(cams, out["joints"], want_ndc=False)
        s = engine.silhouette_forward(dm, ndc, S)
        sil[f0 * views:f1 * views, 0] = (s > 0.5).to(torch.uint8)
        tj[f0 * views:f1 * views] = yx
    noise = torch.randn(total * views, J, 2, generator=gen_t)[frame0 * views:(frame0 + frames) * views].to(dev)
    tj = tj + noise
    vis = torch.ones(frames * views, J, dtype=torch.long, device=dev)
    rgb = torch.zeros(frames * views, 3, 1, S)  # placeholder: the fitting path never reads rgb pixels
    rgb = rgb.expand(frames * views, 3, S, S)

    fitter = SMALFitter(dev, (rgb, sil, tj, vis), window, -1, False, tables=tables, config=cfg, views=views, frame0=frame0,
                        n_frames_total=n_frames_total)
    fitter.set_cameras(R, T)
    with torch.no_grad():
        fitter._pose.copy_(pose0.to(dev))
        fitter.trans.copy_(trans0.to(dev))
        fitter.betas.copy_(betas0.to(dev))
        fitter.log_beta_scales.copy_(ls0.to(dev))
    fitter.log_beta_scales.requires_grad = True  # stages >= 1 with ALLOW_LIMB_SCALING (optimize_to_joints.py:142-143)
    return fitter
