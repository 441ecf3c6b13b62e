"""Stage / epoch driver of the fit, single- or multi-GPU.

Mirrors the loop of reference smal_fitter/optimize_to_joints.py:110-178 (4 stages from ``OPT_WEIGHTS``, a
fresh Adam(betas=(0.5,0.999)) per stage with ``fov`` in its own lr=1 group, stage 0 freezing
joint rotations / betas / limb scales and masking visibility to the torso joints, every epoch = sum over
windows of the window mean + temporal terms, one backward, one step) on top of ``SMALFitter.fit_step``.
Image / mesh export of the reference driver is out of scope.

Multi-GPU (one process per GPU, ``torch.distributed`` over RCCL): frames are split into contiguous shards
aligned to ``WINDOW_SIZE`` so the sum over windows is a plain sum over ranks.  Per epoch there are two tiny
exchanges: each shard's first / last parameter row goes to its neighbour ranks (temporal halo, (3J+3) floats each way,
point to point) and ONE in-place all-reduce(SUM) of the fitter's shared block (10 loss terms + shared-parameter
gradients), which runs beside the Adam update of the per-frame parameters.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Callable, Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

try:
    import torch.distributed as dist
except Exception:  # pragma: no cover
    dist = None


@dataclass
class ShardPlan:
    """Contiguous frame range of one rank, aligned to loss windows."""

    rank: int
    world: int
    n_total: int
    window: int
    start: int
    stop: int

    @property
    def n_local(self) -> int:
        return self.stop - self.start


def plan_shards(n_total: int, world: int, window: int) -> List[ShardPlan]:
    """Split ``n_total`` frames into ``world`` contiguous shards whose boundaries fall on window boundaries
    (so that no loss window spans two ranks); earlier ranks take the extra windows."""
    window = max(1, min(window, n_total))
    n_win = (n_total + window - 1) // window
    if n_win < world:
        raise ValueError(f"{n_win} windows cannot be spread over {world} ranks; lower WINDOW_SIZE or use fewer GPUs")
    base, extra = divmod(n_win, world)
    plans, w0 = [], 0
    for r in range(world):
        w1 = w0 + base + (1 if r < extra else 0)
        plans.append(ShardPlan(r, world, n_total, window, w0 * window, min(n_total, w1 * window)))
        w0 = w1
    return plans


class PendingHalo:
    """Temporal-halo rows on their way: ``wait()`` returns (row before my first frame, row after my last frame) on the device.
    Posting (``post_halos``) and waiting are separate so that the skinning and rasteriser kernels of an iteration run while the
    two tiny messages are in flight - only ``fit_epilogue`` (the temporal terms) reads the rows."""

    def __init__(self, reqs, prev_row, next_row, dev, keep=()):
        self._reqs, self._prev, self._next, self._dev, self._keep = reqs, prev_row, next_row, dev, keep
        self._done = None

    def wait(self):
        if self._done is None:
            for req in self._reqs:
                req.wait()  # (RCCL: orders the current stream behind the receive; gloo: blocks the host)
            back = lambda t: None if t is None else t.to(self._dev)  # noqa: E731
            self._done = (back(self._prev), back(self._next))
            self._reqs = self._keep = ()
        return self._done


def post_halos(first_row: torch.Tensor, last_row: torch.Tensor, rank: int, world: int, group=None, host_staged: bool = False,
               recv_prev: Optional[torch.Tensor] = None, recv_next: Optional[torch.Tensor] = None) -> PendingHalo:
    """Temporal halo: every shard sends its first parameter row to the rank before it and its last row to
    the rank after it - point to point, one batch of non-blocking sends / receives per rank, returned as a ``PendingHalo``.
    ``host_staged`` moves the rows through host memory (gloo rehearsals); RCCL runs keep them on the device.  ``recv_prev`` /
    ``recv_next``: receive straight into these (persistent) device buffers - what a captured iteration reads."""
    dev = first_row.device
    if world == 1:
        return PendingHalo((), None, None, dev)
    stage = (lambda t: t.detach().cpu().contiguous()) if host_staged else (lambda t: t.detach().contiguous())
    first, last = stage(first_row), stage(last_row)
    into = lambda buf: torch.empty_like(first) if (buf is None or host_staged) else buf  # noqa: E731
    prev_row = into(recv_prev) if rank > 0 else None
    next_row = into(recv_next) if rank + 1 < world else None
    ops = []
    if rank > 0:
        ops += [dist.P2POp(dist.isend, first, rank - 1, group), dist.P2POp(dist.irecv, prev_row, rank - 1, group)]
    if rank + 1 < world:
        ops += [dist.P2POp(dist.isend, last, rank + 1, group), dist.P2POp(dist.irecv, next_row, rank + 1, group)]
    return PendingHalo(dist.batch_isend_irecv(ops), prev_row, next_row, dev, keep=(first, last))


def exchange_halos(first_row: torch.Tensor, last_row: torch.Tensor, rank: int, world: int, group=None,
                   host_staged: bool = False) -> Tuple[Optional[torch.Tensor], Optional[torch.Tensor]]:
    """``post_halos`` + ``wait()``: returns (row before my first frame, row after my last frame)."""
    return post_halos(first_row, last_row, rank, world, group, host_staged).wait()


class _Done:
    def wait(self):
        return None


def allreduce_block(flat: torch.Tensor, group=None, host_staged: bool = False, async_op: bool = True):
    """All-reduce(SUM) ONE contiguous tensor in place - the fitter's shared block ``[10 loss terms | d_betas | d_fov | shared
    scale-table gradients]``, which the kernels already wrote next to each other (no concatenation, no copy back).
    Returns a handle whose ``wait()`` orders the result before what the current stream does next; with ``async_op`` the
    collective runs beside whatever is launched in between (the per-frame Adam update)."""
    if host_staged:
        host = flat.cpu()
        dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
        flat.copy_(host)
        return _Done()
    work = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group, async_op=async_op)
    return work if work is not None else _Done()


def allreduce_shared(shared: Dict[str, torch.Tensor], objs: torch.Tensor, group=None, host_staged: bool = False) -> None:
    """All-reduce(SUM) of separately allocated tensors [shared-parameter gradients..., loss terms] through one fused buffer;
    results written back in place.  (General form; ``SMALFitter.fit_step`` reduces its shared block in place with
    ``allreduce_block``.)"""
    names = sorted(shared)
    flat = torch.cat([shared[k].reshape(-1) for k in names] + [objs.reshape(-1)])
    allreduce_block(flat, group, host_staged, async_op=False).wait()
    o = 0
    for k in names:
        n = shared[k].numel()
        shared[k].copy_(flat[o:o + n].reshape(shared[k].shape))
        o += n
    objs.copy_(flat[o:o + objs.numel()])


@dataclass
class StageSpec:
    weights: Sequence[float]  # w_j2d, w_reproj, w_betas, w_pose, w_limit, w_splay
    w_temp: float
    epochs: int
    lr: float


def stages_from_config(cfg) -> List[StageSpec]:
    tab = np.array(cfg.OPT_WEIGHTS, dtype=np.float64).T
    return [StageSpec(list(row[:6]), float(row[6]), int(row[7]), float(row[8])) for row in tab]


def configure_stage(fitter, stage_id: int, full_visibility: torch.Tensor) -> None:
    """Parameter freezing and visibility masking of optimize_to_joints.py:129-145."""
    cfg = fitter.config
    if stage_id == 0:
        fitter.joint_rotations.requires_grad = False
        fitter.betas.requires_grad = False
        fitter.log_beta_scales.requires_grad = False
        fitter.fov.requires_grad = True
        vis = torch.zeros_like(full_visibility)
        vis[:, cfg.TORSO_JOINTS] = full_visibility[:, cfg.TORSO_JOINTS]
        fitter.target_visibility = vis
    else:
        fitter.joint_rotations.requires_grad = True
        fitter.betas.requires_grad = True
        fitter.fov.requires_grad = True
        if cfg.ALLOW_LIMB_SCALING:
            fitter.log_beta_scales.requires_grad = True
        fitter.target_visibility = full_visibility.clone()


def optimize(fitter, stages: Optional[List[StageSpec]] = None, rank: int = 0, world: int = 1, group=None,
             on_epoch: Optional[Callable[[int, int, torch.Tensor], None]] = None, max_epochs: Optional[int] = None,
             use_graph: bool = True, host_staged: bool = False):
    """Run the staged optimisation on this rank's shard; returns the per-epoch loss terms of the last stage.

    Single rank: every stage captures its iteration (losses, backward, Adam) once in a hipGraph and replays it per epoch
    (``SMALFitter.fit_step_graph``) - the device-side schedule of SURVEY.md 8(f) row 4; ``use_graph=False`` launches
    the kernels one by one instead (identical results).  Several ranks: the shared-parameter gradients pass through the all-reduce
    between backward and the optimiser step, so the iteration is TWO graphs with the collective between them
    (``SMALFitter.fit_step_graph_ranks``: losses + backward | all-reduce | Adam); ``use_graph=False`` is the eager step, which posts
    the temporal halo first and waits for it only in front of the kernel that reads it.  ``host_staged``: collectives through host
    memory (gloo rehearsals)."""
    cfg = fitter.config
    stages = stages or stages_from_config(cfg)
    full_vis = fitter.target_visibility.clone()
    hook = None
    if world > 1:
        hook = lambda block: allreduce_block(block, group, host_staged)  # noqa: E731
    graph = use_graph and world == 1
    history = []
    for stage_id, st in enumerate(stages):
        configure_stage(fitter, stage_id, full_vis)
        fitter.begin_stage(st.lr, fov_lr=1.0)
        epochs = st.epochs if max_epochs is None else min(st.epochs, max_epochs)
        history = []
        for epoch in range(epochs):
            if graph:
                objs = fitter.fit_step_graph(st.weights, st.w_temp, window=cfg.WINDOW_SIZE).clone()
            elif use_graph:  # several ranks: the iteration as two graphs around the collective
                objs = fitter.fit_step_graph_ranks(st.weights, st.w_temp, cfg.WINDOW_SIZE, rank, world, group, hook,
                                                   host_staged=host_staged).clone()
            else:
                halo = None
                if world > 1:  # posted now, waited for right before the one kernel that reads the rows
                    first, last = fitter.boundary_rows()
                    halo = post_halos(first, last, rank, world, group, host_staged=host_staged)
                objs = fitter.fit_step(st.weights, st.w_temp, window=cfg.WINDOW_SIZE, halo=halo, shared_grad_hook=hook)
            history.append(objs)
            if on_epoch is not None:
                on_epoch(stage_id, epoch, objs)
        fitter.straddling_faces()  # once per stage (it synchronises): warns when the mesh has reached the camera's clipping plane
    return history
