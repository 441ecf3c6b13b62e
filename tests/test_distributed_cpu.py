"""Multi-rank path on CPU: shard planning, the point-to-point temporal halo and the in-place all-reduce of the shared block,
exercised with world_size 2 and 3 over gloo (the same functions run over RCCL on the GPUs)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from smilify_amd import optimize


def test_plan_shards_window_aligned():
    for n_total, world, window in [(4096, 8, 10), (65536, 8, 10), (25, 2, 10), (100, 3, 7), (16, 4, 4)]:
        plans = optimize.plan_shards(n_total, world, window)
        assert plans[0].start == 0 and plans[-1].stop == n_total
        for a, b in zip(plans[:-1], plans[1:]):
            assert a.stop == b.start and a.stop % window == 0 and a.n_local > 0
        sizes = [p.n_local for p in plans]
        assert max(sizes) - min(sizes) <= 2 * window
    with pytest.raises(ValueError):
        optimize.plan_shards(10, 4, 10)


def test_stage_table_matches_reference_config():
    from smilify_amd.config import FitterConfig

    st = optimize.stages_from_config(FitterConfig())
    assert [s.epochs for s in st] == [600, 400, 600, 600]
    assert st[1].weights == [10.0, 500.0, 1.0, 1.0, 100.0, 0.1] and st[1].w_temp == 100.0 and st[1].lr == 5e-3
    assert st[0].weights[0] == 25.0 and st[0].w_temp == 500.0 and st[0].lr == 9e-2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_total, window, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    plan = optimize.plan_shards(n_total, world, window)[rank]
    E = 3 * 5 + 3
    g = torch.Generator().manual_seed(0)
    rows_all = torch.randn(n_total, E, generator=g)  # every rank can regenerate the whole sequence
    mine = rows_all[plan.start:plan.stop]
    pending = optimize.post_halos(mine[0].clone(), mine[-1].clone(), rank, world)  # posted first ...
    work_meanwhile = (mine * 2).sum()  # (stands for the skinning / rasteriser kernels that run while the rows travel)
    prev_row, next_row = pending.wait()  # ... waited for in front of the one consumer
    assert pending.wait() == (prev_row, next_row) and torch.isfinite(work_meanwhile)
    again = optimize.exchange_halos(mine[0].clone(), mine[-1].clone(), rank, world)  # (the blocking form: same rows)
    for a_, b_ in zip(again, (prev_row, next_row)):
        assert (a_ is None and b_ is None) or torch.equal(a_, b_)
    if plan.start > 0:
        assert torch.equal(prev_row, rows_all[plan.start - 1])
    else:
        assert prev_row is None
    if plan.stop < n_total:
        assert torch.equal(next_row, rows_all[plan.stop])
    else:
        assert next_row is None
    # "gradient" of a shared parameter = sum over my frames; loss terms likewise
    shared = {"betas": mine[:, :3].sum(0).clone(), "fov": mine[:, 3:4].sum(0).clone()}
    objs = torch.zeros(10)
    objs[0] = mine.abs().sum()
    # temporal term: pair (i, i+1) is owned by the rank holding i
    nxt = torch.cat([mine[1:], next_row[None]]) if next_row is not None else mine[1:]
    objs[6] = ((mine[: nxt.shape[0]] - nxt) ** 2).sum()
    # the fitter's layout: ONE contiguous block [loss terms | d_betas | d_fov], summed in place while other work proceeds
    block = torch.cat([objs, shared["betas"], shared["fov"]])
    handle = optimize.allreduce_block(block)
    busy = mine.sum()  # (anything: stands for the per-frame Adam update that runs beside the collective)
    handle.wait()
    optimize.allreduce_shared(shared, objs)  # the general form over separate tensors gives the same sums
    assert torch.allclose(block, torch.cat([objs, shared["betas"], shared["fov"]]), rtol=1e-6, atol=1e-6) and torch.isfinite(busy)
    np.save(os.path.join(out_dir, f"r{rank}.npy"), torch.cat([block[10:13], block[13:14], block[:10]]).numpy())
    dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_halo_exchange_and_shared_allreduce_gloo(tmp_path, world):
    """world = 8 is the node size the driver's scaling run uses: eight processes, the chain of seven halo pairs, one block sum."""
    n_total, window = (50, 10) if world < 8 else (170, 10)  # (170 frames over 8 ranks: uneven, window-aligned shards)
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n_total, window, str(tmp_path)), nprocs=world, join=True)
    g = torch.Generator().manual_seed(0)
    rows = torch.randn(n_total, 18, generator=g)
    want = torch.cat([rows[:, :3].sum(0), rows[:, 3:4].sum(0), torch.tensor([rows.abs().sum().item()]), torch.zeros(5),
                      torch.tensor([((rows[:-1] - rows[1:]) ** 2).sum().item()]), torch.zeros(3)]).numpy()
    for r in range(world):
        got = np.load(os.path.join(str(tmp_path), f"r{r}.npy"))
        np.testing.assert_allclose(got, want, rtol=1e-5, atol=1e-5)
