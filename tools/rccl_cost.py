"""One-rank RCCL cost of the per-iteration collective of the sharded fit (the box has one GPU): the in-place asynchronous
all-reduce of the shared block (10 loss terms + d_betas + d_fov + shared tables, <= 350 floats = 1.4 KB) as
optimize.allreduce_block issues it, host time per call and device time per call (events around 200 back-to-back calls).
The N > 1 term this cannot measure is the xGMI hop itself; what it does measure is the fixed software cost the 8-GPU
budget of SURVEY.md 8(e) (20 - 40 us) has to absorb.  python tools/rccl_cost.py [port]"""
import os, sys, time
import torch, torch.distributed as dist
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
torch.cuda.set_device(0)
dev = torch.device("cuda", 0)
port = int(sys.argv[1]) if len(sys.argv) > 1 else 29655
dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=dev)
from smilify_amd import optimize
for n in (16, 350, 4096):
    block = torch.arange(float(n), device=dev)
    for _ in range(20):
        optimize.allreduce_block(block).wait()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter(); e0.record()
    for _ in range(200):
        h = optimize.allreduce_block(block)
        h.wait()
    e1.record(); torch.cuda.synchronize()
    host = (time.perf_counter() - t0) / 200 * 1e6
    print(f"all-reduce of {n:5d} floats, one rank, in place, async + wait: {host:7.1f} us per call on the host, {e0.elapsed_time(e1) / 200 * 1e3:7.1f} us per call on the device stream")
first, last = torch.ones(168, device=dev), torch.ones(168, device=dev)
t0 = time.perf_counter()
for _ in range(200):
    p = optimize.post_halos(first, last, 0, 1)
    p.wait()
print(f"halo post + wait with no neighbour (one rank): {(time.perf_counter() - t0) / 200 * 1e6:.1f} us per call on the host")
dist.barrier(); torch.cuda.synchronize(); dist.destroy_process_group()
