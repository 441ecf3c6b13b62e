"""Per-frame betas through ``SMAL.__call__`` (the neural caller of the reference, smal_fitter/neuralSMIL/smil_image_regressor.py:2663): is the
shape blend ``v_shaped = v_template + betas @ shapedirs`` a limiter there, and would MFMA help?  Runs the LBS forward with one beta row per frame
(``shared_beta=False``) for rocprofv3 --kernel-trace --stats, and prints the algorithmic bytes / flops of the blend so that the kernel's duration can
be priced against the HBM and the fp32 MFMA / vector peaks.

    rocprofv3 --kernel-trace --stats -d gpurun_out/shape -o shape -- python3 tools/shape_blend_probe.py --frames 4096
"""
import argparse
import os
import sys
import time

import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from smilify_amd import engine, model_io  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SMILy_STICK")
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--reps", type=int, default=20)
args = ap.parse_args()
dev = torch.device("cuda:0")
t = model_io.load_model(os.path.join(REPO, "data", "models", args.model + ".npz"))
dm = engine.DeviceModel(t, dev)
B, J, V, nB = args.frames, t.J, t.V, t.nB
g = torch.Generator().manual_seed(1)
beta = (0.5 * torch.randn(B, nB, generator=g)).to(dev)
theta = (0.15 * torch.randn(B, J, 3, generator=g)).to(dev)
trans = (0.05 * torch.randn(B, 3, generator=g)).to(dev)
for it in range(args.reps + 2):
    if it == 2:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    out = engine.lbs_forward(dm, beta, theta, trans=trans, shared_beta=False)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
blend_bytes = B * (12 * V + 4 * nB) + 4 * nB * 3 * V  # v_shaped written once per frame, betas read, shapedirs + template read once (cache resident)
blend_flops = 2.0 * B * nB * 3 * V
print(f"{args.model}: B={B} V={V} J={J} nB={nB}  lbs_forward (per-frame betas) {dt * 1e3:.3f} ms per call")
print(f"shape blend: {blend_bytes / 1e6:.1f} MB algorithmic (-> {blend_bytes / 8e12 * 1e6:.1f} us at 8 TB/s), {blend_flops / 1e9:.3f} GFLOP "
      f"(-> {blend_flops / 157e12 * 1e6:.2f} us at the 157 TFLOP/s fp32 vector peak): arithmetic intensity {blend_flops / blend_bytes:.2f} flop/B - HBM-bound by 20 x whatever unit does the FMAs")
