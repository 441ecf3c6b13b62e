"""Run only the fused rasteriser on a bench workload (for rocprofv3 counter passes) and print work statistics."""
import argparse
import os
import sys
import time

import numpy as np
import torch

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO)
from smilify_amd import engine, model_io, synthetic  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--model", default="SMILy_STICK")
ap.add_argument("--frames", type=int, default=512)
ap.add_argument("--views", type=int, default=1)
ap.add_argument("--S", type=int, default=256)
ap.add_argument("--radius", type=float, default=2.7)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--mode", default="fused", choices=["fused", "fwd"])
ap.add_argument("--quick", action="store_true", help="timing only (skip the per-tile list statistics)")
ap.add_argument("--tie-rule", default="depth_face_id", choices=["depth_face_id", "reference_queue"])
args = ap.parse_args()
dev = torch.device("cuda:0")
tables = model_io.load_model(os.path.join(REPO, "data", "models", args.model + ".npz"))
f = synthetic.make_problem(tables, args.frames, args.views, args.S, dev, radius=args.radius)
f._refresh_targets()
dm = f.device_model
lbs = engine.lbs_forward(dm, f.betas.detach(), f._pose, trans=f.trans.detach().contiguous(), shared_beta=True, trans_after_joints=True)
cam = f.renderer.cameras
cams = engine.CameraSet(cam.R.contiguous(), cam.T.contiguous(), f.fov.detach(), None, args.views, args.S)
ndc, _ = engine.project(cams, lbs["verts"], want_yx=False)
N = ndc.shape[0]
cfg = engine.fit_config(args.frames, dm.J, dm.nB, 10, synthetic.STAGE1_WEIGHTS)
ps = engine.pix_scale(cfg, args.views, args.S, dev)
rs = engine.raster_settings(tie_rule=args.tie_rule)
for it in range(args.reps + 1):
    if it == 1:
        torch.cuda.synchronize(); t0 = time.perf_counter()
    if args.mode == "fused":
        engine.silhouette_l1_fused(dm, ndc, args.S, f._sil_dev, f._sil_sum, ps, rs)
    else:
        engine.silhouette_forward(dm, ndc, args.S, rs)
torch.cuda.synchronize()
dt = (time.perf_counter() - t0) / args.reps
if args.quick:
    if args.tie_rule != "depth_face_id":
        st = engine.raster_stats(dm, N)
        print(f"tie_rule {args.tie_rule}: {st['tie_pixels']} pixels replayed ({st['tie_pixels'] / N:.1f} per image, {st['tie_pixels'] / max(st['tiles'], 1):.2f} per touched tile)")
    print(f"images {N}  time/launch {dt*1e3:.3f} ms  {dt/N*1e6:.2f} us/image  [SMIL_RESIDENT={os.environ.get('SMIL_RESIDENT')} SMIL_WRAP={os.environ.get('SMIL_WRAP')}]")
    sys.exit(0)
ws = dm._ws
FT = (dm.F + 63) // 64 * 64 + 2048  # rows of the per-image face tables (raster.hip: faces_padded(F) + CLIP_FX)
off = ((N * FT * 4 + 255) // 256) * 256
ctr = ws[off:off + 128].view(torch.int32).cpu().numpy().reshape(8, 4).sum(0)  # (partition, cost class) counters
n_work = int(ctr[:4].sum())
tb = ws[: N * FT * 4].view(torch.int32).reshape(N, FT).cpu().numpy().astype(np.uint32)
tx0, ty0, tx1, ty1 = tb & 255, (tb >> 8) & 255, (tb >> 16) & 255, tb >> 24
valid = tx0 <= tx1
tiles_per_face = np.where(valid, (tx1.astype(int) - tx0 + 1) * (ty1.astype(int) - ty0 + 1), 0)
print(f"work items per cost class (pairs >= 65536 / 16384 / 4096 / rest): {ctr[:4].tolist()}")
print(f"images {N}  time/launch {dt*1e3:.2f} ms  {dt/N*1e6:.1f} us/image  work items {n_work} ({n_work/N:.1f} tiles/image)  "
      f"valid faces/image {valid.sum(1).mean():.0f}  (tile,face) pairs/image {tiles_per_face.sum(1).mean():.0f}  "
      f"avg list length {tiles_per_face.sum()/max(n_work,1):.0f}")
# distribution of per-tile list lengths (faces whose tile box contains the tile)
T = (args.S + 7) // 8
import collections
lens = []
for n in range(N):
    cnt = np.zeros((T + 1, T + 1), np.int64)
    v = valid[n]
    np.add.at(cnt, (ty0[n][v], tx0[n][v]), 1)
    np.add.at(cnt, (ty1[n][v].astype(int) + 1, tx0[n][v]), -1)
    np.add.at(cnt, (ty0[n][v], tx1[n][v].astype(int) + 1), -1)
    np.add.at(cnt, (ty1[n][v].astype(int) + 1, tx1[n][v].astype(int) + 1), 1)
    c = cnt.cumsum(0).cumsum(1)[:T, :T]
    lens.append(c[c > 0])
lens = np.concatenate(lens)
print("tiles", len(lens), "list length percentiles 50/90/99/99.9/max:", [int(np.percentile(lens, q)) for q in (50, 90, 99, 99.9)], int(lens.max()),
      " >1024:", int((lens > 1024).sum()), " >2048:", int((lens > 2048).sum()), " >4096:", int((lens > 4096).sum()))
