"""Copy the measurement artefacts of tools/measure_round.sh from gpurun_out/ into profiles/ (tracked), named per round.

    python tools/collect_profiles.py r2
"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r3"
src, dst = os.path.join(REPO, "gpurun_out", tag), os.path.join(REPO, "profiles")
os.makedirs(dst, exist_ok=True)
for w in ("cfg2b", "cfg2", "cfg3", "cfg4", "cfg5s"):
    p = os.path.join(src, f"bench_{w}.json")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_bench_{w}.json"))
for w in ("cfg2b", "cfg3"):  # the same workloads under tie_rule = reference_queue (bench.py --tie-rule)
    p = os.path.join(src, f"bench_tie_{w}.json")
    if os.path.exists(p) and os.path.getsize(p):
        shutil.copy(p, os.path.join(dst, f"{tag}_bench_tie_{w}.json"))
for name, out in (("kernel_stats.csv", f"{tag}_kernel_stats.csv"), ("latency.txt", f"{tag}_latency.txt")):
    if os.path.exists(os.path.join(src, name)):
        shutil.copy(os.path.join(src, name), os.path.join(dst, out))
for cfg, n_img in (("cfg2b", 4096), ("cfg3", 4608)):
    base = os.path.join(REPO, "gpurun_out", f"pmc_{cfg}")
    if not os.path.isdir(base):
        continue
    for name in ("fetch", "write", "sq"):
        found = [os.path.join(d, x) for d, _, fs in os.walk(os.path.join(base, name)) for x in fs if x.endswith("counter_collection.csv")]
        f = found[0] if found else ""
        if f:  # keep only the tile kernel's rows (the full file also lists every torch fill kernel)
            rows = [r for r in csv.DictReader(open(f)) if "k_raster" in r["Kernel_Name"]]
            with open(os.path.join(dst, f"{tag}_pmc_{name}_{cfg}.csv"), "w", newline="") as fh:
                w = csv.DictWriter(fh, fieldnames=list(rows[0].keys()))
                w.writeheader()
                w.writerows(rows)
    subprocess.check_call([sys.executable, os.path.join(REPO, "tools", "pmc_summary.py"), cfg, str(n_img), tag], stdout=subprocess.DEVNULL)
# launches of one single-frame iteration
tr = glob.glob(os.path.join(src, "b1", "*kernel_trace.csv"))
if tr:
    rows = sorted(csv.DictReader(open(tr[0])), key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "k_adam_multi" in r["Kernel_Name"]]
    a, b = idx[-2], idx[-1]
    t0 = int(rows[a + 1]["Start_Timestamp"])
    with open(os.path.join(dst, f"{tag}_b1_trace.txt"), "w") as fh:
        fh.write("one fit iteration on ONE frame (STICK @256^2), rocprofv3 --kernel-trace of tools/dbg/b1_trace.py: start, duration, kernel\n")
        for r in rows[a + 1:b + 1]:
            fh.write(f"{(int(r['Start_Timestamp']) - t0) / 1e3:8.1f} us +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f} us  {r['Kernel_Name'][:100]}\n")
        fh.write(f"{b - a} launches; {(int(rows[b]['End_Timestamp']) - int(rows[a]['End_Timestamp'])) / 1e3:.1f} us from optimiser step to optimiser step (under the profiler)\n")
print(sorted(os.path.basename(p) for p in glob.glob(os.path.join(dst, f"{tag}_*"))))
for w in ("cfg2b", "cfg2", "cfg3", "cfg4", "cfg5s"):
    p = os.path.join(dst, f"{tag}_bench_{w}.json")
    if os.path.exists(p):
        d = json.load(open(p))
        print(w, f"{d['ms_per_step']:.2f} ms/step  {d['value']:.2f} fit-iters/s ({d.get('frame_iters_per_sec', 0):.0f} frame-iters/s)  kernel {d['roofline']['kernel_ms']:.2f} ms  frac {d['roofline']['frac']:.4f}")
